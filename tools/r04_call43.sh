#!/bin/bash
mkdir -p gpurun_out/c43
for m in slabs offsets shift; do
  timeout 400 python tools/placeexp.py 1e9 $m > gpurun_out/c43/$m.txt 2>&1
  tail -40 gpurun_out/c43/$m.txt
done
