#!/bin/bash
# A/B builds of the one-pass cfg-4 kernel: tools/fu_variants.sh TAG "-DFLAG ..."  ->  finmlkit_amd/lib/ab/libfmk_hip_TAG.so
# (fmk_barflow.hip recompiled with the flags, every other object of the current build reused)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; flags=$2
mkdir -p "$ROOT/finmlkit_amd/lib/ab"
cd "$ROOT/finmlkit_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c fmk_barflow.hip -o /tmp/fmk_barflow_$tag.o
objs=$(ls ../lib/obj/*.o | grep -v "fmk_barflow.o\|fmk_diag.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libfmk_hip_$tag.so $objs /tmp/fmk_barflow_$tag.o
echo "built lib/ab/libfmk_hip_$tag.so ($flags)"
