#!/bin/bash
# A/B build of ONE source file of the library: tools/ab_variant.sh TAG FILE.hip "-DFLAG ..."  ->  finmlkit_amd/lib/ab/libfmk_hip_TAG.so
# (FILE recompiled with the flags, every other object of the current build reused; run a script against it with tools/ab_lib.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; file=$2; flags=$3
mkdir -p "$ROOT/finmlkit_amd/lib/ab"
cd "$ROOT/finmlkit_amd/csrc"
base=$(basename "$file" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c "$base.hip" -o "/tmp/${base}_$tag.o"
objs=$(ls ../lib/obj/*.o | grep -v "/$base.o\|fmk_diag.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../lib/ab/libfmk_hip_$tag.so" $objs "/tmp/${base}_$tag.o"
echo "built lib/ab/libfmk_hip_$tag.so ($base.hip $flags)"
