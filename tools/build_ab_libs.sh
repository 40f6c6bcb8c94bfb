#!/bin/bash
# Builds the end-of-round libraries of rounds 1-3 (sources of 2349744^, 80570ca^, bee7b92^) into finmlkit_amd/lib/ab/ for
# tools/drift_ab.sh.  Sources are taken from THIS repository's history (git archive), nothing is kept but the .so files.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/finmlkit_amd/lib/ab"
for pair in r1:2349744^ r2:80570ca^ r3:bee7b92^; do
  tag=${pair%%:*}; rev=${pair#*:}
  out="$ROOT/finmlkit_amd/lib/ab/libfmk_hip_$tag.so"
  [ -f "$out" ] && continue
  tmp=$(mktemp -d /tmp/fmk_ab_XXXX)
  git -C "$ROOT" archive "$rev" finmlkit_amd/csrc include | tar -x -C "$tmp"
  make -s -j8 -C "$tmp/finmlkit_amd/csrc" >/dev/null 2>&1
  cp "$tmp/finmlkit_amd/lib/libfmk_hip.so" "$out"
  rm -rf "$tmp"
  echo "built $out ($rev)"
done
