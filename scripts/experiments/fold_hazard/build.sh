#!/bin/bash
# Builds the two experiment libraries (SLP vectorisation on / off) next to this script and prints which variant shows which
# ISA pattern (isa_patterns.py).  Runs in the build container (hipcc cross-compiles gfx950 without a GPU).
set -e
D=$(cd "$(dirname "$0")" && pwd)
T=$(mktemp -d)
for v in slp noslp; do
  F=""; [ $v = noslp ] && F="-fno-slp-vectorize"
  mkdir -p $T/$v
  (cd $T/$v && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -save-temps -c $D/fold_variants.hip -o f.o 2>/dev/null)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/$v/f.o -o $D/libfold_$v.so
  echo "== $v"; python3 $D/isa_patterns.py $T/$v/fold_variants-hip-amdgcn-amd-amdhsa-gfx950.s
done
rm -rf $T
