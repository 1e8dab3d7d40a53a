#!/bin/bash
# A/B of library builds under a microbenchmark: bash scripts/ab_micro.sh <script.py> [args]; variants/*.so are swapped in
cd $GRAFT_REPO_ROOT
L=image-in-speech-watermarking_amd/libwm_hip.so
cp $L /tmp/base.so
for v in base $(ls variants/*.so 2>/dev/null); do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp $v $L; fi
  echo "== $v"
  python "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so $L
