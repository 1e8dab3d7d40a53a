#!/bin/bash
# A/B of library builds on one box: bash scripts/ab_variants.sh [bench args] ; variants/*.so are swapped in for libwm_hip.so
# (interleaved rounds; the base build is restored at the end)
cd $GRAFT_REPO_ROOT
L=image-in-speech-watermarking_amd/libwm_hip.so
cp $L /tmp/base.so
for r in 1 2; do
  for v in base $(ls variants/*.so 2>/dev/null); do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp $v $L; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-identity-leg "$@" 2>/dev/null > /tmp/ab.json
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
rb = d.get("reference_batch", {}).get("ms_per_step")
print(sys.argv[1], round(d["ms_per_step"], 2), round(rb, 2) if rb else None, flush=True)
PY
  done
done
cp /tmp/base.so $L
