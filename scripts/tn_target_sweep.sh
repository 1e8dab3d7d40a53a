#!/bin/bash
# Whole-step sweep of the split-K workgroup target of the 128 x 128 weight-gradient kernels (WM_TN_TARGET, default 512), interleaved:
#   bash scripts/tn_target_sweep.sh "512 384 256 768" [rounds]
T=${1:-"512 384 256"}; R=${2:-2}
QUIET="--no-cpu-baseline --no-kernel-timing --no-identity-leg --no-parity-mode-leg --no-secondary-legs --ber-train-steps 0"
for r in $(seq 1 $R); do
  for t in $T; do
    WM_TN_TARGET=$t python bench.py --steps 10 --warmup 3 $QUIET 2>/tmp/tn.err > /tmp/tn.json || { tail -5 /tmp/tn.err; exit 1; }
    python - "$t" <<'PY'
import json, sys
d = json.load(open("/tmp/tn.json"))
rb = d.get("reference_batch") or {}
print("WM_TN_TARGET", sys.argv[1], "ms/step", round(d["ms_per_step"], 2), "B24 ms", round(rb.get("ms_per_step", 0.0), 2), flush=True)
PY
  done
done
