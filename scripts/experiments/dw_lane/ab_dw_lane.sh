#!/bin/bash
# Interleaved A/B of the weight-gradient lane (WM_DW_LANE=0 / 1 with dw_lane.patch applied) at the batches of the bench line:
#   bash scripts/experiments/dw_lane/ab_dw_lane.sh [rounds]
#   B = 256 (headline) with its B = 24 reference-batch leg, B = 96, and config 5's per-GPU shape (2 clips x 10 tiles)
R=${1:-2}
QUIET="--no-cpu-baseline --no-kernel-timing --no-identity-leg --no-parity-mode-leg --no-secondary-legs --ber-train-steps 0"
for r in $(seq 1 $R); do
  for on in 0 1; do
    export WM_DW_LANE=$on
    python bench.py --steps 10 --warmup 3 $QUIET 2>/tmp/ab.err > /tmp/ab.json || { tail -5 /tmp/ab.err; exit 1; }
    python bench.py --steps 20 --warmup 3 --clip-seconds 10 --batch 20 $QUIET 2>/tmp/ab5.err > /tmp/ab5.json || { tail -5 /tmp/ab5.err; exit 1; }
    python bench.py --steps 10 --warmup 3 --batch 96 --no-reference-batch $QUIET 2>/tmp/ab96.err > /tmp/ab96.json || { tail -5 /tmp/ab96.err; exit 1; }
    python - "WM_DW_LANE=$on" <<'PY'
import json, sys
d, d5, d96 = json.load(open("/tmp/ab.json")), json.load(open("/tmp/ab5.json")), json.load(open("/tmp/ab96.json"))
rb = d.get("reference_batch") or {}
print(sys.argv[1], "B256 ms/step", round(d["ms_per_step"], 2), "value", round(d["value"], 1), "| B24 ms", round(rb.get("ms_per_step", 0.0), 2),
      "value", round(rb.get("value", 0.0), 1), "| B96 ms", round(d96["ms_per_step"], 2), "| config5 (B20) ms", round(d5["ms_per_step"], 2), "value", round(d5["value"], 1),
      "| loss", [round(v, 4) for v in d["loss_terms"]], flush=True)
PY
  done
done
