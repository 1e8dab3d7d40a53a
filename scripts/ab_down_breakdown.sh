#!/bin/bash
# per-family kernel time of the step with the im2col form and the grid form of Downsample (eager event pairs of bench.py)
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-identity-leg --no-parity-mode-leg --no-secondary-legs --no-reference-batch --ber-train-steps 0"
for on in 0 1; do
  WM_DOWN_DIRECT=$on python bench.py $Q 2>/tmp/abd.err > /tmp/abd_$on.json || { tail -5 /tmp/abd.err; exit 1; }
done
python - <<'PY'
import json
a = {r["kernel"]: r for r in json.load(open("/tmp/abd_0.json"))["kernel_breakdown"]}
b = {r["kernel"]: r for r in json.load(open("/tmp/abd_1.json"))["kernel_breakdown"]}
print("family                          im2col form: launches ms | grid form: launches ms | delta ms")
tot = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, {}).get("ms", 0) + b.get(k, {}).get("ms", 0))):
    ra, rb = a.get(k, {"launches": 0, "ms": 0.0}), b.get(k, {"launches": 0, "ms": 0.0})
    d = rb["ms"] - ra["ms"]
    tot += d
    if abs(d) > 0.05 or ra["launches"] != rb["launches"]:
        print(f"{k:32s} {ra['launches']:5d} {ra['ms']:8.3f} | {rb['launches']:5d} {rb['ms']:8.3f} | {d:+.3f}")
print("sum of deltas", round(tot, 3), "| step ms", json.load(open("/tmp/abd_0.json"))["ms_per_step"], json.load(open("/tmp/abd_1.json"))["ms_per_step"])
PY
