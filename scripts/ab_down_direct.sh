QUIET="--no-cpu-baseline --no-kernel-timing --no-identity-leg --no-parity-mode-leg --no-secondary-legs --ber-train-steps 0"
for r in 1 2; do
  for on in 0 1; do
    WM_DOWN_DIRECT=$on python bench.py --steps 10 --warmup 3 $QUIET 2>/tmp/ab.err > /tmp/ab.json || { tail -5 /tmp/ab.err; exit 1; }
    python - "WM_DOWN_DIRECT=$on" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
rb = d.get("reference_batch") or {}
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 2), "value", round(d["value"], 1), "B24 ms", round(rb.get("ms_per_step", 0.0), 2), flush=True)
PY
  done
done
