#!/bin/bash
# A/B of one environment toggle on the same box: bash scripts/ab_env.sh VAR   (alternates unset / VAR=1, three rounds)
V=$1
for r in 1 2 3; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $V=1; else unset $V; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null > /tmp/ab.json
    python - "$V=$on" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], round(d["ms_per_step"], 2), round(d["reference_batch"]["ms_per_step"], 2), flush=True)
PY
  done
done
