#!/bin/bash
# the complete default bench line (all legs) at two batches, with wall time: bash scripts/bench_full_check.sh [B ...]
cd $GRAFT_REPO_ROOT
for b in ${@:-256 384}; do
  t0=$(date +%s); python bench.py --batch $b 2> gpurun_out/bfull_$b.err | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print($b, round(d['value'], 1), round(d['ms_per_step'], 2), round(d['reference_batch']['value'], 1), round(d['identity_attack']['value'], 1),
      round(d['roofline']['frac'], 3), d['roofline']['traffic'] is not None, round(d['cpu_baseline']['value'], 2), d['ber']['attacked']['hip_fp32_bits_equal_cpu'])"
  echo "wall $(( $(date +%s) - t0 )) s"; grep -i "error\|TrainStep" gpurun_out/bfull_$b.err | head -3
done
