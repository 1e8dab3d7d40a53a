#!/bin/bash
# short rocprofv3 kernel trace of the bench step; prints the rows matching $1 (default: conv)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_k -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-batch --no-identity-leg > gpurun_out/tr_k.log 2>&1
python scripts/summarize_rocprof.py gpurun_out/tr_k gpurun_out/tr_k_sum > /dev/null
grep -i "${1:-conv}" gpurun_out/tr_k_sum_kernel_stats.md | cut -c1-150
find gpurun_out/tr_k -name "*.csv" -size +1M -delete
