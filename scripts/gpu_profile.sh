#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace of the same command, separate PMC passes for HBM traffic.
#   bash scripts/gpu_profile.sh <tag>     ->  gpurun_out/prof_<tag>/{<tag>_bench.json, <tag>_kernel_stats.md, <tag>_families.{json,md}}
set -o pipefail
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=$OUT/prof_$R
mkdir -p $P
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-batch > $P/trace.log 2>&1 || { tail -5 $P/trace.log; exit 1; }
python scripts/summarize_rocprof.py $P/trace $P/${R}_trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-timing --no-reference-batch > $P/pmc_fetch.log 2>&1 || { tail -5 $P/pmc_fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-timing --no-reference-batch > $P/pmc_write.log 2>&1 || { tail -5 $P/pmc_write.log; exit 1; }
python scripts/summarize_rocprof.py --families $P/trace $P/pmc_fetch $P/pmc_write $P/${R}_families
# the bench line last, so that it can pick the PMC traffic of this very build up
mkdir -p profiles && cp $P/${R}_families.json profiles/r01_families_b96_bf16.json
python bench.py --steps 20 --warmup 5 > $P/${R}_bench.json 2> $P/bench.err || { tail -5 $P/bench.err; exit 1; }
cat $P/${R}_bench.json | cut -c1-600
# keep only the small summaries in the merge-back (raw traces can exceed the 64 MiB limit)
find $P -name "*.csv" -size +1M -delete
ls -la $P
