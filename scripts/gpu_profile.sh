#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace of the same command, PMC passes for HBM traffic.
set -o pipefail
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/prof_$R
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_$R.json 2> $OUT/bench_$R.err || { tail -5 $OUT/bench_$R.err; exit 1; }
cat $OUT/bench_$R.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$R/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/prof_$R/trace.log 2>&1 || { tail -5 $OUT/prof_$R/trace.log; exit 1; }
python scripts/summarize_rocprof.py $OUT/prof_$R/trace $OUT/prof_$R/${R}_trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_$R/pmc_fetch -- python3 bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-timing > $OUT/prof_$R/pmc_fetch.log 2>&1 || { tail -5 $OUT/prof_$R/pmc_fetch.log; exit 1; }
python scripts/summarize_rocprof.py $OUT/prof_$R/pmc_fetch $OUT/prof_$R/${R}_pmc_fetch --pmc
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_$R/pmc_write -- python3 bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-timing > $OUT/prof_$R/pmc_write.log 2>&1 || { tail -5 $OUT/prof_$R/pmc_write.log; exit 1; }
python scripts/summarize_rocprof.py $OUT/prof_$R/pmc_write $OUT/prof_$R/${R}_pmc_write --pmc
# keep only the small summaries in the merge-back (raw traces can exceed the 64 MiB limit)
find $OUT/prof_$R -name "*.csv" -size +2M -delete
ls -la $OUT/prof_$R
