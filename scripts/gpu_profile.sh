#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace of the bench command, separate PMC passes for HBM traffic
# (FETCH_SIZE, WRITE_SIZE) and MFMA busy cycles, then the bench line itself (which picks the PMC summary of this build up).
#   bash scripts/gpu_profile.sh <round-tag> [bench args, e.g. --batch 24 / --attack none]
#   ->  gpurun_out/prof_<tag>/…  and  profiles/<tag>_{bench,kernel_stats,families}_b<B>_<mode>_<attack>.*
set -o pipefail
R=${1:-r02}; shift
ARGS="$@"
case " $ARGS " in *" --gpus "*) echo "gpu_profile.sh profiles ONE rank: drop --gpus (a self-launching bench under rocprofv3 is the forbidden hop)"; exit 2;; esac
B=256; MODE=bf16; ATT=full
set -- $ARGS
while [ $# -gt 0 ]; do case $1 in --batch) B=$2; shift;; --mode) MODE=$2; shift;; --attack) ATT=$2; shift;; esac; shift; done
TAG=b${B}_${MODE}_${ATT}
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=$OUT/prof_${R}_$TAG
mkdir -p $P
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-kernel-timing --no-reference-batch --no-identity-leg --no-parity-mode-leg --no-secondary-legs --ber-train-steps 0"
SID=$(python -c "import importlib; print(importlib.import_module('image-in-speech-watermarking_amd.profiling').source_id())")
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --steps 5 --warmup 2 $QUIET $ARGS > $P/trace.log 2>&1 || { tail -5 $P/trace.log; exit 1; }
python scripts/summarize_rocprof.py $P/trace $P/${R}_trace
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P/pmc_$C -- python3 bench.py --steps 1 --warmup 2 --no-graph $QUIET $ARGS > $P/pmc_$C.log 2>&1 || { tail -5 $P/pmc_$C.log; exit 1; }
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_mfma -- python3 bench.py --steps 1 --warmup 2 --no-graph $QUIET $ARGS > $P/pmc_mfma.log 2>&1 || { tail -5 $P/pmc_mfma.log; exit 1; }
python scripts/summarize_rocprof.py --families $P/trace $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE $P/${R}_families_$TAG $P/pmc_mfma $SID
mkdir -p profiles
cp $P/${R}_families_$TAG.json $P/${R}_families_$TAG.md profiles/
cp $P/${R}_trace_kernel_stats.md profiles/${R}_rocprofv3_kernel_stats_$TAG.md
# the bench line last, so that it can pick the PMC traffic of this very build up
python bench.py --steps 20 --warmup 5 $ARGS > $P/${R}_bench_$TAG.json 2> $P/bench.err || { tail -5 $P/bench.err; exit 1; }
cp $P/${R}_bench_$TAG.json profiles/
cp profiles/${R}_*_$TAG.* $P/ 2>/dev/null
cut -c1-1500 $P/${R}_bench_$TAG.json
# keep only the small summaries in the merge-back (raw traces can exceed the 64 MiB limit)
find $P -name "*.csv" -size +1M -delete
ls -la $P
