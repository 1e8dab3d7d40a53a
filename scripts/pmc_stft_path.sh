python -m pytest tests/test_kernels_gpu.py -x -q -k "stft or attack" > gpurun_out/t_st.log 2>&1; tail -2 gpurun_out/t_st.log
python scripts/stft_microbench.py 192 2>&1 | grep -v amdgpu | tail -8
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_st_$C -- python3 scripts/stft_microbench.py 192 > /dev/null 2>&1; done
python - <<PY
import csv,glob,collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_st_%s/*/*counter_collection.csv"%C)[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "synth" in r["Kernel_Name"] or "analysis" in r["Kernel_Name"]: acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(C,k,len(v),sum(v)/len(v))
PY
find gpurun_out/pmc_st_* -name "*.csv" -size +1M -delete
