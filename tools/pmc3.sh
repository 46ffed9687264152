R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | head -c 6000; echo
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc3 -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(f"{R}/gpurun_out/pmc3/run_counter_collection.csv")):
        k=r["Kernel_Name"]
        if k.startswith("hgs_k_render"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        print(k, {n: round(sum(v)/len(v)) for n,v in c.items()})
except Exception as e:
    print("failed", e)
PY
rm -rf $R/gpurun_out/pmc3
done
