R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o run -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc1","pmc2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{R}/gpurun_out/{d}/run_counter_collection.csv")):
        k=r["Kernel_Name"]
        if k.startswith("hgs_k_render") or k.startswith("hgs_k_fwd"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        print(k, {n: round(sum(v)/len(v)) for n,v in c.items()})
PY
