# Round-end measurement artifacts (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <tag> <commit>     e.g. r03 $(git rev-parse --short HEAD)   (the box has no .git: pass the hash)
# 1) PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) -> profiles/pmc_traffic.json
# 2) SQ counter passes for the blend / sort kernels -> gpurun_out/sq_<tag>.json
# 3) rocprofv3 --kernel-trace --stats of the default bench command (extras and CPU baseline off)
# 4) the default bench line (with cpu_baseline and extras) -> gpurun_out/bench_<tag>.json
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o run -- $BENCH > $O/pmc_fetch_$TAG.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o run -- $BENCH > $O/pmc_write_$TAG.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch_$TAG/run_counter_collection.csv $O/pmc_write_$TAG/run_counter_collection.csv "${2:-unknown}" > $O/pmc_traffic_$TAG.json
cp $O/pmc_traffic_$TAG.json $R/profiles/pmc_traffic.json
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_sq${i}_$TAG -o run -- $BENCH > $O/pmc_sq${i}_$TAG.log 2>&1
done
python - "$TAG" <<'PY'
import csv, collections, json, os, sys, glob
R=os.environ["GRAFT_REPO_ROOT"]; tag=sys.argv[1]
out=collections.defaultdict(dict)
for d in sorted(glob.glob(f"{R}/gpurun_out/pmc_sq*_{tag}")):
    f=os.path.join(d,"run_counter_collection.csv")
    if not os.path.exists(f): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if k.startswith("hgs_k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        for n,v in c.items():
            out[k][n]=sum(v)/len(v)
json.dump(out, open(f"{R}/gpurun_out/sq_{tag}.json","w"), indent=1)
for k in ("hgs_k_render_bwd","hgs_k_render_fwd_store","hgs_k_sort_lds","hgs_k_pair_reduce"):
    print(k, {n: round(v) for n,v in out.get(k,{}).items()})
PY
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python $R/bench.py --no-cpu-baseline --no-extra > $O/prof_$TAG.log 2>&1
cd $R
timeout 600 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -c 600 $O/bench_$TAG.json
cut -d, -f1-4 $O/prof_$TAG/run_kernel_stats.csv | head -14
cat $O/pmc_traffic_$TAG.json | head -20
