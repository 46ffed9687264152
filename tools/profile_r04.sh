#!/bin/bash
# Round-4 measurement artifacts (run on the GPU box through gpurun):  bash tools/profile_r04.sh <commit>
#  configs[1] (the headline): PMC FETCH/WRITE passes, three SQ-counter passes, rocprofv3 --kernel-trace --stats of the default command
#  configs[3] (500k, SH 3) and the 8-view batched call: kernel stats + PMC FETCH/WRITE
# Counters are collected in their own runs (--pmc with --kernel-trace only).  Outputs -> gpurun_out/r04_*; copy what is to be judged to profiles/.
C=${1:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run_set() {  # tag, bench args
  local TAG=$1; shift
  local BENCH="python $R/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 $@"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o run -- $BENCH > $O/pmc_fetch_$TAG.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o run -- $BENCH > $O/pmc_write_$TAG.log 2>&1
  python $R/tools/pmc_traffic.py $O/pmc_fetch_$TAG/run_counter_collection.csv $O/pmc_write_$TAG/run_counter_collection.csv "$C" > $O/${TAG}_pmc_traffic.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python $R/bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 $@ > $O/prof_$TAG.log 2>&1
  cp $O/prof_$TAG/run_kernel_stats.csv $O/${TAG}_kernel_stats.csv
  echo "== $TAG"; cut -d, -f1-4 $O/${TAG}_kernel_stats.csv | head -12
}
run_set r04
run_set r04_cfg3 --points 500000 --sh-degree 3
run_set r04_8views --views 8
# SQ counters, configs[1]
BENCH="python $R/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5"
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_sq${i}_r04 -o run -- $BENCH > $O/pmc_sq${i}_r04.log 2>&1
done
python - "$C" <<'PY'
import csv, collections, json, os, sys, glob
R=os.environ["GRAFT_REPO_ROOT"]; O=f"{R}/gpurun_out"
out=collections.defaultdict(dict)
for d in sorted(glob.glob(f"{O}/pmc_sq*_r04")):
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
# kernel durations from the stats pass -> VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (1024 SIMDs * clock * t)
dur={}
for r in csv.DictReader(open(f"{O}/r04_kernel_stats.csv")):
    dur[r["Name"]]=float(r["AverageNs"])
CLK=2.4e9
busy={}
for short,k in (("render_bwd","hgs_k_render_bwd"),("render_fwd","hgs_k_render_fwd_store"),("sort","hgs_k_sort_lds"),("pair_reduce","hgs_k_pair_reduce_em")):
    if k in out and k in dur and "SQ_ACTIVE_INST_VALU" in out[k]:
        busy[short]=out[k]["SQ_ACTIVE_INST_VALU"]*4.0/(1024*CLK*dur[k]*1e-9)
res={"_commit":sys.argv[1],"_note":"rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5` (configs[1]); "
     "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x 2.4 GHz x the kernel's average duration from the --stats pass)",
     "valu_busy":busy,"kernel_avg_ns":{k:dur[k] for k in dur if k.startswith("hgs_k_")},"counters":out}
json.dump(res, open(f"{O}/r04_sq_counters.json","w"), indent=1)
print("valu_busy", {k:round(v,3) for k,v in busy.items()})
PY
for t in r04 r04_cfg3 r04_8views; do echo "-- $t traffic"; python -c "
import json;d=json.load(open('$O/${t}_pmc_traffic.json'));print({k:round(v/1e6,1) for k,v in d.items() if not k.startswith('_')})"; done
