#!/bin/bash
# tools/gpu_round.sh MODE ...   - what a gpurun call runs on the GPU box (the repo root is taken from this file's place).
#   tests                 the whole `-m gpu` suite + smoke()                                    -> gpurun_out/suite_*.log, parity_fullsize.json
#   suite                 tests + the default bench line                                        -> ... + bench_default.json
#   ab NAME ...           short benches (AB_CONFIGS, default "1v 8v cfg3": one view, 8 views batched, configs[3]) of the in-tree
#                         build and of every variants/NAME/libhgs_rast.so (tools/mkvariant.sh; LD_PRELOAD) -> gpurun_out/ab_*.json
#   quick                 parity + batch suites only (fast gate for a kernel change)
#   profile COMMIT        the round's artifacts: rocprofv3 --kernel-trace --stats, PMC FETCH/WRITE traffic (their own passes),
#                         for configs[1], configs[3] and the 8-view batched call; three SQ-counter passes for configs[1]
#                         -> gpurun_out/r06_*; copy what is to be judged into profiles/
#   timeline NAME         device timelines of a -DHGS_TIMELINE variant (tools/timeline.py)      -> gpurun_out/timeline_NAME.txt
#   gap                   per-kernel GPU idle gaps of a step (kernel + HIP-call trace, tools/gap_trace.py) -> gpurun_out/r06_gap_trace.txt
# Modes can be chained:  bash tools/gpu_round.sh quick -- ab a b -- profile abc123
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=r06

bench_line() {  # name, preload, extra bench args...
  local N=$1 PRE=$2; shift 2
  (cd $R && LD_PRELOAD=$PRE timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 "$@" 2>$O/ab_$N.err > $O/ab_$N.json)
  python - "$N" "$O/ab_$N.json" <<'PY'
import json, sys
n, p = sys.argv[1], sys.argv[2]
try:
    b = json.load(open(p))
    print(n, "ms %.4f" % b["ms_per_step"], {k: round(v, 1) for k, v in b["stage_us"].items()})
except Exception as e:
    print(n, "FAILED", e)
PY
}

mode_tests() {
  cd $R
  HGS_PARITY_STATS=$O/parity_fullsize.json timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/suite_gpu.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/suite_smoke.log
}

mode_suite() {
  mode_tests
  cd $R
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
}

mode_quick() {
  cd $R
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q 2>&1 | tail -8 | tee $O/quick_gpu.log
}

mode_ab() {
  local names=("base" "$@")
  for N in "${names[@]}"; do
    local PRE=""; [ "$N" != base ] && PRE=$R/variants/$N/libhgs_rast.so
    for cfg in ${AB_CONFIGS:-1v 8v cfg3}; do
      case $cfg in
        1v) bench_line ${N}_1v "$PRE" ;;
        8v) bench_line ${N}_8v "$PRE" --views 8 --steps 40 ;;
        cfg3) bench_line ${N}_cfg3 "$PRE" --points 500000 --sh-degree 3 --steps 40 ;;
      esac
    done
  done
}

run_set() {  # tag, commit, bench args
  local T=$1 C=$2; shift 2
  local BENCH="python $R/bench.py --no-cpu-baseline --no-extra --init-seconds 0 --steps 20 --warmup 5 $@"
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$T -o run -- $BENCH > $O/pmc_fetch_$T.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$T -o run -- $BENCH > $O/pmc_write_$T.log 2>&1
  python $R/tools/pmc_traffic.py $O/pmc_fetch_$T/run_counter_collection.csv $O/pmc_write_$T/run_counter_collection.csv "$C" > $O/${T}_pmc_traffic.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o run -- python $R/bench.py --no-cpu-baseline --no-extra --init-seconds 0 --steps 100 --warmup 10 $@ > $O/prof_$T.log 2>&1
  cp $O/prof_$T/run_kernel_stats.csv $O/${T}_kernel_stats.csv
  echo "== $T"; cut -d, -f1-4 $O/${T}_kernel_stats.csv | head -12
}

mode_profile() {
  local C=${1:-unknown}
  run_set $TAG "$C"
  run_set ${TAG}_cfg3 "$C" --points 500000 --sh-degree 3
  run_set ${TAG}_8views "$C" --views 8
  cd /tmp; export TMPDIR=/tmp
  local BENCH="python $R/bench.py --no-cpu-baseline --no-extra --init-seconds 0 --steps 20 --warmup 5"
  local i=0
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pmc_sq${i}_$TAG -o run -- $BENCH > $O/pmc_sq${i}_$TAG.log 2>&1
  done
  GRAFT_REPO_ROOT=$R python - "$C" "$TAG" <<'PY'
import csv, collections, json, os, sys, glob
R = os.environ["GRAFT_REPO_ROOT"]; O = f"{R}/gpurun_out"; tag = sys.argv[2]
out = collections.defaultdict(dict)
for d in sorted(glob.glob(f"{O}/pmc_sq*_{tag}")):
    f = os.path.join(d, "run_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("hgs_k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for n, v in c.items():
            out[k][n] = sum(v) / len(v)
dur = {}
for r in csv.DictReader(open(f"{O}/{tag}_kernel_stats.csv")):
    dur[r["Name"]] = float(r["AverageNs"])
CLK = 2.4e9
busy = {}
for short, k in (("render_bwd", "hgs_k_render_bwd"), ("render_fwd", "hgs_k_render_fwd_store"), ("sort", "hgs_k_sort_lds"),
                 ("pair_reduce", "hgs_k_pair_reduce_em")):
    if k in out and k in dur and "SQ_ACTIVE_INST_VALU" in out[k]:
        busy[short] = out[k]["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * CLK * dur[k] * 1e-9)
res = {"_commit": sys.argv[1],
       "_note": "rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5` (configs[1]); "
                "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x 2.4 GHz x the kernel's average duration from the --stats pass)",
       "valu_busy": busy, "kernel_avg_ns": {k: dur[k] for k in dur if k.startswith("hgs_k_")}, "counters": out}
json.dump(res, open(f"{O}/{tag}_sq_counters.json", "w"), indent=1)
print("valu_busy", {k: round(v, 3) for k, v in busy.items()})
PY
  for t in $TAG ${TAG}_cfg3 ${TAG}_8views; do echo "-- $t traffic"; python -c "
import json;d=json.load(open('$O/${t}_pmc_traffic.json'));print({k:round(v/1e6,1) for k,v in d.items() if not k.startswith('_')})"; done
}

mode_gap() {   # kernel-trace + HIP-call trace of the bench loop -> per-kernel idle gaps (tools/gap_trace.py); no counters in this run
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/gap_$TAG -o run -- python $R/bench.py --no-cpu-baseline --no-extra --init-seconds 0 --steps 100 --warmup 10 > $O/gap_$TAG.log 2>&1
  python $R/tools/gap_trace.py $O/gap_$TAG > $O/${TAG}_gap_trace.txt 2>&1; cat $O/${TAG}_gap_trace.txt
  rm -rf $O/gap_$TAG
}

mode_timeline() {
  local N=$1
  (cd $R && HGS_LIB=$R/variants/$N/libhgs_rast.so LD_PRELOAD=$R/variants/$N/libhgs_rast.so timeout 300 python tools/timeline.py > $O/timeline_$N.txt 2>&1; tail -40 $O/timeline_$N.txt)
}

args=()
run_mode() { [ ${#args[@]} -eq 0 ] && return; local m=${args[0]}; echo "##### ${args[*]}"; mode_$m "${args[@]:1}"; args=(); }
for a in "$@"; do
  if [ "$a" == "--" ]; then run_mode; else args+=("$a"); fi
done
run_mode
