#!/bin/bash
# traffic (PMC FETCH / WRITE) + kernel stats of the in-tree build, single view and 8 views, and the device timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run_set() {
  local TAG=$1; shift
  local BENCH="python $R/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 $@"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o run -- $BENCH > $O/pmc_fetch_$TAG.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o run -- $BENCH > $O/pmc_write_$TAG.log 2>&1
  python $R/tools/pmc_traffic.py $O/pmc_fetch_$TAG/run_counter_collection.csv $O/pmc_write_$TAG/run_counter_collection.csv x > $O/${TAG}_pmc_traffic.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python $R/bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 $@ > $O/prof_$TAG.log 2>&1
  cp $O/prof_$TAG/run_kernel_stats.csv $O/${TAG}_kernel_stats.csv
  echo "== $TAG"; cut -d, -f1-4 $O/${TAG}_kernel_stats.csv | head -10
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print({k:round(v/1e6,1) for k,v in d.items() if isinstance(v,float)})" $O/${TAG}_pmc_traffic.json
}
run_set x1
run_set x8 --views 8
cd $R
if [ -f variants/timeline/libhgs_rast.so ]; then
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/timeline_a.txt 2>&1
grep "== render_fwd" -A12 $O/timeline_a.txt; grep "== render_bwd" -A8 $O/timeline_a.txt
fi
