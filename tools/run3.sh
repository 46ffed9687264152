#!/bin/bash
# parity + batch suites of the in-tree build, short benches of it and of variants/<name> builds, the device timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -12 > $O/pt_fast.log
tail -6 $O/pt_fast.log
bash tools/abv.sh "$@"
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/timeline_a.txt 2>&1
grep -A8 "rank sort per tile" $O/timeline_a.txt; grep "== sort_lds" -A1 $O/timeline_a.txt
