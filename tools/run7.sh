#!/bin/bash
# run5 (parity + batch suites, single-view / 8-view A/B against variants) + the device timeline of the in-tree sources
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/run5.sh "$@"
if [ -f variants/timeline/libhgs_rast.so ]; then
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/timeline_a.txt 2>&1
grep "== render_fwd" -A7 $O/timeline_a.txt; grep "== render_bwd" -A3 $O/timeline_a.txt
fi
