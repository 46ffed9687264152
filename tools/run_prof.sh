#!/bin/bash
# round artifacts without the test suite: the r04 profiles (PMC, kernel stats, SQ counters) and the device timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/profile_r04.sh "$1" > $O/profile_r04.log 2>&1; tail -40 $O/profile_r04.log
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/r04_timeline.txt 2>&1
grep "== render_fwd" -A6 $O/r04_timeline.txt | cut -c1-260
