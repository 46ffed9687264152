#!/bin/bash
# round-end artifacts in one call: full GPU suite + smoke + default bench line, the r04 profiles, the device timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/run_full.sh
bash tools/profile_r04.sh "$1" > $O/profile_r04.log 2>&1; tail -25 $O/profile_r04.log
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/r04_timeline.txt 2>&1
grep -A7 "rank sort per tile" $O/r04_timeline.txt | cut -c1-260
