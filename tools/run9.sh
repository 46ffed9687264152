#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/timeline_a.txt 2>&1
grep "== render_fwd" -A20 $O/timeline_a.txt | cut -c1-400
