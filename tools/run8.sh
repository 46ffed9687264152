#!/bin/bash
# parity suite of the in-tree build (fail fast), single-view A/B against variants, then the device timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 > $O/pt_fast.log
tail -3 $O/pt_fast.log
bash tools/abv.sh "$@"
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/timeline_a.txt 2>&1
grep "== render_fwd" -A7 $O/timeline_a.txt
