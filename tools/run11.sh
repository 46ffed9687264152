#!/bin/bash
# a variant build under the parity + batch suites (LD_PRELOAD), then single-view and 8-view A/B against the in-tree build
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
LD_PRELOAD=$R/variants/$V/libhgs_rast.so HGS_LIB=$R/variants/$V/libhgs_rast.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -6 > $O/pt_$V.log
tail -4 $O/pt_$V.log
one() { LD_PRELOAD=$2 timeout 200 python bench.py --no-cpu-baseline --no-extra --warmup 5 $3 2>/dev/null > $O/ab_$1.json
  python - "$1" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.load(open(f"gpurun_out/ab_{n}.json")); print(n,"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
except Exception as e: print(n,"FAILED",e)
PY
}
one c1_base "" "--steps 100"
one c1_$V $R/variants/$V/libhgs_rast.so "--steps 100"
one v8_base "" "--views 8 --steps 40"
one v8_$V $R/variants/$V/libhgs_rast.so "--views 8 --steps 40"
