#!/bin/bash
# parity + batch suites of the in-tree build, then single-view and 8-view benches of it and of variants/<name> builds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -4 > $O/pt_fast.log
tail -3 $O/pt_fast.log
bash tools/abv.sh "$@"
run8() { LD_PRELOAD=$2 timeout 200 python bench.py --no-cpu-baseline --no-extra --views 8 --steps 40 --warmup 5 2>/dev/null > $O/ab8_$1.json
  python - "$1" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.load(open(f"gpurun_out/ab8_{n}.json")); print("8views",n,"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
except Exception as e: print(n,"FAILED",e)
PY
}
run8 base ""
for v in "$@"; do run8 $v $R/variants/$v/libhgs_rast.so; done
