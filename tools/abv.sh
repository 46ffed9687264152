#!/bin/bash
# tools/abv.sh variant1 variant2 ...   (on the GPU box): short single-view bench of the in-tree build and of each variants/NAME build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
run() { # name, preload
  LD_PRELOAD=$2 timeout 100 python bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 2>/dev/null > $O/abv_$1.json
  python - "$1" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.load(open(f"gpurun_out/abv_{n}.json"))
    print(n,"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
except Exception as e: print(n,"FAILED",e)
PY
}
run base ""
for v in "$@"; do run $v $R/variants/$v/libhgs_rast.so; done
