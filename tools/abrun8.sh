#!/bin/bash
# tools/abrun.sh OUTTAG variant1 variant2 ...   (on the GPU box) : short bench of the in-tree build and of each variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
run() { # name, preload
  true
  LD_PRELOAD=$2 timeout 100 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 --views 8 2>/dev/null > $O/ab8_$1.json
  python - "$1" <<'PY'
import json,sys
n=sys.argv[1]
for pre in ("ab","ab8"):
    try:
        b=json.load(open(f"gpurun_out/{pre}_{n}.json"))
        print(pre,n,"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
    except Exception as e: print(pre,n,"FAILED",e)
PY
}
run base ""
for v in "$@"; do run $v $R/variants/$v/libhgs_rast.so; done
