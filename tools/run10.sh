#!/bin/bash
# A/B of the in-tree build against variants on three workloads: configs[1], 8 views batched, configs[3] (500k, SH 3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
one() { # tag name preload args...
  local TAG=$1 N=$2 P=$3; shift 3
  LD_PRELOAD=$P timeout 300 python bench.py --no-cpu-baseline --no-extra --warmup 5 "$@" 2>/dev/null > $O/ab_${TAG}_$N.json
  python - "$TAG" "$N" <<'PY'
import json,sys
t,n=sys.argv[1:3]
try:
    b=json.load(open(f"gpurun_out/ab_{t}_{n}.json")); print(t,n,"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
except Exception as e: print(t,n,"FAILED",e)
PY
}
for v in base "$@"; do
  P=""; [ $v != base ] && P=$R/variants/$v/libhgs_rast.so
  one c1 $v "$P" --steps 100
  one v8 $v "$P" --views 8 --steps 40
  one c3 $v "$P" --points 500000 --sh-degree 3 --steps 40
done
