#!/bin/bash
# tools/abenv.sh "ENV=.. ENV2=.." ...   (on the GPU box): short single-view bench of the in-tree build under each environment
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 100 python bench.py --no-cpu-baseline --no-extra --steps 150 --warmup 20 2>/dev/null > $O/abenv_$i.json
  python - "$e" $i <<'PY'
import json,sys
try:
    b=json.load(open(f"gpurun_out/abenv_{sys.argv[2]}.json"))
    print(sys.argv[1],"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
