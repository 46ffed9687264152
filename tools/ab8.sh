#!/bin/bash
# tools/ab8.sh "ENV=.." ... (on the GPU box): 8-view batched bench of the in-tree build under each environment
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 100 python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 --views 8 2>/dev/null > $O/ab8env_$i.json
  python - "$e" $i <<'PY'
import json,sys
try:
    b=json.load(open(f"gpurun_out/ab8env_{sys.argv[2]}.json"))
    print("8v",sys.argv[1],"ms %.4f"%b["ms_per_step"],{k:round(v,1) for k,v in b["stage_us"].items()})
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
