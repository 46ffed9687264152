#!/bin/bash
# first GPU pass of a change: fast parity suites, a bench line with extras, the device timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -15 > $O/pt_fast.log
tail -5 $O/pt_fast.log
timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_a.json 2> $O/bench_a.err
python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/bench_a.json"))
    print("ms %.4f"%b["ms_per_step"], {k:round(v,1) for k,v in b["stage_us"].items()})
    print("host", b["host"])
    for k,e in (b.get("extra") or {}).items():
        print(k, "ms %.4f"%e["ms_per_step"], {a:round(v,1) for a,v in e["stage_us"].items()})
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/bench_a.err").read()[-2000:])
PY
HGS_LIB=$R/variants/timeline/libhgs_rast.so LD_PRELOAD=$R/variants/timeline/libhgs_rast.so timeout 120 python tools/timeline.py > $O/timeline_a.txt 2>&1
head -40 $O/timeline_a.txt
