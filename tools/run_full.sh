#!/bin/bash
# the round-end check: the whole GPU suite, smoke, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
HGS_PARITY_STATS=$O/parity_fullsize.json timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -25 > $O/pt_full.log
tail -12 $O/pt_full.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
try:
    b=json.load(open("gpurun_out/bench_full.json"))
    print("ms %.4f value %.3e"%(b["ms_per_step"], b["value"]), {k:round(v,1) for k,v in b["stage_us"].items()})
    print("roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in b["roofline"].items() if k in ("kernel","achieved","frac","avg_us","traffic")}, "path", round(b["roofline"]["path"]["frac"],4))
    print("host", b["host"]); print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"])
    for k,e in (b.get("extra") or {}).items():
        print(k, "ms %.4f"%e["ms_per_step"], {a:round(v,1) for a,v in e["stage_us"].items()})
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/bench_full.err").read()[-3000:])
PY
