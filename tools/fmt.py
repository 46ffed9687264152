import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
line = sys.stdin.read().strip().splitlines()[-1]
d = json.loads(line)
print(tag, round(d["ms_per_step"], 4), d.get("host"), {k: round(x, 1) for k, x in d["stage_us"].items()})
