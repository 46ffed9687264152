for cfg in "--points 100000 --sh-degree 0" "--points 500000 --sh-degree 3"; do
  timeout 120 python bench.py --no-cpu-baseline --steps 30 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_us']; print('$cfg', round(d['ms_per_step'],4), d['value'], {k: round(x,1) for k,x in s.items()})"
done
