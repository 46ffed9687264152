for cfg in "--points 1000 --sh-degree 0" "--points 1000 --sh-degree 0 --async-mode" "--points 100000 --sh-degree 0" "--points 100000 --sh-degree 0 --async-mode" "--points 100000 --sh-degree 0" "--points 500000 --sh-degree 3 --async-mode"; do
  timeout 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_us']; print('$cfg', round(d['ms_per_step'],4), round(sum(s.values()),1))"
done
