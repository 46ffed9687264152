for cfg in "--points 100000 --sh-degree 0" "--points 100000 --sh-degree 0" "--points 100000 --sh-degree 0 --async-mode" "--points 500000 --sh-degree 3"; do
  timeout 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 $cfg 2>/dev/null | python tools/fmt.py "$cfg"
done
