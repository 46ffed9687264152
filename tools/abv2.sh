for v in "$@"; do
  P=""; [ "$v" != "main" ] && P="$GRAFT_REPO_ROOT/variants/$v/libhgs_rast.so"
  for cfg in "--points 100000 --sh-degree 0" "--points 500000 --sh-degree 3"; do
  LD_PRELOAD=$P timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 20 $cfg 2>/dev/null | python tools/fmt.py "$v $cfg"
  done
done
