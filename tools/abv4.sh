for v in "$@"; do
  P=""; [ "$v" != "main" ] && P="$GRAFT_REPO_ROOT/variants/$v/libhgs_rast.so"
  LD_PRELOAD=$P timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --points 500000 --sh-degree 3 2>/dev/null | python tools/fmt.py "$v cfg4"
done
