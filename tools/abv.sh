# tools/abv.sh v1 v2 ...: bench config 2 with each variant library preloaded ("main" = in-tree)
for v in "$@"; do
  P=""; [ "$v" != "main" ] && P="$GRAFT_REPO_ROOT/variants/$v/libhgs_rast.so"
  LD_PRELOAD=$P timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python tools/fmt.py "$v"
done
