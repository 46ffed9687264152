for v in "" _pf4 _pf2; do
for cfg in "--points 100000 --sh-degree 0"; do
  HGS_LIB=$GRAFT_REPO_ROOT/humangaussian_amd/libhgs_rast$v.so timeout 120 python bench.py --no-cpu-baseline --steps 20 $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_us']; print('[$v] $cfg', round(d['ms_per_step'],4), {k: round(x,1) for k,x in s.items() if 'render' in k or k=='scan'})"
done; done
