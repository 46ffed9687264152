for v in "" _nolds _noshift _noexp _dpp; do
  HGS_LIB=$GRAFT_REPO_ROOT/humangaussian_amd/libhgs_rast$v.so timeout 120 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_us']; print('variant[$v]', round(d['ms_per_step'],4), {k: round(x,1) for k,x in s.items() if 'render' in k})"
done
