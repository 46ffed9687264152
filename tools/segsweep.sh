for t in "256 512" "256 1024" "256 2048" "512 1024" "512 2048" "128 512" "256 512"; do
  set -- $t
  sed -i "s/^#define HGS_SEG [0-9]*/#define HGS_SEG $1/; s/^#define HGS_SEG_THRESH [0-9]*/#define HGS_SEG_THRESH $2/" humangaussian_amd/csrc/hgs_common.h
  python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
  echo "SEG $1 THRESH $2"
  for cfg in "--points 100000 --sh-degree 0" "--points 500000 --sh-degree 3"; do
    timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 20 $cfg 2>/dev/null | python tools/fmt.py "$cfg"
  done
done
