TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof4_$TAG -o run -- python $R/bench.py --no-cpu-baseline --points 500000 --sh-degree 3 --steps 50 --warmup 10 > $O/prof4_$TAG.log 2>&1
cut -c1-150 $O/prof4_$TAG/run_kernel_stats.csv | head -16
