# tools/prof_stats.sh TAG [bench args...]   (run on the GPU box through gpurun)
# rocprofv3 --kernel-trace --stats of a short bench run -> gpurun_out/prof_TAG/ + a compact per-kernel table
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python $R/bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 "$@" > $O/prof_$TAG.log 2>&1
cd $R
F=$(ls $O/prof_$TAG/*/run_kernel_stats.csv $O/prof_$TAG/run_kernel_stats.csv 2>/dev/null | head -1)
cp "$F" $O/kernel_stats_$TAG.csv 2>/dev/null
cut -d, -f1-4 $O/kernel_stats_$TAG.csv | head -20
