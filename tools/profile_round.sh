# Round-end measurement artifacts (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>     e.g. r01_f
# 1) PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) -> pmc_traffic.json
# 2) rocprofv3 --kernel-trace --stats of the default bench command
# 3) the default bench line (with cpu_baseline) -> bench_<tag>.json
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$TAG -o run -- $BENCH > $O/pmc_fetch_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$TAG -o run -- $BENCH > $O/pmc_write_$TAG.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch_$TAG/run_counter_collection.csv $O/pmc_write_$TAG/run_counter_collection.csv > $O/pmc_traffic_$TAG.json
cp $O/pmc_traffic_$TAG.json $R/profiles/pmc_traffic.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python $R/bench.py --no-cpu-baseline > $O/prof_$TAG.log 2>&1
cd $R
timeout 900 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
tail -c 1500 $O/bench_$TAG.json
head -12 $O/prof_$TAG/run_kernel_stats.csv | cut -c1-160
cat $O/pmc_traffic_$TAG.json
