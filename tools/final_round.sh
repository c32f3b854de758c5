# Everything profiles/ holds for one round, from the final code, in one lease:
#   bash tools/final_round.sh <tag> <commit>    -> gpurun_out/<tag>/...
cd $GRAFT_REPO_ROOT
T=${1:-final}
COMMIT=${2:-unknown}
O=gpurun_out/$T
mkdir -p $O
bash tools/profile_round.sh $T $COMMIT > $O/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err
python tools/gemm256_bench.py --json $O/gemm256_bench.json > $O/gemm256_bench.log 2>&1
bash tools/config_table.sh $O/configs_n1.jsonl > $O/config_table.log 2>&1
bash tools/launch_census.sh $T --mesh 250x200 --dtype bf16 --no-graph > $O/launch_census.log 2>&1
cp $O/launch_census.txt $O/launch_census_c2_bf16.txt
for i in 1 2; do
  for m in "--no-graph" "--partitioned --no-graph" "--partitioned" "--graph"; do
    python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --steps 60 --warmup 10 $m 2>/dev/null >> $O/rank_proxy_125k.jsonl
  done
done
ls -la $O
