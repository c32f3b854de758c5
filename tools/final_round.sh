# Everything profiles/ holds for one round, from the final code, in one lease:
#   bash tools/final_round.sh <tag>     -> gpurun_out/<tag>/...
cd $GRAFT_REPO_ROOT
T=${1:-final}
O=gpurun_out/$T
mkdir -p $O
bash tools/profile_round.sh $T > $O/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err
bash tools/config_table.sh $O/configs_n1.jsonl > $O/config_table.log 2>&1
bash tools/launch_census.sh $T --mesh 250x200 --dtype bf16 > $O/launch_census.log 2>&1
cp $O/launch_census.txt $O/launch_census_c2_bf16.txt
for i in 1 2; do
  for m in "" "--partitioned --no-graph" "--partitioned" "--graph"; do
    python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-cpu-baseline --no-launch-timer --steps 60 --warmup 10 $m 2>/dev/null >> $O/rank_proxy_125k.jsonl
  done
done
python tools/overlap_probe.py 2>/dev/null | grep "^C=" > $O/overlap_probe.txt
python tools/overlap_probe.py --N 512 --Kp 768 2>/dev/null | grep "^C=" >> $O/overlap_probe.txt
python tools/overhead_bench.py 2>/dev/null | grep "per call" > $O/capi_call_overhead.txt
python tools/host_profile.py 354x354 bf16 40 > $O/host_profile_125k.txt 2>&1
python tools/host_profile.py 354x354 bf16 40 partitioned > $O/host_profile_125k_partitioned.txt 2>&1
ls -la $O
