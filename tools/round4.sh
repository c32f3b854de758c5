# Everything profiles/r04_* holds, from the current code, in one lease:   bash tools/round4.sh <commit>
cd $GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
O=gpurun_out/r04
mkdir -p $O
bash tools/pmc_ring.sh $O/pmc_ring > $O/pmc_ring.log 2>&1
python3 tools/pmc_ring_summary.py $O/pmc_ring $O/pmc_issue_ring.json > $O/pmc_ring_summary.log 2>&1
find $O/pmc_ring -name "*.csv" -size +2M -delete
bash tools/profile_round.sh r04 $COMMIT > $O/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err
bash tools/config_table.sh $O/configs_n1.jsonl > $O/config_table.log 2>&1
bash tools/block_ab.sh $O/block_ab.jsonl > $O/block_ab.txt 2>&1
rm -f $O/planes_ab.jsonl; bash tools/planes_ab.sh $O/planes_ab.jsonl > /dev/null 2>&1
: > $O/host_call_census.jsonl
for w in "--mesh 100x50 --dtype fp32" "--mesh 250x200 --dtype fp32" "--mesh 250x200 --dtype bf16" "--mesh 250x200 --dtype fp32 --model mgcn"; do
  python tools/host_call_census.py $w 2>/dev/null >> $O/host_call_census.jsonl
done
rm -f $O/rank_proxy_125k.jsonl $O/host_call_census.jsonl.old
for i in 1 2; do
  # unpartitioned block of a rank's size; the rank: blocks phase by phase (default) / per-module eager / per-module replayed
  for m in "--no-graph" "--partitioned" "--partitioned --no-phases" "--partitioned --no-phases --graph" "--graph"; do
    python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --steps 60 --warmup 10 $m 2>/dev/null >> $O/rank_proxy_125k.jsonl
  done
done
ls -la $O
