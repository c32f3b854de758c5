# Everything profiles/r05_* holds that comes from the final code, in one lease:   bash tools/round5.sh <commit>
cd $GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
O=gpurun_out/r05
mkdir -p $O
bash tools/profile_round.sh r05 $COMMIT > $O/profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err
bash tools/config_table.sh $O/configs_n1.jsonl > $O/config_table.log 2>&1
python tools/gemm_f32split_bench.py --out $O/gemm_f32split_bench.json > $O/gemm_f32split_bench.log 2>&1
python tools/gemm_f32split_bench.py --V 50000 --no-check --out $O/gemm_f32split_bench_50k.json >> $O/gemm_f32split_bench.log 2>&1
cat $O/gemm_f32split_bench_50k.json >> $O/gemm_f32split_bench.json
python tools/ring_f32_probe.py 1000x1000 > $O/ring_f32_probe.txt 2>&1
rm -f $O/rank_proxy_125k.jsonl
for i in 1 2; do
  # an unpartitioned block of a rank's size; the rank: blocks phase by phase with the collectives enqueued by the library
  # (default) / issued one by one through torch.distributed / the per-module path
  python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --no-graph 2>/dev/null >> $O/rank_proxy_125k.jsonl
  python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --partitioned 2>/dev/null >> $O/rank_proxy_125k.jsonl
  SEMIGCN_DIST_NATIVE=0 python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --partitioned 2>/dev/null >> $O/rank_proxy_125k.jsonl
  python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --partitioned --no-phases 2>/dev/null >> $O/rank_proxy_125k.jsonl
done
ls -la $O
