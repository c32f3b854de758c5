# Everything profiles/r06_* holds that comes from the final code, in one lease:   bash tools/round6.sh <commit>
cd $GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
O=gpurun_out/r06f
mkdir -p $O
# kernel stats + PMC traffic of the headline workload, both precisions
bash tools/profile_round.sh r06f $COMMIT > $O/profile_round.log 2>&1
# the driver's command line
python bench.py --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err
# one line per BASELINE configuration
bash tools/config_table.sh $O/configs_n1.jsonl > $O/config_table.log 2>&1
# c3 (MGCN) with its roofline / pool kernels, 50 K and 1 M, and their kernel traces
python bench.py --model mgcn --dtype fp32 --mesh 250x200 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/bench_mgcn_50k_fp32.json
python bench.py --model mgcn --dtype bf16 --mesh 1000x1000 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > $O/bench_mgcn_1m_bf16.json
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mgcn50k -- python3 bench.py --model mgcn --dtype fp32 --mesh 250x200 --steps 10 --warmup 3 --no-cpu-baseline --no-launch-timer > $O/prof_mgcn50k.log 2>&1; \
  f=$(find $O/prof_mgcn50k -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_mgcn_50k_kernel_stats.csv; rm -rf $O/prof_mgcn50k; \
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mgcn1m -- python3 bench.py --model mgcn --dtype bf16 --mesh 1000x1000 --steps 5 --warmup 2 --no-cpu-baseline --no-launch-timer > $O/prof_mgcn1m.log 2>&1; \
  f=$(find $O/prof_mgcn1m -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_mgcn_1m_kernel_stats.csv; rm -rf $O/prof_mgcn1m )
# float32 products per shape at 1 M, 50 K and 5 K rows
python tools/gemm_f32split_bench.py --out $O/gemm_f32split_bench.json > $O/gemm_f32split_bench.log 2>&1
python tools/gemm_f32split_bench.py --V 50000 --no-check --out $O/gemm_f32split_bench_50k.json >> $O/gemm_f32split_bench.log 2>&1
python tools/gemm_f32split_bench.py --V 5000 --no-check --out $O/gemm_f32split_bench_5k.json >> $O/gemm_f32split_bench.log 2>&1
# one rank of an 8-way job
rm -f $O/rank_proxy_125k.jsonl
for i in 1 2; do
  python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --no-graph 2>/dev/null >> $O/rank_proxy_125k.jsonl
  python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --partitioned 2>/dev/null >> $O/rank_proxy_125k.jsonl
  SEMIGCN_DIST_NATIVE=0 python bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 60 --warmup 10 --partitioned 2>/dev/null >> $O/rank_proxy_125k.jsonl
done
ls -la $O
