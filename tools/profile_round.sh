cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for dt in fp32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$dt -- python3 bench.py --dtype $dt --single-dtype --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/prof_$dt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$dt/pmc_$c -- python3 bench.py --dtype $dt --single-dtype --no-cpu-baseline --no-launch-timer --steps 2 --warmup 1 > gpurun_out/pmc_${dt}_$c.log 2>&1
  done
  python3 tools/pmc_summary.py gpurun_out/pmc_$dt gpurun_out/pmc_traffic_$dt.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --dtype $dt --single-dtype --no-cpu-baseline --no-launch-timer --steps 2 --warmup 1"
  f=$(find gpurun_out/prof_$dt -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats_$dt.csv
  rm -rf gpurun_out/prof_$dt gpurun_out/pmc_$dt
done
ls -la gpurun_out | tail -12
