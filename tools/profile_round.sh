# rocprofv3 summaries of the bench workload for profiles/: kernel stats + PMC traffic of both feature precisions.
#   bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/kernel_stats_{bf16,fp32}.csv, pmc_traffic_{bf16,fp32}.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-prof}
COMMIT=${2:-unknown}
O=gpurun_out/$T
cd $R
mkdir -p $O
for dt in bf16 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$dt -- python3 bench.py --dtype $dt --single-dtype --no-cpu-baseline --no-second-order --steps 5 --warmup 2 > $O/prof_$dt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_$dt/pmc_$c -- python3 bench.py --dtype $dt --single-dtype --no-cpu-baseline --no-second-order --no-launch-timer --steps 2 --warmup 1 > $O/pmc_${dt}_$c.log 2>&1
  done
  python3 tools/pmc_summary.py $O/pmc_$dt $O/pmc_traffic_$dt.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --dtype $dt --single-dtype --no-cpu-baseline --no-second-order --no-launch-timer --steps 2 --warmup 1" $COMMIT
  f=$(find $O/prof_$dt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$dt.csv
  rm -rf $O/prof_$dt $O/pmc_$dt
done
ls -la $O | tail -12
