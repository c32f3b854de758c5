# Kernel statistics of ONE rank of an 8-way job (125 K vertices, every collective issued on a one-rank RCCL communicator):
#   bash tools/rank_trace.sh <out-dir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/${1:-gpurun_out/r04/rank_trace}
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 20 --warmup 5 --partitioned > $O/run.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/rank_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python3 bench.py --mesh 354x354 --dtype bf16 --single-dtype --no-second-order --no-cpu-baseline --no-launch-timer --no-distributed-estimate --steps 20 --warmup 5 > $O/run2.log 2>&1
f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/block_kernel_stats.csv
rm -rf $O/prof $O/prof2
python3 - $O <<'PY'
import csv, sys
for name in ("rank_kernel_stats.csv", "block_kernel_stats.csv"):
    rows = list(csv.DictReader(open(sys.argv[1] + "/" + name)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(name, "total kernel ms over 25 iterations", round(tot / 1e6, 2), "per iteration", round(tot / 1e6 / 25, 3))
    for r in rows[:18]:
        print(f"  {r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/25:8.3f} ms/it {float(r['AverageNs'])/1e3:8.1f} us")
PY
