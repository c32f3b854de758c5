# Steady-state kernel launches per training iteration: two rocprofv3 kernel traces of the same bench command that differ
# only in --steps; the difference of the per-kernel call counts / the difference in steps is what ONE iteration launches.
#   bash tools/launch_census.sh <tag> [bench.py args...]     -> gpurun_out/<tag>/launch_census.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-census}; shift
O=gpurun_out/$T
cd $R
mkdir -p $O
for s in 10 30; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/census_$s -- python3 bench.py "$@" --single-dtype --no-cpu-baseline --no-launch-timer --steps $s --warmup 5 > $O/census_$s.log 2>&1
  f=$(find $O/census_$s -name "*kernel_stats.csv" | head -1); cp "$f" $O/census_$s.csv
  rm -rf $O/census_$s
done
python3 - "$O" <<'PY'
import csv, sys
o = sys.argv[1]
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load(f"{o}/census_10.csv"), load(f"{o}/census_30.csv")
rows = []
for k, (nb, tb) in b.items():
    na, ta = a.get(k, (0, 0.0))
    if nb != na:
        rows.append(((nb - na) / 20.0, (tb - ta) / 20.0 / 1e3, k))
rows.sort(reverse=True)
with open(f"{o}/launch_census.txt", "w") as f:
    f.write(f"launches per iteration: {sum(r[0] for r in rows):.1f}   kernel time per iteration: {sum(r[1] for r in rows) / 1e3:.3f} ms\n")
    own = sum(r[0] for r in rows if "sg::" in r[2])
    f.write(f"  of them the library's own kernels: {own:.1f}\n")
    for n, t, k in rows:
        f.write(f"{n:7.1f} {t:9.1f} us  {k[:160]}\n")
print(open(f"{o}/launch_census.txt").read()[:6000])
PY
