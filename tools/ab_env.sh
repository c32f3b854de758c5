# A/B of one environment switch on the bench line, alternating in one lease:   bash tools/ab_env.sh VAR "v1 v2" "bench args" [rounds]
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; ARGS=$3; N=${4:-2}
for i in $(seq $N); do
  for v in $VALS; do
    ms=$(env $VAR=$v python bench.py $ARGS --single-dtype --no-second-order --no-cpu-baseline --no-distributed-estimate 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "$VAR=$v $ARGS: $ms ms"
  done
done
