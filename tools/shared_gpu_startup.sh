# Start-up and run time of the N-rank code path with the ranks sharing ONE GPU (gloo-staged collectives), under the per-rank
# supervisor: how long does an attempt take before its first timed step, against bench.py's ATTEMPT_TIMEOUT_S?
#   bash tools/shared_gpu_startup.sh <out.jsonl>
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/shared_gpu_ranks.jsonl}
mkdir -p $(dirname $O)
: > $O
export SEMIGCN_BENCH_SHARE_GPU=1
for cfg in "8 2000x2000" "8 1000x1000" "2 1000x1000"; do
  set -- $cfg
  t0=$(date +%s.%N)
  python bench.py --gpus $1 --mesh $2 --steps 3 --warmup 5 --no-cpu-baseline 2> $O.err.$1.$2 | grep "^{" > $O.line
  t1=$(date +%s.%N)
  python - "$O" "$1" "$2" "$t0" "$t1" "$O.err.$1.$2" <<'PY'
import json, re, sys
out, n, mesh, t0, t1, err = sys.argv[1:]
line = open(out + ".line").read().strip()
rec = {"gpus_asked": int(n), "mesh": mesh, "ranks_share_one_gpu": True, "wall_s_whole_command": round(float(t1) - float(t0), 1)}
marks = re.findall(r"\[bench \+\s*([0-9.]+)s\] (.*)", open(err).read())
rec["log_marks_s"] = {m[1][:60]: float(m[0]) for m in marks if any(k in m[1] for k in ("model built", "warm-up iteration 0", "timed region done", "mesh generated"))}
if line:
    d = json.loads(line)
    rec.update({"ms_per_step": d["ms_per_step"], "distributed": d.get("distributed")})
else:
    rec["error"] = open(err).read()[-500:]
open(out, "a").write(json.dumps(rec) + "\n")
print(json.dumps(rec)[:600])
PY
done
rm -f $O.line
