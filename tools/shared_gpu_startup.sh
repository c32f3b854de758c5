# N consecutive start-ups of the 8-rank code path with the ranks sharing ONE GPU (gloo-staged collectives), under the per-rank
# supervisor: does every attempt reach its first timed step, how long does it take, and -- when one does not -- what were its
# workers doing (start-up marks and faulthandler dumps from the supervisor's marker directory, which bench.py prints).
#   bash tools/shared_gpu_startup.sh <out.jsonl> [runs=20] [mesh=2000x2000] [gpus=8]
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/shared_gpu_ranks.jsonl}
N=${2:-20}
MESH=${3:-2000x2000}
G=${4:-8}
mkdir -p $(dirname $O)
: > $O
export SEMIGCN_BENCH_SHARE_GPU=1
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  timeout 1500 python bench.py --gpus $G --mesh $MESH --steps 3 --warmup 5 --no-cpu-baseline 2> $O.err | grep "^{" > $O.line
  t1=$(date +%s.%N)
  python - "$O" "$i" "$G" "$MESH" "$t0" "$t1" "$O.err" <<'PY'
import json, re, sys
out, i, n, mesh, t0, t1, err = sys.argv[1:]
line = open(out + ".line").read().strip()
text = open(err).read()
rec = {"run": int(i), "gpus_asked": int(n), "mesh": mesh, "ranks_share_one_gpu": True, "wall_s_whole_command": round(float(t1) - float(t0), 1)}
sup = re.findall(r"bench.py supervisor \(rank \d+\): (rank .*)", text)
if line:
    d = json.loads(line)
    dd = d.get("distributed") or {}
    rec.update({"ms_per_step": round(d["ms_per_step"], 2), "attempt": dd.get("attempt"), "per_module_path": dd.get("per_module_path"),
                "first_attempt_failure": dd.get("first_attempt_failure"), "collectives_per_iteration": dd.get("collectives_per_iteration"),
                "block_calls_per_iteration": dd.get("block_calls_per_iteration"), "startup_marks_s_rank0": dd.get("startup_marks_s")})
else:
    rec["error"] = text[-1500:]
if sup:
    rec["supervisor_messages"] = sorted(set(sup))[:6]
    tb = re.findall(r"traceback of attempt \d+:\n(.*?)(?=\nbench.py supervisor|\Z)", text, flags=re.S)
    if tb:
        rec["traceback_excerpt"] = tb[0][-1500:]
open(out, "a").write(json.dumps(rec) + "\n")
print(json.dumps({k: v for k, v in rec.items() if k != "startup_marks_s_rank0"})[:400])
PY
done
rm -f $O.line
