# Runs of blocks below the C ABI (sg_block_chain_*) against one call per block and against the per-module path, EAGER:
#   bash tools/block_ab.sh <out.jsonl>
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/block_ab.jsonl}
: > $O
run() { python bench.py "$@" --no-cpu-baseline --single-dtype --no-second-order --no-launch-timer --no-graph 2>>$O.err | grep "^{" >> $O; }
for mode in chains blocks modules; do
  unset SEMIGCN_NO_BLOCK_CALLS SEMIGCN_NO_BLOCK_CHAINS
  [ $mode = blocks ] && export SEMIGCN_NO_BLOCK_CHAINS=1
  [ $mode = modules ] && export SEMIGCN_NO_BLOCK_CALLS=1
  run --mesh 100x50 --dtype fp32 --steps 100 --warmup 10
  run --mesh 250x200 --dtype fp32 --steps 40 --warmup 5
  run --mesh 250x200 --dtype bf16 --steps 40 --warmup 5
  run --mesh 250x200 --model mgcn --dtype fp32 --steps 40 --warmup 5
  run --mesh 1000x1000 --dtype bf16 --steps 10 --warmup 3
done
python - $O <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
n = len(rows) // 3
for a, b, c in zip(rows[:n], rows[n:2 * n], rows[2 * n:]):
    print(f"{a['config']['workload'][:4]} V={a['config']['V']:8d} {a['dtype'][:4]}  chains {a['ms_per_step']:8.3f} ms   one call per block {b['ms_per_step']:8.3f} ms   per-module {c['ms_per_step']:8.3f} ms")
PY
