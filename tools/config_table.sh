# One bench line per BASELINE.json configuration on ONE GPU (jsonl), for profiles/ and BASELINE.md section 4.
#   bash tools/config_table.sh <out.jsonl>
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/configs.jsonl}
: > $O
run() { python bench.py "$@" --no-cpu-baseline --single-dtype --no-second-order 2>/dev/null | grep "^{" >> $O; }
# the reference's own sizes, EAGER (bench.py's default since round 4: runs of blocks below the C ABI) ...
run --mesh 100x50 --dtype fp32 --steps 50 --warmup 5 --no-graph
run --mesh 250x200 --dtype fp32 --steps 40 --warmup 5 --no-graph
run --mesh 250x200 --dtype bf16 --steps 40 --warmup 5 --no-graph
run --mesh 250x200 --model mgcn --dtype fp32 --steps 40 --warmup 5 --no-graph
# ... and replayed from a hipGraph after the replay == eager check (--graph; the default of rounds 2-3)
run --mesh 100x50 --dtype fp32 --steps 50 --warmup 5 --graph
run --mesh 250x200 --dtype fp32 --steps 40 --warmup 5 --graph
run --mesh 250x200 --model mgcn --dtype fp32 --steps 40 --warmup 5 --graph
# large meshes
run --mesh 1000x1000 --model mgcn --dtype fp32 --steps 10 --warmup 3
run --mesh 1000x1000 --model mgcn --dtype bf16 --steps 10 --warmup 3
run --mesh 1000x1000 --dtype bf16 --mesh-recipe diagonal --steps 10 --warmup 3
run --mesh 2000x2000 --dtype bf16 --steps 6 --warmup 2
run --mesh 2000x2000 --dtype fp32 --steps 6 --warmup 2
wc -l $O
