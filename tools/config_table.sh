# One bench line per BASELINE.json configuration on ONE GPU (jsonl), for profiles/ and BASELINE.md section 4.
#   bash tools/config_table.sh <out.jsonl>
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/configs.jsonl}
: > $O
run() { python bench.py "$@" --no-cpu-baseline --single-dtype 2>/dev/null | grep "^{" >> $O; }
run --mesh 100x50 --dtype fp32 --steps 50 --warmup 5
run --mesh 250x200 --dtype fp32 --steps 40 --warmup 5
run --mesh 250x200 --dtype bf16 --steps 40 --warmup 5
run --mesh 250x200 --model mgcn --dtype fp32 --steps 40 --warmup 5
run --mesh 1000x1000 --model mgcn --dtype fp32 --steps 10 --warmup 3
run --mesh 1000x1000 --dtype bf16 --permute --steps 10 --warmup 3
run --mesh 1000x1000 --dtype fp32 --permute --steps 10 --warmup 3
run --mesh 2000x2000 --dtype bf16 --steps 6 --warmup 2
run --mesh 2000x2000 --dtype fp32 --steps 6 --warmup 2
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py --mesh 100x50 --dtype fp32 --steps 50 --warmup 5 --graph --no-cpu-baseline --single-dtype 2>/dev/null | grep "^{" >> $O
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py --mesh 250x200 --dtype fp32 --steps 40 --warmup 5 --graph --no-cpu-baseline --single-dtype 2>/dev/null | grep "^{" >> $O
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py --mesh 250x200 --model mgcn --dtype fp32 --steps 40 --warmup 5 --graph --no-cpu-baseline --single-dtype 2>/dev/null | grep "^{" >> $O
wc -l $O
