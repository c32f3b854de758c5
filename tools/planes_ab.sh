# A/B of the narrow layers' plane layout (sg_block_planar) on the headline workload, both orders in one lease:
#   bash tools/planes_ab.sh <out.jsonl>
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/r04/planes_ab.jsonl}
mkdir -p $(dirname $OUT)
for i in 1 2; do
  for m in "" "--no-planes"; do
    python bench.py --steps 20 --warmup 5 --single-dtype --no-second-order --no-cpu-baseline --no-distributed-estimate $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
nar=[k for k in d['aggregation_kernels'] if k['C']<=64]
print(json.dumps({'planes': '$m'=='' , 'ms_per_step': round(d['ms_per_step'],3), 'narrow_agg_ms': round(sum(k['total_ms'] for k in nar)/d['steps'],3),
  'all_agg_ms': round(sum(k['total_ms'] for k in d['aggregation_kernels'])/d['steps'],3), 'dense_ms': d['dense_products']['ms_per_iteration'],
  'narrow': [[k['C'],k['epilogue_operands'],round(k['mean_ms']*1e3,1),round(k['achieved_GBs']/8000,3)] for k in nar]}))" >> $OUT
  done
done
cat $OUT
