# PMC passes over the persistent 256 x 256 product (and the BLAS library beside it) at V = 1 M: HBM bytes, LDS conflicts,
# where the wavefront cycles go.   bash tools/gemm256_probe.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/${1:-gpurun_out/gemm256_probe}
mkdir -p $OUT
cd $R
CMD="python3 tools/gemm256_bench.py --rounds 1 --reps 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "gemm_nt_256" if "gemm_nt_256" in k else ("gemm_nt_128" if "gemm_nt_bf16" in k else ("blas:" + k[:60] if k.startswith(("Cijk", "Custom_Cijk")) else None))
            if name is None:
                continue
            key = (name, r.get("Grid_Size"), r["Counter_Name"])
            out.setdefault(key, []).append(float(r["Counter_Value"]))
res = collections.defaultdict(dict)
for (name, grid, c), v in out.items():
    res[f"{name} grid={grid}"][c] = [round(sum(v) / len(v), 1), len(v)]
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, v)
PY
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
