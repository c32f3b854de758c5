# rocprofv3 PMC passes over the split-bf16 fp32 products (csrc/gemm_split.hip): how busy is the matrix pipe, where do the
# wavefronts' cycles go.  Separate passes (8 SQ slots each), counters only.
#   bash tools/pmc_split.sh <outdir> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/${1:-gpurun_out/pmc_split}
shift
mkdir -p $OUT
cd $R
CMD="python3 tools/gemm_f32split_bench.py --quick --no-check --reps 2 --out $OUT/bench.json $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
python3 tools/pmc_split_summary.py $OUT $OUT/summary.json
