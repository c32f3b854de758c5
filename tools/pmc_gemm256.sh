# rocprofv3 PMC passes over the persistent 256 x 256 bf16 products (csrc/gemm_mfma256.hip: gemm_nt_256 / gemm_tn_256, and
# any experimental sibling): how busy is the matrix pipe, at what clock, where do the wavefronts' cycles go.  Separate passes
# (8 SQ slots each), counters only; a kernel trace for the durations.
#   bash tools/pmc_gemm256.sh <outdir> [bench args, e.g. --only tile256]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/${1:-gpurun_out/pmc_gemm256}
shift
mkdir -p $OUT
cd $R
CMD="python3 tools/gemm256_bench.py --rounds 1 --reps 2 --json $OUT/bench.json $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
python3 tools/pmc_split_summary.py $OUT $OUT/summary.json "gemm_[nt][nt]_s?256"
