# rocprofv3 PMC passes over sg::spmm_ring on its dominant launch (bf16, C = 256, one epilogue operand, V = 1 M): where the
# wavefronts' cycles go (parked / issue-stalled / issuing), what the LDS and the matrix cores do, what L2 sees.
# One counter group per pass, the program directly after `--` (no shell hop between the profiler and the GPU process).
#   bash tools/pmc_ring.sh <outdir> ; python tools/pmc_ring_summary.py <outdir> <out.json>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/${1:-gpurun_out/pmc_ring}
mkdir -p $OUT
cd $R
CMD="python3 tools/agg_bench.py --channels 256 --dtypes bf16 --epilogue 1 --rounds 1 --reps 3"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/tc1 -- $CMD > $OUT/tc1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
find $OUT -name "*counter_collection.csv" | head; find $OUT -name "*kernel_stats.csv" | head -2
tail -3 $OUT/sq1.log $OUT/sq2.log $OUT/tc1.log
