# rocprofv3 PMC passes over the dominant aggregation launches (bf16 / fp32 C=256 with one epilogue operand):
# where do the wavefronts' cycles go (issue vs wait), and what does L2 see.  Separate passes: 8 SQ slots, 4 TCC slots.
#   bash tools/pmc_issue.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/${1:-gpurun_out/pmc_issue}
mkdir -p $OUT
cd $R
CMD="python3 tools/agg_bench.py --channels 256,64 --dtypes bf16,fp32 --epilogue 1 --rounds 1 --reps 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tc1 -- $CMD > $OUT/tc1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tc2 -- $CMD > $OUT/tc2.log 2>&1
find $OUT -name "*counter_collection.csv" | head
