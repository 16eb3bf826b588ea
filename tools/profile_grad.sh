#!/bin/bash
# Kernel trace + SQ counters of the C2u value+gradient path (tools/c2u_bench.py) -> gpurun_out/grad/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/grad; rm -rf $OUT; mkdir -p $OUT
CMD="python tools/c2u_bench.py"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $CMD > $OUT/trace.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o r -- $CMD > $OUT/pmc_sq.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/pmc_lds -o r -- $CMD > $OUT/pmc_lds.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $CMD > $OUT/pmc_fetch.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $CMD > $OUT/pmc_write.log 2>&1 </dev/null
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -12
