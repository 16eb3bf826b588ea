#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/profiles/ (copied to profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $CMD > $OUT/trace.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $CMD > $OUT/pmc_fetch.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $CMD > $OUT/pmc_write.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o r -- $CMD > $OUT/pmc_sq.log 2>&1 </dev/null
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err </dev/null
ls -R $OUT | head -30
tail -1 $OUT/bench.json | cut -c1-400
