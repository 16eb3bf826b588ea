#!/bin/bash
# rocprofv3 kernel stats of BASELINE config 4 (SMGPR M=200, N=5000): FITC factorisation + H=40 rollouts (tools/c4_bench.py)
# -> gpurun_out/c4/ ; copy r_kernel_stats.csv to profiles/rNN_c4_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/c4; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python tools/c4_bench.py > $OUT/trace.log 2>&1 </dev/null
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -14
tail -2 $OUT/trace.log
