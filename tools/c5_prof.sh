cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/c5prof; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python tools/c5_nograph.py > $OUT/trace.log 2>&1 </dev/null
tail -4 $OUT/trace.log | cut -c1-200
head -16 $OUT/trace/r_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
