cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gradprof; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python tools/grad_bench.py > $OUT/trace.log 2>&1 </dev/null
tail -2 $OUT/trace.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-200
