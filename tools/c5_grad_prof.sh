# kernel trace of value+gradient rollouts at config-5 size (tools/small_model_bench.py) -> gpurun_out/c5grad/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/c5grad; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python tools/small_model_bench.py > $OUT/trace.log 2>&1 </dev/null
tail -2 $OUT/trace.log | cut -c1-200
python tools/kstats.py $OUT/trace/r_kernel_trace.csv | head -12
