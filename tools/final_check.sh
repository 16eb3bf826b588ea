mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/gputest.log | tail -1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; grep -a "^{" gpurun_out/bench_final.log | cut -c1-200
