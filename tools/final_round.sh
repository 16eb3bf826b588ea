#!/bin/bash
# Round-end evidence in one gpurun call: GPU tests, smoke, soak, rocprofv3 profiles (forward + value-and-gradient), bench.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputest.log | cut -c1-160
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python tools/soak.py 2>&1 | tail -2
bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; tail -1 gpurun_out/pr.log | cut -c1-200
bash tools/profile_grad.sh > gpurun_out/pg.log 2>&1; tail -3 gpurun_out/pg.log | cut -c1-160
timeout 500 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; grep -a "^{" gpurun_out/bench_final.log | cut -c1-300
