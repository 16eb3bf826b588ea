#!/bin/bash
# Round-4 evidence in one gpurun call: forward profile (kernel trace + PMC passes + bench line), value-and-gradient profile,
# config 4 / config 5 / factorisation kernel traces.  Outputs under gpurun_out/ (summarised into profiles/r04_* afterwards).
mkdir -p gpurun_out
bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; tail -1 gpurun_out/pr.log | cut -c1-200
bash tools/profile_grad.sh > gpurun_out/pg.log 2>&1; tail -3 gpurun_out/pg.log | cut -c1-160
bash tools/profile_c4.sh > gpurun_out/c4.log 2>&1; tail -2 gpurun_out/c4.log | cut -c1-160
bash tools/c5_grad_prof.sh > gpurun_out/c5g.log 2>&1; tail -3 gpurun_out/c5g.log | cut -c1-160
bash tools/c5_prof.sh > gpurun_out/c5.log 2>&1; tail -3 gpurun_out/c5.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/factprof; mkdir -p gpurun_out/factprof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/factprof/trace -o r -- python tools/fact_bench.py > gpurun_out/factprof/trace.log 2>&1 </dev/null
tail -1 gpurun_out/factprof/trace.log
python tools/fact_ab.py 2>&1 | grep graph > gpurun_out/fact_ab.log; cat gpurun_out/fact_ab.log
python tools/small_step_ab.py > gpurun_out/small_step_ab.log 2>&1; cat gpurun_out/small_step_ab.log
python tools/grad_bench.py > gpurun_out/grad_bench.log 2>&1; cat gpurun_out/grad_bench.log
python tools/restart_lanes_bench.py > gpurun_out/restart_lanes.log 2>&1; cat gpurun_out/restart_lanes.log
python tools/small_phases_c5.py > gpurun_out/small_phases_c5.log 2>&1; cat gpurun_out/small_phases_c5.log
python tools/small_phases.py 2>&1 | tail -2 > gpurun_out/small_phases_c4.log; cat gpurun_out/small_phases_c4.log
python examples/inverted_pendulum.py 2>&1 | tail -4 > gpurun_out/c5_loop.log; cat gpurun_out/c5_loop.log
