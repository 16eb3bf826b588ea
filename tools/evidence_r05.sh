#!/bin/bash
# Round-5 evidence in one gpurun call: forward profile (kernel trace + PMC passes + the bench line), value-and-gradient
# profile, config-4 and factorisation kernel traces, head phase stamps.  Outputs under gpurun_out/ -- summarised into
# profiles/r05_* by tools/pmc_summary.py r05 / tools/grad_pmc_summary.py r05 (run here, in the repo, afterwards).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; tail -1 gpurun_out/pr.log | cut -c1-300
bash tools/profile_grad.sh > gpurun_out/pg.log 2>&1; tail -3 gpurun_out/pg.log | cut -c1-160
bash tools/profile_c4.sh > gpurun_out/c4.log 2>&1; tail -2 gpurun_out/c4.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/factprof; mkdir -p gpurun_out/factprof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/factprof/trace -o r -- python tools/fact_bench.py > gpurun_out/factprof/trace.log 2>&1 </dev/null
tail -1 gpurun_out/factprof/trace.log
python tools/head_phases.py 2>&1 | tail -4 > gpurun_out/head_phases_c2.log; cat gpurun_out/head_phases_c2.log
python tools/head_blocks.py 2>&1 | tail -3 > gpurun_out/head_blocks_c2.log; cat gpurun_out/head_blocks_c2.log
python tools/pair_waves.py 2>&1 | head -1 > gpurun_out/pair_waves_c2.log; cat gpurun_out/pair_waves_c2.log
python tools/grad_bench.py > gpurun_out/grad_bench.log 2>&1; cat gpurun_out/grad_bench.log
