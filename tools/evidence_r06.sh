#!/bin/bash
# Round-6 evidence in one gpurun call: forward profile (kernel trace + PMC passes + the bench line), value-and-gradient
# profile (+ the timeline of its tail behind the last sweep), config-4, FITC-objective and factorisation kernel traces, head
# phase stamps, the VALU-write -> MFMA-read microbenchmark.  Outputs under gpurun_out/ -- summarised into profiles/r06_* by
# tools/pmc_summary.py r06 / tools/grad_pmc_summary.py r06 (run here, in the repo, afterwards).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
bash tools/profile_round.sh > gpurun_out/pr.log 2>&1; tail -1 gpurun_out/pr.log | cut -c1-300
bash tools/profile_grad.sh > gpurun_out/pg.log 2>&1; tail -3 gpurun_out/pg.log | cut -c1-160
bash tools/profile_c4.sh > gpurun_out/c4.log 2>&1; tail -2 gpurun_out/c4.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/factprof; mkdir -p gpurun_out/factprof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/factprof/trace -o r -- python tools/fact_bench.py > gpurun_out/factprof/trace.log 2>&1 </dev/null
tail -1 gpurun_out/factprof/trace.log
rm -rf gpurun_out/fitcprof; mkdir -p gpurun_out/fitcprof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fitcprof/trace -o r -- python tools/fitc_obj_bench.py > gpurun_out/fitcprof/trace.log 2>&1 </dev/null
tail -1 gpurun_out/fitcprof/trace.log
rm -rf gpurun_out/gtail; mkdir -p gpurun_out/gtail
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gtail/trace -o r -- python tools/grad_ab.py save /tmp/x.npz > gpurun_out/gtail/trace.log 2>&1 </dev/null
python tools/grad_tail.py gpurun_out/gtail/trace > gpurun_out/grad_tail.txt; cat gpurun_out/grad_tail.txt
python tools/head_phases.py 2>&1 | tail -4 > gpurun_out/head_phases_c2.log; cat gpurun_out/head_phases_c2.log
python tools/head_blocks.py 2>&1 | tail -3 > gpurun_out/head_blocks_c2.log; cat gpurun_out/head_blocks_c2.log
python tools/pair_waves.py 2>&1 | head -1 > gpurun_out/pair_waves_c2.log; cat gpurun_out/pair_waves_c2.log
python tools/grad_bench.py > gpurun_out/grad_bench.log 2>&1; cat gpurun_out/grad_bench.log
AB_TAG="host chain (rounds 1-5)" PILCO_HOST_CHAIN=1 python tools/grad_ab.py save /tmp/g_host.npz > gpurun_out/grad_ab.log 2>&1
AB_TAG="device chain" python tools/grad_ab.py cmp /tmp/g_host.npz >> gpurun_out/grad_ab.log 2>&1; cat gpurun_out/grad_ab.log
AB_TAG="VALU kernel-derivative reductions" PILCO_FITC_KGRAD_VALU=1 python tools/fitc_obj_bench.py save /tmp/f.npz > gpurun_out/fitc_ab.log 2>&1
AB_TAG="MFMA kernel-derivative reductions" python tools/fitc_obj_bench.py cmp /tmp/f.npz >> gpurun_out/fitc_ab.log 2>&1; cat gpurun_out/fitc_ab.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -Wno-uninitialized tools/ubench_srcc_war.hip -o /tmp/srcc 2>/dev/null && /tmp/srcc > gpurun_out/ubench_valu_mfma.txt; cat gpurun_out/ubench_valu_mfma.txt
python tools/pair_clock.py > gpurun_out/pair_clock.log 2>&1; cat gpurun_out/pair_clock.log
[ -f exp/lib_jacstamps.so ] && PILCO_LIB=exp/lib_jacstamps.so python tools/jac_phases.py > gpurun_out/jac_phases.log 2>&1; cat gpurun_out/jac_phases.log   # (make -C pilco_amd/csrc BUILD=build_js OUT=exp/lib_jacstamps.so EXTRA=-DJAC_STAMPS)
