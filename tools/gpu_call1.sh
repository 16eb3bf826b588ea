#!/bin/bash
# round 5, first GPU call: parity suite on the new operand layout, then A/B of r4 / layout / layout+kernarg-warm builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q --tb=short 2>&1 | tail -40 ) > $O/gputest1.log
for l in exp/lib_r4.so exp/lib_b.so exp/lib_bw.so; do PILCO_LIB=$l timeout 400 python tools/ab_libs.py >> $O/ab1.log 2>&1; done
for l in exp/lib_bw.so exp/lib_b.so exp/lib_r4.so; do PILCO_LIB=$l timeout 300 python tools/ab_libs.py c2 c2u >> $O/ab1.log 2>&1; done
timeout 200 python tools/head_phases.py > $O/phases_bw.log 2>&1
PILCO_LIB=exp/lib_r4.so timeout 200 python tools/head_phases.py > $O/phases_r4.log 2>&1
tail -5 $O/gputest1.log; cat $O/ab1.log; tail -3 $O/phases_bw.log; tail -3 $O/phases_r4.log
