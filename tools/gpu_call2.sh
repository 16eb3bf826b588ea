#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 ) > $O/gputest2.log
for l in exp/lib_r4.so exp/lib_bw.so exp/lib_c_nopre.so exp/lib_c.so; do PILCO_LIB=$l timeout 400 python tools/ab_libs.py >> $O/ab2.log 2>&1; done
for l in exp/lib_c.so exp/lib_c_nopre.so exp/lib_r4.so; do PILCO_LIB=$l timeout 300 python tools/ab_libs.py c2 c2u >> $O/ab2.log 2>&1; done
timeout 200 python tools/head_phases.py > $O/phases_c.log 2>&1
tail -30 $O/gputest2.log; cat $O/ab2.log; tail -3 $O/phases_c.log
