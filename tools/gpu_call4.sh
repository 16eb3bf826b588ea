#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 ) > $O/gputest4.log
for l in exp/lib_r4.so exp/lib_d.so; do PILCO_LIB=$l timeout 400 python tools/ab_libs.py c2 c2u >> $O/ab4.log 2>&1; done
tail -30 $O/gputest4.log; cat $O/ab4.log
