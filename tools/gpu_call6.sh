#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
for l in exp/lib_d.so exp/lib_nostore.so exp/lib_d.so exp/lib_nostore.so; do PILCO_LIB=$l timeout 400 python tools/ab_libs.py c2 >> $O/ab6.log 2>&1; done
PILCO_LIB=exp/lib_nostore.so timeout 200 python tools/head_phases.py > $O/phases_nostore.log 2>&1
cat $O/ab6.log; tail -3 $O/phases_nostore.log
