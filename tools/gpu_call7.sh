#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
PILCO_LIB=exp/lib_d.so timeout 200 python tools/head_blocks.py > $O/blocks_d.log 2>&1
PILCO_LIB=exp/lib_d.so timeout 200 python tools/head_blocks.py 11 >> $O/blocks_d.log 2>&1
cat $O/blocks_d.log
