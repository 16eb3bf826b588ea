#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
for l in exp/lib_r4.so exp/lib_p.so exp/lib_c_nopre.so; do
  echo "=== $l" >> $O/bisect3.log
  ( PILCO_LIB=$l timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -k "rbf_controller_golden or (fused_heads_with_an_rbf and 225)" 2>&1 | tail -15 ) >> $O/bisect3.log
done
cat $O/bisect3.log
