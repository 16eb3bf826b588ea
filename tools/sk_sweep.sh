#!/bin/bash
# usage: bash tools/sk_sweep.sh : headline rollout against the stream-K cut's knob -- the cost units of a diagonal / off-diagonal
# column step (PILCO_SK_UNITS=ud,uo).  (Round 6 also swept shares by dispatch round with it: docs/dead_ends.md, profiles/r06_sk_cut_sweep.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O; L=$O/sk_sweep.log; : > $L
for rep in 1 2; do
for u in 5,4 6,5 7,6 9,8 4,3 11,8; do
  echo "units $u : $(PILCO_SK_UNITS=$u timeout 120 python tools/headline_ab.py 2>&1 | tail -1)" >> $L
done
done
cat $L
