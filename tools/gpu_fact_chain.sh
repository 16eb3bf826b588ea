#!/bin/bash
# usage: bash tools/gpu_fact_chain.sh <lib> ... : kernel trace of the factorisation's launch chain per library
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
for l in "$@"; do
  t=$(basename $l .so); rm -rf $O/fchain_$t
  FACT_REPS=60 PILCO_LIB=$l timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/fchain_$t -o r -- python tools/fact_bench.py > $O/fchain_$t.log 2>&1 </dev/null
  python tools/fact_chain.py $O/fchain_$t > $O/fchain_$t.txt; tail -1 $O/fchain_$t.txt
done
