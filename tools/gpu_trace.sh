#!/bin/bash
# usage: bash tools/gpu_trace.sh <lib tag> ... : rocprofv3 kernel trace of the headline loop per library
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
for l in "$@"; do
  rm -rf $O/trace_$l
  PILCO_LIB=exp/lib_$l.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$l -o r -- python tools/headline_ab.py > $O/trace_$l.log 2>&1 </dev/null
  echo "== $l"; python tools/kstats.py $O/trace_$l | head -3; tail -1 $O/trace_$l.log
done
