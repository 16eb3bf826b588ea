#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
CMD="python tools/headline_ab.py"
for l in r4 d; do
  PILCO_LIB=exp/lib_$l.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$l -o r -- $CMD > $O/trace_$l.log 2>&1 </dev/null
  echo "== $l" >> $O/kstats5.log; python tools/kstats.py $O/trace_$l | head -6 >> $O/kstats5.log
done
cat $O/kstats5.log
