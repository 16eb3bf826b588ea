#!/bin/bash
# kernel trace of the training objectives' evaluation (tools/nlml_bench.py): per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O; rm -rf $O/nlml_trace
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nlml_trace -o r -- python tools/nlml_bench.py > $O/nlml_trace.log 2>&1 </dev/null
python tools/kstats.py $O/nlml_trace | head -30; tail -2 $O/nlml_trace.log
