#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06/rank_model; rm -rf $O; mkdir -p $O
for cfg in "1 0" "2 2" "2 0" "4 2" "4 0" "8 2" "8 0"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/W$1v$2 -o r -- python tools/rank_model.py run $1 $2 > $O/W$1v$2.log 2>&1 </dev/null
done
python tools/rank_model.py summarise $O/W1v0 $O/W2v2 $O/W2v0 $O/W4v2 $O/W4v0 $O/W8v2 $O/W8v0 > $O/r06_rank_model.json
cat $O/r06_rank_model.json
