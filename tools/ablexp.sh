#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for abl in 0 1 2 4 8 16; do
  rm -rf /tmp/ab$abl; PILCO_ABL=$abl timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$abl -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ab$abl.log 2>&1 </dev/null
  echo "ABL=$abl $(grep -E 'k_mm_prep|k_glue|k_mm_pair' /tmp/ab$abl/r_kernel_stats.csv | awk -F, '{gsub(/"/,"",$1); split($1,a,"("); printf "%s=%.1fus ", substr(a[1],1,22), $4/1000}')"
done
