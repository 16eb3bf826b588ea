#!/bin/bash
# usage: bash tools/gpu_ab.sh <tag> <what...> -- <lib> <lib> ...   (A/B in one call; default product library = the test target)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
TAG=$1; shift
WHAT=""
while [ "$1" != "--" ] && [ -n "$1" ]; do WHAT="$WHAT $1"; shift; done
shift
LIBS="$@"
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -25 ) > $O/gputest_$TAG.log
for rep in 1 2; do for l in $LIBS; do PILCO_LIB=$l timeout 400 python tools/ab_libs.py $WHAT >> $O/ab_$TAG.log 2>&1; done; done
PILCO_LIB=$(echo $LIBS | awk '{print $NF}') timeout 200 python tools/head_blocks.py > $O/blocks_$TAG.log 2>&1
tail -4 $O/gputest_$TAG.log; cat $O/ab_$TAG.log; tail -6 $O/blocks_$TAG.log
