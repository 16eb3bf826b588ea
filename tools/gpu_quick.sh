#!/bin/bash
# usage: bash tools/gpu_quick.sh <tag> <what...> -- <lib> <lib> ...   (A/B without the test suite, then the head's phases of the last library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
TAG=$1; shift
WHAT=""
while [ "$1" != "--" ] && [ -n "$1" ]; do WHAT="$WHAT $1"; shift; done
shift
LIBS="$@"
for rep in 1 2; do for l in $LIBS; do PILCO_LIB=$l timeout 400 python tools/ab_libs.py $WHAT >> $O/ab_$TAG.log 2>&1; done; done
cat $O/ab_$TAG.log
PILCO_LIB=$(echo $LIBS | awk '{print $NF}') timeout 200 python tools/head_phases.py 10 2>&1 | tail -4
