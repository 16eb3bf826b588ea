#!/bin/bash
# usage: bash tools/gpu_py.sh <lib or -> <script> [args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$1; shift
if [ "$L" != "-" ]; then export PILCO_LIB=$L; fi
timeout 300 python "$@" 2>&1 | tail -40
