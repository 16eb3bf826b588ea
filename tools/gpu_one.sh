#!/bin/bash
# usage: bash tools/gpu_one.sh "<pytest -k expression>"  : selected GPU tests against the product library
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -k "$1" 2>&1 | tail -30
