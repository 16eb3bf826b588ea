#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 200 python tools/head_phases.py ${1:-10} 2>&1 | tail -4
