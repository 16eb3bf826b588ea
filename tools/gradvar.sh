#!/bin/bash
# usage: tools/gradvar.sh -- R-grad timing + bwd kernel times for the in-tree library and every exp/*.so variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in pilco_amd/libpilco_hip.so exp/*.so; do
  echo "== $lib"
  rm -rf gpurun_out/gv; PILCO_LIB=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gv -o r -- python tools/grad_bench.py 2>&1 </dev/null | grep "C2u"
  python tools/kstats.py gpurun_out/gv | grep "bwd"
done
