#!/bin/bash
# usage: tools/variants.sh  -- runs bench.py against every exp/*.so variant and prints the pair-kernel time
for lib in exp/*.so; do
  for njb in ${NJBS:-2 4}; do
    out=$(PILCO_LIB=$PWD/$lib PILCO_PAIR_NJB=$njb timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1)
    echo "$lib njb=$njb $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rollouts/s %.1f ms/rollout %.3f pair_us %.1f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))" 2>&1 | tail -1)"
  done
done
