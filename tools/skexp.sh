#!/bin/bash
for lib in exp/lib_*.so; do
  out=$(PILCO_LIB=$PWD/$lib timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1)
  echo "$lib $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rollouts/s %.1f ms/rollout %.3f pair_us %.1f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))" 2>&1 | tail -1)"
done
