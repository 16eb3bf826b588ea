#!/bin/bash
# usage: tools/skunits.sh -- bench.py under different stream-K cost-unit ratios (diagonal,off-diagonal)
for u in ${UNITS:-3,2 5,4 11,10 1,1}; do
  out=$(PILCO_SK_UNITS=$u timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -1)
  echo "units=$u $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rollouts/s %.1f ms/rollout %.3f pair_us %.1f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))" 2>&1 | tail -1)"
done
