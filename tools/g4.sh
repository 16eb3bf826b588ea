PILCO_GRAD_TIMING=1 timeout 200 python tools/grad_bench.py 2>&1 | tail -4
