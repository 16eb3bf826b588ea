python tools/grad_bench.py 2>&1 | grep -v "^chain"
python tools/grad_bench.py 2>&1 | grep -v "^chain"
PILCO_GRAD_TIMING=1 python tools/grad_bench.py 2>&1 | grep "pilco grad\] forward" | tail -3
