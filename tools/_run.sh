python tools/small_step_ab.py
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
