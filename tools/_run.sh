python tools/small_step_ab.py 2>&1 | head -4
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_launch or cascade or golden or batched or two_models" 2>&1 | tail -4
