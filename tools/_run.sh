python tools/small_step_ab.py 2>&1
python tools/small_phases.py 2>&1 | head -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_launch or cascade or golden or batched or two_models or sparse or random_shapes" 2>&1 | tail -4
