mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "jacobian_tape or gradient or adjoint or vjp or safe_pilco or native_rollout_grad or random_shapes" > gpurun_out/g2.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/g2.log
timeout 200 python tools/grad_bench.py 2>&1 | tail -5
