mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gputest.log
PILCO_BENCH_SHARE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/mp2.log 2>&1; echo "mp2 rc=$?"; grep -a "^{" gpurun_out/mp2.log | cut -c1-600; tail -5 gpurun_out/mp2.log | cut -c1-300
PILCO_BENCH_SHARE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 10 --warmup 2 > gpurun_out/mp4.log 2>&1; echo "mp4 rc=$?"; grep -a "^{" gpurun_out/mp4.log | cut -c1-600
timeout 500 python bench.py > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; grep -a "^{" gpurun_out/bench1.log | cut -c1-1500
