#!/bin/bash
# GPU tests + forced-distributed bench path at world size 1 (used via gpurun)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
NERF_BENCH_FORCE_DIST=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "dist bench rc=$?"; tail -3 gpurun_out/bench_dist1.err; cat gpurun_out/bench_dist1.json | cut -c1-400
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])"
