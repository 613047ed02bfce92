#!/bin/bash
# first GPU validation: diagnostic, smoke, GPU parity tests, short bench (used via gpurun)
mkdir -p gpurun_out
timeout -k 10 200 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "diag rc=$?"; tail -12 gpurun_out/diag.log
timeout -k 10 240 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout -k 10 700 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
