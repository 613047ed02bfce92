#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -q -m gpu 2>&1 | tail -4
timeout -k 10 200 python bench.py --dtype bf16 --steps 5 --warmup 1 > gpurun_out/bench_bf16_c3.json 2> gpurun_out/bench_bf16.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/bench_bf16_c3.json')); print('bf16 800x800:', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
timeout -k 10 200 python bench.py --dtype bf16 --ssaa 2 --steps 3 --warmup 1 > gpurun_out/bench_bf16_c5.json 2>> gpurun_out/bench_bf16.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/bench_bf16_c5.json')); print('bf16 C5 1600x1600 rays:', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['workload'])"
