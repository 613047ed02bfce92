#!/bin/bash
# bench + rocprofv3 summaries (used via gpurun). $1 = round tag
TAG=${1:-r01}
# ONLY_LDS=1: just the LDS-conflict PMC pass (into an existing gpurun_out/prof_TAG)
ARGS=${2:-}   # extra bench.py arguments, e.g. "--dtype bf16" (then the CPU-baseline bench and the CLI run are skipped)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O/prof_$TAG
cd /tmp && export TMPDIR=/tmp
if [ -n "$ONLY_LDS" ]; then timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/prof_$TAG/pmc_lds -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $ARGS > $O/prof_$TAG/pmc_lds.log 2>&1; echo "lds pass rc=$?"; tail -2 $O/prof_$TAG/pmc_lds.log; exit 0; fi
python3 -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0))); print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)"
timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra $ARGS > $O/bench_gpu_$TAG.json 2> $O/bench_gpu_$TAG.err || { tail -5 $O/bench_gpu_$TAG.err; exit 1; }
cat $O/bench_gpu_$TAG.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG/trace -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra $ARGS > $O/prof_$TAG/trace.log 2>&1 || { tail -5 $O/prof_$TAG/trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/prof_$TAG/pmc_sq -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $ARGS > $O/prof_$TAG/pmc_sq.log 2>&1 || { tail -5 $O/prof_$TAG/pmc_sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_$TAG/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $ARGS > $O/prof_$TAG/pmc_fetch.log 2>&1 || { tail -5 $O/prof_$TAG/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_$TAG/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $ARGS > $O/prof_$TAG/pmc_write.log 2>&1 || { tail -5 $O/prof_$TAG/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/prof_$TAG/pmc_lds -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra $ARGS > $O/prof_$TAG/pmc_lds.log 2>&1 || { tail -5 $O/prof_$TAG/pmc_lds.log; exit 1; }
if [ -n "$ARGS" ]; then find $O/prof_$TAG -name "*.csv" | head -30; exit 0; fi
timeout -k 10 400 python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cat $O/bench_$TAG.json
cd $R && timeout -k 10 120 ./nerf-rs_amd/nerf_cli --scene lego_rust --out $O/output_256.ppm --frames 2 > $O/cli_$TAG.log 2>&1; tail -3 $O/cli_$TAG.log
find $O/prof_$TAG -name "*.csv" | head -30
