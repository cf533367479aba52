#!/bin/bash
# round 4, GPU call L: whole GPU suite on the new defaults (pooled leaves and split selected automatically), later split starts, and the bench lines
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04l
( timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 2>&1 | tail -12 ) > gpurun_out/r04l/gpu_suite.log
( SWEEP_TAG=r04l1 SWEEP_OPT=SPLIT_PEEK:64,128,256,512 IDKPT_SPLIT=2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline 2>&1 | tail -9 ) > gpurun_out/r04l/peek.txt
( time timeout 900 python bench.py > gpurun_out/r04l/bench_default.json 2> gpurun_out/r04l/bench_default.err ) 2> gpurun_out/r04l/bench_default.time
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04l/bench_driver.json 2> gpurun_out/r04l/bench_driver.err ) 2> gpurun_out/r04l/bench_driver.time
tail -n 4 gpurun_out/r04l/gpu_suite.log; cat gpurun_out/r04l/peek.txt; tail -n 3 gpurun_out/r04l/bench_default.err gpurun_out/r04l/bench_default.time; tail -n 3 gpurun_out/r04l/bench_driver.time; head -c 1500 gpurun_out/r04l/bench_default.json
