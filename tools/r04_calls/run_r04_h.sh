#!/bin/bash
# round 4, GPU call H: pooled leaf phase (leaf_pool) parity + A/B; whole GPU suite on the current tree (split auto rule, scene versions, bands ...)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04h
( IDKPT_LEAF_POOL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_worklist.py tests/test_gpu_nocounters.py tests/test_gpu_batching.py tests/test_gpu_fullsize.py tests/test_gpu_glref.py -q -m gpu --maxfail=6 2>&1 | tail -8 ) > gpurun_out/r04h/tests_pool.log
( IDKPT_LEAF_POOL=1 timeout 600 python tools/fuzz_parity.py 150 10100 2>&1 | grep -v ": OK" | tail -12 ) > gpurun_out/r04h/fuzz_pool.log
( SWEEP_TAG=r04h SWEEP_OPT=LEAF_POOL:0,1 SWEEP_BATCHES=32,1 SWEEP_DEPTHS=2,5 timeout 1200 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -26 ) > gpurun_out/r04h/sweep_pool.txt
( timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 2>&1 | tail -12 ) > gpurun_out/r04h/gpu_suite.log
tail -n 3 gpurun_out/r04h/tests_pool.log; cat gpurun_out/r04h/fuzz_pool.log gpurun_out/r04h/sweep_pool.txt; tail -n 5 gpurun_out/r04h/gpu_suite.log
