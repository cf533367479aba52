#!/bin/bash
# round 4, GPU call C: speculative touch (spec) parity + A/B; split with hitless donors; device builder vs oracle
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04c
( timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_builder.py -q -m gpu --maxfail=6 2>&1 | tail -15 ) > gpurun_out/r04c/tests_a.log
( IDKPT_SPEC=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_worklist.py tests/test_gpu_nocounters.py tests/test_gpu_batching.py -q -m gpu --maxfail=6 2>&1 | tail -15 ) > gpurun_out/r04c/tests_spec.log
( IDKPT_SPEC=2 timeout 600 python tools/fuzz_parity.py 100 9500 2>&1 | grep -v ": OK" | tail -20 ) > gpurun_out/r04c/fuzz_spec.log
( SWEEP_TAG=r04c SWEEP_OPT=SPEC:0,2 SWEEP_BATCHES=1,3,32 SWEEP_DEPTHS=2,5 timeout 1200 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -40 ) > gpurun_out/r04c/sweep_spec.txt
( SWEEP_TAG=r04c2 SWEEP_OPT=SPLIT:0,2 SWEEP_BATCHES=1 SWEEP_DEPTHS=2 IDKPT_SPEC=0 timeout 600 python tools/sweep_r03.py headline atrium 2>&1 | tail -10 ) > gpurun_out/r04c/sweep_split_hitless.txt
( SHARD_MODS=1,2,4,8 SHARD_BANDS=8 SHARD_OPTS="spec=0;spec=2;spec=2,split=2" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -14 ) > gpurun_out/r04c/shard_spec.txt
tail -4 gpurun_out/r04c/tests_a.log gpurun_out/r04c/tests_spec.log; cat gpurun_out/r04c/fuzz_spec.log gpurun_out/r04c/sweep_spec.txt gpurun_out/r04c/sweep_split_hitless.txt gpurun_out/r04c/shard_spec.txt
