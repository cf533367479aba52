#!/bin/bash
# round 4, GPU call B: the split kernel — parity (tests + fuzz), then A/B against the plain kernel on single frames, small batches and N-GPU shards
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04b
( timeout 900 python -m pytest tests/test_gpu_split.py -q -m gpu --maxfail=6 2>&1 | tail -40 ) > gpurun_out/r04b/split_tests.log
( timeout 600 python tools/fuzz_parity.py 120 9000 2>&1 | grep -v ": OK" | tail -20 ) > gpurun_out/r04b/fuzz.log
( SWEEP_TAG=r04b SWEEP_OPT=SPLIT:0,2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2,5 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -30 ) > gpurun_out/r04b/sweep_split.txt
( SHARD_MODS=1,2,4,8 SHARD_BANDS=8 SHARD_OPTS="split=0;split=2" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -12 ) > gpurun_out/r04b/shard_split.txt
tail -12 gpurun_out/r04b/split_tests.log; cat gpurun_out/r04b/fuzz.log gpurun_out/r04b/sweep_split.txt gpurun_out/r04b/shard_split.txt
