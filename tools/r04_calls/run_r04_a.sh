#!/bin/bash
# round 4, GPU call A: whole GPU suite (incl. the new row-band and scene-version tests), then the measurements those two changes are for
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04a
( timeout 1500 python -m pytest tests -q -m gpu --maxfail=8 2>&1 | tail -60 ) > gpurun_out/r04a/gpu_suite.log
( timeout 600 python tools/bench_animated.py 1000000 64 2>&1 | tail -12 ) > gpurun_out/r04a/animated.txt
( SHARD_MODS=1,2,4,8 SHARD_BANDS=1,8 timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -12 ) > gpurun_out/r04a/shard_small_batch.txt
tail -25 gpurun_out/r04a/gpu_suite.log; cat gpurun_out/r04a/animated.txt gpurun_out/r04a/shard_small_batch.txt
