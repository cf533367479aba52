#!/bin/bash
# round 4, GPU call P: k_trace_fused — parity (tests + fuzz at RayDepth 2), then A/B on single frames, small batches and N-GPU shards
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04p
( timeout 900 python -m pytest tests/test_gpu_fused.py -q -m gpu --maxfail=6 2>&1 | tail -40 ) > gpurun_out/r04p/fused_tests.log
( FUZZ_DEPTH=2 timeout 600 python tools/fuzz_parity.py 150 12000 2>&1 | grep -v ": OK" | tail -20 ) > gpurun_out/r04p/fuzz_d2.log
( SWEEP_TAG=r04p SWEEP_OPT=FUSED:0,2 SWEEP_BATCHES=1,3,8 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -30 ) > gpurun_out/r04p/sweep_fused.txt
( SWEEP_TAG=r04p2 SWEEP_OPT=FUSED_SHADE_MIN:1,8,16,32 IDKPT_FUSED=2 SWEEP_BATCHES=1 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline atrium 2>&1 | tail -30 ) > gpurun_out/r04p/sweep_shade_min.txt
( SHARD_MODS=1,2,4,8 SHARD_BANDS=8 SHARD_OPTS="fused=0;fused=2;fused=2,split=0" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -14 ) > gpurun_out/r04p/shard_fused.txt
tail -12 gpurun_out/r04p/fused_tests.log; cat gpurun_out/r04p/fuzz_d2.log gpurun_out/r04p/sweep_fused.txt gpurun_out/r04p/sweep_shade_min.txt gpurun_out/r04p/shard_fused.txt
