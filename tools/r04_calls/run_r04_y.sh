#!/bin/bash
# round 4, GPU call Y: quad records (k_trace2q): parity, then A/B on single frames, small batches, batches of 32, one rank of 8
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04y
( timeout 900 python -m pytest tests/test_gpu_quad.py -q -m gpu --maxfail=6 2>&1 | tail -25 ) > gpurun_out/r04y/quad_tests.log
( IDKPT_QUAD=2 timeout 600 python tools/fuzz_parity.py 100 15000 2>&1 | grep -v ": OK" | tail -8 ) > gpurun_out/r04y/fuzz_quad.log
( IDKPT_FUSED=0 SWEEP_TAG=r04y SWEEP_OPT=QUAD:0,2 SWEEP_BATCHES=1,3,32 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -30 ) > gpurun_out/r04y/sweep_quad.txt
( IDKPT_FUSED=0 IDKPT_SPLIT=0 SWEEP_TAG=r04y2 SWEEP_OPT=QUAD:0,2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2,5 timeout 900 python tools/sweep_r03.py headline 2>&1 | tail -30 ) > gpurun_out/r04y/sweep_quad_nosplit.txt
( SHARD_MODS=1,8 SHARD_BANDS=8 SHARD_OPTS="quad=0;quad=2;quad=2,leaf_min=8" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -8 ) > gpurun_out/r04y/shard_quad.txt
tail -12 gpurun_out/r04y/quad_tests.log; cat gpurun_out/r04y/fuzz_quad.log gpurun_out/r04y/sweep_quad.txt gpurun_out/r04y/sweep_quad_nosplit.txt gpurun_out/r04y/shard_quad.txt
