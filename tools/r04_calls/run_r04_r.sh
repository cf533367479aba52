#!/bin/bash
# round 4, GPU call R: k_trace2s with the work list handed out scattered (split_scatter): parity, then single frames / small batches / one rank of 8
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04r
( IDKPT_SPLIT_SCATTER=3 timeout 900 python -m pytest tests/test_gpu_split.py -q -m gpu --maxfail=6 2>&1 | tail -5 ) > gpurun_out/r04r/split_tests_scatter3.log
( IDKPT_SPLIT_SCATTER=0 timeout 900 python -m pytest tests/test_gpu_split.py -q -m gpu --maxfail=6 2>&1 | tail -5 ) > gpurun_out/r04r/split_tests_scatter0.log
( IDKPT_FUSED=0 SWEEP_TAG=r04r SWEEP_OPT=SPLIT_SCATTER:6,5,4,3,2,0 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline 2>&1 | tail -30 ) > gpurun_out/r04r/sweep_scatter.txt
( IDKPT_FUSED=0 IDKPT_SPLIT=2 SWEEP_TAG=r04r2 SWEEP_OPT=SPLIT_SCATTER:6,3,0 SWEEP_BATCHES=1 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py atrium interior 2>&1 | tail -30 ) > gpurun_out/r04r/sweep_scatter_dense.txt
( SHARD_MODS=8 SHARD_BANDS=8 SHARD_OPTS="split_scatter=6;split_scatter=4;split_scatter=3;split_scatter=0" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -6 ) > gpurun_out/r04r/shard_scatter.txt
cat gpurun_out/r04r/*.log gpurun_out/r04r/sweep_scatter.txt gpurun_out/r04r/sweep_scatter_dense.txt gpurun_out/r04r/shard_scatter.txt
