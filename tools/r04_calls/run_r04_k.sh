#!/bin/bash
# round 4, GPU call K: split kernel — how soon a busy wave learns that the work list is empty (split_peek), and the grid of a small launch when idle lanes can help
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04k
( IDKPT_SPLIT=2 timeout 300 python -m pytest tests/test_gpu_split.py -q -m gpu --maxfail=3 2>&1 | tail -4 ) > gpurun_out/r04k/tests.log
( SWEEP_TAG=r04k1 SWEEP_OPT=SPLIT_PEEK:64,16,8,4,2,1 IDKPT_SPLIT=2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline 2>&1 | tail -13 ) > gpurun_out/r04k/peek.txt
( SWEEP_TAG=r04k2 SWEEP_OPT=GRID_RAYS_X4:6,5,4,3,2 IDKPT_SPLIT=2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline 2>&1 | tail -11 ) > gpurun_out/r04k/grid.txt
( SWEEP_TAG=r04k3 SWEEP_OPT=SPLIT:0,2 SWEEP_BATCHES=1 SWEEP_DEPTHS=2,5 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -13 ) > gpurun_out/r04k/split_now.txt
( SHARD_MODS=4,8 SHARD_BANDS=8 SHARD_OPTS="split=0;split=2;split=2,split_peek=2;split=2,grid_rays_x4=4" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -10 ) > gpurun_out/r04k/shards.txt
cat gpurun_out/r04k/tests.log gpurun_out/r04k/peek.txt gpurun_out/r04k/grid.txt gpurun_out/r04k/split_now.txt gpurun_out/r04k/shards.txt
