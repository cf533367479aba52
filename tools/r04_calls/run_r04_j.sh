#!/bin/bash
# round 4, GPU call J: pooled leaf phase per launch kind (mask: 1 primary, 2 first bounce, 4 later bounces)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04j
( SWEEP_TAG=r04j SWEEP_OPT=LEAF_POOL:0,1,3,7,2,4 IDKPT_POOL_MIN=0 SWEEP_BATCHES=32 SWEEP_DEPTHS=2,5 timeout 1500 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -40 ) > gpurun_out/r04j/pool_mask.txt
cat gpurun_out/r04j/pool_mask.txt
