#!/bin/bash
# round 4, GPU call I: pooled leaf phase — threshold sweep
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04i
( SWEEP_TAG=r04i0 SWEEP_OPT=LEAF_POOL:0 SWEEP_BATCHES=32,3 SWEEP_DEPTHS=2,5 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -13 ) > gpurun_out/r04i/base.txt
( SWEEP_TAG=r04i1 SWEEP_OPT=POOL_MIN:0,12,20,32,48 IDKPT_LEAF_POOL=1 SWEEP_BATCHES=32,3 SWEEP_DEPTHS=2,5 timeout 1500 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -62 ) > gpurun_out/r04i/pool_min.txt
cat gpurun_out/r04i/base.txt gpurun_out/r04i/pool_min.txt
