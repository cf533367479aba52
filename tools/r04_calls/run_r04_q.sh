#!/bin/bash
# round 4, GPU call Q: the leaf's first triangle requested in the node step (k_trace2 DBG 32, developer build): A/B against the plain (100) kernel, and instrumented
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04q
export IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so
( IDKPT_FUSED=0 SWEEP_TAG=r04q SWEEP_OPT=TRACE_VARIANT:100,121,0 SWEEP_BATCHES=32,1 SWEEP_DEPTHS=2,5 timeout 1200 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -40 ) > gpurun_out/r04q/sweep_pref.txt
( PHASE_VARIANT=122 timeout 600 python tools/phase_profile.py 2>&1 | tail -8 ) > gpurun_out/r04q/phase_profile_122.txt
cat gpurun_out/r04q/sweep_pref.txt gpurun_out/r04q/phase_profile_122.txt
