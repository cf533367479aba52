#!/bin/bash
# round 4, GPU call AN: adv_min re-swept with the multi modes' refill threshold at 16; MODE 0's refill threshold re-measured (developer build, plain kernel: 32 / 24 / 16)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04an
for a in 1 4 8 12; do
  ( IDKPT_ADV_MIN=$a timeout 400 python tools/bench_multi.py 1000000 3 headline 2>&1 >/dev/null | grep -v one_blas | sed "s/^/adv_min $a: /" ) >> gpurun_out/r04an/multi_adv_refill16.txt
done
( IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so IDKPT_FUSED=0 SWEEP_TAG=r04an SWEEP_OPT=TRACE_VARIANT:100,903,904 SWEEP_BATCHES=32 SWEEP_DEPTHS=2,5 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -20 ) > gpurun_out/r04an/sweep_refill_mode0.txt
cat gpurun_out/r04an/multi_adv_refill16.txt gpurun_out/r04an/sweep_refill_mode0.txt
