#!/bin/bash
# round 4, GPU call O: instance entries / TLAS steps taken by several lanes together (adv_min); instrumented A/B of the pooled leaf phase on one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04o
( IDKPT_ADV_MIN=16 timeout 600 python -m pytest tests/test_gpu_instances.py tests/test_gpu_scene_updates.py -q -m gpu --maxfail=4 2>&1 | tail -4 ) > gpurun_out/r04o/tests.log
for a in 1 8 16 24 32; do
  ( IDKPT_ADV_MIN=$a timeout 600 python tools/bench_multi.py 1000000 3 headline > gpurun_out/r04o/multi_headline_adv$a.json 2> gpurun_out/r04o/multi_headline_adv$a.txt )
done
for a in 1 16; do
  ( IDKPT_ADV_MIN=$a timeout 600 python tools/bench_multi.py 1000000 3 interior > gpurun_out/r04o/multi_interior_adv$a.json 2> gpurun_out/r04o/multi_interior_adv$a.txt )
done
for v in 113 116; do
  ( IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so PHASE_VARIANT=$v timeout 600 python tools/phase_profile.py 2>&1 | tail -8 ) > gpurun_out/r04o/phase_profile_$v.txt
done
tail -n 2 gpurun_out/r04o/tests.log; for f in gpurun_out/r04o/multi_*_adv*.txt; do echo "== $f"; grep -v one_blas $f; done; cat gpurun_out/r04o/phase_profile_113.txt gpurun_out/r04o/phase_profile_116.txt
