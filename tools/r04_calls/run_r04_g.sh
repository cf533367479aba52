#!/bin/bash
# round 4, GPU call G: restructured split kernel (cold state in LDS, no-pieces fast path), 4 vs 5 waves per SIMD; leaf pooling statistics
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04g
( timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_instances.py -q -m gpu --maxfail=6 2>&1 | tail -8 ) > gpurun_out/r04g/tests.log
( IDKPT_SPLIT=2 timeout 600 python tools/fuzz_parity.py 120 9900 2>&1 | grep -v ": OK" | tail -20 ) > gpurun_out/r04g/fuzz.log
for occ in 5 1; do
  ( SWEEP_TAG=r04g_o$occ SWEEP_OPT=SPLIT:0,2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2,5 IDKPT_SPLIT_OCC=$occ timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -26 ) > gpurun_out/r04g/sweep_split_occ$occ.txt
done
( SHARD_MODS=1,2,4,8 SHARD_BANDS=8 SHARD_OPTS="split=0;split=2;split=2,split_occ=1" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -14 ) > gpurun_out/r04g/shard_split.txt
( IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so timeout 600 python tools/phase_profile.py 2>&1 | tail -8 ) > gpurun_out/r04g/phase_profile.txt
tail -n 4 gpurun_out/r04g/tests.log; cat gpurun_out/r04g/fuzz.log gpurun_out/r04g/sweep_split_occ5.txt gpurun_out/r04g/sweep_split_occ1.txt gpurun_out/r04g/shard_split.txt gpurun_out/r04g/phase_profile.txt
