#!/bin/bash
# round 4, GPU call F: per-instance trace-ready records (k_trace2 MODE 3 / 4) parity + A/B; split with bound propagation
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04f
( timeout 900 python -m pytest tests/test_gpu_instances.py tests/test_gpu_split.py tests/test_gpu_versions.py tests/test_gpu_glref_full.py tests/test_gpu_glref.py tests/test_gpu_scene_updates.py -q -m gpu --maxfail=8 2>&1 | tail -25 ) > gpurun_out/r04f/tests.log
( timeout 600 python tools/fuzz_parity.py 150 9700 2>&1 | grep -v ": OK" | tail -20 ) > gpurun_out/r04f/fuzz.log
for rec in 1 0; do for view in headline interior; do
  ( IDKPT_INSTANCE_RECORDS=$rec timeout 600 python tools/bench_multi.py 1000000 3 $view > gpurun_out/r04f/multi_${view}_rec$rec.json 2> gpurun_out/r04f/multi_${view}_rec$rec.txt )
done; done
for donor in 1 0; do
  ( SWEEP_TAG=r04f_d$donor SWEEP_OPT=SPLIT:0,2 SWEEP_BATCHES=1,3 SWEEP_DEPTHS=2 IDKPT_SPLIT_DONOR=$donor timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -14 ) > gpurun_out/r04f/sweep_split_donor$donor.txt
done
( SHARD_MODS=1,2,4,8 SHARD_BANDS=8 SHARD_OPTS="split=0;split=2;split=2,split_donor=0" timeout 900 python tools/shard_small_batch.py 8 20 2>&1 | tail -14 ) > gpurun_out/r04f/shard_split.txt
tail -n 6 gpurun_out/r04f/tests.log; cat gpurun_out/r04f/fuzz.log; for f in gpurun_out/r04f/multi_*.txt; do echo "== $f"; cat $f; done; cat gpurun_out/r04f/sweep_split_donor1.txt gpurun_out/r04f/sweep_split_donor0.txt gpurun_out/r04f/shard_split.txt
