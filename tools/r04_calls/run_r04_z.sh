#!/bin/bash
# round 4, GPU call Z: two parked leaves per lane (k_trace2p): parity on the suites that render frames, then A/B against the plain and the pooled kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z
( IDKPT_PARK=7 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_fused.py tests/test_gpu_defer.py tests/test_gpu_glref.py -q -m gpu --maxfail=6 2>&1 | tail -8 ) > gpurun_out/r04z/tests_park.log
( IDKPT_PARK=7 timeout 600 python tools/fuzz_parity.py 150 18000 2>&1 | grep -v ": OK" | tail -8 ) > gpurun_out/r04z/fuzz_park.log
( IDKPT_FUSED=0 IDKPT_LEAF_POOL=0 SWEEP_TAG=r04z SWEEP_OPT=PARK:0,7,1,2 SWEEP_BATCHES=32,1 SWEEP_DEPTHS=2,5 timeout 1200 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -50 ) > gpurun_out/r04z/sweep_park_vs_plain.txt
( IDKPT_FUSED=0 SWEEP_TAG=r04z2 SWEEP_OPT=PARK:0,7 SWEEP_BATCHES=32 SWEEP_DEPTHS=2 timeout 900 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -10 ) > gpurun_out/r04z/sweep_park_vs_pooled.txt
tail -4 gpurun_out/r04z/tests_park.log; cat gpurun_out/r04z/fuzz_park.log gpurun_out/r04z/sweep_park_vs_plain.txt gpurun_out/r04z/sweep_park_vs_pooled.txt
