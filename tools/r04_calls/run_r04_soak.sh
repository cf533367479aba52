#!/bin/bash
# round 4, last soak on the final tree: 400 random API sequences (the deferring side draws the split / fused / quad / park / pooled kernels), 1500 fuzz seeds, 300 at RayDepth 2
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04soak
( IDKPT_RANDOM_API_SEEDS=400 timeout 1500 python -m pytest tests/test_gpu_zz_random_api.py -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r04soak/random_api_400.log
( timeout 2400 python tools/fuzz_parity.py 1500 30000 2>&1 | grep -v ": OK" | tail -6 ) > gpurun_out/r04soak/fuzz_1500.log
( FUZZ_DEPTH=2 timeout 900 python tools/fuzz_parity.py 300 32000 2>&1 | grep -v ": OK" | tail -6 ) > gpurun_out/r04soak/fuzz_d2_300.log
tail -2 gpurun_out/r04soak/random_api_400.log; cat gpurun_out/r04soak/fuzz_1500.log gpurun_out/r04soak/fuzz_d2_300.log
