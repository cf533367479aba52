#!/bin/bash
# round 4, GPU call AD: tile pre-classification with a textured sky (class 8): parity (whole suite + fuzz: a third of the fuzz cases have cube-map skies), bench_sky
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ad
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r04ad/gpu_suite.log
( timeout 900 python tools/fuzz_parity.py 400 23000 2>&1 | grep -v ": OK" | tail -6 ) > gpurun_out/r04ad/fuzz.log
( timeout 600 python tools/bench_sky.py 2>&1 | grep -v "amdgpu.ids" | tail -4 ) > gpurun_out/r04ad/sky.txt
tail -3 gpurun_out/r04ad/gpu_suite.log; cat gpurun_out/r04ad/fuzz.log gpurun_out/r04ad/sky.txt
