#!/bin/bash
# round 4, GPU call AC: the headline under an HDR cube-map sky; the group selftest with explicit bands beyond RayDepth 2
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ac
( timeout 600 python tools/bench_sky.py 2>&1 | grep -v "amdgpu.ids" | tail -4 ) > gpurun_out/r04ac/sky.txt
( timeout 300 python tools/scale_selftest.py --gpus 2 2>&1 | grep "selftest" | tail -10 ) > gpurun_out/r04ac/selftest_group.txt
cat gpurun_out/r04ac/sky.txt gpurun_out/r04ac/selftest_group.txt
