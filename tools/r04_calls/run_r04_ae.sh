#!/bin/bash
# round 4, GPU call AE: BASELINE configs[3] (soup-4M, 4K, RayDepth 9) dealt over 8 GPUs in bands vs strips, every shard timed alone
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ae
( timeout 900 python tools/shard_config4.py 4000000 8 32 2>&1 | grep -v "amdgpu.ids" | tail -5 ) > gpurun_out/r04ae/shard_config4.txt
cat gpurun_out/r04ae/shard_config4.txt
