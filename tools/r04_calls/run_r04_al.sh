#!/bin/bash
# round 4, GPU call AL: refill threshold of the instance-loop / TLAS kernels (developer build: trace_variant 16 / 24 / 48 against 32)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04al
export IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so
for v in 0 16 24 48; do
  ( IDKPT_TRACE_VARIANT=$v timeout 400 python tools/bench_multi.py 1000000 3 headline 2>&1 >/dev/null | grep -v one_blas | sed "s/^/refill variant $v: /" ) >> gpurun_out/r04al/multi_refill.txt
done
cat gpurun_out/r04al/multi_refill.txt
