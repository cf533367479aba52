#!/bin/bash
# round 4, GPU call AM: refill threshold of the instance-loop / TLAS kernels, 8 / 12 / 16 against 32, headline and interior views
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04am
export IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so
for view in headline interior; do for v in 0 8 12 16; do
  ( IDKPT_TRACE_VARIANT=$v timeout 400 python tools/bench_multi.py 1000000 3 $view 2>&1 >/dev/null | grep -v one_blas | sed "s/^/$view refill variant $v: /" ) >> gpurun_out/r04am/multi_refill.txt
done; done
cat gpurun_out/r04am/multi_refill.txt
