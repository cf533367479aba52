#!/bin/bash
# round 4, GPU call AO: leaf_min of the instance-loop / TLAS kernels (option, no rebuild): 8 / 12 / 20 / 24 against the default 16
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ao
for v in 0 8 12 20 24; do
  ( IDKPT_LEAF_MIN=$v timeout 400 python tools/bench_multi.py 1000000 3 headline 2>&1 >/dev/null | grep -v one_blas | sed "s/^/leaf_min $v: /" ) >> gpurun_out/r04ao/multi_leaf_min.txt
done
cat gpurun_out/r04ao/multi_leaf_min.txt
