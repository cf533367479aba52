#!/bin/bash
# round 5, GPU call q: samples per group of the pixel-major primary list (32 = two passes per wave), and the tests that run batches
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 600 python tools/ab_group_max.py 2>&1 | tail -4 ) > $OUT/ab_group_max.log
( IDKPT_GEN_PIXEL_MAJOR=2 timeout 900 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_versions.py tests/test_gpu_parity.py tests/test_gpu_inst_tlas.py -x -q -m gpu 2>&1 | tail -3 ) > $OUT/tests_pm.log
( timeout 600 python tools/fuzz_parity.py 200 99000 2>&1 | grep -v ": OK" | tail -3 ) > $OUT/fuzz_200.log
cat $OUT/ab_group_max.log; tail -2 $OUT/tests_pm.log; cat $OUT/fuzz_200.log
