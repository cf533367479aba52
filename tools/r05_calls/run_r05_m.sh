#!/bin/bash
# round 5, GPU call m: experiment — primary work list pixel-major over 16 samples (gen_pixel_major) vs tile-major
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 900 python tools/ab_pixel_major.py 2>&1 | tail -12 ) > $OUT/ab_pixel_major.log
( IDKPT_GEN_PIXEL_MAJOR=2 timeout 900 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_versions.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 ) > $OUT/tests_pm.log
cat $OUT/ab_pixel_major.log; tail -3 $OUT/tests_pm.log
