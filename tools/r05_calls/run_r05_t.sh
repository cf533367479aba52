#!/bin/bash
# round 5, GPU call t: the first bounce traced in the primary list's pixel-major order (bounce_pixel_major): A/B, batch tests, fuzz
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05t; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( AB_OPTION=bounce_pixel_major timeout 600 python tools/ab_pixel_major.py 2>&1 | tail -8 ) > $OUT/ab_bounce_pm.log
( IDKPT_GEN_PIXEL_MAJOR=2 timeout 900 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_versions.py tests/test_gpu_parity.py tests/test_gpu_inst_tlas.py tests/test_gpu_wide.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4 ) > $OUT/tests_pm.log
( timeout 600 python tools/fuzz_parity.py 250 120000 2>&1 | grep -v ": OK" | tail -3 ) > $OUT/fuzz_250.log
cat $OUT/ab_bounce_pm.log; tail -3 $OUT/tests_pm.log; cat $OUT/fuzz_250.log
