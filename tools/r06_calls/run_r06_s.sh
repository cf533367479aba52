#!/bin/bash
# call s: the refit test of the unified tree, the random-API test, long fuzz runs on the final tree (default draw, multi-BLAS, same-space)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06s; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_zz_random_api.py tests/test_gpu_scene_updates.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
timeout 2400 python tools/fuzz_parity.py 1200 200000 > $O/fuzz_1200.log 2>&1; echo "rc $?" >> $O/fuzz_1200.log
tail -2 $O/fuzz_1200.log
FUZZ_BLASES=2,14 timeout 1500 python tools/fuzz_parity.py 500 210000 > $O/fuzz_blases_500.log 2>&1; echo "rc $?" >> $O/fuzz_blases_500.log
tail -2 $O/fuzz_blases_500.log
FUZZ_SAME_SPACE=1 FUZZ_BLASES=2,14 timeout 1500 python tools/fuzz_parity.py 500 220000 > $O/fuzz_same_space_500.log 2>&1; echo "rc $?" >> $O/fuzz_same_space_500.log
tail -2 $O/fuzz_same_space_500.log
