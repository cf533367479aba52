#!/bin/bash
# call q: k_trace2 FAST (regrouped node pairs): the parity suites, A B A B against the reference's layout
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06q; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_defer.py tests/test_gpu_fullsize.py tests/test_gpu_scene_updates.py tests/test_gpu_versions.py tests/test_gpu_boundary.py tests/test_gpu_worklist.py tests/test_gpu_nocounters.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
FUZZ_BLASES=1,1 timeout 1200 python tools/fuzz_parity.py 300 130000 > $O/fuzz_one_blas_300.log 2>&1; echo "rc $?" >> $O/fuzz_one_blas_300.log
tail -2 $O/fuzz_one_blas_300.log
timeout 1200 python tools/ab_option.py pair_nodes 0 1 > $O/ab_pair_nodes.log 2>&1; echo "rc $?" >> $O/ab_pair_nodes.log
cat $O/ab_pair_nodes.log | cut -c1-600
