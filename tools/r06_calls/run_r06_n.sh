#!/bin/bash
# call n: the general unified array (k_trace_inst TREE 2): the instance-TLAS tests under the "general_array" parameter, multi-BLAS fuzz, tools/bench_braid.py --quick (rotated parts)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06n; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
FUZZ_BLASES=2,14 timeout 1500 python tools/fuzz_parity.py 400 110000 > $O/fuzz_blases_400.log 2>&1; echo "rc $?" >> $O/fuzz_blases_400.log
tail -2 $O/fuzz_blases_400.log; grep -c "unified launches [1-9]" $O/fuzz_blases_400.log
timeout 1500 python tools/bench_braid.py --quick > $O/bench_braid_quick.json 2> $O/bench_braid_quick.err; echo "rc $?"
grep "^{" $O/bench_braid_quick.err | grep rotated | cut -c1-2500
