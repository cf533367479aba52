#!/bin/bash
# call f: the unified tree of same-space multi-instance scenes (k_braid + k_unify_* + k_trace_inst<.., UNI>): its tests, the instance suites, tools/bench_braid.py --quick
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -15 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_configscale.py tests/test_gpu_multi.py tests/test_gpu_boundary.py -x -q > $O/tests_more.log 2>&1; echo "rc $?" >> $O/tests_more.log
tail -3 $O/tests_more.log
timeout 1500 python tools/bench_braid.py --quick > $O/bench_braid_quick.json 2> $O/bench_braid_quick.err; echo "rc $?"
cat $O/bench_braid_quick.err | tail -20
