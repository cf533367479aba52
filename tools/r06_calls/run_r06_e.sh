#!/bin/bash
# call e: partial re-braiding of the own TLAS (k_braid): the instance-TLAS tests under three budgets, the configscale atrium-87 case, tools/bench_braid.py --quick
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_instances.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_configscale.py -x -q -k "atrium" > $O/tests_atrium.log 2>&1; echo "rc $?" >> $O/tests_atrium.log
tail -3 $O/tests_atrium.log
timeout 1500 python tools/bench_braid.py --quick > $O/bench_braid_quick.json 2> $O/bench_braid_quick.err; echo "rc $?"
cat $O/bench_braid_quick.err | tail -20
