#!/bin/bash
# call u: singular / NaN transforms (the PLOC guard), with a hard timeout so that a hang cannot outlive the call
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_inst_tlas.py -x -q -k "singular" > $O/tests_singular.log 2>&1; echo "rc $?" >> $O/tests_singular.log
tail -15 $O/tests_singular.log
timeout 900 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_versions.py tests/test_gpu_scene_updates.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
