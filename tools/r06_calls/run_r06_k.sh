#!/bin/bash
# call k: the packet walk over the unified tree: tests, same-space fuzz, bench_braid --quick
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_inst_tlas.py tests/test_gpu_packet.py -x -q > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -3 $O/tests.log
FUZZ_SAME_SPACE=1 FUZZ_BLASES=2,14 timeout 1200 python tools/fuzz_parity.py 400 90000 > $O/fuzz_same_space_400.log 2>&1; echo "rc $?" >> $O/fuzz_same_space_400.log
tail -2 $O/fuzz_same_space_400.log
timeout 1500 python tools/bench_braid.py --quick > $O/bench_braid_quick.json 2> $O/bench_braid_quick.err; echo "rc $?"
grep "^{" $O/bench_braid_quick.err | cut -c1-1500
