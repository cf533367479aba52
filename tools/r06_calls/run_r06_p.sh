#!/bin/bash
# call p: full-size soak of the round's instance-scene paths against k_trace2's loop (GPU against GPU), and a longer fuzz of the default draw
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06p; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 3000 python tools/soak_inst_tlas.py 48 > $O/soak_inst.log 2>&1; echo "rc $?" >> $O/soak_inst.log
cat $O/soak_inst.log | cut -c1-3000
timeout 1800 python tools/fuzz_parity.py 800 120000 > $O/fuzz_800.log 2>&1; echo "rc $?" >> $O/fuzz_800.log
tail -2 $O/fuzz_800.log
