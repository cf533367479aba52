#!/bin/bash
# call g: soak of the unified tree (fuzz seeds drawing same-space scenes of 2-14 BLASes, and the default draw with the new options), the full tools/bench_braid.py
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06g; mkdir -p $O; cd $GRAFT_REPO_ROOT
FUZZ_SAME_SPACE=1 FUZZ_BLASES=2,14 timeout 1500 python tools/fuzz_parity.py 500 40000 > $O/fuzz_same_space_500.log 2>&1; echo "rc $?" >> $O/fuzz_same_space_500.log
tail -2 $O/fuzz_same_space_500.log; grep -c "unified launches [1-9]" $O/fuzz_same_space_500.log
timeout 1200 python tools/fuzz_parity.py 300 50000 > $O/fuzz_300.log 2>&1; echo "rc $?" >> $O/fuzz_300.log
tail -2 $O/fuzz_300.log
FUZZ_BLASES=2,14 timeout 1200 python tools/fuzz_parity.py 300 60000 > $O/fuzz_blases_300.log 2>&1; echo "rc $?" >> $O/fuzz_blases_300.log
tail -2 $O/fuzz_blases_300.log
timeout 2400 python tools/bench_braid.py > $O/bench_braid.json 2> $O/bench_braid.err; echo "rc $?"
