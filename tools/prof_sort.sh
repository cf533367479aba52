#!/bin/bash
# Developer experiment: does hit-triangle ordering make the bounce traversal faster?  depth 3, sort off vs on, kernel stats.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sortexp; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for S in 0 1; do
  SORT=$S timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$S -o t -- python tools/profile_frame.py 1000000 3 64 32 > $OUT/s$S.log 2>&1
  tail -1 $OUT/s$S.log
done
