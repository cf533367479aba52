#!/bin/bash
# round 5, GPU call s: refill / parked-leaf thresholds of the primary launch under the pixel-major list (developer library)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05s; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( IDKPT_LIB_PATH=$GRAFT_REPO_ROOT/idkengine_amd/libidkpt_dev.so timeout 600 python tools/sweep_refill_pm.py 2>&1 | tail -20 ) > $OUT/sweep_refill_pm.log
cat $OUT/sweep_refill_pm.log
