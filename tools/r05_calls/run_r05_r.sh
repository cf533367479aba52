#!/bin/bash
# round 5, GPU call r: full-size soak of the instance-loop kernels against k_trace2's loop (GPU against GPU, every checkpoint bit for bit)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 800 python tools/soak_inst_tlas.py 200 2>&1 | tail -6 ) > $OUT/soak_inst_tlas.log
cat $OUT/soak_inst_tlas.log
