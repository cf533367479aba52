#!/bin/bash
# Developer tool: rocprofv3 kernel stats + two PMC passes for the headline frame.  Usage: tools/prof.sh <tag> [n_tris] [depth]
TAG=${1:-p}; NT=${2:-1000000}; DEPTH=${3:-2}; BATCH=${4:-1}; FR=${5:-16}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/profile_frame.py $NT $DEPTH $FR $BATCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o c -- python tools/profile_frame.py $NT $DEPTH $BATCH $BATCH > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o c -- python tools/profile_frame.py $NT $DEPTH $BATCH $BATCH > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc3 -o c -- python tools/profile_frame.py $NT $DEPTH $BATCH $BATCH > $OUT/pmc3.log 2>&1
find $OUT -name "*.csv" | head -20
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -20 "$F"
grep -h "ms/frame" $OUT/*.log
