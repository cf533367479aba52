#!/bin/bash
# kernel-trace stats only. Usage: tools/prof_trace.sh <tag> <ntris> <depth> <batch> <frames>
TAG=${1:-p}; NT=${2:-1000000}; DEPTH=${3:-2}; BATCH=${4:-1}; FR=${5:-16}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/profile_frame.py $NT $DEPTH $FR $BATCH > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cut -c1-60,200-400 "$F" | head -24
grep -h "ms/frame" $OUT/*.log
