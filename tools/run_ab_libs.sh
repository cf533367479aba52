#!/bin/bash
# A/B of whole library builds on one box (developer tool): tools/ab/*.so (git-ignored, travel with gpurun), each through tools/sweep_r03.py with the
# default options.  usage: bash tools/run_ab_libs.sh <tag> lib1.so lib2.so ...   -> gpurun_out/<tag>/ab_<lib>.log
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  IDKPT_LIB_PATH=$PWD/$LIB SWEEP_ONLY_REF=1 SWEEP_TAG=$TAG/$NAME timeout 400 python tools/sweep_r03.py > gpurun_out/$TAG/ab_$NAME.log 2>&1
  echo "== $NAME"; grep "Mray/s" gpurun_out/$TAG/ab_$NAME.log
done
