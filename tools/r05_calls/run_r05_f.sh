#!/bin/bash
# round 5, GPU call f: A/B of library builds on one box (tools/ab_libs.py): base (commit before), z slabs of both boxes packed, + pooled kernel forced to 6 waves per SIMD
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1500 python tools/ab_libs.py --rounds 2 tools/ab/base.so tools/ab/zpack.so tools/ab/zpack_pool6.so 2>&1 | grep -v "^\[W\|amdgpu.ids" | tee $OUT/ab.log
cp gpurun_out/ab_libs.json $OUT/
