# A/B on one box: the library built from an older commit against the working tree's (developer tool).  Build the old one first, e.g.:
#   mkdir -p /tmp/ab tools/ab && git archive 15f8033 idkengine_amd/csrc include | tar -x -C /tmp/ab && (cd /tmp/ab/idkengine_amd/csrc && \
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -o $REPO/tools/ab/libidkpt_head.so idkpt.hip)
# (idkengine_amd/_lib.py honours IDKPT_LIB_PATH; tools/ab/*.so is git-ignored but travels with gpurun)
mkdir -p gpurun_out/r02s
( echo "== lib head"; IDKPT_LIB_PATH=$PWD/tools/ab/libidkpt_head.so SWEEP_OUT=/tmp/x.json timeout 200 python tools/sweep_trace.py 100
  echo "== lib new"; SWEEP_OUT=/tmp/y.json timeout 200 python tools/sweep_trace.py 100 ) > gpurun_out/r02s/ab.log 2>&1
cat gpurun_out/r02s/ab.log
