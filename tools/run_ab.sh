# A/B on one box: the library built from HEAD (tools/ab/libidkpt_head.so) against the working tree's (developer tool)
mkdir -p gpurun_out/r02s
( echo "== lib head"; IDKPT_LIB_PATH=$PWD/tools/ab/libidkpt_head.so SWEEP_OUT=/tmp/x.json timeout 200 python tools/sweep_trace.py 100
  echo "== lib new"; SWEEP_OUT=/tmp/y.json timeout 200 python tools/sweep_trace.py 100 ) > gpurun_out/r02s/ab6.log 2>&1
cat gpurun_out/r02s/ab6.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02s/gpu_suite.log 2>&1; tail -15 gpurun_out/r02s/gpu_suite.log
