mkdir -p gpurun_out/r02s
SWEEP_OUT=/tmp/y.json timeout 200 python tools/sweep_trace.py 100 2>&1 | tee gpurun_out/r02s/trim.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -i "passed\|failed\|error" | tail -3
