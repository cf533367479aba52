# quick check of a kernel change on the GPU box: parity suite, then the three views with the feature off / on (developer tool)
mkdir -p gpurun_out/r02s
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -i "passed\|failed\|error" | tail -3
for C in 0 1; do echo "== IDKPT_CULL_POPS=$C"; IDKPT_CULL_POPS=$C SWEEP_OUT=/tmp/y.json timeout 200 python tools/sweep_trace.py 100 2>&1; done | tee gpurun_out/r02s/cull.log
