#!/bin/bash
# Final soak on the final kernel: GPU suite (log kept), randomised parity fuzz, random API sequences.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; grep -n "passed\|failed" $OUT/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
IDKPT_RANDOM_API_SEEDS=120 timeout 300 python -m pytest tests/test_gpu_zz_random_api.py -q 2>&1 | tail -1
timeout 400 python tools/fuzz_parity.py 1000 6000 > $OUT/fuzz.log 2>&1; tail -1 $OUT/fuzz.log
