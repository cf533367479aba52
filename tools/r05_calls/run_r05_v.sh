#!/bin/bash
# round 5, GPU call v: after the last one-line change (the sparse-view test per sample of the previous batch) — batch / sample / parity tests and a short fuzz
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05v; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bench.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) > $OUT/tests.log
( timeout 200 python tools/fuzz_parity.py 60 140000 2>&1 | grep -v ": OK" | tail -3 ) > $OUT/fuzz_60.log
cat $OUT/tests.log $OUT/fuzz_60.log
