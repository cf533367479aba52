#!/bin/bash
# round 5, GPU call g: after the persistent group workers and the wide-node clean-up — multi-device / wide / transport tests, the whole suite, long soaks
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $OUT/gpu_suite.log
( timeout 400 python tools/scale_selftest.py --gpus 3 2>&1 | grep "selftest" | tail -5 ) > $OUT/selftest_group3.txt
( timeout 2400 python tools/fuzz_parity.py 2000 50000 2>&1 | grep -v ": OK" | tail -6 ) > $OUT/fuzz_2000.log
( for s in $(seq 0 7); do RAPI_FIRST=$((s*50)) timeout 600 python -m pytest tests/test_gpu_zz_random_api.py -q 2>&1 | tail -1; done ) > $OUT/random_api.log
tail -3 $OUT/gpu_suite.log; cat $OUT/selftest_group3.txt; cat $OUT/fuzz_2000.log; cat $OUT/random_api.log
