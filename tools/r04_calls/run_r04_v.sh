#!/bin/bash
# round 4, GPU call V: the whole GPU suite after the query scheduler change + fuzz
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04v
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > gpurun_out/r04v/gpu_suite.log
( timeout 900 python tools/fuzz_parity.py 200 13000 2>&1 | grep -v ": OK" | tail -8 ) > gpurun_out/r04v/fuzz.log
( FUZZ_DEPTH=2 timeout 600 python tools/fuzz_parity.py 100 14000 2>&1 | grep -v ": OK" | tail -8 ) > gpurun_out/r04v/fuzz_d2.log
tail -6 gpurun_out/r04v/gpu_suite.log; cat gpurun_out/r04v/fuzz.log gpurun_out/r04v/fuzz_d2.log
