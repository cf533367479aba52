#!/bin/bash
# round 4, GPU call AB: interleaved rows exact at any depth inside one multi-device context (threaded members + band exchange)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ab
( timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_versions.py tests/test_gpu_boundary.py -q -m gpu --maxfail=6 2>&1 | tail -15 ) > gpurun_out/r04ab/tests.log
( timeout 300 python tools/scale_selftest.py --gpus 2 2>&1 | grep "selftest" | tail -8 ) > gpurun_out/r04ab/selftest_group.txt
tail -6 gpurun_out/r04ab/tests.log; cat gpurun_out/r04ab/selftest_group.txt
