#!/bin/bash
# round 4, GPU call AI: the multi-device suites with the members enqueued by one host thread each at every depth (group_threads 1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ai
( IDKPT_GROUP_THREADS=1 timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_versions.py tests/test_gpu_boundary.py -q -m gpu --maxfail=4 2>&1 | grep "passed\|failed\|Error" | tail -5 ) > gpurun_out/r04ai/tests_threads1.log
( IDKPT_GROUP_THREADS=1 timeout 300 python tools/scale_selftest.py --gpus 4 2>&1 | grep "selftest" | tail -8 ) > gpurun_out/r04ai/selftest_group4_threads1.txt
cat gpurun_out/r04ai/tests_threads1.log gpurun_out/r04ai/selftest_group4_threads1.txt
