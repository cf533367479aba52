#!/bin/bash
# round 4, GPU call AF: AUTO = bands at every RayDepth (group contexts): the multi / versions / boundary suites, both selftests
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04af
( timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_versions.py tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_gpu_configscale.py -q -m gpu --maxfail=6 2>&1 | tail -12 ) > gpurun_out/r04af/tests.log
( timeout 300 python tools/scale_selftest.py --gpus 2 2>&1 | grep "selftest" | tail -10 ) > gpurun_out/r04af/selftest_group.txt
( timeout 300 python tools/scale_selftest.py --gpus 3 2>&1 | grep "selftest" | tail -10 ) > gpurun_out/r04af/selftest_group3.txt
timeout 200 python bench.py --gpus 2 --steps 16 --warmup 4 --depth 5 --no-extras --no-cpu-baseline --no-pmc --repeats 3 > gpurun_out/r04af/group2_d5.json 2> gpurun_out/r04af/group2_d5.err
tail -5 gpurun_out/r04af/tests.log; cat gpurun_out/r04af/selftest_group.txt gpurun_out/r04af/selftest_group3.txt; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04af/group2_d5.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['n_gpu']['selftest'])"
