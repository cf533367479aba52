#!/bin/bash
# round 4, GPU call AG: the device-side band exchange (idkptSetBandExchangeDevice): lockstep test, both selftests, the 2-rank bench with --exact-deep-paths
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ag
( timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_boundary.py -q -m gpu --maxfail=6 2>&1 | tail -12 ) > gpurun_out/r04ag/tests.log
( IDKPT_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/scale_selftest.py 2>&1 | grep "selftest\|Error\|error" | tail -8 ) > gpurun_out/r04ag/selftest_ranks.txt
IDKPT_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --steps 8 --warmup 2 --depth 5 --exact-deep-paths --no-extras --no-cpu-baseline --no-pmc --repeats 2 > gpurun_out/r04ag/ranks2_exact_d5.json 2> gpurun_out/r04ag/ranks2_exact_d5.err
tail -4 gpurun_out/r04ag/tests.log; cat gpurun_out/r04ag/selftest_ranks.txt; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04ag/ranks2_exact_d5.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['sharding'][:120], d['config']['n_gpu']['selftest'])"; tail -3 gpurun_out/r04ag/ranks2_exact_d5.err
