#!/bin/bash
# round 4, GPU call AA: band exchange (exact deep paths with the balanced deal), random API with kernel options, the multi / instances suites
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04aa
( timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_zz_random_api.py tests/test_gpu_quad.py tests/test_gpu_boundary.py -q -m gpu --maxfail=6 2>&1 | tail -15 ) > gpurun_out/r04aa/tests.log
( timeout 600 python tools/fuzz_parity.py 200 19000 2>&1 | grep -v ": OK" | tail -8 ) > gpurun_out/r04aa/fuzz.log
IDKPT_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 8 --warmup 2 --depth 5 --exact-deep-paths --no-extras --no-cpu-baseline --no-pmc --repeats 2 > gpurun_out/r04aa/ranks2_exact_d5.json 2> gpurun_out/r04aa/ranks2_exact_d5.err
tail -6 gpurun_out/r04aa/tests.log; cat gpurun_out/r04aa/fuzz.log; tail -c 1500 gpurun_out/r04aa/ranks2_exact_d5.json; tail -3 gpurun_out/r04aa/ranks2_exact_d5.err
