#!/bin/bash
# round 4, GPU call AH: one multi-device context, members enqueued by one host thread each also at RayDepth 2 (group_threads): 2 / 4 / 8 members sharing the GPU, the driver's command
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04ah
for n in 2 4 8; do for t in 0 1; do
  IDKPT_GROUP_THREADS=$t timeout 300 python bench.py --gpus $n --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc --repeats 9 > gpurun_out/r04ah/group${n}_threads$t.json 2>> gpurun_out/r04ah/err.log
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04ah/group${n}_threads$t.json') if l.startswith('{')][-1]); print('members $n threads $t:', d['value'], d['ms_per_step'], d['config']['n_gpu']['selftest']['bits_equal'])"
done; done
( IDKPT_GROUP_THREADS=1 timeout 600 python -m pytest tests/test_gpu_multi.py -q -m gpu --maxfail=4 2>&1 | tail -3 )
