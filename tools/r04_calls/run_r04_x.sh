#!/bin/bash
# round 4, GPU call X: any-hit queries on the scheduler: parity, rates
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04x
( timeout 900 python -m pytest tests/test_gpu_queries.py tests/test_gpu_glref.py tests/test_gpu_instances.py tests/test_gpu_multi.py tests/test_gpu_boundary.py tests/test_metamorphic.py -q -m gpu --maxfail=6 2>&1 | tail -15 ) > gpurun_out/r04x/tests.log
( timeout 600 python - <<'PY'
import json, sys
import torch
sys.path.insert(0, '.')
import bench
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer
torch.cuda.init()
sc = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1)
pt = PathTracer(bench.W, bench.H); pt.UploadScene(sc)
for name, cam in (("headline", S.Camera(bench.W, bench.H)), ("interior", S.Camera(bench.W, bench.H, position=(0.0, 0.0, 0.0)))):
    pt.SetCamera(cam)
    r = bench.query_extras(S, pt, sc, cam)
    print(name, json.dumps({k: v for k, v in r.items() if k != "workload"}))
PY
) > gpurun_out/r04x/queries.txt 2> gpurun_out/r04x/queries.err
tail -5 gpurun_out/r04x/tests.log; cat gpurun_out/r04x/queries.txt; tail -2 gpurun_out/r04x/queries.err
