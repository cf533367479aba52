#!/bin/bash
# round 4, GPU call S: device-pointer queries (test + rates), the suites that the fused / scatter / adv_min changes touch
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04s
( timeout 900 python -m pytest tests/test_gpu_queries.py tests/test_gpu_fused.py tests/test_gpu_split.py tests/test_gpu_defer.py tests/test_gpu_boundary.py -q -m gpu --maxfail=6 2>&1 | tail -15 ) > gpurun_out/r04s/tests.log
( timeout 600 python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer
sc = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1); cam = S.Camera(bench.W, bench.H)
pt = PathTracer(bench.W, bench.H); pt.UploadScene(sc); pt.SetCamera(cam)
print(json.dumps(bench.query_extras(S, pt, sc, cam)))
PY
) > gpurun_out/r04s/queries.json 2> gpurun_out/r04s/queries.err
tail -5 gpurun_out/r04s/tests.log; cat gpurun_out/r04s/queries.json; tail -3 gpurun_out/r04s/queries.err
