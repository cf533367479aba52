#!/bin/bash
# round 4, GPU call T: device-resident query rates; the whole GPU suite on the current tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04t
( timeout 600 python - <<'PY'
import json, sys
import torch
sys.path.insert(0, '.')
import bench
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer
torch.cuda.init()
sc = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1); cam = S.Camera(bench.W, bench.H)
pt = PathTracer(bench.W, bench.H); pt.UploadScene(sc); pt.SetCamera(cam)
print(json.dumps(bench.query_extras(S, pt, sc, cam)))
cam2 = S.Camera(bench.W, bench.H, position=(0.0, 0.0, 0.0)); pt.SetCamera(cam2)
print(json.dumps(bench.query_extras(S, pt, sc, cam2)))
PY
) > gpurun_out/r04t/queries.json 2> gpurun_out/r04t/queries.err
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > gpurun_out/r04t/gpu_suite.log
cat gpurun_out/r04t/queries.json; tail -3 gpurun_out/r04t/queries.err; tail -6 gpurun_out/r04t/gpu_suite.log
