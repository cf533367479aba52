IDKBVH_TIMING=1 IDKPT_BVH_TIMING=1 python - <<'PY' 2>&1 | tail -40
import time, torch
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder, GpuBuilder
from idkengine_amd.pathtracer import PathTracer
pt = PathTracer(64, 64)
b = GpuBuilder(pt)
for i in range(3):
    t=time.time(); sc=S.soup_scene(1000000,b,seed=1); print("scene total", time.time()-t, flush=True)
print("---- cpu", flush=True)
b = NativeBuilder()
for i in range(2):
    t=time.time(); sc=S.soup_scene(1000000,b,seed=1); print("scene total", time.time()-t, flush=True)
PY
