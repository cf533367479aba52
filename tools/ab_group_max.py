"""Round 5 experiment: pixel-major primary list, samples per group (gen_group_max) 2 / 4 / 8 / 16 at 32 samples in flight."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402
W, H, B = bench.W, bench.H, 32
pt = PathTracer(W, H); pt.enable_timing(True)
soup = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1)
for name in ("headline", "interior"):
    pt.UploadScene(soup); pt.SetCamera(bench.view_camera(S, name, W, H)); pt.RayDepth = 2
    row = {}
    for g in (16, 8, 4, 2, 16, 8, 4, 2):
        pt.set_option("gen_group_max", g)
        rays, dt = bench.timed_batch(pt, B, B, reps=5)
        st = pt.stats()
        row.setdefault(str(g), []).append((round(rays / dt / 1e6, 1), round(st["trace_ms_total"] / max(st["trace_launches"], 1), 3)))
    print(json.dumps({name: row}), flush=True)
pt.Dispose()
