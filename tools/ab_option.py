"""A B A B of one developer option on the bench views (1920x1080, soup-1M / atrium-1M, RayDepth 2, 32 and 20 samples in flight), bit-identical frames asserted.
usage: python tools/ab_option.py <option> <valueA> <valueB> [view ...]      views: headline headline_20 interior atrium headline_one_frame interior_d5"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

opt, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
want = set(sys.argv[4:])
W, H = bench.W, bench.H
pt = PathTracer(W, H); pt.enable_timing(True)
soup = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1); atrium = S.atrium_scene(bench.N_TRIS, NativeBuilder())
rows = (("headline", soup, bench.view_camera(S, "headline", W, H), 2, 32), ("headline_20", soup, bench.view_camera(S, "headline", W, H), 2, 20), ("interior", soup, bench.view_camera(S, "interior", W, H), 2, 32),
        ("interior_d5", soup, bench.view_camera(S, "interior", W, H), 5, 32), ("atrium", atrium, S.atrium_camera(W, H), 2, 32), ("headline_one_frame", soup, bench.view_camera(S, "headline", W, H), 2, 1))
for name, sc, cam, depth, B in rows:
    if want and name not in want:
        continue
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth
    row = {}; ref = None
    for v in (a, b, a, b):
        pt.set_option(opt, v)
        rays, dt = bench.timed_batch(pt, B, max(B, 16), reps=5)
        st = pt.stats()
        img = np.ascontiguousarray(pt.Result).view(np.uint32)
        if ref is None: ref = img.copy()
        assert (img == ref).all(), "frames differ"
        row.setdefault(f"{opt}={v}", []).append({"mray_s": round(rays / dt / 1e6, 1), "trace_ms_per_launch": round(st["trace_ms_total"] / max(st["trace_launches"], 1), 4)})
    print(json.dumps({name: row}), flush=True)
pt.Dispose()
