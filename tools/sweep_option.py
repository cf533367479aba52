"""One developer option swept over the bench views (1920x1080, RayDepth 2, 32 samples in flight unless the view says otherwise); bit-identical frames asserted.
usage: python tools/sweep_option.py <option> v1,v2,... [view ...] [-- other_option=value ...]     views: headline headline_20 interior atrium"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

args = sys.argv[1:]
extra = {}
if "--" in args:
    k = args.index("--"); extra = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args[k + 1:]}; args = args[:k]
opt, values, want = args[0], [int(v) for v in args[1].split(",")], set(args[2:])
W, H = bench.W, bench.H
pt = PathTracer(W, H); pt.enable_timing(True)
for k, v in extra.items():
    pt.set_option(k, v)
soup = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1); atrium = S.atrium_scene(bench.N_TRIS, NativeBuilder())
rows = (("headline", soup, bench.view_camera(S, "headline", W, H), 2, 32), ("headline_20", soup, bench.view_camera(S, "headline", W, H), 2, 20), ("interior", soup, bench.view_camera(S, "interior", W, H), 2, 32),
        ("atrium", atrium, S.atrium_camera(W, H), 2, 32))
for name, sc, cam, depth, B in rows:
    if want and name not in want:
        continue
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth
    row = {}; ref = None
    for v in values + values:
        pt.set_option(opt, v)
        rays, dt = bench.timed_batch(pt, B, max(B, 16), reps=4)
        img = np.ascontiguousarray(pt.Result).view(np.uint32)
        if ref is None: ref = img.copy()
        assert (img == ref).all(), "frames differ"
        row.setdefault(str(v), []).append(round(rays / dt / 1e6, 1))
    print(json.dumps({name: row}), flush=True)
pt.Dispose()
