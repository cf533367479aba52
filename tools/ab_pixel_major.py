"""Round 5 experiment: k_gen_primary's append order (option gen_pixel_major): a traversal wave = one 8x8 tile of one sample (0) or 4 pixels x 16 samples (n: from batches of n samples on).
Headline / interior / atrium views, 32 samples in flight; bit-identical frames are asserted."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H, B = bench.W, bench.H, 32
pt = PathTracer(W, H); pt.enable_timing(True)
res = {}
soup = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1)
atrium = S.atrium_scene(bench.N_TRIS, NativeBuilder())
for name, sc, cam, depth, B in (("headline", soup, bench.view_camera(S, "headline", W, H), 2, 32), ("headline_20_samples", soup, bench.view_camera(S, "headline", W, H), 2, 20), ("interior", soup, bench.view_camera(S, "interior", W, H), 2, 32),
                               ("interior_20_samples", soup, bench.view_camera(S, "interior", W, H), 2, 20), ("atrium", atrium, S.atrium_camera(W, H), 2, 32), ("interior_d5", soup, bench.view_camera(S, "interior", W, H), 5, 32)):
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth
    row = {}; ref = None
    which = os.environ.get("AB_OPTION", "gen_pixel_major")            # AB_OPTION=bounce_pixel_major: the first bounce's list (0 / 1) under the pixel-major primary list
    for opt in ((0, 2, 0, 2) if which == "bounce_pixel_major" else (0, 8, 0, 8)):
        pt.set_option(which, opt)
        rays, dt = bench.timed_batch(pt, B, B, reps=5)
        st = pt.stats()
        img = np.ascontiguousarray(pt.Result).view(np.uint32)
        if ref is None: ref = img.copy()
        assert (img == ref).all(), "frames differ"
        row.setdefault(str(opt), []).append({"mray_s": round(rays / dt / 1e6, 1), "trace_ms_per_launch": round(st["trace_ms_total"] / max(st["trace_launches"], 1), 4)})
    res[name] = row
    print(json.dumps({name: row}), flush=True)
pt.Dispose()
