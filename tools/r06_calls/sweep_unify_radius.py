"""call i: PLOC search radius of the unified tree's top (option inst_unify_radius) on the atrium as 87 BLASes and the 3-part same-space soup."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
import bench_braid
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer
W, H, B = bench.W, bench.H, 32
pt = PathTracer(W, H)
for name, sc, cam in (("atrium_87", S.atrium_scene(bench.N_TRIS, NativeBuilder(), per_mesh_blas=True), S.atrium_camera(W, H)), ("soup3_interior", bench_braid.same_space_soup(3), bench.view_camera(S, "interior", W, H))):
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = bench.RAY_DEPTH
    for budget in (1024, 4096):
        for radius in (15, 40, 100, 250):
            pt.set_option("inst_unify", budget); pt.set_option("inst_unify_radius", radius)
            m = bench_braid.measure(pt, B); st = pt.stats()
            print(json.dumps({"scene": name, "budget": budget, "radius": radius, "mray_s": m["mray_s"], "single": m["single_frame_mray_s"], "entries": st["inst_unified_entries"], "depth": st["inst_unified_top_depth"]}), flush=True)
pt.Dispose()
