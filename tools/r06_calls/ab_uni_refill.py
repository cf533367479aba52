import json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools")
import bench, bench_braid
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer
W, H, B = bench.W, bench.H, 32
pt = PathTracer(W, H)
for name, sc, cam in (("atrium_87", S.atrium_scene(bench.N_TRIS, NativeBuilder(), per_mesh_blas=True), S.atrium_camera(W, H)), ("soup3_one_space_interior", bench_braid.same_space_soup(3), bench.view_camera(S, "interior", W, H))):
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2
    for v in (16, 32, 16, 32):
        pt.set_option("uni_refill", v)
        m = bench_braid.measure(pt, B)
        print(json.dumps({"scene": name, "uni_refill": v, "mray_s": m["mray_s"], "single": m["single_frame_mray_s"]}), flush=True)
