"""Developer tool (GPU box): the wide-node walk (option wide = 1, kernels_wide.hpp) against k_trace2 alone (wide = 0) on the three bench views, 32 samples in
flight and one frame at a time; checks that both give the same image and ray state bit for bit and reports the rays handed to the exact kernel.
usage: python tools/wide_ab.py [tris] [extra IDKPT_NAME=value ...]   -> gpurun_out/wide_ab.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402


def counts(sc, cam, batch):
    os.environ["IDKPT_WIDE_COUNT"] = "1"
    pt = PathTracer(W, H); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2; pt.set_max_batch(batch)
    for _ in range(batch):
        pt.Compute()
    pt.synchronize(); st = pt.stats(); pt.Dispose()
    os.environ.pop("IDKPT_WIDE_COUNT", None)
    return {k: st[k] / batch for k in ("wide_node_visits", "wide_leaf_records", "wide_triangle_tests", "wide_flagged_rays", "rays_traced")}


if __name__ == "__main__":
    tris = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000000
    extra = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    soup = S.soup_scene(tris, NativeBuilder(), seed=1); atrium = S.atrium_scene(tris, NativeBuilder())
    views = {"headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0))), "atrium": (atrium, S.atrium_camera(W, H))}
    report = {}
    for vname, (sc, cam) in views.items():
        for batch, frames in ((32, 96), (1, 40)):
            res = {}
            for wide in (0, 1):
                env = dict(extra); env["IDKPT_WIDE"] = wide
                r, img, rays = run(sc, cam, 0, batch, frames, env=env)
                res[wide] = (r, img, rays)
            same = res[0][1].tobytes() == res[1][1].tobytes() and res[0][2].tobytes() == res[1][2].tobytes()
            c = counts(sc, cam, min(batch, 4))
            row = {"off": res[0][0], "on": res[1][0], "speedup": res[1][0]["mray_s"] / res[0][0]["mray_s"], "trace_speedup": res[0][0]["trace_ms_per_frame"] / max(1e-9, res[1][0]["trace_ms_per_frame"]), "bit_identical": same, "per_frame": c}
            report[f"{vname}_b{batch}"] = row
            print(f"{vname:9s} batch {batch:2d}: k_trace2 {res[0][0]['mray_s']:8.1f} Mray/s (trace {res[0][0]['trace_ms_per_frame']:.3f} ms/frame) | wide {res[1][0]['mray_s']:8.1f} Mray/s (trace {res[1][0]['trace_ms_per_frame']:.3f} ms/frame) "
                  f"x{row['speedup']:.3f} (trace x{row['trace_speedup']:.3f}) | identical {same} | per frame: {c['wide_node_visits']:.0f} node visits, {c['wide_leaf_records']:.0f} leaf records, {c['wide_triangle_tests']:.0f} triangle tests, {c['wide_flagged_rays']:.1f} flagged of {c['rays_traced']:.0f} rays", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open("gpurun_out/wide_ab.json", "w"), indent=1)
