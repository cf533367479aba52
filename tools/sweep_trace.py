"""Developer tool (GPU box): time the traversal-kernel scheduling variants (IDKPT_TRACE_VARIANT) on the headline frame and on the
interior view, at 32 samples in flight and one frame at a time, and check every variant's image / ray state against k_trace2's
(variant 100).  Usage: python tools/sweep_trace.py [variants...]   -> gpurun_out/sweep_trace.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = 1920, 1080


def run(sc, cam, variant, batch, frames, depth=2, sort=0, env=None):
    os.environ["IDKPT_TRACE_VARIANT"] = str(variant)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    pt = PathTracer(W, H)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth; pt.DoRaySorting = sort; pt.set_max_batch(batch)
    for _ in range(max(batch, 4)):
        pt.Compute()
    pt.synchronize(); pt.reset_stats(); pt.enable_timing(True); pt.ResetAccumulation()
    t0 = time.perf_counter()
    if batch == 1:
        for _ in range(frames):
            pt.ResetAccumulation(); pt.Compute(); pt.synchronize()      # one frame at a time
    else:
        for _ in range(frames):
            pt.Compute()
        pt.synchronize()
    dt = time.perf_counter() - t0
    st = pt.stats()
    img = pt.Result; rays = pt.rays()
    pt.Dispose()
    for k in (env or {}):          # (some knobs are read at launch time, not at context creation)
        os.environ.pop(k, None)
    return {"ms_per_frame": dt / frames * 1e3, "mray_s": st["rays_traced"] / dt / 1e6, "trace_ms_per_launch": st["trace_ms_total"] / max(1, st["trace_launches"]),
            "trace_ms_per_frame": st["trace_ms_total"] / frames, "rays_per_frame": st["rays_traced"] / frames}, img, rays


if __name__ == "__main__":
    variants = [int(v) for v in sys.argv[1:]] or [100, 208, 216, 232, 308, 316, 332, 408]
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1)
    atrium = S.atrium_scene(1000000, NativeBuilder())
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    report = {}
    for vname, (sc, cam) in views.items():
        ref = {}
        for batch, frames in ((32, 96), (1, 40)):
            for v in variants:
                r, img, rays = run(sc, cam, v, batch, frames)
                key = (batch,)
                if v == variants[0]:
                    ref[key] = (img, rays)
                    r["parity"] = "ref"
                else:
                    r["parity"] = bool((img.view(np.uint32) == ref[key][0].view(np.uint32)).all() and rays.tobytes() == ref[key][1].tobytes())
                report[f"{vname}/b{batch}/v{v}"] = r
                print(f"{vname:9s} batch {batch:2d} variant {v:3d}: {r['mray_s']:8.1f} Mray/s  {r['ms_per_frame']:.3f} ms/frame  trace {r['trace_ms_per_frame']:.3f} ms/frame  parity {r['parity']}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open(os.environ.get("SWEEP_OUT", "gpurun_out/sweep_trace.json"), "w"), indent=1)
