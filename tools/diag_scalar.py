"""Developer tool (GPU box): where does the trace time of a view go (primary launch alone = RayDepth 1, both launches = RayDepth 2), what do the
instrumented builds count (node steps, stepping lanes, steps served through the scalar cache), per IDKPT_TRACE_VARIANT.
Usage: python tools/diag_scalar.py [atrium|interior|headline ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = 1920, 1080


def run(sc, cam, variant, depth, batch=32, frames=64):
    os.environ["IDKPT_TRACE_VARIANT"] = str(variant)
    pt = PathTracer(W, H)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth; pt.set_max_batch(batch)
    for _ in range(batch):
        pt.Compute()
    pt.synchronize(); pt.reset_stats(); pt.enable_timing(True); pt.ResetAccumulation()
    t0 = time.perf_counter()
    for _ in range(frames):
        pt.Compute()
    pt.synchronize()
    dt = time.perf_counter() - t0
    sys.stderr.flush()
    st = pt.stats()        # (the instrumented variants print their counters to stderr here)
    pt.Dispose()
    return dt / frames * 1e3, st["trace_ms_total"] / frames, st["rays_traced"] / frames


if __name__ == "__main__":
    names = sys.argv[1:] or ["atrium", "interior", "headline"]
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1) if any(n != "atrium" for n in names) else None
    atrium = S.atrium_scene(1000000, NativeBuilder()) if "atrium" in names else None
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    for n in names:
        sc, cam = views[n]
        for depth in (1, 2):
            for v in (100, 113):
                ms, tr, rays = run(sc, cam, v, depth)
                print(f"{n:9s} depth {depth} variant {v:3d}: {ms:.3f} ms/frame  trace {tr:.3f} ms/frame  rays/frame {rays:.0f}", flush=True)
