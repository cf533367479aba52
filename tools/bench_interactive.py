"""Secondary benchmark: the interactive regime — every frame has a new camera and its own image (ResetAccumulation per frame in the
reference).  F frames in flight through the frame ring (idkptSetFrameRing(2F), idkptSetMaxBatch(F)); F = 1 is the reference's
one-frame-at-a-time schedule.  Usage: python tools/bench_interactive.py [n_tris=1000000] [frames=256]"""
import os, sys, time, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    W, H = 1920, 1080
    sc = S.soup_scene(n, NativeBuilder(), seed=1)
    cams = [S.Camera(W, H, position=(2.0 * math.sin(0.02 * k), 1.0 * math.sin(0.013 * k), 25.0)) for k in range(64)]   # a slowly moving camera
    for F in (1, 2, 4, 8, 16, 32):
        pt = PathTracer(W, H); pt.UploadScene(sc); pt.RayDepth = 2
        pt.SetFrameRing(max(2, 2 * F)); pt.set_max_batch(F)
        def run(count):
            for k in range(count):
                pt.BeginFrame(); pt.SetCamera(cams[k % len(cams)]); pt.Compute()
            pt.synchronize()
        run(2 * F + 4); pt.reset_stats()
        t0 = time.perf_counter(); run(frames); dt = time.perf_counter() - t0
        st = pt.stats()
        print({"frames_in_flight": F, "ms_per_frame": round(dt / frames * 1e3, 3), "frames_per_s": round(frames / dt), "Mray_per_s": round(st["rays_traced"] / dt / 1e6, 1),
               "latency_ms_until_a_frame_is_complete": round(F * dt / frames * 1e3, 2)})
        pt.Dispose()
