"""Developer tool: one-frame-at-a-time latency of small frames (launch-bound regime): Cornell 256x256 and 640x360, RayDepth 2 / 5."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer
sc = S.cornell_scene(NativeBuilder(), "mixed")
for (w, h, d) in ((256, 256, 2), (256, 256, 5), (640, 360, 5), (1920, 1080, 5)):
    pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(S.cornell_camera(w, h)); pt.RayDepth = d; pt.set_max_batch(1)
    for _ in range(20):
        pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); pt.ResetAccumulation(); pt.Compute(); pt.synchronize(); ts.append(time.perf_counter() - t0)
    st = pt.stats()
    print(f"{w}x{h} depth {d}: median {statistics.median(ts)*1e6:.1f} us/frame, {st['rays_traced']/220/statistics.median(ts)/1e6:.1f} Mray/s")
    pt.Dispose()
