"""Developer tool: N single frames (idkptSetMaxBatch(1); ResetAccumulation, Compute, Synchronize) of the headline workload, for
`rocprofv3 --kernel-trace --stats` (per-kernel share of the one-frame-at-a-time latency).  Prints wall time per frame."""
import os
import sys
import time
import statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = 1920, 1080
view = sys.argv[2] if len(sys.argv) > 2 else "headline"      # headline | interior | atrium
if view == "atrium":
    sc = S.atrium_scene(1000000, NativeBuilder()); cam = S.atrium_camera(W, H)
else:
    sc = S.soup_scene(1000000, NativeBuilder(), seed=1); cam = S.Camera(W, H) if view == "headline" else S.Camera(W, H, position=(0.0, 0.0, 0.0))
pt = PathTracer(W, H); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2
pt.set_max_batch(1)
if os.environ.get("SFP_NO_TIMING"):
    pt.enable_timing(False) if hasattr(pt, "enable_timing") else None
for _ in range(5):
    pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
ts = []
for _ in range(frames):
    t0 = time.perf_counter(); pt.ResetAccumulation(); pt.Compute(); pt.synchronize(); ts.append(time.perf_counter() - t0)
print(f"single frame: median {statistics.median(ts) * 1e3:.4f} ms, min {min(ts) * 1e3:.4f} ms over {frames} frames; rays/frame {pt.stats()['rays_traced'] / (frames + 5):.0f}")
