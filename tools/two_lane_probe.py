"""Experiment: two contexts of one process on one GPU (own streams), each rendering half of the samples, batches interleaved — do the shading kernels of one hide behind the traversal of
the other when the persistent traversal grids leave wave slots free (option trace_waves)?  Against ONE context with all samples in flight.  Headline view, RayDepth 2.
usage: python tools/two_lane_probe.py [trace_waves ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = bench.W, bench.H
view = os.environ.get("VIEW", "headline")
sc = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1); cam = bench.view_camera(S, view, W, H)


def run(pts, B, steps, waves):
    for p in pts:
        p.set_option("trace_waves", waves); p.set_max_batch(B)
    def region():
        for p in pts:
            p.ResetAccumulation()
        for _ in range(steps):
            for p in pts:
                p.Compute()
        for p in pts:
            p.flush()
        for p in pts:
            p.synchronize()
    for _ in range(2):
        region()
    for p in pts:
        p.reset_stats()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); region(); ts.append(time.perf_counter() - t0)
    rays = sum(p.stats()["rays_traced"] for p in pts) / 7
    return round(rays / sorted(ts)[3] / 1e6, 1)


one = PathTracer(W, H); one.UploadScene(sc); one.SetCamera(cam); one.RayDepth = 2
two = [PathTracer(W, H) for _ in range(2)]
for p in two:
    p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 2
for waves in [int(v) for v in sys.argv[1:]] or [0, 16, 12, 10]:
    print(json.dumps({"view": view, "trace_waves": waves, "one_context_32_in_flight": run([one], 32, 32, waves), "one_context_16_in_flight": run([one], 16, 32, waves),
                      "two_contexts_16_each": run(two, 16, 16, waves)}), flush=True)
