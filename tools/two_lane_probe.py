"""Developer tool (GPU box): how much of a batch's non-traversal work (ray generation, shading, compaction: ~19 % of a 32-sample headline batch) can hide
behind ANOTHER batch's traversal launches?  Two contexts of one process on one GPU, each on its own stream, each rendering whole frames in batches of
`batch` samples; their batches are issued alternately without a host synchronisation, so that one context's k_shade / k_gen_primary can run while the
other's persistent k_trace2 occupies the chip — if the traversal grid leaves room (option trace_waves: 24 resident waves per CU take 450 of a SIMD's
512 VGPRs, 20 leave room for one shading wave per SIMD).  Compared with ONE context rendering the same number of samples.
usage: python tools/two_lane_probe.py [view=headline|interior|atrium] [batch=32] [batches=6]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = 1920, 1080


def make(sc, cam, batch, waves, depth):
    p = PathTracer(W, H)
    if waves:
        p.set_option("trace_waves", waves)
    p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = depth; p.set_max_batch(batch)
    return p


def region(pts, batch, batches):
    for p in pts:
        p.ResetAccumulation()
    for b in range(batches):
        for p in pts:
            for _ in range(batch):
                p.Compute()              # the batch-th call launches the whole batch asynchronously on the context's stream
    for p in pts:
        p.flush()
    for p in pts:
        p.synchronize()


if __name__ == "__main__":
    view = sys.argv[1] if len(sys.argv) > 1 else "headline"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    batches = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    depth = int(os.environ.get("PROBE_DEPTH", "2"))
    if view == "atrium":
        sc = S.atrium_scene(1000000, NativeBuilder()); cam = S.atrium_camera(W, H)
    else:
        sc = S.soup_scene(1000000, NativeBuilder(), seed=1); cam = S.Camera(W, H) if view == "headline" else S.Camera(W, H, position=(0.0, 0.0, 0.0))
    for contexts, waves in ((1, 0), (1, 20), (2, 0), (2, 20), (2, 16), (2, 12)):
        pts = [make(sc, cam, batch, waves, depth) for _ in range(contexts)]
        per = batches // contexts      # batches per context: the same total number of samples in every configuration
        for _ in range(2):
            region(pts, batch, per)
        for p in pts:
            p.reset_stats()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); region(pts, batch, per); ts.append(time.perf_counter() - t0)
        rays = sum(p.stats()["rays_traced"] for p in pts) / 5
        med = sorted(ts)[2]
        print(f"{view} depth {depth}: {contexts} context(s), trace_waves {waves or 'default'}, {per} batches of {batch} each: {med * 1e3:8.3f} ms, {rays / med / 1e6:8.1f} Mray/s", flush=True)
        for p in pts:
            p.Dispose()
