"""Developer tool (one GPU): what does dealing the rows y % N cost in ray coherence?  Times the N row shards of the headline frame one after
another (each with N x 32 samples in flight = as many rays per launch as the whole frame at 32) and compares their sum with the whole frame."""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer

W, H = 1920, 1080


def run(sc, cam, mod, rem, batch, strip=None):
    pt = PathTracer(W, H, row_modulo=mod, row_remainder=rem)
    if strip:
        pt.SetRowRange(*strip)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2; pt.set_max_batch(batch)
    for _ in range(batch):
        pt.Compute()
    pt.synchronize(); pt.reset_stats()
    ts = []
    for _ in range(5):
        pt.ResetAccumulation()
        t0 = time.perf_counter()
        for _ in range(batch):
            pt.Compute()
        pt.synchronize()
        ts.append((time.perf_counter() - t0) / batch)
    rays = pt.stats()["rays_traced"] / (5 * batch)
    pt.Dispose()
    return statistics.median(ts), rays


if __name__ == "__main__":
    sc = S.soup_scene(1000000, NativeBuilder(), seed=1)
    for vname, cam in (("headline", S.Camera(W, H)), ("interior", S.Camera(W, H, position=(0.0, 0.0, 0.0)))):
        t1, r1 = run(sc, cam, 1, 0, 32)
        print(f"{vname}: whole frame, 32 in flight: {t1*1e3:.4f} ms/step, {r1/t1/1e6:.0f} Mray/s", flush=True)
        for n in (2, 4, 8):
            tot = 0.0; worst = 0.0
            for r in range(n):
                t, rays = run(sc, cam, n, r, min(256, 32 * n)); tot += t; worst = max(worst, t)
            print(f"  rows y%{n}: sum of the {n} shards {tot*1e3:.4f} ms/step ({t1/tot:.3f} of ideal), slowest shard {worst*1e3:.4f} ms -> {n} GPUs would run at {r1/worst/1e6:.0f} Mray/s = {t1/worst:.2f}x", flush=True)
            base, extra = divmod(H, n)
            tots = 0.0; worsts = 0.0
            for r in range(n):
                first = r * base + min(r, extra); cnt = base + (1 if r < extra else 0)
                t, rays = run(sc, cam, 1, 0, min(256, 32 * n), strip=(first, cnt)); tots += t; worsts = max(worsts, t)
            print(f"  strips  : sum {tots*1e3:.4f} ms/step ({t1/tots:.3f} of ideal), slowest strip {worsts*1e3:.4f} ms -> {t1/worsts:.2f}x", flush=True)
