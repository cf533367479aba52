"""Developer tool (one GPU): BASELINE.json configs[3] — the multi-GPU config of north_star (soup-4M, 3840x2160, RayDepth 9) — dealt over N = 8 GPUs in bands of 8 rows
and in contiguous strips: every shard is timed alone with the same number of samples in flight; whole-frame time / slowest shard = what N GPUs reach on the rendering
alone (the deep-path exchange, idkptSetBandExchange / idkptSetBounceExchange, is a few hundred uint32 per bounce).  Both deals are exact at this depth (round 4).
usage: python tools/shard_config4.py [n_tris=4000000] [N=8] [batch=8]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import DeviceBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H, DEPTH = 3840, 2160, 9


def run(sc, cam, batch, **kw):
    strip = kw.pop("strip", None)
    pt = PathTracer(W, H, **kw)
    if strip:
        pt.SetRowRange(*strip)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = DEPTH; pt.set_max_batch(batch)
    for _ in range(batch):
        pt.Compute()
    pt.synchronize(); pt.reset_stats(); ts = []
    for _ in range(3):
        pt.ResetAccumulation(); t0 = time.perf_counter()
        for _ in range(batch):
            pt.Compute()
        pt.synchronize(); ts.append((time.perf_counter() - t0) / batch)
    rays = pt.stats()["rays_traced"] / (3 * batch)
    pt.Dispose()
    return statistics.median(ts), rays


if __name__ == "__main__":
    n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    bpt = PathTracer(8, 8); sc = S.soup_scene(n_tris, DeviceBuilder(bpt), seed=1); bpt.Dispose()
    cam = S.Camera(W, H)
    t1, r1 = run(sc, cam, batch)
    print(f"soup-{n_tris}, {W}x{H}, RayDepth {DEPTH}, {batch} samples in flight: whole frame {t1*1e3:.3f} ms/step, {r1/t1/1e6:.0f} Mray/s", flush=True)
    tb = [run(sc, cam, batch, row_modulo=n, row_remainder=r, row_band=8)[0] for r in range(n)]
    print(f"  bands of 8 rows, (y // 8) % {n}: shards {' '.join(f'{t*1e3:.3f}' for t in tb)} ms; sum {sum(tb)*1e3:.3f} ({t1/sum(tb):.3f} of ideal), slowest {max(tb)*1e3:.3f} -> {t1/max(tb):.2f}x = {r1/max(tb)/1e6:.0f} Mray/s", flush=True)
    base, extra = divmod(H, n)
    ts = [run(sc, cam, batch, strip=(r * base + min(r, extra), base + (1 if r < extra else 0)))[0] for r in range(n)]
    print(f"  contiguous strips: shards {' '.join(f'{t*1e3:.3f}' for t in ts)} ms; sum {sum(ts)*1e3:.3f} ({t1/sum(ts):.3f} of ideal), slowest {max(ts)*1e3:.3f} -> {t1/max(ts):.2f}x = {r1/max(ts)/1e6:.0f} Mray/s", flush=True)
