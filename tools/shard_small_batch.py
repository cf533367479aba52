"""Developer tool (GPU box): what ONE rank of an N-GPU strong-scaling run does — the driver's 20 steps on 1/N of the frame's rows (bands of SHARD_BANDS rows,
(y // band) % N == r) — timed on one GPU, with the scheduling options that matter for small launches.
usage: [SHARD_SCENE=soup|atrium] [SHARD_BANDS=1,8] [SHARD_OPTS=grid|none|"k=v,k=v;k=v"] [SHARD_MODS=1,2,4,8] python tools/shard_small_batch.py [N=8] [steps=20]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = 1920, 1080

if __name__ == "__main__" and os.environ.get("SHARD_TWO") != "1":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if os.environ.get("SHARD_SCENE", "soup") == "atrium":      # the Sponza-class hall (every pixel traverses, chains a third as long): the same projection beside the soup's
        sc = S.atrium_scene(1000000, NativeBuilder()); cam = S.atrium_camera(W, H)
    else:
        sc = S.soup_scene(1000000, NativeBuilder(), seed=1); cam = S.Camera(W, H)
    mods = [int(v) for v in os.environ["SHARD_MODS"].split(",")] if os.environ.get("SHARD_MODS") else [1, N]
    bands = [int(v) for v in os.environ.get("SHARD_BANDS", "8").split(",")]
    for mod, band in [(m, b) for m in mods for b in (bands if m > 1 else bands[:1])]:
        sets = ({}, {"leaf_min": 12}, {"leaf_min": 16}, {"leaf_min": 20}, {"grid_hint": 0}, {"trace_waves": 16}, {"trace_waves": 32})
        so = os.environ.get("SHARD_OPTS", "none")
        if so == "grid":
            sets = ({}, {"grid_rays_x4": 0}, {"grid_rays_x4": 8}, {"grid_rays_x4": 12}, {"grid_rays_x4": 16}, {"grid_rays_x4": 24}, {"trace_waves": 20}, {"trace_waves": 16}, {"trace_waves": 12})
        elif so == "none":
            sets = ({},)
        elif so != "sched":
            sets = tuple({kv.split("=")[0]: int(kv.split("=")[1]) for kv in grp.split(",") if kv} for grp in so.split(";"))
        for opts in sets:
            pt = PathTracer(W, H, row_modulo=mod, row_remainder=0, row_band=band)
            for k, v in opts.items():
                pt.set_option(k, v)
            pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2; pt.set_max_batch(min(256, 32 * mod))
            for _ in range(2):
                pt.ResetAccumulation()
                for _ in range(steps):
                    pt.Compute()
                pt.synchronize()
            pt.reset_stats(); ts = []
            for _ in range(9):
                pt.ResetAccumulation()
                t0 = time.perf_counter()
                for _ in range(steps):
                    pt.Compute()
                pt.image_device_ptr(0); pt.synchronize()
                ts.append(time.perf_counter() - t0)
            rays = pt.stats()["rays_traced"] / 9
            med = sorted(ts)[4]
            print(f"rows (y // {band}) % {mod} == 0, {steps} steps, options {opts}: {med * 1e3:7.3f} ms per region, {rays / med / 1e6:8.1f} Mray/s on this GPU -> x{mod} = {rays * mod / med / 1e6:9.1f} Mray/s if every rank took as long", flush=True)
            pt.Dispose()


def two_contexts(N, steps, sc, cam, parts=2):
    """The same shard (rows y % N == 0) rendered by `parts` contexts of one process on one GPU, each on its own stream with rows y % (parts*N) == k*N: do the
    launch tails of one overlap with the bulk of the other?"""
    pts = []
    for k in range(parts):
        p = PathTracer(W, H, row_modulo=parts * N, row_remainder=k * N); p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 2; p.set_max_batch(256); pts.append(p)

    def region():
        for p in pts:
            p.ResetAccumulation()
        for _ in range(steps):
            for p in pts:
                p.Compute()
        for p in pts:
            p.flush()
        for p in pts:
            p.image_device_ptr(0)
        for p in pts:
            p.synchronize()
    for _ in range(2):
        region()
    for p in pts:
        p.reset_stats()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); region(); ts.append(time.perf_counter() - t0)
    rays = sum(p.stats()["rays_traced"] for p in pts) / 9
    med = sorted(ts)[4]
    print(f"rows y % {N} == 0 by {parts} contexts on one GPU, {steps} steps: {med * 1e3:7.3f} ms per region, {rays / med / 1e6:8.1f} Mray/s on this GPU -> x{N} = {rays * N / med / 1e6:9.1f} Mray/s", flush=True)
    for p in pts:
        p.Dispose()


if __name__ == "__main__" and os.environ.get("SHARD_TWO") == "1":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sc = S.soup_scene(1000000, NativeBuilder(), seed=1); cam = S.Camera(W, H)
    for n in (1, 2, 4, 8):
        for parts in (1, 2, 3, 4):
            two_contexts(n, steps, sc, cam, parts)
