"""Secondary benchmark: what the headline costs with the engine's default kind of sky — an HDR cube map (SkyBoxManager.cs:44,74) — instead of a sky that is constant per
face.  The tile pre-classification (k_classify_tiles: whole 8x8 tiles proven to miss the scene get one flag byte and no ray) needs a per-face constant colour; with a textured
sky every culled pixel generates its ray, samples the cube map and stores its radiance (k_gen_primary's miss branch, FirstHit/compute.glsl:225-233).
usage: python tools/bench_sky.py [n_tris=1000000] [face=64]   (one line per sky; 32 samples in flight and one frame at a time)"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = 1920, 1080


def measure(pt, B):
    pt.set_max_batch(B)
    for _ in range(B):
        pt.Compute()
    pt.synchronize(); pt.reset_stats(); ts = []
    for _ in range(5):
        pt.ResetAccumulation(); t0 = time.perf_counter()
        for _ in range(2 * B):
            pt.Compute()
        pt.synchronize(); ts.append(time.perf_counter() - t0)
    rays = pt.stats()["rays_traced"] / 5.0
    return rays / statistics.median(ts) / 1e6


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    face = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    sc = S.soup_scene(n, NativeBuilder(), seed=1)
    cam = S.Camera(W, H)
    rng = np.random.default_rng(2)
    hdr = np.zeros((6, face, face, 4), np.float32); hdr[..., :3] = rng.uniform(0.0, 4.0, (6, face, face, 3)); hdr[..., 3] = 1.0
    for name, sky in (("constant per face (the bench's white sky)", sc.sky_faces.copy()), (f"HDR cube map, {face}x{face} texels per face", hdr)):
        sc.sky_faces = sky
        pt = PathTracer(W, H); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2
        print(f"{name:48s}: {measure(pt, 32):8.1f} Mray/s with 32 samples in flight, {measure(pt, 1):8.1f} one frame at a time", flush=True)
        pt.Dispose()
