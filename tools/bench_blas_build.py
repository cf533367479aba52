"""Developer tool (GPU box): whole-BLAS build time of the three builders — libidkbvh on the host cores (NativeBuilder), host + SweepSAH core on the
GPU (GpuBuilder, round 2), everything on the GPU (DeviceBuilder, idkptBuildBlas) — on the bench scenes, with a byte comparison of the outputs.
Usage: python tools/bench_blas_build.py [tris=1000000] [reps=5]      IDKPT_BVH_TIMING=1 prints the phases of the device build"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder, GpuBuilder, DeviceBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402


class Capture:
    def __init__(self, inner):
        self.inner, self.calls = inner, []

    def build_blas(self, positions, tris, refittable):
        self.calls.append((np.array(positions, np.float32), np.array(tris), bool(refittable)))
        return self.inner.build_blas(positions, tris, refittable)

    def __getattr__(self, k):
        return getattr(self.inner, k)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    pt = PathTracer(8, 8)
    nb = NativeBuilder()
    for scene, make in (("soup", lambda b: S.soup_scene(n, b, seed=1)), ("soup refittable", lambda b: S.soup_scene(n, b, seed=1, refittable=True)), ("atrium", lambda b: S.atrium_scene(n, b))):
        cap = Capture(nb); make(cap)
        positions, tris, refittable = cap.calls[0]
        ref = nb.build_blas(positions, tris, refittable)
        for name, b in (("host (libidkbvh)", nb), ("host + GPU core", GpuBuilder(pt)), ("device (idkptBuildBlas)", DeviceBuilder(pt))):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); r = b.build_blas(positions, tris, refittable); ts.append((time.perf_counter() - t0) * 1e3)
            same = all(r[k].tobytes() == ref[k].tobytes() for k in ("nodes", "triangles", "parents", "leaves")) and r["required_stack_size"] == ref["required_stack_size"]
            print(f"{scene:16s} {len(tris):8d} tris -> {r['fragments']:8d} fragments, {len(r['nodes']):8d} nodes | {name:24s}: median {sorted(ts)[len(ts) // 2]:8.2f} ms  min {min(ts):8.2f} ms  "
                  f"(device part {getattr(b, 'last_device_ms', float('nan')):7.2f} ms)  identical to the host build: {same}", flush=True)
    pt.Dispose()
