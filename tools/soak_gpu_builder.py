"""Developer tool: random fragment sets through idkptBuildBlasCore and libidkbvh's CPU core; every byte must agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from idkengine_amd.bvh import NativeBuilder, GpuBuilder
from idkengine_amd.pathtracer import PathTracer
from idkengine_amd import gputypes as T
nb = NativeBuilder(); pt = PathTracer(8, 8); gb = GpuBuilder(pt)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 120000), rng.integers(120000, 400000)]))
    kind = rng.integers(0, 4)
    if kind == 0: p = rng.uniform(-10, 10, (n, 1, 3)) + rng.uniform(-0.2, 0.2, (n, 3, 3))
    elif kind == 1: p = np.round(rng.uniform(-4, 4, (n, 3, 3)) * 2) / 2                      # grid-aligned: many equal keys and costs
    elif kind == 2: p = rng.normal(0, 1, (n, 1, 3)) ** 3 + rng.uniform(-0.01, 0.01, (n, 3, 3))   # clustered
    else: p = rng.uniform(-1, 1, (n, 3, 3)) * np.array([100.0, 1.0, 0.01])                     # anisotropic
    positions = p.astype(np.float32).reshape(-1, 3)
    tris = np.zeros(n, T.GpuBlasTriangle); tris["X"] = np.arange(n) * 3; tris["Y"] = tris["X"] + 1; tris["Z"] = tris["X"] + 2
    refit = bool(rng.integers(0, 2))
    boxes, cn, co = nb.core_arrays(positions, tris, refit)
    gn, go = gb.core_on_gpu(boxes)
    ok = gn.tobytes() == cn.tobytes() and go.tobytes() == co.tobytes()
    bad += not ok
    print(it, "n", n, "frags", len(boxes), "kind", int(kind), "refit", refit, "levels", gb.last_levels, "OK" if ok else "MISMATCH", flush=True)
print("mismatches", bad)
