"""Round 5 (developer library): refill threshold of the primary launch under the pixel-major list (trace_variant 911-914 = 16 / 24 / 48 / 8; shipped: 32) and the parked-leaf
threshold (leaf_min), 32 samples in flight."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402

soup = S.soup_scene(1000000, NativeBuilder(), seed=1)
views = {"headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
CONFIGS = [("refill 32 (shipped)", 100, {}), ("refill 8", 914, {}), ("refill 16", 911, {}), ("refill 24", 912, {}), ("refill 48", 913, {}), ("refill 32 again", 100, {})] + [(f"leaf_min {l}", 100, {"IDKPT_LEAF_MIN": l}) for l in (12, 20, 24)]
for vname, (sc, cam) in views.items():
    ref = None
    for label, variant, env in CONFIGS:
        r, img, rays = run(sc, cam, variant, 32, 96, env=env)
        if ref is None: ref = (img, rays); par = "ref"
        else: par = bool((img.view(np.uint32) == ref[0].view(np.uint32)).all() and rays.tobytes() == ref[1].tobytes())
        print(f"{vname:9s} {label:22s}: {r['mray_s']:8.1f} Mray/s  trace {r['trace_ms_per_launch']:.3f} ms/launch  parity {par}", flush=True)
