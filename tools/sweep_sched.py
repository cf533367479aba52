"""Developer tool (GPU box): scheduling thresholds of k_trace2 (IDKPT_TRACE_VARIANT 9xx) and the size of its persistent grid / LDS padding
(IDKPT_TRACE_WAVES, IDKPT_LDS_PAD), per view and batch size.  Usage: python tools/sweep_sched.py [views...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402

CONFIGS = [("default", 100, {})] + [(f"leafMin {l}", 100, {"IDKPT_LEAF_MIN": l}) for l in (8, 12, 16, 20, 24)] + [("R32 L20 (template)", 901, {}), ("R40 L16 (template)", 902, {})] + \
          [(f"grid {w} waves/CU", 100, {"IDKPT_TRACE_WAVES": w}) for w in (12, 16, 20, 24, 28, 40)]
SORT = int(os.environ.get("SWEEP_SORT", 0))

if __name__ == "__main__":
    names = sys.argv[1:] or ["atrium", "headline", "interior"]
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1) if any(n != "atrium" for n in names) else None
    atrium = S.atrium_scene(1000000, NativeBuilder()) if "atrium" in names else None
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    for vname in names:
        sc, cam = views[vname]
        for batch, frames in ((32, 96), (1, 40)):
            ref = None
            for label, variant, env in CONFIGS:
                r, img, rays = run(sc, cam, variant, batch, frames, env=env, sort=SORT)
                if ref is None:
                    ref = (img, rays); par = "ref"
                else:
                    par = bool((img.view(np.uint32) == ref[0].view(np.uint32)).all() and rays.tobytes() == ref[1].tobytes())
                print(f"{vname:9s} batch {batch:2d} {label:40s}: {r['mray_s']:8.1f} Mray/s  {r['ms_per_frame']:.3f} ms/frame  trace {r['trace_ms_per_frame']:.3f} ms/frame  parity {par}", flush=True)
