"""Developer tool (GPU box): work-list chunk size of k_trace2 (IDKPT_GRAB_DIV / IDKPT_GRAB_MAX) and kernel variants, per view and batch size,
with a bit-wise parity check against the first configuration.  Usage: python tools/sweep_grab.py [views...]  -> gpurun_out/sweep_grab.json"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402

CONFIGS = [("default (8 slices, runs of 1024)", 100, {}),
           ("runs of 256", 100, {"IDKPT_GRAB_UNIT_LOG2": 8}),
           ("runs of 4096", 100, {"IDKPT_GRAB_UNIT_LOG2": 12}),
           ("reserve 128 per atomic", 100, {"IDKPT_GRAB_FIXED": 128})]

if __name__ == "__main__":
    names = sys.argv[1:] or ["atrium", "headline", "interior"]
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1) if any(n != "atrium" for n in names) else None
    atrium = S.atrium_scene(1000000, NativeBuilder()) if "atrium" in names else None
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    report = {}
    for vname in names:
        sc, cam = views[vname]
        for depth in (2,):
            for batch, frames in ((32, 96), (1, 40)):
                if depth == 5 and batch == 1:
                    continue
                ref = None
                for label, variant, env in CONFIGS:
                    r, img, rays = run(sc, cam, variant, batch, frames, depth=depth, env=env)
                    if ref is None:
                        ref = (img, rays); r["parity"] = "ref"
                    else:
                        r["parity"] = bool((img.view(np.uint32) == ref[0].view(np.uint32)).all() and rays.tobytes() == ref[1].tobytes())
                    report[f"{vname}/d{depth}/b{batch}/{label}"] = r
                    print(f"{vname:9s} depth {depth} batch {batch:2d} {label:34s}: {r['mray_s']:8.1f} Mray/s  {r['ms_per_frame']:.3f} ms/frame  trace {r['trace_ms_per_frame']:.3f} ms/frame  parity {r['parity']}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open(os.environ.get("SWEEP_OUT", "gpurun_out/sweep_grab.json"), "w"), indent=1)
