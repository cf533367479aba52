"""Developer tool (GPU box), round 3: derived node order (IDKPT_NODE_LAYOUT / IDKPT_TREELET_DEPTH) x trace order of the bounce launches
(IDKPT_TRACE_ORDER), per view, RayDepth and batch size; every configuration is compared bit for bit (image + complete ray state) with the
reference order / queue order run of the same view.  Usage: python tools/sweep_r03.py [views...]   -> gpurun_out/<SWEEP_TAG>/sweep_r03.json"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402

CONFIGS = [("reference order, queue order", {"IDKPT_NODE_LAYOUT": 0, "IDKPT_TRACE_ORDER": 0}),
           ("couples depth-first, queue order", {"IDKPT_NODE_LAYOUT": 1, "IDKPT_TRACE_ORDER": 0}),
           ("couples treelets(3), queue order", {"IDKPT_NODE_LAYOUT": 2, "IDKPT_TREELET_DEPTH": 3, "IDKPT_TRACE_ORDER": 0}),
           ("reference order, trace order", {"IDKPT_NODE_LAYOUT": 0, "IDKPT_TRACE_ORDER": 2}),
           ("couples depth-first, trace order", {"IDKPT_NODE_LAYOUT": 1, "IDKPT_TRACE_ORDER": 2}),
           ("couples treelets(3), trace order", {"IDKPT_NODE_LAYOUT": 2, "IDKPT_TREELET_DEPTH": 3, "IDKPT_TRACE_ORDER": 2})]
if os.environ.get("SWEEP_ONLY_REF") == "1":      # A/B of whole libraries (IDKPT_LIB_PATH): the default configuration only
    CONFIGS = [("default options", {})]
if os.environ.get("SWEEP_OPT"):                  # one library option swept: SWEEP_OPT=GRAB_UNIT_LOG2:10,13,16 (run length of the work-list slices), TRACE_WAVES:8,12,24 ...
    _name, _vals = os.environ["SWEEP_OPT"].split(":")
    CONFIGS = [(f"{_name.lower()} {v}", {"IDKPT_" + _name: int(v)}) for v in _vals.split(",")]
DEPTHS = [int(d) for d in os.environ.get("SWEEP_DEPTHS", "2,5").split(",")]
BATCHES = [int(b) for b in os.environ.get("SWEEP_BATCHES", "32,1").split(",")]

if __name__ == "__main__":
    names = sys.argv[1:] or ["headline", "interior", "atrium"]
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1) if any(n != "atrium" for n in names) else None
    atrium = S.atrium_scene(1000000, NativeBuilder()) if "atrium" in names else None
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    report = {}
    for vname in names:
        sc, cam = views[vname]
        for depth in DEPTHS:
            for batch in BATCHES:
                frames = (96 if batch > 1 else 30) if depth <= 2 else (64 if batch > 1 else 20)
                ref = None
                for label, env in CONFIGS:
                    if batch == 1 and env.get("IDKPT_TRACE_ORDER") and env.get("IDKPT_NODE_LAYOUT") == 2:
                        continue
                    r, img, rays = run(sc, cam, 100, batch, frames, depth=depth, env=env)
                    if ref is None:
                        ref = (img, rays); par = "ref"
                    else:
                        par = bool((img.view(np.uint32) == ref[0].view(np.uint32)).all() and rays.tobytes() == ref[1].tobytes())
                    r["parity"] = par
                    report[f"{vname}/d{depth}/b{batch}/{label}"] = r
                    print(f"{vname:9s} depth {depth} batch {batch:2d} {label:36s}: {r['mray_s']:8.1f} Mray/s  {r['ms_per_frame']:.3f} ms/frame  trace {r['trace_ms_per_frame']:.3f} ms/frame  parity {par}", flush=True)
    out = os.path.join("gpurun_out", os.environ.get("SWEEP_TAG", "r03a")); os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "sweep_r03.json"), "w"), indent=1)
