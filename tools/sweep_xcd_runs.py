import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from sweep_trace import run, W, H
soup = S.soup_scene(1000000, NativeBuilder(), seed=1); atrium = S.atrium_scene(1000000, NativeBuilder())
views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
for vname, (sc, cam) in views.items():
    for depth in (2, 5):
        for sort in (0, 1):
            for unit in (10, 16, 20, 23):
                r, _, _ = run(sc, cam, 100, 32, 64, depth=depth, sort=sort, env={"IDKPT_GRAB_UNIT_LOG2": unit})
                print(f"{vname:9s} depth {depth} sort {sort} unit 2^{unit:2d}: {r['mray_s']:8.1f} Mray/s  trace {r['trace_ms_per_frame']:.3f} ms/frame", flush=True)
