"""Developer tool (GPU box): round trip of the work-counter atomic as the waves see it (instrumented build, IDKPT_TRACE_VARIANT=113), per view,
batch size and reservation policy.  Usage: python tools/diag_grab.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402

if __name__ == "__main__":
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1)
    atrium = S.atrium_scene(1000000, NativeBuilder())
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    for vname, (sc, cam) in views.items():
        for batch, frames in ((32, 64), (1, 32)):
            for depth in (1, 2):
                for label, env in (("legacy", {"IDKPT_GRAB_MAX": 0}), ("fixed128", {"IDKPT_GRAB_FIXED": 128}), ("adaptive", {})):
                    print(f"-- {vname} batch {batch} depth {depth} {label}", flush=True)
                    sys.stdout.flush()
                    r, _, _ = run(sc, cam, 113, batch, frames, depth=depth, env=env)
                    print(f"   {r['mray_s']:.1f} Mray/s trace {r['trace_ms_per_frame']:.3f} ms/frame", flush=True)
