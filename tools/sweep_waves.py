"""Developer tool (GPU box): size of k_trace2's persistent grid (IDKPT_TRACE_WAVES one-wave workgroups per CU) and LDS padding per workgroup.
Usage: python tools/sweep_waves.py [views...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from sweep_trace import run, W, H  # noqa: E402

if __name__ == "__main__":
    names = sys.argv[1:] or ["atrium", "headline", "interior"]
    soup = S.soup_scene(1000000, NativeBuilder(), seed=1) if any(n != "atrium" for n in names) else None
    atrium = S.atrium_scene(1000000, NativeBuilder()) if "atrium" in names else None
    views = {"atrium": (atrium, S.atrium_camera(W, H)), "headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0)))}
    for vname in names:
        sc, cam = views[vname]
        for batch, frames in ((32, 96), (1, 40)):
            for variant in (100,):
                for waves, pad in ((12, 0), (16, 0), (20, 0), (22, 0), (24, 0), (26, 0), (28, 0), (30, 0), (32, 0), (40, 0), (24, 1536), (32, 1536)):
                    r, _, _ = run(sc, cam, variant, batch, frames, env={"IDKPT_TRACE_WAVES": waves, "IDKPT_LDS_PAD": pad})
                    print(f"{vname:9s} batch {batch:2d} variant {variant} waves/CU {waves:2d} pad {pad:4d}: {r['mray_s']:8.1f} Mray/s  {r['ms_per_frame']:.3f} ms/frame  trace {r['trace_ms_per_frame']:.3f} ms/frame", flush=True)
