"""Developer tool: render N frames of the headline workload (for rocprofv3). Uses a cached scene file if present."""
import os, sys, time, pickle
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from idkengine_amd import scenes as S
from idkengine_amd.pathtracer import PathTracer

def get_scene(n):
    from idkengine_amd.bvh import NativeBuilder
    parts = int(os.environ.get("PARTS", "1"))
    if os.environ.get("VIEW") == "atrium":
        return S.atrium_scene(n, NativeBuilder())
    return S.soup_scene(n, NativeBuilder()) if parts == 1 else S.soup_scene_multi(n, NativeBuilder(), parts)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    sc = get_scene(n)
    pt = PathTracer(1920, 1080)
    view = os.environ.get("VIEW", "headline")   # headline | interior | atrium
    cam = S.atrium_camera(1920, 1080) if view == "atrium" else (S.Camera(1920, 1080, position=(0.0, 0.0, 0.0)) if view == "interior" else S.Camera(1920, 1080))
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth; pt.DoRaySorting = int(os.environ.get('SORT', '0')); pt.set_max_batch(batch)
    reset = batch == 1                          # one frame at a time: Reset -> Compute (-> flush); batches: consecutive samples of one accumulation
    for _ in range(max(3, batch)):
        if reset: pt.ResetAccumulation()
        pt.Compute()
    pt.synchronize(); pt.reset_stats()
    t0 = time.time()
    for _ in range(frames):
        if reset: pt.ResetAccumulation()
        pt.Compute()
    pt.synchronize()
    dt = (time.time() - t0) / frames
    st = pt.stats()
    print("ms/frame", dt * 1e3, "Mray/s", st["rays_traced"] / frames / dt / 1e6, st["alive_counts"][:depth + 1], "flagged/frame", st["wide_flagged_rays"] / frames)
