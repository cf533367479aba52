"""Secondary benchmark (SURVEY.md §8d config 5 stand-in): an animated frame entirely on the device.

Per frame: upload 2 joint matrices (96 B) -> skin all vertices (Skinning/compute.glsl) -> BLAS refit (BLASRefit/compute.glsl)
-> TLAS rebuild (BVH.TlasBuild) -> one 1-spp path-traced frame (RayDepth 2).  The reference does the refit on the GPU too, but reads
the skinned positions back and rebuilds the TLAS on the CPU every frame (ModelManager.cs:263-361, Bvh/BVH.cs:278-298, 472-489).
With F frames in flight (idkptSetSceneVersions(2 F) + idkptSetFrameRing(F) + idkptSetMaxBatch(F)): the updates run at once, into scene states no queued
frame reads, and F frames with F different geometries are traced by one set of launches (DESIGN.md 4 "Scene versions").
Usage: python tools/bench_animated.py [n_tris=1000000] [frames=64] [in-flight list, default 1,2,4,8,16,32]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from idkengine_amd import scenes as S, gputypes as T
from idkengine_amd.bvh import NativeBuilder
from idkengine_amd.pathtracer import PathTracer

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    flights = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32]
    W, H = 1920, 1080
    sc = S.soup_scene(n, NativeBuilder(), seed=1, refittable=True)          # refittable => no PreSplit (Bvh/BVH.cs:325)
    nv = len(sc.vertex_positions)
    un = np.zeros(nv, T.GpuUnskinnedVertex)
    un["Position"] = sc.vertex_positions; un["Normal"] = sc.vertices["Normal"]; un["Tangent"] = sc.vertices["Tangent"]
    un["JointIndices"][:, 1] = 1
    wgt = (0.5 + 0.5 * np.sin(sc.vertex_positions[:, 0] * 0.7)).astype(np.float32)                # blend of two joints along x
    un["JointWeights"][:, 0] = wgt; un["JointWeights"][:, 1] = 1.0 - wgt
    pt = PathTracer(W, H); pt.UploadScene(sc); pt.SetCamera(S.Camera(W, H)); pt.RayDepth = 2
    pt.UploadUnskinnedVertices(un)

    def joints(t):
        j = np.zeros((2, 3, 4), np.float32); j[0, :, :3] = np.eye(3); j[1, :, :3] = np.eye(3)
        c, s_ = np.cos(0.05 * np.sin(t)), np.sin(0.05 * np.sin(t))
        j[1, :, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]; j[0, :, 3] = (0.0, 0.05 * np.sin(1.3 * t), 0.0)
        return j

    def frame(t, render=True, ring=False):
        pt.UpdateBuffer(T.IDKPT_BUF_JOINT_MATRICES, joints(t))
        pt.Skin(0, 0, 0, nv)
        pt.RefitBlas(0)
        pt.BuildTlasOnDevice()
        if render:
            if ring:
                pt.BeginFrame()
            else:
                pt.ResetAccumulation()
            pt.Compute()

    for k in range(5):
        frame(0.1 * k)
    pt.synchronize()
    t0 = time.perf_counter()
    for k in range(frames):
        frame(0.5 + 0.1 * k, render=False)
    pt.synchronize(); t_anim = (time.perf_counter() - t0) / frames
    pt.reset_stats()
    t0 = time.perf_counter()
    for k in range(frames):
        frame(0.5 + 0.1 * k)
    pt.synchronize(); t_all = (time.perf_counter() - t0) / frames
    st = pt.stats()
    print({"tris": n, "vertices": nv, "ms_skin_refit_tlas": round(t_anim * 1e3, 3), "ms_animated_frame_incl_1spp_depth2": round(t_all * 1e3, 3),
           "Mray_per_s_incl_animation": round(st["rays_traced"] / frames / t_all / 1e6, 1)}, flush=True)
    import hashlib
    ref_hash = None
    for F in flights:
        pt.SetSceneVersions(max(1, 2 * F if F > 1 else 1)); pt.SetFrameRing(F); pt.set_max_batch(F)
        for rep in range(2):                       # first repetition: warm-up (arena growth, grids)
            pt.synchronize(); pt.reset_stats()
            t0 = time.perf_counter()
            for k in range(frames):
                frame(0.5 + 0.1 * k, ring=True)
            pt.flush(); pt.synchronize(); t_all = (time.perf_counter() - t0) / frames
        st = pt.stats()
        last = hashlib.sha256(pt.FrameResult((frames - 1) % F).tobytes()).hexdigest()[:16]     # the last frame's image: the same bits for every F
        ref_hash = ref_hash or last
        print({"frames_in_flight": F, "scene_versions": max(1, 2 * F if F > 1 else 1), "ms_per_animated_frame": round(t_all * 1e3, 4), "Mray_per_s_incl_animation": round(st["rays_traced"] / frames / t_all / 1e6, 1),
               "last_frame_sha": last, "same_bits_as_one_at_a_time": last == ref_hash}, flush=True)
