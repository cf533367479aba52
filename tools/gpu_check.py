"""Developer tool (runs on the GPU box via gpurun): GPU-vs-oracle parity summary on several scenes + a quick timing.
Not a test and not the bench; prints a compact report and writes gpurun_out/gpu_check.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402
from oracle import oracle as O  # noqa: E402

report = {}


def compare(name, sc, cam, w, h, depth, spp=1, tlas=False, lights=False, sort=False, aov=False, debug=False, focal=None):
    pt = PathTracer(w, h)
    pt.UploadScene(sc); pt.SetCamera(cam)
    pt.RayDepth = depth; pt.SamplesPerPixel = spp; pt.UseTlas = 1 if tlas else 0; pt.DoTraceLights = 1 if lights else 0
    pt.DoRaySorting = 1 if sort else 0; pt.OutputAOVs = 1 if aov else 0
    if focal:
        pt.FocalLength, pt.LenseRadius = focal
    if debug:
        pt.DoDebugBVHTraversal = True
    pt.enable_primary_hit_capture(True); pt.enable_counters(True)
    pt.ResetAccumulation(); pt.reset_stats()
    pt.Compute()
    g_img = pt.Result; g_rays = pt.rays(); g_t, g_tri, g_bary = pt.primary_hits(); g_stats = pt.stats(); g_alive = pt.alive_queue()
    o = O.OraclePathTracer(sc, w, h)
    o.set_camera(cam)
    s = o.settings
    s.RayDepth = 1 if debug else depth; s.SamplesPerPixel = spp; s.UseTlas = 1 if tlas else 0; s.Gpu.DoTraceLights = 1 if lights else 0
    s.DoRaySorting = 1 if sort else 0; s.OutputAOVs = 1 if aov else 0; s.Gpu.DoDebugBVHTraversal = 1 if debug else 0
    if focal:
        s.Gpu.FocalLength, s.Gpu.LenseRadius = focal
    o.enable_counters(True)
    o.render()
    o_img = o.image(); o_rays = o.rays(); o_t, o_tri, o_bary = o.primary_hits(); o_stats = o.stats(); o_alive = o.alive_queue()
    res = {
        "tri_mismatch": int((g_tri != o_tri).sum()), "t_mismatch": int((g_t.view(np.uint32) != o_t.view(np.uint32)).sum()),
        "bary_mismatch": int((g_bary.view(np.uint32) != o_bary.view(np.uint32)).any(axis=1).sum()),
        "img_bits_mismatch_px": int((g_img.view(np.uint32) != o_img.view(np.uint32)).any(axis=2).sum()),
        "img_max_abs": float(np.nanmax(np.abs(g_img - o_img))), "img_nan": int(np.isnan(g_img).sum()),
        "rays_bits_mismatch": int((g_rays.view(np.uint32).reshape(-1, 12) != o_rays.view(np.uint32).reshape(-1, 12)).any(axis=1).sum()),
        "alive_equal": bool(len(g_alive) == len(o_alive) and (g_alive == o_alive).all()),
        "gpu_alive": g_stats["alive_counts"][:depth + 1], "ref_alive": o_stats["alive_counts"][:depth + 1],
        "gpu_rays": g_stats["rays_traced"], "ref_rays": o_stats["rays_traced"],
        "gpu_pairs": g_stats["node_pair_visits"], "ref_pairs": o_stats["node_pair_visits"], "gpu_tris": g_stats["triangle_tests"], "ref_tris": o_stats["triangle_tests"],
    }
    if aov:
        for k, wh in (("albedo", 1), ("normal", 2)):
            res[k + "_bits_mismatch_px"] = int((pt.download(wh).view(np.uint32) != o.image(wh).view(np.uint32)).any(axis=2).sum())
    report[name] = res
    print(name, json.dumps(res), flush=True)
    pt.Dispose(); o.close()


def timing(name, sc, cam, w, h, depth, frames=20, sort=False):
    pt = PathTracer(w, h)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth; pt.DoRaySorting = 1 if sort else 0
    for _ in range(3):
        pt.ResetAccumulation(); pt.Compute()
    pt.synchronize(); pt.reset_stats()
    t0 = time.time()
    for _ in range(frames):
        pt.ResetAccumulation(); pt.Compute()
    pt.synchronize()
    dt = (time.time() - t0) / frames
    st = pt.stats()
    rays = st["rays_traced"] / frames
    res = {"ms_per_frame": dt * 1e3, "rays_per_frame": rays, "mray_s": rays / dt / 1e6, "alive": st["alive_counts"][:depth + 1]}
    report[name] = res
    print(name, json.dumps(res), flush=True)
    pt.Dispose()


if __name__ == "__main__":
    b = O.OracleBuilder()
    quick = "--quick" in sys.argv
    cam = S.cornell_camera(256, 256)
    compare("cornell_diffuse_d2", S.cornell_scene(b), cam, 256, 256, 2)
    compare("cornell_diffuse_d7_spp3_aov", S.cornell_scene(b), cam, 256, 256, 7, spp=3, aov=True)
    compare("cornell_mixed_d7", S.cornell_scene(b, variant="mixed"), cam, 256, 256, 7)
    compare("cornell_mixed_inst_tlas_d5", S.cornell_scene(b, variant="mixed", instanced=True), cam, 256, 256, 5, tlas=True)
    compare("cornell_mixed_inst_notlas_d5_sort", S.cornell_scene(b, variant="mixed", instanced=True), cam, 256, 256, 5, sort=True)
    compare("cornell_debug", S.cornell_scene(b), cam, 256, 256, 1, debug=True)
    compare("cornell_lens", S.cornell_scene(b), cam, 256, 256, 3, focal=(3.0, 0.05))
    soup = S.soup_scene(100000, b)
    camS = S.Camera(640, 360)
    compare("soup100k_d2", soup, camS, 640, 360, 2)
    compare("soup100k_d5_sort", soup, camS, 640, 360, 5, sort=True)
    if not quick:
        t0 = time.time(); soup1m = S.soup_scene(1000000, b); print("built 1M soup in", time.time() - t0, flush=True)
        cam1 = S.Camera(1920, 1080)
        compare("soup1m_1080p_d2", soup1m, cam1, 1920, 1080, 2)
        timing("time_soup1m_1080p_d2", soup1m, cam1, 1920, 1080, 2)
        timing("time_soup1m_1080p_d5", soup1m, cam1, 1920, 1080, 5)
        timing("time_soup1m_1080p_d5_sort", soup1m, cam1, 1920, 1080, 5, sort=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open("gpurun_out/gpu_check.json", "w"), indent=1)
