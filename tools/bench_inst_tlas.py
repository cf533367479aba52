"""Round 5: the instance loop through the library's own TLAS (csrc/kernels_trace_inst.hpp) against the exact loop and against the TLAS mode, per scene shape:
the atrium as 87 BLASes (bench.py's multi_blas.atrium_per_mesh block), the soup in 3 / 12 / 60 rotated parts.  One JSON document on stdout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402


def modes(pt, B):
    out = {}
    for name, tlas, own, ovl, sieve, sovl in (("own_tlas", 0, 2, 100, 0, 100), ("exact_loop", 0, 0, 100, 0, 100), ("sieved_loop", 0, 0, 100, 2, 100), ("default", 0, 8, 10, 8, 50), ("tlas_mode", 1, 2, 100, 0, 100)):
        if tlas:
            pt.BuildTlasOnDevice()
        pt.UseTlas = tlas; pt.set_option("inst_tlas", own); pt.set_option("inst_tlas_overlap", ovl); pt.set_option("inst_sieve", sieve); pt.set_option("inst_sieve_overlap", sovl)
        rays, dt = bench.timed_batch(pt, B, B, reps=3)
        st = pt.stats()
        out[name] = {"mray_s": round(rays / dt / 1e6, 1), "single_frame_mray_s": bench.single_frame(pt, bench.RAY_DEPTH, frames=8)["mray_s"], "flagged_share": round(st["inst_tlas_flagged_rays"] / max(st["rays_traced"], 1), 5)}
    pt.UseTlas = 0; pt.set_option("inst_sieve", 8); pt.set_option("inst_sieve_overlap", 50); pt.set_option("inst_tlas", 8); pt.set_option("inst_tlas_overlap", 10)
    out["own_over_exact"] = round(out["own_tlas"]["mray_s"] / out["exact_loop"]["mray_s"], 3)
    return out


def main():
    W, H, B = bench.W, bench.H, 32
    res = {}
    pt = PathTracer(W, H)
    sc = S.atrium_scene(bench.N_TRIS, NativeBuilder(), per_mesh_blas=True)
    pt.UploadScene(sc); pt.SetCamera(S.atrium_camera(W, H)); pt.RayDepth = bench.RAY_DEPTH
    if "--profile-atrium" in sys.argv:    # for rocprofv3 --kernel-trace: the default mode only, 8 frames one at a time and one batch of 32
        pt.set_option("inst_tlas", 8)
        print(json.dumps({"single_frame": bench.single_frame(pt, bench.RAY_DEPTH, frames=8), "batched_mray_s": round((lambda r: r[0] / r[1] / 1e6)(bench.timed_batch(pt, B, B, reps=1)), 1)}))
        pt.Dispose(); return
    res["atrium_87_blases"] = modes(pt, B)
    # clusters: 64 small soups at random places; (e / 10)^2 of them meet a random line (e = half size of a cluster, 10 = half size of the scene)
    import numpy as np
    for e in (2.24, 3.16, 4.47, 5.48):
        rng = np.random.default_rng(7); blases = []
        for k in range(64):
            p, i, nrm, tan = S.flat_shaded(S.soup_triangles(bench.N_TRIS // 64, 100 + k, e, 0.15))
            blases.append({"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.8, 0.8, 1.0)), nrm, tan)], "transform": S.rotation_y(float(rng.uniform(0, 360))) @ S.translation(tuple(rng.uniform(-(10 - e), 10 - e, 3)))})
        sc = S.assemble(blases, NativeBuilder(), build_tlas=False)
        for view in ("headline", "interior"):
            pt.UploadScene(sc); pt.SetCamera(bench.view_camera(S, view, W, H)); pt.RayDepth = bench.RAY_DEPTH
            key = f"clusters_64_e{e}_{view}"; res[key] = modes(pt, B)
            print(json.dumps({key: res[key]}), file=sys.stderr, flush=True)
    for parts in (3, 12, 60):
        for view in ("headline", "interior"):
            sc = S.soup_scene_multi(bench.N_TRIS, NativeBuilder(), parts=parts, seed=1)
            pt.UploadScene(sc); pt.SetCamera(bench.view_camera(S, view, W, H)); pt.RayDepth = bench.RAY_DEPTH
            res[f"soup_{parts}_parts_{view}"] = modes(pt, B)
            print(json.dumps({f"soup_{parts}_parts_{view}": res[f"soup_{parts}_parts_{view}"]}), file=sys.stderr, flush=True)
    pt.Dispose()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
