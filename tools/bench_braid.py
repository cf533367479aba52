"""Round 6: partial re-braiding (k_braid, csrc/kernels_scene.hpp) under the library's own TLAS (option inst_braid) and under the unified tree of same-space scenes (option inst_unify,
k_unify_* + k_trace_inst<.., UNI>) on the scenes whose BLASes overlap: the atrium as 87 BLASes
(one per mesh, the reference's shape), the soup in 3 / 12 / 60 interleaved parts under one transform, and in 3 / 12 rotated parts (not one space).  Per scene and view: the exact loop (k_trace2 MODE 1), the own TLAS over
whole instances (inst_braid 0), over braided entries (budgets), and what the library's defaults pick.  Batched (32 samples in flight) | one frame at a time.  One JSON document on stdout.

    python tools/bench_braid.py [--quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

QUICK = "--quick" in sys.argv
BUDGETS = (0, 2048) if QUICK else (0, 256, 2048)                     # entries of the own TLAS (inst_braid): its TLAS phase pays per entry
UNIFY = (2, 1024, 4096) if QUICK else (2, 256, 1024, 4096, 16384)   # subtrees under the unified tree's top (inst_unify; 2 < instances = whole BLASes): same-space scenes only


def measure(pt, B):
    rays, dt = bench.timed_batch(pt, B, B, reps=2 if QUICK else 3)
    st = pt.stats()
    return {"mray_s": round(rays / dt / 1e6, 1), "single_frame_mray_s": bench.single_frame(pt, bench.RAY_DEPTH, frames=6)["mray_s"], "retraced_share": round(st["inst_tlas_flagged_rays"] / max(st["rays_traced"], 1), 5)}


def modes(pt, B):
    out = {}
    pt.UseTlas = 0
    pt.set_option("inst_unify", 0); pt.set_option("inst_braid", 0)
    pt.set_option("inst_tlas", 0); pt.set_option("inst_sieve", 0)
    out["exact_loop"] = measure(pt, B)
    pt.set_option("inst_tlas", 2); pt.set_option("inst_tlas_overlap", 100)
    for b in BUDGETS:
        pt.set_option("inst_braid", b)
        out["own_tlas_whole_instances" if b == 0 else f"own_tlas_braid_{b}"] = measure(pt, B)
    pt.set_option("inst_braid", 0); pt.set_option("inst_general", 0)
    for u in UNIFY:
        pt.set_option("inst_unify", u)
        m = measure(pt, B); st = pt.stats()
        if st["inst_unified_entries"] == 0:
            out["unified"] = "not one space (or a BLAS used twice): the unified tree does not apply"; break
        m.update(entries=st["inst_unified_entries"], top_depth=st["inst_unified_top_depth"]); out[f"unified_{u}"] = m
    if "unified" in out:      # not one space: the general array (k_trace_inst TREE 2: a world-space top whose entries take the ray into their instance's space)
        pt.set_option("inst_general", 2)
        for u in UNIFY:
            pt.set_option("inst_unify", u)
            m = measure(pt, B); st = pt.stats()
            m.update(entries=st["inst_unified_entries"], top_depth=st["inst_unified_top_depth"]); out[f"general_{u}"] = m
    # the library's defaults
    pt.set_option("inst_unify", 4096); pt.set_option("inst_general", 0); pt.set_option("inst_braid", 0); pt.set_option("inst_tlas", 8); pt.set_option("inst_tlas_overlap", 10); pt.set_option("inst_sieve", 8); pt.set_option("inst_sieve_overlap", 50)
    out["default"] = measure(pt, B)
    uni = [k for k in out if k.startswith("unified_")]
    if uni:
        best = max(uni, key=lambda k: out[k]["mray_s"])
        out["best_unified"] = best; out["best_unified_over_whole_instance_tlas"] = round(out[best]["mray_s"] / out["own_tlas_whole_instances"]["mray_s"], 3); out["best_unified_over_exact_loop"] = round(out[best]["mray_s"] / out["exact_loop"]["mray_s"], 3)
    return out


def same_space_soup(parts):
    """bench.py's soup in `parts` BLASes under ONE transform (soup_scene_multi rotates its parts: that scene stays with the loop / the own TLAS)."""
    blases = []
    per = bench.N_TRIS // parts
    for k in range(parts):
        p, i, nrm, tan = S.flat_shaded(S.soup_triangles(per if k < parts - 1 else bench.N_TRIS - per * (parts - 1), 1 + 17 * k, 10.0, 0.15))
        blases.append({"meshes": [S.MeshInput(p, i, S.make_material(base_color=(0.8, 0.8, 0.8, 1.0), metallic=0.0, roughness=1.0), nrm, tan)], "transform": None})
    return S.assemble(blases, NativeBuilder())


def main():
    W, H, B = bench.W, bench.H, 32
    res = {}
    pt = PathTracer(W, H)

    def run(key, sc, cam):
        pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = bench.RAY_DEPTH
        res[key] = modes(pt, B)
        print(json.dumps({key: res[key]}), file=sys.stderr, flush=True)

    # the same triangles in one BLAS, for scale
    one = S.atrium_scene(bench.N_TRIS, NativeBuilder())
    pt.UploadScene(one); pt.SetCamera(S.atrium_camera(W, H)); pt.RayDepth = bench.RAY_DEPTH
    rays, dt = bench.timed_batch(pt, B, B, reps=2)
    res["atrium_one_blas"] = {"mray_s": round(rays / dt / 1e6, 1), "single_frame_mray_s": bench.single_frame(pt, bench.RAY_DEPTH, frames=6)["mray_s"]}
    run("atrium_87_blases", S.atrium_scene(bench.N_TRIS, NativeBuilder(), per_mesh_blas=True), S.atrium_camera(W, H))
    one = S.soup_scene(bench.N_TRIS, NativeBuilder())
    for view in ("headline", "interior"):
        pt.UploadScene(one); pt.SetCamera(bench.view_camera(S, view, W, H)); pt.RayDepth = bench.RAY_DEPTH
        rays, dt = bench.timed_batch(pt, B, B, reps=2)
        res[f"soup_one_blas_{view}"] = {"mray_s": round(rays / dt / 1e6, 1), "single_frame_mray_s": bench.single_frame(pt, bench.RAY_DEPTH, frames=6)["mray_s"]}
    for parts in ((3,) if QUICK else (3, 12, 60)):
        sc = same_space_soup(parts)
        for view in ("headline", "interior"):
            run(f"soup_{parts}_parts_one_space_{view}", sc, bench.view_camera(S, view, W, H))
    for parts in ((3,) if QUICK else (3, 12)):
        sc = S.soup_scene_multi(bench.N_TRIS, NativeBuilder(), parts=parts, seed=1)
        for view in ("headline", "interior"):
            run(f"soup_{parts}_rotated_parts_{view}", sc, bench.view_camera(S, view, W, H))
    pt.Dispose()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
