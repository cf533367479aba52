"""Golden-vector cases shared by the generator (make_golden.py), the CPU tests (oracle vs fixtures) and the GPU tests
(HIP path vs fixtures).  The reference ships no golden vectors, so these are minted from the oracle —
they pin the oracle against regressions and give the GPU tests a checker that needs no oracle run."""
import os
from idkengine_amd import scenes as S

_MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


def lucy_scene(b, **kw):
    """Lucy (8 954 triangles; reference: Resource/Models/LucyCompressed), diffuse stone, roughness 0.55 as in its glTF."""
    return S.mesh_scene(os.path.join(_MODELS, "lucy.npz"), b, **kw)


def helmet_scene(b, **kw):
    """DamagedHelmet (15 452 triangles, shared vertices; reference: Resource/Models/HelmetCompressed), metal with factor-only material."""
    return S.mesh_scene(os.path.join(_MODELS, "helmet.npz"), b, material=S.make_material((0.75, 0.7, 0.6, 1.0), metallic=0.7, roughness=0.35, emissive=(0.02, 0.02, 0.03)), **kw)


def lucy_camera(w, h):
    return S.Camera(w, h, position=(0.2, -0.9, 9.5), fovy_deg=40.0)


def helmet_camera(w, h):
    return S.Camera(w, h, position=(0.6, 0.3, 3.0), view_dir=(-0.2, -0.1, -1.0), fovy_deg=45.0)

CASES = {
    # name: (scene factory(builder), camera factory(w,h), w, h, settings overrides)
    "cornell_mixed_d5": (lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 64, 64, dict(RayDepth=5)),
    "cornell_inst_tlas_aov_d4": (lambda b: S.cornell_scene(b, "mixed", True), S.cornell_camera, 48, 48, dict(RayDepth=4, UseTlas=1, OutputAOVs=1, SamplesPerPixel=2)),
    "presplit_sort_d4": (lambda b: S.presplit_scene(b), S.presplit_camera, 96, 54, dict(RayDepth=4, DoRaySorting=1)),
    "soup20k_d2": (lambda b: S.soup_scene(20000, b, seed=11), lambda w, h: S.Camera(w, h), 160, 90, dict(RayDepth=2)),
    "cornell_debugcost": (lambda b: S.cornell_scene(b), S.cornell_camera, 64, 64, dict(DoDebugBVHTraversal=1, RayDepth=1)),
    "cornell_lens_norr_d3": (lambda b: S.cornell_scene(b), S.cornell_camera, 64, 64, dict(RayDepth=3, FocalLength=3.0, LenseRadius=0.05, DoRussianRoulette=0)),
    # real geometry from the reference's own assets (tests/golden/make_models.py)
    "lucy_d5": (lucy_scene, lucy_camera, 96, 128, dict(RayDepth=5)),
    "helmet_sort_d4": (helmet_scene, helmet_camera, 112, 96, dict(RayDepth=4, DoRaySorting=1)),
}

BVH_CASES = {
    "cornell": lambda b: S.cornell_scene(b),
    "cornell_instanced": lambda b: S.cornell_scene(b, "mixed", True),
    "soup1000": lambda b: S.soup_scene(1000, b, seed=3),
    "soup60000": lambda b: S.soup_scene(60000, b, seed=4),
    "soup1000_refit": lambda b: S.soup_scene(1000, b, seed=3, refittable=True),
    "presplit": lambda b: S.presplit_scene(b),
    "lucy": lucy_scene,
    "helmet": helmet_scene,
    "helmet_refit": lambda b: helmet_scene(b, refittable=True),
}


def apply_settings(settings, overrides):
    """settings: gputypes.Settings; Gpu.* fields are addressed by bare name like the PathTracer properties."""
    for k, v in overrides.items():
        if hasattr(settings.Gpu, k):
            setattr(settings.Gpu, k, v)
        else:
            setattr(settings, k, v)
    return settings


def sah_cost(sc, traversal_cost=1.0, triangle_cost=1.1):
    """Global SAH cost of the first BLAS (BLAS.ComputeGlobalSAH, Bvh/BLAS.cs: sum over nodes of HalfArea * cost / root HalfArea), in float64."""
    import numpy as np
    d = sc.blas_descs[0]
    n = sc.blas_nodes[d["NodeOffset"]: d["NodeOffset"] + d["NodeCount"]]
    ext = (n["Max"].astype(np.float64) - n["Min"].astype(np.float64))
    half = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
    cost = np.where(n["TriCount"] > 0, n["TriCount"] * triangle_cost, traversal_cost)
    used = np.ones(len(n), bool); used[0] = False
    return float((half[used] * cost[used]).sum() / half[1])
