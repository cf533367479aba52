"""Golden-vector cases shared by the generator (make_golden.py), the CPU tests (oracle vs fixtures) and the GPU tests
(HIP path vs fixtures).  The reference ships no golden vectors, so these are minted from the oracle —
they pin the oracle against regressions and give the GPU tests a checker that needs no oracle run."""
from idkengine_amd import scenes as S

CASES = {
    # name: (scene factory(builder), camera factory(w,h), w, h, settings overrides)
    "cornell_mixed_d5": (lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 64, 64, dict(RayDepth=5)),
    "cornell_inst_tlas_aov_d4": (lambda b: S.cornell_scene(b, "mixed", True), S.cornell_camera, 48, 48, dict(RayDepth=4, UseTlas=1, OutputAOVs=1, SamplesPerPixel=2)),
    "presplit_sort_d4": (lambda b: S.presplit_scene(b), S.presplit_camera, 96, 54, dict(RayDepth=4, DoRaySorting=1)),
    "soup20k_d2": (lambda b: S.soup_scene(20000, b, seed=11), lambda w, h: S.Camera(w, h), 160, 90, dict(RayDepth=2)),
    "cornell_debugcost": (lambda b: S.cornell_scene(b), S.cornell_camera, 64, 64, dict(DoDebugBVHTraversal=1, RayDepth=1)),
    "cornell_lens_norr_d3": (lambda b: S.cornell_scene(b), S.cornell_camera, 64, 64, dict(RayDepth=3, FocalLength=3.0, LenseRadius=0.05, DoRussianRoulette=0)),
}

BVH_CASES = {
    "cornell": lambda b: S.cornell_scene(b),
    "cornell_instanced": lambda b: S.cornell_scene(b, "mixed", True),
    "soup1000": lambda b: S.soup_scene(1000, b, seed=3),
    "soup60000": lambda b: S.soup_scene(60000, b, seed=4),
    "soup1000_refit": lambda b: S.soup_scene(1000, b, seed=3, refittable=True),
    "presplit": lambda b: S.presplit_scene(b),
}


def apply_settings(settings, overrides):
    """settings: gputypes.Settings; Gpu.* fields are addressed by bare name like the PathTracer properties."""
    for k, v in overrides.items():
        if hasattr(settings.Gpu, k):
            setattr(settings.Gpu, k, v)
        else:
            setattr(settings, k, v)
    return settings
