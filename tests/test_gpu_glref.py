"""HIP path (through the C-ABI) vs the REFERENCE's own shader outputs (tests/golden/glref/, minted by oracle/glref/make_vectors.py
from /root/reference's GLSL on Mesa llvmpipe).  No oracle involved: every bounce is compared with what the reference's NHit dispatch
produced from the same input state, and the whole frame with the reference's free-running frame."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
import glref_cases  # noqa: E402
import glref_check  # noqa: E402
from gpu_helpers import gpu_render  # noqa: E402

pytestmark = pytest.mark.gpu
FIXTURES = os.path.join(HERE, "golden", "glref")


@pytest.mark.parametrize("name", list(glref_cases.GLREF_CASES))
def test_hip_path_matches_reference_shaders(name, native_builder):
    from idkengine_amd import gputypes as T
    fac, camf, w, h, ov = glref_cases.GLREF_CASES[name]
    sc = fac(native_builder); cam = camf(w, h)
    fx = np.load(os.path.join(FIXTURES, name + ".npz"))

    def state_at(d):
        pt = gpu_render(sc, cam, w, h, **dict(ov, RayDepth=d, SamplesPerPixel=1))
        r, q = pt.rays().copy(), pt.alive_queue().copy(); pt.Dispose()
        return r, q
    pt = gpu_render(sc, cam, w, h, **ov)
    aov = bool(configs.apply_settings(T.Settings.default(), ov).OutputAOVs)
    final = dict(image=pt.Result, counts=pt.stats()["alive_counts"], albedo=pt.AlbedoTexture if aov else None, normal=pt.NormalTexture if aov else None)
    rep = glref_check.check_case(fx, state_at, final, strict=True, name=name)
    pt.Dispose()
    assert all(s["flips"] == 0 and s["beyond_tol"] == 0 and s["queue_identical"] for s in rep["stages"]), rep


def _mv():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "glref"))
    import make_vectors
    return make_vectors


def test_hip_ray_queries_match_reference_functions(native_builder):
    """idkptTraceRays vs the reference's own TraceRay / TraceRayAny (BVHIntersect.glsl:183-411) run on llvmpipe."""
    from idkengine_amd.pathtracer import PathTracer
    fx = np.load(os.path.join(FIXTURES, "queries.npz"))
    sc = _mv().query_scene(native_builder)
    pt = PathTracer(8, 8); pt.UploadScene(sc)
    for tlas in (0, 1):
        pt.UseTlas = tlas
        for any_hit in (0, 1):
            for lights in (0, 1):
                got = pt.TraceRays(fx["rays"], any_hit=bool(any_hit), trace_lights=bool(lights))
                glref_check.check_query_hits(got, fx[f"hits_tlas{tlas}_any{any_hit}_lights{lights}"])
    pt.Dispose()


def test_hip_rt_shadows_match_reference_shader(native_builder):
    """idkptTraceShadows vs Shaders/ShadowsRayTraced/compute.glsl run on llvmpipe."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import scenes as S, gputypes as T
    mv = _mv()
    fx = np.load(os.path.join(FIXTURES, "shadows.npz"))
    w, h = mv.SHADOW_SIZE
    cam = S.cornell_camera(w, h)
    for variant, tlas in mv.SHADOW_CONFIGS:
        sc = mv.shadow_scene(native_builder, variant)
        pt = PathTracer(8, 8); pt.UploadScene(sc); pt.UseTlas = tlas
        depth, normal = fx[f"depth_{variant}_{tlas}"], fx[f"normal_{variant}_{tlas}"]
        for light, samples, noise in mv.SHADOW_PARAMS:
            p = T.ShadowParams.make(cam.inv_proj_view, w, h, light_index=light, samples=samples, noise_index=noise, jitter=(0.0005, -0.0003))
            got = pt.TraceShadows(p, depth, normal, visibility=np.full((h, w), np.float32(-3.0)))
            glref_check.check_shadow_image(got, fx[f"vis_{variant}_{tlas}_{light}_{samples}_{noise}"])
        pt.Dispose()


def test_hip_refit_and_skinning_match_reference_shaders(native_builder):
    """idkptRefitBlas vs Shaders/BLASRefit/compute.glsl (every node, bit for bit); idkptSkin vs Shaders/Skinning/compute.glsl: positions and the
    previous-position copy bit for bit; the re-compressed 11/11/10-bit normals and tangents identical up to one quantisation step where llvmpipe's
    inversesqrt rounds the other way."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    mv = _mv()
    fx = np.load(os.path.join(FIXTURES, "updates.npz"))
    sc, moved, un, joints, sk = mv.update_inputs(native_builder)
    pt = PathTracer(8, 8); pt.UploadScene(sc)
    pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, moved)
    pt.RefitBlas(0)
    assert pt.DownloadBuffer(T.IDKPT_BUF_BLAS_NODES, T.GpuBlasNode, len(sc.blas_nodes)).tobytes() == fx["refit_nodes"].tobytes()
    pt.UploadUnskinnedVertices(un); pt.UpdateBuffer(T.IDKPT_BUF_JOINT_MATRICES, joints)
    pt.Skin(sk["input_offset"], sk["output_offset"], sk["joint_offset"], sk["count"]); pt.synchronize()
    lo, hi = sk["output_offset"], sk["output_offset"] + sk["count"]
    pos = pt.DownloadBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, np.float32, 3 * len(moved)).reshape(-1, 3)
    assert pos[lo:hi].tobytes() == fx["skin_positions"].tobytes()
    assert pos[:lo].tobytes() == moved[:lo].tobytes() and pos[hi:].tobytes() == moved[hi:].tobytes()
    verts = pt.DownloadBuffer(T.IDKPT_BUF_VERTICES, T.GpuVertex, len(sc.vertices))
    for field, key in (("Normal", "skin_normals"), ("Tangent", "skin_tangents")):
        got, ref = verts[field][lo:hi].astype(np.int64), fx[key].astype(np.int64)
        same = got == ref
        assert same.mean() >= 0.98, (field, same.mean())
        for shift, mask in ((0, 2047), (11, 2047), (22, 1023)):
            assert np.abs(((got >> shift) & mask) - ((ref >> shift) & mask)).max() <= 1, field
    pt.Dispose()
