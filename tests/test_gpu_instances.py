"""Instance list / TLAS paths of the persistent traversal kernel and the device TLAS build (BVHIntersect.glsl:205-287, Bvh/TLAS.cs)."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402,F401
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal, _queries  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def test_multi_instance_fast_path_equals_generic(native_builder, monkeypatch):
    """Several BLAS instances without a TLAS (the reference's default mode) run on the persistent traversal kernel with the
    per-lane instance loop; it must agree bit-for-bit with the general kernel, with and without sample batching."""
    sc = S.cornell_scene(native_builder, "mixed", True); cam = S.cornell_camera(200, 120)
    a = gpu_render(sc, cam, 200, 120, RayDepth=5, DoRaySorting=1, SamplesPerPixel=2)
    from idkengine_amd.pathtracer import PathTracer
    c = PathTracer(200, 120); c.UploadScene(sc); c.SetCamera(cam); c.RayDepth = 5; c.DoRaySorting = 1; c.SamplesPerPixel = 2
    c.set_max_batch(4); c.enable_counters(True); c.Compute(); c.flush()
    monkeypatch.setenv("IDKPT_FORCE_GENERIC", "1")
    b = gpu_render(sc, cam, 200, 120, RayDepth=5, DoRaySorting=1, SamplesPerPixel=2)
    assert (bits(a.Result) == bits(b.Result)).all() and a.rays().tobytes() == b.rays().tobytes()
    assert (bits(c.Result) == bits(b.Result)).all()
    for k in ("node_pair_visits", "triangle_tests", "rays_traced"):
        assert a.stats()[k] == b.stats()[k] == c.stats()[k], k
    a.Dispose(); b.Dispose(); c.Dispose()


@pytest.mark.parametrize("use_tlas", [0, 1])
def test_many_instances_fast_path(native_builder, oracle_mod, monkeypatch, use_tlas):
    """12 rotated BLAS instances (deep PLOC TLAS when use_tlas=1): persistent kernel (instance loop / in-kernel TLAS walk) vs the
    oracle and vs the general kernel, batched, with exact visit counters."""
    sc = S.soup_scene_multi(6000, native_builder, parts=12, seed=5); w, h = 160, 96; cam = S.Camera(w, h)
    ov = dict(RayDepth=4, UseTlas=use_tlas, SamplesPerPixel=3, DoRaySorting=1)
    o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    a = gpu_render(sc, cam, w, h, **ov)
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    c = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); c.UploadScene(sc); c.SetCamera(cam)
    c.set_max_batch(3); c.enable_counters(True); c.Compute(); c.flush()
    monkeypatch.setenv("IDKPT_FORCE_GENERIC", "1")
    b = gpu_render(sc, cam, w, h, **ov)
    assert (bits(a.Result) == bits(o.image(0))).all()
    assert (bits(a.Result) == bits(b.Result)).all() and (bits(c.Result) == bits(b.Result)).all()
    os_ = o.stats()
    for k in ("node_pair_visits", "triangle_tests", "rays_traced"):
        assert a.stats()[k] == b.stats()[k] == c.stats()[k] == os_[k], k
    monkeypatch.delenv("IDKPT_FORCE_GENERIC")
    d = gpu_render(sc, cam, w, h, counters=False, **ov)      # counters off: the plain (non-counting) build of the traversal kernel
    assert_equal(d, o, counters=False)
    a.Dispose(); b.Dispose(); c.Dispose(); d.Dispose(); o.close()


@pytest.mark.parametrize("use_tlas", [1, 0])
def test_many_small_instances_deep_tlas(native_builder, oracle_mod, use_tlas):
    """600 BLAS instances of a few triangles each: a deep PLOC TLAS (per-lane TLAS stack in LDS) or a long instance list, with the
    ray-query entry point on top — frame, counters and 20 000 closest/any-hit queries equal the oracle."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene_multi(3600, native_builder, parts=600, seed=13, extent=4.0, edge=0.4); w, h = 160, 96; cam = S.Camera(w, h, position=(0.0, 0.0, 11.0), fovy_deg=60.0)
    ov = dict(RayDepth=3, UseTlas=use_tlas)
    pt = gpu_render(sc, cam, w, h, **ov); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    assert_equal(pt, o)
    p3 = gpu_render(sc, cam, w, h, counters=False, **ov)     # plain (non-counting) build
    assert_equal(p3, o, counters=False); p3.Dispose()
    rays = _queries(20000, 17, 6.0)
    for any_hit in (False, True):
        assert pt.TraceRays(rays, any_hit=any_hit).tobytes() == oracle_mod.trace_rays(sc, rays, any_hit=any_hit, use_tlas=bool(use_tlas)).tobytes()
    pt.Dispose(); o.close()


@pytest.mark.parametrize("parts,tris", [(1, 500), (2, 600), (3, 900), (12, 6000), (200, 4000), (1500, 6000)])
def test_device_tlas_build_matches_host_build(native_builder, oracle_builder, parts, tris):
    """TLAS rebuild on the device (idkptBuildTlasOnDevice: instance world bounds + Morton order + PLOC) must give the node array
    of the serial host build (TLAS.Build, Bvh/TLAS.cs:28-141) bit for bit, also after the transforms moved."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc = S.soup_scene_multi(tris, native_builder, parts=parts, seed=9) if parts > 1 else S.soup_scene(tris, native_builder, seed=9)
    pt = PathTracer(64, 64); pt.UploadScene(sc)
    pt.BuildTlasOnDevice()
    got = pt.DownloadBuffer(T.IDKPT_BUF_TLAS_NODES, T.GpuTlasNode, 2 * parts - 1)
    assert got.tobytes() == sc.tlas_nodes.tobytes()
    # move every instance (animated frame), rebuild on both sides
    rng = np.random.default_rng(parts)
    xf = sc.mesh_transforms.copy()
    for i in range(len(xf)):
        m = S.rotation_y(float(rng.uniform(0, 360))) @ S.translation(tuple(rng.uniform(-6, 6, 3)))
        xf[i] = S.transform_from_matrix(m)[0]
    sc.mesh_transforms = xf
    pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf)
    pt.BuildTlasOnDevice()
    S.rebuild_tlas(sc, oracle_builder)
    got = pt.DownloadBuffer(T.IDKPT_BUF_TLAS_NODES, T.GpuTlasNode, 2 * parts - 1)
    assert got.tobytes() == sc.tlas_nodes.tobytes()
    if parts == 12:   # and the frame traced through the device-built TLAS equals the frame through the uploaded one
        cam = S.Camera(96, 64)
        a = PathTracer(96, 64); a.UploadScene(sc); a.SetCamera(cam); a.UseTlas = 1; a.RayDepth = 3; a.Compute()
        b = PathTracer(96, 64); b.UploadScene(sc); b.SetCamera(cam); b.BuildTlasOnDevice(); b.UseTlas = 1; b.RayDepth = 3; b.Compute()
        assert (bits(a.Result) == bits(b.Result)).all()
        a.Dispose(); b.Dispose()
    pt.Dispose()


@pytest.mark.parametrize("parts,use_tlas", [(2, 0), (3, 1), (8, 0), (8, 1), (9, 1)])
def test_instance_entries_computed_in_the_kernel(native_builder, oracle_mod, parts, use_tlas):
    """The instance loop / the TLAS walk of k_trace2 (MODE 1 / 2: RayTransform, 1/dir and the root-box test per (ray, instance) inside the persistent kernel): the oracle's
    frame bit for bit (image, ray records, alive queue, primary hits, visit counters), with lights in front of the geometry, batched, sorted."""
    sc = S.soup_scene_multi(5000, native_builder, parts=parts, seed=11 + parts); w, h = 150, 90
    sc.lights = S.make_lights([((0.0, 3.0, 14.0), 0.8, (9.0, 8.0, 7.0))])
    cam = S.Camera(w, h, position=(1.0, 0.5, 24.0))
    ov = dict(RayDepth=4, UseTlas=use_tlas, DoTraceLights=1, DoRaySorting=1)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=3, **ov)
    a = gpu_render(sc, cam, w, h, frames=3, **ov)
    assert_equal(a, o)
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    b = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); b.UploadScene(sc); b.SetCamera(cam); b.set_max_batch(3)
    for _ in range(3):
        b.Compute()
    assert (bits(b.Result) == bits(o.image(0))).all() and b.rays().tobytes() == o.rays().tobytes() and (b.alive_queue() == o.alive_queue()).all()
    a.Dispose(); b.Dispose(); o.close()
