"""GPU parity tests proper: the HIP path (through the C-ABI, libidkpt.so) against the CPU oracle on identical seeded
inputs, against the committed golden fixtures, and — at BASELINE.json's full size — through size-independent properties.
Bar: bit-exact for everything (ids, T, barycentrics, radiance, ray state, queues, visit counters): both sides execute
the same IEEE-754 binary32 operation sequence (DESIGN.md "Numerics"); the 1e-4 relative tolerance north_star allows is
therefore asserted as exact equality, with the looser bound kept as a named constant for reference."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402

pytestmark = pytest.mark.gpu
NORTH_STAR_REL_TOL = 1e-4   # BASELINE.json; the tests below demand 0


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def gpu_render(sc, cam, w, h, counters=True, capture=True, frames=1, **ov):
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    st = configs.apply_settings(T.Settings.default(), ov)
    pt = PathTracer(w, h, settings=st)
    pt.UploadScene(sc); pt.SetCamera(cam)
    pt.enable_counters(counters); pt.enable_primary_hit_capture(capture)
    for _ in range(frames):
        pt.Compute()
    return pt


def oracle_render(O, sc, cam, w, h, frames=1, **ov):
    o = O.OraclePathTracer(sc, w, h); o.set_camera(cam)
    configs.apply_settings(o.settings, ov)
    o.enable_counters(True)
    for _ in range(frames):
        o.render()
    return o


def assert_equal(pt, o, aov=False):
    assert (bits(pt.Result) == bits(o.image(0))).all()
    gt, gtri, gb = pt.primary_hits(); ot, otri, ob = o.primary_hits()
    assert (gtri == otri).all() and (bits(gt) == bits(ot)).all() and (bits(gb) == bits(ob)).all()
    assert pt.rays().tobytes() == o.rays().tobytes()
    assert (pt.alive_queue() == o.alive_queue()).all()
    gs, os_ = pt.stats(), o.stats()
    assert gs["rays_traced"] == os_["rays_traced"] and gs["node_pair_visits"] == os_["node_pair_visits"] and gs["triangle_tests"] == os_["triangle_tests"]
    if aov:
        assert (bits(pt.AlbedoTexture) == bits(o.image(1))).all() and (bits(pt.NormalTexture) == bits(o.image(2))).all()


MATRIX = [
    ("cornell_d2", lambda b: S.cornell_scene(b), S.cornell_camera, 256, 256, dict(RayDepth=2)),
    ("cornell_d7_spp3_aov", lambda b: S.cornell_scene(b), S.cornell_camera, 128, 128, dict(RayDepth=7, SamplesPerPixel=3, OutputAOVs=1)),
    ("cornell_mixed_d7", lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 192, 192, dict(RayDepth=7)),
    ("cornell_inst_tlas_d5", lambda b: S.cornell_scene(b, "mixed", True), S.cornell_camera, 128, 128, dict(RayDepth=5, UseTlas=1)),
    ("cornell_inst_notlas_sort_d5", lambda b: S.cornell_scene(b, "mixed", True), S.cornell_camera, 128, 128, dict(RayDepth=5, DoRaySorting=1)),
    ("cornell_debugcost", lambda b: S.cornell_scene(b), S.cornell_camera, 128, 128, dict(DoDebugBVHTraversal=1, RayDepth=1)),
    ("cornell_lens_norr", lambda b: S.cornell_scene(b), S.cornell_camera, 128, 128, dict(RayDepth=4, FocalLength=3.0, LenseRadius=0.05, DoRussianRoulette=0)),
    ("presplit_sort_d6", lambda b: S.presplit_scene(b), S.presplit_camera, 320, 180, dict(RayDepth=6, DoRaySorting=1)),
    ("soup100k_d2", lambda b: S.soup_scene(100000, b), lambda w, h: S.Camera(w, h), 640, 360, dict(RayDepth=2)),
    ("soup100k_d5_sort", lambda b: S.soup_scene(100000, b), lambda w, h: S.Camera(w, h), 640, 360, dict(RayDepth=5, DoRaySorting=1)),
    ("ragged_size_77x33", lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 77, 33, dict(RayDepth=4)),   # not a multiple of 8 / 64
    ("tiny_1x1", lambda b: S.cornell_scene(b), S.cornell_camera, 1, 1, dict(RayDepth=3)),
    # edge cases: widest legal image (FirstHit packs x into 12 bits), a frame where every primary ray misses (empty queues through
    # the sort and every bounce), a single-triangle BLAS (root with one duplicated leaf), primary rays only, less than one tile
    ("max_width_4096x3", lambda b: S.soup_scene(3000, b, seed=4), lambda w, h: S.Camera(w, h, fovy_deg=0.04), 4096, 3, dict(RayDepth=3)),
    ("all_rays_miss_d5_sort", lambda b: S.cornell_scene(b, "mixed", True), lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 30.0), view_dir=(0.0, 0.0, 1.0)), 64, 40, dict(RayDepth=5, DoRaySorting=1, SamplesPerPixel=2)),
    ("all_rays_miss_tlas", lambda b: S.cornell_scene(b, "mixed", True), lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 30.0), view_dir=(0.0, 0.0, 1.0)), 64, 40, dict(RayDepth=3, UseTlas=1)),
    ("single_triangle_blas", lambda b: S.assemble([{"meshes": [S.MeshInput(*S.flat_shaded(np.float32([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]]))[:2], S.make_material((0.9, 0.5, 0.2, 1.0)),
                                                                            *S.flat_shaded(np.float32([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]]))[2:])]}], b),
     lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 3.0), fovy_deg=50.0), 48, 48, dict(RayDepth=4)),
    ("primary_only_d1", lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 100, 60, dict(RayDepth=1, SamplesPerPixel=2)),
    ("sub_tile_5x3", lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 5, 3, dict(RayDepth=6, DoRaySorting=1)),
]


@pytest.mark.parametrize("name,mk_scene,mk_cam,w,h,ov", MATRIX, ids=[m[0] for m in MATRIX])
def test_gpu_equals_oracle(name, mk_scene, mk_cam, w, h, ov, oracle_mod, native_builder):
    sc = mk_scene(native_builder); cam = mk_cam(w, h)
    pt = gpu_render(sc, cam, w, h, **ov); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    assert_equal(pt, o, aov=bool(ov.get("OutputAOVs")))
    pt.Dispose(); o.close()


def test_lights_as_surfaces(oracle_mod, native_builder):
    """DoTraceLights: brute-force sphere lights (BVHIntersect.glsl:189-203) + light-as-surface shading (FirstHit:161-168)."""
    from idkengine_amd import gputypes as T
    sc = S.cornell_scene(native_builder, "mixed")
    lights = np.zeros(2, T.GpuLight)
    lights[0]["Position"] = (0.3, 0.2, 0.4); lights[0]["Radius"] = 0.18; lights[0]["Color"] = (6.0, 5.0, 3.0); lights[0]["PointShadowIndex"] = -1
    lights[1]["Position"] = (-0.5, -0.4, 0.1); lights[1]["Radius"] = 0.1; lights[1]["Color"] = (1.0, 2.0, 8.0); lights[1]["PointShadowIndex"] = -1
    sc.lights = lights
    cam = S.cornell_camera(128, 128)
    for extra in (dict(), dict(DoRaySorting=1)):
        ov = dict(RayDepth=5, DoTraceLights=1, **extra)
        pt = gpu_render(sc, cam, 128, 128, **ov); o = oracle_render(oracle_mod, sc, cam, 128, 128, **ov)
        assert_equal(pt, o)
        pt.Dispose(); o.close()


def test_alpha_blend_and_cutoff_materials(oracle_mod, native_builder):
    """Stochastic alpha blending (AlphaCutoff == 2.0 draws an RNG value, FirstHit:141-146) and alpha cutoff pass-through."""
    m = S.cornell_meshes("mixed")
    m["short"].material = S.make_material((0.9, 0.3, 0.3, 0.4), alpha_cutoff=2.0)     # blend, alpha 0.4
    m["tall"].material = S.make_material((0.3, 0.9, 0.3, 0.3), alpha_cutoff=0.5)      # cutoff: always skipped
    sc = S.assemble([{"meshes": m["walls"] + [m["short"], m["tall"]]}], native_builder, sky_color=(0.2, 0.2, 0.2))
    cam = S.cornell_camera(128, 128)
    pt = gpu_render(sc, cam, 128, 128, RayDepth=6); o = oracle_render(oracle_mod, sc, cam, 128, 128, RayDepth=6)
    assert_equal(pt, o)
    pt.Dispose(); o.close()


def test_textures_and_six_face_sky(oracle_mod, native_builder):
    """Texture-table stand-in for bindless samplers (1x1 exact + bilinear) and a 6-face sky with distinct colours."""
    rng = np.random.default_rng(5)
    m = S.cornell_meshes("diffuse")
    uv = rng.uniform(0, 1, (len(m["tall"].positions), 2)).astype(np.float32)
    m["tall"].uvs = uv
    m["tall"].material["BaseColorTexture"] = 1; m["tall"].material["EmissiveTexture"] = 2; m["tall"].material["EmissiveFactor"] = (0.5, 0.5, 0.5)
    sc = S.assemble([{"meshes": m["walls"] + [m["short"], m["tall"]]}], native_builder, sky_color=None)
    sc.textures = [rng.uniform(0.2, 1.0, (8, 8, 4)).astype(np.float32), np.float32([[[0.2, 0.7, 0.1, 1.0]]])]
    sky = np.zeros((6, 2, 2, 4), np.float32); sky[..., :3] = rng.uniform(0, 1, (6, 2, 2, 3)); sc.sky_faces = sky
    cam = S.cornell_camera(128, 128)
    pt = gpu_render(sc, cam, 128, 128, RayDepth=5, OutputAOVs=1); o = oracle_render(oracle_mod, sc, cam, 128, 128, RayDepth=5, OutputAOVs=1)
    assert_equal(pt, o, aov=True)
    pt.Dispose(); o.close()


@pytest.mark.parametrize("name", list(configs.CASES))
def test_gpu_matches_golden_fixture(name, native_builder):
    """No oracle involved: HIP path vs the committed vectors (tests/golden, minted by make_golden.py)."""
    mk_scene, mk_cam, w, h, ov = configs.CASES[name]
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    pt = gpu_render(mk_scene(native_builder), mk_cam(w, h), w, h, **ov)
    assert (bits(pt.Result) == bits(g["result"])).all()
    t, tri, bary = pt.primary_hits()
    assert (tri == g["prim_tri"]).all() and (bits(t) == bits(g["prim_t"])).all() and (bits(bary) == bits(g["prim_bary"])).all()
    assert pt.rays().tobytes() == g["rays"].tobytes() and (pt.alive_queue() == g["alive"]).all()
    st = pt.stats()
    assert st["rays_traced"] == int(g["rays_traced"]) and st["node_pair_visits"] == int(g["pairs"]) and st["triangle_tests"] == int(g["tris"])
    if "albedo" in g:
        assert (bits(pt.AlbedoTexture) == bits(g["albedo"])).all() and (bits(pt.NormalTexture) == bits(g["normal"])).all()
    pt.Dispose()


def test_generic_path_equals_fast_path(native_builder, monkeypatch):
    """The single-BLAS fast path (gen+cull, persistent while-while) and the general path (multi-instance/TLAS capable)
    must agree bit-for-bit, including visit counters."""
    sc = S.soup_scene(50000, native_builder, seed=2); cam = S.Camera(320, 180)
    a = gpu_render(sc, cam, 320, 180, RayDepth=4)
    monkeypatch.setenv("IDKPT_FORCE_GENERIC", "1")
    b = gpu_render(sc, cam, 320, 180, RayDepth=4)
    assert (bits(a.Result) == bits(b.Result)).all() and a.rays().tobytes() == b.rays().tobytes()
    assert a.stats()["node_pair_visits"] == b.stats()["node_pair_visits"] and a.stats()["triangle_tests"] == b.stats()["triangle_tests"]
    a.Dispose(); b.Dispose()


@pytest.mark.parametrize("lens", [0.0, 0.05, 0.4])
def test_tile_preclassification_is_conservative(native_builder, oracle_mod, monkeypatch, lens):
    """k_classify_tiles skips ray generation for tiles whose whole beam provably misses the root box; with it, without it
    (IDKPT_NO_TILE_CULL) and the oracle must agree on every bit (image, ray state incl. the lazily regenerated planes), for a thin
    and a wide lens, a ragged size, interleaved rows and a camera close to the box edge."""
    sc = S.soup_scene(20000, native_builder, seed=8, extent=3.0)
    w, h = 333, 187
    for cam in (S.Camera(w, h, position=(0.0, 0.0, 9.0)), S.Camera(w, h, position=(2.9, 1.0, 3.4), view_dir=(-0.3, -0.1, -1.0), fovy_deg=70.0)):
        ov = dict(RayDepth=2, LenseRadius=lens, FocalLength=6.0, SamplesPerPixel=2)
        a = gpu_render(sc, cam, w, h, **ov)
        o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        assert_equal(a, o)
        monkeypatch.setenv("IDKPT_NO_TILE_CULL", "1")
        b = gpu_render(sc, cam, w, h, **ov)
        monkeypatch.delenv("IDKPT_NO_TILE_CULL")
        assert (bits(a.Result) == bits(b.Result)).all() and a.rays().tobytes() == b.rays().tobytes()
        assert a.stats()["alive_counts"][:2] == b.stats()["alive_counts"][:2]
        # the lazily completed ray state must not depend on what happens to the scene afterwards (the sky decides the miss radiance)
        c = gpu_render(sc, cam, w, h, **ov)
        c.UploadScene(S.soup_scene(500, native_builder, seed=3, sky_color=(0.1, 0.2, 0.3)))
        assert c.rays().tobytes() == b.rays().tobytes()
        a.Dispose(); b.Dispose(); c.Dispose(); o.close()
    # row-sharded context: the tile's rows are every 3rd image row
    from idkengine_amd.pathtracer import PathTracer
    cam = S.Camera(w, h, position=(0.0, 0.0, 9.0))
    full = gpu_render(sc, cam, w, h, RayDepth=2, LenseRadius=lens, FocalLength=6.0)
    for rem in range(3):
        p = PathTracer(w, h, row_modulo=3, row_remainder=rem); p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 2; p.LenseRadius = lens; p.FocalLength = 6.0
        p.Compute()
        assert (bits(p.Result) == bits(full.Result[rem::3])).all()
        p.Dispose()
    full.Dispose()


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
    a.Dispose(); b.Dispose(); c.Dispose(); o.close()


@pytest.mark.parametrize("use_tlas", [1, 0])
def test_many_small_instances_deep_tlas(native_builder, oracle_mod, use_tlas):
    """600 BLAS instances of a few triangles each: a deep PLOC TLAS (per-lane TLAS stack in LDS) or a long instance list, with the
    ray-query entry point on top — frame, counters and 20 000 closest/any-hit queries equal the oracle."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene_multi(3600, native_builder, parts=600, seed=13, extent=4.0, edge=0.4); w, h = 160, 96; cam = S.Camera(w, h, position=(0.0, 0.0, 11.0), fovy_deg=60.0)
    ov = dict(RayDepth=3, UseTlas=use_tlas)
    pt = gpu_render(sc, cam, w, h, **ov); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    assert_equal(pt, o)
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


def _queries(n, seed, extent, max_dist=3.4028235e+38):
    from idkengine_amd import gputypes as T
    rng = np.random.default_rng(seed)
    r = np.zeros(n, T.RayQuery)
    r["Origin"] = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r["Direction"] = d.astype(np.float32); r["MaxDist"] = max_dist
    return r


@pytest.mark.parametrize("use_tlas", [0, 1])
@pytest.mark.parametrize("any_hit", [False, True])
@pytest.mark.parametrize("lights", [False, True])
def test_ray_queries_match_oracle(native_builder, oracle_mod, use_tlas, any_hit, lights):
    """idkptTraceRays (TraceRay / TraceRayAny with maxDist + traceLights, BVHIntersect.glsl:183-411) == oracle, every field."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.cornell_scene(native_builder, "mixed", True)
    sc.lights = S.make_lights([((0.0, 0.55, 0.2), 0.12, (20.0, 20.0, 20.0)), ((-0.5, -0.2, 0.6), 0.08, (5.0, 2.0, 2.0))])
    rays = _queries(20000, 11 + use_tlas, 1.1)
    rays["MaxDist"][::3] = np.random.default_rng(5).uniform(0.05, 2.0, len(rays[::3])).astype(np.float32)   # a third of the rays are range-limited
    pt = PathTracer(8, 8); pt.UploadScene(sc); pt.UseTlas = use_tlas
    got = pt.TraceRays(rays, any_hit=any_hit, trace_lights=lights)
    want = oracle_mod.trace_rays(sc, rays, any_hit=any_hit, trace_lights=lights, use_tlas=bool(use_tlas))
    assert got.tobytes() == want.tobytes()
    assert (got["Hit"] != 0).any() and (got["Hit"] == 0).any()
    pt.Dispose()


def test_ray_queries_on_soup_match_oracle(native_builder, oracle_mod):
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene_multi(30000, native_builder, parts=5, seed=21)
    rays = _queries(50000, 3, 12.0)
    pt = PathTracer(8, 8); pt.UploadScene(sc)
    for any_hit in (False, True):
        for tl in (0, 1):
            pt.UseTlas = tl
            got = pt.TraceRays(rays, any_hit=any_hit)
            assert got.tobytes() == oracle_mod.trace_rays(sc, rays, any_hit=any_hit, use_tlas=bool(tl)).tobytes()
    pt.Dispose()


@pytest.mark.parametrize("variant,use_tlas", [("mixed", 0), ("mixed", 1), ("blend", 0)])
def test_rt_shadows_match_oracle(native_builder, oracle_mod, variant, use_tlas):
    """idkptTraceShadows (Shaders/ShadowsRayTraced/compute.glsl) == oracle bit for bit on a stand-in G-buffer."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc = S.cornell_scene(native_builder, "mixed", True)
    if variant == "blend":   # make the short box alpha-blended and the tall box alpha-tested: exercises the continue-through-surface loop
        sc.materials["AlphaCutoff"][-2] = 2.0; sc.materials["BaseColorFactor"][-2] = (sc.materials["BaseColorFactor"][-2] & 0x00FFFFFF) | (0x60 << 24)
        sc.materials["AlphaCutoff"][-1] = 0.5; sc.materials["BaseColorFactor"][-1] = (sc.materials["BaseColorFactor"][-1] & 0x00FFFFFF) | (0x40 << 24)
    sc.lights = S.make_lights([((0.0, 0.55, 0.2), 0.12, (20.0, 20.0, 20.0)), ((-0.5, -0.2, 0.6), 0.08, (5.0, 2.0, 2.0))])
    w, h = 96, 80
    cam = S.cornell_camera(w, h)
    rays = S.primary_ray_queries(cam, w, h)
    pt = PathTracer(8, 8); pt.UploadScene(sc); pt.UseTlas = use_tlas
    hits = pt.TraceRays(rays)
    depth, normal = S.gbuffer_from_hits(sc, cam, w, h, rays, hits)
    for light, samples, noise in ((0, 1, 0), (0, 4, 8), (1, 3, 5)):
        p = T.ShadowParams.make(cam.inv_proj_view, w, h, light_index=light, samples=samples, noise_index=noise, jitter=(0.0005, -0.0003))
        keep = np.full((h, w), np.float32(-3.0))
        got = pt.TraceShadows(p, depth, normal, visibility=keep)
        want = oracle_mod.trace_shadows(sc, p, depth, normal, visibility=keep, use_tlas=bool(use_tlas))
        assert (bits(got) == bits(want)).all()
        assert (got == 1.0).any() and (got == 0.0).any()
    pt.Dispose()


def test_plain_c_host_matches_python_host(native_builder, tmp_path):
    """The boundary is a C ABI: a plain C11 program (tests/c_driver/abi_driver.c; gcc, include/idkpt.h, -lidkpt; no Python, torch or
    C++ on its side) uploads the same arrays, renders, and must produce the same bits and counters as the Python host."""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "abi_driver")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(root, "include"), os.path.join(HERE, "c_driver", "abi_driver.c"),
                           "-L", os.path.join(root, "idkengine_amd"), "-lidkpt", "-Wl,-rpath," + os.path.join(root, "idkengine_amd"), "-o", exe])
    sc = S.cornell_scene(native_builder, "mixed", True); w, h = 96, 64; cam = S.cornell_camera(w, h)
    for name in ("blas_nodes", "blas_triangles", "blas_descs", "blas_instances", "tlas_nodes", "vertex_positions", "vertices", "meshes", "materials", "mesh_transforms", "lights"):
        np.ascontiguousarray(getattr(sc, name)).tofile(str(tmp_path / (name + ".bin")))
    np.ascontiguousarray(sc.sky_faces, np.float32).tofile(str(tmp_path / "sky_faces.bin"))
    np.concatenate([cam.inv_projection, cam.inv_view, cam.position.astype(np.float32)]).astype(np.float32).tofile(str(tmp_path / "camera.bin"))
    for use_tlas in (0, 1):
        out = subprocess.run([exe, str(tmp_path), str(w), str(h), "4", "2", str(use_tlas)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.startswith("ok ")
        got = np.fromfile(str(tmp_path / "result.bin"), np.float32).reshape(h, w, 4)
        rays, pairs, tris, acc = (int(x) for x in open(str(tmp_path / "stats.txt")).read().split())
        pt = gpu_render(sc, cam, w, h, RayDepth=4, SamplesPerPixel=2, UseTlas=use_tlas)
        st = pt.stats()
        assert (bits(got) == bits(pt.Result)).all()
        assert (rays, pairs, tris, acc) == (st["rays_traced"], st["node_pair_visits"], st["triangle_tests"], pt.AccumulatedSamples)
        pt.Dispose()


@pytest.fixture(scope="module")
def soup1m(native_builder):
    return S.soup_scene(1000000, native_builder, seed=1)


def test_full_size_headline_frame_properties(soup1m, oracle_mod):
    """BASELINE.json configs[2] at full size: 1M triangles, 1920x1080, RayDepth 2.  Size-independent properties +
    a strided-row exact comparison against the oracle (every 16th row = 67 rows, seconds on CPU)."""
    w, h = 1920, 1080; cam = S.Camera(w, h)
    pt = gpu_render(soup1m, cam, w, h, RayDepth=2)
    img = pt.Result; st = pt.stats()
    assert np.isfinite(img).all() and img[..., :3].max() <= 1.0 and (img[..., 3] == 1.0).all()
    t, tri, _ = pt.primary_hits()
    assert (img.reshape(-1, 4)[tri == 0xFFFFFFFF, :3] == 1.0).all()                 # white sky on every miss
    assert st["alive_counts"][1] == int((tri != 0xFFFFFFFF).sum())                   # every primary hit continues (opaque diffuse, no RR on the first hit)
    assert st["rays_traced"] == w * h + st["alive_counts"][1]
    # determinism: same frame twice -> identical bits
    pt.ResetAccumulation(); pt.Compute()
    assert (bits(pt.Result) == bits(img)).all()
    # sort on == sort off at depth 2
    pt.DoRaySorting = 1; pt.ResetAccumulation(); pt.Compute()
    assert (bits(pt.Result) == bits(img)).all()
    # exact oracle comparison on a row shard (rows y % 16 == 3)
    o = oracle_mod.OraclePathTracer(soup1m, w, h, row_modulo=16, row_remainder=3); o.set_camera(cam); o.settings.RayDepth = 2; o.render()
    assert (bits(img[3::16]) == bits(o.image())).all()
    pt.Dispose(); o.close()


@pytest.mark.parametrize("depth,sort,batch", [(2, 0, 4), (5, 0, 2), (5, 1, 3)])
def test_full_size_frames_are_bit_exact(soup1m, oracle_mod, depth, sort, batch):
    """BASELINE.json's full size (1M triangles, 1920x1080), whole frames against the oracle (OpenMP over the host cores of the GPU
    box: a few seconds): image, ray state, queue, counters — 3 accumulated samples traced `batch` at a time."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    w, h = 1920, 1080; cam = S.Camera(w, h)
    ov = dict(RayDepth=depth, DoRaySorting=sort)
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); pt.UploadScene(soup1m); pt.SetCamera(cam)
    pt.set_max_batch(batch); pt.enable_counters(True)
    o = oracle_render(oracle_mod, soup1m, cam, w, h, frames=3, **ov)
    for _ in range(3):
        pt.Compute()
    assert (bits(pt.Result) == bits(o.image(0))).all()
    assert pt.rays().tobytes() == o.rays().tobytes() and (pt.alive_queue() == o.alive_queue()).all()
    gs, os_ = pt.stats(), o.stats()
    assert gs["rays_traced"] == os_["rays_traced"] and gs["node_pair_visits"] == os_["node_pair_visits"] and gs["triangle_tests"] == os_["triangle_tests"]
    pt.Dispose(); o.close()


@pytest.mark.parametrize("batch", [1, 3])
def test_exact_deep_paths_across_contexts(native_builder, oracle_mod, batch):
    """idkptSetRowRange + idkptSetBounceExchange: three contexts (one per strip, driven by three host threads in lockstep, the
    exchange function summing the counts of the strips above) reproduce the single-context frame bit for bit at RayDepth 6 —
    image, ray state and the total ray count — also when several accumulated samples are traced per batch."""
    import threading
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import dist as D, gputypes as T
    sc = S.soup_scene(30000, native_builder, seed=6, extent=3.0); w, h = 200, 131; cam = S.Camera(w, h, position=(0.0, 0.0, 7.0))
    ov = dict(RayDepth=6)
    frames = 3
    one = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); one.UploadScene(sc); one.SetCamera(cam)
    for _ in range(frames):
        one.Compute()
    want = one.Result; want_rays = one.rays(); want_count = one.stats()["rays_traced"]
    world = 3
    barrier = threading.Barrier(world)
    board = {}

    def exchange_for(rank):
        def fn(bounce, counts):
            board[(bounce, rank)] = counts.copy()
            barrier.wait(timeout=60)
            base = np.zeros(len(counts), np.uint32)
            for r in range(rank):
                base += board[(bounce, r)]
            barrier.wait(timeout=60)                     # nobody overwrites the board before everybody has read it
            return base
        return fn

    pts, errs = [], []
    for r in range(world):
        p = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); p.UploadScene(sc); p.SetCamera(cam)
        first, count = D.strip_of_rank(h, world, r); p.SetRowRange(first, count); p.SetBounceExchange(exchange_for(r)); p.set_max_batch(batch)
        pts.append(p)

    def run(p):
        try:
            for _ in range(frames):
                p.Compute()
            p.flush(); p.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append(e); barrier.abort()
    threads = [threading.Thread(target=run, args=(p,)) for p in pts]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errs, errs
    got = np.concatenate([p.Result for p in pts]); got_rays = np.concatenate([p.rays() for p in pts])
    assert (bits(got) == bits(want)).all()
    assert got_rays.tobytes() == want_rays.tobytes()
    assert sum(p.stats()["rays_traced"] for p in pts) == want_count
    # control: the same strips without the exchange differ at this depth
    q = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); q.UploadScene(sc); q.SetCamera(cam)
    first, count = D.strip_of_rank(h, world, 1); q.SetRowRange(first, count)
    for _ in range(frames):
        q.Compute()
    assert (bits(q.Result) != bits(want[first:first + count])).any()
    for p in pts + [one, q]:
        p.Dispose()


@pytest.mark.parametrize("batch,use_tlas", [(1, 0), (4, 0), (5, 1)])
def test_frame_ring_frames_equal_stand_alone_frames(native_builder, oracle_mod, batch, use_tlas):
    """idkptSetFrameRing: 6 frames with 6 different cameras (and 2 spp each) queued back to back into a ring of 8 slots and traced
    `batch` samples at a time — every frame's image must equal that frame rendered alone, and the oracle, bit for bit (per-sample
    camera, per-sample tile classification, per-frame result images)."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc = S.soup_scene_multi(9000, native_builder, parts=3, seed=2, extent=3.0) if use_tlas else S.soup_scene(9000, native_builder, seed=2, extent=3.0)
    w, h = 150, 90
    cams = [S.Camera(w, h, position=(0.3 * k - 0.8, 0.1 * k, 8.0 - 0.7 * k), view_dir=(0.05 * k - 0.1, -0.02 * k, -1.0), fovy_deg=60.0 + 5 * k) for k in range(6)]
    ov = dict(RayDepth=4, SamplesPerPixel=2, UseTlas=use_tlas)
    ring = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); ring.UploadScene(sc)
    ring.SetFrameRing(8); ring.set_max_batch(batch)
    slots = []
    for cam in cams:
        slots.append(ring.BeginFrame()); ring.SetCamera(cam); ring.Compute()
    assert slots == list(range(6))                    # slots are handed out from 0 after idkptSetFrameRing
    for k, cam in enumerate(cams):
        alone = gpu_render(sc, cam, w, h, **ov)
        got = ring.FrameResult(slots[k])
        assert (bits(got) == bits(alone.Result)).all(), k
        if k in (0, 5):
            o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
            assert (bits(got) == bits(o.image(0))).all()
            o.close()
        alone.Dispose()
    # progressive accumulation inside one slot still works with the ring on: 2 more samples into the last frame
    ring.Compute()
    two = gpu_render(sc, cams[-1], w, h, frames=2, **ov)
    assert (bits(ring.FrameResult(slots[-1])) == bits(two.Result)).all() and ring.AccumulatedSamples == 4
    ring.Dispose(); two.Dispose()


def test_maximum_batch_of_256_samples(native_builder):
    """idkptSetMaxBatch(256): 200 accumulated samples of a small frame traced by ONE set of launches (5 radix passes over key + sample
    index) equal 200 samples traced one at a time."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.cornell_scene(native_builder, "mixed", True); w, h = 33, 21; cam = S.cornell_camera(w, h)
    res = []
    for batch in (256, 1):
        p = PathTracer(w, h); p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 5; p.DoRaySorting = 1; p.set_max_batch(batch)
        for _ in range(200):
            p.Compute()
        res.append((p.Result, p.rays(), p.stats()["rays_traced"], p.AccumulatedSamples)); p.Dispose()
    assert (bits(res[0][0]) == bits(res[1][0])).all() and res[0][1].tobytes() == res[1][1].tobytes() and res[0][2:] == res[1][2:]
    with pytest.raises(Exception):
        p = PathTracer(w, h); p.set_max_batch(257)


@pytest.mark.parametrize("seed", range(6))
def test_random_api_sequences_match_unbatched_replay(native_builder, seed):
    """State-machine check of the deferral logic: a random sequence of host calls (camera moves, Compute, ResetAccumulation, settings,
    SetMaxBatch, SetSize, scene swap, reads in between) must leave the same image as the same logical sequence replayed on a context
    that never defers (max batch 1)."""
    from idkengine_amd.pathtracer import PathTracer
    rng = np.random.default_rng(100 + seed)
    scenes = [S.cornell_scene(native_builder, "mixed", True), S.soup_scene(4000, native_builder, seed=3, extent=2.5)]
    sizes = [(64, 40), (57, 33)]
    cams = lambda w, h: [S.cornell_camera(w, h), S.Camera(w, h, position=(0.3, 0.2, 5.0), fovy_deg=55.0), S.Camera(w, h, position=(-0.4, 0.1, 4.0), fovy_deg=70.0)]   # noqa: E731
    a = PathTracer(*sizes[0]); b = PathTracer(*sizes[0])
    a.set_max_batch(int(rng.integers(2, 9)))
    size = sizes[0]
    for p in (a, b):
        p.UploadScene(scenes[0]); p.SetCamera(cams(*size)[0]); p.RayDepth = 3
    for step in range(24):
        op = rng.choice(["cam", "compute", "compute", "compute", "reset", "depth", "sort", "batch", "size", "scene", "read", "spp"])
        if op == "cam":
            k = int(rng.integers(0, 3))
            for p in (a, b):
                p.SetCamera(cams(*size)[k])
        elif op == "compute":
            for p in (a, b):
                p.Compute()
        elif op == "reset":
            for p in (a, b):
                p.ResetAccumulation()
        elif op == "depth":
            d = int(rng.integers(1, 6))
            for p in (a, b):
                p.RayDepth = d
        elif op == "sort":
            v = int(rng.integers(0, 2))
            for p in (a, b):
                p.DoRaySorting = v
        elif op == "spp":
            v = int(rng.integers(1, 4))
            for p in (a, b):
                p.SamplesPerPixel = v
        elif op == "batch":
            a.set_max_batch(int(rng.integers(1, 9)))
        elif op == "size":
            size = sizes[int(rng.integers(0, 2))]
            for p in (a, b):
                p.SetSize(*size); p.SetCamera(cams(*size)[0])
        elif op == "scene":
            k = int(rng.integers(0, 2))
            for p in (a, b):
                p.UploadScene(scenes[k])
        elif op == "read":
            assert (bits(a.Result) == bits(b.Result)).all(), (seed, step)
            assert a.AccumulatedSamples == b.AccumulatedSamples
    assert (bits(a.Result) == bits(b.Result)).all(), seed
    assert a.AccumulatedSamples == b.AccumulatedSamples
    for p in (a, b):                       # the ray state is only defined right after a sample
        p.Compute()
    assert a.rays().tobytes() == b.rays().tobytes() and (bits(a.Result) == bits(b.Result)).all()
    a.Dispose(); b.Dispose()


def test_row_sharded_contexts_reassemble_the_frame(soup1m):
    """Two contexts on one GPU, rows y%2==r: the multi-GPU sharding of dist.py without the transport."""
    from idkengine_amd.pathtracer import PathTracer
    w, h = 960, 540; cam = S.Camera(w, h)
    full = gpu_render(soup1m, cam, w, h, RayDepth=2, counters=False, capture=False)
    want = full.Result
    out = np.zeros_like(want)
    for r in range(2):
        p = PathTracer(w, h, row_modulo=2, row_remainder=r); p.UploadScene(soup1m); p.SetCamera(cam); p.RayDepth = 2
        p.Compute(); out[r::2] = p.Result; p.Dispose()
    assert (bits(out) == bits(want)).all()
    full.Dispose()


def test_spp_accumulation_equals_repeated_compute(native_builder):
    sc = S.cornell_scene(native_builder, "mixed"); cam = S.cornell_camera(96, 96)
    a = gpu_render(sc, cam, 96, 96, RayDepth=4, SamplesPerPixel=4)
    b = gpu_render(sc, cam, 96, 96, frames=4, RayDepth=4)
    assert (bits(a.Result) == bits(b.Result)).all() and a.AccumulatedSamples == b.AccumulatedSamples == 4
    a.Dispose(); b.Dispose()


@pytest.mark.parametrize("batch", [2, 4, 8])
def test_deferred_batching_is_bit_identical(batch, native_builder, oracle_mod):
    """idkptSetMaxBatch: up to `batch` consecutive samples are traced by one set of launches.  Accumulating 5 samples
    (one full batch + a partial one, or a single partial one) must equal 5 sequential samples bit-for-bit — radiance, AOVs, the last sample's ray
    state / queue / hit records and the exact ray + visit counters — with sorting on (sample-tagged keys) and depth 5."""
    sc = S.presplit_scene(native_builder); cam = S.presplit_camera(160, 90)
    ov = dict(RayDepth=5, DoRaySorting=1, OutputAOVs=1)
    a = gpu_render(sc, cam, 160, 90, frames=5, **ov)
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    b = PathTracer(160, 90, settings=configs.apply_settings(T.Settings.default(), ov))
    b.UploadScene(sc); b.SetCamera(cam); b.enable_counters(True); b.enable_primary_hit_capture(True)
    b.set_max_batch(batch)
    for _ in range(5):
        b.Compute()
    assert b.AccumulatedSamples == 5
    assert (bits(a.Result) == bits(b.Result)).all()
    assert (bits(a.AlbedoTexture) == bits(b.AlbedoTexture)).all() and (bits(a.NormalTexture) == bits(b.NormalTexture)).all()
    assert a.rays().tobytes() == b.rays().tobytes() and (a.alive_queue() == b.alive_queue()).all()
    at, atri, ab = a.primary_hits(); bt, btri, bb = b.primary_hits()
    assert (atri == btri).all() and (bits(at) == bits(bt)).all() and (bits(ab) == bits(bb)).all()
    sa, sb = a.stats(), b.stats()
    for k in ("rays_traced", "primary_rays", "frames", "node_pair_visits", "triangle_tests"):
        assert sa[k] == sb[k], k
    assert sa["alive_counts"][1:5] == sb["alive_counts"][1:5]
    # and against the oracle
    o = oracle_render(oracle_mod, sc, cam, 160, 90, frames=5, **ov)
    assert (bits(b.Result) == bits(o.image(0))).all() and b.rays().tobytes() == o.rays().tobytes()
    a.Dispose(); b.Dispose(); o.close()


def test_batching_with_ragged_size_and_row_shard(native_builder, oracle_mod):
    """N = 77*11 rows is not a multiple of 64: sample segments are padded (Npad) and the padding must stay inert; combined
    with row sharding (rows y%3==1) and batch 5."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.cornell_scene(native_builder, "mixed"); cam = S.cornell_camera(77, 33)
    p = PathTracer(77, 33, row_modulo=3, row_remainder=1); p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 5; p.DoRaySorting = 1; p.set_max_batch(5)
    for _ in range(7):
        p.Compute()
    o = oracle_mod.OraclePathTracer(sc, 77, 33, row_modulo=3, row_remainder=1); o.set_camera(cam); o.settings.RayDepth = 5; o.settings.DoRaySorting = 1
    for _ in range(7):
        o.render()
    assert (bits(p.Result) == bits(o.image())).all() and p.rays().tobytes() == o.rays().tobytes() and (p.alive_queue() == o.alive_queue()).all()
    assert p.stats()["rays_traced"] == o.stats()["rays_traced"]
    p.Dispose(); o.close()


def test_batched_independent_frames_with_reset(native_builder):
    """The bench pattern: ResetAccumulation + Compute per step, 8 steps deferred into batches of 4; also a camera change
    in the middle must flush (pending samples belong to the old camera)."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene(30000, native_builder, seed=21); cam = S.Camera(320, 180); cam2 = S.Camera(320, 180, position=(2.0, 1.0, 24.0))
    ref = gpu_render(sc, cam, 320, 180, RayDepth=3); want = ref.Result
    ref2 = gpu_render(sc, cam2, 320, 180, RayDepth=3); want2 = ref2.Result
    p = PathTracer(320, 180); p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 3; p.set_max_batch(4)
    for _ in range(8):
        p.ResetAccumulation(); p.Compute()
    assert (bits(p.Result) == bits(want)).all() and p.stats()["frames"] == 8
    p.ResetAccumulation(); p.Compute(); p.ResetAccumulation(); p.Compute()       # 2 pending under cam
    p.SetCamera(cam2)                                                              # flushes them
    p.ResetAccumulation(); p.Compute()
    assert (bits(p.Result) == bits(want2)).all()
    assert p.stats()["rays_traced"] == 10 * ref.stats()["rays_traced"] + ref2.stats()["rays_traced"]
    ref.Dispose(); ref2.Dispose(); p.Dispose()


def test_refit_and_skinning_match_oracle(oracle_mod, oracle_builder, native_builder):
    """Config 5 stand-in: refittable soup, positions displaced, GPU BLAS refit (BLASRefit/compute.glsl) vs BLAS.Refit,
    then a frame on the refitted BVH vs the oracle on the CPU-refitted BVH.  Skinning (Skinning/compute.glsl) with two
    joints vs a numpy restatement."""
    from idkengine_amd import gputypes as T, _lib  # noqa: F401
    sc = S.soup_scene(20000, native_builder, seed=12, refittable=True); cam = S.Camera(320, 180)
    pt = gpu_render(sc, cam, 320, 180, RayDepth=3)
    rng = np.random.default_rng(3)
    moved = (sc.vertex_positions + np.sin(sc.vertex_positions[:, ::-1] * 1.7).astype(np.float32) * np.float32(0.05) + rng.normal(0, 0.01, sc.vertex_positions.shape)).astype(np.float32)
    pt.UpdateBuffer(1, moved)                       # IDKPT_BUF_VERTEX_POSITIONS
    pt.RefitBlas(0)
    got = pt.DownloadBuffer(6, T.GpuBlasNode, len(sc.blas_nodes))      # IDKPT_BUF_BLAS_NODES
    want = oracle_builder.refit(sc.blas_nodes, moved, sc.blas_triangles)
    assert got.tobytes() == want.tobytes()
    pt.ResetAccumulation(); pt.Compute()
    sc2 = sc; sc2.vertex_positions = moved; sc2.blas_nodes = want
    o = oracle_render(oracle_mod, sc2, cam, 320, 180, RayDepth=3)
    assert (bits(pt.Result) == bits(o.image())).all()
    o.close()
    # --- skinning
    n = 500
    un = np.zeros(n, T.GpuUnskinnedVertex)
    un["Position"] = sc.vertex_positions[:n]; un["Normal"] = sc.vertices["Normal"][:n]; un["Tangent"] = sc.vertices["Tangent"][:n]
    un["JointIndices"] = rng.integers(0, 2, (n, 4)); wts = rng.uniform(0, 1, (n, 4)).astype(np.float32); un["JointWeights"] = wts / wts.sum(1, keepdims=True)
    joints = np.zeros((2, 3, 4), np.float32); joints[0, :, :3] = np.eye(3); joints[0, :, 3] = (0.1, 0.0, -0.2)
    c, s_ = np.cos(0.3), np.sin(0.3); joints[1, :, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]; joints[1, :, 3] = (0, 0.3, 0)
    pt.UploadUnskinnedVertices(un); pt.UpdateBuffer(8, joints)          # IDKPT_BUF_JOINT_MATRICES
    pt.Skin(0, 0, 0, n); pt.synchronize()
    pos = pt.DownloadBuffer(1, np.float32, 3 * n).reshape(n, 3)
    f = np.float32
    M = np.zeros((n, 3, 4), f)
    for r in range(3):
        for k in range(4):
            acc = None
            for j in range(4):
                term = un["JointWeights"][:, j].astype(f) * joints[un["JointIndices"][:, j], r, k].astype(f)
                acc = term if acc is None else (acc + term).astype(f)
            M[:, r, k] = acc
    p = un["Position"].astype(f)
    want_pos = np.stack([(((M[:, i, 0] * p[:, 0] + M[:, i, 1] * p[:, 1]).astype(f) + M[:, i, 2] * p[:, 2]).astype(f) + M[:, i, 3] * f(1.0)).astype(f) for i in range(3)], 1)
    assert (bits(pos) == bits(want_pos)).all()
    pt.Dispose()


def test_sharded_frame_over_rccl_world1(native_builder):
    """The multi-GPU driver path of dist.py / bench.py on one GPU: process group "nccl" (RCCL) with world_size 1, scene
    broadcast through GPU tensors, renderer on torch's stream, zero-copy alias of the device image, all_gather."""
    import torch
    import torch.distributed as dist
    from idkengine_amd import dist as D
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sc = D.broadcast_scene(S.cornell_scene(native_builder, "mixed"), src=0, device=torch.device("cuda", 0))
        cam = S.cornell_camera(96, 64)
        r = D.GpuShardRenderer(96, 64, 1, 0, 0); r.upload_scene(sc); r.set_camera(cam); r.pt.RayDepth = 2; r.pt.set_max_batch(4)
        frame = D.ShardedFrame(r, 96, 64)
        for _ in range(4):
            frame.render()
        full_t = frame.gather(); torch.cuda.synchronize()
        full = full_t.cpu().numpy()
        ref = gpu_render(sc, cam, 96, 64, RayDepth=2)
        assert (bits(full) == bits(ref.Result)).all()
        # frame ring over the same transport: 3 frames with their own cameras in flight, one all-gather carrying all three
        r.pt.SetFrameRing(6); r.pt.set_max_batch(3)
        cams = [S.cornell_camera(96, 64), S.Camera(96, 64, position=(0.2, 0.1, 3.0), fovy_deg=45.0), S.Camera(96, 64, position=(-0.3, 0.0, 3.2), fovy_deg=50.0)]
        slots = []
        for c in cams:
            slots.append(r.pt.BeginFrame()); r.set_camera(c); r.pt.Compute()
        assert slots == [0, 1, 2]                    # the first frame after idkptSetFrameRing uses slot 0
        frames_t = frame.gather_frames(slots[0], 3); torch.cuda.synchronize()
        for k, c in enumerate(cams):
            alone = gpu_render(sc, c, 96, 64, RayDepth=2)
            assert (bits(frames_t[k].cpu().numpy()) == bits(alone.Result)).all()
            alone.Dispose()
        ref.Dispose(); r.pt.Dispose()
    finally:
        dist.destroy_process_group()


def test_error_paths_fail_loudly(native_builder):
    from idkengine_amd.pathtracer import PathTracer, IdkPtError
    pt = PathTracer(64, 64)
    with pytest.raises(IdkPtError):
        pt.Compute()                                  # no scene uploaded
    sc = S.cornell_scene(native_builder)
    bad = S.cornell_scene(native_builder); bad.blas_triangles = bad.blas_triangles.copy(); bad.blas_triangles["X"][0] = 10 ** 6
    with pytest.raises(IdkPtError):
        pt.UploadScene(bad)                           # out-of-range vertex index is rejected on the host, never reaches the GPU
    pt.UploadScene(sc)
    with pytest.raises(IdkPtError):
        pt.UseTlas = 1; pt.BuildTlas(np.zeros(0, sc.tlas_nodes.dtype))
    with pytest.raises(IdkPtError):
        pt.RefitBlas(0)                               # BLAS is not refittable
    with pytest.raises(IdkPtError):
        pt.SetSize(8192, 64)                          # FirstHit seeds pack x into 12 bits
    # the adjacent entry points report misuse the same way
    from idkengine_amd import gputypes as T
    pt.UseTlas = 0
    cam = S.cornell_camera(16, 16)
    with pytest.raises(IdkPtError):
        pt.TraceShadows(T.ShadowParams.make(cam.inv_proj_view, 16, 16, light_index=0), np.zeros((16, 16), np.float32), np.zeros((16, 16, 2), np.float32))   # the scene has no lights
    with pytest.raises(IdkPtError):
        pt.SetFrameRing(0)
    with pytest.raises(IdkPtError):
        pt.SetFrameRing(2); pt.FrameResult(5)         # slot outside the ring
    with pytest.raises(IdkPtError):
        pt.SetRowRange(60, 10)                        # strip exceeds the image
    assert len(pt.TraceRays(np.zeros(0, T.RayQuery))) == 0   # empty query is fine
    pt.Dispose()
    # more samples in flight than the device can hold: a clean error, the previous configuration stays usable
    big = PathTracer(4096, 16384); big.UploadScene(sc); big.SetCamera(S.cornell_camera(4096, 16384))
    with pytest.raises(IdkPtError, match="samples in flight"):
        big.set_max_batch(256)                        # 3 ray planes alone would need 3 x 275 GB
    big.SetSize(64, 64); big.SetCamera(S.cornell_camera(64, 64)); big.RayDepth = 2; big.Compute()
    ref = PathTracer(64, 64); ref.UploadScene(sc); ref.SetCamera(S.cornell_camera(64, 64)); ref.RayDepth = 2; ref.Compute()
    assert (bits(big.Result) == bits(ref.Result)).all()
    big.Dispose(); ref.Dispose()
