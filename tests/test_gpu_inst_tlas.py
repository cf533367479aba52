"""The reference's instance loop (several BLAS instances, no UseTlas: BVHIntersect.glsl:275-287) walked through the library's own TLAS (csrc/kernels_trace_inst.hpp,
developer option "inst_tlas"): the loop's results bit for bit — image, ray records, alive queue, primary hits — against the CPU oracle's loop, on scenes chosen for what the
argument has to flag: exact ties between instances, instanced BLASes, PreSplit fragments, transforms that are not rigid, stale leaf boxes."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[2048, 0, 24, -1], ids=["braid2048", "whole_instances", "braid24", "general_array"])
def _from_two_instances_on(monkeypatch, request):
    # the own TLAS's leaves: subtrees of the instances' BLASes (partial re-braiding, k_braid — the default budget opens these small scenes down to their leaves' parents; 24 entries
    # stops half way, so entries of very different sizes sit side by side) or whole instances; "general_array": the same scenes through k_trace_inst<.., TREE 2> (one array, a
    # world-space top whose entries take the ray into their instance's space: option inst_general) instead of the own TLAS
    monkeypatch.setenv("IDKPT_INST_BRAID", str(max(request.param, 0)))
    monkeypatch.setenv("IDKPT_INST_GENERAL", "2" if request.param == -1 else "0")
    monkeypatch.setenv("IDKPT_INST_TLAS", "2")        # (the default starts at 8 instances ...)
    monkeypatch.setenv("IDKPT_INST_TLAS_OVERLAP", "100")   # (... and asks for little overlap between them: these scenes interleave their BLASes on purpose)


def _check(oracle_mod, sc, cam, w, h, frames=1, **ov):
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=frames, **ov)
    a = gpu_render(sc, cam, w, h, counters=False, frames=frames, **ov)
    assert_equal(a, o, counters=False)
    st = a.stats(); a.Dispose(); o.close()
    return st["inst_tlas_flagged_rays"], st["rays_traced"]


@pytest.mark.parametrize("parts,tris,depth,sort,lights", [(2, 3000, 4, 1, 0), (3, 6000, 5, 0, 1), (12, 6000, 4, 1, 1), (60, 12000, 3, 0, 0), (600, 3600, 3, 1, 0)])
def test_soup_parts_equal_the_loop(native_builder, oracle_mod, parts, tris, depth, sort, lights):
    sc = S.soup_scene_multi(tris, native_builder, parts=parts, seed=3 + parts, **(dict(extent=4.0, edge=0.4) if parts == 600 else {})); w, h = 160, 96
    if lights:
        sc.lights = S.make_lights([((0.0, 3.0, 14.0), 0.8, (9.0, 8.0, 7.0))])
    cam = S.Camera(w, h, position=(0.0, 0.0, 11.0), fovy_deg=60.0) if parts == 600 else S.Camera(w, h, position=(1.0, 0.5, 24.0))
    flagged, rays = _check(oracle_mod, sc, cam, w, h, frames=2, RayDepth=depth, DoRaySorting=sort, DoTraceLights=lights)
    assert flagged < 0.02 * rays, (flagged, rays)


def test_batched_samples_and_the_option_switched_off(native_builder, oracle_mod, monkeypatch):
    """Three samples in one batch through the own TLAS == the same frame with the option off (the exact loop) == the oracle; the flagged-ray total only moves with the option on."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene_multi(8000, native_builder, parts=9, seed=21); w, h = 200, 120; cam = S.Camera(w, h, position=(1.0, 0.5, 24.0))
    ov = dict(RayDepth=4, SamplesPerPixel=3)
    o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    res = []
    for opt in (2, 0):
        pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); pt.set_option("inst_tlas", opt); pt.set_option("inst_sieve", 0)
        pt.UploadScene(sc); pt.SetCamera(cam); pt.set_max_batch(3); pt.Compute(); pt.flush()
        assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes() and (pt.alive_queue() == o.alive_queue()).all()
        res.append(pt.stats()["inst_tlas_flagged_rays"]); pt.Dispose()
    assert res[1] == 0
    o.close()


@pytest.mark.parametrize("parts,tris,depth,lights", [(2, 3000, 4, 1), (12, 6000, 4, 0), (70, 14000, 3, 0), (600, 3600, 3, 0)])
def test_the_sieved_exact_loop_as_the_main_kernel(native_builder, oracle_mod, monkeypatch, parts, tris, depth, lights):
    """Option inst_sieve: the kernel that traces the own-TLAS walk's flagged rays (k_trace_inst<P, EXACT>: the loop itself, the instances a ray cannot meet sieved out when the
    wave takes the ray; 70 and 600 instances: masks of 3 and 19 words) run over every ray.  Nothing is flagged on this path."""
    monkeypatch.setenv("IDKPT_INST_TLAS", "0"); monkeypatch.setenv("IDKPT_INST_SIEVE", "2"); monkeypatch.setenv("IDKPT_INST_SIEVE_OVERLAP", "100")
    sc = S.soup_scene_multi(tris, native_builder, parts=parts, seed=40 + parts, **(dict(extent=4.0, edge=0.4) if parts == 600 else {})); w, h = 160, 96
    if lights:
        sc.lights = S.make_lights([((0.0, 3.0, 14.0), 0.8, (9.0, 8.0, 7.0))])
    cam = S.Camera(w, h, position=(0.0, 0.0, 11.0), fovy_deg=60.0) if parts == 600 else S.Camera(w, h, position=(1.0, 0.5, 24.0))
    flagged, rays = _check(oracle_mod, sc, cam, w, h, frames=2, RayDepth=depth, DoRaySorting=1, DoTraceLights=lights)
    assert flagged == 0


@pytest.mark.parametrize("parts,tris", [(12, 6000), (600, 3600)])
def test_closest_hit_queries_through_the_sieve(native_builder, oracle_mod, monkeypatch, parts, tris):
    """idkptTraceRays (closest hit) on a several-instance scene without UseTlas runs the exact loop with its instance sieve wherever a frame would use it or the own TLAS; any-hit
    queries (first found = list order) keep k_trace2's loop.  20 000 random rays, with and without a maximal distance: the oracle's hits, byte for byte."""
    from idkengine_amd.pathtracer import PathTracer
    from gpu_helpers import _queries
    monkeypatch.setenv("IDKPT_INST_SIEVE", "2"); monkeypatch.setenv("IDKPT_INST_SIEVE_OVERLAP", "100")
    sc = S.soup_scene_multi(tris, native_builder, parts=parts, seed=77, **(dict(extent=4.0, edge=0.4) if parts == 600 else {}))
    pt = PathTracer(64, 64); pt.UploadScene(sc)
    for max_dist in (3.4028235e+38, 5.0):
        rays = _queries(20000, 23, 6.0, max_dist=max_dist)
        for any_hit in (False, True):
            assert pt.TraceRays(rays, any_hit=any_hit).tobytes() == oracle_mod.trace_rays(sc, rays, any_hit=any_hit, use_tlas=False).tobytes()
    pt.Dispose()


def _instanced(native_builder, transforms, blas_ids, n_blas=2, tris=1500, seed=5, presplit=False):
    """n_blas soups (optionally with scene-spanning triangles that PreSplit cuts up) and an instance list that uses them several times."""
    blases = []
    for k in range(n_blas):
        tp = S.soup_triangles(tris, seed + 31 * k, 3.0, 0.3)
        if presplit:
            tp = np.concatenate([tp, np.float32([[[-3, -3, -2.5], [3, -3, -2.4], [0, 3, 2.6]], [[-3, 0.1, -3], [3, 0.2, 3], [-3, 0.3, 3]]])])
        p, i, nrm, tan = S.flat_shaded(tp)
        blases.append({"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.7, 0.6, 1.0)), nrm, tan)]})
    sc = S.assemble(blases, native_builder, build_tlas=False)
    inst = np.zeros(len(blas_ids), T.GpuBlasInstance); inst["BlasId"] = blas_ids; inst["MeshTransformId"] = np.arange(len(blas_ids))
    sc.blas_instances = inst
    sc.mesh_transforms = np.concatenate([S.transform_from_matrix(m) for m in transforms])
    return sc


def test_instanced_blases_with_exact_ties(native_builder, oracle_mod):
    """Seven instances of two BLASes; instances 1 and 4 are the same BLAS under the same matrix (every hit of one ties with the other: the loop reports the lower instance,
    i.e. MeshTransformId 1), instance 6 repeats instance 2 shifted by a hair (near ties inside the window)."""
    m = [np.eye(4), S.rotation_y(30.0) @ S.translation((2.0, 0.0, 0.0)), S.rotation_y(-50.0) @ S.translation((-2.5, 0.5, 1.0)), S.translation((0.0, 3.0, -2.0)),
         S.rotation_y(30.0) @ S.translation((2.0, 0.0, 0.0)), S.rotation_y(75.0) @ S.translation((1.0, -3.0, 0.0)), S.rotation_y(-50.0) @ S.translation((-2.5, 0.5 + 1e-6, 1.0))]
    sc = _instanced(native_builder, m, [0, 1, 0, 1, 1, 0, 0])
    w, h = 192, 128; cam = S.Camera(w, h, position=(0.5, 0.5, 13.0), fovy_deg=60.0)
    flagged, rays = _check(oracle_mod, sc, cam, w, h, frames=2, RayDepth=4)
    assert flagged > 100, flagged                                        # the tied instances cover a good part of the frame


def test_presplit_fragments_are_handed_back(native_builder, oracle_mod):
    sc = _instanced(native_builder, [np.eye(4), S.rotation_y(40.0) @ S.translation((1.0, 0.2, 0.0)), S.rotation_y(-70.0) @ S.translation((-1.0, -0.3, 0.5))], [0, 1, 0], presplit=True)
    assert len(sc.blas_triangles) > 2 * 1502                             # PreSplit made fragments
    w, h = 192, 128; cam = S.Camera(w, h, position=(0.5, 0.8, 9.0), fovy_deg=60.0)
    flagged, rays = _check(oracle_mod, sc, cam, w, h, frames=2, RayDepth=3)
    assert flagged > 100, flagged                                        # the big triangles fill most of the view and every hit on one is a fragment's


def test_non_rigid_transforms(native_builder, oracle_mod):
    """Non-uniform scale and shear: the world box of an instance is the image of its root box under the inverse of InvModel (computed by the library in double and padded),
    and T stays a world-space distance because the loop does not renormalise the BLAS-space direction (Ray.glsl:7-12)."""
    sh = np.eye(4); sh[0, 1] = 0.4; sh[2, 0] = -0.3
    m = [np.diag([2.0, 0.5, 1.5, 1.0]), sh @ S.translation((3.0, 0.0, 0.0)), S.rotation_y(20.0) @ np.diag([0.7, 1.8, 0.9, 1.0]) @ S.translation((-3.0, 1.0, 0.0)), S.translation((0.0, -2.0, 1.0)),
         np.diag([40.0, 40.0, 0.02, 1.0]) @ S.translation((0.0, 0.0, -6.0))]
    sc = _instanced(native_builder, m, [0, 1, 1, 0, 1])
    w, h = 192, 128; cam = S.Camera(w, h, position=(0.5, 0.5, 14.0), fovy_deg=60.0)
    flagged, rays = _check(oracle_mod, sc, cam, w, h, frames=2, RayDepth=3)
    assert flagged < 0.05 * rays, (flagged, rays)


def test_updates_rebuild_the_own_tlas_and_the_marks(native_builder, oracle_mod):
    """Moving the instances (idkptUpdateBuffer of the transforms) and moving vertices (positions: triangles leave their leaf boxes -> marked) between frames: every frame equals
    the oracle's frame of the scene in that state."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene_multi(6000, native_builder, parts=10, seed=8); w, h = 160, 96; cam = S.Camera(w, h, position=(1.0, 0.5, 24.0))
    ov = dict(RayDepth=3)
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); pt.UploadScene(sc); pt.SetCamera(cam)
    def same():
        o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        pt.ResetAccumulation(); pt.Compute()
        assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes()
        o.close()
    same()
    rng = np.random.default_rng(4)
    xf = sc.mesh_transforms.copy()
    for i in range(len(xf)):
        xf[i] = S.transform_from_matrix(S.rotation_y(float(rng.uniform(0, 360))) @ S.translation(tuple(rng.uniform(-5, 5, 3))))[0]
    sc.mesh_transforms = xf; pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf)
    same()
    f0 = pt.stats()["inst_tlas_flagged_rays"]
    pos = sc.vertex_positions.copy(); pos[::7] += np.float32(0.05)       # every seventh vertex leaves its leaf box (no refit: the boxes are stale, as the loop sees them)
    sc.vertex_positions = pos; pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, pos)
    same()
    assert pt.stats()["inst_tlas_flagged_rays"] > f0
    pt.Dispose()


# ---- the unified tree (developer option "inst_unify"; k_braid / k_unify_* in csrc/kernels_scene.hpp, k_trace_inst<.., UNI>): every instance carries the same InvModel -------------
def _same_space(native_builder, parts, tris, seed, matrix=None, presplit=False):
    """`parts` interleaved soups, one BLAS each, all under ONE transform (identity, or `matrix`): the reference's usual static scene — one BLAS per mesh of a model."""
    blases = []
    per = tris // parts
    for k in range(parts):
        tp = S.soup_triangles(per, seed + 17 * k, 6.0, 0.35)
        if presplit and k < 2:
            tp = np.concatenate([tp, np.float32([[[-6, -6, -5.0 + k], [6, -6, -4.8 + k], [0, 6, 5.2 - k]]])])
        p, i, nrm, tan = S.flat_shaded(tp)
        blases.append({"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.75, 0.7, 1.0)), nrm, tan)], "transform": matrix})
    return S.assemble(blases, native_builder)


@pytest.mark.parametrize("parts,tris,depth,sort,lights,budget,shape", [(2, 3000, 4, 1, 0, 4096, "identity"), (9, 9000, 5, 0, 1, 4096, "sheared"), (40, 16000, 3, 1, 0, 64, "sheared"), (87, 30000, 3, 0, 0, 4096, "identity"),
                                                                           (5, 6000, 4, 0, 0, 2, "presplit")])
@pytest.mark.parametrize("packet", [1, 2], ids=["lane_walk", "packet_walk_forced"])
def test_unified_tree_equals_the_loop(native_builder, oracle_mod, monkeypatch, request, parts, tris, depth, sort, lights, budget, shape, packet):
    """Scenes of several BLASes in one space — identity, and a transform that is neither rigid nor axis-aligned (the ray is taken into the BLAS space once, with the loop's own
    arithmetic); PreSplit fragments (marked triangles: handed back); budgets from "whole instances only" (2 < instances) to 4 096 subtrees: the loop's image, ray records, alive queue
    and primary hits, bit for bit, and the launches really walked the unified tree."""
    if request.node.callspec.params.get("_from_two_instances_on") != 2048:
        pytest.skip("one run per case: the unified tree does not read inst_braid")
    monkeypatch.setenv("IDKPT_INST_UNIFY", str(budget))
    monkeypatch.setenv("IDKPT_PACKET", str(packet))          # 2: the primary launch is k_trace_packet<.., UNI> over the same tree (its unvouched rays to the exact loop)
    m = None
    if shape == "sheared":
        sh = np.eye(4); sh[0, 1] = 0.35; sh[2, 0] = -0.2
        m = S.rotation_y(33.0) @ np.diag([1.3, 0.8, 1.1, 1.0]) @ sh @ S.translation((0.5, -0.25, 1.0))
    sc = _same_space(native_builder, parts, tris, 50 + parts, matrix=m, presplit=shape == "presplit")
    if lights:
        sc.lights = S.make_lights([((0.0, 3.0, 14.0), 0.8, (9.0, 8.0, 7.0))])
    w, h = 160, 96; cam = S.Camera(w, h, position=(1.0, 0.5, 19.0))
    ov = dict(RayDepth=depth, DoRaySorting=sort, DoTraceLights=lights)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=2, **ov)
    a = gpu_render(sc, cam, w, h, counters=False, frames=2, **ov)
    assert_equal(a, o, counters=False)
    st = a.stats(); a.Dispose(); o.close()
    assert st["inst_unified_launches"] >= depth and st["inst_unified_entries"] >= parts and st["inst_unified_top_depth"] >= 2, st
    assert (st["packet_packets"] > 0) == (packet == 2), st
    assert st["inst_tlas_flagged_rays"] < (0.6 if shape == "presplit" else 0.02) * st["rays_traced"], st


def test_unified_tree_is_not_used_where_transforms_differ_or_a_blas_is_instanced(native_builder, oracle_mod, monkeypatch, request):
    if request.node.callspec.params.get("_from_two_instances_on") != 2048:
        pytest.skip("one run")
    monkeypatch.setenv("IDKPT_INST_UNIFY", "4096")
    w, h = 160, 96; cam = S.Camera(w, h, position=(1.0, 0.5, 24.0))
    sc = S.soup_scene_multi(6000, native_builder, parts=4, seed=9)                         # rotated parts: four InvModels
    flagged, rays = _check(oracle_mod, sc, cam, w, h, RayDepth=3)
    from idkengine_amd.pathtracer import PathTracer
    for scene in (sc, _instanced(native_builder, [np.eye(4), np.eye(4), np.eye(4)], [0, 1, 0])):   # ... and one InvModel, but BLAS 0 twice
        pt = PathTracer(w, h); pt.UploadScene(scene); pt.SetCamera(cam); pt.RayDepth = 3; pt.Compute(); pt.flush()
        st = pt.stats(); pt.Dispose()
        assert st["inst_unified_launches"] == 0 and st["inst_unified_entries"] == 0, st


def test_unified_tree_follows_transform_and_vertex_updates(native_builder, oracle_mod, monkeypatch, request):
    """All transforms moved together (still one space: re-derived, still unified), then one of them alone (no longer one space: the own TLAS / the loop takes over), then vertices moved
    without a refit (stale boxes: marked triangles): every frame equals the oracle's frame of the scene in that state."""
    if request.node.callspec.params.get("_from_two_instances_on") != 2048:
        pytest.skip("one run")
    from idkengine_amd.pathtracer import PathTracer
    monkeypatch.setenv("IDKPT_INST_UNIFY", "4096")
    sc = _same_space(native_builder, 6, 6000, 71); w, h = 160, 96; cam = S.Camera(w, h, position=(1.0, 0.5, 19.0))
    ov = dict(RayDepth=3)
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); pt.UploadScene(sc); pt.SetCamera(cam)
    def same():
        o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        pt.ResetAccumulation(); pt.Compute()
        assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes()
        o.close()
    same(); assert pt.stats()["inst_unified_entries"] >= 6
    xf = sc.mesh_transforms.copy()
    for i in range(len(xf)):
        xf[i] = S.transform_from_matrix(S.rotation_y(40.0) @ S.translation((1.0, 0.5, -1.0)))[0]
    sc.mesh_transforms = xf; pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf)
    same(); assert pt.stats()["inst_unified_entries"] >= 6
    xf = xf.copy(); xf[2] = S.transform_from_matrix(S.rotation_y(-20.0) @ S.translation((-2.0, 0.0, 1.0)))[0]
    sc.mesh_transforms = xf; pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf)
    n0 = pt.stats()["inst_unified_launches"]
    same(); st = pt.stats(); assert st["inst_unified_entries"] == 0 and st["inst_unified_launches"] == n0, st
    xf = xf.copy(); xf[2] = xf[0]
    sc.mesh_transforms = xf; pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf)
    pos = sc.vertex_positions.copy(); pos[::7] += np.float32(0.05)
    sc.vertex_positions = pos; pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, pos)
    f0 = pt.stats()["inst_tlas_flagged_rays"]
    same(); st = pt.stats(); assert st["inst_unified_entries"] >= 6 and st["inst_tlas_flagged_rays"] > f0, st
    pt.Dispose()


def test_general_array_is_what_ran(native_builder, oracle_mod, request):
    """Under the "general_array" parameter the scenes of this file whose instances carry different transforms (or share a BLAS) are walked as ONE array (TREE 2); under the others by
    the own TLAS: the statistics say which."""
    w, h = 160, 96; cam = S.Camera(w, h, position=(1.0, 0.5, 24.0))
    sc = S.soup_scene_multi(6000, native_builder, parts=5, seed=13)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=2, RayDepth=4)
    a = gpu_render(sc, cam, w, h, counters=False, frames=2, RayDepth=4)
    assert_equal(a, o, counters=False)
    st = a.stats(); a.Dispose(); o.close()
    general = request.node.callspec.params.get("_from_two_instances_on") == -1
    assert (st["inst_unified_launches"] > 0) == general and (st["inst_unified_entries"] >= 5) == general, st


def test_unified_tree_is_rederived_after_a_refit(native_builder, oracle_mod, monkeypatch, request):
    """Four refittable BLASes in one space: vertices moved, every BLAS refitted on the device (BLASRefit), the unified tree re-derived from the new boxes: the frame equals the oracle's
    frame of the scene with the moved vertices and the refitted nodes (the refit itself is pinned in tests/test_gpu_scene_updates.py and against the reference's shader in test_glref.py)."""
    if request.node.callspec.params.get("_from_two_instances_on") != 2048:
        pytest.skip("one run")
    from idkengine_amd.pathtracer import PathTracer
    monkeypatch.setenv("IDKPT_INST_UNIFY", "4096")
    blases = []
    for k in range(4):
        p, i, nrm, tan = S.flat_shaded(S.soup_triangles(1500, 90 + 17 * k, 5.0, 0.4))
        blases.append({"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.75, 0.7, 1.0)), nrm, tan)], "refittable": True})
    sc = S.assemble(blases, native_builder); w, h = 160, 96; cam = S.Camera(w, h, position=(1.0, 0.5, 17.0))
    ov = dict(RayDepth=3)
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); pt.UploadScene(sc); pt.SetCamera(cam)
    def same():
        o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        pt.ResetAccumulation(); pt.Compute()
        assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes()
        o.close()
    same(); assert pt.stats()["inst_unified_entries"] >= 4
    rng = np.random.default_rng(5)
    moved = (sc.vertex_positions + np.sin(sc.vertex_positions[:, ::-1] * 1.3).astype(np.float32) * np.float32(0.08) + rng.normal(0, 0.01, sc.vertex_positions.shape)).astype(np.float32)
    pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, moved)
    for b in range(4):
        pt.RefitBlas(b)
    sc.vertex_positions = moved
    sc.blas_nodes = pt.DownloadBuffer(T.IDKPT_BUF_BLAS_NODES, T.GpuBlasNode, len(sc.blas_nodes))
    n0 = pt.stats()["inst_unified_launches"]
    same(); st = pt.stats(); assert st["inst_unified_entries"] >= 4 and st["inst_unified_launches"] > n0, st
    pt.Dispose()


def test_singular_and_non_finite_transforms_terminate_and_equal_the_loop(native_builder, oracle_mod):
    """An instance hidden by a zero scale (Model = InvModel = 0) and one whose InvModel is NaN: their padded world boxes are "all of space" with infinite union areas, which no
    PLOC partner ever picks — the tree builders end anyway (k_tlas_build's guard), the walks hand the rays that meet such an instance to the exact loop, and the frame is the oracle's."""
    m = [np.eye(4), S.rotation_y(30.0) @ S.translation((2.0, 0.0, 0.0)), S.rotation_y(-50.0) @ S.translation((-2.5, 0.5, 1.0)), S.translation((0.0, 3.0, -2.0)),
         S.rotation_y(75.0) @ S.translation((1.0, -3.0, 0.0)), S.translation((0.5, 0.5, 0.5)), S.translation((-1.0, 1.0, 0.0)), S.rotation_y(10.0), S.translation((3.0, -1.0, 1.0))]
    sc = _instanced(native_builder, m, [0, 1, 0, 1, 1, 0, 0, 1, 0])
    xf = sc.mesh_transforms.copy()
    xf["Model"][5] = 0.0; xf["InvModel"][5] = 0.0
    xf["InvModel"][7] = np.float32("nan")
    sc.mesh_transforms = xf
    w, h = 160, 96; cam = S.Camera(w, h, position=(0.5, 0.5, 13.0), fovy_deg=60.0)
    _check(oracle_mod, sc, cam, w, h, frames=2, RayDepth=3)
