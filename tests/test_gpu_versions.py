"""Scene versions (idkptSetSceneVersions): animated frames — skinning, BLAS refit, moved instances, TLAS rebuild (ModelManager.Update,
Source/ModelManager.cs:263-361; Shaders/Skinning/compute.glsl, Shaders/BLASRefit/compute.glsl, Bvh/BVH.cs:278-298) — queued with different states of the
geometry and traced by ONE batch must equal, bit for bit, updating and rendering every frame alone."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402,F401
from gpu_helpers import bits  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def _scene(native_builder, kind):
    """kind 'one': ONE refittable BLAS, every vertex skinned (k_trace2 MODE 0; the updates rewrite whole buffers).  'list' / 'tlas': a skinned refittable
    BLAS plus two rigid instances that move every frame, through the instance loop (MODE 1) / the TLAS (MODE 2); only a part of the vertex, node and
    triangle-record arrays is rewritten per frame, the rest is carried over into the new state."""
    if kind == "one":
        sc = S.soup_scene(9000, native_builder, seed=31, extent=2.0, refittable=True)
        return sc, len(sc.vertex_positions)
    tp = S.soup_triangles(5000, seed=14, extent=1.5, edge=0.25)
    p, i, nrm, tan = S.flat_shaded(tp)
    tp2 = S.soup_triangles(2000, seed=15, extent=1.0, edge=0.3)
    p2, i2, n2, t2 = S.flat_shaded(tp2)
    sc = S.assemble([{"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.6, 0.5, 1.0)), nrm, tan)], "refittable": True},
                     {"meshes": [S.MeshInput(p2, i2, S.make_material((0.5, 0.7, 0.9, 1.0), metallic=0.5, roughness=0.3), n2, t2)], "transform": S.translation((3.0, 0.0, 0.0))},
                     {"meshes": [S.MeshInput(p2, i2, S.make_material((0.6, 0.9, 0.5, 1.0), emissive=(0.5, 0.2, 0.1)), n2, t2)], "transform": S.translation((-3.0, 0.5, 0.0))}], native_builder, sky_color=(0.7, 0.8, 1.0))
    return sc, len(p)


def _unskinned(sc, nskin, T):
    rng = np.random.default_rng(5)
    un = np.zeros(nskin, T.GpuUnskinnedVertex)
    un["Position"] = sc.vertex_positions[:nskin]; un["Normal"] = sc.vertices["Normal"][:nskin]; un["Tangent"] = sc.vertices["Tangent"][:nskin]
    un["JointIndices"] = rng.integers(0, 2, (nskin, 4)); wts = rng.uniform(0, 1, (nskin, 4)).astype(np.float32); un["JointWeights"] = wts / wts.sum(1, keepdims=True)
    return un


def _update(pt, sc, T, kind, nskin, frame):
    a = 0.2 * (frame + 1)
    joints = np.zeros((2, 3, 4), np.float32)
    joints[0, :, :3] = np.eye(3); joints[0, :, 3] = (0.1 * a, 0.05, -0.2 * a)
    c_, s_ = np.cos(a), np.sin(a); joints[1, :, :3] = [[c_, 0, s_], [0, 1, 0], [-s_, 0, c_]]; joints[1, :, 3] = (0.0, 0.3 * a, 0.0)
    pt.UpdateBuffer(T.IDKPT_BUF_JOINT_MATRICES, joints); pt.Skin(0, 0, 0, nskin); pt.RefitBlas(0)
    if kind != "one":
        xf = sc.mesh_transforms.copy()
        xf[1] = S.transform_from_matrix(S.rotation_y(40.0 * a) @ S.translation((3.0 - a, 0.2 * a, 0.0)))[0]
        xf[2] = S.transform_from_matrix(S.rotation_y(-25.0 * a) @ S.translation((-3.0 + 0.5 * a, 0.5, a)))[0]
        pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf)
        if kind == "tlas":
            pt.BuildTlasOnDevice()


def _run(native_builder, kind, depth, frames, versions, batch, devices, probe_at=None, cams=None):
    """`frames` animated frames, one sample each, into a frame ring; versions = 1: every update launches what is queued (the reference's order of events)."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc, nskin = _scene(native_builder, kind)
    w, h = 136, 90
    pt = PathTracer(w, h, devices=([0] * devices if devices > 1 else None))
    pt.UploadScene(sc); pt.UseTlas = 1 if kind == "tlas" else 0; pt.RayDepth = depth
    pt.UploadUnskinnedVertices(_unskinned(sc, nskin, T))
    pt.SetSceneVersions(versions); pt.SetFrameRing(frames); pt.set_max_batch(batch)
    pt.enable_counters(True)
    slots, probes = [], []
    for f in range(frames):
        _update(pt, sc, T, kind, nskin, f)
        slots.append(pt.BeginFrame())
        pt.SetCamera(cams[f] if cams else S.Camera(w, h, position=(0.0, 0.5, 9.0), fovy_deg=60.0))
        pt.Compute()
        if probe_at is not None and f == probe_at:
            probes.append(pt.rays().tobytes())                      # a reader in the middle of the sequence: launches what is queued, completes the deferred bounce
    images = [pt.FrameResult(s) for s in slots]
    state = (pt.DownloadBuffer(T.IDKPT_BUF_BLAS_NODES, T.GpuBlasNode, len(sc.blas_nodes)).tobytes(), pt.DownloadBuffer(T.IDKPT_BUF_VERTICES, T.GpuVertex, len(sc.vertices)).tobytes(),
             pt.DownloadBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, T.GpuMeshTransform, len(sc.mesh_transforms)).tobytes(), pt.DownloadBuffer(T.IDKPT_BUF_TLAS_NODES, T.GpuTlasNode, len(sc.tlas_nodes)).tobytes() if kind == "tlas" else b"")
    st = pt.stats()
    rays = pt.rays().tobytes() if devices == 1 or depth > 2 else b""     # (row-dealt members at RayDepth <= 2: images are exact, the unread continuation of the last bounce is not, test_gpu_multi.py)
    pt.Dispose()
    return images, state, (st["rays_traced"], st["node_pair_visits"], st["triangle_tests"]), probes, rays


@pytest.mark.parametrize("kind,depth,devices", [("one", 2, 1), ("one", 5, 1), ("list", 3, 1), ("tlas", 4, 1), ("one", 2, 2), ("tlas", 4, 2)])
def test_animated_frames_in_one_batch_equal_frames_rendered_alone(native_builder, kind, depth, devices):
    frames = 6
    want = _run(native_builder, kind, depth, frames, versions=1, batch=1, devices=devices)
    assert len({im.tobytes() for im in want[0]}) == frames             # the frames do differ
    for versions, batch, probe in ((12, 6, None), (7, 6, None), (3, 6, None), (2, 4, 2), (12, 3, 4)):
        got = _run(native_builder, kind, depth, frames, versions=versions, batch=batch, devices=devices, probe_at=probe)
        for f in range(frames):
            assert (bits(got[0][f]) == bits(want[0][f])).all(), (kind, versions, batch, f)
        assert got[1] == want[1], (kind, versions, batch)               # the final state of the geometry
        assert got[2] == want[2], (kind, versions, batch)               # rays traced, node-pair visits, triangle tests
        assert got[4] == want[4], (kind, versions, batch)               # per-pixel ray state of the last frame
    # a reader in the middle sees the same state either way
    a = _run(native_builder, kind, depth, frames, versions=1, batch=1, devices=devices, probe_at=3)
    b = _run(native_builder, kind, depth, frames, versions=8, batch=6, devices=devices, probe_at=3)
    if devices == 1 or depth > 2:
        assert a[3] == b[3]


def test_scene_versions_with_moving_cameras_and_unversioned_updates(native_builder):
    """Every frame its own camera AND its own geometry in one batch; an update of something that is not versioned (a material) launches what is queued first."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd._lib import IdkPtError
    from idkengine_amd import gputypes as T
    w, h = 136, 90
    cams = [S.Camera(w, h, position=(0.3 * k - 0.8, 0.5 + 0.1 * k, 9.0 - 0.4 * k), fovy_deg=55.0 + k) for k in range(6)]
    want = _run(native_builder, "tlas", 3, 6, versions=1, batch=1, devices=1, cams=cams)
    got = _run(native_builder, "tlas", 3, 6, versions=12, batch=6, devices=1, cams=cams)
    for f in range(6):
        assert (bits(got[0][f]) == bits(want[0][f])).all(), f
    # material patch between two queued frames
    sc, nskin = _scene(native_builder, "one")
    res = []
    for versions, batch in ((1, 1), (8, 4)):
        pt = PathTracer(w, h); pt.UploadScene(sc); pt.RayDepth = 3; pt.UploadUnskinnedVertices(_unskinned(sc, nskin, T))
        pt.SetSceneVersions(versions); pt.SetFrameRing(4); pt.set_max_batch(batch); pt.SetCamera(S.Camera(w, h, position=(0.0, 0.0, 6.0)))
        slots = []
        for f in range(4):
            _update(pt, sc, T, "one", nskin, f)
            if f == 2:
                m = sc.materials.copy(); m["EmissiveFactor"][0] = (0.4, 0.1, 0.0); pt.UpdateBuffer(T.IDKPT_BUF_MATERIALS, m)
            slots.append(pt.BeginFrame()); pt.Compute()
        res.append([pt.FrameResult(s) for s in slots]); pt.Dispose()
    for f in range(4):
        assert (bits(res[0][f]) == bits(res[1][f])).all(), f
    q = PathTracer(16, 16)
    with pytest.raises(IdkPtError, match="1..64"):
        q.SetSceneVersions(0)
    q.Dispose()
