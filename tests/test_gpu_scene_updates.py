"""R1 BLASRefit / R2 Skinning (Shaders/BLASRefit/compute.glsl, Shaders/Skinning/compute.glsl) and the animated-frame sequence."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402,F401
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal  # noqa: E402,F401

pytestmark = pytest.mark.gpu


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


def _skin_numpy(un, joints):
    """Shaders/Skinning/compute.glsl:14-47 for the positions, binary32, operation for operation (weights * joint matrices summed in order)."""
    f = np.float32
    n = len(un)
    M = np.zeros((n, 3, 4), f)
    for r in range(3):
        for k in range(4):
            acc = None
            for j in range(4):
                term = un["JointWeights"][:, j].astype(f) * joints[un["JointIndices"][:, j], r, k].astype(f)
                acc = term if acc is None else (acc + term).astype(f)
            M[:, r, k] = acc
    p = un["Position"].astype(f)
    return np.stack([(((M[:, i, 0] * p[:, 0] + M[:, i, 1] * p[:, 1]).astype(f) + M[:, i, 2] * p[:, 2]).astype(f) + M[:, i, 3] * f(1.0)).astype(f) for i in range(3)], 1)


@pytest.mark.parametrize("devices", [1, 2])
def test_animated_frames_skin_refit_tlas_render(oracle_mod, oracle_builder, native_builder, devices):
    """SURVEY 8(f) N1 / BASELINE config 5 in one sequence, entirely on the device, three animated frames: joint matrices -> Skinning ->
    BLAS refit of the skinned mesh -> moved instance transforms -> TLAS rebuild on the device -> one frame through the TLAS.  Every
    frame must equal the oracle rendering the scene that numpy skinning + BLAS.Refit + TLAS.Build produce on the host
    (ModelManager.cs:263-361, Bvh/BVH.cs:278-298,472-489).  devices = 2: the same through one multi-device context."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    import copy
    tp = S.soup_triangles(6000, seed=14, extent=1.5, edge=0.25)
    p, i, nrm, tan = S.flat_shaded(tp)
    tp2 = S.soup_triangles(2500, seed=15, extent=1.0, edge=0.3)
    p2, i2, n2, t2 = S.flat_shaded(tp2)
    sc = S.assemble([{"meshes": [S.MeshInput(p, i, S.make_material((0.8, 0.6, 0.5, 1.0)), nrm, tan)], "refittable": True},
                     {"meshes": [S.MeshInput(p2, i2, S.make_material((0.5, 0.7, 0.9, 1.0), metallic=0.5, roughness=0.3), n2, t2)], "transform": S.translation((3.0, 0.0, 0.0))},
                     {"meshes": [S.MeshInput(p2, i2, S.make_material((0.6, 0.9, 0.5, 1.0)), n2, t2)], "transform": S.translation((-3.0, 0.5, 0.0))}], native_builder, sky_color=(0.7, 0.8, 1.0))
    w, h = 144, 96; cam = S.Camera(w, h, position=(0.0, 0.5, 9.0), fovy_deg=60.0)
    ids = [0] * devices if devices > 1 else None
    pt = PathTracer(w, h, devices=ids); pt.UploadScene(sc); pt.SetCamera(cam); pt.UseTlas = 1; pt.RayDepth = 4
    nskin = len(p)                                           # the first BLAS's vertices are the skinned ones
    rng = np.random.default_rng(5)
    un = np.zeros(nskin, T.GpuUnskinnedVertex)
    un["Position"] = sc.vertex_positions[:nskin]; un["Normal"] = sc.vertices["Normal"][:nskin]; un["Tangent"] = sc.vertices["Tangent"][:nskin]
    un["JointIndices"] = rng.integers(0, 2, (nskin, 4)); wts = rng.uniform(0, 1, (nskin, 4)).astype(np.float32); un["JointWeights"] = wts / wts.sum(1, keepdims=True)
    pt.UploadUnskinnedVertices(un)
    host = copy.copy(sc); host.vertex_positions = sc.vertex_positions.copy(); host.blas_nodes = sc.blas_nodes.copy(); host.mesh_transforms = sc.mesh_transforms.copy()
    o = None
    for frame in range(3):
        a = 0.25 * (frame + 1)
        joints = np.zeros((2, 3, 4), np.float32)
        joints[0, :, :3] = np.eye(3); joints[0, :, 3] = (0.1 * a, 0.05, -0.2 * a)
        c_, s_ = np.cos(a), np.sin(a); joints[1, :, :3] = [[c_, 0, s_], [0, 1, 0], [-s_, 0, c_]]; joints[1, :, 3] = (0.0, 0.3 * a, 0.0)
        xf = sc.mesh_transforms.copy()
        xf[1] = S.transform_from_matrix(S.rotation_y(40.0 * a) @ S.translation((3.0 - a, 0.2 * a, 0.0)))[0]
        xf[2] = S.transform_from_matrix(S.rotation_y(-25.0 * a) @ S.translation((-3.0 + 0.5 * a, 0.5, a)))[0]
        # ---- device
        pt.UpdateBuffer(T.IDKPT_BUF_JOINT_MATRICES, joints); pt.Skin(0, 0, 0, nskin); pt.RefitBlas(0)
        pt.UpdateBuffer(T.IDKPT_BUF_MESH_TRANSFORMS, xf); pt.BuildTlasOnDevice()
        pt.ResetAccumulation(); pt.Compute()
        # ---- host equivalent for the oracle
        host.vertex_positions[:nskin] = _skin_numpy(un, joints)
        d0 = host.blas_descs[0]
        nodes0 = oracle_builder.refit(host.blas_nodes[d0["NodeOffset"]: d0["NodeOffset"] + d0["NodeCount"]], host.vertex_positions,
                                      host.blas_triangles[d0["TriangleOffset"]: d0["TriangleOffset"] + d0["TriangleCount"]])
        host.blas_nodes[d0["NodeOffset"]: d0["NodeOffset"] + d0["NodeCount"]] = nodes0
        host.mesh_transforms = xf
        S.rebuild_tlas(host, oracle_builder)
        host.vertices = pt.DownloadBuffer(T.IDKPT_BUF_VERTICES, T.GpuVertex, len(sc.vertices))    # skinned + re-compressed normals / tangents (positions are checked against numpy below)
        assert pt.DownloadBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, np.float32, 3 * nskin).tobytes() == host.vertex_positions[:nskin].tobytes(), frame
        assert pt.DownloadBuffer(T.IDKPT_BUF_BLAS_NODES, T.GpuBlasNode, len(host.blas_nodes)).tobytes() == host.blas_nodes.tobytes(), frame
        assert pt.DownloadBuffer(T.IDKPT_BUF_TLAS_NODES, T.GpuTlasNode, len(host.tlas_nodes)).tobytes() == host.tlas_nodes.tobytes(), frame
        if o is not None:
            o.close()
        o = oracle_render(oracle_mod, host, cam, w, h, RayDepth=4, UseTlas=1)
        assert (bits(pt.Result) == bits(o.image(0))).all(), frame
    o.close(); pt.Dispose()
