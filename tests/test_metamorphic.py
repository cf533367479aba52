"""A second opinion on the oracle and the HIP path that shares code with neither: a float64 brute-force closest-hit
(Moeller-Trumbore, no BVH) over every triangle, and two metamorphic relations that hold for any correct tracer —
  * a rigidly moved scene seen by a rigidly moved ray hits the same triangles (BVHIntersect.glsl:231-232: rays go to BLAS-local space,
    T stays a world-space distance);
  * k instances of one BLAS hit what the mesh with the k copies baked into one BLAS hits.
The traversal under test works in binary32 with the reference's operation order, the checker in binary64 with a different formula, so
the comparisons allow what rounding allows: T within 1e-4 relative (north_star's tolerance), triangle identity except where the two
nearest candidates are closer together than that."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402

REL = 1e-4


def world_triangles(sc):
    """(n,3,3) float64 world-space corners of every stored BLAS triangle of every instance, + (instance, blas triangle id) per row."""
    tris, owner = [], []
    for ii, inst in enumerate(sc.blas_instances):
        d = sc.blas_descs[inst["BlasId"]]
        t = sc.blas_triangles[d["TriangleOffset"]: d["TriangleOffset"] + d["TriangleCount"]]
        p = sc.vertex_positions[np.stack([t["X"], t["Y"], t["Z"]], 1).reshape(-1)].astype(np.float64).reshape(-1, 3, 3)
        m = sc.mesh_transforms[inst["MeshTransformId"]]["Model"].astype(np.float64)          # 3x4: world = M[:, :3] @ p + M[:, 3]
        tris.append(p @ m[:, :3].T + m[:, 3])
        owner.append(np.stack([np.full(len(t), ii), np.arange(len(t)) + d["TriangleOffset"]], 1))
    return np.concatenate(tris), np.concatenate(owner)


_BRUTE = None


def brute_force(tris, origins, dirs):
    """Closest hit of every ray against every triangle, float64 Moeller-Trumbore in plain C with OpenMP (tests/c_driver/brute_force.c,
    built here with gcc); returns (t, row index or -1, runner-up t)."""
    global _BRUTE
    import ctypes as C
    import subprocess
    import tempfile
    if _BRUTE is None:
        so = os.path.join(tempfile.mkdtemp(prefix="idkpt_brute_"), "libbrute.so")
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", os.path.join(HERE, "c_driver", "brute_force.c"), "-o", so, "-lm"])
        _BRUTE = C.CDLL(so)
        _BRUTE.brute_force.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _BRUTE.brute_force.restype = None
    tr = np.ascontiguousarray(tris, np.float64).reshape(-1, 9); o = np.ascontiguousarray(origins, np.float64); d = np.ascontiguousarray(dirs, np.float64)
    t = np.zeros(len(o)); i = np.zeros(len(o), np.int64); t2 = np.zeros(len(o))
    _BRUTE.brute_force(tr.ctypes.data, len(tr), o.ctypes.data, d.ctypes.data, len(o), t.ctypes.data, i.ctypes.data, t2.ctypes.data)
    return t, i, t2


def random_rays(n, seed, extent):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, T.RayQuery)
    r["Origin"] = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r["Direction"] = d.astype(np.float32); r["MaxDist"] = 3.4028235e+38
    return r


def check_against_brute_force(sc, rays, hits):
    tris, owner = world_triangles(sc)
    bt, bi, bt2 = brute_force(tris, rays["Origin"].astype(np.float64), rays["Direction"].astype(np.float64))
    hit = hits["Hit"] != 0
    # grazing hits (edge / vertex / parallel) can be found by one arithmetic and missed by the other: they show up as a hit whose runner-up
    # or whose own barycentrics are within rounding of the edge; everything else must agree on hit / miss
    disagree = hit != np.isfinite(bt)
    assert disagree.mean() < 2e-3, disagree.mean()
    both = hit & np.isfinite(bt)
    rel = np.abs(hits["T"][both].astype(np.float64) - bt[both]) / np.maximum(bt[both], 1e-6)
    ambiguous = np.abs(bt2[both] - bt[both]) <= REL * np.maximum(bt[both], 1e-6) * 4       # two candidates nearly equally near
    assert (rel[~ambiguous] < REL).mean() > 0.999, float((rel[~ambiguous] < REL).mean())
    same_tri = owner[bi[both], 1] == hits["TriangleId"][both]
    assert same_tri[~ambiguous & (rel < REL)].mean() > 0.999
    return int(both.sum())


SOUP_N, RAYS_N = 100000, 10000


@pytest.fixture(scope="module")
def soup100k(native_builder):
    return S.soup_scene(SOUP_N, native_builder, seed=31, extent=4.0, edge=0.25)


def test_oracle_closest_hit_equals_brute_force(oracle_mod, soup100k):
    """10 000 random rays into a 100 000-triangle soup: the oracle's BVH traversal (the checker of every GPU parity test) against
    a brute force that knows nothing about BVHs."""
    rays = random_rays(RAYS_N, 5, 4.0)
    n = check_against_brute_force(soup100k, rays, oracle_mod.trace_rays(soup100k, rays))
    assert n > RAYS_N // 2


def _moved(sc_builder, native_builder, M):
    return sc_builder(native_builder, transform=M)


def _rigid(deg_y, deg_x, t):
    ay, ax = np.radians(deg_y), np.radians(deg_x)
    ry = np.eye(4); ry[0, 0] = np.cos(ay); ry[0, 2] = -np.sin(ay); ry[2, 0] = np.sin(ay); ry[2, 2] = np.cos(ay)
    rx = np.eye(4); rx[1, 1] = np.cos(ax); rx[1, 2] = np.sin(ax); rx[2, 1] = -np.sin(ax); rx[2, 2] = np.cos(ax)
    tr = np.eye(4); tr[3, :3] = t
    return ry @ rx @ tr                       # OpenTK row-vector convention: p_world = p_local @ M


def _relation_rigid(trace, native_builder):
    import configs
    M = _rigid(37.0, -21.0, (1.5, -0.7, 2.2))
    a = configs.helmet_scene(native_builder); b = configs.helmet_scene(native_builder, transform=M)
    rays = random_rays(20000, 9, 1.6)
    # aim most rays at the mesh
    rays["Direction"] = (-rays["Origin"] / np.maximum(np.linalg.norm(rays["Origin"], axis=1, keepdims=True), 1e-6) + np.random.default_rng(2).normal(0, 0.35, (len(rays), 3))).astype(np.float32)
    rays["Direction"] /= np.linalg.norm(rays["Direction"], axis=1, keepdims=True)
    moved = rays.copy()
    moved["Origin"] = (np.c_[rays["Origin"].astype(np.float64), np.ones(len(rays))] @ M)[:, :3].astype(np.float32)
    moved["Direction"] = (rays["Direction"].astype(np.float64) @ M[:3, :3]).astype(np.float32)
    ha, hb = trace(a, rays), trace(b, moved)
    assert (ha["Hit"] != 0).mean() > 0.3
    assert ((ha["Hit"] != 0) != (hb["Hit"] != 0)).mean() < 2e-3
    both = (ha["Hit"] != 0) & (hb["Hit"] != 0)
    assert (ha["TriangleId"][both] == hb["TriangleId"][both]).mean() > 0.998              # same BLAS -> same triangle numbering
    same = both & (ha["TriangleId"] == hb["TriangleId"])
    assert (np.abs(ha["T"][same] - hb["T"][same]) <= REL * np.maximum(ha["T"][same], 1e-6)).mean() > 0.999   # T is a world-space distance in both


def _relation_instances(trace, native_builder):
    rng = np.random.default_rng(12)
    tp = S.soup_triangles(3000, seed=8, extent=1.0, edge=0.3)
    p, i, nrm, tan = S.flat_shaded(tp)
    mat = S.make_material((0.7, 0.7, 0.7, 1.0))
    Ms = [_rigid(20.0 * k, 11.0 * k, (2.6 * (k % 3) - 2.6, 2.4 * (k // 3) - 1.2, 0.3 * k)) for k in range(6)]
    inst = S.assemble([{"meshes": [S.MeshInput(p, i, mat, nrm, tan)], "transform": M} for M in Ms], native_builder)
    baked_tp = np.concatenate([(np.c_[tp.reshape(-1, 3).astype(np.float64), np.ones(len(tp) * 3)] @ M)[:, :3].astype(np.float32).reshape(-1, 3, 3) for M in Ms])
    pb, ib, nb, tb = S.flat_shaded(baked_tp)
    baked = S.assemble([{"meshes": [S.MeshInput(pb, ib, mat, nb, tb)]}], native_builder)
    rays = random_rays(20000, 4, 4.0)
    for use_tlas in (False, True):
        hi, hb = trace(inst, rays, use_tlas), trace(baked, rays, False)
        assert ((hi["Hit"] != 0) != (hb["Hit"] != 0)).mean() < 2e-3
        both = (hi["Hit"] != 0) & (hb["Hit"] != 0)
        assert both.mean() > 0.2
        assert (np.abs(hi["T"][both] - hb["T"][both]) <= REL * np.maximum(hb["T"][both], 1e-6)).mean() > 0.995      # (baking rounds the moved corners to binary32: hits within rounding of an edge change triangle)
    del rng


def test_oracle_metamorphic_relations(oracle_mod, native_builder):
    _relation_rigid(lambda sc, r: oracle_mod.trace_rays(sc, r), native_builder)
    _relation_instances(lambda sc, r, tlas: oracle_mod.trace_rays(sc, r, use_tlas=tlas), native_builder)


def _gpu_trace(sc, rays, use_tlas=False):
    from idkengine_amd.pathtracer import PathTracer
    pt = PathTracer(8, 8); pt.UploadScene(sc); pt.UseTlas = 1 if use_tlas else 0
    out = pt.TraceRays(rays)
    pt.Dispose()
    return out


@pytest.mark.gpu
def test_gpu_closest_hit_equals_brute_force(soup100k):
    rays = random_rays(RAYS_N, 5, 4.0)
    check_against_brute_force(soup100k, rays, _gpu_trace(soup100k, rays))


@pytest.mark.gpu
def test_gpu_metamorphic_relations(native_builder):
    _relation_rigid(_gpu_trace, native_builder)
    _relation_instances(_gpu_trace, native_builder)


@pytest.mark.gpu
def test_gpu_path_tracer_primary_hits_equal_brute_force(native_builder):
    """The path tracer's own primary rays (FirstHit: jittered pixel rays through the persistent traversal kernel), checked the same way."""
    import configs
    from idkengine_amd.pathtracer import PathTracer
    sc = configs.lucy_scene(native_builder); w, h = 160, 200; cam = configs.lucy_camera(w, h)
    pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 1; pt.enable_primary_hit_capture(True); pt.Compute()
    t, tri, _ = pt.primary_hits(); r = pt.rays()
    o = r["Origin"].astype(np.float64)                       # RayDepth 1: the stored ray is still the primary ray of every pixel that missed
    tris, owner = world_triangles(sc)
    miss = tri == 0xFFFFFFFF
    # pixels that hit: origin moved to the hit point (FirstHit:129); go back along the stored direction is not possible (direction was
    # replaced by the bounce direction), so check the hit POINT instead: it must lie on the reported triangle's plane, inside it
    hp = o[~miss]; tr_ = tris[np.searchsorted(owner[:, 1], tri[~miss])]
    n = np.cross(tr_[:, 1] - tr_[:, 0], tr_[:, 2] - tr_[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
    dist = np.abs(((hp - tr_[:, 0]) * n).sum(1))
    assert (dist < 2e-3 + 1e-4 * np.abs(hp).max()).mean() > 0.999        # (the shaded origin is pushed 0.001 along the geometric normal, FirstHit:221)
    assert (~miss).mean() > 0.15
    pt.Dispose()
