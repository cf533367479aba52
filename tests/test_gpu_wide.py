"""The wide-node walk on the GPU (csrc/kernels_wide.hpp, csrc/wide_nodes.hpp): the structure the device derives is the host build's byte for byte, and frames
traced through it — with the rays it does not vouch for re-traced by the exact BVH2 kernel — are the oracle's bit for bit: image, every ray record, the alive queue,
primary hit records (T, barycentrics, TriangleId).  The counting build of k_trace2 (idkptEnableCounters) never takes this path, so everything here runs with the
counters off, which is also what bench.py times."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_build(tmp_path_factory):
    so = tmp_path_factory.mktemp("widecheck") / "libwidecheck.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(HERE, "c_driver", "wide_check.cpp"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.wide_host_build.restype = C.c_int
    L.wide_host_build.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]

    def build(nodes, tri_verts):
        nodes = np.ascontiguousarray(nodes); tri_verts = np.ascontiguousarray(tri_verts, np.float32)
        units = C.c_uint(0)
        n = L.wide_host_build(nodes.ctypes.data, len(nodes), tri_verts.ctypes.data, None, 0, None, 0, C.byref(units))
        wn = np.zeros(n * 16, np.uint32); wl = np.zeros(units.value * 4, np.uint32)
        L.wide_host_build(nodes.ctypes.data, len(nodes), tri_verts.ctypes.data, wn.ctypes.data, n, wl.ctypes.data, units.value, C.byref(units))
        return wn, wl
    return build


def _tri_verts(sc, d):
    t = sc.blas_triangles[d["TriangleOffset"]:d["TriangleOffset"] + d["TriangleCount"]]; p = sc.vertex_positions.reshape(-1, 3)
    tv = np.zeros((len(t), 3, 4), np.float32)
    tv[:, 0, :3] = p[t["X"]]; tv[:, 1, :3] = p[t["Y"]]; tv[:, 2, :3] = p[t["Z"]]
    return tv


STRUCT_SCENES = [
    ("cornell", lambda b: S.cornell_scene(b, "mixed")),
    ("cornell_instanced", lambda b: S.cornell_scene(b, "mixed", True)),          # several BLASes: one derivation each
    ("soup20k", lambda b: S.soup_scene(20000, b, seed=5)),
    ("presplit", lambda b: S.presplit_scene(b)),                                   # marked triangles (PreSplit fragments), triangles shared by leaf pairs
    ("atrium40k", lambda b: S.atrium_scene(40000, b)),
    ("soup_multi", lambda b: S.soup_scene_multi(30000, b, parts=3, seed=2)),
    ("lucy", configs.lucy_scene),
]


@pytest.mark.parametrize("name,mk", STRUCT_SCENES, ids=[s[0] for s in STRUCT_SCENES])
def test_device_derivation_equals_host_build(name, mk, native_builder, host_build):
    from idkengine_amd.pathtracer import PathTracer
    sc = mk(native_builder)
    pt = PathTracer(16, 16); pt.UploadScene(sc)
    nb = len(sc.blas_descs)
    counts = pt.DownloadBuffer(T.IDKPT_BUF_WIDE_COUNTS, np.uint32, 2 * nb).reshape(nb, 2)
    node_off = leaf_off = 0
    for b in range(nb):
        d = sc.blas_descs[b]
        pairs = int(d["NodeCount"]) // 2 + 1
        wn, wl = host_build(sc.blas_nodes[d["NodeOffset"]:d["NodeOffset"] + d["NodeCount"]], _tri_verts(sc, d))
        assert counts[b, 0] == len(wn) // 16 and counts[b, 1] == len(wl) // 4, (name, b, counts[b], len(wn) // 16, len(wl) // 4)
        got_n = pt.DownloadBuffer(T.IDKPT_BUF_WIDE_NODES, np.uint32, len(wn), offset_bytes=64 * node_off)
        got_l = pt.DownloadBuffer(T.IDKPT_BUF_WIDE_LEAVES, np.uint32, len(wl), offset_bytes=16 * leaf_off)
        assert got_n.tobytes() == wn.tobytes(), (name, b)
        assert got_l.tobytes() == wl.tobytes(), (name, b)
        node_off += pairs; leaf_off += 5 * (pairs + 1) + 3 * int(d["TriangleCount"]) + 4
    if name == "presplit":
        wn, wl = host_build(sc.blas_nodes[:sc.blas_descs[0]["NodeCount"]], _tri_verts(sc, sc.blas_descs[0]))
        assert (wl.reshape(-1, 4)[:, 3] == 1).sum() > 10      # the three scene-spanning triangles were split: their copies are marked
    pt.Dispose()


WALK_CASES = [
    ("cornell_mixed_d7", lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 192, 192, dict(RayDepth=7)),
    ("presplit_sort_d6", lambda b: S.presplit_scene(b), S.presplit_camera, 320, 180, dict(RayDepth=6, DoRaySorting=1)),        # a third of the rays end on marked triangles -> exact kernel
    ("soup100k_interior_d4", lambda b: S.soup_scene(100000, b), lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.2, 0.1, -1.0)), 640, 360, dict(RayDepth=4)),
    ("atrium60k_d5_sort", lambda b: S.atrium_scene(60000, b), S.atrium_camera, 256, 144, dict(RayDepth=5, DoRaySorting=1)),
    ("helmet_d5_aov", configs.helmet_scene, configs.helmet_camera, 320, 256, dict(RayDepth=5, OutputAOVs=1)),
    ("lucy_lens_d4", configs.lucy_scene, configs.lucy_camera, 240, 320, dict(RayDepth=4, FocalLength=9.0, LenseRadius=0.04)),
    ("axis_aligned_rays", lambda b: S.cornell_scene(b, "mixed"), lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 3.4), fovy_deg=1e-4), 64, 64, dict(RayDepth=3)),   # directions (0, 0, -1) up to rounding: 1/dir overflows -> not vouched for
]


def _env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name,mk_scene,mk_cam,w,h,ov", WALK_CASES, ids=[m[0] for m in WALK_CASES])
def test_wide_walk_equals_oracle(name, mk_scene, mk_cam, w, h, ov, oracle_mod, native_builder):
    sc = mk_scene(native_builder); cam = mk_cam(w, h)
    o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    aov = bool(ov.get("OutputAOVs"))
    stats = {}
    for label, env in (("default", {"IDKPT_WIDE": "1"}), ("off", {"IDKPT_WIDE": "0"}), ("short_stack", {"IDKPT_WIDE": "1", "IDKPT_WIDE_CAP": "4"}), ("one_wave_per_cu", {"IDKPT_WIDE": "1", "IDKPT_TRACE_WAVES": "1", "IDKPT_LEAF_MIN": "1"}),
                       ("odd_hand_out", {"IDKPT_WIDE": "1", "IDKPT_GRAB_UNIT_LOG2": "6", "IDKPT_GRAB_FIXED": "100", "IDKPT_LEAF_MIN": "64"})):
        pt = _env(env, lambda: gpu_render(sc, cam, w, h, counters=False, **ov))
        _env(env, lambda: assert_equal(pt, o, aov=aov, counters=False))
        stats[label] = pt.stats()
        pt.Dispose()
    o.close()
    assert stats["off"]["wide_flagged_rays"] == 0
    assert stats["short_stack"]["wide_flagged_rays"] >= stats["default"]["wide_flagged_rays"]
    if name == "presplit_sort_d6":
        assert stats["default"]["wide_flagged_rays"] > 1000                  # the marked triangles cover most of the view
    if name == "soup100k_interior_d4":
        assert stats["default"]["wide_flagged_rays"] < 0.01 * stats["default"]["rays_traced"]


def test_wide_walk_batched_samples_lights_and_counts(oracle_mod, native_builder):
    """Several samples in one launch, sphere lights as the rays' initial T (BVHIntersect.glsl:189-203), and the walk's own counters (developer option wide_count)."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.cornell_scene(native_builder, "mixed")
    lights = np.zeros(2, T.GpuLight)
    lights[0]["Position"] = (0.3, 0.2, 0.4); lights[0]["Radius"] = 0.18; lights[0]["Color"] = (6.0, 5.0, 3.0); lights[0]["PointShadowIndex"] = -1
    lights[1]["Position"] = (-0.5, -0.4, 0.1); lights[1]["Radius"] = 0.1; lights[1]["Color"] = (1.0, 2.0, 8.0); lights[1]["PointShadowIndex"] = -1
    sc.lights = lights
    w = h = 128; cam = S.cornell_camera(w, h)
    for extra in (dict(), dict(DoRaySorting=1)):
        ov = dict(RayDepth=5, DoTraceLights=1, **extra)
        pt = _env({"IDKPT_WIDE": "1"}, lambda: gpu_render(sc, cam, w, h, counters=False, **ov)); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        assert_equal(pt, o, counters=False)
        assert pt.stats()["wide_flagged_rays"] >= 0
        pt.Dispose(); o.close()
    sc = S.soup_scene(20000, native_builder, seed=9); w, h = 250, 130
    cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(-0.3, 0.2, -1.0))
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=5, RayDepth=3)
    pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 3; pt.set_max_batch(5); pt.set_option("wide", 1); pt.set_option("wide_count", 1)
    for _ in range(5):
        pt.Compute()
    assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes() and (pt.alive_queue() == o.alive_queue()).all()
    st = pt.stats(); ost = o.stats()
    # the walk's own work: fewer node fetches than the reference's pair visits (the point of the structure), the triangle tests of the same order
    assert 0 < st["wide_node_visits"] < 0.75 * ost["node_pair_visits"], (st, ost)
    assert st["wide_leaf_records"] > 0 and 0.5 * ost["triangle_tests"] < st["wide_triangle_tests"] < 1.5 * ost["triangle_tests"]
    pt.Dispose(); o.close()


def test_wide_nodes_follow_refit_and_node_patches(oracle_mod, oracle_builder, native_builder, host_build):
    """k_wide_fill after a refit (boxes and positions moved, topology kept), k_wide_topo after a node patch (idkptUpdateBuffer on the BLAS nodes)."""
    sc = S.soup_scene(20000, native_builder, seed=12, refittable=True); w, h = 320, 180; cam = S.Camera(w, h, position=(0.0, 0.0, 4.0))
    pt = _env({"IDKPT_WIDE": "1"}, lambda: gpu_render(sc, cam, w, h, counters=False, RayDepth=3))
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=3)
    assert (bits(pt.Result) == bits(o.image())).all(); o.close()
    rng = np.random.default_rng(3)
    moved = (sc.vertex_positions + np.sin(sc.vertex_positions[:, ::-1] * 1.7).astype(np.float32) * np.float32(0.05) + rng.normal(0, 0.01, sc.vertex_positions.shape)).astype(np.float32)
    pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, moved); pt.RefitBlas(0)
    want = oracle_builder.refit(sc.blas_nodes, moved, sc.blas_triangles)
    sc.vertex_positions = moved; sc.blas_nodes = want
    d = sc.blas_descs[0]
    wn, wl = host_build(want[:d["NodeCount"]], _tri_verts(sc, d))
    # (after the refit the topology is still the upload's — expansion by the ORIGINAL half areas — with boxes and leaf records re-derived: compared through the frame)
    pt.ResetAccumulation(); pt.Compute()
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=3)
    assert (bits(pt.Result) == bits(o.image())).all() and pt.rays().tobytes() == o.rays().tobytes()
    o.close()
    # a node patch: the same bytes written back through idkptUpdateBuffer invalidates and re-derives everything; the structure is then the host build of the refitted tree
    pt.UpdateBuffer(T.IDKPT_BUF_BLAS_NODES, want)
    cnt = pt.DownloadBuffer(T.IDKPT_BUF_WIDE_COUNTS, np.uint32, 2)
    assert cnt[0] == len(wn) // 16 and cnt[1] == len(wl) // 4
    assert pt.DownloadBuffer(T.IDKPT_BUF_WIDE_NODES, np.uint32, len(wn)).tobytes() == wn.tobytes()
    assert pt.DownloadBuffer(T.IDKPT_BUF_WIDE_LEAVES, np.uint32, len(wl)).tobytes() == wl.tobytes()
    pt.ResetAccumulation(); pt.Compute()
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=3)
    assert (bits(pt.Result) == bits(o.image())).all()
    pt.Dispose(); o.close()
