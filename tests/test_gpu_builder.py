"""idkptBuildBlasCore: the SweepSAH core of the BLAS build on the GPU (idkengine_amd/csrc/bvh_gpu.hpp) against libidkbvh's CPU core —
the node array before compaction and the final x-sorted id order, byte for byte — and the finished BLAS (nodes, triangles, parent / leaf
indices, RequiredStackSize, SAH) against NativeBuilder AND, directly, against the oracle's restatement of the C# builder (oracle/ref_bvh_build.cpp:
BLAS.Build + PreSplitting.PreSplit + OptimizeStackSize + RemoveEmptySubtrees + GetUnindexedTriangles, Bvh/BLAS.cs:159-243, Bvh/PreSplitting.cs:26-160),
so that on the MI355X the device builder is held to the checker and not only to the product's own host builder."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402

pytestmark = pytest.mark.gpu


class _Capture:
    """Builder stand-in that records what scenes.assemble feeds build_blas, so the same inputs can go to both builders."""
    def __init__(self, inner):
        self.inner, self.calls = inner, []

    def build_blas(self, positions, tris, refittable):
        self.calls.append((np.array(positions, np.float32), np.array(tris), bool(refittable)))
        return self.inner.build_blas(positions, tris, refittable)

    def __getattr__(self, k):
        return getattr(self.inner, k)


CASES = dict(configs.BVH_CASES)
CASES["soup200k"] = lambda b: S.soup_scene(200000, b, seed=8)
CASES["atrium60k"] = lambda b: S.atrium_scene(60000, b)


@pytest.mark.parametrize("name", list(CASES))
def test_gpu_core_equals_cpu_core_and_finished_blas(name, native_builder):
    from idkengine_amd.bvh import GpuBuilder
    from idkengine_amd.pathtracer import PathTracer
    cap = _Capture(native_builder)
    CASES[name](cap)
    pt = PathTracer(8, 8)
    gb = GpuBuilder(pt)
    assert cap.calls
    for positions, tris, refittable in cap.calls:
        boxes, cpu_nodes, cpu_order = native_builder.core_arrays(positions, tris, refittable)
        gpu_nodes, gpu_order = gb.core_on_gpu(boxes)
        assert gpu_order.tobytes() == cpu_order.tobytes()
        assert gpu_nodes.tobytes() == cpu_nodes.tobytes()
        a = gb.build_blas(positions, tris, refittable); b = native_builder.build_blas(positions, tris, refittable)
        for k in ("nodes", "triangles", "parents", "leaves"):
            assert a[k].tobytes() == b[k].tobytes(), k
        assert a["required_stack_size"] == b["required_stack_size"] and a["sah"] == b["sah"] and a["fragments"] == b["fragments"]
    pt.Dispose()


def test_signed_zero_bounds_follow_the_sse_tie_rule(native_builder):
    """Boxes that differ only in the sign of a zero coordinate: minps/maxps return the second operand on equal values, so the stored bounds
    depend on the accumulation order — which must be the reference's."""
    from idkengine_amd.bvh import GpuBuilder
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    rng = np.random.default_rng(4)
    n = 3000
    p = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    p[::3, :, 0] = np.where(rng.random((len(p[::3]), 3)) < 0.5, np.float32(0.0), np.float32(-0.0))   # triangles lying in the plane x = +-0
    p[1::7, 0, 1] = np.float32(-0.0); p[2::5, 1, 2] = np.float32(0.0)
    positions = p.reshape(-1, 3)
    tris = np.zeros(n, T.GpuBlasTriangle); tris["X"] = np.arange(n) * 3; tris["Y"] = tris["X"] + 1; tris["Z"] = tris["X"] + 2
    pt = PathTracer(8, 8); gb = GpuBuilder(pt)
    for refittable in (True, False):
        boxes, cpu_nodes, cpu_order = native_builder.core_arrays(positions, tris, refittable)
        gpu_nodes, gpu_order = gb.core_on_gpu(boxes)
        assert gpu_order.tobytes() == cpu_order.tobytes() and gpu_nodes.tobytes() == cpu_nodes.tobytes()
    pt.Dispose()


@pytest.mark.parametrize("small", ["0", "5", "128"])
def test_level_synchronous_and_per_thread_paths_agree(small, native_builder, monkeypatch):
    """IDKPT_BVH_SMALL = 0: every node goes through the chunked level-synchronous kernels; 128: subtrees of up to 128 fragments are finished by
    one thread.  Both must give the CPU core's bytes; so must degenerate inputs (1, 2, 3 fragments; all fragments identical)."""
    from idkengine_amd.bvh import GpuBuilder
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    monkeypatch.setenv("IDKPT_BVH_SMALL", small)
    pt = PathTracer(8, 8); gb = GpuBuilder(pt)
    rng = np.random.default_rng(int(small) + 1)
    for n in (1, 2, 3, 7, 300, 5000):
        p = rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)
        if n == 7:
            p[:] = p[0]                                            # identical fragments: every cost ties
        positions = p.reshape(-1, 3)
        tris = np.zeros(n, T.GpuBlasTriangle); tris["X"] = np.arange(n) * 3; tris["Y"] = tris["X"] + 1; tris["Z"] = tris["X"] + 2
        for refittable in (True, False):
            boxes, cpu_nodes, cpu_order = native_builder.core_arrays(positions, tris, refittable)
            gpu_nodes, gpu_order = gb.core_on_gpu(boxes)
            assert gpu_order.tobytes() == cpu_order.tobytes() and gpu_nodes.tobytes() == cpu_nodes.tobytes(), (n, refittable)
            a = gb.build_blas(positions, tris, refittable); b = native_builder.build_blas(positions, tris, refittable)
            assert a["nodes"].tobytes() == b["nodes"].tobytes() and a["triangles"].tobytes() == b["triangles"].tobytes()
    pt.Dispose()


# ---- idkptBuildBlas: the whole build on the device (csrc/bvh_gpu_full.hpp)

FULL_CASES = dict(CASES)
FULL_CASES["soup1"] = lambda b: S.soup_scene(1, b)
FULL_CASES["soup1_refit"] = lambda b: S.soup_scene(1, b, refittable=True)            # single-leaf root duplicated into nodes 2, 3
FULL_CASES["soup2"] = lambda b: S.soup_scene(2, b)
FULL_CASES["soup3"] = lambda b: S.soup_scene(3, b, seed=5)
FULL_CASES["soup300k_refit"] = lambda b: S.soup_scene(300000, b, seed=12, refittable=True)   # stack optimisation (>= 16) without PreSplit
FULL_CASES["atrium300k"] = lambda b: S.atrium_scene(300000, b)


def _same_blas(a, b, what, sah_rel=0.0):
    assert a["fragments"] == b["fragments"] and a["required_stack_size"] == b["required_stack_size"], (what, a["fragments"], b["fragments"], a["required_stack_size"], b["required_stack_size"])
    for k in ("nodes", "triangles", "parents", "leaves"):
        assert np.asarray(a[k]).shape == np.asarray(b[k]).shape, (what, k, np.asarray(a[k]).shape, np.asarray(b[k]).shape)
        assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), (what, k)
    assert abs(a["sah"] - b["sah"]) <= sah_rel * abs(b["sah"]), (what, a["sah"], b["sah"])


@pytest.mark.parametrize("name", list(FULL_CASES))
def test_device_build_equals_native_build(name, native_builder, oracle_builder):
    """PreSplit (device cbrtf, split counts, grid splits), SweepSAH, OptimizeStackSize (parallel sums + the reference's decisions), RemoveEmptySubtrees
    as a stream compaction, both un-indexing variants, parent / leaf indices: every output array byte for byte, RequiredStackSize and the fragment
    count exactly, the SAH to rounding (a parallel binary64 sum)."""
    from idkengine_amd.bvh import DeviceBuilder
    from idkengine_amd.pathtracer import PathTracer
    cap = _Capture(native_builder)
    FULL_CASES[name](cap)
    pt = PathTracer(8, 8)
    db = DeviceBuilder(pt)
    assert cap.calls
    for positions, tris, refittable in cap.calls:
        a = db.build_blas(positions, tris, refittable); b = native_builder.build_blas(positions, tris, refittable)
        _same_blas(a, b, "device vs product host builder", 1e-12)
        o = oracle_builder.build_blas(positions, tris, refittable)                    # the checker itself, on the GPU box
        _same_blas(a, o, "device vs oracle restatement of BLAS.Build", 1e-12)
    pt.Dispose()


def test_device_cbrtf_equals_host_cbrtf(tmp_path):
    """The device's cbrtf (glibc 2.35's algorithm in binary64 steps) against the host's, bit for bit, on 6 M inputs: a stride through all bit patterns
    (every exponent, subnormals, negatives, zeros, infinities, NaNs) and the magnitudes PreSplit priorities have."""
    import ctypes as C
    import subprocess
    from idkengine_amd.pathtracer import PathTracer
    so = tmp_path / "libcbrtcheck.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(HERE, "c_driver", "cbrt_check.c"), "-o", str(so), "-lm"])
    L = C.CDLL(str(so)); L.host_cbrtf_array.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    rng = np.random.default_rng(2)
    x = np.concatenate([np.arange(0, 1 << 32, 1021, dtype=np.uint64).astype(np.uint32).view(np.float32),
                        (rng.uniform(0, 1, 1 << 21).astype(np.float32) ** 3 * np.float32(1e-3)), np.float32([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, 8.0, 27.0, 1e-45, -1e-45])])
    x = np.ascontiguousarray(x, np.float32)
    want = np.empty_like(x); L.host_cbrtf_array(x.ctypes.data, want.ctypes.data, len(x))
    pt = PathTracer(8, 8)
    got = np.empty_like(x)
    pt._check(pt._L.idkptCbrtProbe(pt._ctx, x.ctypes.data, got.ctypes.data, len(x)))
    pt.Dispose()
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all()
    assert (got.view(np.uint32)[~nan] == want.view(np.uint32)[~nan]).all()


def test_device_build_signed_zeros_and_degenerate_inputs(native_builder, oracle_builder):
    """Signed zeros (minps / maxps tie rule in the scene box, the triangle boxes and the clipped split boxes), identical triangles (every cost ties,
    the tree degenerates into a chain: deep enough for the stack optimisation), zero-area triangles (priority 0, cbrt(0))."""
    from idkengine_amd.bvh import DeviceBuilder
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    rng = np.random.default_rng(14)
    pt = PathTracer(8, 8); db = DeviceBuilder(pt)
    sets = []
    n = 3000
    p = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    p[::3, :, 0] = np.where(rng.random((len(p[::3]), 3)) < 0.5, np.float32(0.0), np.float32(-0.0))
    p[1::7, 0, 1] = np.float32(-0.0); p[2::5, 1, 2] = np.float32(0.0)
    sets.append(p)
    q = np.repeat(rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32), 40, axis=0)          # 40 identical triangles
    sets.append(q)
    z = rng.uniform(-1, 1, (500, 3, 3)).astype(np.float32); z[::4, 2] = z[::4, 1]           # every 4th triangle has zero area
    z[5] *= np.float32(40.0)                                                                 # and one is large: many splits
    sets.append(z)
    for tri_pos in sets:
        positions = tri_pos.reshape(-1, 3); m = len(tri_pos)
        tris = np.zeros(m, T.GpuBlasTriangle); tris["X"] = np.arange(m) * 3; tris["Y"] = tris["X"] + 1; tris["Z"] = tris["X"] + 2
        for refittable in (True, False):
            a = db.build_blas(positions, tris, refittable); b = native_builder.build_blas(positions, tris, refittable)
            assert a["fragments"] == b["fragments"] and a["required_stack_size"] == b["required_stack_size"]
            for k in ("nodes", "triangles", "parents", "leaves"):
                assert a[k].tobytes() == b[k].tobytes(), (k, m, refittable)
            _same_blas(a, oracle_builder.build_blas(positions, tris, refittable), ("device vs oracle", m, refittable), 1e-12)
    pt.Dispose()


@pytest.mark.parametrize("factor", [0.0, 0.3, 1.0, 2.5])
def test_device_build_presplit_factors(factor, native_builder):
    """PreSplitting.Settings.SplitFactor other than the default (the reference suggests 1.0 for Bistro, BLAS.cs:33-35): split counts, and with them every
    fragment, follow the factor on both builders."""
    from idkengine_amd.bvh import DeviceBuilder, NativeBuilder
    from idkengine_amd.pathtracer import PathTracer
    nb = NativeBuilder(presplit_factor=factor)
    pt = PathTracer(8, 8); db = DeviceBuilder(pt, presplit_factor=factor)
    for make in (configs.lucy_scene, configs.helmet_scene, lambda b: S.presplit_scene(b)):
        cap = _Capture(nb); make(cap)
        for positions, tris, refittable in cap.calls:
            a = db.build_blas(positions, tris, refittable); b = nb.build_blas(positions, tris, refittable)
            assert a["fragments"] == b["fragments"] and a["required_stack_size"] == b["required_stack_size"]
            for k in ("nodes", "triangles"):
                assert a[k].tobytes() == b[k].tobytes(), (k, factor)
    pt.Dispose()


def test_device_build_stack_optimisation_fallback_path(native_builder, monkeypatch):
    """OptimizeStackSize is decided on the device from parallel sums with an error bound; a decision inside the bound would fall back to the reference's own
    serial walk on a host copy of the tree.  That has never happened, so the path is forced here ("bvh_stackopt_host") on trees that need the optimisation
    (RequiredStackSize >= 16 before it): same bytes as the normal path and as the host builder."""
    from idkengine_amd.bvh import DeviceBuilder
    from idkengine_amd.pathtracer import PathTracer
    pt = PathTracer(8, 8); db = DeviceBuilder(pt)
    pt.set_option("bvh_stackopt_host", 1)
    for make in (lambda b: S.soup_scene(60000, b, seed=4), lambda b: S.soup_scene(200000, b, seed=8, refittable=True), configs.helmet_scene):
        cap = _Capture(native_builder); make(cap)
        for positions, tris, refittable in cap.calls:
            a = db.build_blas(positions, tris, refittable); b = native_builder.build_blas(positions, tris, refittable)
            assert a["required_stack_size"] == b["required_stack_size"]
            for k in ("nodes", "triangles", "parents", "leaves"):
                assert a[k].tobytes() == b[k].tobytes(), k
    pt.Dispose()


def test_device_build_refuses_what_the_host_builder_refuses(native_builder):
    """include/idkpt.h: non-finite vertex positions, vertex ids out of range and a PreSplit that asks for more than 2^27 fragments are refused with
    IDKPT_ERR_INVALID_ARGUMENT before anything is built (idkbvhBuildBlas refuses the same: tests/test_builder.py); the context keeps working."""
    from idkengine_amd.bvh import DeviceBuilder
    from idkengine_amd.pathtracer import PathTracer, IdkPtError
    from idkengine_amd import gputypes as T
    rng = np.random.default_rng(3)
    pos = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    tris = np.zeros(100, T.GpuBlasTriangle); tris["X"] = np.arange(100) * 3; tris["Y"] = tris["X"] + 1; tris["Z"] = tris["X"] + 2
    pt = PathTracer(8, 8); db = DeviceBuilder(pt)
    for bad in (np.nan, np.inf, -np.inf):
        p = pos.copy(); p[151, 1] = bad
        for refittable in (False, True):
            with pytest.raises(IdkPtError, match="not finite"):
                db.build_blas(p, tris, refittable)
    t2 = tris.copy(); t2["Z"][7] = 300
    with pytest.raises(IdkPtError, match="out of range"):
        db.build_blas(pos, t2, False)
    big = DeviceBuilder(pt, presplit_factor=1e9)
    with pytest.raises(IdkPtError, match="2\\^27"):
        big.build_blas(pos, tris, False)
    a = big.build_blas(pos, tris, True); b = native_builder.build_blas(pos, tris, True)              # refittable: no PreSplit, the factor is not looked at
    assert a["nodes"].tobytes() == b["nodes"].tobytes()
    a = db.build_blas(pos, tris, False); b = native_builder.build_blas(pos, tris, False)             # and the context still builds
    assert a["fragments"] == b["fragments"] and a["nodes"].tobytes() == b["nodes"].tobytes() and a["triangles"].tobytes() == b["triangles"].tobytes()
    pt.Dispose()
