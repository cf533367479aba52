"""idkptBuildBlasCore: the SweepSAH core of the BLAS build on the GPU (idkengine_amd/csrc/bvh_gpu.hpp) against libidkbvh's CPU core —
the node array before compaction and the final x-sorted id order, byte for byte — and the finished BLAS (nodes, triangles, parent / leaf
indices, RequiredStackSize, SAH) against NativeBuilder, which tests/test_builder.py holds to the oracle's restatement of the C# builder."""
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
