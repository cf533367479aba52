"""HIP path (through the C-ABI) vs the REFERENCE's own shaders on whole frames at BASELINE size (tests/golden/glref_full/: 1920x1080, the 1M-triangle
scenes bench.py times; see tests/test_glref_full.py and oracle/glref/make_full_vectors.py).  Stage by stage: the HIP path's state after FirstHit and
after every bounce is, bit for bit, the state that was compared with the reference's llvmpipe run ray by ray at generation (sha256 of 100 MB of ray
records + the alive queue), and on the fixture's sample of the rays it is compared with the reference's records directly, under the gate of
tests/glref_check.py (1e-4 pure relative, no outliers; the listed closest-hit exceptions: a few rays per million, named one by one in the fixture)."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import glref_cases  # noqa: E402
import glref_check  # noqa: E402
from gpu_helpers import gpu_render  # noqa: E402
from test_glref_full import FIXTURES, full_scene  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(glref_cases.FULL_CASES))
def test_hip_whole_frames_are_the_states_compared_with_the_reference(name, native_builder):
    skey, camf, w, h, ov = glref_cases.FULL_CASES[name]
    sc = full_scene(skey, native_builder); cam = camf(w, h)
    fx = np.load(os.path.join(FIXTURES, name + ".npz"))

    def state_at(d):
        pt = gpu_render(sc, cam, w, h, counters=False, capture=False, **dict(ov, RayDepth=d, SamplesPerPixel=1))
        r, q = pt.rays().copy(), pt.alive_queue().copy(); pt.Dispose()
        return r, q
    rep = glref_check.check_full_case(fx, state_at, strict=True, name=name)
    assert len(rep["stages"]) >= (1 if "debugcost" in name else 2)
    if "free_image_hash" in fx:  # the reference's whole free-running frame (FirstHit, NHit, FinalDraw; accumulated samples) against the Result image of the C-ABI
        pt = gpu_render(sc, cam, w, h, counters=False, capture=False, frames=int(fx["free_samples"]), **ov)
        glref_check.check_full_frame(fx, pt.Result, name=name); pt.Dispose()
    if "ref_cost_sum" in fx:     # the reference's own traversal-cost counter against the HIP path's P / T counters (the numerator of bench.py's roofline)
        pt = gpu_render(sc, cam, w, h, counters=True, capture=False, **ov); st = pt.stats(); pt.Dispose()
        glref_check.check_traversal_cost(fx, st["node_pair_visits"], st["triangle_tests"])
    assert all(s["state_is_the_compared_state"] and s["beyond_tol_in_sample"] == 0 for s in rep["stages"]), rep


def test_hip_refit_and_skinning_at_config5_size_match_reference_shaders(native_builder):
    """The device refit (level-synchronous) and the skinning kernel on the refittable 1M-triangle scene / 3 M vertices against the reference's BLASRefit and
    Skinning shaders run on llvmpipe over the whole scene: all 1.85 M nodes and all 3 M skinned positions bit-identical (sha256), re-compressed normals /
    tangents within one quantisation step on the sampled vertices (llvmpipe's inversesqrt)."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    from test_glref_full import _mfv, _sha
    fx = np.load(os.path.join(FIXTURES, "updates_1m.npz"))
    sc, moved, un, joints = _mfv().full_update_inputs(native_builder)
    pt = PathTracer(8, 8); pt.UploadScene(sc)
    pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, moved)
    pt.RefitBlas(0)
    nodes = pt.DownloadBuffer(T.IDKPT_BUF_BLAS_NODES, T.GpuBlasNode, len(sc.blas_nodes))
    assert np.array_equal(_sha(nodes), fx["refit_nodes_hash"]) and nodes[::509].tobytes() == fx["refit_nodes_sample"].tobytes()
    pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, sc.vertex_positions)
    pt.UploadUnskinnedVertices(un); pt.UpdateBuffer(T.IDKPT_BUF_JOINT_MATRICES, joints)
    pt.Skin(0, 0, 0, len(un)); pt.synchronize()
    pos = pt.DownloadBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, np.float32, 3 * len(moved)).reshape(-1, 3)
    assert np.array_equal(_sha(pos), fx["skin_positions_hash"]) and pos[::1021].tobytes() == fx["skin_positions_sample"].tobytes()
    verts = pt.DownloadBuffer(T.IDKPT_BUF_VERTICES, T.GpuVertex, len(sc.vertices))
    for field, key in (("Normal", "skin_normals_sample"), ("Tangent", "skin_tangents_sample")):
        got, ref = verts[field][::1021].astype(np.int64), fx[key].astype(np.int64)
        assert (got == ref).mean() >= 0.98, (field, (got == ref).mean())
        for shift, mask in ((0, 2047), (11, 2047), (22, 1023)):
            assert np.abs(((got >> shift) & mask) - ((ref >> shift) & mask)).max() <= 1, field
    pt.Dispose()
