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
    if "ref_cost_sum" in fx:     # the reference's own traversal-cost counter against the HIP path's P / T counters (the numerator of bench.py's roofline)
        pt = gpu_render(sc, cam, w, h, counters=True, capture=False, **ov); st = pt.stats(); pt.Dispose()
        glref_check.check_traversal_cost(fx, st["node_pair_visits"], st["triangle_tests"])
    assert all(s["state_is_the_compared_state"] and s["beyond_tol_in_sample"] == 0 for s in rep["stages"]), rep
