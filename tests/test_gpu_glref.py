"""HIP path (through the C-ABI) vs the REFERENCE's own shader outputs (tests/golden/glref/, minted by oracle/glref/make_vectors.py
from /root/reference's GLSL on Mesa llvmpipe).  No oracle involved: every bounce is compared with what the reference's NHit dispatch
produced from the same input state, and the whole frame with the reference's free-running frame."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
import glref_cases  # noqa: E402
import glref_check  # noqa: E402
from gpu_helpers import gpu_render  # noqa: E402

pytestmark = pytest.mark.gpu
FIXTURES = os.path.join(HERE, "golden", "glref")


@pytest.mark.parametrize("name", list(glref_cases.GLREF_CASES))
def test_hip_path_matches_reference_shaders(name, native_builder):
    from idkengine_amd import gputypes as T
    fac, camf, w, h, ov = glref_cases.GLREF_CASES[name]
    sc = fac(native_builder); cam = camf(w, h)
    fx = np.load(os.path.join(FIXTURES, name + ".npz"))

    def state_at(d):
        pt = gpu_render(sc, cam, w, h, **dict(ov, RayDepth=d, SamplesPerPixel=1))
        r, q = pt.rays().copy(), pt.alive_queue().copy(); pt.Dispose()
        return r, q
    pt = gpu_render(sc, cam, w, h, **ov)
    aov = bool(configs.apply_settings(T.Settings.default(), ov).OutputAOVs)
    final = dict(image=pt.Result, counts=pt.stats()["alive_counts"], albedo=pt.AlbedoTexture if aov else None, normal=pt.NormalTexture if aov else None)
    rep = glref_check.check_case(fx, state_at, final, strict=True)
    pt.Dispose()
    assert all(s["flips"] == 0 and s["beyond_tol"] == 0 and s["queue_identical"] for s in rep["stages"]), rep
