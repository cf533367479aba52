"""The NON-counting build of the traversal kernel is the one bench.py times; most parity tests run the counting build (they also compare the
visit counters).  Here every single-BLAS case of the glref table, and a launch with more rays than the chip has lanes, render with the
counters OFF and must give the oracle's frame bit for bit: image, every ray record, the alive queue, AOVs."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
import glref_cases  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from gpu_helpers import bits, gpu_render, oracle_render  # noqa: E402

pytestmark = pytest.mark.gpu
SINGLE_BLAS = [n for n, c in glref_cases.GLREF_CASES.items() if not c[4].get("UseTlas") and not c[4].get("DoDebugBVHTraversal") and "inst" not in n and "multi" not in n]


def _same(pt, o, aov):
    assert (bits(pt.Result) == bits(o.image(0))).all()
    assert pt.rays().tobytes() == o.rays().tobytes()
    assert (pt.alive_queue() == o.alive_queue()).all()
    assert pt.stats()["rays_traced"] == o.stats()["rays_traced"]
    if aov:
        assert (bits(pt.AlbedoTexture) == bits(o.image(1))).all() and (bits(pt.NormalTexture) == bits(o.image(2))).all()


@pytest.mark.parametrize("name", SINGLE_BLAS)
def test_non_counting_traversal_matches_oracle(name, oracle_mod, native_builder):
    fac, camf, w, h, ov = glref_cases.GLREF_CASES[name]
    sc = fac(native_builder); cam = camf(w, h)
    pt = gpu_render(sc, cam, w, h, counters=False, **ov); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    _same(pt, o, bool(ov.get("OutputAOVs")))
    pt.Dispose(); o.close()


def test_non_counting_traversal_on_a_full_chip_launch(oracle_mod, native_builder):
    """More rays than the chip holds lanes for: the persistent waves refill several times; every pixel traverses."""
    sc = S.soup_scene(120000, native_builder, seed=31)
    w, h = 1280, 720
    cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.3, 0.2, -1.0), fovy_deg=95.0)
    for depth in (2, 5):
        pt = gpu_render(sc, cam, w, h, counters=False, RayDepth=depth); o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=depth)
        _same(pt, o, False)
        pt.Dispose(); o.close()
