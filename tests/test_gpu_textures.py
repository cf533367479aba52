"""The texture table on the GPU (idkpt_texture: per-image wrap modes, magnification filter, 8-bit storage; idkptUpdateTexture): frames are the oracle's bit for bit — the oracle
whose texture unit tests/test_oracle_kats.py holds against the GL specification tap by tap and tests/test_glref.py against the reference's shaders on llvmpipe (the HIP path against
those fixtures: tests/test_gpu_glref.py, cases sampler_states_*).  Reference: Utils/ModelLoader.cs:1166-1197 (sampler state), Shaders/include/Surface.glsl:49-77 (the taps)."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import glref_cases  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal  # noqa: E402

pytestmark = pytest.mark.gpu
CAM = lambda w, h: S.Camera(w, h, position=(0.0, 0.1, 3.0), fovy_deg=48.0)  # noqa: E731


def _mixed_8bit_scene(b):
    """The sampler wall with every storage format: RGBA8 and SRGB8_A8 under both filters, RGBA32F on the rest."""
    sc = glref_cases._sampler_scene(b)
    rng = np.random.default_rng(77)
    tex = []
    for k, t in enumerate(sc.textures):
        if k % 3 == 0:
            tex.append(t)
        else:
            tex.append(T.TextureImage(rng.integers(0, 256, t.data.shape, dtype=np.uint8), t.wrap_s, t.wrap_t, t.mag_filter, srgb=(k % 3 == 2)))
    sc.textures = tex
    return sc


@pytest.mark.parametrize("name,fac", [("float_states", glref_cases._sampler_scene), ("rgba8_nearest", lambda b: glref_cases._sampler_scene(b, True)), ("mixed_8bit", _mixed_8bit_scene)])
def test_sampler_states_and_formats_equal_oracle(name, fac, oracle_mod, native_builder):
    sc = fac(native_builder)
    for (w, h, ov) in ((192, 112, dict(RayDepth=4, OutputAOVs=1)), (97, 61, dict(RayDepth=3, DoRaySorting=1))):
        cam = CAM(w, h)
        pt = gpu_render(sc, cam, w, h, **ov); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        assert_equal(pt, o, aov=bool(ov.get("OutputAOVs")))
        pt.Dispose(); o.close()


def test_update_texture_contents_size_format_and_state(oracle_mod, native_builder):
    """idkptUpdateTexture: a larger image under another format and sampler state replaces image 3, a smaller one image 11; queued samples are launched first."""
    from idkengine_amd.pathtracer import PathTracer
    sc = glref_cases._sampler_scene(native_builder); w, h = 160, 96; cam = CAM(w, h)
    rng = np.random.default_rng(5)
    pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 3; pt.set_max_batch(4)
    o = oracle_mod.OraclePathTracer(sc, w, h); o.set_camera(cam); o.settings.RayDepth = 3
    for _ in range(3):                                                   # three samples queued (not yet launched) when the update arrives: they see the OLD image
        pt.Compute(); o.render()
    big = T.TextureImage(rng.integers(0, 256, (16, 12, 4), dtype=np.uint8), T.IDKPT_WRAP_MIRRORED_REPEAT, T.IDKPT_WRAP_CLAMP_TO_EDGE, T.IDKPT_FILTER_LINEAR, srgb=True)
    small = T.TextureImage(rng.uniform(0, 1, (2, 3, 4)).astype(np.float32), T.IDKPT_WRAP_CLAMP_TO_EDGE, T.IDKPT_WRAP_REPEAT, T.IDKPT_FILTER_NEAREST)
    pt.UpdateTexture(3, big); pt.UpdateTexture(11, small)
    assert (bits(pt.Result) == bits(o.image(0))).all()                   # the three samples of the old table
    o.set_texture(3, big); o.set_texture(11, small)
    pt.ResetAccumulation(); o.reset_accumulation()
    for _ in range(2):
        pt.Compute(); o.render()
    assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes() and (pt.alive_queue() == o.alive_queue()).all()
    with pytest.raises(Exception):
        pt.UpdateTexture(len(sc.textures), small)                        # index out of range
    bad = T.TextureImage(small.data); bad.wrap_s = 7
    with pytest.raises(Exception):
        pt.UpdateTexture(0, bad)
    pt.Dispose(); o.close()


def test_multi_device_context_replicates_sampler_state(oracle_mod, native_builder):
    """idkptCreate(2) on one GPU: member 1 gets the texture table — texels AND state — by device-to-device copy (dev_CloneSceneFrom)."""
    from idkengine_amd.pathtracer import PathTracer
    sc = _mixed_8bit_scene(native_builder); w, h = 128, 80; cam = CAM(w, h)
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=2)
    pt = PathTracer(w, h, devices=[0, 0]); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2
    pt.Compute()
    assert (bits(pt.Result) == bits(o.image(0))).all()
    pt.Dispose(); o.close()
