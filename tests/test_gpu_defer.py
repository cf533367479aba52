"""The deferred last bounce (kernels_shade.hpp k_shade_last, idkpt.hip finish_deferred): without AOVs the frame needs only the radiance of a sample's last bounce —
the sky on the rays that miss and, in scenes that emit or with light hits, the radiance part of the hit shading; ray state, alive queue and counts of that bounce are
produced when somebody asks for them.  Results must not depend on it: images without ever asking, state when asked (before and after more frames, and across a
scene update), scenes where it must not apply, and the path for throughputs that are not finite."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from gpu_helpers import bits  # noqa: E402

pytestmark = pytest.mark.gpu


def _pt(sc, cam, w, h, defer, batch=1, **ov):
    from idkengine_amd.pathtracer import PathTracer
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov))
    pt.set_option("defer_last", defer)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.set_max_batch(batch)
    return pt


def _oracle(O, sc, cam, w, h, **ov):
    o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
    return o


@pytest.mark.parametrize("depth,batch,sort", [(2, 1, 0), (2, 3, 0), (4, 2, 1), (7, 1, 0)])
def test_images_never_need_the_deferred_part_and_state_follows_on_demand(depth, batch, sort, oracle_mod, native_builder):
    sc = S.soup_scene(20000, native_builder, seed=5); w, h = 160, 90
    cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.2, 0.1, -1.0))
    ov = dict(RayDepth=depth, DoRaySorting=sort)
    pt = _pt(sc, cam, w, h, 1, batch, **ov); o = _oracle(oracle_mod, sc, cam, w, h, **ov)
    for frame in range(2 * batch):
        pt.Compute(); o.render()
    assert (bits(pt.Result) == bits(o.image(0))).all()              # the image alone: nothing of the last bounce's continuation was computed
    for frame in range(batch):
        pt.Compute(); o.render()
    assert (bits(pt.Result) == bits(o.image(0))).all()
    assert pt.stats()["rays_traced"] == o.stats()["rays_traced"]
    assert pt.rays().tobytes() == o.rays().tobytes()                # now it is asked for
    assert (pt.alive_queue() == o.alive_queue()).all()
    assert (bits(pt.Result) == bits(o.image(0))).all()              # and asking did not touch the frame
    pt.Compute(); o.render()
    assert pt.rays().tobytes() == o.rays().tobytes() and (bits(pt.Result) == bits(o.image(0))).all()
    pt.Dispose(); o.close()


def test_state_of_a_deferred_frame_survives_a_scene_update(oracle_mod, native_builder):
    """The continuation is computed from the scene the frame was traced in: a scene update completes it first."""
    sc = S.soup_scene(5000, native_builder, seed=12, refittable=True); w, h = 96, 64
    cam = S.Camera(w, h, position=(0.0, 0.0, 0.0))
    pt = _pt(sc, cam, w, h, 1, RayDepth=3); o = _oracle(oracle_mod, sc, cam, w, h, RayDepth=3)
    pt.Compute(); o.render()
    want_rays, want_q = o.rays().copy(), o.alive_queue().copy()
    mats = sc.materials.copy(); mats["EmissiveFactor"][:] = (3.0, 2.0, 1.0)
    pt.UpdateBuffer(T.IDKPT_BUF_MATERIALS, mats)                   # every surface emits from now on
    assert pt.rays().tobytes() == want_rays.tobytes() and (pt.alive_queue() == want_q).all()
    sc2 = S.soup_scene(5000, native_builder, seed=12, refittable=True); sc2.materials = mats
    o2 = _oracle(oracle_mod, sc2, cam, w, h, RayDepth=3)
    pt.ResetAccumulation(); pt.Compute(); o2.render()               # and the next frame is shaded eagerly, with the emission
    assert (bits(pt.Result) == bits(o2.image(0))).all() and pt.rays().tobytes() == o2.rays().tobytes()
    pt.Dispose(); o.close(); o2.close()


@pytest.mark.parametrize("case", ["emissive", "aovs", "lights"])
def test_emissive_scenes_light_hits_and_aovs(case, oracle_mod, native_builder):
    """Emission and light hits: every hit of the last bounce runs the radiance part of the shading (k_shade_last<true>); AOVs: nothing is deferred."""
    w, h = 64, 64; cam = S.cornell_camera(w, h)
    ov = dict(RayDepth=3)
    if case == "emissive":
        sc = S.cornell_scene(native_builder, "mixed")
        assert (sc.materials["EmissiveFactor"] != 0).any() or (sc.meshes["EmissiveBias"] != 0).any()
    elif case == "aovs":
        sc = S.cornell_scene(native_builder, "diffuse"); ov["OutputAOVs"] = 1
    else:
        sc = S.cornell_scene(native_builder, "diffuse"); ov["DoTraceLights"] = 1
        sc.lights = S.make_lights([((0.0, 0.5, 0.2), 0.15, (20.0, 20.0, 20.0))])
    pt = _pt(sc, cam, w, h, 1, **ov); o = _oracle(oracle_mod, sc, cam, w, h, **ov)
    for _ in range(2):
        pt.Compute(); o.render()
    assert (bits(pt.Result) == bits(o.image(0))).all()
    if case == "aovs":
        assert (bits(pt.AlbedoTexture) == bits(o.image(1))).all() and (bits(pt.NormalTexture) == bits(o.image(2))).all()
    assert pt.rays().tobytes() == o.rays().tobytes()
    pt.Dispose(); o.close()


def test_throughput_that_is_not_finite_takes_the_full_shading_path(native_builder):
    """A texture with absurd values drives the throughput to infinity within a few bounces; 0 x inf is not 0, so those hits are shaded in full inside k_shade_last.
    NaN payloads are the arithmetic unit's business, so the comparison is between the two modes of the same library (images and state, bit for bit)."""
    rng = np.random.default_rng(2)
    m = S.cornell_meshes("diffuse")
    for part in ("tall", "short"):
        m[part].uvs = rng.uniform(0, 1, (len(m[part].positions), 2)).astype(np.float32); m[part].material["BaseColorTexture"] = 1
    for wall in m["walls"]:
        wall.uvs = rng.uniform(0, 1, (len(wall.positions), 2)).astype(np.float32); wall.material["BaseColorTexture"] = 1
    sc = S.assemble([{"meshes": m["walls"] + [m["short"], m["tall"]]}], native_builder, sky_color=(1.0, 1.0, 1.0))
    sc.textures = [np.full((2, 2, 4), 1e20, np.float32)]
    w, h = 64, 64; cam = S.cornell_camera(w, h)
    out = []
    for defer in (0, 1):
        pt = _pt(sc, cam, w, h, defer, RayDepth=5, DoRussianRoulette=0)
        for _ in range(2):
            pt.Compute()
        img = pt.Result.copy(); out.append((img, pt.rays().copy(), pt.alive_queue().copy())); pt.Dispose()
    assert not np.isfinite(out[0][1]["Throughput"]).all()           # the case is what it claims to be
    assert (bits(out[0][0]) == bits(out[1][0])).all() and out[0][1].tobytes() == out[1][1].tobytes() and (out[0][2] == out[1][2]).all()
