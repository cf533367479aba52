"""k_trace_fused (csrc/kernels_trace_fused.hpp): at RayDepth 2 a lane that finishes a primary ray shades it and traces the bounce ray in the same persistent launch
(FirstHit/compute.glsl:44-98 and the traversal of NHit/compute.glsl:40-89 without the barrier between them, PathTracer.cs:214-271); the bounce's hit records are stored
per ray id and the deferred kernels look them up through the queue.  Everything a host can see must equal the two-launch schedule's and the oracle's bit for bit:
image, every ray record after the frame (the on-demand continuation of the last bounce included), alive queue, ray counts — with emission (every hit of the last
bounce may add radiance), sphere lights, textures, several samples per launch, one wave per CU, every shading-phase threshold, and across a scene update."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from gpu_helpers import bits, oracle_render  # noqa: E402

pytestmark = pytest.mark.gpu


def _render(sc, cam, w, h, opts, frames, batch, **ov):
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    pt = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov))
    for k, v in opts.items():
        pt.set_option(k, v)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.set_max_batch(batch)
    for _ in range(frames):
        pt.Compute()
    return pt


def _same(pt, o):
    assert (bits(pt.Result) == bits(o.image(0))).all()
    assert pt.rays().tobytes() == o.rays().tobytes()
    assert (pt.alive_queue() == o.alive_queue()).all()
    assert pt.stats()["rays_traced"] == o.stats()["rays_traced"]


def _case(case, native_builder):
    ov = dict(RayDepth=2)
    if case == "cornell":
        sc, w, h = S.cornell_scene(native_builder, variant="mixed"), 160, 120; cam = S.cornell_camera(w, h)
    elif case == "cornell_lights":
        sc, w, h = S.cornell_scene(native_builder, variant="mixed"), 128, 96; cam = S.cornell_camera(w, h)
        from idkengine_amd import gputypes as T
        lights = np.zeros(2, T.GpuLight); lights["Position"] = [(0.0, 0.6, 0.2), (0.3, -0.2, 1.5)]; lights["Radius"] = [0.15, 0.1]; lights["Color"] = [(6.0, 5.0, 4.0), (1.0, 2.0, 6.0)]
        sc.lights = lights; ov["DoTraceLights"] = 1
    elif case == "soup_inside":
        sc, w, h = S.soup_scene(40000, native_builder, seed=8), 200, 120; cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.2, 0.1, -1.0))
    elif case == "soup_outside":
        sc, w, h = S.soup_scene(40000, native_builder, seed=9), 200, 120; cam = S.Camera(w, h)
    elif case == "soup_lens":
        sc, w, h = S.soup_scene(20000, native_builder, seed=5), 131, 77; cam = S.Camera(w, h); ov.update(FocalLength=20.0, LenseRadius=0.4, DoRussianRoulette=0)
    else:
        sc, w, h = S.atrium_scene(30000, native_builder), 192, 108; cam = S.atrium_camera(w, h)
    return sc, cam, w, h, ov


CASES = ["cornell", "cornell_lights", "soup_inside", "soup_outside", "soup_lens", "atrium"]


@pytest.mark.parametrize("case", CASES)
def test_fused_launch_leaves_what_the_two_launches_leave(case, oracle_mod, native_builder):
    sc, cam, w, h, ov = _case(case, native_builder)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=3, **ov)
    for opts, batch in (({"fused": 2}, 1), ({"fused": 2, "trace_waves": 1, "leaf_min": 1, "fused_shade_min": 1}, 1), ({"fused": 2, "fused_shade_min": 64, "trace_waves": 2}, 1),
                        ({"fused": 2, "fused_shade_min": 8}, 3), ({"fused": 2, "grab_unit_log2": 6, "no_lean_primary": 1}, 3), ({"fused": 1}, 1), ({"fused": 0}, 1)):
        pt = _render(sc, cam, w, h, opts, 3, batch, **ov)
        _same(pt, o)
        pt.Dispose()
    o.close()


def test_fused_is_only_used_where_it_is_exact(oracle_mod, native_builder):
    """Deeper paths seed their RNG per queue slot (NHit/compute.glsl:54), AOVs and the eager last bounce need the whole NHit: those frames keep the wavefront schedule,
    whatever the option says; a change of RayDepth between frames switches schedules without a seam."""
    sc, w, h = S.soup_scene(20000, native_builder, seed=4), 160, 96; cam = S.Camera(w, h)
    for ov in (dict(RayDepth=3), dict(RayDepth=2, OutputAOVs=1), dict(RayDepth=1), dict(RayDepth=2, DoRaySorting=1)):
        o = oracle_render(oracle_mod, sc, cam, w, h, frames=2, **ov)
        pt = _render(sc, cam, w, h, {"fused": 2}, 2, 1, **ov)
        _same(pt, o)
        if ov.get("OutputAOVs"):
            assert (bits(pt.AlbedoTexture) == bits(o.image(1))).all() and (bits(pt.NormalTexture) == bits(o.image(2))).all()
        pt.Dispose(); o.close()
    pt = _render(sc, cam, w, h, {"fused": 2, "defer_last": 0}, 2, 1, RayDepth=2)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=2, RayDepth=2)
    _same(pt, o)
    # RayDepth 2 -> 4 -> 2 on one context (a new RayDepth restarts the accumulation, PathTracer.cs:16-25)
    pt.set_option("defer_last", 1)
    pt.RayDepth = 4; pt.Compute(); o.settings.RayDepth = 4; o.reset_accumulation(); o.render()
    _same(pt, o)
    pt.RayDepth = 2; pt.Compute(); pt.Compute(); o.settings.RayDepth = 2; o.reset_accumulation(); o.render(); o.render()
    _same(pt, o)
    pt.Dispose(); o.close()
