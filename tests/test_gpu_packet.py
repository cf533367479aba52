"""The packet walk of the primary launches (csrc/kernels_packet.hpp): frames whose primary rays are traced as wave-uniform packets — with the rays the walk does not vouch for
re-traced by the exact BVH2 kernel — are the oracle's bit for bit: image, every ray record, the alive queue, primary hit records (T, barycentrics, TriangleId).  The counting
build of k_trace2 (idkptEnableCounters) never takes this path, so everything here runs with the counters off, which is also what bench.py times."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal  # noqa: E402

pytestmark = pytest.mark.gpu


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


WALK_CASES = [
    ("cornell_mixed_d7", lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 192, 192, dict(RayDepth=7)),
    ("presplit_sort_d6", lambda b: S.presplit_scene(b), S.presplit_camera, 320, 180, dict(RayDepth=6, DoRaySorting=1)),        # a third of the rays end on marked triangles -> exact kernel
    ("soup100k_interior_d4", lambda b: S.soup_scene(100000, b), lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.2, 0.1, -1.0)), 640, 360, dict(RayDepth=4)),
    ("soup100k_outside_d2", lambda b: S.soup_scene(100000, b), lambda w, h: S.Camera(w, h), 636, 357, dict(RayDepth=2)),          # sky tiles, ragged frame size, divergent packets
    ("atrium60k_d5_sort", lambda b: S.atrium_scene(60000, b), S.atrium_camera, 256, 144, dict(RayDepth=5, DoRaySorting=1)),
    ("helmet_d5_aov", configs.helmet_scene, configs.helmet_camera, 320, 256, dict(RayDepth=5, OutputAOVs=1)),
    ("lucy_lens_d4", configs.lucy_scene, configs.lucy_camera, 240, 320, dict(RayDepth=4, FocalLength=9.0, LenseRadius=0.04)),
    ("axis_aligned_rays", lambda b: S.cornell_scene(b, "mixed"), lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 3.4), fovy_deg=1e-4), 64, 64, dict(RayDepth=3)),   # directions (0, 0, -1) up to rounding: 1/dir overflows -> not vouched for
]


@pytest.mark.parametrize("name,mk_scene,mk_cam,w,h,ov", WALK_CASES, ids=[m[0] for m in WALK_CASES])
def test_packet_walk_equals_oracle(name, mk_scene, mk_cam, w, h, ov, oracle_mod, native_builder):
    sc = mk_scene(native_builder); cam = mk_cam(w, h)
    o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
    aov = bool(ov.get("OutputAOVs"))
    stats = {}
    for label, env in (("forced", {"IDKPT_PACKET": "2"}), ("off", {"IDKPT_PACKET": "0"}), ("one_wave_per_cu", {"IDKPT_PACKET": "2", "IDKPT_PACKET_WAVES": "1"}),
                       ("short_runs", {"IDKPT_PACKET": "2", "IDKPT_GRAB_UNIT_LOG2": "6"}), ("long_runs", {"IDKPT_PACKET": "2", "IDKPT_GRAB_UNIT_LOG2": "14", "IDKPT_PACKET_WAVES": "3"})):
        pt = _env(env, lambda: gpu_render(sc, cam, w, h, counters=False, **ov))
        _env(env, lambda: assert_equal(pt, o, aov=aov, counters=False))
        stats[label] = pt.stats()
        pt.Dispose()
    o.close()
    assert stats["off"]["packet_packets"] == 0 and stats["off"]["packet_flagged_rays"] == 0
    st = stats["forced"]
    if name != "axis_aligned_rays":
        assert st["packet_packets"] > 0 and st["packet_node_steps"] > 0 and 0 < st["packet_live_lanes"] <= 64 * st["packet_node_steps"], st
    else:
        assert st["packet_flagged_rays"] > 0                                        # non-finite 1/dir: every entering ray goes to the exact kernel
    if name == "presplit_sort_d6":
        assert st["packet_flagged_rays"] > 1000                                     # the marked triangles cover most of the view
    if name == "soup100k_interior_d4":
        assert st["packet_flagged_rays"] < 0.01 * st["rays_traced"]


def test_packet_walk_batched_samples_pixel_major_and_lights(oracle_mod, native_builder):
    """Several samples in one launch (the pixel-major list the packet walk is meant for: a wave = 4 pixels x 16 samples), the automatic choice, and sphere lights as the rays'
    initial T (BVHIntersect.glsl:189-203)."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene(20000, native_builder, seed=9); w, h = 250, 130
    cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(-0.3, 0.2, -1.0))
    for frames, batch in ((16, 16), (20, 20), (9, 9), (5, 5)):
        o = oracle_render(oracle_mod, sc, cam, w, h, frames=frames, RayDepth=3)
        for mode in (2, 1):
            pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 3; pt.set_max_batch(batch); pt.set_option("packet", mode)
            for _ in range(frames):
                pt.Compute()
            assert (bits(pt.Result) == bits(o.image(0))).all() and pt.rays().tobytes() == o.rays().tobytes() and (pt.alive_queue() == o.alive_queue()).all(), (frames, mode)
            st = pt.stats()
            if mode == 2 or batch >= 8:
                assert st["packet_packets"] > 0, (frames, mode, st)                  # (automatic: the first batch of a view always probes)
            else:
                assert st["packet_packets"] == 0, (frames, mode, st)                 # (automatic: tile-major lists keep k_trace2)
            pt.Dispose()
        o.close()
    sc = S.cornell_scene(native_builder, "mixed")
    lights = np.zeros(2, T.GpuLight)
    lights[0]["Position"] = (0.3, 0.2, 0.4); lights[0]["Radius"] = 0.18; lights[0]["Color"] = (6.0, 5.0, 3.0); lights[0]["PointShadowIndex"] = -1
    lights[1]["Position"] = (-0.5, -0.4, 0.1); lights[1]["Radius"] = 0.1; lights[1]["Color"] = (1.0, 2.0, 8.0); lights[1]["PointShadowIndex"] = -1
    sc.lights = lights
    w = h = 128; cam = S.cornell_camera(w, h)
    for extra in (dict(), dict(DoRaySorting=1)):
        ov = dict(RayDepth=5, DoTraceLights=1, **extra)
        pt = _env({"IDKPT_PACKET": "2"}, lambda: gpu_render(sc, cam, w, h, counters=False, **ov)); o = oracle_render(oracle_mod, sc, cam, w, h, **ov)
        assert_equal(pt, o, counters=False)
        assert pt.stats()["packet_packets"] > 0
        pt.Dispose(); o.close()


def test_packet_choice_follows_the_kernels_own_counters(native_builder):
    """packet = 1: a view whose waves want the same nodes keeps the packet walk, a view whose pixels are wider than its triangles drops it after the probe — and the frames are the
    same bits either way (compared with packet = 0)."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene(200000, native_builder, seed=3); w, h = 640, 360
    views = {"interior": S.Camera(w, h, position=(0.0, 0.0, 0.0)), "outside": S.Camera(w, h, position=(0.0, 0.0, 60.0), fovy_deg=60.0)}
    for name, cam in views.items():
        imgs = {}
        for mode in (0, 1):
            pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 2; pt.set_max_batch(16); pt.set_option("packet", mode)
            per_batch = []
            for b in range(6):
                before = pt.stats()["packet_packets"]
                for _ in range(16):
                    pt.Compute()
                pt.synchronize()
                per_batch.append(pt.stats()["packet_packets"] - before)
            imgs[mode] = bits(pt.Result).copy()
            st = pt.stats()
            if mode == 1:
                live = st["packet_live_lanes"] / max(1, 64 * st["packet_node_steps"])
                if name == "interior":
                    assert live > 0.6 and per_batch[-1] > 0, (name, live, per_batch)
                else:
                    assert live < 0.6 and per_batch[0] > 0 and per_batch[-1] == 0, (name, live, per_batch)   # probed, then dropped
            pt.Dispose()
        assert (imgs[0] == imgs[1]).all(), name


def test_packet_marks_follow_refit(oracle_mod, oracle_builder, native_builder):
    """The per-triangle marks (k_mark_triangles) are re-derived after a refit moved boxes and positions."""
    sc = S.soup_scene(20000, native_builder, seed=12, refittable=True); w, h = 320, 180; cam = S.Camera(w, h, position=(0.0, 0.0, 4.0))
    pt = _env({"IDKPT_PACKET": "2"}, lambda: gpu_render(sc, cam, w, h, counters=False, RayDepth=3))
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=3)
    assert (bits(pt.Result) == bits(o.image())).all(); o.close()
    rng = np.random.default_rng(3)
    moved = (sc.vertex_positions + np.sin(sc.vertex_positions[:, ::-1] * 1.7).astype(np.float32) * np.float32(0.05) + rng.normal(0, 0.01, sc.vertex_positions.shape)).astype(np.float32)
    pt.UpdateBuffer(T.IDKPT_BUF_VERTEX_POSITIONS, moved); pt.RefitBlas(0)
    sc.blas_nodes = oracle_builder.refit(sc.blas_nodes, moved, sc.blas_triangles); sc.vertex_positions = moved
    pt.ResetAccumulation(); pt.Compute()
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=3)
    assert (bits(pt.Result) == bits(o.image())).all() and pt.rays().tobytes() == o.rays().tobytes()
    assert pt.stats()["packet_packets"] > 0
    pt.Dispose(); o.close()
