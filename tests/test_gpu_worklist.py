"""k_trace2 hands its work list out through eight interleaved counters (kernels_trace.hpp: slices of runs of 2^IDKPT_GRAB_UNIT_LOG2 entries,
optional reservations of IDKPT_GRAB_FIXED entries per atomic), parks leaves until `leafMin` lanes wait, and runs on a grid the host sizes.
Which lane traces which entry must not matter: every combination of those developer knobs — slices that are empty, that run out early, whose
last run is partial, reservations that straddle the end of a slice, a grid of one wave per CU, leaves tested one at a time or all together —
has to give the oracle's frame bit for bit (image, every ray record, alive queue, visit counters).  The knobs are read at launch time."""
import os
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
from idkengine_amd import scenes as S  # noqa: E402
from gpu_helpers import gpu_render, oracle_render, assert_equal  # noqa: E402

pytestmark = pytest.mark.gpu

KNOBS = [
    {},                                                                                  # the defaults (runs of 1024, no reservation)
    {"IDKPT_GRAB_UNIT_LOG2": "6"},                                                       # a refill straddles runs 512 entries apart
    {"IDKPT_GRAB_UNIT_LOG2": "16"},                                                      # fewer runs than slices: most slices are empty, waves move on at once
    {"IDKPT_GRAB_FIXED": "100"},                                                         # reservations that are no multiple of anything
    {"IDKPT_GRAB_FIXED": "1000", "IDKPT_GRAB_UNIT_LOG2": "8"},                           # a reservation longer than a run and than most slices
    {"IDKPT_TRACE_WAVES": "1", "IDKPT_LEAF_MIN": "1"},                                   # one wave per CU, leaves tested as soon as one lane parks
    {"IDKPT_TRACE_WAVES": "40", "IDKPT_LEAF_MIN": "64", "IDKPT_GRAB_FIXED": "64"},       # more workgroups than fit, leaves only when every lane is parked or stuck
]


def _with_env(env, fn):
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


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join(f"{a[6:].lower()}={b}" for a, b in k.items()) or "defaults")
def test_work_list_hand_out_is_order_free(knobs, oracle_mod, native_builder):
    cases = [(S.cornell_scene(native_builder, variant="mixed"), S.cornell_camera(97, 61), 97, 61, 4),                       # 5 917 rays: fewer than one run per slice
             (S.soup_scene(20000, native_builder, seed=5), S.Camera(320, 200, position=(0.0, 0.0, 0.0), view_dir=(0.2, 0.1, -1.0)), 320, 200, 3)]   # every pixel traverses, 64 000 rays
    for sc, cam, w, h, depth in cases:
        o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=depth)
        pt = _with_env(knobs, lambda: gpu_render(sc, cam, w, h, RayDepth=depth))          # (Compute inside: the knobs are read when the launches are issued)
        _with_env(knobs, lambda: assert_equal(pt, o))
        pt.Dispose(); o.close()


def test_batched_samples_with_odd_knobs(oracle_mod, native_builder):
    """The same with several samples in flight (one work list over all samples of the batch) on the non-counting build."""
    sc = S.soup_scene(20000, native_builder, seed=9); w, h = 250, 130
    cam = S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(-0.3, 0.2, -1.0))
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=5, RayDepth=3)
    for knobs in ({"IDKPT_GRAB_UNIT_LOG2": "7", "IDKPT_GRAB_FIXED": "200"}, {"IDKPT_GRAB_UNIT_LOG2": "12", "IDKPT_TRACE_WAVES": "2"}):
        def run():
            from idkengine_amd.pathtracer import PathTracer
            pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 3; pt.set_max_batch(5)
            for _ in range(5):
                pt.Compute()
            img = pt.Result; rays = pt.rays(); q = pt.alive_queue(); n = pt.stats()["rays_traced"]
            pt.Dispose()
            return img, rays, q, n
        img, rays, q, n = _with_env(knobs, run)
        from gpu_helpers import bits
        assert (bits(img) == bits(o.image(0))).all() and rays.tobytes() == o.rays().tobytes() and (q == o.alive_queue()).all() and n == o.stats()["rays_traced"]
    o.close()
