"""State-machine check of the deferral logic (runs last: a failure here must not hide the other files under pytest -x)."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402,F401
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal  # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("IDKPT_RANDOM_API_SEEDS", "100"))))   # (3 s on the GPU; seed 77 found a stale deferred bounce after idkptSetMaxBatch in round 3)
def test_random_api_sequences_match_unbatched_replay(native_builder, seed):
    """State-machine check of the deferral logic: a random sequence of host calls (camera moves, Compute, ResetAccumulation, settings,
    SetMaxBatch, SetSize, scene swap, reads of the image and of the ray state in between) must leave the same image and the same ray state as the same
    logical sequence replayed on a context that never defers anything (max batch 1, last bounce shaded eagerly)."""
    from idkengine_amd.pathtracer import PathTracer
    rng = np.random.default_rng(100 + seed)
    scenes = [S.cornell_scene(native_builder, "mixed", True), S.soup_scene(4000, native_builder, seed=3, extent=2.5)]
    sizes = [(64, 40), (57, 33)]
    cams = lambda w, h: [S.cornell_camera(w, h), S.Camera(w, h, position=(0.3, 0.2, 5.0), fovy_deg=55.0), S.Camera(w, h, position=(-0.4, 0.1, 4.0), fovy_deg=70.0)]   # noqa: E731
    a = PathTracer(*sizes[0]); b = PathTracer(*sizes[0])
    if os.environ.get("IDKPT_RAPI_A_EAGER") == "1":
        a.set_option("defer_last", 0)                    # (debugging aid: both sides eager)
    b.set_option("defer_last", 0)                        # the replay also shades every last bounce eagerly; `a` defers it where it may (the soup scene does not emit)
    a.set_max_batch(int(rng.integers(2, 9)))
    # free choices of the implementation on the deferring side (their own generator: the call sequences of earlier rounds stay what they were): the kernels a launch
    # may be given — fused FirstHit + NHit, split, quad records, two parked leaves, pooled leaves — must never show in what a host reads
    orng = np.random.default_rng(9000 + seed)
    for name, values in (("fused", [1, 2, 2, 0]), ("split", [1, 2, 3, 0]), ("wide", [0, 0, 1]), ("wide_cap", [0, 0, 5]), ("leaf_pool", [-1, 7, 0, 1]), ("trace_waves", [0, 0, 1])):
        a.set_option(name, int(orng.choice(values)))
    size = sizes[0]
    for p in (a, b):
        p.UploadScene(scenes[0]); p.SetCamera(cams(*size)[0]); p.RayDepth = 3
    state_valid = False                                   # the ray state of the last sample is kept until the wavefront buffers are reallocated (idkptSetMaxBatch / idkptSetSize keep images only)
    for step in range(24):
        op = rng.choice(["cam", "compute", "compute", "compute", "reset", "depth", "sort", "batch", "size", "scene", "read", "spp", "state"])
        if op == "cam":
            k = int(rng.integers(0, 3))
            for p in (a, b):
                p.SetCamera(cams(*size)[k])
        elif op == "compute":
            for p in (a, b):
                p.Compute()
            state_valid = True
        elif op == "reset":
            for p in (a, b):
                p.ResetAccumulation()
        elif op == "depth":
            d = int(rng.integers(1, 6))
            for p in (a, b):
                p.RayDepth = d
        elif op == "sort":
            v = int(rng.integers(0, 2))
            for p in (a, b):
                p.DoRaySorting = v
        elif op == "spp":
            v = int(rng.integers(1, 4))
            for p in (a, b):
                p.SamplesPerPixel = v
        elif op == "batch":
            a.set_max_batch(int(rng.integers(1, 9))); state_valid = False
        elif op == "size":
            size = sizes[int(rng.integers(0, 2))]; state_valid = False
            for p in (a, b):
                p.SetSize(*size); p.SetCamera(cams(*size)[0])
        elif op == "scene":
            k = int(rng.integers(0, 2))
            for p in (a, b):
                p.UploadScene(scenes[k])
        elif op == "state":                               # ray state and alive queue of the most recent sample (completes a deferred last bounce on `a`)
            if state_valid:
                assert a.rays().tobytes() == b.rays().tobytes(), (seed, step)
                assert (a.alive_queue() == b.alive_queue()).all(), (seed, step)
        elif op == "read":
            assert (bits(a.Result) == bits(b.Result)).all(), (seed, step)
            assert a.AccumulatedSamples == b.AccumulatedSamples
    assert (bits(a.Result) == bits(b.Result)).all(), seed
    assert a.AccumulatedSamples == b.AccumulatedSamples
    for p in (a, b):                       # the ray state is only defined right after a sample
        p.Compute()
    assert a.rays().tobytes() == b.rays().tobytes() and (bits(a.Result) == bits(b.Result)).all()
    a.Dispose(); b.Dispose()
