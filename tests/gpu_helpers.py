"""Shared helpers of the GPU parity tests (tests/test_gpu_*.py): render through the C-ABI (libidkpt.so) / through the CPU oracle
and compare bit patterns.  Bar: bit-exact for everything (ids, T, barycentrics, radiance, ray state, queues, visit counters): both
sides execute the same IEEE-754 binary32 operation sequence (DESIGN.md "Numerics"); the 1e-4 relative tolerance north_star allows
is therefore asserted as exact equality, with the looser bound kept as a named constant for reference."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import configs  # noqa: E402

NORTH_STAR_REL_TOL = 1e-4   # BASELINE.json; the tests demand 0


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def gpu_render(sc, cam, w, h, counters=True, capture=True, frames=1, **ov):
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    st = configs.apply_settings(T.Settings.default(), ov)
    pt = PathTracer(w, h, settings=st)
    pt.UploadScene(sc); pt.SetCamera(cam)
    pt.enable_counters(counters); pt.enable_primary_hit_capture(capture)
    for _ in range(frames):
        pt.Compute()
    return pt


def oracle_render(O, sc, cam, w, h, frames=1, **ov):
    o = O.OraclePathTracer(sc, w, h); o.set_camera(cam)
    configs.apply_settings(o.settings, ov)
    o.enable_counters(True)
    for _ in range(frames):
        o.render()
    return o


def assert_equal(pt, o, aov=False, counters=True):
    assert (bits(pt.Result) == bits(o.image(0))).all()
    gt, gtri, gb = pt.primary_hits(); ot, otri, ob = o.primary_hits()
    assert (gtri == otri).all() and (bits(gt) == bits(ot)).all() and (bits(gb) == bits(ob)).all()
    assert pt.rays().tobytes() == o.rays().tobytes()
    assert (pt.alive_queue() == o.alive_queue()).all()
    gs, os_ = pt.stats(), o.stats()
    assert gs["rays_traced"] == os_["rays_traced"]
    if counters:
        assert gs["node_pair_visits"] == os_["node_pair_visits"] and gs["triangle_tests"] == os_["triangle_tests"]
    if aov:
        assert (bits(pt.AlbedoTexture) == bits(o.image(1))).all() and (bits(pt.NormalTexture) == bits(o.image(2))).all()


def _queries(n, seed, extent, max_dist=3.4028235e+38):
    from idkengine_amd import gputypes as T
    rng = np.random.default_rng(seed)
    r = np.zeros(n, T.RayQuery)
    r["Origin"] = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r["Direction"] = d.astype(np.float32); r["MaxDist"] = max_dist
    return r


def read_device_image(pt, ptr, nbytes, shape):
    """Reads `nbytes` at device pointer `ptr` through the context's own stream (idkptGetStream) — what a consumer that is ordered behind
    the library's work does.  Uses torch as the HIP front end (one HIP runtime per process: conftest imports torch first)."""
    import ctypes as C
    import torch
    stream = C.c_void_p(); pt._check(pt._L.idkptGetStream(pt._ctx, C.byref(stream)))
    ext = torch.cuda.ExternalStream(stream.value)
    holder = type("DevArray", (), {"__cuda_array_interface__": {"shape": (nbytes // 4,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}})()
    with torch.cuda.stream(ext):
        t = torch.as_tensor(holder, device="cuda").to("cpu", non_blocking=False)
    return t.numpy().reshape(shape).copy()
