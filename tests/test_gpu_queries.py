"""Adjacent consumers of the traversal core: batched ray queries (closest / any hit) and ray-traced shadows."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import configs  # noqa: E402,F401
from idkengine_amd import scenes as S  # noqa: E402,F401
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal, _queries  # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_tlas", [0, 1])
@pytest.mark.parametrize("any_hit", [False, True])
@pytest.mark.parametrize("lights", [False, True])
def test_ray_queries_match_oracle(native_builder, oracle_mod, use_tlas, any_hit, lights):
    """idkptTraceRays (TraceRay / TraceRayAny with maxDist + traceLights, BVHIntersect.glsl:183-411) == oracle, every field."""
    from idkengine_amd.pathtracer import PathTracer
    sc = S.cornell_scene(native_builder, "mixed", True)
    sc.lights = S.make_lights([((0.0, 0.55, 0.2), 0.12, (20.0, 20.0, 20.0)), ((-0.5, -0.2, 0.6), 0.08, (5.0, 2.0, 2.0))])
    rays = _queries(20000, 11 + use_tlas, 1.1)
    rays["MaxDist"][::3] = np.random.default_rng(5).uniform(0.05, 2.0, len(rays[::3])).astype(np.float32)   # a third of the rays are range-limited
    pt = PathTracer(8, 8); pt.UploadScene(sc); pt.UseTlas = use_tlas
    got = pt.TraceRays(rays, any_hit=any_hit, trace_lights=lights)
    want = oracle_mod.trace_rays(sc, rays, any_hit=any_hit, trace_lights=lights, use_tlas=bool(use_tlas))
    assert got.tobytes() == want.tobytes()
    assert (got["Hit"] != 0).any() and (got["Hit"] == 0).any()
    pt.Dispose()


def test_ray_queries_on_soup_match_oracle(native_builder, oracle_mod):
    from idkengine_amd.pathtracer import PathTracer
    sc = S.soup_scene_multi(30000, native_builder, parts=5, seed=21)
    rays = _queries(50000, 3, 12.0)
    pt = PathTracer(8, 8); pt.UploadScene(sc)
    for any_hit in (False, True):
        for tl in (0, 1):
            pt.UseTlas = tl
            got = pt.TraceRays(rays, any_hit=any_hit)
            assert got.tobytes() == oracle_mod.trace_rays(sc, rays, any_hit=any_hit, use_tlas=bool(tl)).tobytes()
    pt.Dispose()


@pytest.mark.parametrize("variant,use_tlas", [("mixed", 0), ("mixed", 1), ("blend", 0)])
def test_rt_shadows_match_oracle(native_builder, oracle_mod, variant, use_tlas):
    """idkptTraceShadows (Shaders/ShadowsRayTraced/compute.glsl) == oracle bit for bit on a stand-in G-buffer."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc = S.cornell_scene(native_builder, "mixed", True)
    if variant == "blend":   # make the short box alpha-blended and the tall box alpha-tested: exercises the continue-through-surface loop
        sc.materials["AlphaCutoff"][-2] = 2.0; sc.materials["BaseColorFactor"][-2] = (sc.materials["BaseColorFactor"][-2] & 0x00FFFFFF) | (0x60 << 24)
        sc.materials["AlphaCutoff"][-1] = 0.5; sc.materials["BaseColorFactor"][-1] = (sc.materials["BaseColorFactor"][-1] & 0x00FFFFFF) | (0x40 << 24)
    sc.lights = S.make_lights([((0.0, 0.55, 0.2), 0.12, (20.0, 20.0, 20.0)), ((-0.5, -0.2, 0.6), 0.08, (5.0, 2.0, 2.0))])
    w, h = 96, 80
    cam = S.cornell_camera(w, h)
    rays = S.primary_ray_queries(cam, w, h)
    pt = PathTracer(8, 8); pt.UploadScene(sc); pt.UseTlas = use_tlas
    hits = pt.TraceRays(rays)
    depth, normal = S.gbuffer_from_hits(sc, cam, w, h, rays, hits)
    for light, samples, noise in ((0, 1, 0), (0, 4, 8), (1, 3, 5)):
        p = T.ShadowParams.make(cam.inv_proj_view, w, h, light_index=light, samples=samples, noise_index=noise, jitter=(0.0005, -0.0003))
        keep = np.full((h, w), np.float32(-3.0))
        got = pt.TraceShadows(p, depth, normal, visibility=keep)
        want = oracle_mod.trace_shadows(sc, p, depth, normal, visibility=keep, use_tlas=bool(use_tlas))
        assert (bits(got) == bits(want)).all()
        assert (got == 1.0).any() and (got == 0.0).any()
    pt.Dispose()


def test_device_pointer_queries_equal_the_host_pointer_calls(native_builder, oracle_mod):
    """idkptTraceRaysDevice / idkptTraceShadowsDevice: the same kernels on buffers that already live on the GPU (the engine's G-buffer, ShadowsRayTraced/compute.glsl:9-13);
    asynchronous in the context's stream order, results bit for bit those of the host-pointer calls and of the oracle; a multi-device context refuses them."""
    import torch
    from idkengine_amd.pathtracer import PathTracer, IdkPtError
    from idkengine_amd import gputypes as T
    sc = S.cornell_scene(native_builder, "mixed", True)
    sc.lights = S.make_lights([((0.0, 0.55, 0.2), 0.12, (20.0, 20.0, 20.0))])
    w, h = 96, 80
    cam = S.cornell_camera(w, h)
    rays = S.primary_ray_queries(cam, w, h)
    pt = PathTracer(8, 8); pt.UploadScene(sc)
    dev = torch.device("cuda", 0)
    d_rays = torch.from_numpy(np.frombuffer(rays.tobytes(), np.uint8).copy()).to(dev)
    for any_hit in (False, True):
        d_hits = torch.zeros(len(rays) * T.RayHit.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        pt.TraceRaysDevice(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), any_hit=any_hit, trace_lights=True)
        pt.synchronize()
        got = np.frombuffer(d_hits.cpu().numpy().tobytes(), T.RayHit)
        assert got.tobytes() == pt.TraceRays(rays, any_hit=any_hit, trace_lights=True).tobytes() == oracle_mod.trace_rays(sc, rays, any_hit=any_hit, trace_lights=True).tobytes()
        # the host mirror's TraceRays given a device tensor takes the device entry point by itself and returns a device tensor
        dev_hits = pt.TraceRays(d_rays, any_hit=any_hit, trace_lights=True)
        assert dev_hits.is_cuda and dev_hits.cpu().numpy().tobytes() == got.tobytes()
    hits = pt.TraceRays(rays)
    depth, normal = S.gbuffer_from_hits(sc, cam, w, h, rays, hits)
    p = T.ShadowParams.make(cam.inv_proj_view, w, h, light_index=0, samples=3, noise_index=4, jitter=(0.0005, -0.0003))
    keep = np.full((h, w), np.float32(-3.0))
    d_depth = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).to(dev); d_normal = torch.from_numpy(np.ascontiguousarray(normal, np.float32)).to(dev); d_vis = torch.from_numpy(keep.copy()).to(dev)
    torch.cuda.synchronize()
    pt.Compute()                                                       # (queued rendering and device-pointer queries share the stream: order is the call order)
    pt.TraceShadowsDevice(p, d_depth.data_ptr(), d_normal.data_ptr(), d_vis.data_ptr())
    pt.synchronize()
    assert (bits(d_vis.cpu().numpy()) == bits(pt.TraceShadows(p, depth, normal, visibility=keep))).all()
    assert (bits(d_vis.cpu().numpy()) == bits(oracle_mod.trace_shadows(sc, p, depth, normal, visibility=keep))).all()
    with pytest.raises(IdkPtError):
        pt.TraceRaysDevice(0, 0, 5)                                    # null pointers
    pt.Dispose()
    grp = PathTracer(8, 8, devices=[0, 0]); grp.UploadScene(sc)
    with pytest.raises(IdkPtError):
        grp.TraceRaysDevice(d_rays.data_ptr(), d_rays.data_ptr(), 4)
    grp.Dispose()


@pytest.mark.parametrize("scene", ["one_blas", "instances", "tlas"])
def test_queries_on_the_frame_scheduler_equal_the_thread_per_ray_kernel(native_builder, oracle_mod, scene):
    """idkptTraceRays (closest and any hit) runs on k_trace2's persistent-wave scheduler (k_query_prepare -> k_trace2 -> k_query_finish); option query_scheduler 0 keeps the
    thread-per-ray kernel.  Both equal the oracle's TraceRay / TraceRayAny (BVHIntersect.glsl:107-411) in every field: ragged counts, rays limited to a range shorter than their first hit,
    rays that start inside / outside the root box, sphere lights in front of and behind the geometry, visit counters of the frame untouched."""
    from idkengine_amd.pathtracer import PathTracer
    if scene == "one_blas":
        sc = S.soup_scene(30000, native_builder, seed=12)
    else:
        sc = S.soup_scene_multi(30000, native_builder, parts=4, seed=13)
    sc.lights = S.make_lights([((0.0, 3.0, 2.0), 1.5, (20.0, 20.0, 20.0)), ((-14.0, -2.0, 6.0), 0.8, (5.0, 2.0, 2.0))])
    rays = _queries(40001, 17, 14.0)
    rng = np.random.default_rng(6)
    rays["MaxDist"][::3] = rng.uniform(0.05, 6.0, len(rays[::3])).astype(np.float32)
    rays["MaxDist"][5] = 0.0
    for n in (40001, 65, 64, 63, 1):
        want = None
        for sched in (1, 0):
            pt = PathTracer(16, 16); pt.set_option("query_scheduler", sched); pt.UploadScene(sc); pt.UseTlas = int(scene == "tlas")
            pt.enable_counters(True)
            for lights in (False, True):
                for any_hit in (False, True):                   # (TraceRayAny takes the same route: k_trace2's ANY instantiations, BVHIntersect.glsl:107-181, 299-411)
                    got = pt.TraceRays(rays[:n], any_hit=any_hit, trace_lights=lights)
                    ref = oracle_mod.trace_rays(sc, rays[:n], any_hit=any_hit, trace_lights=lights, use_tlas=scene == "tlas")
                    assert got.tobytes() == ref.tobytes()
                    if n > 1000:
                        assert (got["Hit"] != 0).any() and (got["Hit"] == 0).any()
            st = pt.stats(); assert st["node_pair_visits"] == 0 and st["triangle_tests"] == 0       # (queries do not count as frame traversal)
            pt.Dispose()
