"""Secondary benchmark: the reference's DEFAULT scene shape — several models, one BLAS each, no TLAS (Source/Application.cs:484 adds three models;
Bvh/BVH.cs:17-25 defaults to the instance loop of BVHIntersect.glsl:275-287; `UseTlas` walks BVHIntersect.glsl:205-272 instead) — against the
same triangles in ONE BLAS.  Per mode: Mray/s (32 samples in flight and one frame at a time), and the traversal kernel's per-visit rate
(64 B x node-pair visits + 48 B x triangle tests + 72 B x rays, exact counts of the counting build, / HIP-event time of the k_trace2 launches):
what the mode costs per unit of traversal work, independent of how much more work a 3-BLAS layout asks for.
usage: python tools/bench_multi.py [n_tris=1000000] [parts=3] [view=headline|interior]   (one JSON line per mode; also imported by bench.py)"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

W, H = 1920, 1080


def measure(pt, depth, B, steps):
    """-> dict: batched Mray/s, one-frame-at-a-time Mray/s, algorithmic bytes per ray and per-visit rate of the traversal launches"""
    pt.RayDepth = depth
    # exact visit counts of one displayed frame (B samples, counting build)
    pt.set_max_batch(1); pt.enable_counters(True); pt.reset_stats(); pt.ResetAccumulation()
    trav_total = 0
    for _ in range(B):
        pt.Compute(); pt.synchronize(); cs = pt.stats()
        trav_total += cs["alive_counts"][0] + sum(cs["alive_counts"][1:depth])      # rays of this sample that entered the traversal kernel
    pairs, tris = cs["node_pair_visits"], cs["triangle_tests"]
    pt.enable_counters(False)
    # batched
    pt.set_max_batch(B)
    for _ in range(B):
        pt.Compute()
    pt.synchronize(); pt.reset_stats(); pt.enable_timing(True)
    ts = []
    for _ in range(5):
        pt.ResetAccumulation(); t0 = time.perf_counter()
        for _ in range(steps):
            pt.Compute()
        pt.synchronize(); ts.append(time.perf_counter() - t0)
    st = pt.stats(); pt.enable_timing(False)
    rays = st["rays_traced"] / 5.0
    trav = trav_total / float(B)                                                # traversed rays per step (mean over the displayed frame's samples)
    dt = statistics.median(ts)
    frames_counted = 5.0 * steps
    alg_bytes = (64.0 * pairs + 48.0 * tris) / B * frames_counted + 72.0 * trav * frames_counted
    trace_s = st["trace_ms_total"] * 1e-3
    out = {"mray_s": round(rays / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 4), "rays_per_step": int(rays / steps), "traversed_rays_per_step": int(trav),
           "node_pair_visits_per_step": int(pairs / B), "triangle_tests_per_step": int(tris / B), "samples_in_flight": B,
           "trace_ms_per_step": round(trace_s * 1e3 / frames_counted, 4), "per_visit_rate_gbs": round(alg_bytes / trace_s / 1e9, 1) if trace_s > 0 else None,
           "alg_bytes_per_traversed_ray": round(alg_bytes / frames_counted / max(1, trav), 1)}
    # one frame at a time
    pt.set_max_batch(1)
    for _ in range(5):
        pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
    pt.reset_stats(); ts = []
    for _ in range(20):
        t0 = time.perf_counter(); pt.ResetAccumulation(); pt.Compute(); pt.synchronize(); ts.append(time.perf_counter() - t0)
    out["single_frame_mray_s"] = round(pt.stats()["rays_traced"] / 20.0 / statistics.median(ts) / 1e6, 1)
    return out


def run(S, PathTracer, builder, n_tris=1_000_000, parts=3, view="headline", depth=2, B=32, w=W, h=H, pt=None):
    cam = S.Camera(w, h) if view == "headline" else S.Camera(w, h, position=(0.0, 0.0, 0.0))
    own = pt is None
    pt = pt or PathTracer(w, h)
    res = {"workload": f"soup-{n_tris} split into {parts} BLASes with their own (rotated, shifted) transforms (idkengine_amd/scenes.py:soup_scene_multi), {w}x{h}, RayDepth {depth}, {view} view; "
                       f"per_visit_rate = (64 B x node-pair visits + 48 B x triangle tests + 72 B x rays) / time of the k_trace2 launches"}
    multi = S.soup_scene_multi(n_tris, builder, parts=parts, seed=1)
    for name, use_tlas in (("instance_loop", 0), ("tlas", 1)):
        pt.UploadScene(multi); pt.SetCamera(cam); pt.UseTlas = use_tlas
        res[name] = measure(pt, depth, B, 2 * B)
    pt.UseTlas = 0
    one = S.soup_scene(n_tris, builder, seed=1)
    pt.UploadScene(one); pt.SetCamera(cam)
    res["one_blas_same_triangle_count"] = measure(pt, depth, B, 2 * B)
    base = res["one_blas_same_triangle_count"]["per_visit_rate_gbs"]
    for name in ("instance_loop", "tlas"):
        res[name]["per_visit_rate_vs_one_blas"] = round(res[name]["per_visit_rate_gbs"] / base, 3) if base else None
    if own:
        pt.Dispose()
    return res


if __name__ == "__main__":
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from idkengine_amd.pathtracer import PathTracer
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    view = sys.argv[3] if len(sys.argv) > 3 else "headline"
    r = run(S, PathTracer, NativeBuilder(), n, parts, view)
    print(json.dumps(r), flush=True)
    for k in ("one_blas_same_triangle_count", "instance_loop", "tlas"):
        e = r[k]
        print(f"{k:30s} {e['mray_s']:8.1f} Mray/s batched  {e['single_frame_mray_s']:8.1f} one frame at a time  trace {e['trace_ms_per_step']:.4f} ms/step  {e['node_pair_visits_per_step'] / max(1, e['traversed_rays_per_step']):6.1f} pairs/ray  per-visit rate {e['per_visit_rate_gbs']} GB/s", file=sys.stderr, flush=True)
