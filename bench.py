#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native wavefront path tracer (driver contract: see task statement).

Metric (BASELINE.json): Mray/s (primary + 1 bounce) at 1920x1080 on the synthetic 1M-triangle SweepSAH scene.
  step      = one pass of the hot path over the whole frame: FirstHit -> NHit -> FinalDraw at 1 spp, RayDepth 2
              (PathTracer.Compute, Source/Render/PathTracer.cs:214-271), scene and BVH already resident in HBM.
              Steps accumulate progressively like the reference's render loop (AccumulatedSamples 0,1,2,...); after every
              `samples_in_flight` steps the displayed frame is complete: its row shards are exchanged (N > 1) and the
              accumulation is reset (PathTracer.ResetAccumulation), so every sample that is traced ends up in an exchanged image.
  value     = (N + sum_j A_j) rays of all ranks / wall time (max over ranks), exact integer ray counts from the GPU queues.
  N GPUs    = image rows dealt round-robin to the ranks (idkengine_amd/dist.py); the only exchange is the RCCL all-gather of
              the finished row shards, inside the timed region.  Total work per step is fixed -> "strong" scaling.
  roofline  = traversal kernel (k_trace2): algorithmic bytes (64*P + 52*T + 104 per traversed ray) / HIP-event time of
              its launches during the timed region, against 8 TB/s HBM.
  cpu_baseline = the oracle's CPU port of the same path (all host cores, OpenMP) on a bounded sample of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
N_TRIS = 1_000_000
RAY_DEPTH = 2
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STATE_BYTES_PER_TRAVERSED_RAY = 104  # 48 B ray fetch + 4 B index + 20 B hit record + 32 B root node (DESIGN.md "Roofline")


def main():
    global W, H
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tris", type=int, default=N_TRIS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=RAY_DEPTH, help="RayDepth (headline = 2); other values are secondary-table runs")
    ap.add_argument("--sort", type=int, default=0, help="DoRaySorting (headline = 0)")
    ap.add_argument("--width", type=int, default=W, help="secondary-table runs only (headline = 1920)")
    ap.add_argument("--height", type=int, default=H, help="secondary-table runs only (headline = 1080)")
    ap.add_argument("--exact-deep-paths", action="store_true", help="N > 1 only: contiguous strips + per-bounce count exchange (gloo control group) so that RayDepth > 2 output equals the 1-GPU output bit for bit; default = interleaved rows (exact at RayDepth 2)")
    ap.add_argument("--interactive", type=int, default=0, metavar="F", help="secondary mode: every step is a NEW frame (own camera, own image, ResetAccumulation semantics) with F frames in flight through the frame ring; every finished frame is exchanged when N > 1")
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU-frame-equivalent the library may defer and trace together (idkptSetMaxBatch; results are bit-identical); multiplied by the GPU count because each rank only holds 1/N of every frame, capped at 256")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from idkengine_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.batch = max(1, min(256, args.batch * world))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
    # developer smoke test of the N > 1 flow on a single-GPU box: IDKPT_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and uses gloo
    # (RCCL refuses two ranks on one device); numbers from such a run are meaningless, only the control flow is exercised
    one_device = os.environ.get("IDKPT_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- scene: rank 0 builds (native SweepSAH + PreSplit builder), RCCL broadcast to the others
    t0 = time.time()
    scene = S.soup_scene(args.tris, NativeBuilder(), seed=1) if rank == 0 else None
    build_s = time.time() - t0
    if world > 1:
        scene = D.broadcast_scene(scene, src=0, device=device)
    W, H = args.width, args.height
    cam = S.Camera(W, H)

    control = dist.new_group(backend="gloo") if (world > 1 and args.exact_deep_paths) else None   # CPU-side group for the tiny count exchange
    r = D.GpuShardRenderer(W, H, world, rank, local_rank, exact_deep_paths=bool(control), control_group=control)
    r.upload_scene(scene); r.set_camera(cam)
    pt = r.pt
    depth = args.depth
    pt.RayDepth = depth; pt.SamplesPerPixel = 1; pt.DoRaySorting = args.sort
    frame = D.ShardedFrame(r, W, H) if world > 1 else None

    if args.interactive > 0:
        return interactive(args, r, frame, pt, world, rank, device, scene, build_s, dist, torch, S)
    B = args.batch
    step_no = [0]

    def finish_frame():
        if frame is not None:
            frame.gather()                           # the displayed frame's exchange: all-gather of the accumulated row shards

    def step():
        if step_no[0] % B == 0:
            pt.ResetAccumulation()                   # a new displayed frame starts (PathTracer.cs:334-342)
        pt.Compute()                                 # one 1-spp pass over the whole frame (deferred by the library)
        step_no[0] += 1
        if step_no[0] % B == 0:
            finish_frame()

    # ---- untimed counter pass over one displayed frame (B samples, one at a time): exact cumulative P (node-pair visits),
    #      T (triangle tests), traversed rays and traced rays after each sample index of this rank's rows
    pt.enable_counters(True); pt.reset_stats(); pt.ResetAccumulation()
    cum = [(0, 0, 0, 0)]
    for _ in range(B):
        pt.Compute(); pt.synchronize()
        cs = pt.stats()
        trav = cs["alive_counts"][0] + sum(cs["alive_counts"][1:depth])   # rays of this sample that entered the traversal kernel
        cum.append((cs["node_pair_visits"], cs["triangle_tests"], cum[-1][2] + trav, cs["rays_traced"]))
    pt.enable_counters(False)
    q, rem = divmod(args.steps, B)
    pairs, tri_tests, traversed, rays_expected = (q * cum[B][i] + cum[rem][i] for i in range(4))

    pt.set_max_batch(B)
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    if step_no[0] % B:
        finish_frame()
    pt.synchronize(); pt.reset_stats(); pt.enable_timing(True)
    step_no[0] = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if step_no[0] % B:
        finish_frame()                               # the last, partial displayed frame is exchanged too
    pt.synchronize()                                 # launches whatever is still deferred and waits for it
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    st = pt.stats()
    pt.enable_timing(False)

    rays_total = torch.tensor([float(st["rays_traced"])], dtype=torch.float64, device=device)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays_total = rays_total.item(); dt = tmax.item()
    assert st["rays_traced"] == rays_expected, (st["rays_traced"], rays_expected, args.steps)  # every step traced its complete sample (exact ray count)

    if rank == 0:
        value = rays_total / dt / 1e6
        # roofline of the traversal kernel (both instantiations of k_trace2: primary + bounce), this rank
        alg_bytes_total = 64.0 * pairs + 52.0 * tri_tests + STATE_BYTES_PER_TRAVERSED_RAY * traversed   # over all timed steps
        launches = max(1, st["trace_launches"])
        alg_bytes_launch = alg_bytes_total / launches
        avg_launch_s = st["trace_ms_total"] * 1e-3 / launches
        achieved = alg_bytes_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes/launch from the committed PMC summary of this same command
        headline = (args.tris, depth, args.sort, W, H, args.batch) == (N_TRIS, RAY_DEPTH, 0, 1920, 1080, min(256, 32 * world))
        if os.path.exists(prof) and headline:        # the PMC pass was taken on the headline command only
            try:
                traffic = json.load(open(prof)).get(f"n{world}", {}).get("traversal_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s (primary+1 bounce) at 1920x1080, 1M-tri scene" if headline else f"Mray/s (RayDepth {depth}) at {W}x{H}, {args.tris}-tri scene (secondary config)", "value": round(value, 2), "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"soup-{args.tris} (seeded random triangles, SweepSAH+PreSplit BVH, 1 BLAS), {W}x{H}, 1 spp, RayDepth {depth}, sort {'on' if args.sort else 'off'}, white sky",
                       "rays_per_step": int(rays_total / args.steps), "samples_in_flight": B, "displayed_frame": f"{B} accumulated samples, then exchange + ResetAccumulation", "sharding": ("contiguous strips + per-bounce alive-count exchange + all-gather" if args.exact_deep_paths else "rows round-robin over ranks + all-gather") if world > 1 else "none",
                       "bvh_build_s": round(build_s, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_trace2 (persistent while-while BVH traversal)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "alg_bytes_per_launch": int(alg_bytes_launch), "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches": int(launches),
                         "node_pair_visits_per_step": int(pairs / args.steps), "triangle_tests_per_step": int(tri_tests / args.steps), "traversed_rays_per_step": int(traversed / args.steps),
                         "hbm_copy_measured_gbs": hbm_copy_gbs(torch, device)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cam, depth)
        print(json.dumps(out), flush=True)
    r.pt.Dispose()
    if world > 1:
        dist.destroy_process_group()


def interactive(args, r, frame, pt, world, rank, device, scene, build_s, dist, torch, S):
    """Secondary mode (not the headline): F frames in flight, each with its own camera and result image (frame ring); after every
    F frames the F finished images are exchanged (one all-gather carrying every frame's row shard)."""
    import math
    F = max(1, min(64, args.interactive))
    pt.SetFrameRing(2 * F); pt.set_max_batch(F)
    cams = [S.Camera(W, H, position=(2.0 * math.sin(0.02 * k), 1.0 * math.sin(0.013 * k), 25.0)) for k in range(64)]
    state = {"k": 0, "first": None}

    def step():
        slot = pt.BeginFrame(); pt.SetCamera(cams[state["k"] % len(cams)]); pt.Compute()
        if state["k"] % F == 0:
            state["first"] = slot
        state["k"] += 1
        if state["k"] % F == 0 and frame is not None:
            frame.gather_frames(state["first"], F)                 # slots of one group are consecutive (ring size 2F)

    for _ in range(((args.warmup + F - 1) // F) * F):
        step()
    pt.synchronize(); pt.reset_stats()
    steps = ((args.steps + F - 1) // F) * F                         # whole groups
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    pt.synchronize(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    st = pt.stats()
    rays = torch.tensor([float(st["rays_traced"])], dtype=torch.float64, device=device); tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        dt = tmax.item()
        print(json.dumps({"metric": f"Mray/s, interactive: a new camera every frame, {F} frames in flight (secondary)", "value": round(rays.item() / dt / 1e6, 2), "unit": "Mray/s",
                          "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"soup-{args.tris}, {W}x{H}, 1 spp, RayDepth {args.depth}, every frame its own camera and image", "frames_in_flight": F,
                                     "frame_latency_ms": round(F * dt / steps * 1e3, 3), "exchange": "all-gather of every finished frame's row shards, once per F frames" if world > 1 else "none"}}), flush=True)
    r.pt.Dispose()
    if world > 1:
        dist.destroy_process_group()


def hbm_copy_gbs(torch, device):
    """Achievable HBM bandwidth on this box: device-to-device copy of 1 GiB, read + write bytes / time."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device); b = torch.empty_like(a)
    a.zero_(); b.copy_(a); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    return round(2.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def cpu_baseline(scene, cam, depth):
    """The oracle (CPU port of the reference path, GLSL semantics, OpenMP over all host cores) on a bounded sample of the
    same frame: every m-th row, m chosen so the run takes roughly 10-30 s."""
    from oracle import oracle as O   # allowed here: cpu_baseline leg only
    cores = os.cpu_count() or 1

    def run(mod):
        o = O.OraclePathTracer(scene, W, H, row_modulo=mod, row_remainder=0)
        o.set_camera(cam); o.settings.RayDepth = depth
        t0 = time.perf_counter(); o.render(); dt = time.perf_counter() - t0
        rays = o.stats()["rays_traced"]; rows = o.rows
        o.close()
        return rays, dt, rows
    rays, dt, rows = run(32)                       # probe: 34 rows
    full_est = dt * 32
    mod = 1 if full_est <= 30.0 else max(1, int(full_est / 20.0 + 0.999))
    # bounded sample of ~10-30 s: the row subset is repeated until at least 10 s of CPU work have been timed
    tot_rays, tot_dt, reps = 0, 0.0, 0
    while tot_dt < 10.0 and reps < 400:
        rays, dt, rows = run(mod)
        tot_rays += rays; tot_dt += dt; reps += 1
    # the reference's own CPU path (C# semantics: Gui.Test -> BVH.Intersect -> BLAS.Intersect, Render/Gui.cs:1484-1503): primary rays only
    p_rays, p_dt = 0, 0.0
    while p_dt < 3.0:
        t0 = time.perf_counter(); r = O.cpu_trace_primary(scene, cam, W, H, want_hits=False); p_dt += time.perf_counter() - t0
        p_rays += int(r["rays"])
    return {"value": round(tot_rays / tot_dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port",
            "primary_only_csharp_semantics": {"value": round(p_rays / p_dt / 1e6, 3), "unit": "Mray/s", "sample": f"{p_rays // (W * H)} full {W}x{H} frames of centre-of-pixel primary rays, closest hit only (no shading), {p_dt:.1f} s"},
            "sample": f"rows y%{mod}==0 of the same {W}x{H} frame ({rows} rows, {rays} rays, RayDepth {depth}) x {reps} repetitions = {tot_dt:.1f} s of CPU work; "
                      "C++/OpenMP restatement of the reference path incl. shading (the C# binary cannot run here: no .NET)"}


if __name__ == "__main__":
    main()
