#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native wavefront path tracer (driver contract: see task statement).

Metric (BASELINE.json): Mray/s (primary + 1 bounce) at 1920x1080 on the synthetic 1M-triangle SweepSAH scene (configs[2]).
  step      = one pass of the hot path over the whole frame: FirstHit -> NHit -> FinalDraw at 1 spp, RayDepth 2
              (PathTracer.Compute, Source/Render/PathTracer.cs:214-271), scene and BVH already resident in HBM.
              Steps accumulate progressively like the reference's render loop (AccumulatedSamples 0,1,2,...); after every
              `samples_in_flight` steps the displayed frame is complete: its row shards are exchanged (N > 1) and the
              accumulation is reset (PathTracer.ResetAccumulation), so every sample that is traced ends up in an exchanged image.
  value     = (N + sum_j A_j) rays of all ranks / wall time of the timed region (max over ranks), exact integer ray counts from the
              GPU queues.  The region (exactly --steps steps between barrier + synchronize) is repeated --repeats times; the MEDIAN
              repetition is reported (`ms_per_step` x `steps` = that repetition), all repetitions are listed under "repeat_ms".
  honesty   = the same line carries `traversed_mray_s` (only the rays that enter the BVH traversal kernel: the pre-cull answers the
              others from the root box), `single_frame` (one frame at a time with a synchronisation per frame, SURVEY 8(d)'s
              protocol: what a host sees that cannot keep 32 samples in flight) and `interior` (camera inside the soup: every pixel
              traverses — the stand-in for a Sponza-class view).
  N GPUs    = the FRAME is sharded (BASELINE.json's metric: one 1920x1080 frame on 1/2/4/8 GPUs; total work per step fixed -> "strong" scaling),
              whichever way the run is started: one process per GPU (torch.distributed launch, the driver's command; --shard rows, the default):
              the frame's rows are dealt over the ranks (bit-identical to one GPU), the displayed frame's row shards are all-gathered over
              RCCL inside the timed region; `python bench.py --gpus N` without a launcher drives ONE multi-device context
              (idkptCreate(deviceCount = N): rows / strips per device, scene replicated by peer copies, frame gathered on device 0); --spawn starts
              ranks instead (torch.distributed.run, 127.0.0.1).  Both print the same metric string and "scaling": "strong".
              --shard samples (explicit, never under the headline metric string): every rank renders WHOLE frames for the sample indices rank,
              rank + N, ... (idkptSetSampleSequence); per-GPU work is fixed -> "weak"; a secondary mode for hosts that want samples per second.
  roofline  = traversal kernel (k_trace2), fixed denominators from /opt/skills/guides/MI355X_MICROARCH.md.  achieved = algorithmic bytes (64 B per
              node-pair visit + 48 B per triangle test + 72 B per traversed ray, exact visit counts from the counting build) / HIP-event time of
              its launches in the timed region.  `frac` = achieved / 34.5 TB/s (aggregate L2, `bound: "l2"`): the 110 MB working set lives in L2 +
              Infinity Cache, so HBM does not bind and the algorithmic rate can exceed 8 TB/s — both HBM fractions (algorithmic, and from the
              FETCH_SIZE + WRITE_SIZE counters) are reported beside it under `hbm`.  `gather_measured` is a third, separately labelled figure: the
              rate at which THIS box fetches independent 64-B blocks (tools/ubench_lines), the access pattern of incoherent traversal.
              `traffic` and `pmc` are measured for THIS run: the timed region is re-executed under rocprofv3 --pmc (separate passes for
              FETCH_SIZE, WRITE_SIZE and the L2/L1 request counters; --kernel-trace only beside them) and the timed launches are averaged.
  cpu_baseline = the oracle's CPU port of the same path (all host cores, OpenMP), scene and threads kept warm, on a bounded sample.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC: without it RCCL's hipIpcGetMemHandle fails (multi-process runs)

W, H = 1920, 1080
N_TRIS = 1_000_000
RAY_DEPTH = 2
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # same guide, "L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s aggregate
NODE_PAIR_BYTES, TRI_BYTES, RAY_BYTES = 64, 48, 72     # node pair; triVerts entry; 48 B ray record + 4 B list entry + 20 B hit record


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same flags>`."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def view_camera(S, view, w, h):
    if view == "interior":
        return S.Camera(w, h, position=(0.0, 0.0, 0.0))       # inside the soup ([-10,10]^3): every primary ray enters the traversal
    return S.Camera(w, h)                                     # z = 25, looking at the soup, FovY 102


def main():
    global W, H
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--repeats", type=int, default=0, help="the timed region of --steps steps is repeated this often; the median repetition is reported.  0 (default): 7, "
                    "or as many as it takes to time >= 150 ms in total (at most 25): a region of a few milliseconds (--steps 20) is shorter than the GPU's clock ramp after the idle gap between regions")
    ap.add_argument("--tris", type=int, default=N_TRIS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-frame and interior-view measurements")
    ap.add_argument("--view", choices=["headline", "interior"], default="headline", help="headline = BASELINE.json configs[2]; interior = camera inside the soup (secondary)")
    ap.add_argument("--scene", choices=["soup", "atrium"], default="soup", help="soup = BASELINE.json configs[2] (headline); atrium = procedural Sponza-class hall of --tris triangles, camera inside (secondary)")
    ap.add_argument("--depth", type=int, default=RAY_DEPTH, help="RayDepth (headline = 2); other values are secondary-table runs")
    ap.add_argument("--sort", type=int, default=0, help="DoRaySorting (headline = 0)")
    ap.add_argument("--width", type=int, default=W, help="secondary-table runs only (headline = 1920)")
    ap.add_argument("--height", type=int, default=H, help="secondary-table runs only (headline = 1080)")
    ap.add_argument("--exact-deep-paths", action="store_true", help="N > 1 only: per-bounce exchange of the per-band alive counts (RCCL all-gather on the render stream, idkptSetBandExchangeDevice) so that RayDepth > 2 output equals the 1-GPU output bit for bit with the same interleaved deal; default = no exchange (exact at RayDepth 2)")
    ap.add_argument("--interactive", type=int, default=0, metavar="F", help="secondary mode: every step is a NEW frame (own camera, own image, ResetAccumulation semantics) with F frames in flight through the frame ring; every finished frame is exchanged when N > 1")
    ap.add_argument("--spawn", action="store_true", help="--gpus N without a launcher: start N processes (torch.distributed.run, one rank per GPU, RCCL) instead of the default ONE process driving ONE multi-device context (idkptCreate(deviceCount = N))")
    ap.add_argument("--shard", choices=["rows", "samples"], default="rows", help="one process per GPU (torch.distributed launch) only.  rows (default): the frame's rows are dealt over the ranks (bit-identical to one GPU, strong scaling: BASELINE.json's metric).  samples: every rank renders WHOLE frames for its own sample indices (idkptSetSampleSequence(rank, N)), one all-reduce per displayed frame, weak scaling — a secondary mode, never reported under the headline metric string")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic / roofline.pmc for this run (N = 1 only)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # internal: the re-execution of the timed region under rocprofv3
    ap.add_argument("--cpu-build", action="store_true", help="build the BLAS entirely on the host (libidkbvh) instead of running the SweepSAH core on the GPU; the result is the same bytes")
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU-frame-equivalent the library may defer and trace together (idkptSetMaxBatch; results are bit-identical); multiplied by the GPU count because each rank only holds 1/N of every frame, capped at 256")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and args.spawn:
        self_launch(args)                                # does not return
    group = args.gpus if (world == 1 and args.gpus > 1) else 1     # one process, one multi-device context (the C-ABI's own N-GPU mode)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from idkengine_amd import dist as D

    sample_parallel = world > 1 and args.shard == "samples" and not args.exact_deep_paths and args.interactive == 0
    args.batch = max(1, min(256, args.batch * (1 if sample_parallel else world * group)))   # row sharding: a rank holds 1/N of every frame, so N times the samples keep its launches as large
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
    scene, builder_kind, blas_build_ms = None, "cpu", None
    if rank == 0:
        if args.cpu_build:
            builder = NativeBuilder()
        else:   # the whole BLAS build on the GPU (idkptBuildBlas: PreSplit, SweepSAH, stack-size optimisation, compaction, un-indexing): same bytes as the CPU build
            from idkengine_amd.bvh import DeviceBuilder
            from idkengine_amd.pathtracer import PathTracer as _PT
            _bpt = _PT(8, 8); builder = DeviceBuilder(_bpt); builder_kind = "device (idkptBuildBlas)"
        scene = S.soup_scene(args.tris, builder, seed=1) if args.scene == "soup" else S.atrium_scene(args.tris, builder)
        blas_build_ms = round(builder.last_build_ms, 1)
        if not args.cpu_build:
            _bpt.Dispose()
    build_s = time.time() - t0
    if world > 1:
        scene = D.broadcast_scene(scene, src=0, device=device)
    W, H = args.width, args.height
    cam = view_camera(S, args.view, W, H) if args.scene == "soup" else S.atrium_camera(W, H)

    depth = args.depth
    if group > 1:
        # ONE context on `group` devices: the library replicates the scene over xGMI, deals the rows and gathers the frame itself
        from idkengine_amd.pathtracer import PathTracer
        ndev = torch.cuda.device_count()
        pt = PathTracer(W, H, devices=[d % ndev for d in range(group)])   # (more members than GPUs: members share GPUs — said so in the metric string and in config.n_gpu)
        pt.UploadScene(scene); pt.SetCamera(cam)
        r = type("Single", (), {"pt": pt})()

        class _GroupFrame:                                   # the displayed frame's exchange: rows of all devices gathered on the first one
            def gather(self):
                return pt.image_device_ptr(0)
        frame = _GroupFrame()
    else:
        if sample_parallel:
            r = D.SampleParallelRenderer(W, H, world, rank, 0 if one_device else local_rank)
            r.upload_scene(scene); r.set_camera(cam)
            pt = r.pt
            frame = D.SampleParallelFrame(r)
        else:
            r = D.GpuShardRenderer(W, H, world, rank, local_rank, exact_deep_paths=bool(world > 1 and args.exact_deep_paths))   # (the per-band count exchange is enqueued on the render stream: RCCL all-gather + prefix sum)
            r.upload_scene(scene); r.set_camera(cam)
            pt = r.pt
            frame = D.ShardedFrame(r, W, H) if world > 1 else None
    pt.RayDepth = depth; pt.SamplesPerPixel = 1; pt.DoRaySorting = args.sort

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
    cum = counter_pass(pt, B, depth)
    q, rem = divmod(args.steps, B)
    pairs, tri_tests, traversed, rays_expected = (q * cum[B][i] + cum[rem][i] for i in range(4))   # per repetition

    pt.set_max_batch(B)
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    if step_no[0] % B:
        finish_frame()
    pt.synchronize(); pt.reset_stats(); pt.enable_timing(True)
    repeat_s, rays_rank = [], 0
    want_reps = args.repeats if args.repeats > 0 else 7
    while len(repeat_s) < want_reps:
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
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        repeat_s.append(tmax.item())
        if args.repeats <= 0 and len(repeat_s) == 1:                      # (the same count on every rank: derived from the all-reduced time)
            want_reps = int(min(25, max(7, -(-0.15 // max(repeat_s[0], 1e-6)))))
    args.repeats = len(repeat_s)
    st = pt.stats()
    pt.enable_timing(False)
    reps = len(repeat_s)
    assert st["rays_traced"] == rays_expected * reps, (st["rays_traced"], rays_expected, reps)  # every step traced its complete sample (exact ray count)
    rays_total = torch.tensor([float(rays_expected), float(traversed)], dtype=torch.float64, device=device)   # per repetition, this rank
    if world > 1:
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
    rays_rep, traversed_rep = rays_total.tolist()
    dt = statistics.median(repeat_s)
    ranks_counted = None
    if world > 1:                                        # (a collective: every rank takes part, rank 0 reports)
        seen = torch.ones(1, dtype=torch.float64, device=device); dist.all_reduce(seen); ranks_counted = int(seen.item())
    selftest = None
    if world * group > 1 and not sample_parallel and (depth <= 2 or ((args.exact_deep_paths or group > 1) and not args.sort)) and not args.pmc_child:   # (beyond RayDepth 2 the shards are exact with the count exchange: --exact-deep-paths, or one multi-device context)
        # the N-GPU frame against ONE device, bit for bit (what tools/scale_selftest.py checks, here on the bench's own frame): two accumulated samples, the sharded
        # frame exchanged as in the timed region, rank 0 renders the same two samples on its own device alone and compares
        try:
            import hashlib
            pt.set_max_batch(2); pt.ResetAccumulation(); pt.Compute(); pt.Compute()
            if group > 1:
                got = pt.Result.tobytes()
            else:
                full = frame.gather(); torch.cuda.synchronize(); got = full.cpu().numpy().tobytes() if rank == 0 else None
            if rank == 0:
                from idkengine_amd.pathtracer import PathTracer as _One
                one = _One(W, H, device=local_rank); one.UploadScene(scene); one.SetCamera(cam); one.RayDepth = depth; one.DoRaySorting = args.sort; one.set_max_batch(2)
                one.Compute(); one.Compute(); want = one.Result.tobytes(); one.Dispose()
                selftest = {"frame": "2 accumulated samples of the bench frame, sharded + exchanged vs rank 0's device alone", "bits_equal": got == want, "sha16": hashlib.sha256(got).hexdigest()[:16]}
            pt.set_max_batch(B)
        except Exception as e:   # noqa: BLE001
            selftest = {"error": str(e)}

    if rank == 0:
        value = rays_rep / dt / 1e6
        shared = group > torch.cuda.device_count() or (world > 1 and one_device)      # members / ranks share GPUs: a control-flow check, not an N-GPU measurement
        headline = (not sample_parallel) and not shared and (args.tris, depth, args.sort, W, H, args.batch, args.view, args.scene) == (N_TRIS, RAY_DEPTH, 0, 1920, 1080, min(256, 32 * world * group), "headline", "soup")
        view_txt = "camera at z = 25 outside the soup (SURVEY 8d config 3)" if args.view == "headline" else "camera INSIDE the soup at the origin"
        out = {
            "metric": "Mray/s (primary+1 bounce) at 1920x1080, 1M-tri scene" if headline else f"Mray/s (RayDepth {depth}) at {W}x{H}, {args.tris}-tri {args.scene} scene, {args.view if args.scene == 'soup' else 'interior'} view (secondary config{', sample-parallel: whole frames per GPU' if sample_parallel else ''}{', ' + str(world * group) + ' members SHARING ' + str(torch.cuda.device_count()) + ' visible GPU(s): control flow only, not an N-GPU number' if shared else ''})", "value": round(value, 2), "unit": "Mray/s",
            "n_gpus": world * group, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if sample_parallel else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": reps, "repeat_ms": [round(x * 1e3, 3) for x in repeat_s], "statistic": "median repetition of the timed region",
            "traversed_mray_s": round(traversed_rep / dt / 1e6, 2),
            **transport_report(dist, world, group, pt, ranks_counted, one_device),
            "config": {"headline_notes": ([f"{100.0 * (1.0 - traversed_rep / max(1.0, rays_rep)):.0f} % of the counted rays never enter the BVH: primary rays whose 8x8 tile or whose own root-box test misses the scene "
                                          "(pre-classified sky; the same frame bit for bit) - the rate of the rays that do traverse is traversed_mray_s",
                                          "defer_last=1: the continuation of the last bounce (next direction, throughput, queue) is produced on demand, not every frame - eager_last_bounce is the rate without it",
                                          f"{B} samples in flight per launch - single_frame is the rate of a host that synchronises after every frame (SURVEY 8(d)'s protocol)"] if args.scene == "soup" and args.view == "headline" else None),
                       "workload": (f"soup-{args.tris} (seeded random triangles, SweepSAH+PreSplit BVH, 1 BLAS), {W}x{H}, 1 spp, RayDepth {depth}, sort {'on' if args.sort else 'off'}, white sky, {view_txt}" if args.scene == "soup" else
                                     f"atrium-{args.tris} (procedural two-storey colonnaded hall, connected surfaces, {len(scene.blas_triangles)} BLAS triangles, 1 BLAS), {W}x{H}, 1 spp, RayDepth {depth}, sort {'on' if args.sort else 'off'}, white sky, camera inside looking down the hall"),
                       "rays_per_step": int(rays_rep / args.steps), "traversed_rays_per_step": int(traversed_rep / args.steps),
                       "samples_in_flight": B, "displayed_frame": f"{B} accumulated samples, then exchange + ResetAccumulation",
                       "last_bounce": "every ray of the last bounce is traced and its radiance (sky on a miss; this scene has no emission, so hits add none) reaches the frame; the rest of that bounce's shading - new direction, throughput, Russian roulette, next queue: outputs the reference computes and nothing reads - is produced on demand (idkptDownloadRays / idkptDownloadAliveQueue / scene updates), bit-identical (DESIGN.md 4; option defer_last)", "sharding": ("one process, one multi-device context (idkptCreate(deviceCount = N)): scene replicated by peer copies, rows dealt round-robin in bands of 8 (beyond RayDepth 2 with the per-band alive-count exchange: exact at any depth), frame gathered on device 0" if group > 1 else (("sample-parallel: every rank renders whole frames for the sample indices rank, rank + N, ... (idkptSetSampleSequence); the displayed frame of N x samples_in_flight samples is the all-reduced mean of the ranks' accumulations; nothing is exchanged inside a frame" if sample_parallel else ("rows in bands of 8 round-robin over ranks + per-bounce exchange of the per-band alive counts enqueued on the render stream (idkptSetBandExchangeDevice: exact at any RayDepth, no host synchronisation) + all-gather" if args.exact_deep_paths else "rows round-robin over ranks + all-gather")) if world > 1 else "none")),
                       "bvh_build_s": round(build_s, 2), "blas_build_ms": blas_build_ms, "bvh_builder": builder_kind,
                       "blas_build_note": "blas_build_ms is the wall time of DeviceBuilder.build_blas as this script sees it: the device build (16 ms) + the download of nodes and triangles (23 ms together, profiles/r03_blas_build.txt) + numpy marshalling of 1 M triangles and the first call's allocations; untimed set-up, outside the metric",
                       "n_gpu": n_gpu_report(torch, dist, world, group, pt, st, B, depth, args, ranks_counted, selftest)},
            "roofline": roofline(st, pairs * reps / group, tri_tests * reps / group, traversed * reps / group, args, world * group, B if rem == 0 else (rem if q == 0 else None), torch, device),
        }
        if args.pmc_child:
            out = {"pmc_child": True, "trace_launches": int(st["trace_launches"]), "ms_per_step": out["ms_per_step"]}
        if world * group == 1 and not args.no_extras:
            # the same timed region with the last bounce shaded eagerly (option defer_last 0: new direction, throughput, Russian roulette and next queue of the bounce after
            # which nothing is traced, every frame, as the reference does): what `value` would be without producing those unread outputs on demand (config.last_bounce)
            pt.set_option("defer_last", 0)
            eager = []
            for _rep in range(3):
                step_no[0] = 0; torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                if step_no[0] % B:
                    finish_frame()
                pt.synchronize(); torch.cuda.synchronize(); eager.append(time.perf_counter() - t0)
            pt.set_option("defer_last", 1)
            out["eager_last_bounce"] = {"mray_s": round(rays_rep / statistics.median(eager) / 1e6, 2), "ms_per_step": round(statistics.median(eager) / args.steps * 1e3, 4), "repeats": 3,
                                        "what": "the timed region re-run with idkptSetDeveloperOption(defer_last, 0): bit-identical frames and ray state, the last bounce's continuation computed every frame"}
            out["single_frame"] = safe(single_frame, pt, depth)
            # the strict readings of the same workload beside `value` (config.headline_notes says what separates them): one frame at a time (SURVEY 8(d)'s protocol), the last bounce
            # shaded eagerly, and the rays that actually enter the BVH
            out["value_eager"] = out["eager_last_bounce"]["mray_s"]
            out["value_one_frame"] = out["single_frame"].get("mray_s") if isinstance(out["single_frame"], dict) else None
            out["value_traversed"] = out["traversed_mray_s"]
            if args.scene == "soup":
                # secondary blocks: each one on its own — a failure inside one is reported in its place and costs neither the metric line nor the other blocks
                out["interior"] = safe(interior_extras, S, pt, W, H, B)
                out["atrium"] = safe(atrium_extras, S, NativeBuilder, pt, W, H, B)
                out["multi_blas"] = safe(multi_blas_extras, S, NativeBuilder, pt, B)
                if isinstance(out["multi_blas"], dict):
                    out["multi_blas"]["atrium_per_mesh"] = safe(atrium_per_mesh_extras, S, NativeBuilder, pt, W, H, B)
                out["animated"] = safe(animated_extras, S, NativeBuilder, pt)
                safe(pt.UploadScene, scene)
                out["wide_nodes"] = safe(wide_extras, S, NativeBuilder, pt, scene, cam, W, H, B)
                safe(pt.UploadScene, scene)
                out["packet"] = safe(packet_extras, S, NativeBuilder, pt, scene, cam, W, H, B)
                safe(pt.UploadScene, scene)
                out["sort_on_vs_off"] = safe(sort_extras, S, NativeBuilder, pt, scene, cam, W, H, B)
                safe(pt.UploadScene, scene)
                out["queries"] = safe(query_extras, S, pt, scene, cam)
            pt.UseTlas = 0; pt.SetCamera(cam); pt.RayDepth = depth; pt.set_max_batch(B)
        if world * group == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = safe(cpu_baseline, S, scene, depth, args.view)
        print(json.dumps(out), flush=True)
    r.pt.Dispose()
    if world > 1:
        dist.destroy_process_group()


def safe(fn, *a, **k):
    """A secondary measurement must not cost the metric line: its failure is reported in its place."""
    try:
        return fn(*a, **k)
    except Exception as e:   # noqa: BLE001
        import traceback
        return {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc().strip().splitlines()[-3:]}


def transport_report(dist, world, group, pt, ranks_counted, one_device):
    """Top-level answer to "did RCCL carry the N > 1 run, and how many ranks did it see": torch.distributed ranks -> the backend and the count an all-reduce of ones
    returned; one multi-device context -> idkptGetTransportInfo (RCCL communicator formed inside the library, or its peer copies and why)."""
    if world > 1:
        backend = dist.get_backend()
        return {"transport": "rccl (torch.distributed, backend nccl)" if backend == "nccl" else f"{backend} (control-flow check: ranks share one GPU)" if one_device else backend,
                "rccl_ranks_seen": ranks_counted if backend == "nccl" else None, "ranks": world}
    if group > 1:
        try:
            info = pt.transport_info()
            return {"transport": info["transport"] + (" (inside idkptCreate(N): ncclCommInitAll, ncclBroadcast of the scene, grouped ncclSend/ncclRecv gather)" if info["transport"] == "rccl" else f" ({info['detail']})"),
                    "rccl_ranks_seen": info["rccl_ranks"] if info["transport"] == "rccl" else None, "ranks": group}
        except Exception as e:   # noqa: BLE001
            return {"transport": f"unknown ({e})", "rccl_ranks_seen": None, "ranks": group}
    return {"transport": "none (one GPU)", "rccl_ranks_seen": None, "ranks": 1}


def n_gpu_report(torch, dist, world, group, pt=None, st=None, B=None, depth=None, args=None, ranks_counted=None, selftest=None):
    """What the N > 1 run actually ran on (tools/scale_selftest.py checks the same things and the bits): devices, peer access, ranks RCCL saw, whether members
    share GPUs, and how large this rank's traversal launches were — a poor scaling point then reads as "launches at the latency floor of their longest rays"
    (~1 M rays per launch) or as "exchange", not as a riddle."""
    if world * group == 1:
        return None
    try:
        ndev = torch.cuda.device_count()
        ids = sorted({d % ndev for d in range(group)}) if group > 1 else list(range(min(ndev, world)))
        rep = {"visible_devices": ndev, "members_or_ranks": world * group, "devices_used": ids, "members_share_gpus": bool(group > ndev or (world > 1 and os.environ.get("IDKPT_BENCH_ONE_DEVICE") == "1")),
               "peer_access": [[1 if i == j else int(torch.cuda.can_device_access_peer(i, j)) for j in ids] for i in ids]}
        if world > 1:
            rep["ranks"] = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "ranks_counted_by_all_reduce": ranks_counted}   # (every rank added 1 in an all-reduce: the collective itself says how many ranks RCCL connected)
        if st is not None and B:
            ac = st["alive_counts"]; per_batch = min(B, args.steps) if args is not None else B
            rep["rank0_launches"] = {"samples_per_launch": per_batch, "primary_rays_per_launch": int(ac[0]), "bounce_rays_per_launch": [int(a * per_batch) for a in ac[1:depth]],   # ([0]: the whole batch's active list; [j]: the last sample's alive count x samples)
                                     "avg_trace_launch_us": round(st["trace_ms_total"] * 1e3 / max(1, st["trace_launches"]), 1), "trace_launches": int(st["trace_launches"]),
                                     "row_deal": "bands of 8 rows, (y // 8) % N" if world > 1 else "one multi-device context (idkptSetGroupSharding AUTO: bands of 8 rows)"}
        rep["selftest"] = selftest
        return rep
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)}


def counter_pass(pt, B, depth):
    """B samples one at a time on the counting build: cumulative (node-pair visits, triangle tests, traversed rays, traced rays)."""
    pt.set_max_batch(1)
    pt.enable_counters(True); pt.reset_stats(); pt.ResetAccumulation()
    cum = [(0, 0, 0, 0)]
    for _ in range(B):
        pt.Compute(); pt.synchronize()
        cs = pt.stats()
        trav = cs["alive_counts"][0] + sum(cs["alive_counts"][1:depth])   # rays of this sample that entered the traversal kernel
        cum.append((cs["node_pair_visits"], cs["triangle_tests"], cum[-1][2] + trav, cs["rays_traced"]))
    pt.enable_counters(False)
    return cum


def file_sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def gather_ceiling(set_log2_blocks=16):
    """The chip's rate of independent 64-B block fetches (tools/ubench_lines mode 0: every lane loads 4 x 16 B of its own random 64-B
    block = the node-pair fetch; 2^16 blocks = 4 MB set = L2 hits, 32 waves/CU), measured now, on this box.  GB/s of 64-B blocks."""
    exe = os.path.join(ROOT, "tools", "ubench_lines.bin")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "ubench_lines.hip"), "-o", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = subprocess.run([exe, str(set_log2_blocks), "0", "32"], capture_output=True, text=True, timeout=120).stdout
        for line in txt.splitlines():
            if line.startswith("RESULT"):
                kv = dict(p.split("=") for p in line.split()[1:])
                return float(kv["gbs"])
    except Exception:
        pass
    return None


def pmc_passes(args, launches):
    """Re-executes this run's timed region (same scene, view, depth, steps, warm-up; one repetition, no extras) under rocprofv3 --pmc, one
    pass per counter set (the guide's HBM/rocprofv3 section: FETCH_SIZE and WRITE_SIZE do not fit one pass; --kernel-trace only beside --pmc),
    and averages the counters over the timed launches of the non-counting k_trace2 instantiations (the last `launches` dispatches of each
    child: warm-up comes first, nothing follows).  Returns None when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = os.environ.get("IDKPT_BENCH_ROCPROFV3") or shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)   # (the override: the test that takes the tool away)
    if exe is None or not os.path.exists(exe):
        return None
    sets = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
            "sq": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES"]}   # issue side: what a cache-resident, coherent launch is limited by
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--no-pmc", "--no-extras", "--no-cpu-baseline", "--repeats", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
             "--tris", str(args.tris), "--view", args.view, "--scene", args.scene, "--depth", str(args.depth), "--sort", str(args.sort), "--width", str(args.width), "--height", str(args.height),
             "--batch", str(args.batch)] + (["--cpu-build"] if args.cpu_build else [])
    got = {}
    tmp = tempfile.mkdtemp(prefix="idkpt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for name, ctrs in sets.items():
            d = os.path.join(tmp, name)
            r = subprocess.run([exe, "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "-o", "b", "--"] + child, capture_output=True, text=True, timeout=180, env=env, cwd="/tmp")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if name == "sq":
                    continue                                   # (the issue-side pass is an extra: the traffic figures stand without it)
                return None
            n_child = None
            for line in r.stdout.splitlines():
                if line.startswith("{") and "pmc_child" in line:
                    n_child = json.loads(line)["trace_launches"]
            if n_child != launches:
                return None                                    # not the same launch schedule: do not report someone else's numbers
            per = {}
            for row in csv.DictReader(open(files[0])):
                k = row["Kernel_Name"]
                if "k_trace2<" not in k:
                    continue
                targs = k.split("k_trace2<")[1].split(">")[0].replace(" ", "").split(",")
                if targs[1] != "false":
                    continue                                   # the counting instantiation (untimed counter pass)
                per.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for c, v in per.items():
                v.sort()
                timed = [x for _, x in v[-launches:]]
                if len(timed) != launches:
                    return None
                got[c] = sum(timed) / launches
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got


def roofline(st, pairs, tri_tests, traversed, args, world, samples_per_launch, torch, device):
    """Traversal kernel (both instantiations of k_trace2: primary + bounce), this rank, over all timed repetitions.  Fixed denominators."""
    alg_bytes_total = float(NODE_PAIR_BYTES) * pairs + float(TRI_BYTES) * tri_tests + float(RAY_BYTES) * traversed
    launches = max(1, st["trace_launches"])
    alg_bytes_launch = alg_bytes_total / launches
    avg_launch_s = st["trace_ms_total"] * 1e-3 / launches
    achieved = alg_bytes_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    peak_hit = gather_ceiling(16)           # 4 MB set: every block an L1 miss served by L2
    peak_miss = gather_ceiling(21)          # 128 MB set (the scene's working-set size): every block misses L2, served by Infinity Cache
    pmc = None
    if world == 1 and not args.no_pmc and not args.pmc_child:
        pmc = pmc_passes(args, launches // max(1, args.repeats))
    traffic = None
    out = {"bound": "l2", "kernel": "k_trace2 (persistent while-while BVH traversal)", "achieved": round(achieved, 1), "peak": L2_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / L2_PEAK_GBS, 4), "frac_hbm_algorithmic": round(achieved / HBM_PEAK_GBS, 4), "frac_hbm_counters": None, "traffic": None,
           "peak_source": "MI355X_MICROARCH.md: aggregate L2 bandwidth ~34.5 TB/s (8 XCDs x 4 MiB); the working set of this kernel is L2 + Infinity-Cache resident, so HBM (8 TB/s) is not the roof - both HBM fractions are under `hbm`",
           "alg_bytes_per_launch": int(alg_bytes_launch), "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches": int(launches), "samples_per_launch": samples_per_launch,
           "alg_bytes_definition": "64 B x node-pair visits + 48 B x triangle tests + 72 B x rays entering the kernel (DESIGN.md 5), exact counts of the counting build",
           "node_pair_visits_per_step": int(pairs / max(1, args.steps * max(1, args.repeats))), "triangle_tests_per_step": int(tri_tests / max(1, args.steps * max(1, args.repeats))),
           "hbm": {"peak": HBM_PEAK_GBS, "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 4), "counter_gbs": None, "counter_frac": None, "copy_measured_gbs": hbm_copy_gbs(torch, device),
                   "note": "algorithmic_frac > 1 is possible and says only that the bytes are served from cache; counter_* = (FETCH_SIZE + WRITE_SIZE) of this run's launches"},
           "gather_measured": {"gbs": peak_hit, "frac": round(achieved / peak_hit, 4) if peak_hit else None, "l2_miss_set_gbs": peak_miss, "ubench_sha16": file_sha16(os.path.join(ROOT, "tools", "ubench_lines.bin")),
                               "what": "not a guide peak: tools/ubench_lines.bin run by this bench - independent random 64-B block fetches (4 x 16-B loads per lane, the node-pair fetch), 32 waves/CU, from a 4 MB set (L2 hits) and from a 128 MB set (L2 misses served by the Infinity Cache)"}}
    if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0          # KiB -> bytes; FETCH_SIZE calibrated at 1.03 on this 64-B gather pattern (profiles/r01_bench_pmc_summary.json), no doubling
        out["traffic"] = int(traffic)
        out["hbm"]["counter_gbs"] = round(traffic / avg_launch_s / 1e9, 1); out["hbm"]["counter_frac"] = round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4)
        out["frac_hbm_counters"] = out["hbm"]["counter_frac"]
        h, m, rq, ac = pmc.get("TCC_HIT_sum"), pmc.get("TCC_MISS_sum"), pmc.get("TCP_TCC_READ_REQ_sum"), pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum")
        out["pmc"] = {"source": "this run: its timed region re-executed under rocprofv3 --kernel-trace --pmc (3 passes), mean over the timed k_trace2 launches",
                      "fetch_bytes_per_launch": int(pmc["FETCH_SIZE"] * 1024.0), "write_bytes_per_launch": int(pmc["WRITE_SIZE"] * 1024.0),
                      "l1_miss_bytes_per_launch": int(rq * 64.0) if rq is not None else None, "l2_miss_bytes_per_launch": int(m * 128.0) if m is not None else None,
                      "l1_hit_rate": round(1.0 - rq / ac, 4) if rq is not None and ac else None, "l2_hit_rate": round(h / (h + m), 4) if h is not None and m is not None and h + m > 0 else None,
                      "units": "L1 misses = TCP_TCC_READ_REQ x 64 B; L2 misses = TCC_MISS x 128-B lines; FETCH/WRITE_SIZE KiB x 1024"}
        if pmc.get("SQ_INSTS_VALU"):
            # issue side (a fourth pass): wave64 VALU instructions of a launch against the slots its duration offers (256 CUs x 4 SIMDs, 2.4 clk per wave64
            # instruction on this chip: profiles/r02_ubench_halfwave.txt, 2.4 GHz) — the figure that says something about launches whose working set is cache-resident
            slots = avg_launch_s * 2.4e9 * 1024.0 / 2.4
            out["pmc"]["issue"] = {"valu_insts_per_launch": int(pmc["SQ_INSTS_VALU"]), "valu_issue_frac_at_2p4clk": round(pmc["SQ_INSTS_VALU"] / slots, 4) if slots > 0 else None,
                                   "valu_lane_utilisation": round(pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"]), 4) if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU") else None,
                                   "vmem_read_insts_per_launch": int(pmc.get("SQ_INSTS_VMEM_RD", 0)), "lds_insts_per_launch": int(pmc.get("SQ_INSTS_LDS", 0)), "salu_insts_per_launch": int(pmc.get("SQ_INSTS_SALU", 0)),
                                   "waves_per_launch": int(pmc.get("SQ_WAVES", 0))}
    else:
        out["pmc"] = None
        out["pmc_reason"] = ("--no-pmc" if args.no_pmc else "not an N = 1 run" if world != 1 else "the rocprofv3 --pmc child passes did not produce counters for this launch schedule (rocprofv3 missing or failed): achieved / frac above are from HIP events alone")
    return out


def single_frame(pt, depth, frames=40):
    """SURVEY 8(d) protocol: one frame at a time — ResetAccumulation, Compute, wait for it — nothing deferred, nothing batched."""
    pt.set_max_batch(1)
    for _ in range(5):
        pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
    pt.reset_stats()
    ts = []
    for _ in range(frames):
        t0 = time.perf_counter()
        pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
        ts.append(time.perf_counter() - t0)
    st = pt.stats()
    rays = st["rays_traced"] / frames
    trav = st["alive_counts"][0] + sum(st["alive_counts"][1:depth])
    med = statistics.median(ts)
    return {"mray_s": round(rays / med / 1e6, 1), "ms_per_frame": round(med * 1e3, 4), "traversed_mray_s": round(trav / med / 1e6, 1), "frames": frames,
            "protocol": "idkptSetMaxBatch(1); per frame: ResetAccumulation, Compute, Synchronize; median frame"}


def timed_batch(pt, B, steps, reps=5):
    pt.set_max_batch(B)
    for _ in range(B):
        pt.Compute()
    pt.synchronize(); pt.reset_stats()
    ts = []
    for _ in range(reps):
        pt.ResetAccumulation()
        t0 = time.perf_counter()
        for _ in range(steps):
            pt.Compute()
        pt.synchronize()
        ts.append(time.perf_counter() - t0)
    st = pt.stats()
    return st["rays_traced"] / reps, statistics.median(ts)


def interior_extras(S, pt, w, h, B):
    """Secondary workload: camera inside the soup, every pixel traverses (>= 95 % of the primary rays hit).  Same library calls."""
    out = {"workload": "soup-1M interior: camera at the origin inside the soup, 1920x1080, sort off"}
    pt.SetCamera(view_camera(S, "interior", w, h))
    for depth in (2, 5):
        pt.RayDepth = depth
        rays, dt = timed_batch(pt, B, 2 * B)
        st = pt.stats()
        out[f"depth{depth}"] = {"mray_s": round(rays / dt / 1e6, 1), "ms_per_step": round(dt / (2 * B) * 1e3, 4), "rays_per_step": int(rays / (2 * B)), "samples_in_flight": B,
                                "primary_hit_fraction": round(st["alive_counts"][1] / float(w * h), 4) if depth > 1 else None}
        if depth == 2:
            out["depth2"]["single_frame"] = single_frame(pt, depth, frames=20)
    return out


def atrium_extras(S, NativeBuilder, pt, w, h, B):
    """Secondary workload, the "Sponza-class" stand-in proper: a procedural colonnaded hall (connected surfaces, empty space, occlusion; the
    reference's Sponza.gltf comes without its geometry buffer), camera inside, 1M triangles at RayDepth 2 and ~262 k triangles at RayDepth 5
    (BASELINE.json configs[1]: Sponza ~260k tris, 1 spp, 4 bounces)."""
    out = {"workload": "procedural atrium (idkengine_amd/scenes.py:atrium_scene), 1920x1080, camera inside looking down the hall, sort off"}
    for tris, depth in ((1_000_000, 2), (262_000, 5)):
        sc = S.atrium_scene(tris, NativeBuilder())
        pt.UploadScene(sc); pt.SetCamera(S.atrium_camera(w, h)); pt.RayDepth = depth
        rays, dt = timed_batch(pt, B, 2 * B)
        st = pt.stats()
        e = {"blas_triangles": int(len(sc.blas_triangles)), "mray_s": round(rays / dt / 1e6, 1), "ms_per_step": round(dt / (2 * B) * 1e3, 4), "rays_per_step": int(rays / (2 * B)),
             "samples_in_flight": B, "primary_hit_fraction": round(st["alive_counts"][1] / float(w * h), 4)}
        e["single_frame"] = single_frame(pt, depth, frames=20)
        out[f"atrium_{tris // 1000}k_depth{depth}"] = e
    return out


def wide_extras(S, NativeBuilder, pt, scene, cam, w, h, B):
    """Secondary block: the wide-node walk (developer option "wide", csrc/kernels_wide.hpp: 4-wide quantised nodes derived from the reference's BVH2, rays it does not
    vouch for re-traced by k_trace2; bit-identical frames, tests/test_gpu_wide.py) against the default on the three views — it is NOT the default: profiles/r05_wide_nodes.md."""
    out = {"what": "option wide = 1 vs the default (k_trace2 alone): Mray/s with the bench's samples in flight and one frame at a time, rays re-traced by the exact kernel, and the walk's own fetch counts"}
    atrium = S.atrium_scene(N_TRIS, NativeBuilder())
    for name, sc, cm in (("headline", scene, cam), ("interior", scene, view_camera(S, "interior", w, h)), ("atrium", atrium, S.atrium_camera(w, h))):
        pt.UploadScene(sc); pt.SetCamera(cm); pt.RayDepth = RAY_DEPTH
        row = {}
        for wide in (0, 1):
            pt.set_option("wide", wide)
            rays, dt = timed_batch(pt, B, 2 * B)
            row["wide" if wide else "default"] = {"mray_s": round(rays / dt / 1e6, 1), "single_frame_mray_s": single_frame(pt, RAY_DEPTH, frames=12)["mray_s"]}
        pt.set_option("wide", 1); pt.set_option("wide_count", 1); pt.set_max_batch(4); pt.reset_stats(); pt.ResetAccumulation()
        for _ in range(4):
            pt.Compute()
        pt.synchronize(); st = pt.stats()
        pt.set_option("wide_count", 0); pt.set_option("wide", 0)
        row["per_step"] = {"wide_node_visits": int(st["wide_node_visits"] / 4), "leaf_records": int(st["wide_leaf_records"] / 4), "triangle_tests": int(st["wide_triangle_tests"] / 4),
                           "rays_retraced_exactly": round(st["wide_flagged_rays"] / 4.0, 1), "rays": int(st["rays_traced"] / 4)}
        row["speedup_batched"] = round(row["wide"]["mray_s"] / row["default"]["mray_s"], 3)
        out[name] = row
    return out


def packet_extras(S, NativeBuilder, pt, scene, cam, w, h, B):
    """Secondary block: the primary launch as a packet launch (developer option "packet", csrc/kernels_packet.hpp: one shared BVH2 walk per wave of 64 pixel-major work-list
    entries, node pairs through the scalar cache; rays it does not vouch for re-traced by k_trace2; bit-identical frames, tests/test_gpu_packet.py).  Default = 1: by the
    kernel's own counters (live lanes per node step) — on where every pixel traverses, off on the headline view whose pixels are wider than its triangles
    (profiles/r06_packet.md).  RayDepth 1 rows time the primary launch alone."""
    out = {"what": "option packet 0 (k_trace2 alone) / 2 (always) / 1 (default: by measurement): Mray/s with the bench's samples in flight; live = lanes live per node step / 64; rays re-traced by the exact kernel"}
    atrium = S.atrium_scene(N_TRIS, NativeBuilder())
    for name, sc, cm in (("headline", scene, cam), ("interior", scene, view_camera(S, "interior", w, h)), ("atrium", atrium, S.atrium_camera(w, h))):
        pt.UploadScene(sc); pt.SetCamera(cm)
        row = {}
        for depth in (1, RAY_DEPTH):
            pt.RayDepth = depth
            for mode, key in ((0, "off"), (2, "forced"), (1, "default")):
                pt.set_option("packet", mode)
                rays, dt = timed_batch(pt, B, 2 * B, reps=3)
                st = pt.stats()
                e = {"mray_s": round(rays / dt / 1e6, 1)}
                if st["packet_packets"]:
                    e.update(live=round(st["packet_live_lanes"] / (64.0 * max(1, st["packet_node_steps"])), 3), node_steps_per_packet=round(st["packet_node_steps"] / st["packet_packets"], 1),
                             triangle_rounds_per_packet=round(st["packet_triangle_rounds"] / st["packet_packets"], 1), retraced_fraction=round(st["packet_flagged_rays"] / max(1, st["packet_rays_entered"]), 5))
                row.setdefault("primary_only" if depth == 1 else f"depth{depth}", {})[key] = e
        for k in row:
            row[k]["forced_vs_off"] = round(row[k]["forced"]["mray_s"] / row[k]["off"]["mray_s"], 3); row[k]["default_vs_off"] = round(row[k]["default"]["mray_s"] / row[k]["off"]["mray_s"], 3)
        out[name] = row
    pt.set_option("packet", 1)
    return out


def sort_extras(S, NativeBuilder, pt, scene, cam, w, h, B):
    """BASELINE.json configs[2]'s second half: DoRaySorting on vs off (PathTracer.cs:273-297: a counting sort of every bounce's queue by the previous hit's triangle id, from the
    second bounce on) — the headline view at RayDepth 2 (nothing is sorted there: the first sort precedes bounce 2) and 5, the interior view and the 262 k-triangle atrium at RayDepth 5,
    the regime the reference built it for (Gui.cs:745-749)."""
    out = {"what": "Mray/s with DoRaySorting 0 / 1, same protocol as the headline (samples in flight, defer_last); sort_pays = on / off"}
    atrium = S.atrium_scene(262_000, NativeBuilder())
    for name, sc, cm, depth in (("headline_depth2", scene, cam, 2), ("headline_depth5", scene, cam, 5), ("interior_depth5", scene, view_camera(S, "interior", w, h), 5), ("atrium_262k_depth5", atrium, S.atrium_camera(w, h), 5)):
        pt.UploadScene(sc); pt.SetCamera(cm); pt.RayDepth = depth
        row = {}
        for srt in (0, 1):
            pt.DoRaySorting = srt
            rays, dt = timed_batch(pt, B, 2 * B, reps=3)
            row["sort_on" if srt else "sort_off"] = {"mray_s": round(rays / dt / 1e6, 1), "ms_per_step": round(dt / (2 * B) * 1e3, 4)}
        row["sort_pays"] = round(row["sort_on"]["mray_s"] / row["sort_off"]["mray_s"], 3)
        out[name] = row
    pt.DoRaySorting = 0
    return out


def atrium_per_mesh_extras(S, NativeBuilder, pt, w, h, B):
    """The atrium as the reference would hold it: one BLAS per mesh (87 BLASes, Bvh/BVH.cs:156), through the instance loop (the default) and through a TLAS built on
    the device (idkptBuildTlasOnDevice), against the same triangles hoisted into one BLAS (atrium block)."""
    sc = S.atrium_scene(N_TRIS, NativeBuilder(), per_mesh_blas=True)
    out = {"workload": f"procedural atrium, {len(sc.blas_descs)} BLASes (one per mesh), {len(sc.blas_triangles)} BLAS triangles, {w}x{h}, RayDepth {RAY_DEPTH}, camera inside"}
    pt.UploadScene(sc); pt.SetCamera(S.atrium_camera(w, h)); pt.RayDepth = RAY_DEPTH
    # instance_loop: what a host that has not set UseTlas gets — the loop's hits.  The library's defaults: every instance of this scene carries the same transform, so the scene is
    # walked as ONE tree in the instances' common BLAS space (round 6: k_braid + k_unify_* + k_trace_inst TREE 1, the primary launch as packets over the same tree; rays whose result
    # could depend on the loop's order are traced again by the exact loop with its instance sieve; profiles/r06_braid.md).  instance_loop_own_tlas: round 5's path, the library's own TLAS
    # over whole instances (what scenes of different transforms and little overlap get); instance_loop_sieved: the exact loop with the sieve as the main kernel (scenes of more
    # overlap); instance_loop_k_trace2: the loop inside k_trace2 (MODE 1: round 4's path)
    for name, tlas, own, sieve, unify in (("instance_loop", 0, 8, 8, 4096), ("instance_loop_own_tlas", 0, 8, 8, 0), ("instance_loop_sieved", 0, 0, 8, 0), ("instance_loop_k_trace2", 0, 0, 0, 0), ("tlas_built_on_device", 1, 8, 8, 4096)):
        if tlas:
            pt.BuildTlasOnDevice()
        pt.UseTlas = tlas
        pt.set_option("inst_tlas", own); pt.set_option("inst_sieve", sieve); pt.set_option("inst_unify", unify)
        pt.reset_stats()
        rays, dt = timed_batch(pt, B, B, reps=3)
        st = pt.stats(); flagged = st["inst_tlas_flagged_rays"] + st["packet_flagged_rays"]; traced = st["rays_traced"]
        out[name] = {"mray_s": round(rays / dt / 1e6, 1), "ms_per_step": round(dt / B * 1e3, 4), "single_frame_mray_s": single_frame(pt, RAY_DEPTH, frames=8)["mray_s"]}
        if name == "instance_loop":
            out[name].update(rays_retraced_by_the_exact_loop=round(flagged / max(traced, 1), 5), unified_tree_entries=st["inst_unified_entries"], unified_tree_top_depth=st["inst_unified_top_depth"])
    pt.UseTlas = 0; pt.set_option("inst_tlas", 8); pt.set_option("inst_sieve", 8); pt.set_option("inst_unify", 4096)
    return out


def multi_blas_extras(S, NativeBuilder, pt, B):
    """Secondary workload: the reference's DEFAULT scene shape, several models with one BLAS each and no TLAS (Source/Application.cs:484, Bvh/BVH.cs:17-25:
    the instance loop of BVHIntersect.glsl:275-287 = k_trace2 MODE 1) and the same through the TLAS (MODE 2), against the same triangles in one BLAS."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_multi
    from idkengine_amd.pathtracer import PathTracer
    out = bench_multi.run(S, PathTracer, NativeBuilder(), N_TRIS, 3, "headline", RAY_DEPTH, B, W, H, pt=pt)
    inner = bench_multi.run(S, PathTracer, NativeBuilder(), N_TRIS, 3, "interior", RAY_DEPTH, B, W, H, pt=pt)
    out["interior_view"] = {k: inner[k] for k in ("instance_loop", "tlas", "one_blas_same_triangle_count")}
    pt.UseTlas = 0
    return out


def animated_extras(S, NativeBuilder, pt, frames=64):
    """Secondary workload (BASELINE.json configs[4], stand-in): refittable soup-1M, every vertex skinned by two joints, BLAS refit and TLAS rebuild on the device
    every frame (ModelManager.cs:263-361), one 1-spp RayDepth-2 frame per geometry; one frame at a time and with frames in flight through scene versions."""
    import numpy as np
    from idkengine_amd import gputypes as T
    sc = S.soup_scene(N_TRIS, NativeBuilder(), seed=1, refittable=True)
    nv = len(sc.vertex_positions)
    un = np.zeros(nv, T.GpuUnskinnedVertex)
    un["Position"] = sc.vertex_positions; un["Normal"] = sc.vertices["Normal"]; un["Tangent"] = sc.vertices["Tangent"]; un["JointIndices"][:, 1] = 1
    wgt = (0.5 + 0.5 * np.sin(sc.vertex_positions[:, 0] * 0.7)).astype(np.float32); un["JointWeights"][:, 0] = wgt; un["JointWeights"][:, 1] = 1.0 - wgt
    pt.UploadScene(sc); pt.SetCamera(S.Camera(W, H)); pt.RayDepth = 2; pt.UseTlas = 0
    pt.UploadUnskinnedVertices(un)

    def joints(t):
        j = np.zeros((2, 3, 4), np.float32); j[0, :, :3] = np.eye(3)
        c, s_ = np.cos(0.05 * np.sin(t)), np.sin(0.05 * np.sin(t))
        j[1, :, :3] = [[c, 0, s_], [0, 1, 0], [-s_, 0, c]]; j[0, :, 3] = (0.0, 0.05 * np.sin(1.3 * t), 0.0)
        return j

    out = {"workload": "refittable soup-1M (3 M vertices), per frame: 2 joint matrices -> idkptSkin -> idkptRefitBlas -> idkptBuildTlasOnDevice -> one 1-spp RayDepth-2 frame, 1920x1080"}
    for F in (1, 8, 32):
        pt.SetSceneVersions(2 * F if F > 1 else 1); pt.SetFrameRing(F); pt.set_max_batch(F)
        for rep_ in range(2):
            pt.synchronize(); pt.reset_stats(); t0 = time.perf_counter()
            for k in range(frames):
                pt.UpdateBuffer(T.IDKPT_BUF_JOINT_MATRICES, joints(0.5 + 0.1 * k)); pt.Skin(0, 0, 0, nv); pt.RefitBlas(0); pt.BuildTlasOnDevice()
                pt.BeginFrame(); pt.Compute()
            pt.flush(); pt.synchronize(); dt = (time.perf_counter() - t0) / frames
        out[f"frames_in_flight_{F}"] = {"mray_s": round(pt.stats()["rays_traced"] / frames / dt / 1e6, 1), "ms_per_animated_frame": round(dt * 1e3, 4), "scene_versions": 2 * F if F > 1 else 1}
    pt.SetSceneVersions(1); pt.SetFrameRing(1)
    return out


def query_extras(S, pt, scene, cam):
    """The adjacent consumers: idkptTraceRays (closest / any hit, BVHIntersect.glsl:183-411) on one primary ray per pixel, and idkptTraceShadows
    (Shaders/ShadowsRayTraced/compute.glsl) on the G-buffer of those hits.  Thread-per-ray kernels on the general traversal code; the times are whole calls,
    i.e. they include the PCIe copies of the ray / hit arrays (64 B per ray) and of the G-buffer."""
    import numpy as np
    from idkengine_amd import gputypes as T
    rays = S.primary_ray_queries(cam, W, H)
    pt.TraceRays(rays[:4096])
    out = {"workload": f"soup-{N_TRIS}, {W}x{H}: one ray per pixel centre; times are whole calls incl. host <-> device copies (32 B per ray in, 32 B per hit out)"}
    for name, any_hit in (("closest_hit", False), ("any_hit", True)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); hits = pt.TraceRays(rays, any_hit=any_hit); ts.append(time.perf_counter() - t0)
        out[name] = {"mray_s": round(len(rays) / statistics.median(ts) / 1e6, 1), "ms_per_call": round(statistics.median(ts) * 1e3, 3), "rays": int(len(rays))}
    hits = pt.TraceRays(rays)
    depth, normal = S.gbuffer_from_hits(scene, cam, W, H, rays, hits)
    lights = S.make_lights([((0.0, 30.0, 10.0), 0.5, (50.0, 50.0, 50.0))])
    pt.UpdateBuffer(T.IDKPT_BUF_LIGHTS, lights); pt._check(pt._L.idkptSetLightCount(pt._ctx, 1))
    prm = T.ShadowParams.make(cam.inv_proj_view, W, H, light_index=0, samples=1, noise_index=0, jitter=(0.0, 0.0))
    pt.TraceShadows(prm, depth, normal)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); pt.TraceShadows(prm, depth, normal); ts.append(time.perf_counter() - t0)
    out["rt_shadows"] = {"mray_s": round(W * H / statistics.median(ts) / 1e6, 1), "ms_per_call": round(statistics.median(ts) * 1e3, 3), "shadow_rays": W * H, "samples": 1}
    # the same queries with rays / G-buffer / results resident in HBM (idkptTraceRaysDevice / idkptTraceShadowsDevice): what the kernels themselves do
    try:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        d_rays = torch.from_numpy(np.frombuffer(rays.tobytes(), np.uint8).copy()).to(dev); d_hits = torch.zeros(len(rays) * 32, dtype=torch.uint8, device=dev)
        d_depth = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).to(dev); d_normal = torch.from_numpy(np.ascontiguousarray(normal, np.float32)).to(dev); d_vis = torch.zeros(H * W, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        res = {}
        for name, fn in (("closest_hit", lambda: pt.TraceRaysDevice(d_rays.data_ptr(), d_hits.data_ptr(), len(rays))), ("any_hit", lambda: pt.TraceRaysDevice(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), any_hit=True)),
                         ("rt_shadows", lambda: pt.TraceShadowsDevice(prm, d_depth.data_ptr(), d_normal.data_ptr(), d_vis.data_ptr()))):
            fn(); pt.synchronize(); ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(4):
                    fn()
                pt.synchronize(); ts.append((time.perf_counter() - t0) / 4.0)
            res[name] = {"mray_s": round(len(rays) / statistics.median(ts) / 1e6, 1), "ms_per_call": round(statistics.median(ts) * 1e3, 3)}
        out["device_resident"] = res
    except Exception as e:   # noqa: BLE001
        out["device_resident"] = {"error": str(e)}
    pt._check(pt._L.idkptSetLightCount(pt._ctx, 0))
    return out


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(S, scene, depth, view):
    """The oracle (CPU port of the reference path, GLSL semantics, OpenMP over all host cores) on a bounded sample of the same frame:
    every m-th row, m chosen so the run takes roughly 10-20 s; scene, buffers and thread pool are created once and kept warm.
    `value` is the whole path; `parallel_section` times only the per-invocation sections (FirstHit / NHit / FinalDraw loops, what
    the GPU kernels replace) without the serial queue compaction in between."""
    from oracle import oracle as O   # allowed here: cpu_baseline leg only
    cores = os.cpu_count() or 1
    eff = [cores]                                           # cores the process can actually occupy (set below: min(threads, cgroup quota))

    def measure(cam, budget_s):
        probe = O.OraclePathTracer(scene, W, H, row_modulo=32, row_remainder=0); probe.set_camera(cam); probe.settings.RayDepth = depth
        probe.render(); probe.timing()                      # warm-up (thread pool, page faults)
        t0 = time.perf_counter(); probe.render(); dt = time.perf_counter() - t0
        probe.close()
        full_est = dt * 32
        mod = 1 if full_est <= 2.0 else max(1, int(full_est / 2.0 + 0.999))       # one repetition ~ <= 2 s
        o = O.OraclePathTracer(scene, W, H, row_modulo=mod, row_remainder=0); o.set_camera(cam); o.settings.RayDepth = depth
        o.render(); o.timing(); r0 = o.stats()["rays_traced"]                     # warm-up of this instance
        tot_dt, reps = 0.0, 0
        while tot_dt < budget_s and reps < 400:
            t0 = time.perf_counter(); o.render(); tot_dt += time.perf_counter() - t0; reps += 1
        par_s, _ = o.timing()
        rays = o.stats()["rays_traced"] - r0; rows = o.rows
        o.close()
        return {"value": round(rays / tot_dt / 1e6, 3), "parallel_section": round(rays / par_s / 1e6, 3), "per_core": round(rays / par_s / 1e6 / eff[0], 4),
                "sample": f"rows y%{mod}==0 of the {W}x{H} frame ({rows} rows, RayDepth {depth}) x {reps} repetitions = {tot_dt:.1f} s of CPU work ({par_s:.1f} s inside the OpenMP sections)"}

    # the host may expose more hardware threads than it lets a process use well (cgroup quota, SMT, 2 sockets): pick the OpenMP thread count
    # that is fastest on a short probe and report THAT count as the cores used
    quota = None
    try:                                                    # a container CPU quota (cgroup v2 cpu.max) is the real core count of this process
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    cand = {16, 32, 64, 128, cores}
    if quota:                                               # (under a quota more than 2x its threads only oversubscribe: a noisy probe picked 128 threads on a 16-core quota once and halved the baseline)
        cand = {max(1, int(quota)), max(1, int(2 * quota))}
    best = (0.0, cores)
    for nthr in sorted(t for t in cand if t <= cores):
        O.set_num_threads(nthr)
        probe = O.OraclePathTracer(scene, W, H, row_modulo=8, row_remainder=0); probe.set_camera(view_camera(S, view, W, H)); probe.settings.RayDepth = depth
        probe.render(); r0 = probe.stats()["rays_traced"]
        t0 = time.perf_counter(); probe.render(); probe.render(); dtp = time.perf_counter() - t0
        rate = (probe.stats()["rays_traced"] - r0) / dtp
        probe.close()
        if rate > best[0]:
            best = (rate, nthr)
    hw_threads, cores = cores, best[1]
    eff[0] = min(cores, int(quota)) if quota else cores
    O.set_num_threads(cores)
    head = measure(view_camera(S, view, W, H), 10.0)
    # the reference's own CPU path (C# semantics: Gui.Test -> BVH.Intersect -> BLAS.Intersect, Render/Gui.cs:1484-1503): primary rays only
    cam = view_camera(S, view, W, H)
    O.cpu_trace_primary(scene, cam, W, H, threads=cores, want_hits=False)
    p_rays, p_dt = 0, 0.0
    while p_dt < 3.0:
        t0 = time.perf_counter(); rr = O.cpu_trace_primary(scene, cam, W, H, threads=cores, want_hits=False); p_dt += time.perf_counter() - t0
        p_rays += int(rr["rays"])
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = None
    out = {"value": head["value"], "unit": "Mray/s", "cores": eff[0], "omp_threads": cores, "hw_threads": hw_threads, "cgroup_cpu_quota": quota, "sched_affinity": affinity, "cpu": cpu_model(), "kind": "port", "parallel_section_mray_s": head["parallel_section"],
           "mray_s_per_core": head["per_core"], "sample": head["sample"] + f"; {cores} OpenMP threads (the fastest on a probe of {'1x/2x the container CPU quota of ' + str(quota) if quota else '16/32/64/128/' + str(hw_threads)}; `cores` = min(threads, quota): what the process can actually occupy); C++/OpenMP restatement of the reference path incl. shading (the C# binary cannot run here: no .NET); scene and threads warm",
           "primary_only_csharp_semantics": {"value": round(p_rays / p_dt / 1e6, 3), "unit": "Mray/s", "per_core": round(p_rays / p_dt / 1e6 / eff[0], 4),
                                             "sample": f"{p_rays // (W * H)} full {W}x{H} frames of centre-of-pixel primary rays, closest hit only (no shading), {p_dt:.1f} s"},
           "note": "a reported baseline, not the target: the GPU/CPU ratio says nothing about kernel quality (roofline.frac does)"}
    # BASELINE.json configs[0]: Cornell box (32 triangles), 256x256, through the reference's CPU traversal path (C# semantics, Gui.cs:1484-1503) — primary rays —
    # and, as "primary + 1 bounce", the oracle's port of the whole path at RayDepth 2 on the same frame
    try:
        from idkengine_amd.bvh import NativeBuilder
        csc = S.cornell_scene(NativeBuilder(), variant="diffuse"); ccam = S.cornell_camera(256, 256)
        O.cpu_trace_primary(csc, ccam, 256, 256, threads=cores, want_hits=False)
        c_rays, c_dt = 0, 0.0
        while c_dt < 1.0:
            t0 = time.perf_counter(); rr = O.cpu_trace_primary(csc, ccam, 256, 256, threads=cores, want_hits=False); c_dt += time.perf_counter() - t0; c_rays += int(rr["rays"])
        co = O.OraclePathTracer(csc, 256, 256); co.set_camera(ccam); co.settings.RayDepth = 2; co.render(); r0 = co.stats()["rays_traced"]
        o_dt, reps = 0.0, 0
        while o_dt < 1.0:
            t0 = time.perf_counter(); co.render(); o_dt += time.perf_counter() - t0; reps += 1
        o_rays = co.stats()["rays_traced"] - r0; co.close()
        out["cornell_256"] = {"workload": "BASELINE.json configs[0]: Cornell box (32 triangles), 256x256, 1 spp", "cores": eff[0], "omp_threads": cores,
                              "primary_only_csharp_semantics_mray_s": round(c_rays / c_dt / 1e6, 3), "primary_plus_1_bounce_oracle_port_mray_s": round(o_rays / o_dt / 1e6, 3),
                              "sample": f"{c_rays // 65536} frames of primary rays in {c_dt:.1f} s; {reps} RayDepth-2 frames of the oracle port in {o_dt:.1f} s"}
    except Exception as e:   # noqa: BLE001
        out["cornell_256"] = {"error": str(e)}
    if view == "headline":
        inter = measure(view_camera(S, "interior", W, H), 6.0)
        out["interior_view"] = {"value": inter["value"], "parallel_section_mray_s": inter["parallel_section"], "mray_s_per_core": inter["per_core"], "sample": inter["sample"]}
    return out


if __name__ == "__main__":
    main()
