"""N-GPU self-test of both multi-GPU hosts, meant for the first run on a real multi-GPU node (the build box has one GPU):

  python tools/scale_selftest.py [--gpus N]
      one process, ONE multi-device context (idkptCreate(deviceCount = N)): prints the visible devices and the peer-access matrix, renders a
      small frame on 1 device and on N members (ids wrap around when fewer GPUs are visible), with xGMI peer copies and with every copy staged
      through the host (option "force_no_peer"), at RayDepth 2 and 5 (bands of 8 rows; beyond RayDepth 2 with the per-band count exchange), plus explicit strips + device-side count exchange at RayDepth 5, and compares the bits.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/scale_selftest.py
      one process per GPU: prints rank -> device, checks that RCCL ("nccl") sees N ranks (all-reduce of the ranks), renders the same frame
      row-sharded over the ranks (dist.GpuShardRenderer + ShardedFrame, all-gather over RCCL) and rank 0 compares it with its own 1-device frame.

Every line starts with [selftest]; the last one is "[selftest] PASS" or "[selftest] FAIL ...".  Exit code 0 / 1."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only (RCCL between processes)


def log(*a):
    print("[selftest]", *a, flush=True)


def peer_matrix(torch, ids):
    return [[1 if i == j else int(torch.cuda.can_device_access_peer(i, j)) for j in ids] for i in ids]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from idkengine_amd.pathtracer import PathTracer
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev == 0:
        log("FAIL no GPU visible"); return 1
    w, h = 192, 121
    sc = S.soup_scene(20000, NativeBuilder(), seed=7, extent=3.0); cam = S.Camera(w, h, position=(0.0, 0.0, 8.0))
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)   # noqa: E731
    ok = True

    def one_device(depth, frames, device=0):
        p = PathTracer(w, h, device=device); p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = depth; p.set_max_batch(2)
        for _ in range(frames):
            p.Compute()
        img = p.Result; p.Dispose()
        return img

    if world == 1:
        n = args.gpus or ndev
        ids = [d % ndev for d in range(n)]
        log(f"one process, {ndev} visible device(s): " + ", ".join(f"{i}={torch.cuda.get_device_name(i)}" for i in range(ndev)))
        log(f"members -> devices {ids}; peer-access matrix over devices {sorted(set(ids))}: {peer_matrix(torch, sorted(set(ids)))}")
        rc, ver, detail = PathTracer.transport_self_test(0)
        log(f"RCCL self test on device 0 (one-rank communicator: broadcast, send / recv, all-gather): status {rc}, version {ver}, {detail}")
        ok &= rc == 0
        for depth in (2, 5):
            want = one_device(depth, 3)
            for no_peer, transport in ((0, 0), (0, 1), (1, 1)):       # RCCL where the communicators can be formed (distinct devices); peer copies; host-staged copies
                g = PathTracer(w, h, devices=ids)
                g.set_option("force_no_peer", no_peer); g.set_option("transport", transport)
                g.UploadScene(sc); g.SetCamera(cam); g.RayDepth = depth; g.set_max_batch(2)
                for _ in range(3):
                    g.Compute()
                same = bool((bits(g.Result) == bits(want)).all())
                ptr, nbytes = g.image_device_ptr(0); g.synchronize()      # the gather of the members' rows on device 0 runs too
                info = g.transport_info()
                log(f"RayDepth {depth}, transport {info['transport']} ({info['rccl_ranks']} RCCL ranks; {info['detail']}){', host-staged copies' if no_peer else ''}: {n}-member context == 1 device: {same}")
                ok &= same and nbytes == w * h * 16
                if transport == 0 and len(set(ids)) == len(ids) and len(ids) > 1:
                    ok &= info["transport"] == "rccl" and info["rccl_ranks"] == len(ids)          # on a real N-GPU node RCCL must have carried this run
                if depth > 2 and not no_peer and transport == 1:        # round 2's deal: contiguous strips + device-side, event-ordered count exchange (no host synchronisation)
                    g.SetGroupSharding(2)            # (a change of layout restarts the accumulation)
                    for _ in range(3):
                        g.Compute()
                    same = bool((bits(g.Result) == bits(want)).all()); ok &= same
                    log(f"RayDepth {depth}, explicit strips + device-side count exchange: {n}-member context == 1 device: {same}")
                g.Dispose()
    else:
        import torch.distributed as dist
        from idkengine_amd import dist as D
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        one_dev = os.environ.get("IDKPT_BENCH_ONE_DEVICE") == "1"          # developer: every rank on GPU 0 over gloo (control flow only)
        dev_index = 0 if one_dev else local_rank
        torch.cuda.set_device(dev_index); device = torch.device("cuda", dev_index)
        dist.init_process_group("gloo" if one_dev else "nccl", rank=rank, world_size=world, **({} if one_dev else {"device_id": device}))
        t = torch.tensor([float(rank + 1)], device=device if not one_dev else "cpu"); dist.all_reduce(t)
        seen = dist.get_world_size(); sum_ok = abs(t.item() - world * (world + 1) / 2) < 1e-6
        log(f"rank {rank}/{world} on cuda:{dev_index} ({torch.cuda.get_device_name(dev_index)}); backend {dist.get_backend()}; world seen {seen}; all-reduce of ranks {'ok' if sum_ok else 'WRONG'}")
        ok &= sum_ok and seen == world
        scene = D.broadcast_scene(sc if rank == 0 else None, src=0, device=device)
        r = D.GpuShardRenderer(w, h, world, rank, dev_index); r.upload_scene(scene); r.set_camera(cam); r.pt.RayDepth = 2; r.pt.set_max_batch(2)
        frame = D.ShardedFrame(r, w, h)
        for _ in range(3):
            r.pt.Compute()                           # three accumulated samples of this rank's rows (frame.render() would reset the accumulation)
        full = frame.gather(); torch.cuda.synchronize()
        if rank == 0:
            same = bool((bits(full.cpu().numpy()) == bits(one_device(2, 3, dev_index))).all())
            log(f"row-sharded frame over {world} ranks (all-gather) == 1 device: {same}")
            ok &= same
        r.pt.Dispose()
        # the same deal at RayDepth 5 with the per-band count exchange enqueued on the render stream (idkptSetBandExchangeDevice: all-gather + prefix sum): exact beyond RayDepth 2 as well
        r = D.GpuShardRenderer(w, h, world, rank, dev_index, exact_deep_paths=True); r.upload_scene(scene); r.set_camera(cam); r.pt.RayDepth = 5; r.pt.set_max_batch(2)
        frame = D.ShardedFrame(r, w, h)
        for _ in range(3):
            r.pt.Compute()
        full = frame.gather(); torch.cuda.synchronize()
        if rank == 0:
            same = bool((bits(full.cpu().numpy()) == bits(one_device(5, 3, dev_index))).all())
            log(f"RayDepth 5, rows in bands + per-band count exchange over {world} ranks == 1 device: {same}")
            ok &= same
        r.pt.Dispose()
        flag = torch.tensor([1.0 if ok else 0.0], device=device if not one_dev else "cpu"); dist.all_reduce(flag, op=dist.ReduceOp.MIN); ok = flag.item() > 0.5
        dist.destroy_process_group()
    if rank == 0:
        log("PASS" if ok else "FAIL (see the lines above)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
