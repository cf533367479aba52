"""Multi-GPU sharding exercised on one GPU: several contexts (rows / strips), exact deep paths, RCCL world-1."""
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


@pytest.mark.parametrize("batch", [1, 3])
def test_exact_deep_paths_across_contexts(native_builder, oracle_mod, batch):
    """idkptSetRowRange + idkptSetBounceExchange: three contexts (one per strip, driven by three host threads in lockstep, the
    exchange function summing the counts of the strips above) reproduce the single-context frame bit for bit at RayDepth 6 —
    image, ray state and the total ray count — also when several accumulated samples are traced per batch."""
    import threading
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import dist as D, gputypes as T
    sc = S.soup_scene(30000, native_builder, seed=6, extent=3.0); w, h = 200, 131; cam = S.Camera(w, h, position=(0.0, 0.0, 7.0))
    ov = dict(RayDepth=6)
    frames = 3
    one = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); one.UploadScene(sc); one.SetCamera(cam)
    for _ in range(frames):
        one.Compute()
    want = one.Result; want_rays = one.rays(); want_count = one.stats()["rays_traced"]
    world = 3
    barrier = threading.Barrier(world)
    board = {}

    def exchange_for(rank):
        def fn(bounce, counts):
            board[(bounce, rank)] = counts.copy()
            barrier.wait(timeout=60)
            base = np.zeros(len(counts), np.uint32)
            for r in range(rank):
                base += board[(bounce, r)]
            barrier.wait(timeout=60)                     # nobody overwrites the board before everybody has read it
            return base
        return fn

    pts, errs = [], []
    for r in range(world):
        p = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); p.UploadScene(sc); p.SetCamera(cam)
        first, count = D.strip_of_rank(h, world, r); p.SetRowRange(first, count); p.SetBounceExchange(exchange_for(r)); p.set_max_batch(batch)
        pts.append(p)

    def run(p):
        try:
            for _ in range(frames):
                p.Compute()
            p.flush(); p.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append(e); barrier.abort()
    threads = [threading.Thread(target=run, args=(p,)) for p in pts]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errs, errs
    got = np.concatenate([p.Result for p in pts]); got_rays = np.concatenate([p.rays() for p in pts])
    assert (bits(got) == bits(want)).all()
    assert got_rays.tobytes() == want_rays.tobytes()
    assert sum(p.stats()["rays_traced"] for p in pts) == want_count
    # control: the same strips without the exchange differ at this depth
    q = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov)); q.UploadScene(sc); q.SetCamera(cam)
    first, count = D.strip_of_rank(h, world, 1); q.SetRowRange(first, count)
    for _ in range(frames):
        q.Compute()
    assert (bits(q.Result) != bits(want[first:first + count])).any()
    for p in pts + [one, q]:
        p.Dispose()


def test_sharded_frame_over_rccl_world1(native_builder):
    """The multi-GPU driver path of dist.py / bench.py on one GPU: process group "nccl" (RCCL) with world_size 1, scene
    broadcast through GPU tensors, renderer on torch's stream, zero-copy alias of the device image, all_gather."""
    import torch
    import torch.distributed as dist
    from idkengine_amd import dist as D
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sc = D.broadcast_scene(S.cornell_scene(native_builder, "mixed"), src=0, device=torch.device("cuda", 0))
        cam = S.cornell_camera(96, 64)
        r = D.GpuShardRenderer(96, 64, 1, 0, 0); r.upload_scene(sc); r.set_camera(cam); r.pt.RayDepth = 2; r.pt.set_max_batch(4)
        frame = D.ShardedFrame(r, 96, 64)
        for _ in range(4):
            frame.render()
        full_t = frame.gather(); torch.cuda.synchronize()
        full = full_t.cpu().numpy()
        ref = gpu_render(sc, cam, 96, 64, RayDepth=2)
        assert (bits(full) == bits(ref.Result)).all()
        # frame ring over the same transport: 3 frames with their own cameras in flight, one all-gather carrying all three
        r.pt.SetFrameRing(6); r.pt.set_max_batch(3)
        cams = [S.cornell_camera(96, 64), S.Camera(96, 64, position=(0.2, 0.1, 3.0), fovy_deg=45.0), S.Camera(96, 64, position=(-0.3, 0.0, 3.2), fovy_deg=50.0)]
        slots = []
        for c in cams:
            slots.append(r.pt.BeginFrame()); r.set_camera(c); r.pt.Compute()
        assert slots == [0, 1, 2]                    # the first frame after idkptSetFrameRing uses slot 0
        frames_t = frame.gather_frames(slots[0], 3); torch.cuda.synchronize()
        for k, c in enumerate(cams):
            alone = gpu_render(sc, c, 96, 64, RayDepth=2)
            assert (bits(frames_t[k].cpu().numpy()) == bits(alone.Result)).all()
            alone.Dispose()
        ref.Dispose(); r.pt.Dispose()
    finally:
        dist.destroy_process_group()
