"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: scene broadcast and row-interleaved sharding +
all-gather (idkengine_amd/dist.py).  The per-rank renderer is the oracle here (no GPU in this container); on the GPU box
the same ShardedFrame drives idkengine_amd.PathTracer over RCCL."""
import os
import sys
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 64, 47   # odd height: ranks get different row counts (24 / 23)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleShardRenderer:
    def __init__(self, scene, cam, world, rank, row_band=1):
        from oracle import oracle as O
        self.row_band = row_band
        self.pt = O.OraclePathTracer(scene, W, H, row_modulo=world, row_remainder=rank, row_band=row_band)
        self.pt.set_camera(cam); self.pt.settings.RayDepth = 2
        self.rows, self.width = self.pt.rows, W

    def render(self):
        self.pt.reset_accumulation(); self.pt.render()

    def local_image(self):
        return torch.from_numpy(self.pt.image())


def _worker(rank, world, port, q, row_band=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from idkengine_amd import scenes as S, dist as D
    from idkengine_amd.bvh import NativeBuilder
    scene = S.cornell_scene(NativeBuilder(), "mixed", instanced=True) if rank == 0 else None
    scene = D.broadcast_scene(scene, src=0)
    cam = S.cornell_camera(W, H)
    frame = D.ShardedFrame(OracleShardRenderer(scene, cam, world, rank, row_band), W, H)
    frame.render()
    full = frame.gather().numpy()
    import hashlib
    nodes_hash = hashlib.sha256(scene.blas_nodes.tobytes() + scene.tlas_nodes.tobytes() + scene.materials.tobytes() + scene.sky_faces.tobytes()).hexdigest()
    q.put((rank, full, nodes_hash, frame.r.rows))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("row_band", [1, 8])
def test_row_sharded_frame_world2_gloo(row_band):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, row_band)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from oracle import oracle as O
    sc = S.cornell_scene(NativeBuilder(), "mixed", instanced=True)
    ref = O.OraclePathTracer(sc, W, H); ref.set_camera(S.cornell_camera(W, H)); ref.settings.RayDepth = 2; ref.render()
    want = ref.image()
    assert (res[0][3], res[1][3]) == ((24, 23) if row_band == 1 else (24, 23))   # H = 47: rows 0,2,..,46 / bands 0-7, 16-23, 32-39 against 8-15, 24-31, 40-46
    assert res[0][2] == res[1][2]                                          # broadcast scene identical on both ranks
    for _, full, _, _ in res:                                              # every rank holds the full frame, == 1-process frame bit-for-bit
        assert full.shape == (H, W, 4)
        assert (full.view(np.uint32) == want.view(np.uint32)).all()


class OracleStripRenderer:
    """exact deep paths: contiguous strip + per-bounce count exchange (dist.make_count_exchange), oracle as the per-rank renderer."""
    exact = True

    def __init__(self, scene, cam, world, rank, depth, frames):
        from oracle import oracle as O
        from idkengine_amd import dist as D
        self.pt = O.OraclePathTracer(scene, W, H)
        first, count = D.strip_of_rank(H, world, rank)
        self.pt.set_row_range(first, count)
        self.pt.set_bounce_exchange(D.make_count_exchange())
        self.pt.set_camera(cam); self.pt.settings.RayDepth = depth
        self.rows, self.width, self.frames = self.pt.rows, W, frames

    def render(self):
        self.pt.reset_accumulation()
        for _ in range(self.frames):
            self.pt.render()

    def local_image(self):
        return torch.from_numpy(self.pt.image())


def _worker_exact(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from idkengine_amd import scenes as S, dist as D
    from idkengine_amd.bvh import NativeBuilder
    scene = S.cornell_scene(NativeBuilder(), "mixed", instanced=True) if rank == 0 else None
    scene = D.broadcast_scene(scene, src=0)
    frame = D.ShardedFrame(OracleStripRenderer(scene, S.cornell_camera(W, H), world, rank, depth=6, frames=2), W, H)
    frame.render()
    q.put((rank, frame.gather().numpy(), frame.r.pt.stats()["rays_traced"]))
    dist.barrier()
    dist.destroy_process_group()


def test_exact_deep_paths_world3_gloo():
    """RayDepth 6 (order-dependent: NHit seeds from the queue slot), 3 strips of 16/16/15 rows, 2 accumulated samples: with the
    per-bounce count exchange the gathered frame equals the 1-process frame bit for bit, and so does the total ray count."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exact, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from oracle import oracle as O
    sc = S.cornell_scene(NativeBuilder(), "mixed", instanced=True)
    ref = O.OraclePathTracer(sc, W, H); ref.set_camera(S.cornell_camera(W, H)); ref.settings.RayDepth = 6
    ref.render(); ref.render()
    want = ref.image()
    for _, full, _ in res:
        assert (full.view(np.uint32) == want.view(np.uint32)).all()
    assert sum(r[2] for r in res) == ref.stats()["rays_traced"]
    # and without the exchange the strips do NOT reproduce the frame at this depth (the test would be vacuous otherwise)
    from idkengine_amd import dist as D
    parts = []
    for r in range(world):
        o = O.OraclePathTracer(sc, W, H); first, count = D.strip_of_rank(H, world, r); o.set_row_range(first, count)
        o.set_camera(S.cornell_camera(W, H)); o.settings.RayDepth = 6; o.render(); o.render(); parts.append(o.image()); o.close()
    assert (np.concatenate(parts).view(np.uint32) != want.view(np.uint32)).any()


class OracleBandRenderer:
    """exact deep paths with the balanced deal: interleaved bands + per-bounce per-band count exchange (dist.make_band_exchange, idkptSetBandExchange)."""

    def __init__(self, scene, cam, world, rank, depth, frames, row_band):
        from oracle import oracle as O
        from idkengine_amd import dist as D
        self.row_band = row_band
        self.pt = O.OraclePathTracer(scene, W, H, row_modulo=world, row_remainder=rank, row_band=row_band)
        ex = D.make_band_exchange(H, row_band)
        self.pt.set_band_exchange(lambda bounce, counts: ex(bounce, counts))
        self.pt.set_camera(cam); self.pt.settings.RayDepth = depth
        self.rows, self.width, self.frames = self.pt.rows, W, frames

    def render(self):
        self.pt.reset_accumulation()
        for _ in range(self.frames):
            self.pt.render()

    def local_image(self):
        return torch.from_numpy(self.pt.image())


def _worker_bands(rank, world, port, q, row_band):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from idkengine_amd import scenes as S, dist as D
    from idkengine_amd.bvh import NativeBuilder
    scene = S.cornell_scene(NativeBuilder(), "mixed", instanced=True) if rank == 0 else None
    scene = D.broadcast_scene(scene, src=0)
    frame = D.ShardedFrame(OracleBandRenderer(scene, S.cornell_camera(W, H), world, rank, depth=6, frames=2, row_band=row_band), W, H)
    frame.render()
    q.put((rank, frame.gather().numpy(), frame.r.pt.stats()["rays_traced"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("row_band", [8, 1])
def test_exact_deep_paths_with_interleaved_bands_world3_gloo(row_band):
    """RayDepth 6 with the BALANCED deal: rows in interleaved bands of 8 (and single interleaved rows), 3 ranks, 2 accumulated samples.  With the per-bounce exchange of the
    per-band alive counts every rank numbers its queue slots as the one-process frame does (NHit seeds from the slot, NHit/compute.glsl:54): the gathered frame and the
    total ray count equal the 1-process ones bit for bit; without the exchange they do not."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bands, args=(r, world, port, q, row_band)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from idkengine_amd import scenes as S, dist as D
    from idkengine_amd.bvh import NativeBuilder
    from oracle import oracle as O
    sc = S.cornell_scene(NativeBuilder(), "mixed", instanced=True)
    ref = O.OraclePathTracer(sc, W, H); ref.set_camera(S.cornell_camera(W, H)); ref.settings.RayDepth = 6
    ref.render(); ref.render()
    want = ref.image()
    for _, full, _ in res:
        assert (full.view(np.uint32) == want.view(np.uint32)).all()
    assert sum(r[2] for r in res) == ref.stats()["rays_traced"]
    rows1 = D.rows_of_rank(H, world, 1, row_band)
    o = O.OraclePathTracer(sc, W, H, row_modulo=world, row_remainder=1, row_band=row_band); o.set_camera(S.cornell_camera(W, H)); o.settings.RayDepth = 6; o.render(); o.render()
    assert (o.image().view(np.uint32) != want[rows1].view(np.uint32)).any()     # (without the exchange the shard differs at this depth: the test is not vacuous)
    o.close(); ref.close()


def test_band_bases_arithmetic():
    """dist.band_bases: image band g belongs to rank g % N as its (g // N)-th band; its base is the number of alive rays in the bands before it."""
    from idkengine_amd import dist as D
    rng = np.random.default_rng(4)
    world, H2, band = 3, 47, 8                                   # 6 bands: ranks own 2 / 2 / 2
    lbs = [D.local_band_count(H2, world, r, band) for r in range(world)]
    assert lbs == [2, 2, 2] and [D.local_band_count(50, 4, r, 8) for r in range(4)] == [2, 2, 2, 1]
    counts = [rng.integers(0, 50, (2, lbs[r])).astype(np.uint32) for r in range(world)]
    bases = D.band_bases(counts, world)
    for k in range(2):
        flat = [int(counts[g % world][k][g // world]) for g in range(sum(lbs))]
        for g in range(sum(lbs)):
            assert int(bases[g % world][k][g // world]) == sum(flat[:g])


def test_rows_of_rank_partition():
    from idkengine_amd.dist import rows_of_rank, band_of_deal
    for h, world in ((1080, 8), (47, 2), (5, 8)):
        rows = sorted(y for r in range(world) for y in rows_of_rank(h, world, r))
        assert rows == list(range(h))
        for band in (2, 8, 16):
            per_rank = [rows_of_rank(h, world, r, band) for r in range(world)]
            assert sorted(y for v in per_rank for y in v) == list(range(h))
            assert all(v == sorted(v) and all((y // band) % world == r for y in v) for r, v in enumerate(per_rank))
        from idkengine_amd.dist import strip_of_rank
        strips = [strip_of_rank(h, world, r) for r in range(world)]
        assert [y for f, n in strips for y in range(f, f + n)] == list(range(h))
    assert band_of_deal(1080, 8) == 8 and band_of_deal(5, 8) == 1 and band_of_deal(47, 2) == 8


def _worker_samples(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from idkengine_amd import scenes as S, dist as D
    from idkengine_amd.bvh import NativeBuilder
    from oracle import oracle as O
    scene = S.cornell_scene(NativeBuilder(), "mixed") if rank == 0 else None
    scene = D.broadcast_scene(scene, src=0)
    pt = O.OraclePathTracer(scene, W, H); pt.set_camera(S.cornell_camera(W, H)); pt.settings.RayDepth = 4
    pt.set_sample_sequence(rank, world)                       # this rank renders the reference's samples rank, rank + world, ...
    for _ in range(3):
        pt.render()
    full = D.combine_accumulations(torch.from_numpy(pt.image()), world).numpy()
    q.put((rank, full, pt.image().copy(), pt.stats()["rays_traced"]))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_parallel_world3_gloo():
    """Sample-parallel mode (idkengine_amd/dist.py): 3 ranks x 3 whole-frame samples with idkptSetSampleSequence(rank, 3).  Every rank's own
    accumulation is the running mean of the reference's samples rank, rank + 3, rank + 6 (checked against a plain 9-sample run), and the
    combined frame is the mean of the three accumulations — the 9-sample accumulation up to binary32 rounding."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_samples, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from idkengine_amd import scenes as S
    from idkengine_amd.bvh import NativeBuilder
    from oracle import oracle as O
    sc = S.cornell_scene(NativeBuilder(), "mixed")
    ref = O.OraclePathTracer(sc, W, H); ref.set_camera(S.cornell_camera(W, H)); ref.settings.RayDepth = 4
    radiance = []
    for _ in range(9):
        ref.render(); radiance.append(ref.rays()["Radiance"].reshape(H, W, 3).copy())
    nine = ref.image()[..., :3]
    f = np.float32
    for rank, full, own, traced in res:
        acc = np.zeros((H, W, 3), f)
        for i in range(3):                                     # FinalDraw's running mean over this rank's three samples
            wgt = f(1.0) / (f(i) + f(1.0))
            acc = (acc * (f(1.0) - wgt) + radiance[rank + 3 * i] * wgt).astype(f)
        assert (own[..., :3].view(np.uint32) == acc.view(np.uint32)).all()
    for _, full, _, _ in res:
        assert (full.view(np.uint32) == res[0][1].view(np.uint32)).all()                      # the same combined frame on every rank
        np.testing.assert_allclose(full[..., :3], nine, rtol=2e-6, atol=1e-6)                  # = the 9-sample accumulation up to rounding
