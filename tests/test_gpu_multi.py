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
from gpu_helpers import bits, gpu_render, oracle_render, assert_equal, _queries, read_device_image  # noqa: E402,F401

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


@pytest.mark.parametrize("band,world", [(8, 3), (4, 2), (16, 2)])
def test_row_bands_equal_the_oracle_and_tile_the_frame(native_builder, oracle_mod, band, world):
    """idkptSetRowBands: rank r of `world` renders the rows y with (y // band) % world == r.  Every rank's context equals the oracle rendering the same
    rows bit for bit (image, ray state, primary hits, alive queue, visit counters; the frame's last band is partial, one tile row straddles two bands when
    band < 8), and at RayDepth 2 the ranks' images tile the one-context frame exactly."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd._lib import IdkPtError
    from idkengine_amd import gputypes as T, dist as D
    sc = S.soup_scene(20000, native_builder, seed=12, extent=3.0); w, h = 150, 93; cam = S.Camera(w, h, position=(0.0, 0.0, 7.5))
    whole = gpu_render(sc, cam, w, h, frames=2, RayDepth=2)
    full = np.zeros((h, w, 4), np.float32)
    for r in range(world):
        for depth, batch in ((2, 2), (4, 1)):
            st = configs.apply_settings(T.Settings.default(), dict(RayDepth=depth))
            p = PathTracer(w, h, settings=st, row_modulo=world, row_remainder=r, row_band=band)
            p.UploadScene(sc); p.SetCamera(cam); p.enable_counters(True); p.enable_primary_hit_capture(True); p.set_max_batch(batch)
            p.Compute(); p.Compute()
            o = oracle_mod.OraclePathTracer(sc, w, h, row_modulo=world, row_remainder=r, row_band=band); o.set_camera(cam)
            configs.apply_settings(o.settings, dict(RayDepth=depth)); o.enable_counters(True); o.render(); o.render()
            rows = D.rows_of_rank(h, world, r, band)
            assert p.rows == o.rows == len(rows) and p.global_rows() == rows
            assert_equal(p, o)
            if depth == 2:
                full[rows] = p.Result
            p.Dispose(); o.close()
    assert (bits(full) == bits(whole.Result)).all()
    whole.Dispose()
    # argument checks: the band height is a power of two; a remainder without rows is refused
    q = PathTracer(16, 16)
    with pytest.raises(IdkPtError, match="power of two"):
        q._check(q._L.idkptSetRowBands(q._ctx, 6, 2, 0))
    with pytest.raises(IdkPtError, match="no row"):
        q._check(q._L.idkptSetRowBands(q._ctx, 8, 4, 3))          # 16 rows = bands 0 and 1: band index 3 does not exist
    q._check(q._L.idkptSetRowBands(q._ctx, 8, 2, 1)); q.row_modulo, q.row_remainder, q.row_band = 2, 1, 8
    assert q.rows == 8
    q.Dispose()


def _device_ids(members):
    import ctypes as C
    from idkengine_amd import _lib
    n = C.c_int32(0); _lib.load().idkptGetDeviceCount(C.byref(n))
    return [i % max(1, n.value) for i in range(members)]      # wraps around the visible GPUs: on a one-GPU box the members share it


@pytest.mark.parametrize("members", [2, 3])
def test_multi_device_context_equals_one_device(native_builder, oracle_mod, members):
    """idkptCreate(deviceCount = N): ONE handle rendering every frame on N members (bands of 8 rows, beyond RayDepth 2 with the per-band alive-count exchange; or strips + the device-side
    exchange when asked for) must return what a one-device context returns, bit for bit: the three images, the per-pixel ray state, primary hits,
    the alive queue, the ray and visit counters — batched or not, several samples per call, a size no member count divides."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc = S.soup_scene(30000, native_builder, seed=21, extent=3.0); w, h = 200, 117; cam = S.Camera(w, h, position=(0.0, 0.0, 8.0))
    ids = _device_ids(members)
    for depth, sort, batch, spp, aov, mode in ((2, 0, 1, 1, 0, 0), (2, 1, 4, 2, 1, 0), (5, 0, 3, 1, 0, 0), (4, 0, 1, 3, 1, 2), (7, 0, 8, 1, 0, 2), (6, 0, 2, 1, 1, 0)):   # mode 0: auto (bands at every depth), 2: strips + device-side exchange
        st = configs.apply_settings(T.Settings.default(), dict(RayDepth=depth, DoRaySorting=sort, SamplesPerPixel=spp, OutputAOVs=aov))
        a = PathTracer(w, h, settings=st, devices=ids); a.SetGroupSharding(mode); b = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), dict(RayDepth=depth, DoRaySorting=sort, SamplesPerPixel=spp, OutputAOVs=aov)))
        for p in (a, b):
            p.UploadScene(sc); p.SetCamera(cam); p.enable_counters(True); p.enable_primary_hit_capture(True); p.set_max_batch(batch)
            for _ in range(3):
                p.Compute()
        assert a.rows == h and a.Result.shape == (h, w, 4)
        assert (bits(a.Result) == bits(b.Result)).all(), (depth, sort, batch)
        if aov:
            assert (bits(a.AlbedoTexture) == bits(b.AlbedoTexture)).all() and (bits(a.NormalTexture) == bits(b.NormalTexture)).all()
        assert a.AccumulatedSamples == b.AccumulatedSamples == 3 * spp
        ta, ia, ba = a.primary_hits(); tb, ib, bb = b.primary_hits()
        assert (ia == ib).all() and (bits(ta) == bits(tb)).all() and (bits(ba) == bits(bb)).all()
        if depth > 2:
            # strips + count exchange: every queue slot, hence every RNG stream, is the one-device one -> the internal state matches too.
            # (Interleaved rows, RayDepth <= 2: the images are exact because radiance never depends on the slot there, but what the LAST
            # bounce leaves behind for a bounce that is not traced — new direction, roulette survivors — is drawn from slot-seeded streams.)
            assert a.rays().tobytes() == b.rays().tobytes(), (depth, sort, batch)
            assert (a.alive_queue() == b.alive_queue()).all()
        sa, sb = a.stats(), b.stats()
        for k in ("rays_traced", "primary_rays", "node_pair_visits", "triangle_tests", "frames"):
            assert sa[k] == sb[k], k
        assert sa["alive_counts"][:depth] == sb["alive_counts"][:depth]
        a.Dispose(); b.Dispose()
    # explicit strips at RayDepth 2: full internal state parity there as well
    a = PathTracer(w, h, devices=ids); b = PathTracer(w, h)
    a.SetGroupSharding(2)
    for p in (a, b):
        p.UploadScene(sc); p.SetCamera(cam); p.RayDepth = 2; p.set_max_batch(2); p.Compute(); p.Compute()
    assert (bits(a.Result) == bits(b.Result)).all() and a.rays().tobytes() == b.rays().tobytes() and (a.alive_queue() == b.alive_queue()).all()
    a.Dispose(); b.Dispose()
    # ... and the oracle agrees with the group directly (deep paths, strips)
    st = configs.apply_settings(T.Settings.default(), dict(RayDepth=5))
    a = PathTracer(w, h, settings=st, devices=ids); a.UploadScene(sc); a.SetCamera(cam); a.set_max_batch(2)
    o = oracle_render(oracle_mod, sc, cam, w, h, frames=2, RayDepth=5)
    a.Compute(); a.Compute()
    assert (bits(a.Result) == bits(o.image(0))).all() and a.rays().tobytes() == o.rays().tobytes()
    a.Dispose(); o.close()


def test_multi_device_context_api_surface(native_builder, oracle_mod):
    """The rest of the boundary on a multi-device context: resize, explicit sharding modes, frame gather on the first device
    (idkptGetImageDevicePtr), sharded ray queries, replicated scene updates (refit), frame ring, and the calls a group refuses."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd._lib import IdkPtError
    from idkengine_amd import gputypes as T
    ids = _device_ids(2)
    sc = S.soup_scene(8000, native_builder, seed=4, extent=2.5, refittable=True); cam = lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 7.0))   # noqa: E731
    a = PathTracer(96, 64, devices=ids); b = PathTracer(96, 64)
    for p in (a, b):
        p.UploadScene(sc); p.SetCamera(cam(96, 64)); p.RayDepth = 3; p.Compute()
    assert (bits(a.Result) == bits(b.Result)).all()
    # resize + explicit strips at depth 2 / explicit rows
    for mode, depth in ((2, 2), (1, 2), (3, 2), (0, 2), (0, 4), (2, 4)):      # strips, rows, bands of 8, auto (= bands), auto beyond RayDepth 2 (= bands + exchange), strips beyond RayDepth 2
        a.SetGroupSharding(mode)
        for p in (a, b):
            p.SetSize(123, 45); p.SetCamera(cam(123, 45)); p.RayDepth = depth; p.set_max_batch(2); p.Compute(); p.Compute()
        assert (bits(a.Result) == bits(b.Result)).all(), (mode, depth)
        # gather on the first device: the pointer addresses the whole frame, ordered on the context's stream
        ptr, nbytes = a.image_device_ptr(0)
        assert nbytes == 123 * 45 * 16
        out = read_device_image(a, ptr, nbytes, (45, 123, 4))
        assert (bits(out) == bits(b.Result)).all(), mode
    # ray queries are cut into one piece per device
    rays = _queries(5001, 3, 3.0)
    assert a.TraceRays(rays).tobytes() == b.TraceRays(rays).tobytes() == oracle_mod.trace_rays(sc, rays).tobytes()
    assert a.TraceRays(rays, any_hit=True).tobytes() == b.TraceRays(rays, any_hit=True).tobytes()
    # replicated scene update: every member refits its own copy
    moved = (sc.vertex_positions + np.float32(0.03) * np.sin(sc.vertex_positions[:, ::-1] * 2.1).astype(np.float32)).astype(np.float32)
    for p in (a, b):
        p.UpdateBuffer(1, moved); p.RefitBlas(0); p.ResetAccumulation(); p.Compute()
    assert (bits(a.Result) == bits(b.Result)).all()
    assert a.DownloadBuffer(6, T.GpuBlasNode, len(sc.blas_nodes)).tobytes() == b.DownloadBuffer(6, T.GpuBlasNode, len(sc.blas_nodes)).tobytes()
    # frame ring
    for p in (a, b):
        p.SetFrameRing(3); p.set_max_batch(3)
        slots = []
        for k in range(3):
            slots.append(p.BeginFrame()); p.SetCamera(S.Camera(123, 45, position=(0.2 * k, 0.0, 7.0))); p.Compute()
        p._slots = slots
    for k in range(3):
        assert (bits(a.FrameResult(a._slots[k])) == bits(b.FrameResult(b._slots[k]))).all(), k
    # what a group does not take
    with pytest.raises(IdkPtError, match="multi-device"):
        a._check(a._L.idkptSetRowSharding(a._ctx, 2, 0))
    with pytest.raises(IdkPtError, match="multi-device"):
        a.set_stream(0)
    with pytest.raises(IdkPtError):
        a.SetSize(64, 1)                              # fewer rows than devices
    a.SetSize(64, 32); a.SetCamera(cam(64, 32)); a.Compute()     # still usable
    assert a.Result.shape == (32, 64, 4)
    a.Dispose(); b.Dispose()


@pytest.mark.parametrize("members", [2, 3])
def test_multi_device_context_without_peer_access(native_builder, members, monkeypatch):
    """A node whose GPUs refuse peer access (or the option "force_no_peer"): every device-to-device copy of the group layer — scene replication,
    the per-bounce alive-count exchange of the strips, the frame gather on device 0 — is staged through pinned host memory instead.  Same bits."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    monkeypatch.setenv("IDKPT_FORCE_NO_PEER", "1")
    sc = S.soup_scene(20000, native_builder, seed=33, extent=3.0, refittable=True); w, h = 160, 101; cam = S.Camera(w, h, position=(0.0, 0.0, 8.0))
    ids = _device_ids(members)
    for depth, batch, mode in ((2, 2, 0), (5, 3, 2), (5, 2, 0)):      # (mode 2: strips, whose per-bounce count exchange is a device-to-device copy; 0: bands, exchanged on the host beyond RayDepth 2)
        st = lambda: configs.apply_settings(T.Settings.default(), dict(RayDepth=depth))   # noqa: E731
        a = PathTracer(w, h, settings=st(), devices=ids); a.SetGroupSharding(mode); b = PathTracer(w, h, settings=st())
        for p in (a, b):
            p.UploadScene(sc); p.SetCamera(cam); p.set_max_batch(batch)
            for _ in range(4):
                p.Compute()
        assert (bits(a.Result) == bits(b.Result)).all(), depth
        if depth > 2:
            assert a.rays().tobytes() == b.rays().tobytes() and (a.alive_queue() == b.alive_queue()).all()
        assert a.stats()["rays_traced"] == b.stats()["rays_traced"]
        ptr, nbytes = a.image_device_ptr(0)                       # staged gather on device 0, twice in a row (the second one waits for the first's readers)
        out = read_device_image(a, ptr, nbytes, (h, w, 4))
        assert (bits(out) == bits(b.Result)).all()
        a.Compute(); b.Compute()
        ptr, nbytes = a.image_device_ptr(0)
        assert (bits(read_device_image(a, ptr, nbytes, (h, w, 4))) == bits(b.Result)).all()
        a.Dispose(); b.Dispose()



@pytest.mark.parametrize("batch,band,on_device", [(1, 8, False), (3, 8, False), (2, 1, False), (3, 8, True)])
def test_exact_deep_paths_with_interleaved_bands(native_builder, oracle_mod, batch, band, on_device):
    """idkptSetRowBands / idkptSetRowSharding + idkptSetBandExchange: three contexts with the BALANCED deal (interleaved bands of 8 rows, or single interleaved rows), driven by
    three host threads in lockstep, the exchange function running dist.band_bases over everybody's per-(sample, band) counts: the single-context frame bit for bit at
    RayDepth 6 — image, ray state, total ray count — also with several accumulated samples per batch.  And one banded context against the oracle under an arbitrary
    exchange function: both number their slots the same way."""
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
            mine = D.band_bases([board[(bounce, r)] for r in range(world)], world)[rank]
            barrier.wait(timeout=60)                     # nobody overwrites the board before everybody has read it
            return mine
        return fn

    pts, errs = [], []
    for r in range(world):
        p = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov), row_modulo=world, row_remainder=r, row_band=band); p.UploadScene(sc); p.SetCamera(cam)
        if on_device:
            # idkptSetBandExchangeDevice: counts and bases are device buffers and the callback may merely enqueue on the context's stream (dist.make_band_exchange_device does,
            # with an all-gather); this one drains the stream and goes through the same board, which a callback is free to do
            import torch
            from idkengine_amd.dist import _DevArray

            def dev_fn(bounce, samples, bands, d_counts, d_bases, stream, _fn=exchange_for(r)):
                torch.cuda.ExternalStream(stream).synchronize()
                counts = torch.as_tensor(_DevArray(d_counts, (samples, bands), "<i4"), device="cuda:0").cpu().numpy().astype(np.uint32)
                bases = np.asarray(_fn(bounce, counts), np.uint32).astype(np.int64).astype(np.int32)
                torch.as_tensor(_DevArray(d_bases, (samples, bands), "<i4"), device="cuda:0").copy_(torch.from_numpy(bases.reshape(samples, bands)))
                torch.cuda.synchronize()
            p.SetBandExchangeDevice(dev_fn)
        else:
            p.SetBandExchange(exchange_for(r))
        p.set_max_batch(batch)
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
    want_rays2 = want_rays.reshape(h, w)
    for r, p in enumerate(pts):
        rows = D.rows_of_rank(h, world, r, band)
        assert (bits(p.Result) == bits(want[rows])).all()
        assert p.rays().reshape(len(rows), w).tobytes() == want_rays2[rows].tobytes()
    assert sum(p.stats()["rays_traced"] for p in pts) == want_count
    # control: a shard without the exchange differs at this depth
    q = PathTracer(w, h, settings=configs.apply_settings(T.Settings.default(), ov), row_modulo=world, row_remainder=1, row_band=band); q.UploadScene(sc); q.SetCamera(cam)
    for _ in range(frames):
        q.Compute()
    assert (bits(q.Result) != bits(want[D.rows_of_rank(h, world, 1, band)])).any()
    # one banded context == the oracle under the same (arbitrary) exchange function
    fake = lambda bounce, counts: (np.arange(counts.size, dtype=np.uint32).reshape(counts.shape) * np.uint32(977) + np.uint32(31 * bounce))   # noqa: E731
    q.SetBandExchange(fake); q.ResetAccumulation(); q.set_max_batch(1)
    o = oracle_mod.OraclePathTracer(sc, w, h, row_modulo=world, row_remainder=1, row_band=band); o.set_camera(cam); configs.apply_settings(o.settings, ov)
    o.set_band_exchange(fake)
    for _ in range(2):
        q.Compute(); o.render()
    assert (bits(q.Result) == bits(o.image(0))).all() and q.rays().tobytes() == o.rays().tobytes() and (q.alive_queue() == o.alive_queue()).all()
    o.close()
    for p in pts + [one, q]:
        p.Dispose()


@pytest.mark.parametrize("members", [2, 3])
def test_multi_device_context_interleaved_rows_are_exact_at_any_depth(native_builder, members):
    """idkptSetGroupSharding(ROWS / BANDS) beyond RayDepth 2: the members' batches are enqueued by one host thread each and meet at every bounce to exchange their
    per-(sample, band) alive counts (the single-context idkptSetBandExchange inside the group): image, ray state, alive queue and counters equal the one-device
    context's bit for bit — batched, several samples per call, a height no member count divides; with ray sorting on the frame still renders (statistical beyond the first bounce)."""
    from idkengine_amd.pathtracer import PathTracer
    from idkengine_amd import gputypes as T
    sc = S.soup_scene(30000, native_builder, seed=22, extent=3.0); w, h = 200, 117; cam = S.Camera(w, h, position=(0.0, 0.0, 8.0))
    ids = _device_ids(members)
    for mode, depth, batch, spp in ((3, 5, 1, 1), (3, 6, 3, 1), (1, 4, 2, 2), (3, 3, 8, 1)):
        mk = lambda: configs.apply_settings(T.Settings.default(), dict(RayDepth=depth, SamplesPerPixel=spp))   # noqa: E731
        a = PathTracer(w, h, settings=mk(), devices=ids); a.SetGroupSharding(mode); b = PathTracer(w, h, settings=mk())
        for p in (a, b):
            p.UploadScene(sc); p.SetCamera(cam); p.set_max_batch(batch)
            for _ in range(3):
                p.Compute()
        assert (bits(a.Result) == bits(b.Result)).all(), (mode, depth, batch)
        assert a.rays().tobytes() == b.rays().tobytes(), (mode, depth, batch)
        assert (a.alive_queue() == b.alive_queue()).all()
        assert a.stats()["rays_traced"] == b.stats()["rays_traced"]
        a.RayDepth = 2; b.RayDepth = 2                      # back to the pipelined schedule on the same context
        for p in (a, b):
            p.Compute(); p.Compute()
        assert (bits(a.Result) == bits(b.Result)).all()
        a.Dispose(); b.Dispose()
    st = configs.apply_settings(T.Settings.default(), dict(RayDepth=5, DoRaySorting=1))
    a = PathTracer(w, h, settings=st, devices=ids); a.SetGroupSharding(3); a.UploadScene(sc); a.SetCamera(cam); a.Compute(); a.Compute()
    assert np.isfinite(a.Result).all() and a.AccumulatedSamples == 2
    a.Dispose()
