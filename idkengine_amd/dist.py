"""Multi-GPU framebuffer sharding: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in
CPU tests).  The reference is single-GPU; this is the MI355X-native extension north_star asks for:

  * scene/BVH replicated: rank 0 builds, every array is broadcast once per scene (RCCL broadcast, ~100 MB for 1M tris);
  * pixels are independent given the scene (FirstHit/compute.glsl:81-97), so image rows are dealt round-robin in bands of 8 rows:
    rank r renders rows y with (y // 8) % world == r (interleaving balances sky rows against geometry rows; bands of 8 keep every
    8x8 pixel tile — the unit a wave of the ray generation and of the primary traversal works on — whole on every rank);
  * per frame the only exchange is the all-gather of the finished row shards (4.1 MB per GPU at 1080p on 8 GPUs);
  * no data-path collective inside a frame.  At RayDepth 2 (the headline metric) radiance is independent of the queue
    slot, so N-GPU output == 1-GPU output bit-for-bit.  For deeper paths the NHit RNG seeds depend on the
    queue slot: `exact_deep_paths` adds an all-gather of the per-(sample, band) alive counts per bounce (idkptSetBandExchange, make_band_exchange: the interleaved
    deal stays) — or, with "strips", switches to contiguous strips + per-sample counts (idkptSetRowRange / idkptSetBounceExchange, make_count_exchange) — which
    restores bit-exact parity at any depth (sorting off);
    without it parity beyond depth 2 is statistical.
"""
import numpy as np
import torch
import torch.distributed as dist

_SCENE_FIELDS = ["blas_nodes", "blas_triangles", "blas_descs", "blas_instances", "tlas_nodes", "blas_parent_indices", "blas_leaf_indices",
                 "vertex_positions", "vertices", "meshes", "materials", "mesh_transforms", "lights"]


ROW_BAND = 8      # rows per band of the default (interleaved) deal: idkptSetRowBands(8, world, rank)


def rows_of_rank(height, world, rank, band=1):
    """Image rows of rank `rank` when bands of `band` rows are dealt round-robin (band = 1: single rows), in increasing y."""
    if band <= 1:
        return list(range(rank, height, world))
    return [y for y in range(height) if (y // band) % world == rank]


def band_of_deal(height, world, band=ROW_BAND):
    """The band height the default deal uses: `band` rows, or single rows when the image has fewer bands than ranks."""
    return band if (height + band - 1) // band >= world else 1


def strip_of_rank(height, world, rank):
    """Contiguous strip (first_row, row_count) of rank `rank`: rows are dealt as evenly as possible, earlier ranks get the extra row."""
    base, extra = divmod(height, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def make_count_exchange(group=None):
    """Host side of idkptSetBounceExchange for one process per strip: all-gather of the per-sample alive counts (a few uint32 per
    bounce and batch, on the CPU: use a gloo group next to the RCCL one), base[k] = sum of count[k] over the ranks above.
    Ranks own strips in rank order (strip_of_rank), so "above" = lower rank."""
    rank = dist.get_rank(group); world = dist.get_world_size(group)

    def exchange(bounce, local_counts):
        mine = torch.from_numpy(np.ascontiguousarray(local_counts, np.int64).astype(np.int64))
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        base = np.zeros(len(local_counts), np.int64)
        for r in range(rank):
            base += parts[r].numpy()
        return base.astype(np.uint32)
    return exchange


def local_band_count(height, world, rank, band):
    """Bands (complete or the image's last, partial one) that the round-robin deal gives to `rank`."""
    total = (height + band - 1) // band
    return len(range(rank, total, world))


def band_bases(counts_by_rank, world):
    """counts_by_rank[r][k][b] = alive rays of sample k in rank r's b-th band (image band b * world + r).  -> bases[r][k][b] = alive rays of sample k in all image bands
    before that band (the arithmetic of idkptSetBandExchange's host side; also used by the tests)."""
    samples = len(counts_by_rank[0])
    total = sum(len(c[0]) if len(c) else 0 for c in counts_by_rank)
    bases = [np.zeros_like(np.asarray(c, np.uint32)) for c in counts_by_rank]
    for k in range(samples):
        cum = 0
        for gb in range(total):
            r, b = gb % world, gb // world
            bases[r][k][b] = cum
            cum += int(counts_by_rank[r][k][b])
    return bases


def make_band_exchange(height, band, group=None):
    """Host side of idkptSetBandExchange for one process per rank with rows dealt in bands of `band` rows: all-gather of the per-(sample, band) alive counts (a few hundred
    uint32 per bounce and batch, on the CPU: a gloo group next to the RCCL one), then the running sum over the image's bands."""
    rank = dist.get_rank(group); world = dist.get_world_size(group)
    lbs = [local_band_count(height, world, r, band) for r in range(world)]
    lb_max = max(lbs)

    def exchange(bounce, local_counts):
        samples = local_counts.shape[0]
        mine = torch.zeros((samples, lb_max), dtype=torch.int64)
        mine[:, :lbs[rank]] = torch.from_numpy(np.ascontiguousarray(local_counts, np.int64))
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        by_rank = [parts[r].numpy()[:, :lbs[r]].astype(np.uint32) for r in range(world)]
        return band_bases(by_rank, world)[rank]
    return exchange


def broadcast_scene(scene, src=0, device=None, group=None):
    """Replicates a gputypes.Scene from rank `src` to all ranks (returns the scene on every rank).
    On non-source ranks `scene` may be None.  Arrays travel as raw bytes (the structs are the ABI payload)."""
    from . import gputypes as T
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    out = scene if rank == src else T.Scene()
    template = T.Scene()
    meta = None
    if rank == src:
        meta = []
        for f in _SCENE_FIELDS:
            a = np.ascontiguousarray(getattr(scene, f))
            meta.append((a.shape, a.nbytes))
        sky = None if scene.sky_faces is None else np.ascontiguousarray(scene.sky_faces, np.float32)
        meta.append((None if sky is None else sky.shape, 0 if sky is None else sky.nbytes))
        imgs = [T.TextureImage.of(t) for t in scene.textures]
        meta.append((len(imgs), [(t.data.shape, t.data.dtype.str, t.wrap_s, t.wrap_t, t.mag_filter, t.format) for t in imgs], scene.blas_stack_size))
    box = [meta]
    dist.broadcast_object_list(box, src=src, group=group)
    meta = box[0]

    def bcast_bytes(arr, nbytes):
        if nbytes == 0:
            return np.zeros(0, np.uint8)
        if rank == src:
            t = torch.from_numpy(np.frombuffer(np.ascontiguousarray(arr).tobytes(), np.uint8).copy()).to(dev)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=src, group=group)
        return t.cpu().numpy()

    for i, f in enumerate(_SCENE_FIELDS):
        shape, nbytes = meta[i]
        raw = bcast_bytes(getattr(scene, f) if rank == src else None, nbytes)
        if rank != src:
            dt = getattr(template, f).dtype
            setattr(out, f, np.frombuffer(raw.tobytes(), dt).reshape(shape).copy())
    sky_shape, sky_bytes = meta[len(_SCENE_FIELDS)]
    raw = bcast_bytes(scene.sky_faces if rank == src else None, sky_bytes)
    if rank != src:
        out.sky_faces = None if sky_shape is None else np.frombuffer(raw.tobytes(), np.float32).reshape(sky_shape).copy()
    ntex, tex_shapes, stack = meta[len(_SCENE_FIELDS) + 1]
    texs = []
    for k in range(ntex):
        shape, dt, ws, wt, mf, fmt = tex_shapes[k]
        nb = int(np.prod(shape)) * np.dtype(dt).itemsize
        raw = bcast_bytes(T.TextureImage.of(scene.textures[k]).data if rank == src else None, nb)
        texs.append(T.TextureImage(np.frombuffer(raw.tobytes(), np.dtype(dt)).reshape(shape).copy(), ws, wt, mf, srgb=(fmt == T.IDKPT_TEXFMT_SRGB8_A8)))      # texels AND sampler state travel
    if rank != src:
        out.textures = texs
        out.blas_stack_size = stack
    return out


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can alias the library's device image without a copy."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def make_band_exchange_device(height, band, device, stream, group=None):
    """Host side of idkptSetBandExchangeDevice for one process per rank: nothing but ENQUEUED work on `stream` (the stream the context renders on) — an all-gather of the
    per-(sample, band) alive counts (RCCL under the nccl backend) and the running sum over the image's bands as a handful of torch ops — so the exact deep-path mode never
    synchronises the host.  Image band g = b * world + r is rank r's b-th band: [samples, bands, world] flattened IS the image order."""
    rank = dist.get_rank(group); world = dist.get_world_size(group)
    lbs = [local_band_count(height, world, r, band) for r in range(world)]
    lb_max = max(lbs)

    def exchange(bounce, samples, bands, d_counts, d_bases, hip_stream):
        assert bands == lbs[rank] and int(hip_stream) == int(stream.cuda_stream)
        with torch.cuda.stream(stream):
            counts = torch.as_tensor(_DevArray(d_counts, (samples, bands), "<i4"), device=device)       # (alive counts fit 31 bits)
            mine = torch.zeros((samples, lb_max), dtype=torch.int32, device=device); mine[:, :bands] = counts
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            img = torch.stack(parts, dim=2).reshape(samples, lb_max * world).to(torch.int64)          # image-band order; bands a rank does not own count 0
            excl = torch.cumsum(img, dim=1) - img
            mine_bases = excl.reshape(samples, lb_max, world)[:, :bands, rank].to(torch.int32).contiguous()
            torch.as_tensor(_DevArray(d_bases, (samples, bands), "<i4"), device=device).copy_(mine_bases)
    return exchange


class GpuShardRenderer:
    """Adapter: idkengine_amd.PathTracer rendering this rank's rows; exposes the local RGBA rows as a torch tensor that
    aliases the library's device image (no host copy before the RCCL all-gather)."""

    def __init__(self, width, height, world, rank, device_index, exact_deep_paths=False, control_group=None, row_band=None):
        """Rows are dealt in interleaved bands of `row_band` rows (None: 8, or single rows on images with fewer bands than ranks): exact at RayDepth 2 as is.
        exact_deep_paths: + a per-bounce exchange of the per-band alive counts, so that N-GPU output equals 1-GPU output bit for bit at ANY RayDepth (sorting off) with the
        balanced deal — enqueued on the render stream over the default group (RCCL all-gather + prefix sum, idkptSetBandExchangeDevice: no host synchronisation), or, when a
        `control_group` (CPU / gloo) is given, on the host (idkptSetBandExchange: one stream synchronisation per bounce); exact_deep_paths="strips": contiguous strips + per-sample counts (round 2's mode:
        exact as well, but 8 strips of the headline camera scale 4.2x where 8 interleaved shards scale 7.6x)."""
        from .pathtracer import PathTracer
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        strips = exact_deep_paths == "strips"
        self.exact = strips                                              # (ShardedFrame: the rows of this renderer are a contiguous strip)
        self.row_band = 1 if strips or world == 1 else (band_of_deal(height, world) if row_band is None else int(row_band))
        if strips:
            self.pt = PathTracer(width, height, device=device_index)
            first, count = strip_of_rank(height, world, rank)
            self.pt.SetRowRange(first, count)
            self.pt.SetBounceExchange(make_count_exchange(control_group))
        else:
            self.pt = PathTracer(width, height, device=device_index, row_modulo=world, row_remainder=rank, row_band=self.row_band)
        # render on a dedicated torch stream and issue the collectives under it: RCCL work is then ordered after the
        # kernels that produce the image, and later renders are ordered after the collective that reads it
        self.stream = torch.cuda.Stream(device=self.device)
        self.pt.set_stream(self.stream.cuda_stream)
        if exact_deep_paths and not strips and world > 1:
            if control_group is not None:      # a CPU (gloo) control group: host-side exchange, one stream synchronisation per bounce
                self.pt.SetBandExchange(make_band_exchange(height, self.row_band, control_group))
            else:                              # the default group on the render stream: RCCL all-gather + prefix sum enqueued, no host synchronisation
                self.pt.SetBandExchangeDevice(make_band_exchange_device(height, self.row_band, self.device, self.stream))
        self.width, self.height, self.rows = width, height, self.pt.rows

    def upload_scene(self, scene):
        self.pt.UploadScene(scene)

    def set_camera(self, cam):
        self.pt.SetCamera(cam)

    def render(self):
        self.pt.ResetAccumulation()
        self.pt.Compute()

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def local_frames(self, first_slot, count):
        """The finished images of `count` consecutive ring slots as one (count, rows, W, 4) tensor aliasing the library's memory
        (slots are contiguous; idkptGetFrameDevicePtr)."""
        self.pt.flush()
        ptr, nbytes = self.pt.frame_device_ptr(first_slot, 0)
        self.pt.frame_device_ptr(first_slot + count - 1, 0)       # raises if the group runs past the end of the ring (slots must be consecutive)
        return torch.as_tensor(_DevArray(ptr, (count, self.rows, self.width, 4)), device=self.device)

    def local_image(self):
        self.pt.flush()   # launch whatever the library still defers; the collective that follows is stream-ordered behind it
        ptr, nbytes = self.pt.image_device_ptr(0)
        assert nbytes == self.rows * self.width * 16
        return torch.as_tensor(_DevArray(ptr, (self.rows, self.width, 4)), device=self.device)


class ShardedFrame:
    """Sharding of one frame over the ranks of a process group (interleaved bands / rows, or strips) + all-gather of the shards."""

    def __init__(self, renderer, width, height, group=None):
        self.r = renderer
        self.group = group
        self.world = dist.get_world_size(group); self.rank = dist.get_rank(group)
        self.width, self.height = width, height
        self.strips = bool(getattr(renderer, "exact", False))           # contiguous strips (exact deep paths) or interleaved bands / rows
        self.band = int(getattr(renderer, "row_band", 1))
        if self.strips:
            self.rank_rows = [list(range(f, f + n)) for f, n in (strip_of_rank(height, self.world, r) for r in range(self.world))]
        else:
            self.rank_rows = [rows_of_rank(height, self.world, r, self.band) for r in range(self.world)]
        self.max_rows = max(len(v) for v in self.rank_rows)
        assert renderer.rows == len(self.rank_rows[self.rank])
        self._row_index = None

    def render(self):
        self.r.render()

    def gather(self):
        """All-gather of the row shards; returns the full (H, W, 4) image on every rank (torch tensor on the renderer's device)."""
        import contextlib
        ctx = self.r.stream_ctx() if hasattr(self.r, "stream_ctx") else contextlib.nullcontext()
        with ctx:
            return self._gather()

    def gather_frames(self, first_slot, count):
        """Frame ring: all-gather of the row shards of `count` finished frames (consecutive ring slots) in one collective; returns
        (count, H, W, 4) on every rank.  Every frame that was rendered is exchanged."""
        import contextlib
        ctx = self.r.stream_ctx() if hasattr(self.r, "stream_ctx") else contextlib.nullcontext()
        with ctx:
            local = self.r.local_frames(first_slot, count)
            if local.shape[1] < self.max_rows:
                pad = torch.zeros((count, self.max_rows - local.shape[1]) + tuple(local.shape[2:]), dtype=local.dtype, device=local.device)
                local = torch.cat([local, pad], dim=1)
            parts = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(parts, local.contiguous().clone(), group=self.group)   # clone: the collective works on torch-owned memory, the ring slots are free again
            full = torch.empty((count, self.height, self.width, 4), dtype=local.dtype, device=local.device)
            for r in range(self.world):
                self._place(full, parts[r], r, 1)
            return full

    def _place(self, full, part, r, dim):
        """rows of rank r (the first len(rank_rows[r]) of `part` along `dim`) -> their image rows in `full`"""
        rows = self.rank_rows[r]; n = len(rows)
        sel = (slice(None),) * dim
        if n == 0:
            return
        if self.strips:
            full[sel + (slice(rows[0], rows[0] + n),)] = part[sel + (slice(0, n),)]
        elif self.band <= 1:
            full[sel + (slice(r, None, self.world),)] = part[sel + (slice(0, n),)]
        else:
            if self._row_index is None:
                self._row_index = {}
            idx = self._row_index.get((r, str(full.device)))
            if idx is None:
                idx = torch.tensor(rows, dtype=torch.long, device=full.device); self._row_index[(r, str(full.device))] = idx
            full.index_copy_(dim, idx, part[sel + (slice(0, n),)])

    def _gather(self):
        local = self.r.local_image()
        if local.shape[0] < self.max_rows:  # ranks with one row less pad (all_gather needs equal shapes)
            pad = torch.zeros((self.max_rows - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], dim=0)
        parts = [torch.empty_like(local) for _ in range(self.world)]
        dist.all_gather(parts, local.contiguous().clone(), group=self.group)       # clone (4 MB): torch-owned send buffer, decoupled from the library's image
        full = torch.empty((self.height, self.width, 4), dtype=local.dtype, device=local.device)
        for r in range(self.world):
            self._place(full, parts[r], r, 0)
        return full


# ---------------------------------------------------------------------------------------------------------------------------
# Sample-parallel rendering: no sharding of the frame at all.  Rank r of N renders the WHOLE frame for the reference's samples
# r, r + N, r + 2N, ... (idkptSetSampleSequence(r, N): their RNG streams, own running mean), so a displayed frame of N * K samples
# costs every GPU K full-frame passes — the same launches, the same cache behaviour as one GPU alone — and ONE all-reduce of the
# image.  Nothing is exchanged inside a frame.  The displayed image is the mean of the N accumulations: every reference sample
# 0 .. N*K-1 enters exactly once with weight 1 / (N*K); the summation order differs from a single GPU's running mean, so it equals
# the 1-GPU accumulation of the same samples up to binary32 rounding (each rank's own accumulation IS bit-exact, tests/test_gpu_samples.py).
# This is the throughput mode (weak scaling: per-GPU work is fixed); row sharding above is the latency mode (strong scaling, bit-exact).

def combine_accumulations(local, world, group=None):
    """Mean over the ranks of their accumulated images: all-reduce (sum) of a torch-owned copy, then * (1 / world)."""
    buf = local.contiguous().clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf.mul_(1.0 / float(world))


class SampleParallelRenderer:
    """idkengine_amd.PathTracer on this rank's GPU rendering whole frames for the sample indices rank, rank + world, ..."""

    def __init__(self, width, height, world, rank, device_index):
        from .pathtracer import PathTracer
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        self.pt = PathTracer(width, height, device=device_index)
        self.pt.SetSampleSequence(rank, world)
        self.stream = torch.cuda.Stream(device=self.device)
        self.pt.set_stream(self.stream.cuda_stream)
        self.width, self.height, self.rows, self.world = width, height, height, world

    def upload_scene(self, scene):
        self.pt.UploadScene(scene)

    def set_camera(self, cam):
        self.pt.SetCamera(cam)

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def local_image(self):
        self.pt.flush()
        ptr, nbytes = self.pt.image_device_ptr(0)
        assert nbytes == self.height * self.width * 16
        return torch.as_tensor(_DevArray(ptr, (self.height, self.width, 4)), device=self.device)


class SampleParallelFrame:
    """The displayed frame of a sample-parallel group: gather() = mean of the ranks' accumulations, on every rank."""

    def __init__(self, renderer, group=None):
        self.r, self.group = renderer, group
        self.world = dist.get_world_size(group)

    def gather(self):
        with self.r.stream_ctx():
            return combine_accumulations(self.r.local_image(), self.world, self.group)
