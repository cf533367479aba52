"""PathTracer — host-side mirror of IDKEngine's `class PathTracer` (Source/Render/PathTracer.cs:10-365) over the
C-ABI of libidkpt.so.  Same property names, same reset-accumulation behaviour, same lifecycle
(ctor(width,height,settings) / Compute() / SetSize / ResetAccumulation / Dispose); the implicit GL bindings the
reference's shaders read become the explicit UploadScene / SetCamera calls.  The C# drop-in that binds the same
entry points through LibraryImport is shown in INTEGRATION.md.

No CPU fallback: construction raises when libidkpt.so or a GPU is missing.
"""
import ctypes as C
import os
import numpy as np
from . import _lib
from . import gputypes as T
from ._lib import IdkPtError


# idkptSetDeveloperOption names; IDKPT_<NAME> in the environment is forwarded when a PathTracer is created (test / tuning hooks only)
_OPTION_NAMES = ("force_generic", "no_tile_cull", "no_lean_primary", "leaf_min", "grab_unit_log2", "grab_fixed", "lds_pad", "trace_waves", "grid_hint", "grid_rays_x4", "grid_mid_waves", "defer_last", "split", "split_donor", "split_peek", "group_threads", "query_scheduler", "split_scatter", "fused", "fused_shade_min", "leaf_pool", "pool_min", "adv_min",
                 "wide", "wide_cap", "wide_count", "packet", "packet_min_live", "packet_waves", "inst_tlas", "inst_tlas_overlap", "inst_braid", "inst_unify", "inst_unify_radius", "uni_refill", "inst_general", "pair_nodes", "inst_sieve", "inst_sieve_overlap", "gen_pixel_major", "gen_group_max", "bounce_pixel_major", "transport", "bvh_timing", "bvh_small", "bvh_stackopt_host", "force_no_peer", "trace_variant")


class PathTracer:
    def __init__(self, width, height, settings=None, device=0, row_modulo=1, row_remainder=0, devices=None, row_band=1):
        """devices: list of HIP device ids for ONE context that renders on several GPUs (idkptCreate(deviceCount = N)); an id may repeat
        (two members on one GPU).  Otherwise one device (`device`), optionally one row shard of a process-per-GPU run (row_modulo/remainder;
        row_band: rows are dealt in bands of that many rows, idkptSetRowBands)."""
        self._L = _lib.load()
        n = C.c_int32(0)
        self._L.idkptGetDeviceCount(C.byref(n))
        if n.value <= 0:
            raise IdkPtError("no HIP device visible: the path tracer has no CPU fallback")
        ctx = C.c_void_p()
        ids = list(devices) if devices else [device]
        self.device_count = len(ids)
        if self.device_count > 1 and (row_modulo, row_remainder) != (1, 0):
            raise IdkPtError("a multi-device context deals its rows itself")
        dev = (C.c_int32 * len(ids))(*ids)
        rc = self._L.idkptCreate(len(ids), dev, C.byref(ctx))
        if rc != 0:
            raise IdkPtError(f"idkptCreate failed with status {rc}")
        self._ctx = ctx
        # tuning / test options: the library reads no environment; this host mirror forwards IDKPT_<NAME> (tests, tools) to idkptSetDeveloperOption
        for name in _OPTION_NAMES:
            v = os.environ.get("IDKPT_" + name.upper())
            if v is not None and v != "":
                self.set_option(name, int(v))
        self._settings = settings if settings is not None else T.Settings.default()
        self._cached_ray_depth = self._settings.RayDepth
        self._scene = None
        self.width, self.height = width, height
        self.row_modulo, self.row_remainder, self.row_band = row_modulo, row_remainder, (int(row_band) if row_modulo > 1 else 1)
        self._row_limit = None
        if self.device_count == 1:
            if self.row_band > 1:
                self._check(self._L.idkptSetRowBands(ctx, self.row_band, row_modulo, row_remainder))
            else:
                self._check(self._L.idkptSetRowSharding(ctx, row_modulo, row_remainder))
        self._check(self._L.idkptSetSize(ctx, width, height))
        self._push_settings()

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        err = getattr(self, "_callback_error", None)
        if err is not None:            # an exchange callback raised while the library was inside a launch (ctypes cannot propagate it): re-raise here
            self._callback_error = None
            raise err
        if rc != 0:
            msg = C.c_char_p()
            self._L.idkptGetLastError(self._ctx, C.byref(msg))
            raise IdkPtError(f"idkpt status {rc}: {(msg.value or b'').decode()}")

    def _push_settings(self):
        self._check(self._L.idkptSetSettings(self._ctx, C.addressof(self._settings)))

    def global_rows(self):
        """image rows of this context's local rows, in local order"""
        b = self.row_band
        ys = [y for y in range(self.height) if (y // b) % self.row_modulo == self.row_remainder] if self.row_modulo > 1 else list(range(self.row_remainder, self.height))
        return ys if self._row_limit is None else ys[:self._row_limit]

    @property
    def rows(self):
        return len(self.global_rows())

    # ------------------------------------------------------------------ PathTracer.cs public surface
    def _prop(name, gpu=False):  # noqa: N805
        def get(self):
            return getattr(self._settings.Gpu if gpu else self._settings, name)

        def set_(self, v):
            setattr(self._settings.Gpu if gpu else self._settings, name, type(get(self))(v))
            self._push_settings()
        return property(get, set_)

    SamplesPerPixel = _prop("SamplesPerPixel")        # PathTracer.cs:12
    RayDepth = _prop("RayDepth")                      # :16-25 (resets accumulation)
    FocalLength = _prop("FocalLength", gpu=True)      # :39-48
    LenseRadius = _prop("LenseRadius", gpu=True)      # :50-59
    DoTraceLights = _prop("DoTraceLights", gpu=True)  # :83-92
    DoRussianRoulette = _prop("DoRussianRoulette", gpu=True)  # :94-102
    DoRaySorting = _prop("DoRaySorting")              # :104-114
    OutputAOVs = _prop("OutputAOVs")                  # :116-125
    UseTlas = _prop("UseTlas")                        # BVH.GpuUseTlas (Bvh/BVH.cs:16-26)
    BlasStackSize = _prop("BlasStackSize")            # BVH.BlasStackSize (Bvh/BVH.cs:28-45)

    @property
    def DoDebugBVHTraversal(self):                    # :61-81
        return self._settings.Gpu.DoDebugBVHTraversal == 1

    @DoDebugBVHTraversal.setter
    def DoDebugBVHTraversal(self, value):
        self._settings.Gpu.DoDebugBVHTraversal = 1 if value else 0
        if value:
            self._cached_ray_depth = self._settings.RayDepth
            self._settings.RayDepth = 1
        else:
            self._settings.RayDepth = self._cached_ray_depth
        self._push_settings()
        self.ResetAccumulation()

    @property
    def AccumulatedSamples(self):                     # :27-37
        v = C.c_uint32()
        self._check(self._L.idkptGetAccumulatedSamples(self._ctx, C.byref(v)))
        return v.value

    def Compute(self):                                # :214-271
        self._check(self._L.idkptRender(self._ctx))

    def SetSize(self, width, height):                 # :299-332
        self.width, self.height = width, height
        self._check(self._L.idkptSetSize(self._ctx, width, height))

    def SetSampleSequence(self, first, stride):
        """Sample-parallel rendering (idkptSetSampleSequence): this context renders the reference's samples first, first + stride, ..."""
        self._check(self._L.idkptSetSampleSequence(self._ctx, int(first), int(stride)))

    def ResetAccumulation(self):                      # :334-337
        self._check(self._L.idkptResetAccumulation(self._ctx))

    def Dispose(self):                                # :344-365
        if self._ctx:
            self._L.idkptDestroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    @property
    def Result(self):                                 # :143 (Texture.Download equivalent)
        return self.download(0)

    @property
    def AlbedoTexture(self):                          # :167
        return self.download(1)

    @property
    def NormalTexture(self):                          # :168
        return self.download(2)

    # ------------------------------------------------------------------ explicit inputs (implicit GL bindings in the reference)
    def UploadScene(self, scene):
        d, keep = scene.desc()
        self._check(self._L.idkptUploadScene(self._ctx, C.addressof(d)))
        del keep
        self._scene = scene
        if self._settings.BlasStackSize == 0 and scene.blas_stack_size:
            pass  # library derives the LDS stack depth from BlasDescs when BlasStackSize == 0

    def SetCamera(self, cam):
        ip = np.ascontiguousarray(cam.inv_projection, np.float32); iv = np.ascontiguousarray(cam.inv_view, np.float32); vp = np.ascontiguousarray(cam.position, np.float32)
        self._check(self._L.idkptSetPerFrame(self._ctx, ip.ctypes.data, iv.ctypes.data, vp.ctypes.data))

    def UpdateBuffer(self, which, array, offset_bytes=0):
        a = np.ascontiguousarray(array)
        self._check(self._L.idkptUpdateBuffer(self._ctx, which, offset_bytes, a.nbytes, a.ctypes.data))

    def DownloadBuffer(self, which, dtype, count, offset_bytes=0):
        out = np.zeros(count, dtype)
        self._check(self._L.idkptDownloadBuffer(self._ctx, which, offset_bytes, out.nbytes, out.ctypes.data))
        return out

    def BuildTlas(self, nodes):
        n = np.ascontiguousarray(nodes)
        self._check(self._L.idkptBuildTlas(self._ctx, n.ctypes.data, len(n)))

    def BuildTlasOnDevice(self, search_radius=15):
        """BVH.TlasBuild on the device from the resident BLAS roots / instances / mesh transforms (Bvh/BVH.cs:278-298)."""
        self._check(self._L.idkptBuildTlasOnDevice(self._ctx, search_radius))

    def TraceRays(self, rays, any_hit=False, trace_lights=False):
        """Batched TraceRay / TraceRayAny calls (Shaders/include/BVHIntersect.glsl:183-411) with per-ray maxDist.
        rays: array of gputypes.RayQuery; returns an array of gputypes.RayHit."""
        from . import gputypes as T
        if hasattr(rays, "data_ptr") and getattr(rays, "is_cuda", False):
            # rays already resident on this context's device (a torch tensor of count x 32 bytes, any dtype): the device entry point, no PCIe round trip (2 885 vs 252 Mray/s,
            # profiles/r04_queries.md) — the hits come back as a device tensor (count x 8 float32 words = gputypes.RayHit records), asynchronously in the context's stream order
            import torch
            count = rays.numel() * rays.element_size() // T.RayQuery.itemsize
            hit_bytes = T.RayHit.itemsize
            hits = torch.empty(count * hit_bytes // 4, dtype=torch.float32, device=rays.device)
            self.TraceRaysDevice(rays.data_ptr(), hits.data_ptr(), count, any_hit=any_hit, trace_lights=trace_lights)
            self.synchronize()
            return hits.view(count, hit_bytes // 4)
        r = np.ascontiguousarray(rays, T.RayQuery)
        out = np.zeros(len(r), T.RayHit)
        flags = (T.IDKPT_TRACE_ANY_HIT if any_hit else 0) | (T.IDKPT_TRACE_LIGHTS if trace_lights else 0)
        self._check(self._L.idkptTraceRays(self._ctx, r.ctypes.data, len(r), flags, out.ctypes.data))
        return out

    def TraceShadows(self, params, depth, normal_oct, visibility=None):
        """Shaders/ShadowsRayTraced/compute.glsl for one point shadow.  depth (H,W), normal_oct (H,W,2); returns visibility (H,W)."""
        import ctypes as C
        d = np.ascontiguousarray(depth, np.float32); n = np.ascontiguousarray(normal_oct, np.float32)
        v = np.zeros(d.shape, np.float32) if visibility is None else np.ascontiguousarray(visibility, np.float32).copy()
        self._check(self._L.idkptTraceShadows(self._ctx, C.addressof(params), d.ctypes.data, n.ctypes.data, v.ctypes.data))
        return v

    def TraceRaysDevice(self, d_rays, d_hits, count, any_hit=False, trace_lights=False):
        """idkptTraceRaysDevice: d_rays / d_hits are device pointers (ints, e.g. torch tensor.data_ptr()) on this context's device: count x 32 B each.
        Asynchronous in the context's stream order; synchronize() completes it."""
        from . import gputypes as T
        flags = (T.IDKPT_TRACE_ANY_HIT if any_hit else 0) | (T.IDKPT_TRACE_LIGHTS if trace_lights else 0)
        self._check(self._L.idkptTraceRaysDevice(self._ctx, int(d_rays), int(count), flags, int(d_hits)))

    def TraceShadowsDevice(self, params, d_depth, d_normal_oct, d_visibility):
        """idkptTraceShadowsDevice: the three images are device pointers (W*H floats, W*H*2 floats, W*H floats in/out).  Asynchronous in stream order."""
        import ctypes as C
        self._check(self._L.idkptTraceShadowsDevice(self._ctx, C.addressof(params), int(d_depth), int(d_normal_oct), int(d_visibility)))

    def RefitBlas(self, blas_id):
        self._check(self._L.idkptRefitBlas(self._ctx, blas_id))

    def UploadUnskinnedVertices(self, verts):
        v = np.ascontiguousarray(verts)
        self._check(self._L.idkptUploadUnskinnedVertices(self._ctx, v.ctypes.data, len(v)))

    def Skin(self, input_offset, output_offset, joint_offset, count):
        self._check(self._L.idkptSkin(self._ctx, input_offset, output_offset, joint_offset, count))

    def SetSceneVersions(self, versions):
        """idkptSetSceneVersions: up to `versions` states of the animated geometry in flight (scene updates no longer launch the queued samples first)."""
        self._check(self._L.idkptSetSceneVersions(self._ctx, int(versions)))

    def SetGroupSharding(self, mode):
        """idkptSetGroupSharding: 0 auto (= bands of 8 rows at every RayDepth, with the per-band count exchange beyond 2), 1 rows, 2 strips + device-side count exchange, 3 bands of 8 rows."""
        self._check(self._L.idkptSetGroupSharding(self._ctx, int(mode)))

    def SetRowRange(self, first_row, row_count):
        """idkptSetRowRange: this context renders the contiguous strip [first_row, first_row + row_count)."""
        self._check(self._L.idkptSetRowRange(self._ctx, int(first_row), int(row_count)))
        self.row_modulo, self.row_remainder, self._row_limit, self.row_band = 1, int(first_row), int(row_count), 1

    def SetBounceExchange(self, fn):
        """idkptSetBounceExchange: fn(bounce, local_counts ndarray[samples]) -> bases ndarray[samples] (alive rays of the same sample
        held by the contexts that own earlier rows); None disables.  Needed for exact N-GPU == 1-GPU results beyond RayDepth 2."""
        import ctypes as C
        old = getattr(self, "_xfn", None)   # the library launches what is queued before it swaps the pointer: the old trampoline must outlive the call
        if fn is None:
            self._check(self._L.idkptSetBounceExchange(self._ctx, None, None))
            self._xfn = None; del old
            return
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))

        def tramp(user, bounce, n, counts, out):
            try:
                b = fn(int(bounce), np.array([counts[i] for i in range(n)], np.uint32))
                for i in range(n):
                    out[i] = int(b[i])
            except BaseException as e:     # ctypes would swallow it and leave `out` unfilled: zero the bases and re-raise from the next _check
                for i in range(n):
                    out[i] = 0
                self._callback_error = e
        new = proto(tramp)
        self._check(self._L.idkptSetBounceExchange(self._ctx, new, None))
        self._xfn = new; del old       # keep the trampoline alive as long as the context uses it

    def SetBandExchange(self, fn):
        """idkptSetBandExchange (interleaved rows / bands): fn(bounce, local_counts ndarray[samples, bands]) -> bases ndarray[samples, bands] = alive rays of the same
        sample, over ALL contexts, in the image bands before each of this context's bands; None disables.  Exact N-GPU == 1-GPU results at any RayDepth (sorting off)."""
        import ctypes as C
        old = getattr(self, "_bxfn", None)   # (see SetBounceExchange: released only after the library has swapped the pointer)
        if fn is None:
            self._check(self._L.idkptSetBandExchange(self._ctx, None, None))
            self._bxfn = None; del old
            return
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))

        def tramp(user, bounce, samples, bands, counts, out):
            n = samples * bands
            try:
                b = np.asarray(fn(int(bounce), np.ctypeslib.as_array(counts, shape=(n,)).astype(np.uint32).reshape(samples, bands)), np.uint32).reshape(-1)
                for i in range(n):
                    out[i] = int(b[i])
            except BaseException as e:
                for i in range(n):
                    out[i] = 0
                self._callback_error = e
        new = proto(tramp)
        self._check(self._L.idkptSetBandExchange(self._ctx, new, None))
        self._bxfn = new; del old      # keep the trampoline alive as long as the context uses it

    def SetBandExchangeDevice(self, fn):
        """idkptSetBandExchangeDevice: fn(bounce, samples, bands, d_counts_ptr, d_bases_ptr, hip_stream_ptr) ENQUEUES on that stream whatever fills the bases (device
        pointers to uint32[samples * bands]) and returns; no host synchronisation.  None disables."""
        import ctypes as C
        old = getattr(self, "_bxdfn", None)
        if fn is None:
            self._check(self._L.idkptSetBandExchangeDevice(self._ctx, None, None))
            self._bxdfn = None; del old
            return
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)

        def tramp(user, bounce, samples, bands, d_counts, d_bases, stream):
            try:
                fn(int(bounce), int(samples), int(bands), int(d_counts or 0), int(d_bases or 0), int(stream or 0))
            except BaseException as e:     # (nothing was enqueued: the bases of this bounce are whatever the buffer held; the error surfaces at the next _check)
                self._callback_error = e
        new = proto(tramp)
        self._check(self._L.idkptSetBandExchangeDevice(self._ctx, new, None))
        self._bxdfn = new; del old     # keep the trampoline alive as long as the context uses it

    def transport_info(self):
        """idkptGetTransportInfo: how a multi-device context moves its bulk device-to-device traffic ('none' | 'peer-copy' | 'rccl'), the RCCL ranks, version, and the library path / the reason RCCL is not in use."""
        kind = C.c_int32(); ranks = C.c_int32(); ver = C.c_int32(); detail = C.c_char_p()
        self._check(self._L.idkptGetTransportInfo(self._ctx, C.byref(kind), C.byref(ranks), C.byref(ver), C.byref(detail)))
        return {"transport": ("none", "peer-copy", "rccl")[kind.value], "rccl_ranks": ranks.value, "rccl_version": ver.value, "detail": (detail.value or b"").decode()}

    @staticmethod
    def transport_self_test(device=0):
        """idkptTransportSelfTest: (status, rccl version, detail) of a one-rank RCCL round trip on `device`."""
        L = _lib.load(); ver = C.c_int32(); buf = C.create_string_buffer(512)
        rc = L.idkptTransportSelfTest(device, C.byref(ver), buf, 512)
        return rc, ver.value, buf.value.decode()

    def synchronize(self):
        self._check(self._L.idkptSynchronize(self._ctx))

    def download(self, which=0):
        out = np.zeros((self.rows, self.width, 4), np.float32)
        self._check(self._L.idkptDownload(self._ctx, which, out.ctypes.data, out.nbytes))
        return out

    def rays(self):
        out = np.zeros(self.rows * self.width, T.GpuWavefrontRay)
        self._check(self._L.idkptDownloadRays(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def alive_queue(self):
        n = C.c_uint32()
        self._check(self._L.idkptDownloadAliveQueue(self._ctx, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint32)
        if n.value:
            self._check(self._L.idkptDownloadAliveQueue(self._ctx, out.ctypes.data, n.value, C.byref(n)))
        return out

    def enable_primary_hit_capture(self, on=True):
        self._check(self._L.idkptEnablePrimaryHitCapture(self._ctx, 1 if on else 0))

    def primary_hits(self):
        n = self.rows * self.width
        t = np.zeros(n, np.float32); tri = np.zeros(n, np.uint32); bary = np.zeros((n, 2), np.float32)
        self._check(self._L.idkptDownloadPrimaryHits(self._ctx, t.ctypes.data, tri.ctypes.data, bary.ctypes.data, n))
        return t, tri, bary

    def UpdateTexture(self, index, image):
        """idkptUpdateTexture: image `index` of the texture table (a float32 array or a gputypes.TextureImage)."""
        t = T.TextureImage.of(image); rec = T.Texture(); t.fill(rec)
        self._check(self._L.idkptUpdateTexture(self._ctx, int(index), C.byref(rec)))

    def enable_counters(self, on=True):
        self._check(self._L.idkptEnableCounters(self._ctx, 1 if on else 0))

    def enable_timing(self, on=True):
        self._check(self._L.idkptEnableTiming(self._ctx, 1 if on else 0))

    def stats(self):
        s = T.Stats()
        self._check(self._L.idkptGetStatsSized(self._ctx, C.addressof(s), C.sizeof(s)))
        return {"rays_traced": s.RaysTraced, "primary_rays": s.PrimaryRays, "frames": s.Frames, "alive_counts": list(s.LastAliveCounts),
                "last_frame_ms": s.LastFrameMs, "node_pair_visits": s.NodePairVisits, "triangle_tests": s.TriangleTests,
                "trace_ms_total": s.TraceMsTotal, "trace_launches": s.TraceLaunches,
                "wide_flagged_rays": s.WideFlaggedRays, "wide_node_visits": s.WideNodeVisits, "wide_leaf_records": s.WideLeafRecords, "wide_triangle_tests": s.WideTriangleTests,
                "inst_tlas_flagged_rays": s.InstTlasFlaggedRays,
                "packet_flagged_rays": s.PacketFlaggedRays, "packet_packets": s.PacketPackets, "packet_node_steps": s.PacketNodeSteps, "packet_live_lanes": s.PacketLiveLanes,
                "packet_rays_entered": s.PacketRaysEntered, "packet_triangle_rounds": s.PacketTriangleRounds,
                "inst_unified_launches": s.InstUnifiedLaunches, "inst_unified_entries": s.InstUnifiedEntries, "inst_unified_top_depth": s.InstUnifiedTopDepth}

    def reset_stats(self):
        self._check(self._L.idkptResetStats(self._ctx))

    def image_device_ptr(self, which=0):
        p = C.c_void_p(); n = C.c_size_t()
        self._check(self._L.idkptGetImageDevicePtr(self._ctx, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- frame ring: several frames (cameras) in flight, idkptSetFrameRing / idkptBeginFrame / idkptDownloadFrame
    def SetFrameRing(self, frames):
        self._check(self._L.idkptSetFrameRing(self._ctx, int(frames)))

    def BeginFrame(self):
        """The next ring slot becomes current (accumulation restarts); returns the slot to read the finished frame from later."""
        slot = C.c_int32()
        self._check(self._L.idkptBeginFrame(self._ctx, C.byref(slot)))
        return slot.value

    def FrameResult(self, slot, which=0):
        out = np.zeros((self.rows, self.width, 4), np.float32)
        self._check(self._L.idkptDownloadFrame(self._ctx, int(slot), which, out.ctypes.data, out.nbytes))
        return out

    def frame_device_ptr(self, slot, which=0):
        p = C.c_void_p(); n = C.c_size_t()
        self._check(self._L.idkptGetFrameDevicePtr(self._ctx, int(slot), which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_max_batch(self, n):
        """Up to n consecutive samples are deferred and traced together (bit-identical results; idkptSetMaxBatch)."""
        self._check(self._L.idkptSetMaxBatch(self._ctx, n))

    def flush(self):
        self._check(self._L.idkptFlush(self._ctx))

    def set_option(self, name, value):
        """idkptSetDeveloperOption: tuning / test hooks (include/idkpt.h); results are bit-identical under all of them."""
        self._check(self._L.idkptSetDeveloperOption(self._ctx, name.encode(), int(value)))

    def set_stream(self, hip_stream_handle):
        self._check(self._L.idkptSetStream(self._ctx, C.c_void_p(hip_stream_handle)))
