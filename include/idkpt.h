/*
 * idkpt.h — C-ABI of libidkpt.so, the MI355X-native replacement for IDKEngine's
 * wavefront path tracer (class PathTracer, Source/Render/PathTracer.cs:10-365).
 *
 * Every entry point replaces a piece of the reference's PathTracer / BVH GL plumbing;
 * the replaced interface is cited per function.  Conventions follow the reference's own
 * native-interop style (Source/OIDN/OIDN.cs:5-122): opaque handle, int32 status
 * (0 = OK), last-error string, host pointers borrowed only for the duration of a call,
 * library owns all device memory.  No torch / HIP types appear in any signature; device
 * pointers and the stream cross as void*.
 */
#ifndef IDKPT_H
#define IDKPT_H

#include "idkpt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define IDKPT_API __attribute__((visibility("default")))

typedef struct idkpt_ctx idkpt_ctx;

enum idkpt_status {
    IDKPT_OK = 0,
    IDKPT_ERR_UNKNOWN = 1,
    IDKPT_ERR_INVALID_ARGUMENT = 2,
    IDKPT_ERR_INVALID_OPERATION = 3,
    IDKPT_ERR_OUT_OF_MEMORY = 4,
    IDKPT_ERR_NO_DEVICE = 5,
    IDKPT_ERR_HIP = 6,
};

/* Which image idkptDownload / idkptGetImageDevicePtr addresses.
 * PathTracer.Result / AlbedoTexture / NormalTexture (PathTracer.cs:143,167-168), RGBA32F. */
enum idkpt_image { IDKPT_IMAGE_RESULT = 0, IDKPT_IMAGE_ALBEDO = 1, IDKPT_IMAGE_NORMAL = 2 };

/* Which scene array idkptUpdateBuffer patches (ModelManager.UpdateBuffers, ModelManager.cs:594-621;
 * LightManager UBO 2; BVH SSBO 20-27). */
enum idkpt_buffer {
    IDKPT_BUF_MESH_TRANSFORMS = 0, /* SSBO 4  */
    IDKPT_BUF_VERTEX_POSITIONS = 1,/* SSBO 8  */
    IDKPT_BUF_VERTICES = 2,        /* SSBO 7  */
    IDKPT_BUF_MESHES = 3,          /* SSBO 2  */
    IDKPT_BUF_MATERIALS = 4,       /* SSBO 3  */
    IDKPT_BUF_LIGHTS = 5,          /* UBO 2 (GpuLight array; count via idkptSetLightCount) */
    IDKPT_BUF_BLAS_NODES = 6,      /* SSBO 22 */
    IDKPT_BUF_TLAS_NODES = 7,      /* SSBO 27 */
    IDKPT_BUF_JOINT_MATRICES = 8,  /* SSBO 16 (3x4 row-major per joint, Skinning) */
    /* idkptDownloadBuffer only — the traversal structure the library derives from the BLAS nodes (no reference counterpart; csrc/wide_nodes.hpp): */
    IDKPT_BUF_WIDE_NODES = 9,      /* 64-byte wide nodes, BLAS b at 64 * (sum over earlier BLASes of NodeCount / 2 + 1) */
    IDKPT_BUF_WIDE_LEAVES = 10,    /* leaf records, 16-byte units */
    IDKPT_BUF_WIDE_COUNTS = 11,    /* per BLAS: uint32 wide nodes in use, uint32 16-byte units of leaf records in use */
};

/* One image of the texture table + the sampler state it is sampled with (stand-in for the GL bindless samplers of GpuMaterial: a texture object and the GLSampler.SamplerState
 * ModelLoader.GetGLSamplerState derives from the glTF sampler, Utils/ModelLoader.cs:1166-1197).  The path tracer samples in compute shaders (`texture(gpuMaterial.X, uv)`,
 * Shaders/include/Surface.glsl:49-77): no derivatives, level of detail 0, so of the sampler state exactly the two wrap modes and the MAGNIFICATION filter take part (GL 4.6
 * 8.14: lambda = 0 selects the mag filter); texel selection and wrapping are the specification's (8.14.2, table 8.20), the linear filter is mix(mix(t00, t10, a), mix(t01, t11, a), b).
 * 8-bit formats are decoded per texel BEFORE filtering (UNORM: c / 255; sRGB: the specification's transfer function on R, G, B, alpha linear — GL 4.6 8.24), as GL does.
 * The zero value of every state field is the round-5 behaviour (REPEAT, REPEAT, LINEAR, RGBA32F): a zero-initialised struct with width / height / rgba set samples as before. */
enum idkpt_wrap { IDKPT_WRAP_REPEAT = 0 /* GL_REPEAT 0x2901 */, IDKPT_WRAP_CLAMP_TO_EDGE = 1 /* GL_CLAMP_TO_EDGE 0x812F */, IDKPT_WRAP_MIRRORED_REPEAT = 2 /* GL_MIRRORED_REPEAT 0x8370 */ };
enum idkpt_filter { IDKPT_FILTER_LINEAR = 0 /* GL_LINEAR 0x2601 */, IDKPT_FILTER_NEAREST = 1 /* GL_NEAREST 0x2600 */ };
enum idkpt_texture_format { IDKPT_TEXFMT_RGBA32F = 0 /* 16 B / texel */, IDKPT_TEXFMT_RGBA8 = 1 /* GL_RGBA8: 4 B / texel, UNORM */, IDKPT_TEXFMT_SRGB8_A8 = 2 /* GL_SRGB8_ALPHA8: 4 B / texel */ };
typedef struct idkpt_texture {
    int32_t width, height;
    const void* rgba;  /* width*height texels, row-major, row 0 = v 0: 4 floats each (RGBA32F) or 4 bytes each (RGBA8, SRGB8_A8) */
    int32_t wrapS, wrapT;   /* enum idkpt_wrap: GLSampler.SamplerState.WrapModeS / WrapModeT */
    int32_t magFilter;      /* enum idkpt_filter: GLSampler.SamplerState.MagFilter */
    int32_t format;         /* enum idkpt_texture_format */
} idkpt_texture;            /* 32 B (ABI 3; round 5: 16 B) */

/* ---- adjacent consumers of the traversal core ---- */
/* One ray of a batched query = one call of TraceRay(ray, hitInfo, traceLights, maxDist)
 * (Shaders/include/BVHIntersect.glsl:293-297) or TraceRayAny (:299-411).  Direction is used as given (not renormalised). */
typedef struct idkpt_ray {
    float Origin[3];   float MaxDist;     /* maxDist argument (PT uses FLOAT_MAX) */
    float Direction[3]; uint32_t _pad0;
} idkpt_ray;                              /* 32 B */
/* HitInfo (BVHIntersect.glsl:10-16) + the function's bool result.  Miss: T = MaxDist, TriangleId = ~0.
 * Light hit (traceLights): TriangleId = ~0, MeshTransformId = light index.  Fields the reference leaves undefined are 0. */
typedef struct idkpt_hit {
    float T; float BaryX, BaryY; uint32_t TriangleId;
    uint32_t MeshTransformId; uint32_t Hit; uint32_t _pad0, _pad1;
} idkpt_hit;                              /* 32 B */
enum idkpt_trace_flags {
    IDKPT_TRACE_ANY_HIT = 1,              /* TraceRayAny: first intersection found wins, children visited left-first */
    IDKPT_TRACE_LIGHTS  = 2               /* traceLights = true (brute-force sphere lights, :189-203 / :304-320) */
};
/* Shaders/ShadowsRayTraced/compute.glsl for ONE point shadow (= one gl_GlobalInvocationID.z slice): what the reference reads from
 * gBufferDataUBO / shadowsUBO / taaDataUBO / perFrameDataUBO crosses as explicit values. */
typedef struct idkpt_shadow_params {
    float   InvProjView[16];              /* perFrameDataUBO.InvProjView, OpenTK memory order (like idkptSetPerFrame) */
    float   TaaJitter[2];                 /* taaDataUBO.Jitter */
    int32_t Width, Height;                /* imageSize(RayTracedShadowMapImage) == g-buffer size */
    int32_t LightIndex;                   /* pointShadow.LightIndex */
    int32_t RayTracingSamples;            /* uniform RayTracingSamples (>= 1) */
    uint32_t NoiseIndex;                  /* taaEnabled ? (Frame % SampleCount) * RayTracingSamples : 0  (compute.glsl:22-23) */
    uint32_t _pad0;
} idkpt_shadow_params;

/* Everything the reference's PathTracer kernels read through fixed GL bindings
 * (Shaders/include/StaticStorageBuffers.glsl:9-173, StaticUniformBuffers.glsl:9-55), as explicit
 * host arrays owned by the caller: BVH.{BlasNodes,BlasTriangles,BlasesDesc,BlasInstances,TlasNodes}
 * (Bvh/BVH.cs:111-115), ModelManager.{Meshes,GpuMaterials,Vertices,VertexPositions,MeshTransforms}
 * (ModelManager.cs:40-50), lights, skybox.  The library copies; nothing is retained. */
typedef struct idkpt_scene_desc {
    const GpuBlasNode*      BlasNodes;        int32_t BlasNodeCount;        /* SSBO 22 */
    const GpuBlasTriangle*  BlasTriangles;    int32_t BlasTriangleCount;    /* SSBO 23 */
    const GpuBlasDesc*      BlasDescs;        int32_t BlasDescCount;        /* SSBO 20 */
    const GpuBlasInstance*  BlasInstances;    int32_t BlasInstanceCount;    /* SSBO 21 */
    const GpuTlasNode*      TlasNodes;        int32_t TlasNodeCount;        /* SSBO 27 (may be 0 when !UseTlas) */
    const int32_t*          BlasParentIndices;int32_t BlasParentIndexCount; /* SSBO 24 (refit; may be 0) */
    const int32_t*          BlasLeafIndices;  int32_t BlasLeafIndexCount;   /* SSBO 25 (refit; may be 0) */
    const float*            VertexPositions;  int32_t VertexCount;          /* SSBO 8: packed float3 */
    const GpuVertex*        Vertices;                                       /* SSBO 7: VertexCount entries */
    const GpuMesh*          Meshes;           int32_t MeshCount;            /* SSBO 2 */
    const GpuMaterial*      Materials;        int32_t MaterialCount;        /* SSBO 3 */
    const GpuMeshTransform* MeshTransforms;   int32_t MeshTransformCount;   /* SSBO 4 */
    const GpuLight*         Lights;           int32_t LightCount;           /* UBO 2 (<= 256) */
    const float*            SkyFaces;         int32_t SkyFaceSize;          /* UBO 5: 6 faces (+X,-X,+Y,-Y,+Z,-Z) x S x S x RGBA32F, sampled GL_LINEAR + seamless like the reference's skybox; S = 1: six constant face colours (unfiltered); NULL => black */
    const idkpt_texture*    Textures;         int32_t TextureCount;         /* texture table for GpuMaterial handles */
} idkpt_scene_desc;

/* PathTracer's public knobs (PathTracer.cs:12-125) + the two BVH macros the kernels are compiled
 * against in the reference (USE_TLAS, BLAS_STACK_SIZE; Bvh/BVH.cs:16-45). */
typedef struct idkpt_settings {
    GpuSettings Gpu;           /* FocalLength, LenseRadius, DoDebugBVHTraversal, DoTraceLights, DoRussianRoulette */
    int32_t RayDepth;          /* PathTracer.RayDepth, default 7 (PathTracer.cs:211) */
    int32_t SamplesPerPixel;   /* PathTracer.SamplesPerPixel, default 1 (:12) */
    int32_t DoRaySorting;      /* PathTracer.DoRaySorting, default 0 (:173) */
    int32_t OutputAOVs;        /* PathTracer.OutputAOVs, default 0 (:174) */
    int32_t UseTlas;           /* BVH.GpuUseTlas, default 0 (Bvh/BVH.cs:156) */
    int32_t BlasStackSize;     /* BVH.BlasStackSize = max RequiredStackSize (Bvh/BVH.cs:559-567); 0 => derive from BlasDescs */
} idkpt_settings;

typedef struct idkpt_stats {
    uint64_t RaysTraced;        /* N + sum_j A_j over all frames since the last idkptResetStats */
    uint64_t PrimaryRays;
    uint64_t Frames;            /* samples rendered */
    uint32_t LastAliveCounts[16]; /* [j>=1]: alive-queue length entering bounce j of the last sample; [0]: primary rays that reached the traversal kernel */
    float    LastTraceMs;       /* mean HIP-event duration of one traversal-kernel launch since idkptResetStats (0 if timing disabled) */
    float    LastFrameMs;       /* HIP-event time of the last idkptRender */
    uint64_t NodePairVisits;    /* only when counters enabled (idkptEnableCounters) */
    uint64_t TriangleTests;
    double   TraceMsTotal;      /* sum of HIP-event durations of all traversal-kernel launches since idkptResetStats (timing enabled) */
    uint64_t TraceLaunches;     /* number of traversal-kernel launches in TraceMsTotal */
    uint64_t WideFlaggedRays;   /* rays the wide-node walk did not vouch for and the exact BVH2 kernel traced again, since idkptResetStats (kernels_wide.hpp) */
    uint64_t WideNodeVisits;    /* only with the developer option "wide_count": 64-byte wide nodes fetched ... */
    uint64_t WideLeafRecords;   /* ... leaf records fetched (80 bytes: the BVH2 leaf node + its first triangle) ... */
    uint64_t WideTriangleTests; /* ... triangle tests of the wide-node walk (every one beyond a record's first is another 48-byte fetch) */
    uint64_t InstTlasFlaggedRays; /* rays of multi-instance scenes without UseTlas that the walk through the library's own TLAS did not vouch for and the exact instance loop traced
                                   * again, since idkptResetStats (developer option "inst_tlas", kernels_trace_inst.hpp) */
    /* ABI 3 (round 6): the packet walk of the primary launches (developer option "packet", kernels_packet.hpp), since idkptResetStats */
    uint64_t PacketFlaggedRays;  /* rays the packet walk did not vouch for and the exact BVH2 kernel traced again */
    uint64_t PacketPackets;      /* packets (waves of 64 consecutive work-list entries) walked */
    uint64_t PacketNodeSteps;    /* node-pair steps of those walks (one 64-byte scalar fetch each) */
    uint64_t PacketLiveLanes;    /* lanes that were live, summed over the node steps: / (64 x PacketNodeSteps) = what the automatic choice looks at */
    uint64_t PacketRaysEntered;  /* rays that entered the BVH inside a packet */
    uint64_t PacketTriangleRounds; /* wave-wide triangle tests (one 48-byte scalar fetch each) */
    /* the unified tree of same-space multi-instance scenes (developer option "inst_unify", kernels_trace_inst.hpp UNI) */
    uint64_t InstUnifiedLaunches;  /* traversal launches that walked it, since idkptResetStats (their flagged rays count in InstTlasFlaggedRays) */
    uint32_t InstUnifiedEntries;   /* subtrees of the BLASes under its top, as last derived (0: not in use) */
    uint32_t InstUnifiedTopDepth;  /* depth of that top */
} idkpt_stats;
/* The layout above only ever GROWS at its end, and IDKPT_ABI_VERSION counts the growths (and every other change a host compiled against an older header could trip over: new
 * enum values of idkpt_buffer, new fields of idkpt_texture).  A host that does not compile against this header (the C# LibraryImport struct of INTEGRATION.md) passes the size
 * of ITS struct to idkptGetStatsSized and gets exactly that many bytes; idkptGetStats(ctx, out) is idkptGetStatsSized(ctx, out, sizeof(idkpt_stats)) of the header the LIBRARY
 * was built with — for hosts built from the same tree.  idkptGetAbiVersion() lets a host refuse a library older than the header it was written against. */
#define IDKPT_ABI_VERSION 3

/* ---- lifetime --------------------------------------------------------------------------- */
/* new PathTracer(w,h,settings) (PathTracer.cs:170-212).  deviceCount = 1: the reference's situation, one GPU.
 * deviceCount = N > 1 (no reference equivalent: Source/EntryPoint.cs:10-33 is single-GPU): ONE handle that renders every frame on N GPUs of
 * the node — the host keeps calling the same entry points.  The scene crosses PCIe once and is replicated device-to-device over xGMI
 * (hipMemcpyPeerAsync), every device renders its rows of the frame (idkptSetGroupSharding), idkptDownload returns the whole frame (each
 * device copies its rows into the host image itself) and idkptGetImageDevicePtr gathers it on the first device.  Results are identical
 * to a one-device context bit for bit (RayDepth <= 2: always; deeper paths: with DoRaySorting off).  deviceIds may be NULL (devices
 * 0..N-1); an id may appear twice (two members on one GPU: how the multi-device code is exercised on a one-GPU box).
 * A process-per-GPU host (torch.distributed / MPI) can instead create one 1-device context per process and use idkptSetRowSharding /
 * idkptSetRowRange + idkptSetBounceExchange (idkengine_amd/dist.py). */
IDKPT_API int32_t idkptCreate(int32_t deviceCount, const int32_t* deviceIds, idkpt_ctx** outCtx);
IDKPT_API int32_t idkptGetContextDeviceCount(idkpt_ctx* ctx, int32_t* outCount);
/* How the bulk device-to-device traffic of a multi-device context travels — the replication of the scene member 0 uploaded and the gather of the members' rows on
 * device 0 (north_star: "RCCL broadcast of the BVH + gather of tiles over xGMI").  RCCL is probed at run time (dlopen of librccl.so, the NativeLibrary.TryLoad pattern of
 * OIDN/OIDN.cs:11-20; the library does not link against it): one communicator per member (ncclCommInitAll), ncclBroadcast per scene buffer, grouped ncclSend / ncclRecv
 * for the gather.  Where RCCL is absent, cannot form the communicators (two members on one GPU) or a call fails, the context uses xGMI peer copies (hipMemcpyPeerAsync)
 * and reports why.  Developer option "transport": 0 (default) RCCL where usable, 1 peer copies, 2 RCCL or idkptUploadScene fails.  The tiny per-bounce count exchange of
 * IDKPT_SHARD_STRIPS always uses peer copies ordered by events (4 bytes per sample: a collective launch would cost more than it moves).
 * outKind: IDKPT_TRANSPORT_*; outRanks: ranks of the RCCL communicator (0 without RCCL); outRcclVersion: ncclGetVersion (0 if never loaded); outDetail: the library path, or
 * the reason RCCL is not in use (valid until the next call on ctx).  Any out pointer may be NULL. */
enum idkpt_transport { IDKPT_TRANSPORT_NONE = 0, IDKPT_TRANSPORT_PEER_COPY = 1, IDKPT_TRANSPORT_RCCL = 2 };
IDKPT_API int32_t idkptGetTransportInfo(idkpt_ctx* ctx, int32_t* outKind, int32_t* outRanks, int32_t* outRcclVersion, const char** outDetail);
/* Loads RCCL and runs a one-rank communicator on `device` through every call the transport uses (broadcast, send + recv, all-gather), checking the bytes: what a host
 * — or a test on a one-GPU box — runs to know that the RCCL path is usable before it creates an N-device context.  outDetail (optional, detailBytes bytes): "ok" or the reason. */
IDKPT_API int32_t idkptTransportSelfTest(int32_t device, int32_t* outRcclVersion, char* outDetail, size_t detailBytes);
/* How a multi-device context deals the image rows to its devices (ignored by a one-device context):
 *   IDKPT_SHARD_ROWS    row y -> device y % N.  Balances sky rows against geometry rows.  Exact at any RayDepth with DoRaySorting off: up to RayDepth 2
 *                       nothing has to be exchanged (radiance does not depend on the queue slot there); beyond, the members' batches are enqueued by one
 *                       host thread each and meet at every bounce to exchange their per-(sample, row) alive counts (idkptSetBandExchange's contract inside
 *                       the group: one stream synchronisation per member and bounce) so that every member numbers its NHit slots as one device does.
 *   IDKPT_SHARD_BANDS   band of 8 rows k -> device k % N (row y -> device (y / 8) % N): the same balance, and every device keeps whole 8x8 pixel
 *                       tiles — the unit a wave of the ray generation and of the primary traversal works on — so no traversal coherence is lost to
 *                       the split (single rows: 1.5-4.5 % at N = 2..8).  Exact where ROWS is (any RayDepth, the same per-bounce meeting beyond 2).
 *   IDKPT_SHARD_STRIPS  contiguous strips + a device-side exchange of the per-sample alive counts at every bounce (peer copies ordered by
 *                       events, no host synchronisation): every strip numbers its NHit queue slots after the alive rays of the strips above
 *                       it (NHit seeds its RNG from the slot, NHit/compute.glsl:54), so N devices == 1 device at any RayDepth with DoRaySorting off.
 *   IDKPT_SHARD_AUTO    (default) = BANDS (ROWS when the image has fewer bands than devices) at every RayDepth.  (Rounds 2-3 switched to STRIPS beyond RayDepth 2, the
 *                       exchange that needs no host synchronisation; but strips balance badly on views with empty rows — BASELINE configs[3] dealt over 8 GPUs projects
 *                       6.0x with bands against 3.8x with strips — and since round 4 the interleaved deals are exact at any depth as well.)  A change of layout
 *                       restarts the accumulation (like idkptSetSize).
 *   With DoRaySorting ON beyond RayDepth 2 no layout is bit-identical to one device (the sorted slot of a ray depends on every other ray of the frame; the exchange
 *   is skipped from the second bounce on): the N-device image is then another legal execution of the reference's own nondeterministic schedule, equal in distribution only. */
enum idkpt_group_sharding { IDKPT_SHARD_AUTO = 0, IDKPT_SHARD_ROWS = 1, IDKPT_SHARD_STRIPS = 2, IDKPT_SHARD_BANDS = 3 };
IDKPT_API int32_t idkptSetGroupSharding(idkpt_ctx* ctx, int32_t mode);
/* PathTracer.Dispose (PathTracer.cs:344-365) */
IDKPT_API int32_t idkptDestroy(idkpt_ctx* ctx);
/* OIDN.GetDeviceError style (OIDN/OIDN.cs:108-112): pointer stays valid until the next call on ctx */
IDKPT_API int32_t idkptGetLastError(idkpt_ctx* ctx, const char** outMessage);
/* OIDN.SetDeviceErrorFunction style (OIDN/OIDN.cs:108-109; the engine's own debug-callback habit, Render/.../PathTracerPipeline.cs:237-243): optional.  `fn` is called
 * with the status code and the message idkptGetLastError would return, at the moment an entry point fails and before it returns that status — on the calling thread,
 * or, for an error inside one member of a multi-device context whose batches are enqueued by one host thread per device, on that thread.  `message` is only valid
 * during the call.  NULL removes the callback.  The callback must not call back into the library with the same context. */
typedef void (*idkpt_error_fn)(void* user, int32_t status, const char* message);
IDKPT_API int32_t idkptSetErrorCallback(idkpt_ctx* ctx, idkpt_error_fn fn, void* user);
/* Library/device probe usable before Create (NativeLibrary.TryLoad pattern, OIDN/OIDN.cs:11-20) */
IDKPT_API int32_t idkptGetDeviceCount(int32_t* outCount);
IDKPT_API const char* idkptGetVersionString(void);
IDKPT_API int32_t idkptGetAbiVersion(void);   /* IDKPT_ABI_VERSION of the header the library was built with */

/* ---- configuration ---------------------------------------------------------------------- */
/* PathTracer.SetSize (PathTracer.cs:299-332): (re)allocates ray/queue/sort buffers and images, resets accumulation */
IDKPT_API int32_t idkptSetSize(idkpt_ctx* ctx, int32_t width, int32_t height);
/* Multi-GPU framebuffer sharding: this context renders image rows y with y % rowModulo == rowRemainder
 * (rowModulo = world size, rowRemainder = rank).  Images/ray buffers then hold only the local rows, in
 * increasing y.  (1,0) = whole frame.  No reference equivalent (single GPU); DESIGN.md "Multi-GPU". */
IDKPT_API int32_t idkptSetRowSharding(idkpt_ctx* ctx, int32_t rowModulo, int32_t rowRemainder);
/* The same deal in bands of bandRows rows (a power of two, 1..64; 1 = idkptSetRowSharding): this context renders the rows y with
 * (y / bandRows) % rowModulo == rowRemainder, kept in increasing y.  bandRows = 8 keeps every 8x8 pixel tile — the unit a wave of the ray
 * generation and of the primary traversal works on — whole on every rank (DESIGN.md "Multi-GPU"). */
IDKPT_API int32_t idkptSetRowBands(idkpt_ctx* ctx, int32_t bandRows, int32_t rowModulo, int32_t rowRemainder);
/* Contiguous strip instead of interleaved rows: this context renders image rows [firstRow, firstRow + rowCount).  Strips keep every
 * context's pixels contiguous in the canonical (pixel-index) order, which the exact multi-GPU mode below needs; interleaved rows
 * balance better and are the default for RayDepth 2. */
IDKPT_API int32_t idkptSetRowRange(idkpt_ctx* ctx, int32_t firstRow, int32_t rowCount);
/* Exact N-GPU == 1-GPU results for RayDepth > 2 (no reference equivalent: single GPU).  NHit seeds its RNG from the queue slot
 * (NHit/compute.glsl:54), so a context that renders a strip must number its slots from the alive rays of all strips above it.
 * At the start of bounce j the library calls fn(user, j, sampleCount, localCounts, outBases) on the caller's thread (it synchronises
 * the stream first): localCounts[k] = this context's alive rays of in-flight sample k entering bounce j; the host fills
 * outBases[k] = sum of the same count over all contexts that own EARLIER rows (one small all-gather per bounce and batch).
 * Exact with DoRaySorting off; with sorting on the per-context sort is local and parity beyond depth 2 is statistical.  NULL disables. */
typedef void (*idkpt_bounce_exchange_fn)(void* user, int32_t bounce, int32_t sampleCount, const uint32_t* localCounts, uint32_t* outBases);
IDKPT_API int32_t idkptSetBounceExchange(idkpt_ctx* ctx, idkpt_bounce_exchange_fn fn, void* user);
/* The same for INTERLEAVED rows / bands (idkptSetRowSharding, idkptSetRowBands), which balance a frame far better than strips (on the headline camera 8 strips scale
 * 4.2x, 8 interleaved shards 7.6x: the middle strips hold the scene).  A context's alive queue is in local pixel order, so the rays of one local band (bandRows rows; single
 * rows with idkptSetRowSharding) are a contiguous run of slots.  At the start of bounce j the library calls fn(user, j, sampleCount, bandCount, localCounts, outBases) on the
 * caller's thread (stream synchronised): localCounts[k * bandCount + b] = alive rays of in-flight sample k in this context's b-th band (image band b * rowModulo +
 * rowRemainder) entering bounce j; the host fills outBases[k * bandCount + b] = alive rays of sample k, summed over ALL contexts, in the image bands BEFORE that band.
 * Exact (N contexts == 1 context, bit for bit) at any RayDepth with DoRaySorting off; with sorting on the exchange is skipped beyond the first bounce.  Single-device
 * contexts; NULL disables. */
typedef void (*idkpt_band_exchange_fn)(void* user, int32_t bounce, int32_t sampleCount, int32_t bandCount, const uint32_t* localCounts, uint32_t* outBases);
IDKPT_API int32_t idkptSetBandExchange(idkpt_ctx* ctx, idkpt_band_exchange_fn fn, void* user);
/* The same exchange without leaving the device: localCounts / outBases are DEVICE pointers (uint32[sampleCount * bandCount], valid until the next call) and the callback only
 * ENQUEUES, on `hipStream` (the context's stream, behind the kernel that writes the counts), whatever fills outBases — an RCCL all-gather of the counts and a small
 * prefix-sum kernel (idkengine_amd/dist.py make_band_exchange_device does it with torch.distributed on the same stream) — and returns; the library never synchronises
 * the stream for it.  Takes precedence over idkptSetBandExchange when both are set. */
typedef void (*idkpt_band_exchange_device_fn)(void* user, int32_t bounce, int32_t sampleCount, int32_t bandCount, const uint32_t* dLocalCounts, uint32_t* dOutBases, void* hipStream);
IDKPT_API int32_t idkptSetBandExchangeDevice(idkpt_ctx* ctx, idkpt_band_exchange_device_fn fn, void* user);
/* Property setters of PathTracer (PathTracer.cs:12-125); changing anything but DoRussianRoulette/sorting/AOV
 * resets accumulation exactly like the reference setters do. */
IDKPT_API int32_t idkptSetSettings(idkpt_ctx* ctx, const idkpt_settings* settings);
IDKPT_API int32_t idkptGetSettings(idkpt_ctx* ctx, idkpt_settings* outSettings);
/* Upload of UBO 1 (Application.cs:144-159).  Only InvProjection, InvView, ViewPos are consumed. */
IDKPT_API int32_t idkptSetPerFrame(idkpt_ctx* ctx, const float invProjection[16], const float invView[16], const float viewPos[3]);
IDKPT_API int32_t idkptSetPerFrameData(idkpt_ctx* ctx, const GpuPerFrameData* perFrame);

/* ---- scene ------------------------------------------------------------------------------ */
/* Replaces BBG.Buffer.Recreate(...) of SSBO 20-27 (Bvh/BVH.cs:441-451), SSBO 2-8 (ModelManager.cs:594-621),
 * lights UBO and skybox handle.  Builds the library's derived HBM layouts. */
IDKPT_API int32_t idkptUploadScene(idkpt_ctx* ctx, const idkpt_scene_desc* scene);
/* Partial update: BBG.Buffer.UploadElements on one of the scene buffers (e.g. ModelManager.cs:236-261).  A patch of BLAS / TLAS nodes is validated
 * before it reaches the device (child indices, stack need) on a host copy of the array WITH the patch applied, so the array must be a valid tree after
 * every call: a host that streams a rebuilt tree in several pieces uses idkptUploadScene / idkptBuildTlas / idkptBuildTlasOnDevice instead (each such
 * patch also costs a device-to-host copy of the node array and two synchronisations).  Updates of up to 256 KB (joint matrices, transforms) do not wait
 * for the GPU. */
IDKPT_API int32_t idkptUpdateBuffer(idkpt_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, const void* data);
/* Replaces image `index` of the texture table (contents, size, format and sampler state): a streamed-in higher mip as base level, a changed sampler.  Stream-ordered behind the
 * samples already queued (they are launched first); the accumulation is the host's to reset, as after any scene update. */
IDKPT_API int32_t idkptUpdateTexture(idkpt_ctx* ctx, int32_t index, const idkpt_texture* texture);
IDKPT_API int32_t idkptSetLightCount(idkpt_ctx* ctx, int32_t count);
/* BVH.TlasBuild upload (Bvh/BVH.cs:278-298): host-built TLAS nodes replace SSBO 27 */
IDKPT_API int32_t idkptBuildTlas(idkpt_ctx* ctx, const GpuTlasNode* nodes, int32_t nodeCount);
/* BVH.TlasBuild + TLAS.Build (Bvh/BVH.cs:278-298, Bvh/TLAS.cs:28-141) done on the device from the resident BLAS roots, instances
 * and mesh transforms (after idkptUpdateBuffer(MESH_TRANSFORMS)/idkptRefitBlas): no host round trip per animated frame.
 * Node array is bit-identical to the serial host build.  searchRadius: TLAS.BuildSettings.SearchRadius (reference: 15).
 * The depth of a device-built TLAS is not known to the host: the per-lane TLAS stack gets min(instanceCount, 32) rows (the reference's
 * TLAS_STACK_SIZE), and a tree deeper than that sets the device-side overflow flag — reported as IDKPT_ERR_INVALID_OPERATION by the next
 * idkptSynchronize / idkptDownload* / idkptGetStats and, for batches that have already finished, by idkptGetImageDevicePtr /
 * idkptGetFrameDevicePtr.  A zero-copy consumer of those pointers must call idkptSynchronize once per scene change to see it. */
IDKPT_API int32_t idkptBuildTlasOnDevice(idkpt_ctx* ctx, int32_t searchRadius);
/* The SweepSAH core of the BLAS build on the GPU: BLAS.GetBuildData + the recursion of BLAS.Build / TrySplit
 * (Bvh/BLAS.cs:128-243, 730-873) over `fragmentCount` boxes (8 floats each: min.xyz, pad, max.xyz, pad — what libidkbvh's idkbvhBlasFragments
 * returns after idkbvhBlasBegin).  outNodes receives max(2 * fragmentCount, 4) nodes in the builder's id scheme (a subtree's ids are reserved
 * from its fragment count; unused entries zero; not compacted), outSortedIdsX the final order of the x-sorted fragment ids (leaves index
 * into it): exactly the two arrays idkbvhBlasCoreSet takes, byte-identical to idkbvhBlasCoreCpu's.  A host-side service: it uses the context's
 * first device and stream, needs no uploaded scene and changes none.  outLevels (may be NULL): depth of the recursion. */
IDKPT_API int32_t idkptBuildBlasCore(idkpt_ctx* ctx, const float* fragmentBoxes, int32_t fragmentCount, GpuBlasNode* outNodes, int32_t* outSortedIdsX, int32_t* outLevels);
/* The WHOLE BLAS build of one geometry on the device — what BVH.BlasesBuild does per BLAS on the CPU (Bvh/BVH.cs:311-371):
 * PreSplitting.PreSplit (or one box per triangle when isRefittable), BLAS.GetBuildData + BLAS.Build incl. OptimizeStackSize and
 * RemoveEmptySubtrees, GetUnindexedTriangles, GetParentIndices / GetLeafIndices (refittable only), ComputeGlobalSAH.  positions: vertexCount x 3
 * floats; triangles: vertex ids into positions (+ MeshId, copied through).  The results stay on the device until idkptBuildBlasFetch copies them
 * into host arrays sized from outInfo (one build per context at a time).  Output bytes are those of libidkbvh's idkbvhBuildBlas (and therefore of
 * the reference's builder as far as that is pinned); nodes carry BLAS-local indices like BLAS.Build's.  A host-side service like
 * idkptBuildBlasCore: first device of the context, no scene needed, none touched.
 * IDKPT_ERR_INVALID_ARGUMENT (nothing built) for: a vertex id out of range, a position of a vertex some triangle references that is not finite (the
 * reference's builder has no defined result for NaN / infinite boxes; vertices no triangle references are neither checked nor uploaded: a host may pass
 * its global vertex array for every BLAS), a PreSplit that asks for more than 2^27 fragments, a triangle whose split recursion needs more than the 64 stack
 * entries the reference allocates (PreSplitting.cs:57: it throws there).  idkbvhBuildBlas refuses the same inputs. */
typedef struct idkpt_blas_build_info {
    int32_t NodeCount, TriangleCount, RequiredStackSize, ParentIndexCount, LeafIndexCount, FragmentCount, Levels, _pad;
    double Sah;      /* ComputeGlobalSAH of the finished tree (parallel binary64 sum: equal to the reference's tree-order sum up to rounding) */
    double BuildMs;  /* wall time of the call, transfers of the inputs included */
} idkpt_blas_build_info;
IDKPT_API int32_t idkptBuildBlas(idkpt_ctx* ctx, const float* positions, int32_t vertexCount, const GpuBlasTriangle* triangles, int32_t triangleCount, int32_t isRefittable,
                                 float preSplitFactor, idkpt_blas_build_info* outInfo);
IDKPT_API int32_t idkptBuildBlasFetch(idkpt_ctx* ctx, GpuBlasNode* outNodes, GpuBlasTriangle* outTriangles, int32_t* outParentIndices, int32_t* outLeafIndices);
/* test hook: the device cbrtf behind PreSplit's priorities (glibc 2.35's algorithm) on n inputs, for comparison with the host's cbrtf */
IDKPT_API int32_t idkptCbrtProbe(idkpt_ctx* ctx, const float* in, float* out, int32_t n);
/* BVH.GpuBlasesRefit(blasId,1) (Bvh/BVH.cs:472-489, Shaders/BLASRefit/compute.glsl) */
IDKPT_API int32_t idkptRefitBlas(idkpt_ctx* ctx, int32_t blasId);
/* ModelManager skinning dispatch (ModelManager.cs:326-353, Shaders/Skinning/compute.glsl):
 * skins vertexCount vertices from unskinned[inputOffset..] into positions/vertices[outputOffset..] */
IDKPT_API int32_t idkptUploadUnskinnedVertices(idkpt_ctx* ctx, const GpuUnskinnedVertex* verts, int32_t count);
IDKPT_API int32_t idkptSkin(idkpt_ctx* ctx, uint32_t inputVertexOffset, uint32_t outputVertexOffset, uint32_t jointMatricesOffset, uint32_t vertexCount);
/* Batched ray queries through the same BVH/TLAS/light data and the same traversal code as the path tracer (uses the context's
 * UseTlas / BlasStackSize settings).  Host pointers, synchronous.  flags: enum idkpt_trace_flags. */
IDKPT_API int32_t idkptTraceRays(idkpt_ctx* ctx, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits);
/* Ray-traced point-light shadows (Shaders/ShadowsRayTraced/compute.glsl:19-127): depth = W*H floats (gBuffer Depth, 1.0 = sky),
 * normalOct = W*H*2 floats (gBuffer Normal .rg, oct-encoded), visibility = W*H floats in/out (pixels the shader returns early
 * on keep their value).  The shader's GetRandomFloat01() (stochastic alpha) runs on an un-seeded RNG in the reference
 * (InitializeRandomSeed is commented out, :24); here the seed is 0 per invocation. */
IDKPT_API int32_t idkptTraceShadows(idkpt_ctx* ctx, const idkpt_shadow_params* params, const float* depth, const float* normalOct, float* visibility);
/* The same two queries on buffers that already live on the context's device — the engine's G-buffer (ShadowsRayTraced/compute.glsl:9-13 binds gBufferData and the
 * result image directly), ray buffers another pass wrote.  No copies; the kernels are enqueued on the context's stream behind everything issued before and the
 * call returns at once: idkptSynchronize (or a wait on idkptGetStream's stream) completes them.  Same layouts, same results bit for bit as the host-pointer calls
 * (whose time is mostly their PCIe copies: 64 B per ray, 16 B per pixel).  Single-device contexts only (IDKPT_ERR_INVALID_OPERATION otherwise). */
IDKPT_API int32_t idkptTraceRaysDevice(idkpt_ctx* ctx, const idkpt_ray* dRays, size_t count, uint32_t flags, idkpt_hit* dHits);
IDKPT_API int32_t idkptTraceShadowsDevice(idkpt_ctx* ctx, const idkpt_shadow_params* params, const float* dDepth, const float* dNormalOct, float* dVisibility);
/* Read back a scene buffer (tests: refit/skinning results). */
IDKPT_API int32_t idkptDownloadBuffer(idkpt_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, void* dst);

/* ---- scene versions: animated frames in flight (no reference equivalent) ------------------- */
/* The reference's frame loop updates the geometry (ModelManager.Update: skinning, BLAS refit, TLAS rebuild; Source/ModelManager.cs:263-361) and then renders
 * ONE frame with it.  With one scene state on the device, queued samples would have to be launched before every such update — an animated host would render
 * one frame at a time, a quarter of the batched rate.  idkptSetSceneVersions(ctx, n) lets the buffers those updates rewrite (BLAS nodes and their derived
 * triangle records, vertices, TLAS nodes, mesh transforms) hold up to n states each: idkptUpdateBuffer (positions, vertices, transforms, joints) / idkptSkin /
 * idkptRefitBlas / idkptBuildTlas / idkptBuildTlasOnDevice run at once, into a state no queued sample reads, and every queued sample is traced with the
 * geometry that was current when idkptRender queued it — frames with different geometry share one batch (with idkptSetFrameRing / idkptSetMaxBatch).
 * Every frame's image is bit-identical to updating and rendering that frame alone.  When all n states are pinned by queued samples the library launches
 * what is queued (exactly what it does before every update when n = 1, the default).  Memory: about 155 B x n per triangle.  Updates of meshes, materials,
 * lights and settings still launch the queued samples first. */
IDKPT_API int32_t idkptSetSceneVersions(idkpt_ctx* ctx, int32_t versions);

/* ---- frame ring: several frames in flight (no reference equivalent) ------------------------ */
/* The reference renders one frame at a time: Compute(), look at Result, move the camera, ResetAccumulation(), Compute() ...  One
 * 1080p frame does not fill an MI355X, so a host that can tolerate a few frames of latency keeps a ring of result images:
 *   idkptSetFrameRing(ctx, n)      n independent result-image sets (slots); 1 = the reference's behaviour (default)
 *   idkptBeginFrame(ctx, &slot)    next slot (0 for the first frame after idkptSetFrameRing / idkptSetSize, then 1, 2, ... wrapping) becomes
 *                                  current and its accumulation restarts; the camera set by idkptSetPerFrame
 *                                  and the samples queued by idkptRender from now on belong to this frame
 *   idkptDownloadFrame / idkptGetFrameDevicePtr(ctx, slot, ...)   the finished image of a slot (launches what is still deferred)
 * With idkptSetMaxBatch(m) up to m queued samples — of different frames, each with its own camera — are traced by one set of
 * launches; every frame's image is bit-identical to rendering that frame alone.  The ring must hold at least as many slots as
 * frames are in flight (a slot that is reused before it was read is overwritten).  idkptDownload / idkptGetImageDevicePtr /
 * idkptResetAccumulation / idkptGetAccumulatedSamples refer to the current slot. */
IDKPT_API int32_t idkptSetFrameRing(idkpt_ctx* ctx, int32_t frames);
IDKPT_API int32_t idkptBeginFrame(idkpt_ctx* ctx, int32_t* outSlot);
IDKPT_API int32_t idkptDownloadFrame(idkpt_ctx* ctx, int32_t slot, int32_t image, float* rgba, size_t bytes);
IDKPT_API int32_t idkptGetFrameDevicePtr(idkpt_ctx* ctx, int32_t slot, int32_t image, void** outPtr, size_t* outBytes);

/* ---- render ----------------------------------------------------------------------------- */
/* PathTracer.ResetAccumulation (PathTracer.cs:334-337) */
IDKPT_API int32_t idkptResetAccumulation(idkpt_ctx* ctx);
IDKPT_API int32_t idkptGetAccumulatedSamples(idkpt_ctx* ctx, uint32_t* outSamples);
/* Sample-parallel multi-GPU (no reference counterpart; the reference is single-GPU): sample i of an accumulation (i = 0 after
 * idkptResetAccumulation) draws the RNG streams the reference's shaders draw with AccumulatedSamples = first + i * stride
 * (FirstHit/compute.glsl:53, NHit/compute.glsl:54, Shading.glsl), while FinalDraw keeps weighting by 1 / (i + 1).  Default (0, 1) is the
 * reference.  N independent contexts (one per GPU, no sharding, no exchange inside a frame) set (r, N): together they render the reference's
 * samples 0 .. N*K-1 exactly once, and the mean of their N accumulated images is an accumulation over all of them.  Changing the
 * sequence restarts the accumulation. */
IDKPT_API int32_t idkptSetSampleSequence(idkpt_ctx* ctx, uint32_t first, uint32_t stride);
/* PathTracer.Compute (PathTracer.cs:214-271): SamplesPerPixel x [FirstHit, (sort,) NHit x (RayDepth-1), FinalDraw].
 * Asynchronous on the context's stream; no host read-back inside. */
IDKPT_API int32_t idkptRender(idkpt_ctx* ctx);
IDKPT_API int32_t idkptSynchronize(idkpt_ctx* ctx);
/* Deferred batching (no reference equivalent; DESIGN.md "Batching"): idkptRender only queues its samples; up to maxBatch
 * (1..256, default 1) consecutive samples are traced together by one set of kernel launches, each with its own
 * AccumulatedSamples index, queue slots and RNG streams, and FinalDraw folds them in submission order — outputs are
 * bit-identical to maxBatch = 1.  Anything that reads results or changes inputs flushes first. */
IDKPT_API int32_t idkptSetMaxBatch(idkpt_ctx* ctx, int32_t maxBatch);
/* Launch pending samples now, without waiting for them. */
IDKPT_API int32_t idkptFlush(idkpt_ctx* ctx);
/* Texture.Download of Result/Albedo/Normal (PathTracerPipeline.cs:177-188). Synchronises. bytes must be rows*width*16. */
IDKPT_API int32_t idkptDownload(idkpt_ctx* ctx, int32_t image, float* rgba, size_t bytes);
/* Internal wavefront state for parity tests: SSBO 30 (GpuWavefrontRay per local pixel) and the alive queue of the
 * last bounce (SSBO 32 AliveRayIndices). */
IDKPT_API int32_t idkptDownloadRays(idkpt_ctx* ctx, GpuWavefrontRay* rays, size_t bytes);
IDKPT_API int32_t idkptDownloadAliveQueue(idkpt_ctx* ctx, uint32_t* indices, size_t capacityElems, uint32_t* outCount);
/* First-hit records (T, TriangleId, BaryX, BaryY) of the last sample's primary rays, for the bit-exact traversal gate. */
/* enable=1: idkptRender keeps a copy of the primary-ray hit records (costs one 16 B/pixel device copy per sample) */
IDKPT_API int32_t idkptEnablePrimaryHitCapture(idkpt_ctx* ctx, int32_t enable);
IDKPT_API int32_t idkptDownloadPrimaryHits(idkpt_ctx* ctx, float* t, uint32_t* triangleId, float* baryXY, size_t pixelCount);

/* ---- instrumentation -------------------------------------------------------------------- */
IDKPT_API int32_t idkptGetStats(idkpt_ctx* ctx, idkpt_stats* outStats);
IDKPT_API int32_t idkptGetStatsSized(idkpt_ctx* ctx, void* outStats, size_t statsBytes);   /* writes min(statsBytes, sizeof(idkpt_stats)) bytes: see IDKPT_ABI_VERSION */
IDKPT_API int32_t idkptResetStats(idkpt_ctx* ctx);
/* enable=1: trace kernels also count node-pair visits / triangle tests (debugCost terms of BVHIntersect.glsl:45,60) */
IDKPT_API int32_t idkptEnableCounters(idkpt_ctx* ctx, int32_t enable);
/* enable=1: record HIP events (on the context's stream) around each idkptRender and around every traversal-kernel launch */
IDKPT_API int32_t idkptEnableTiming(idkpt_ctx* ctx, int32_t enable);

/* Tuning / test hooks — not part of the reference's interface (PathTracer.cs has no counterpart) and never needed for correct results: every option leaves
 * every output bit-identical (tests/test_gpu_worklist.py, test_gpu_wide.py, test_gpu_zz_random_api.py, tools/fuzz_parity.py draw them).  The library reads no
 * environment variable; the Python host mirror forwards IDKPT_<NAME> for tests and tools.  Unknown names fail.  Every name, its values and its default:
 *   which kernels a launch may be given
 *     "force_generic"    0* / 1      thread-per-ray traversal kernels instead of the persistent k_trace2 (the tests' cross-check)
 *     "split"            0-3 (1*)    k_trace2s, long rays split across idle lanes: 0 never, 1 small launches of sparse views, 2 always, 3 always + every split ray re-traced
 *     "split_donor" 0 / 1*, "split_peek" >= 1 (64*), "split_scatter" 0-6 (6*)      its donor rule, how often a busy wave looks at the work list, hand-out granularity
 *     "fused"            0-2 (1*)    k_trace_fused (FirstHit + shading + last NHit in one launch at RayDepth 2): 0 never, 1 small sparse launches, 2 wherever exact
 *     "fused_shade_min"  1-64 (16*)  lanes that wait for its shading phase
 *     "leaf_pool"        -1* / 0-7   pooled leaf phase of k_trace2 MODE 0: mask 1 primary launches, 2 first bounce, 4 later bounces; -1 by view class
 *     "pool_min"         >= 0 (12*)  (ray, triangle) pairs a wave must have parked before they are pooled
 *     "wide"             0* / 1      k_trace_wide: closest hits over the derived 4-wide nodes, unvouched rays re-traced by k_trace2 (csrc/wide_nodes.hpp; profiles/r05_wide_nodes.md)
 *     "wide_cap"         0* / 4-96   rows of its per-lane stack (0 = 24)        "wide_count" 0* / 1   count its node / leaf-record / triangle fetches (idkpt_stats.Wide*)
 *     "packet"           0-2 (1*)    k_trace_packet: the primary launch of a one-BLAS scene as wave-uniform packets (one shared walk per wave of 64 consecutive work-list entries, node pairs and triangles
 *                                    through the scalar cache), unvouched rays re-traced by k_trace2 (idkpt_stats.Packet*; csrc/kernels_packet.hpp, profiles/r06_packet.md): 0 never, 1 pixel-major lists
 *                                    (batches of >= gen_pixel_major samples) while the kernel's own counters show the wave's rays wanting the same nodes, 2 every primary launch of a one-BLAS scene
 *     "packet_min_live"  0-100 (60*) ... percent of a wave's lanes live in an average node step below which the automatic choice turns the packet walk off      "packet_waves" 0* / 1-32   its waves per CU (0 = 28)
 *     "inst_tlas"        >= 0 (8*)   k_trace_inst: scenes of at least this many BLAS instances rendered WITHOUT UseTlas (the reference's instance loop, BVHIntersect.glsl:275-287) walk a
 *                                    TLAS the library builds for itself; rays whose hit could depend on the loop's order are traced again by the exact loop (idkpt_stats.InstTlasFlaggedRays;
 *                                    csrc/kernels_trace_inst.hpp, profiles/r05_instance_tlas.md).  0 = the loop only.  Not used with the counting build, DoDebugBVHTraversal, scene versions.
 *     "inst_tlas_overlap" 0-100 (10*) ... only while a random line through the scene meets at most this many percent of the instances' boxes (measured on the device at every rebuild)
 *     "inst_braid"       >= 0 (0*)   ... and the tree's leaves are subtrees of the instances' BLASes (partial re-braiding, k_braid in csrc/kernels_scene.hpp: the entries with the largest boxes are opened into
 *                                    their node's children until the list has this many entries; scenes whose BLAS boxes nest).  0 = whole instances (measured: more entries lose, profiles/r06_braid.md)
 *     "inst_unify"       >= 0 (4096*) scenes of >= 2 instances that all carry the same InvModel and use every BLAS at most once (the reference's usual static scene: one BLAS per mesh, Bvh/BVH.cs:156)
 *                                    walk ONE tree in their common BLAS space — a top over at most this many subtrees of the BLASes, the BLASes' own nodes below (k_unify_*, csrc/kernels_scene.hpp);
 *                                    the loop's hits, flagged rays traced again by the exact loop as with "inst_tlas".  0 = off.  profiles/r06_braid.md
 *     "pair_nodes"       0 / 1*      k_trace2 FAST: one-BLAS closest-hit launches step on a derived copy of the node pairs whose fields are regrouped for 2-wide arithmetic (64 bytes per pair more
 *                                    device memory): 53 instead of 62 vector instructions per node step, bit-identical
 *     "inst_general"     >= 0 (0*)   scenes of at least this many instances that are NOT one space walk one array too (k_trace_inst TREE 2: a world-space top whose entries take the ray into their
 *                                    instance's space).  Measured: equal to the own TLAS with whole instances, slower with subtrees (loose world boxes, a RayTransform per entry): off
 *     "uni_refill"       16 / 32*    idle lanes at which a wave of the unified walk takes new rays
 *     "inst_unify_radius" 1-512 (15*) PLOC search radius of the unified tree's top (larger radii measured 2-30 % slower)
 *     "inst_sieve"       >= 0 (8*)   k_trace_inst<EXACT>: scenes of at least this many instances (at most 1024) that keep the loop run it with the instances a ray's line cannot meet sieved out when
 *                                    the wave takes the ray — the loop itself, visit for visit (also: the kernel behind the own-TLAS walk's flagged rays, and idkptTraceRays' closest hits).  0 = k_trace2 MODE 1
 *     "inst_sieve_overlap" 0-100 (50*) ... only up to this overlap (as above)
 *     "query_scheduler"  0 / 1*      idkptTraceRays through k_trace2's scheduler (0: thread-per-ray kernel, the cross-check)
 *     "defer_last"       0 / 1*      without AOVs only the radiance of a sample's last bounce is computed per frame; its continuation when a host asks (idkptDownloadRays ...)
 *     "gen_pixel_major"  >= 0 (8*)   batches of at least this many samples hand their primary rays to the traversal pixel by pixel (a wave = 4 pixels x 16 samples instead of an 8x8 tile of one
 *                                    sample; profiles/r05_pixel_major.md); 0 = never.   "gen_group_max" 1-16 (16*): samples per group of that list
 *     "bounce_pixel_major" 0-2 (1*) the first bounce traced in that order too (its own work list, hits per ray id): 0 never, 1 on sparse views, 2 always
 *     "no_tile_cull"     0* / 1      no per-tile pre-classification of sky tiles      "no_lean_primary" 0* / 1   k_gen_primary stores the full state of surviving rays
 *   scheduling of the traversal kernels
 *     "leaf_min"         0* / 1-64   lanes parked on a leaf before the node phase is left (0: 16 for batches of >= 4 samples, 12 below)
 *     "adv_min"          0* / 1-64   MODE 1 / 2: lanes that enter the next instance / walk the TLAS together (0: 8 for batches of >= 4 samples, 1 below)
 *     "grab_unit_log2"   6-24 (10*)  run length of a work-list slice        "grab_fixed" >= 0 (0*)   entries reserved per atomic (0: what the refill needs)
 *     "trace_waves"      0* / n      one-wave workgroups per CU of the persistent grid (0: what LDS allows, at most 32)      "lds_pad" >= 0 (0*)   extra LDS bytes per workgroup
 *     "grid_hint" (2*), "grid_rays_x4" (6*), "grid_mid_waves" (20*)   grid of a launch from the previous batch's counts: x the queue length; quarter-rays per lane; waves per CU below 14 M rays; 0 = off
 *   multi-device contexts
 *     "transport"        0* / 1 / 2  RCCL where usable / peer copies / RCCL or fail (idkptGetTransportInfo)      "force_no_peer" 0* / 1   stage device-to-device copies through pinned host memory
 *     "group_threads"    -1* / 0 / 1 members' batches enqueued by one host thread each (-1: from 4 members on)
 *   builder                "bvh_timing" 0* / 1 (phase times on stderr), "bvh_small" (32*), "bvh_stackopt_host" 0* / 1 (force the host fallback of OptimizeStackSize)
 *   developer builds only  "trace_variant" (s_memtime-instrumented / probe instantiations: 107, 113, 116, 213, 901-904, 961, 962), "graph_probe"   (libidkpt_dev.so, -DIDKPT_DEVELOPER) */
IDKPT_API int32_t idkptSetDeveloperOption(idkpt_ctx* ctx, const char* name, int32_t value);

/* ---- interop (device pointers as void*, for RCCL gather of row shards by the host process) -- */
/* Launches what is still deferred and returns the image of the current slot (ordered on the context's stream, idkptGetStream).  Multi-device
 * context: the rows of all devices are gathered into a full frame on the first device (peer copies; valid until the next call). */
IDKPT_API int32_t idkptGetImageDevicePtr(idkpt_ctx* ctx, int32_t image, void** outPtr, size_t* outBytes);
/* Use an externally created hipStream_t (e.g. torch's current stream) for all work; NULL = library stream */
IDKPT_API int32_t idkptSetStream(idkpt_ctx* ctx, void* hipStream);
IDKPT_API int32_t idkptGetStream(idkpt_ctx* ctx, void** outHipStream);

#ifdef __cplusplus
}
#endif
#endif /* IDKPT_H */
