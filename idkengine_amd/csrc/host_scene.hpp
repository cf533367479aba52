// host_scene.hpp — validation and residency of the scene: context lifetime, settings, idkptUploadScene / CloneSceneFrom / UpdateBuffer / DownloadBuffer, TLAS builds, refit, skinning.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

// Rows a traversal stack needs for one BLAS: BLAS.ComputeRequiredStackSize (Bvh/BLAS.cs:672-702) evaluated bottom-up.  Requires what the
// validation established first: every child pair lies behind its parent (acyclic), so one reverse sweep over the node array suffices.
static int blas_stack_need(const GpuBlasNode* nodes, int nodeCount)
{
    std::vector<int> need((size_t)nodeCount, 0);                       // need[p] = stack rows while traversing the pair (p, p+1)
    for (int p = nodeCount - 2; p >= 2; p--) {
        const GpuBlasNode& l = nodes[p]; const GpuBlasNode& r = nodes[p + 1];
        const bool tl = l.TriCount == 0 && l.TriStartOrChild != 0, tr = r.TriCount == 0 && r.TriStartOrChild != 0;
        if (tl && tr) need[p] = std::max(need[l.TriStartOrChild], need[r.TriStartOrChild]) + 1;
        else if (tl || tr) need[p] = need[tl ? l.TriStartOrChild : r.TriStartOrChild];
    }
    return nodeCount > 2 ? need[2] : 0;
}

// Host-built TLAS nodes (BVH.TlasBuild / TLAS.Build, Bvh/TLAS.cs:28-141: parents are placed in front of their children): index
// validation + the depth the per-lane TLAS stack must hold.  Returns < 0 with `why` set when the array is not a valid tree.
static int tlas_validate(const GpuTlasNode* nodes, int nodeCount, int instanceCount, const char** why)
{
    if (nodeCount <= 0) return 0;
    const uint32_t* w = (const uint32_t*)nodes;                         // 8 dwords per node: Min.xyz, IsLeaf:1|ChildOrInstanceID:31, Max.xyz, pad
    std::vector<int> need((size_t)nodeCount, 0);
    for (int i = nodeCount - 1; i >= 0; i--) {
        const uint32_t packed = w[8 * (size_t)i + 3], id = packed & 0x7fffffffu;
        if (packed >> 31) { if (id >= (uint32_t)instanceCount) { *why = "TLAS leaf references an instance out of range"; return -1; } continue; }
        if (id <= (uint32_t)i || (uint64_t)id + 1 >= (uint64_t)nodeCount) { *why = "TLAS child index out of range (children must lie behind their parent)"; return -1; }
        need[i] = std::max(need[id], need[id + 1]) + 1;
    }
    return need[0];
}

// BLAS node arrays a host hands over (idkptUploadScene, idkptUpdateBuffer on IDKPT_BUF_BLAS_NODES): index validation so that a bad array cannot
// fault the GPU, plus the traversal stack the trees need.  Returns null when valid, else the reason.
static const char* validate_blas_nodes(const GpuBlasNode* nodes, int nodeCount, const GpuBlasDesc* descs, int descCount, int triangleCount, bool checkClaim, int* outMaxStack)
{
    int maxStack = 1;
    for (int i = 0; i < descCount; i++) {
        const GpuBlasDesc& d = descs[i];
        if (!(d.NodeOffset >= 0 && d.NodeCount >= 4 && d.NodeOffset + d.NodeCount <= nodeCount && d.TriangleOffset >= 0 && d.TriangleOffset + d.TriangleCount <= triangleCount)) return "BlasDesc range out of bounds";
        for (int n = 1; n < d.NodeCount; n++) {
            const GpuBlasNode& nd = nodes[d.NodeOffset + n];
            if (nd.TriCount > 0) { if (!((uint64_t)nd.TriStartOrChild + nd.TriCount <= (uint64_t)d.TriangleCount)) return "leaf triangle range out of bounds"; }
            else if (n == 1 || nd.TriStartOrChild != 0) { if (!(nd.TriStartOrChild >= 2 && nd.TriStartOrChild > (uint32_t)n && nd.TriStartOrChild + 1 < (uint32_t)d.NodeCount)) return "child index out of bounds (children must lie behind their parent)"; }
        }
        // the traversal stack is sized from what the tree really needs (BLAS.ComputeRequiredStackSize, Bvh/BLAS.cs:672-702); a host that
        // claims less in RequiredStackSize would have compiled the reference's shaders with too small a BLAS_STACK_SIZE (Bvh/BVH.cs:559-567)
        const int need = blas_stack_need(nodes + d.NodeOffset, d.NodeCount);
        if (checkClaim && d.RequiredStackSize < need) return "BlasDesc.RequiredStackSize is smaller than the stack the BLAS needs";
        maxStack = std::max(maxStack, need);
    }
    *outMaxStack = maxStack;
    return nullptr;
}

// Are all BLASes nested — every child box inside its parent's box?  (What the builder produces, and what a refit keeps: a parent is the union of its children.
// k_trace2s's exactness argument needs it; a host-patched tree that is not nested simply never gets that kernel.)  NaN coordinates compare false: not nested.
static bool blas_nested(const GpuBlasNode* nodes, const GpuBlasDesc* descs, int descCount)
{
    for (int i = 0; i < descCount; i++) {
        const GpuBlasDesc& d = descs[i];
        for (int n = 1; n < d.NodeCount; n++) {
            const GpuBlasNode& p = nodes[d.NodeOffset + n];
            if (p.TriCount > 0 || p.TriStartOrChild == 0) continue;
            for (int k = 0; k < 2; k++) {
                const GpuBlasNode& c = nodes[d.NodeOffset + p.TriStartOrChild + k];
                if (c.TriCount == 0 && c.TriStartOrChild == 0 && n != 0) continue;          // (an empty node is never entered)
                for (int a = 0; a < 3; a++) if (!(c.Min[a] >= p.Min[a] && c.Max[a] <= p.Max[a])) return false;
            }
        }
    }
    return true;
}

// ---- single-device implementation of the C-ABI (dev_*); the exported entry points and the multi-device group layer are in idkpt_api.hpp

static const char* dev_GetVersionString(void) { return "idkpt 0.1 (gfx950)"; }

static int32_t dev_GetDeviceCount(int32_t* outCount)
{
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    if (outCount) *outCount = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? IDKPT_OK : IDKPT_ERR_NO_DEVICE;
}

static int32_t dev_Create(int32_t deviceCount, const int32_t* deviceIds, dev_ctx** outCtx)
{
    if (!outCtx) return IDKPT_ERR_INVALID_ARGUMENT;
    *outCtx = nullptr;
    if (deviceCount != 1) return IDKPT_ERR_INVALID_ARGUMENT; // one member per device (several devices: the group layer, idkpt_api.hpp)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return IDKPT_ERR_NO_DEVICE;
    int dev = deviceIds ? deviceIds[0] : 0;
    if (dev < 0 || dev >= n) return IDKPT_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(dev) != hipSuccess) return IDKPT_ERR_HIP;
    dev_ctx* ctx = new dev_ctx();
    ctx->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ctx->numCUs = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return IDKPT_ERR_HIP; }
    memset(&ctx->st, 0, sizeof(ctx->st));
    ctx->st.Gpu.FocalLength = 8.0f; ctx->st.Gpu.DoRussianRoulette = 1; ctx->st.RayDepth = 7; ctx->st.SamplesPerPixel = 1;
    ctx->stCaller = ctx->st;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    memset(ctx->invProj, 0, 64); memset(ctx->invView, 0, 64); memset(ctx->viewPos, 0, 12);
    if (hipHostMalloc((void**)&ctx->hCounts, MAX_DEPTH_SLOTS * 4 + 16, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->dCountsMirror, ctx->hCounts, 0) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    memset(ctx->hCounts, 0, MAX_DEPTH_SLOTS * 4 + 16);
    if (hipHostMalloc((void**)&ctx->hOverflow, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->dOverflow, ctx->hOverflow, 0) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    *ctx->hOverflow = 0;
    if (hipHostMalloc((void**)&ctx->hBases, MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->dBasesMirror, ctx->hBases, 0) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    memset(ctx->hBases, 0, MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4);
    (void)hipEventCreate(&ctx->evFrame[0]); (void)hipEventCreate(&ctx->evFrame[1]);
    *outCtx = ctx;
    return IDKPT_OK;
}

static void builder_scratch_free(dev_ctx* ctx);
static int32_t dev_Destroy(dev_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    ctx->pending.clear();
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* all[] = {&ctx->srgbLut, &ctx->pmList, &ctx->pairNodes, &ctx->instRec, &ctx->entRec, &ctx->braidBuf, &ctx->unodes, &ctx->utlas, &ctx->uTabs, &ctx->uniBuf, &ctx->uniEntRec, &ctx->itlas, &ctx->imarks, &ctx->ichunks, &ctx->wnodes, &ctx->wleaf, &ctx->wids, &ctx->wpair, &ctx->wcounts, &ctx->wtotals, &ctx->nodes, &ctx->tris, &ctx->triVerts, &ctx->descs, &ctx->instances, &ctx->tlas, &ctx->parents, &ctx->leaves, &ctx->positions, &ctx->prevPositions, &ctx->vertices, &ctx->meshes,
                     &ctx->materials, &ctx->xforms, &ctx->lights, &ctx->sky, &ctx->texDescs, &ctx->unskinned, &ctx->joints, &ctx->levelNodes, &ctx->tlasScratch, &ctx->queryIn, &ctx->queryOut, &ctx->queryRec, &ctx->queryList, &ctx->bandTab, &ctx->tileClass, &ctx->gbases, &ctx->camTab, &ctx->verTab, &ctx->trRec, &ctx->contFlag, &ctx->blockSums, &ctx->rayO, &ctx->rayT, &ctx->rayR, &ctx->aovA, &ctx->aovN, &ctx->hit,
                     &ctx->hitCost, &ctx->primHit, &ctx->queue[0], &ctx->queue[1], &ctx->keys[0], &ctx->keys[1], &ctx->keysTmp, &ctx->sortKeys, &ctx->sortVals, &ctx->contMask, &ctx->waveCounts,
                     &ctx->counts, &ctx->work, &ctx->qwork, &ctx->radSave, &ctx->deferCount, &ctx->sortHist, &ctx->counters64, &ctx->bases, &ctx->img[0], &ctx->img[1], &ctx->img[2]};
    for (DevBuf* b : all) b->release();
    for (auto& t : ctx->texData) t.release();
    builder_scratch_free(ctx);
    if (ctx->hCounts) (void)hipHostFree(ctx->hCounts);
    if (ctx->hOverflow) (void)hipHostFree(ctx->hOverflow);
    if (ctx->hInstOverlap) (void)hipHostFree(ctx->hInstOverlap);
    if (ctx->hUni) (void)hipHostFree(ctx->hUni);
    if (ctx->hPkStats) (void)hipHostFree(ctx->hPkStats);
    if (ctx->hBases) (void)hipHostFree(ctx->hBases);
    if (ctx->hCams) (void)hipHostFree(ctx->hCams);
    for (int i = 0; i < 2; i++) if (ctx->evCams[i]) (void)hipEventDestroy(ctx->evCams[i]);
    if (ctx->hVerTab) (void)hipHostFree(ctx->hVerTab);
    for (int i = 0; i < 2; i++) if (ctx->evVer[i]) (void)hipEventDestroy(ctx->evVer[i]);
    if (ctx->hStage) (void)hipHostFree(ctx->hStage);
    for (int i = 0; i < 4; i++) if (ctx->evStage[i]) (void)hipEventDestroy(ctx->evStage[i]);
    if (ctx->evFrame[0]) (void)hipEventDestroy(ctx->evFrame[0]);
    if (ctx->evFrame[1]) (void)hipEventDestroy(ctx->evFrame[1]);
    for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
    if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return IDKPT_OK;
}

static int32_t dev_GetLastError(dev_ctx* ctx, const char** outMessage)
{
    if (!ctx || !outMessage) return IDKPT_ERR_INVALID_ARGUMENT;
    *outMessage = ctx->lastError.c_str();
    return IDKPT_OK;
}

static int32_t dev_SetSize(dev_ctx* ctx, int32_t width, int32_t height)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(width > 0 && height > 0 && width <= 4096 && height <= 65536, "idkptSetSize: bad size (FirstHit seeds pack x into 12 bits: width <= 4096)");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    REQUIRE(ctx->rowLimit == 0x7fffffff || ctx->rowRem + ctx->rowLimit <= height, "idkptSetSize: the strip set by idkptSetRowRange exceeds the new image height (set a new range first)");
    REQUIRE((ctx->rowRem << ctx->rowBandLog2) < height, "idkptSetSize: this context's row remainder (idkptSetRowSharding / idkptSetRowBands) is outside the new image height");
    ctx->W = width; ctx->H = height; ctx->rows = std::min(ctx->rowLimit, local_rows(height, ctx->rowMod, ctx->rowRem, ctx->rowBandLog2));
    return alloc_frame(ctx);
}

// size and row layout in one step (group layer): bands (y >> bandLog2) % rowMod == rowRem (rowLimit = 0x7fffffff) or the strip [rowRem, rowRem + rowLimit) (rowMod = 1, bandLog2 = 0)
static int32_t dev_SetLayout(dev_ctx* ctx, int32_t width, int32_t height, int32_t rowMod, int32_t rowRem, int32_t rowLimit, int32_t bandLog2 = 0)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(width > 0 && height > 0 && width <= 4096 && height <= 65536, "idkptSetSize: bad size (FirstHit seeds pack x into 12 bits: width <= 4096)");
    REQUIRE(rowMod >= 1 && rowRem >= 0 && bandLog2 >= 0 && bandLog2 <= 6 && (rowRem << bandLog2) < height && rowLimit >= 1 && (rowMod == 1 ? bandLog2 == 0 : rowRem < rowMod), "internal: bad row layout");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    ctx->W = width; ctx->H = height; ctx->rowMod = rowMod; ctx->rowRem = rowRem; ctx->rowLimit = rowLimit; ctx->rowBandLog2 = bandLog2;
    ctx->rows = std::min(rowLimit, local_rows(height, rowMod, rowRem, bandLog2));
    return alloc_frame(ctx);
}

// rows dealt in bands of bandRows rows: band k of the image (rows [k * bandRows, (k + 1) * bandRows)) belongs to the context with k % rowModulo == rowRemainder
// (bandRows = 1: idkptSetRowSharding)
static int32_t dev_SetRowBands(dev_ctx* ctx, int32_t bandRows, int32_t rowModulo, int32_t rowRemainder)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(bandRows >= 1 && bandRows <= 64 && (bandRows & (bandRows - 1)) == 0, "idkptSetRowBands: bandRows must be a power of two in 1..64");
    REQUIRE(rowModulo >= 1 && rowRemainder >= 0 && rowRemainder < rowModulo, "idkptSetRowSharding / idkptSetRowBands: need 0 <= remainder < modulo");
    int bandLog2 = 0; while ((1 << bandLog2) < bandRows) bandLog2++;
    if (rowModulo == 1) bandLog2 = 0;                                   // the whole frame: bands mean nothing
    REQUIRE(ctx->W <= 0 || (rowRemainder << bandLog2) < ctx->H, "idkptSetRowSharding / idkptSetRowBands: no row of the image has this remainder");
    FLUSH();
    ctx->rowMod = rowModulo; ctx->rowRem = rowRemainder; ctx->rowLimit = 0x7fffffff; ctx->rowBandLog2 = bandLog2;
    if (ctx->W > 0) { HIPC(hipSetDevice(ctx->device)); ctx->rows = local_rows(ctx->H, rowModulo, rowRemainder, bandLog2); return alloc_frame(ctx); }
    return IDKPT_OK;
}
static int32_t dev_SetRowSharding(dev_ctx* ctx, int32_t rowModulo, int32_t rowRemainder) { return dev_SetRowBands(ctx, 1, rowModulo, rowRemainder); }

static int32_t dev_SetRowRange(dev_ctx* ctx, int32_t firstRow, int32_t rowCount)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(firstRow >= 0 && rowCount >= 1, "idkptSetRowRange: need firstRow >= 0 and rowCount >= 1");
    REQUIRE(ctx->W <= 0 || firstRow + rowCount <= ctx->H, "idkptSetRowRange: strip exceeds the image height");
    FLUSH();
    ctx->rowMod = 1; ctx->rowRem = firstRow; ctx->rowLimit = rowCount; ctx->rowBandLog2 = 0;
    if (ctx->W > 0) { HIPC(hipSetDevice(ctx->device)); ctx->rows = std::min(ctx->rowLimit, local_rows(ctx->H, 1, firstRow)); return alloc_frame(ctx); }
    return IDKPT_OK;
}

static int32_t dev_SetBounceExchange(dev_ctx* ctx, idkpt_bounce_exchange_fn fn, void* user)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    FLUSH();
    ctx->exchangeFn = fn; ctx->exchangeUser = user;
    return IDKPT_OK;
}

static int32_t dev_SetBandExchange(dev_ctx* ctx, idkpt_band_exchange_fn fn, void* user)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    FLUSH();
    ctx->bandExchangeFn = fn; ctx->bandExchangeUser = user;
    return IDKPT_OK;
}

static int32_t dev_SetBandExchangeDevice(dev_ctx* ctx, idkpt_band_exchange_device_fn fn, void* user)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    FLUSH();
    ctx->bandExchangeDevFn = fn; ctx->bandExchangeDevUser = user;
    return IDKPT_OK;
}

static int32_t dev_SetSettings(dev_ctx* ctx, const idkpt_settings* s)
{
    if (!ctx || !s) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(s->RayDepth >= 1 && s->RayDepth < MAX_DEPTH_SLOTS - 2, "idkptSetSettings: RayDepth out of range (1..61)");   // (count words MAX_DEPTH_SLOTS - 2 / - 1 are the pixel-major bounce list's and the primary list's lengths: host_schedule.hpp)
    REQUIRE(s->SamplesPerPixel >= 1, "idkptSetSettings: SamplesPerPixel must be >= 1");
    REQUIRE(s->BlasStackSize >= 0, "idkptSetSettings: BlasStackSize must be >= 0 (0 = derive from BlasDescs)");
    // BVH.BlasStackSize is the maximum RequiredStackSize of all BLASes (Bvh/BVH.cs:559-567): a smaller stack cannot hold the traversal
    REQUIRE(s->BlasStackSize == 0 || !ctx->haveScene || s->BlasStackSize >= ctx->sceneStack, "idkptSetSettings: BlasStackSize is smaller than the scene's maximum RequiredStackSize");
    if (memcmp(&ctx->stCaller, s, sizeof(*s)) == 0) return IDKPT_OK;   // the struct the host pushed last time: nothing changed
    FLUSH();                                                            // pending samples were submitted under the old settings
    ctx->stCaller = *s;
    const idkpt_settings o = ctx->st;
    // PathTracer setters that call ResetAccumulation (PathTracer.cs:17-98): RayDepth, FocalLength, LenseRadius, DoDebugBVHTraversal, DoTraceLights
    bool reset = o.RayDepth != s->RayDepth || o.Gpu.FocalLength != s->Gpu.FocalLength || o.Gpu.LenseRadius != s->Gpu.LenseRadius ||
                 o.Gpu.DoDebugBVHTraversal != s->Gpu.DoDebugBVHTraversal || o.Gpu.DoTraceLights != s->Gpu.DoTraceLights || o.UseTlas != s->UseTlas;
    ctx->st = *s;
    if (ctx->st.Gpu.DoDebugBVHTraversal) ctx->st.RayDepth = 1; // PathTracer.cs:67-71
    if (reset) std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);
    return IDKPT_OK;
}
static int32_t dev_GetSettings(dev_ctx* ctx, idkpt_settings* out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = ctx->st; return IDKPT_OK; }

static int32_t dev_SetPerFrame(dev_ctx* ctx, const float invProjection[16], const float invView[16], const float viewPos[3])
{
    if (!ctx || !invProjection || !invView || !viewPos) return IDKPT_ERR_INVALID_ARGUMENT;
    // one camera per batch unless a frame ring is active (then every queued sample carries its own camera)
    if (ctx->ringSize == 1 && (memcmp(ctx->invProj, invProjection, 64) || memcmp(ctx->invView, invView, 64) || memcmp(ctx->viewPos, viewPos, 12))) FLUSH_KEEP();
    memcpy(ctx->invProj, invProjection, 64); memcpy(ctx->invView, invView, 64); memcpy(ctx->viewPos, viewPos, 12);
    return IDKPT_OK;
}
static int32_t dev_SetPerFrameData(dev_ctx* ctx, const GpuPerFrameData* p) { if (!ctx || !p) return IDKPT_ERR_INVALID_ARGUMENT; return dev_SetPerFrame(ctx, p->InvProjection, p->InvView, p->ViewPos); }

// triVerts[first, first + count) from the current positions, into a slot queued samples do not read (ver_writable)
static int regather_triverts(dev_ctx* ctx, uint32_t first, uint32_t count)
{
    if (count == 0) return IDKPT_OK;
    char *src, *dst; int rc = ver_writable(ctx, VB_TRIVERTS, first == 0 && count == (uint32_t)ctx->triCount, &src, &dst); if (rc) return rc;
    hipLaunchKernelGGL(k_gather_triverts, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, ctx->tris.as<uint4>(), ctx->positions.as<float>(), (float4*)dst, first, count);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

static int upload(dev_ctx* ctx, DevBuf& b, const void* src, size_t bytes)
{
    HIPC(b.ensure(std::max<size_t>(bytes, 16)));
    if (bytes) HIPC(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return IDKPT_OK;
}

// ---- texture table (idkpt_texture: include/idkpt.h) -----------------------------------------------------------------------------------------------------------------
static size_t tex_texel_bytes(int32_t format) { return format == IDKPT_TEXFMT_RGBA32F ? 16 : 4; }
static int tex_validate(dev_ctx* ctx, const idkpt_texture& t, const char* who)
{
    if (!(t.width > 0 && t.height > 0 && t.rgba)) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, std::string(who) + ": bad texture (size / data)");
    if (!(t.wrapS >= 0 && t.wrapS <= 2 && t.wrapT >= 0 && t.wrapT <= 2 && t.magFilter >= 0 && t.magFilter <= 1 && t.format >= 0 && t.format <= 2))
        return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, std::string(who) + ": bad texture (wrapS / wrapT are enum idkpt_wrap, magFilter enum idkpt_filter, format enum idkpt_texture_format)");
    return IDKPT_OK;
}
static uint32_t tex_state(const idkpt_texture& t) { return (uint32_t)t.wrapS | ((uint32_t)t.wrapT << 2) | ((uint32_t)t.magFilter << 4) | ((uint32_t)t.format << 5); }
// can a texel of this image be non-finite?  (k_shade_last's "no emission" shortcut: 0 x a texel is only 0 for a finite texel; 8-bit formats decode to [0, 1])
static bool tex_all_finite(const idkpt_texture& t)
{
    if (t.format != IDKPT_TEXFMT_RGBA32F) return true;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(t.rgba); uint32_t bad = 0;
    for (size_t k = 0, e = (size_t)t.width * t.height * 4; k < e; k++) bad |= (uint32_t)((w[k] & 0x7f800000u) == 0x7f800000u);
    return !bad;
}
// the sRGB -> linear transfer function of GL 4.6 8.24 per byte value, evaluated in double and rounded once (the oracle makes the same table with the same expression)
static int tex_srgb_lut(dev_ctx* ctx)
{
    if (ctx->srgbLut.p) return IDKPT_OK;
    float lut[256];
    for (int b = 0; b < 256; b++) { const double cs = (double)b / 255.0; lut[b] = (float)(cs <= 0.04045 ? cs / 12.92 : pow((cs + 0.055) / 1.055, 2.4)); }
    HIPC(ctx->srgbLut.ensure(sizeof(lut)));
    HIPC(hipMemcpy(ctx->srgbLut.p, lut, sizeof(lut), hipMemcpyHostToDevice));      // (lut is on the stack)
    return IDKPT_OK;
}
static int tex_descs_upload(dev_ctx* ctx)
{
    std::vector<TexDesc> td;
    for (size_t i = 0; i < ctx->texData.size(); i++) td.push_back({ctx->texData[i].p, ctx->texDims[i].first, ctx->texDims[i].second, ctx->texState[i], 0u});
    { int rc = upload(ctx, ctx->texDescs, td.data(), td.size() * sizeof(TexDesc)); if (rc) return rc; }
    if (!td.empty()) HIPC(hipStreamSynchronize(ctx->stream));                       // (td is a stack vector)
    return IDKPT_OK;
}

// The fast path stores nothing but a flag for pre-culled pixels of the most recent sample; this completes their ray state (origin,
// direction, miss radiance) from the frame constants of that batch.  Must run while the scene the batch was rendered with is still
// resident (the sky decides the miss radiance): called by idkptDownloadRays and before a new scene replaces the old one.
static DScene make_dscene_last(dev_ctx* ctx);
static int materialize_culled_rays(dev_ctx* ctx)
{
    if (!ctx->lastNeedsRegen) return IDKPT_OK;
    const size_t N = (size_t)ctx->W * ctx->rows;
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    hipLaunchKernelGGL(k_regen_culled, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, make_dscene_last(ctx), ctx->lastFrame, rays, (const uint8_t*)ctx->contFlag.as<uint8_t>(), (uint32_t)(ctx->lastBatch - 1), (uint32_t)N);
    HIPC(hipGetLastError());
    ctx->lastNeedsRegen = false;
    return IDKPT_OK;
}

static int32_t dev_UploadScene(dev_ctx* ctx, const idkpt_scene_desc* sc)
{
    if (!ctx || !sc) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(sc->BlasNodes && sc->BlasNodeCount >= 4, "idkptUploadScene: BlasNodes missing");
    REQUIRE(sc->BlasTriangles && sc->BlasTriangleCount > 0, "idkptUploadScene: BlasTriangles missing");
    REQUIRE(sc->BlasDescs && sc->BlasDescCount > 0 && sc->BlasInstances && sc->BlasInstanceCount > 0, "idkptUploadScene: BlasDescs/BlasInstances missing");
    REQUIRE(sc->VertexPositions && sc->Vertices && sc->VertexCount > 0, "idkptUploadScene: vertices missing");
    REQUIRE(sc->Meshes && sc->MeshCount > 0 && sc->Materials && sc->MaterialCount > 0 && sc->MeshTransforms && sc->MeshTransformCount > 0, "idkptUploadScene: meshes/materials/transforms missing");
    REQUIRE(sc->LightCount >= 0 && sc->LightCount <= IDKPT_MAX_LIGHTS, "idkptUploadScene: more than 256 lights");
    // validate indices so that a bad host array cannot fault the GPU
    for (int i = 0; i < sc->BlasTriangleCount; i++) { const GpuBlasTriangle& t = sc->BlasTriangles[i]; REQUIRE(t.X < (uint32_t)sc->VertexCount && t.Y < (uint32_t)sc->VertexCount && t.Z < (uint32_t)sc->VertexCount && t.MeshId < (uint32_t)sc->MeshCount, "idkptUploadScene: BlasTriangle index out of range"); }
    for (int i = 0; i < sc->MeshCount; i++) REQUIRE(sc->Meshes[i].MaterialId >= 0 && sc->Meshes[i].MaterialId < sc->MaterialCount, "idkptUploadScene: Mesh.MaterialId out of range");
    for (int i = 0; i < sc->BlasInstanceCount; i++) REQUIRE(sc->BlasInstances[i].BlasId < (uint32_t)sc->BlasDescCount && sc->BlasInstances[i].MeshTransformId < (uint32_t)sc->MeshTransformCount, "idkptUploadScene: BlasInstance out of range");
    int maxStack = 1;
    { const char* why = validate_blas_nodes(sc->BlasNodes, sc->BlasNodeCount, sc->BlasDescs, sc->BlasDescCount, sc->BlasTriangleCount, true, &maxStack); REQUIRE(why == nullptr, std::string("idkptUploadScene: ") + (why ? why : "")); }
    REQUIRE(ctx->st.BlasStackSize == 0 || ctx->st.BlasStackSize >= maxStack, "idkptUploadScene: the BlasStackSize set with idkptSetSettings is smaller than this scene's maximum RequiredStackSize");
    int tlasNeed = 1;
    if (sc->TlasNodes && sc->TlasNodeCount > 0) {
        const char* why = nullptr;
        tlasNeed = tlas_validate(sc->TlasNodes, sc->TlasNodeCount, sc->BlasInstanceCount, &why);
        REQUIRE(tlasNeed >= 0, std::string("idkptUploadScene: ") + (why ? why : "bad TLAS"));
        REQUIRE(tlasNeed <= TLAS_STACK_SIZE, "idkptUploadScene: TLAS deeper than TLAS_STACK_SIZE (32)");
    }
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    { int rc = materialize_culled_rays(ctx); if (rc) return rc; }   // while the old sky is still resident
    int rc;
    if ((rc = upload(ctx, ctx->nodes, sc->BlasNodes, (size_t)sc->BlasNodeCount * 32))) return rc;
    if ((rc = upload(ctx, ctx->tris, sc->BlasTriangles, (size_t)sc->BlasTriangleCount * 16))) return rc;
    if ((rc = upload(ctx, ctx->descs, sc->BlasDescs, (size_t)sc->BlasDescCount * sizeof(GpuBlasDesc)))) return rc;
    if ((rc = upload(ctx, ctx->instances, sc->BlasInstances, (size_t)sc->BlasInstanceCount * 8))) return rc;
    if ((rc = upload(ctx, ctx->tlas, sc->TlasNodes, (size_t)(sc->TlasNodes ? sc->TlasNodeCount : 0) * 32))) return rc;
    if ((rc = upload(ctx, ctx->parents, sc->BlasParentIndices, (size_t)(sc->BlasParentIndices ? sc->BlasParentIndexCount : 0) * 4))) return rc;
    if ((rc = upload(ctx, ctx->leaves, sc->BlasLeafIndices, (size_t)(sc->BlasLeafIndices ? sc->BlasLeafIndexCount : 0) * 4))) return rc;
    if ((rc = upload(ctx, ctx->positions, sc->VertexPositions, (size_t)sc->VertexCount * 12))) return rc;
    if ((rc = upload(ctx, ctx->vertices, sc->Vertices, (size_t)sc->VertexCount * 16))) return rc;
    if ((rc = upload(ctx, ctx->meshes, sc->Meshes, (size_t)sc->MeshCount * sizeof(GpuMesh)))) return rc;
    if ((rc = upload(ctx, ctx->materials, sc->Materials, (size_t)sc->MaterialCount * sizeof(GpuMaterial)))) return rc;
    if ((rc = upload(ctx, ctx->xforms, sc->MeshTransforms, (size_t)sc->MeshTransformCount * sizeof(GpuMeshTransform)))) return rc;
    HIPC(ctx->lights.ensure(IDKPT_MAX_LIGHTS * sizeof(GpuLight)));
    if (sc->Lights && sc->LightCount) HIPC(hipMemcpyAsync(ctx->lights.p, sc->Lights, (size_t)sc->LightCount * sizeof(GpuLight), hipMemcpyHostToDevice, ctx->stream));
    {   // can a surface of this scene add radiance?  (k_shade_last: without emission the last bounce's hits leave the radiance alone)
        bool none = true;
        // (... and can the throughput a hit is shaded with be relied on to be finite when it came in finite?  Volumetric absorption multiplies it by exp(-absorbance x T)
        // BEFORE the emission is added (ShadeHit): an infinite Absorbance / AbsorbanceBias with T == 0 makes it NaN, and 0 x NaN is not 0 — such scenes shade every hit)
        auto fin = [](float v) { return v - v == 0.0f; };
        for (int i = 0; i < sc->MaterialCount && none; i++) { const GpuMaterial& m = sc->Materials[i]; none = m.EmissiveFactor[0] == 0.0f && m.EmissiveFactor[1] == 0.0f && m.EmissiveFactor[2] == 0.0f && fin(m.Absorbance[0]) && fin(m.Absorbance[1]) && fin(m.Absorbance[2]); }
        for (int i = 0; i < sc->MeshCount && none; i++) none = sc->Meshes[i].EmissiveBias == 0.0f && fin(sc->Meshes[i].AbsorbanceBias[0]) && fin(sc->Meshes[i].AbsorbanceBias[1]) && fin(sc->Meshes[i].AbsorbanceBias[2]);
        for (int i = 0; i < sc->TextureCount && none; i++) {      // (0 x a texel is only 0 for a finite texel)
            const idkpt_texture& t = sc->Textures[i];
            if (!(t.width > 0 && t.height > 0 && t.rgba)) { none = false; break; }
            none = tex_all_finite(t);
        }
        ctx->sceneNoEmission = none;
    }
    ctx->skySize = (sc->SkyFaces && sc->SkyFaceSize > 0) ? sc->SkyFaceSize : 0;
    if ((rc = upload(ctx, ctx->sky, sc->SkyFaces, (size_t)6 * ctx->skySize * ctx->skySize * 16))) return rc;
    for (auto& t : ctx->texData) t.release();
    ctx->texData.clear(); ctx->texDims.clear(); ctx->texState.clear();
    if ((rc = tex_srgb_lut(ctx))) return rc;
    for (int i = 0; i < sc->TextureCount; i++) {
        const idkpt_texture& t = sc->Textures[i];
        if ((rc = tex_validate(ctx, t, "idkptUploadScene"))) return rc;
        ctx->texData.emplace_back();
        if ((rc = upload(ctx, ctx->texData.back(), t.rgba, (size_t)t.width * t.height * tex_texel_bytes(t.format)))) return rc;
        ctx->texDims.push_back({t.width, t.height}); ctx->texState.push_back(tex_state(t));
    }
    if ((rc = tex_descs_upload(ctx))) return rc;
    HIPC(ctx->triVerts.ensure((size_t)sc->BlasTriangleCount * 48));
    ctx->nodeCount = sc->BlasNodeCount; ctx->triCount = sc->BlasTriangleCount; ctx->instanceCount = sc->BlasInstanceCount; ctx->tlasCount = sc->TlasNodes ? sc->TlasNodeCount : 0;
    ctx->vertexCount = sc->VertexCount; ctx->meshCount = sc->MeshCount; ctx->materialCount = sc->MaterialCount; ctx->xformCount = sc->MeshTransformCount;
    ctx->lightCount = sc->Lights ? sc->LightCount : 0; ctx->textureCount = sc->TextureCount;
    ctx->hDescs.assign(sc->BlasDescs, sc->BlasDescs + sc->BlasDescCount); ctx->hInst0Blas = (int)sc->BlasInstances[0].BlasId;
    ctx->hInstances.assign(sc->BlasInstances, sc->BlasInstances + sc->BlasInstanceCount);
    ctx->hXforms.assign((const char*)sc->MeshTransforms, (const char*)sc->MeshTransforms + (size_t)sc->MeshTransformCount * sizeof(GpuMeshTransform));
    ctx->sceneStack = maxStack; ctx->tlasNeed = std::max(1, tlasNeed);
    ctx->sceneNested = blas_nested(sc->BlasNodes, sc->BlasDescs, sc->BlasDescCount);
    ver_reset(ctx);                                   // one state per versioned buffer, in slot 0 (everything that read the old scene was launched by FLUSH above)
    if ((rc = ver_reserve(ctx))) return rc;
    // refit schedule: internal nodes of every refittable BLAS grouped by depth (children have larger ids than parents)
    ctx->levelOffsets.assign(sc->BlasDescCount, {}); ctx->levelBase.assign(sc->BlasDescCount, 0); ctx->refitCoversAll.assign(sc->BlasDescCount, 0);
    std::vector<int32_t> allLevels;
    for (int bi = 0; bi < sc->BlasDescCount; bi++) {
        const GpuBlasDesc& d = sc->BlasDescs[bi];
        if (!d.IsRefittable) continue;
        std::vector<int> depth(d.NodeCount, 0); int maxD = 0;
        for (int n = 1; n < d.NodeCount; n++) { const GpuBlasNode& nd = sc->BlasNodes[d.NodeOffset + n]; if (nd.TriCount == 0) { int c = (int)nd.TriStartOrChild; depth[c] = depth[c + 1] = depth[n] + 1; maxD = std::max(maxD, depth[n]); } }
        std::vector<std::vector<int32_t>> lv(maxD + 1);
        for (int n = 1; n < d.NodeCount; n++) if (sc->BlasNodes[d.NodeOffset + n].TriCount == 0 && (n == 1 || n >= 2)) lv[depth[n]].push_back(n);
        ctx->levelBase[bi] = (uint32_t)allLevels.size();
        uint32_t off = 0;
        for (auto& l : lv) { ctx->levelOffsets[bi].push_back(off); off += (uint32_t)l.size(); allLevels.insert(allLevels.end(), l.begin(), l.end()); }
        ctx->levelOffsets[bi].push_back(off);
        ctx->refitCoversAll[bi] = (int64_t)off + (int64_t)d.LeafIndicesCount == (int64_t)d.NodeCount - 1;
    }
    if ((rc = upload(ctx, ctx->levelNodes, allLevels.data(), allLevels.size() * 4))) return rc;
    if ((rc = regather_triverts(ctx, 0, (uint32_t)sc->BlasTriangleCount))) return rc;
    HIPC(hipStreamSynchronize(ctx->stream)); // host arrays are only borrowed for the duration of the call
    ctx->haveScene = true;
    std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);
    return IDKPT_OK;
}

// Device-to-device copies of a multi-device context.  xGMI peer copies (hipMemcpyPeerAsync, ordered on `st`) where the runtime grants them; on a
// node whose GPUs refuse peer access — or under the option "force_no_peer" — every copy is staged through pinned host memory instead: wait for
// `st` (so that what the stream order promised about the source holds), D2H on the source device, H2D on the destination device, both blocking.
// Slower (two PCIe crossings and a host synchronisation), same results; the first refusal is logged once.
struct PeerPolicy { bool forceStaged = false, warned = false; void* stage = nullptr; size_t stageBytes = 0; };
static hipError_t member_copy(PeerPolicy* pol, void* dst, int dstDev, const void* src, int srcDev, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    if (!pol || !pol->forceStaged) {
        hipError_t e = hipMemcpyPeerAsync(dst, dstDev, src, srcDev, bytes, st);
        if (e == hipSuccess || !pol) return e;
        (void)hipGetLastError();
        if (!pol->warned) { fprintf(stderr, "[idkpt] warning: peer copy GPU %d -> GPU %d refused (%s); staging device-to-device copies through host memory from now on\n", srcDev, dstDev, hipGetErrorString(e)); pol->warned = true; }
        pol->forceStaged = true;
    }
    hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) return e;
    if (pol->stageBytes < bytes) {
        if (pol->stage) (void)hipHostFree(pol->stage);
        pol->stage = nullptr; pol->stageBytes = 0;
        e = hipHostMalloc(&pol->stage, bytes, hipHostMallocDefault); if (e != hipSuccess) return e;
        pol->stageBytes = bytes;
    }
    int cur = 0; (void)hipGetDevice(&cur);
    e = hipSetDevice(srcDev); if (e == hipSuccess) e = hipMemcpy(pol->stage, src, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipSetDevice(dstDev);
    if (e == hipSuccess) e = hipMemcpy(dst, pol->stage, bytes, hipMemcpyHostToDevice);
    (void)hipSetDevice(cur);
    return e;
}

// Multi-device contexts: the scene one member uploaded (validated, derived layouts built) is replicated to another member device-to-device
// (hipMemcpyPeerAsync: xGMI between MI355X GPUs) instead of crossing PCIe once per GPU — the "broadcast of the BVH" of the group layer.
// Three steps so that the copies themselves can be the member-to-member peer copies below or one RCCL broadcast per buffer over all members (idkpt_api.hpp):
// clone_prepare sizes this member's buffers and lists (destination, source on `src`, bytes); the caller moves the bytes; clone_finish adopts the scene.
struct CloneItem { void* dst; const void* src; size_t bytes; };
static int clone_prepare(dev_ctx* ctx, dev_ctx* src, std::vector<CloneItem>& items)
{
    if (!ctx || !src || !src->haveScene) return IDKPT_ERR_INVALID_ARGUMENT;
    items.clear();
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    { int rc = materialize_culled_rays(ctx); if (rc) return rc; }   // while the old sky is still resident
    DevBuf* d[] = {&ctx->nodes, &ctx->tris, &ctx->triVerts, &ctx->descs, &ctx->instances, &ctx->tlas, &ctx->parents, &ctx->leaves, &ctx->positions, &ctx->vertices, &ctx->meshes,
                   &ctx->materials, &ctx->xforms, &ctx->lights, &ctx->sky, &ctx->levelNodes};
    DevBuf* f[] = {&src->nodes, &src->tris, &src->triVerts, &src->descs, &src->instances, &src->tlas, &src->parents, &src->leaves, &src->positions, &src->vertices, &src->meshes,
                   &src->materials, &src->xforms, &src->lights, &src->sky, &src->levelNodes};
    for (size_t i = 0; i < sizeof(d) / sizeof(d[0]); i++) {
        if (!f[i]->p || f[i]->bytes == 0) continue;
        // a versioned buffer of the source may be an arena of several states: its current one goes to slot 0 here
        int vb = -1; for (int b = 0; b < VB_COUNT; b++) if (f[i] == &vb_buf(src, b)) vb = b;
        const size_t bytes = (vb >= 0 && src->vbytes[vb] > 0) ? src->vbytes[vb] : f[i]->bytes;
        const char* from = (vb >= 0 && src->vbytes[vb] > 0) ? vb_ptr(src, vb, src->vcur[vb]) : (const char*)f[i]->p;
        HIPC(d[i]->ensure(bytes));
        items.push_back({d[i]->p, from, bytes});
    }
    for (auto& t : ctx->texData) t.release();
    ctx->texData.clear(); ctx->texDims = src->texDims; ctx->texState = src->texState;
    for (size_t i = 0; i < src->texData.size(); i++) {
        ctx->texData.emplace_back();
        HIPC(ctx->texData.back().ensure(src->texData[i].bytes));
        items.push_back({ctx->texData.back().p, src->texData[i].p, src->texData[i].bytes});
    }
    return IDKPT_OK;
}
static int clone_finish(dev_ctx* ctx, dev_ctx* src)
{
    HIPC(hipSetDevice(ctx->device));
    { int rc = tex_srgb_lut(ctx); if (rc) return rc; }
    { int rc = tex_descs_upload(ctx); if (rc) return rc; }
    ctx->nodeCount = src->nodeCount; ctx->triCount = src->triCount; ctx->instanceCount = src->instanceCount; ctx->tlasCount = src->tlasCount; ctx->vertexCount = src->vertexCount;
    ctx->meshCount = src->meshCount; ctx->materialCount = src->materialCount; ctx->xformCount = src->xformCount; ctx->lightCount = src->lightCount; ctx->skySize = src->skySize;
    ctx->textureCount = src->textureCount; ctx->hDescs = src->hDescs; ctx->hInstances = src->hInstances; ctx->hXforms = src->hXforms; ctx->hInst0Blas = src->hInst0Blas; ctx->sceneNoEmission = src->sceneNoEmission; ctx->sceneNested = src->sceneNested; ctx->sceneStack = src->sceneStack; ctx->tlasNeed = src->tlasNeed;
    ctx->levelOffsets = src->levelOffsets; ctx->levelBase = src->levelBase; ctx->refitCoversAll = src->refitCoversAll;
    ver_reset(ctx);
    { int rc = ver_reserve(ctx); if (rc) return rc; }
    HIPC(hipStreamSynchronize(ctx->stream));           // td is a stack vector
    ctx->haveScene = true;
    std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);
    return IDKPT_OK;
}

// one member from another by peer copies (xGMI: hipMemcpyPeerAsync; staged through the host where peer access is refused)
static int32_t dev_CloneSceneFrom(dev_ctx* ctx, dev_ctx* src)
{
    std::vector<CloneItem> items;
    int rc = clone_prepare(ctx, src, items); if (rc) return rc;
    for (const CloneItem& it : items) HIPC(member_copy(ctx->peer, it.dst, ctx->device, it.src, src->device, it.bytes, ctx->stream));
    return clone_finish(ctx, src);
}

// idkptUpdateTexture: image `index` of the table gets new contents / size / format / sampler state.  The shading kernels of every queued sample read the table: they are launched first.
static int32_t dev_UpdateTexture(dev_ctx* ctx, int32_t index, const idkpt_texture* t)
{
    if (!ctx || !t) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptUpdateTexture: no scene uploaded");
    REQUIRE(index >= 0 && index < ctx->textureCount, "idkptUpdateTexture: index out of range");
    { int rc = tex_validate(ctx, *t, "idkptUpdateTexture"); if (rc) return rc; }
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));                                        // (the old image may be released below: nothing in flight reads it any more)
    { int rc = upload(ctx, ctx->texData[index], t->rgba, (size_t)t->width * t->height * tex_texel_bytes(t->format)); if (rc) return rc; }
    HIPC(hipStreamSynchronize(ctx->stream));                                        // (the host's array is borrowed for the call only)
    ctx->texDims[index] = {t->width, t->height}; ctx->texState[index] = tex_state(*t);
    if (!tex_all_finite(*t)) ctx->sceneNoEmission = false;                          // (conservative: the shortcut stays off until the next idkptUploadScene)
    return tex_descs_upload(ctx);
}

static int32_t dev_SetLightCount(dev_ctx* ctx, int32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(count >= 0 && count <= IDKPT_MAX_LIGHTS, "idkptSetLightCount: out of range");
    FLUSH();
    ctx->lightCount = count; return IDKPT_OK;
}

// buffer id -> allocation, size of its state, and (versioned buffers) which arena it is (-1: a plain buffer)
static DevBuf* which_buffer(dev_ctx* ctx, int which, size_t* cap, int* vb)
{
    *vb = -1;
    switch (which) {
        case IDKPT_BUF_MESH_TRANSFORMS: *cap = (size_t)ctx->xformCount * sizeof(GpuMeshTransform); *vb = VB_XFORMS; return &ctx->xforms;
        case IDKPT_BUF_VERTEX_POSITIONS: *cap = (size_t)ctx->vertexCount * 12; return &ctx->positions;
        case IDKPT_BUF_VERTICES: *cap = (size_t)ctx->vertexCount * 16; *vb = VB_VERTICES; return &ctx->vertices;
        case IDKPT_BUF_MESHES: *cap = (size_t)ctx->meshCount * sizeof(GpuMesh); return &ctx->meshes;
        case IDKPT_BUF_MATERIALS: *cap = (size_t)ctx->materialCount * sizeof(GpuMaterial); return &ctx->materials;
        case IDKPT_BUF_LIGHTS: *cap = (size_t)IDKPT_MAX_LIGHTS * sizeof(GpuLight); return &ctx->lights;
        case IDKPT_BUF_BLAS_NODES: *cap = (size_t)ctx->nodeCount * 32; *vb = VB_NODES; return &ctx->nodes;
        case IDKPT_BUF_TLAS_NODES: *cap = (size_t)ctx->tlasCount * 32; *vb = VB_TLAS; return &ctx->tlas;
        case IDKPT_BUF_JOINT_MATRICES: *cap = ctx->joints.bytes; return &ctx->joints;
        default: return nullptr;
    }
}

static int32_t dev_UpdateBuffer(dev_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, const void* data)
{
    if (!ctx || !data) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptUpdateBuffer: no scene uploaded");
    HIPC(hipSetDevice(ctx->device));
    // What queued samples still have to read decides whether they are launched first: joint matrices and vertex positions are read by the update kernels only
    // (k_skin, k_gather_triverts: they run in stream order, right now); transforms, vertices and tree nodes are versioned (ver_writable finds or makes room);
    // meshes, materials and lights are read by the shading kernels of every queued sample.
    if (which == IDKPT_BUF_MESHES || which == IDKPT_BUF_MATERIALS || which == IDKPT_BUF_LIGHTS || which == IDKPT_BUF_BLAS_NODES || which == IDKPT_BUF_TLAS_NODES) FLUSH();
    if (which == IDKPT_BUF_JOINT_MATRICES) { size_t need = offsetBytes + bytes; if (need > ctx->joints.bytes) { DevBuf nb; HIPC(nb.ensure(need)); if (ctx->joints.p) { HIPC(hipMemcpyAsync(nb.p, ctx->joints.p, ctx->joints.bytes, hipMemcpyDeviceToDevice, ctx->stream)); } HIPC(hipStreamSynchronize(ctx->stream)); ctx->joints.release(); ctx->joints = nb; } }
    size_t cap = 0; int vb = -1; DevBuf* b = which_buffer(ctx, which, &cap, &vb);
    REQUIRE(b != nullptr, "idkptUpdateBuffer: unknown buffer");
    REQUIRE(offsetBytes + bytes <= cap, "idkptUpdateBuffer: range exceeds buffer");
    if (which == IDKPT_BUF_BLAS_NODES || which == IDKPT_BUF_TLAS_NODES) {
        // Patched tree nodes are validated like uploaded ones BEFORE they reach the device (a bad child index must not fault the GPU, a deeper tree must
        // not overflow the traversal stack), on a host copy of the array with the patch applied; BLAS nodes: the derived order is rebuilt from that copy.
        // The patch must leave a valid tree after EVERY call (idkpt.h): a host that streams a rebuilt tree in pieces uses idkptUploadScene / idkptBuildTlas.
        char* cur = vb_ptr(ctx, vb, ctx->vcur[vb]);           // (nothing is queued or deferred any more: the patch goes in place)
        std::vector<char> h(cap);
        HIPC(hipMemcpyAsync(h.data(), cur, cap, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
        memcpy(h.data() + offsetBytes, data, bytes);
        if (which == IDKPT_BUF_BLAS_NODES) {
            int maxStack = 1;
            const char* why = validate_blas_nodes((const GpuBlasNode*)h.data(), ctx->nodeCount, ctx->hDescs.data(), (int)ctx->hDescs.size(), ctx->triCount, false, &maxStack);
            REQUIRE(why == nullptr, std::string("idkptUpdateBuffer: ") + (why ? why : ""));
            REQUIRE(ctx->st.BlasStackSize == 0 || ctx->st.BlasStackSize >= maxStack, "idkptUpdateBuffer: the patched BLAS needs a deeper traversal stack than the BlasStackSize set with idkptSetSettings");
            HIPC(hipMemcpyAsync(cur + offsetBytes, data, bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->sceneStack = maxStack; ctx->pairValid = false; ctx->wideTopoValid = false; ctx->wideFillValid = false; ctx->itlasValid = false; ctx->instRecValid = false; ctx->imarksValid = false;   // (the tree itself may have changed: the wide nodes, the library's own TLAS and the triangle marks are derived anew)
            ctx->sceneNested = blas_nested((const GpuBlasNode*)h.data(), ctx->hDescs.data(), (int)ctx->hDescs.size());
        } else {
            const char* why = nullptr;
            const int need = tlas_validate((const GpuTlasNode*)h.data(), ctx->tlasCount, ctx->instanceCount, &why);
            REQUIRE(need >= 0, std::string("idkptUpdateBuffer: ") + (why ? why : "bad TLAS"));
            REQUIRE(need <= TLAS_STACK_SIZE, "idkptUpdateBuffer: TLAS deeper than TLAS_STACK_SIZE (32)");
            HIPC(hipMemcpyAsync(cur + offsetBytes, data, bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->tlasNeed = std::max(1, need);
        }
        HIPC(hipStreamSynchronize(ctx->stream));
        return IDKPT_OK;
    }
    if (which == IDKPT_BUF_MESHES || which == IDKPT_BUF_MATERIALS) ctx->sceneNoEmission = false;   // (a patched material may emit: decided again at the next idkptUploadScene)
    char* dst = (char*)b->p;
    if (vb >= 0) { char* src; int rc = ver_writable(ctx, vb, offsetBytes == 0 && bytes == cap, &src, &dst); if (rc) return rc; }
    { int rc = staged_upload(ctx, dst + offsetBytes, data, bytes); if (rc) return rc; }   // (small updates — joints, transforms — do not wait for the stream)
    if (which == IDKPT_BUF_MESH_TRANSFORMS && offsetBytes + bytes <= ctx->hXforms.size()) memcpy(ctx->hXforms.data() + offsetBytes, data, bytes);   // (the host's copy: "do all instances share one InvModel", host_launch.hpp)
    if (which == IDKPT_BUF_VERTEX_POSITIONS) { int rc = regather_triverts(ctx, 0, (uint32_t)ctx->triCount); if (rc) return rc; }
    return IDKPT_OK;
}

static int32_t dev_DownloadBuffer(dev_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, void* dst)
{
    if (!ctx || !dst) return IDKPT_ERR_INVALID_ARGUMENT;
    if (which == IDKPT_BUF_WIDE_NODES || which == IDKPT_BUF_WIDE_LEAVES || which == IDKPT_BUF_WIDE_COUNTS) {   // read-only views of the derived traversal structure (tests, tools)
        if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptDownloadBuffer: no scene uploaded");
        HIPC(hipSetDevice(ctx->device));
        FLUSH();
        { int rc = wide_prepare(ctx); if (rc) return rc; }
        DevBuf& wbuf = which == IDKPT_BUF_WIDE_NODES ? ctx->wnodes : (which == IDKPT_BUF_WIDE_LEAVES ? ctx->wleaf : ctx->wcounts);
        REQUIRE(offsetBytes + bytes <= (which == IDKPT_BUF_WIDE_COUNTS ? ctx->hDescs.size() * 8 : wbuf.bytes), "idkptDownloadBuffer: bad buffer/range");
        HIPC(hipMemcpyAsync(dst, (char*)wbuf.p + offsetBytes, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));
        return IDKPT_OK;
    }
    size_t cap = 0; int vb = -1; DevBuf* b = which_buffer(ctx, which, &cap, &vb);
    REQUIRE(b != nullptr && offsetBytes + bytes <= cap, "idkptDownloadBuffer: bad buffer/range");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipMemcpyAsync(dst, (vb >= 0 ? vb_ptr(ctx, vb, ctx->vcur[vb]) : (char*)b->p) + offsetBytes, bytes, hipMemcpyDeviceToHost, ctx->stream));   // the current state
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

static int32_t dev_BuildTlas(dev_ctx* ctx, const GpuTlasNode* nodes, int32_t nodeCount)
{
    if (!ctx || !nodes || nodeCount <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildTlas: no scene uploaded");
    const char* why = nullptr;
    const int need = tlas_validate(nodes, nodeCount, ctx->instanceCount, &why);
    REQUIRE(need >= 0, std::string("idkptBuildTlas: ") + (why ? why : "bad TLAS"));
    REQUIRE(need <= TLAS_STACK_SIZE, "idkptBuildTlas: TLAS deeper than TLAS_STACK_SIZE (32)");
    HIPC(hipSetDevice(ctx->device));
    if ((size_t)nodeCount * 32 > ctx->vbytes[VB_TLAS] || nodeCount != ctx->tlasCount) {
        // another node count (or more nodes than a slot holds): every queued sample is launched first, then the TLAS buffer is laid out anew
        FLUSH();
        if ((size_t)nodeCount * 32 > ctx->vbytes[VB_TLAS]) {
            HIPC(hipStreamSynchronize(ctx->stream));
            ctx->tlas.release();
            ctx->vbytes[VB_TLAS] = (size_t)nodeCount * 32; ctx->vstride[VB_TLAS] = (ctx->vbytes[VB_TLAS] + 255) / 256 * 256; ctx->valloc[VB_TLAS] = 1; ctx->vcur[VB_TLAS] = 0;
            HIPC(ctx->tlas.ensure(ctx->vstride[VB_TLAS]));
        }
    }
    char *src, *dst; int rc = ver_writable(ctx, VB_TLAS, true, &src, &dst); if (rc) return rc;
    HIPC(hipMemcpyAsync(dst, nodes, (size_t)nodeCount * 32, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->tlasCount = nodeCount; ctx->tlasNeed = std::max(1, need);
    return IDKPT_OK;
}

static int32_t dev_BuildTlasOnDevice(dev_ctx* ctx, int32_t searchRadius)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildTlasOnDevice: no scene uploaded");
    REQUIRE(searchRadius >= 1, "idkptBuildTlasOnDevice: searchRadius must be >= 1 (reference: 15)");
    HIPC(hipSetDevice(ctx->device));
    const int n = ctx->instanceCount, nodeCount = 2 * n - 1;
    if (nodeCount != ctx->tlasCount) FLUSH();            // (queued samples were queued with another node count; a slot always has room for 2n - 1 nodes: ver_reset)
    char *tsrc, *tdst; { int rc = ver_writable(ctx, VB_TLAS, true, &tsrc, &tdst); if (rc) return rc; }
    // scratch: temp nodes (2n-1) + leaves (n) as float4 pairs, keys (n), pref (n)
    const size_t tempOff = 0, leafOff = (size_t)nodeCount * 32, keyOff = leafOff + (size_t)n * 32, prefOff = keyOff + (size_t)n * 4;
    HIPC(ctx->tlasScratch.ensure(prefOff + (size_t)n * 4));
    char* sc = ctx->tlasScratch.as<char>();
    hipLaunchKernelGGL(k_tlas_build, dim3(1), dim3(TLAS_BUILD_THREADS), 0, ctx->stream, (const float4*)vb_cur<float4>(ctx, VB_NODES), ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(),
                       (const float4*)vb_cur<float4>(ctx, VB_XFORMS), n, (int)searchRadius, (float4*)tdst, (float4*)(sc + tempOff), (float4*)(sc + leafOff), (uint32_t*)(sc + keyOff), (int*)(sc + prefOff));
    HIPC(hipGetLastError());
    ctx->tlasCount = nodeCount; ctx->tlasNeed = std::min(TLAS_STACK_SIZE, std::max(1, n));   // depth unknown on the host: all rows a tree over n leaves can need, up to the limit (beyond it: overflow flag)
    return IDKPT_OK;
}

static int32_t dev_RefitBlas(dev_ctx* ctx, int32_t blasId)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRefitBlas: no scene uploaded");
    REQUIRE(blasId >= 0 && blasId < (int)ctx->hDescs.size(), "idkptRefitBlas: blasId out of range");
    const GpuBlasDesc& d = ctx->hDescs[blasId];
    if (!d.IsRefittable || d.LeafIndicesCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRefitBlas: BLAS is not refittable (no leaf/parent indices)");
    HIPC(hipSetDevice(ctx->device));
    // (no launch of the queued samples: the triangle records and the nodes are rewritten in slots they do not read, ver_writable)
    int rc = regather_triverts(ctx, (uint32_t)d.TriangleOffset, (uint32_t)d.TriangleCount); if (rc) return rc;
    // The refit writes every node of this BLAS but its unused node 0 (leaves, then the internal nodes level by level); topology words (.w) are read from the
    // state being replaced, child boxes from the state being written.  A scene that is this one BLAS needs no copy of the old state (node 0 aside).
    const bool whole = ctx->hDescs.size() == 1 && d.NodeOffset == 0 && d.NodeCount == ctx->nodeCount && ctx->refitCoversAll[blasId];
    char *nsrc, *ndst; rc = ver_writable(ctx, VB_NODES, whole, &nsrc, &ndst); if (rc) return rc;
    if (whole && nsrc != ndst) HIPC(hipMemcpyAsync(ndst, nsrc, 32, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_refit_leaves, dim3((d.LeafIndicesCount + 63) / 64), dim3(64), 0, ctx->stream, (const float4*)nsrc, (float4*)ndst, ctx->tris.as<uint4>(), (const float4*)vb_cur<float4>(ctx, VB_TRIVERTS),
                       ctx->leaves.as<int32_t>() + d.LeafIndicesOffset, (uint32_t)d.LeafIndicesCount, (uint32_t)d.NodeOffset, (uint32_t)d.TriangleOffset);
    const std::vector<uint32_t>& off = ctx->levelOffsets[blasId];
    for (int l = (int)off.size() - 2; l >= 0; l--) {
        uint32_t cnt = off[l + 1] - off[l];
        if (!cnt) continue;
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 63) / 64), dim3(64), 0, ctx->stream, (const float4*)nsrc, (float4*)ndst, ctx->levelNodes.as<int32_t>() + ctx->levelBase[blasId] + off[l], cnt, (uint32_t)d.NodeOffset);
    }
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

static int32_t dev_UploadUnskinnedVertices(dev_ctx* ctx, const GpuUnskinnedVertex* verts, int32_t count)
{
    if (!ctx || !verts || count <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    int rc = upload(ctx, ctx->unskinned, verts, (size_t)count * sizeof(GpuUnskinnedVertex)); if (rc) return rc;
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->unskinnedCount = count;
    return IDKPT_OK;
}

static int32_t dev_Skin(dev_ctx* ctx, uint32_t inOff, uint32_t outOff, uint32_t jointOff, uint32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene || ctx->unskinnedCount == 0 || ctx->joints.bytes == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptSkin: needs scene, unskinned vertices and joint matrices");
    REQUIRE((uint64_t)inOff + count <= (uint64_t)ctx->unskinnedCount && (uint64_t)outOff + count <= (uint64_t)ctx->vertexCount, "idkptSkin: range out of bounds");
    HIPC(hipSetDevice(ctx->device));
    HIPC(ctx->prevPositions.ensure((size_t)ctx->vertexCount * 12));
    // positions are read by update kernels only (stream order); the re-compressed normals / tangents go to a vertex slot no queued sample reads (ver_writable)
    char *vsrc, *vdst; { int rc = ver_writable(ctx, VB_VERTICES, outOff == 0 && count == (uint32_t)ctx->vertexCount, &vsrc, &vdst); if (rc) return rc; }
    if (count) hipLaunchKernelGGL(k_skin, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, ctx->unskinned.as<GpuUnskinnedVertex>(), ctx->joints.as<float4>(), ctx->positions.as<float>(),
                                  ctx->prevPositions.as<float>(), (const uint4*)vsrc, (uint4*)vdst, inOff, outOff, jointOff, count);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

static int32_t dev_ResetAccumulation(dev_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->accum[ctx->curSlot] = 0; return IDKPT_OK; }
// Sample-parallel rendering: context r of N renders the reference's samples r, r + N, r + 2N, ... (their RNG streams), each context accumulating
// its own running mean; the mean of the N accumulations is an accumulation over N * K distinct reference samples.
static int32_t dev_SetSampleSequence(dev_ctx* ctx, uint32_t first, uint32_t stride)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(stride >= 1, "idkptSetSampleSequence: stride must be >= 1");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    if (first == ctx->seqFirst && stride == ctx->seqStride) return IDKPT_OK;
    ctx->seqFirst = first; ctx->seqStride = stride;
    std::fill(ctx->accum.begin(), ctx->accum.end(), 0u);           // other RNG streams: the accumulation starts over
    return IDKPT_OK;
}
static int32_t dev_GetAccumulatedSamples(dev_ctx* ctx, uint32_t* out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = ctx->accum[ctx->curSlot]; return IDKPT_OK; }
