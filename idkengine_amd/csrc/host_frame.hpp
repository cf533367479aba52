// host_frame.hpp — wavefront / image buffers of a frame size, the flush rules, scene versions (ver_*), staged small uploads.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

static int alloc_frame_impl(dev_ctx* ctx)
{
    const size_t N = (size_t)ctx->W * ctx->rows;
    ctx->Npad = (uint32_t)((N + 63) / 64 * 64);
    const size_t cap = (size_t)ctx->maxBatch * ctx->Npad;   // ray ids of one batch
    ctx->pending.clear();
    ctx->lastFast = false; ctx->lastNeedsRegen = false; ctx->lastBatch = 1;   // nothing rendered into the new buffers yet
    HIPC(ctx->rayO.ensure(cap * 16)); HIPC(ctx->rayT.ensure(cap * 16)); HIPC(ctx->rayR.ensure(cap * 16));
    HIPC(ctx->aovA.ensure(cap * 16)); HIPC(ctx->aovN.ensure(cap * 16));
    HIPC(ctx->trRec.ensure(cap * 64)); HIPC(ctx->contFlag.ensure(cap));
    HIPC(ctx->blockSums.ensure(((cap + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK * 4 + 16));
    HIPC(ctx->hit.ensure(cap * 32)); HIPC(ctx->hitCost.ensure(cap * 4));
    for (int i = 0; i < 2; i++) { HIPC(ctx->queue[i].ensure(cap * 4)); HIPC(ctx->keys[i].ensure(cap * 4)); }
    HIPC(ctx->keysTmp.ensure(cap * 4)); HIPC(ctx->sortKeys.ensure(cap * 4)); HIPC(ctx->sortVals.ensure(cap * 4));
    size_t nW = (cap + 63) / 64;
    HIPC(ctx->contMask.ensure(nW * 8)); HIPC(ctx->waveCounts.ensure(nW * 4));
    HIPC(ctx->counts.ensure(MAX_DEPTH_SLOTS * 4)); HIPC(ctx->work.ensure(WORK_WORDS * 4)); HIPC(ctx->counters64.ensure(128));
    HIPC(ctx->bases.ensure((size_t)MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4));
    size_t nTiles = (cap + SORT_TILE - 1) / SORT_TILE;
    HIPC(ctx->sortHist.ensure((SORT_RADIX * nTiles + SORT_RADIX) * 4));
    for (int i = 0; i < 3; i++) { HIPC(ctx->img[i].ensure(N * 16 * ctx->ringSize)); HIPC(hipMemsetAsync(ctx->img[i].p, 0, N * 16 * ctx->ringSize, ctx->stream)); }   // slot s at offset s*N
    HIPC(hipMemsetAsync(ctx->counters64.p, 0, 128, ctx->stream));
    ctx->defer.valid = false;      // (callers complete a deferred last bounce before they get here; whatever is left refers to buffers that are gone)
    ctx->countersDirty = true;   // (the first batch resets its counters itself)
    HIPC(hipMemsetAsync(ctx->aovA.p, 0, cap * 16, ctx->stream)); HIPC(hipMemsetAsync(ctx->aovN.p, 0, cap * 16, ctx->stream));
    HIPC(hipMemsetAsync(ctx->contFlag.p, 0, cap, ctx->stream));   // per-batch values are written by k_gen_primary; the pad ids [N, Npad) must read 0
    ctx->accum.assign(ctx->ringSize, 0u); ctx->curSlot = 0; ctx->ringStarted = false;
    return IDKPT_OK;
}

// a failed allocation leaves the context without a usable frame (idkptRender refuses) until a later idkptSetSize / idkptSetMaxBatch succeeds
static int alloc_frame(dev_ctx* ctx) { int rc = alloc_frame_impl(ctx); ctx->frameOk = rc == IDKPT_OK; return rc; }

// maxBatch changed: the wavefront buffers grow, the accumulation images (and their contents) stay
static int alloc_frame_keep_images(dev_ctx* ctx)
{
    const size_t N = (size_t)ctx->W * ctx->rows;
    DevBuf saved[3];
    for (int i = 0; i < 3; i++) { saved[i] = ctx->img[i]; ctx->img[i] = DevBuf(); }
    int rc = alloc_frame(ctx);
    // The restore is ordered on the context's stream, behind the zero-fill alloc_frame_impl queued there: the stream is non-blocking, so
    // a null-stream copy would be unordered against that fill (the fill could land after the restore and wipe the accumulation).
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && rc == IDKPT_OK && e == hipSuccess; i++)
        if (saved[i].p) e = hipMemcpyAsync(ctx->img[i].p, saved[i].p, std::min(saved[i].bytes, N * 16 * ctx->ringSize), hipMemcpyDeviceToDevice, ctx->stream);
    if (rc == IDKPT_OK && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the saved buffers are released right below
    for (int i = 0; i < 3; i++) saved[i].release();
    if (rc == IDKPT_OK && e != hipSuccess) { (void)hipGetLastError(); ctx->frameOk = false; return fail(ctx, IDKPT_ERR_HIP, std::string("alloc_frame_keep_images: ") + hipGetErrorString(e)); }
    return rc;
}

static int flush_batch(dev_ctx* ctx);
// After a stream synchronisation: did any traversal drop a stack push?  (Cannot happen for scenes that passed idkptUploadScene's
// validation with BlasStackSize >= the computed need; the flag is the safety net for buffers patched later with idkptUpdateBuffer
// and for device-built TLASes deeper than TLAS_STACK_SIZE.)  The results of the affected batch are invalid: report, never return them silently.
static int check_overflow(dev_ctx* ctx)
{
    if (!ctx->hOverflow || *(volatile uint32_t*)ctx->hOverflow == 0u) return IDKPT_OK;
    *(volatile uint32_t*)ctx->hOverflow = 0u;
    return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "traversal stack overflow: a BLAS/TLAS is deeper than the traversal stack (BlasStackSize / TLAS_STACK_SIZE); results since the last synchronisation are invalid");
}
#define SYNC_CHECKED() do { HIPC(hipStreamSynchronize(ctx->stream)); int _rc = check_overflow(ctx); if (_rc) return _rc; } while (0)
static int finish_deferred(dev_ctx* ctx);
// FLUSH: issue the samples still queued, and complete a deferred last bounce (readers of ray state / queues, and everything that changes what its kernels would read).
// FLUSH_KEEP: issue only (images, synchronisation, camera, statistics, batching knobs: nothing that looks at or invalidates the deferred part).
// (a member of a multi-device context never launches on its own: the group launches what ALL members have queued, idkpt_api.hpp group_flush_all)
static int flush_any(dev_ctx* ctx);
#define FLUSH() do { int _rc = flush_any(ctx); if (_rc) return _rc; _rc = finish_deferred(ctx); if (_rc) return _rc; } while (0)
#define FLUSH_KEEP() do { int _rc = flush_any(ctx); if (_rc) return _rc; } while (0)

// ---- scene versions -------------------------------------------------------------------------------------------------------------------------------
// A scene update (skinning, refit, TLAS rebuild, a patched transform) used to launch every queued sample first: the kernels of a batch read "the" scene, so
// an animated host rendered one frame at a time (a quarter of the batched rate).  With idkptSetSceneVersions(n > 1) the buffers those updates write — and the
// render kernels read — are arenas of up to n slots.  An update that would overwrite a slot which queued samples (or the deferred last bounce of the launched
// batch) still read moves the CURRENT state of that buffer to a free slot first (copying it unless the update rewrites all of it) and writes there; updates run
// eagerly in stream order, rendering stays deferred, and a batch whose samples saw different slots gets a per-sample table (DScene::ver, VER kernels).
// Everything is on the context's one stream, so kernels already launched are ordered before any later write: only unlaunched samples pin a slot.
// With one slot (the default) the first branch below never finds room and falls through to "launch what is queued": exactly the old behaviour.
static int flush_any(dev_ctx* ctx)
{
    if (ctx->pending.empty()) return IDKPT_OK;
    if (!(ctx->grouped && !ctx->inGroupFlush && ctx->groupFlushAll)) return flush_batch(ctx);
    // the group launches every member's batch and leaves some other member's device current: the caller (an update of THIS member: hipMalloc,
    // event creation, kernel launches follow) continues on its own device
    const int rc = ctx->groupFlushAll(ctx->groupUser);
    const hipError_t e = hipSetDevice(ctx->device);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, IDKPT_ERR_HIP, std::string("hipSetDevice after the group flush: ") + hipGetErrorString(e));
    return IDKPT_OK;
}
static DevBuf& vb_buf(dev_ctx* ctx, int b)
{
    switch (b) { case VB_NODES: return ctx->nodes; case VB_TRIVERTS: return ctx->triVerts; case VB_VERTICES: return ctx->vertices; case VB_TLAS: return ctx->tlas; default: return ctx->xforms; }
}
static char* vb_ptr(dev_ctx* ctx, int b, int slot) { return (char*)vb_buf(ctx, b).p + (size_t)slot * ctx->vstride[b]; }
template <class T> static T* vb_cur(dev_ctx* ctx, int b) { return (T*)vb_ptr(ctx, b, ctx->vcur[b]); }
// after idkptUploadScene / a clone / a re-derived node order: one state per buffer, in slot 0 of whatever allocation the buffer has
static void ver_reset(dev_ctx* ctx)
{
    ctx->wideTopoValid = false; ctx->wideFillValid = false; ctx->itlasValid = false; ctx->instRecValid = false; ctx->imarksValid = false; ctx->ichunkCount = 0; ctx->instOverlapKnown = false; ctx->itlasBuilt = false; ctx->uniValid = false; ctx->uniTabsValid = false;
    ctx->pairValid = false;
    ctx->pkState = 0; ctx->pkBatchesSinceProbe = 0;      // (another scene: the packet walk's decision is probed again, packet_decide)
    const size_t one[VB_COUNT] = {(size_t)ctx->nodeCount * 32, (size_t)ctx->triCount * 48, (size_t)ctx->vertexCount * 16,
                                  (size_t)std::max(ctx->tlasCount, 2 * ctx->instanceCount - 1) * 32, (size_t)ctx->xformCount * sizeof(GpuMeshTransform)};
    for (int b = 0; b < VB_COUNT; b++) { ctx->vbytes[b] = one[b]; ctx->vstride[b] = (one[b] + 255) / 256 * 256; ctx->valloc[b] = 1; ctx->vcur[b] = 0; ctx->lastMask[b] = 0; ctx->lastSlots[b] = 0; }
    ctx->lastMulti = false;
}
// ... and every buffer really holds one whole slot: a scene uploaded without TLAS nodes (legal when !UseTlas) leaves `tlas` a 16-byte allocation while its slot
// is sized for the 2n - 1 nodes idkptBuildTlas / idkptBuildTlasOnDevice write in place later (ver_writable hands out slot 0 without looking at the allocation)
static int ver_reserve(dev_ctx* ctx)
{
    for (int b = 0; b < VB_COUNT; b++) {
        DevBuf& buf = vb_buf(ctx, b);
        if (ctx->vbytes[b] == 0 || buf.bytes >= ctx->vbytes[b]) continue;
        DevBuf nb; HIPC(nb.ensure(ctx->vstride[b]));
        if (buf.p && buf.bytes) HIPC(hipMemcpyAsync(nb.p, buf.p, std::min(buf.bytes, ctx->vbytes[b]), hipMemcpyDeviceToDevice, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));            // (the old allocation is released right below; once per upload, only for a buffer that was short)
        buf.release(); buf = nb;
    }
    return IDKPT_OK;
}
// the arena of buffer b gets room for every slot (first use of a second slot): only slot 0 is in use at that moment
static int ver_grow(dev_ctx* ctx, int b)
{
    if (ctx->valloc[b] >= ctx->verSlots) return IDKPT_OK;
    DevBuf& buf = vb_buf(ctx, b);
    const size_t need = ctx->vstride[b] * (size_t)ctx->verSlots;
    if (buf.bytes < need) {
        DevBuf nb; HIPC(nb.ensure(need));
        if (buf.p && ctx->vbytes[b]) HIPC(hipMemcpyAsync(nb.p, vb_ptr(ctx, b, ctx->vcur[b]), ctx->vbytes[b], hipMemcpyDeviceToDevice, ctx->stream));
        HIPC(hipStreamSynchronize(ctx->stream));            // (the old allocation is released right below; one-time cost per scene)
        buf.release(); buf = nb; ctx->vcur[b] = 0;
    }
    ctx->valloc[b] = ctx->verSlots;
    return IDKPT_OK;
}
// Where an update may write buffer b: *dst = the slot to write, *src = the slot that holds the current state (== *dst when the update can go in place).
// `full`: the update rewrites every byte of the buffer's state (nothing to carry over).  May launch queued samples / complete a deferred bounce when no slot is free.
static int ver_writable(dev_ctx* ctx, int b, bool full, char** src, char** dst)
{
    if (b == VB_NODES) ctx->pairValid = false;
    if (b == VB_NODES || b == VB_TRIVERTS) { ctx->wideFillValid = false; ctx->imarksValid = false; }   // (node boxes or triangle positions are about to change: boxes and leaf records of the wide nodes, and the triangle marks, are re-derived before their next use)
    if (b == VB_NODES || b == VB_XFORMS) { ctx->itlasValid = false; ctx->instRecValid = false; }    // (root boxes or transforms are about to change: the library's own TLAS is rebuilt before its next use)
    const int p = ctx->vcur[b];
    auto free_slot = [&](uint64_t busy) { if (ctx->verSlots > 1 && ctx->vbytes[b] > 0) for (int k = 0; k < ctx->verSlots; k++) if (!((busy >> k) & 1ull)) return k; return -1; };
    uint64_t pend = 0; for (const PendingSample& ps : ctx->pending) pend |= 1ull << ps.vs[b];
    const uint64_t held = ctx->defer.valid ? ctx->lastMask[b] : 0ull;                 // slots the deferred last bounce of the launched batch still reads
    if (!(((pend | held) >> p) & 1ull)) { *src = *dst = vb_ptr(ctx, b, p); return IDKPT_OK; }
    int q = free_slot(pend | held);
    if (q < 0 && held && (!((pend >> p) & 1ull) || free_slot(pend) >= 0)) {
        // no room, and completing the deferred bounce makes some (the current slot itself, or another one): cheaper than launching a short batch
        int rc = finish_deferred(ctx); if (rc) return rc;
        if (!((pend >> p) & 1ull)) { *src = *dst = vb_ptr(ctx, b, p); return IDKPT_OK; }
        q = free_slot(pend);
    }
    if (q < 0) {
        // every slot is pinned by queued samples: they are launched now (and the continuation they defer completed); the write is ordered behind them on the stream
        int rc = flush_any(ctx); if (rc) return rc;
        rc = finish_deferred(ctx); if (rc) return rc;
        *src = *dst = vb_ptr(ctx, b, p);
        return IDKPT_OK;
    }
    { int rc = ver_grow(ctx, b); if (rc) return rc; }
    *src = vb_ptr(ctx, b, ctx->vcur[b]); *dst = vb_ptr(ctx, b, q);
    if (!full) { HIPC(hipMemcpyAsync(*dst, *src, ctx->vbytes[b], hipMemcpyDeviceToDevice, ctx->stream)); *src = *dst; }
    ctx->vcur[b] = q;
    return IDKPT_OK;
}
// An update of something the render kernels read that is NOT versioned (materials, meshes, lights, settings ...): queued samples are launched first.
// Small host -> device update without a stream synchronisation: the bytes are staged in a pinned ring (4 x 256 KB) the copy engine reads later.
#define STAGE_BYTES (256u * 1024u)
static int staged_upload(dev_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return IDKPT_OK;
    if (bytes > STAGE_BYTES) { HIPC(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream)); return IDKPT_OK; }   // (host arrays are only borrowed for the duration of the call)
    if (!ctx->hStage) { HIPC(hipHostMalloc((void**)&ctx->hStage, (size_t)4 * STAGE_BYTES, hipHostMallocDefault)); for (int i = 0; i < 4; i++) HIPC(hipEventCreateWithFlags(&ctx->evStage[i], hipEventDisableTiming)); ctx->stageNext = 0; for (int i = 0; i < 4; i++) HIPC(hipEventRecord(ctx->evStage[i], ctx->stream)); }
    const int k = ctx->stageNext; ctx->stageNext = (k + 1) & 3;
    HIPC(hipEventSynchronize(ctx->evStage[k]));                // the copy that last read this quarter has finished
    memcpy(ctx->hStage + (size_t)k * STAGE_BYTES, src, bytes);
    HIPC(hipMemcpyAsync(dst, ctx->hStage + (size_t)k * STAGE_BYTES, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipEventRecord(ctx->evStage[k], ctx->stream));
    return IDKPT_OK;
}
