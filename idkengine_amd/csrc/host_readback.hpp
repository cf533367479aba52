// host_readback.hpp — what a host reads back: images, frames of the ring, ray state, alive queue, primary hits, statistics; stream interop.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

static int32_t dev_DownloadFrame(dev_ctx* ctx, int32_t slot, int32_t image, float* rgba, size_t bytes)
{
    if (!ctx || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptDownloadFrame: bad image id");
    REQUIRE(slot >= 0 && slot < ctx->ringSize, "idkptDownloadFrame: slot outside the frame ring");
    size_t need = (size_t)ctx->W * ctx->rows * 16;
    REQUIRE(bytes == need && need > 0, "idkptDownloadFrame: bytes must equal localRows*width*16");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    HIPC(hipMemcpyAsync(rgba, image_ptr(ctx, image, slot), need, hipMemcpyDeviceToHost, ctx->stream));
    SYNC_CHECKED();
    return IDKPT_OK;
}

static int32_t dev_GetFrameDevicePtr(dev_ctx* ctx, int32_t slot, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!ctx || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptGetFrameDevicePtr: bad image id");
    REQUIRE(slot >= 0 && slot < ctx->ringSize, "idkptGetFrameDevicePtr: slot outside the frame ring");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();                                        // launches what is still deferred (stream-ordered: a consumer on the context's stream sees the finished image)
    { int rc = check_overflow(ctx); if (rc) return rc; }   // (no wait: reports an overflow of batches that have already finished; a zero-copy consumer sees the rest at its next idkptSynchronize)
    *outPtr = image_ptr(ctx, image, slot);
    if (outBytes) *outBytes = (size_t)ctx->W * ctx->rows * 16;
    return IDKPT_OK;
}

static int32_t dev_Download(dev_ctx* ctx, int32_t image, float* rgba, size_t bytes)
{
    if (!ctx || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptDownload: bad image id");
    size_t need = (size_t)ctx->W * ctx->rows * 16;
    REQUIRE(bytes == need && need > 0, "idkptDownload: bytes must equal localRows*width*16");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    HIPC(hipMemcpyAsync(rgba, image_ptr(ctx, image, ctx->curSlot), need, hipMemcpyDeviceToHost, ctx->stream));
    SYNC_CHECKED();
    return IDKPT_OK;
}

static int32_t dev_DownloadRays(dev_ctx* ctx, GpuWavefrontRay* out, size_t bytes)
{
    if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t N = (size_t)ctx->W * ctx->rows;
    REQUIRE(bytes == N * sizeof(GpuWavefrontRay) && N > 0, "idkptDownloadRays: bytes must equal pixelCount*48");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    const size_t off = (size_t)(ctx->lastBatch - 1) * ctx->Npad * 16; // the most recent sample of the last batch
    { int rc = materialize_culled_rays(ctx); if (rc) return rc; }   // complete what the ray generation left out for pre-culled pixels
    std::vector<float4> a(N), b(N), c(N);
    HIPC(hipMemcpyAsync(a.data(), (char*)ctx->rayO.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(b.data(), (char*)ctx->rayT.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(c.data(), (char*)ctx->rayR.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    SYNC_CHECKED();
    for (size_t i = 0; i < N; i++) {
        GpuWavefrontRay& r = out[i];
        r.Origin[0] = a[i].x; r.Origin[1] = a[i].y; r.Origin[2] = a[i].z; r.PreviousIOROrTraverseCost = a[i].w;
        r.Throughput[0] = b[i].x; r.Throughput[1] = b[i].y; r.Throughput[2] = b[i].z; r.PackedDirectionX = b[i].w;
        r.Radiance[0] = c[i].x; r.Radiance[1] = c[i].y; r.Radiance[2] = c[i].z; r.PackedDirectionY = c[i].w;
    }
    return IDKPT_OK;
}

static int32_t dev_DownloadAliveQueue(dev_ctx* ctx, uint32_t* indices, size_t capacity, uint32_t* outCount)
{
    if (!ctx || !outCount) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    // the most recent sample's segment of the batch-wide queue; entries are ray ids -> subtract the sample's id offset
    const uint32_t* hb = ctx->hBases + (size_t)ctx->lastQueueCountSlot * (MAX_BATCH + 1);
    const uint32_t first = hb[ctx->lastBatch - 1], n = hb[ctx->lastBatch] - first;
    *outCount = n;
    if (indices && n) {
        REQUIRE(capacity >= n, "idkptDownloadAliveQueue: capacity too small");
        HIPC(hipMemcpyAsync(indices, ctx->queue[ctx->lastQueueSide].as<uint32_t>() + first, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
        const uint32_t sub = (uint32_t)(ctx->lastBatch - 1) * ctx->Npad;
        for (uint32_t i = 0; i < n; i++) indices[i] -= sub;
    }
    return IDKPT_OK;
}

static int32_t dev_EnablePrimaryHitCapture(dev_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->capturePrimary = enable != 0; return IDKPT_OK; }

static int32_t dev_DownloadPrimaryHits(dev_ctx* ctx, float* t, uint32_t* triangleId, float* baryXY, size_t pixelCount)
{
    if (!ctx || !t || !triangleId || !baryXY) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t N = (size_t)ctx->W * ctx->rows;
    REQUIRE(pixelCount == N, "idkptDownloadPrimaryHits: pixelCount mismatch");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    if (!ctx->capturePrimary || ctx->primHit.bytes < N * 16) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptDownloadPrimaryHits: call idkptEnablePrimaryHitCapture(ctx,1) before idkptRender");
    std::vector<float4> h(N);
    HIPC(hipMemcpyAsync(h.data(), ctx->primHit.p, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < N; i++) { t[i] = h[i].x; baryXY[2 * i] = h[i].y; baryXY[2 * i + 1] = h[i].z; memcpy(&triangleId[i], &h[i].w, 4); }
    return IDKPT_OK;
}

static int32_t dev_GetStats(dev_ctx* ctx, idkpt_stats* out)
{
    if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    SYNC_CHECKED();
    idkpt_stats s = ctx->stats;
    for (int j = 0; j < 16; j++) { const uint32_t* hb = ctx->hBases + (size_t)j * (MAX_BATCH + 1); s.LastAliveCounts[j] = (j >= 1 && j < ctx->st.RayDepth) ? hb[ctx->lastBatch] - hb[ctx->lastBatch - 1] : 0; }
    // [0]: primary rays that entered the traversal kernel (all pixels, or the survivors of the root-box pre-cull on the fast path)
    s.LastAliveCounts[0] = s.Frames ? (ctx->lastFast ? ctx->hCounts[MAX_DEPTH_SLOTS - 1] : (uint32_t)((size_t)ctx->W * ctx->rows)) : 0;
    s.LastFrameMs = 0.0f; s.LastTraceMs = 0.0f;
    if (ctx->timing && s.Frames > 0) { float ms = 0.0f; if (hipEventElapsedTime(&ms, ctx->evFrame[0], ctx->evFrame[1]) == hipSuccess) s.LastFrameMs = ms; }
    resolve_trace_events(ctx);
    s.TraceMsTotal = ctx->traceMsAcc; s.TraceLaunches = ctx->traceLaunchesAcc;
    s.LastTraceMs = s.TraceLaunches ? (float)(s.TraceMsTotal / (double)s.TraceLaunches) : 0.0f;
    uint64_t c[4] = {0, 0, 0, 0};
    HIPC(hipMemcpyAsync(c, ctx->counters64.p, 32, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
    s.NodePairVisits = c[0]; s.TriangleTests = c[1];
    if (ctx->opt.traceVariant == 107 || ctx->opt.traceVariant == 113 || ctx->opt.traceVariant == 116 || ctx->opt.traceVariant == 213) { uint64_t d[16]; HIPC(hipMemcpyAsync(d, ctx->counters64.p, 128, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream)); fprintf(stderr, "[idkpt prof] cycles refill %llu node %llu leaf %llu other %llu | refills %llu lanes %llu | nodeSteps %llu lanes %llu | leafPhases %llu lanes %llu | leafTests %llu leafTrips %llu\n", (unsigned long long)d[4], (unsigned long long)d[5], (unsigned long long)d[6], (unsigned long long)d[7], (unsigned long long)d[8], (unsigned long long)d[9], (unsigned long long)d[10], (unsigned long long)d[11], (unsigned long long)d[12], (unsigned long long)d[13], (unsigned long long)d[14], (unsigned long long)d[15]); }
    s.RaysTraced = s.PrimaryRays + c[2]; // N per sample + every alive-queue entry that entered a bounce
    if (ctx->wtotals.p) { uint64_t w[16]; HIPC(hipMemcpyAsync(w, ctx->wtotals.p, TOTALS_BYTES, hipMemcpyDeviceToHost, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream)); s.WideFlaggedRays = w[0]; s.WideNodeVisits = w[1]; s.WideLeafRecords = w[2]; s.WideTriangleTests = w[3]; s.InstTlasFlaggedRays = w[4];
        s.PacketFlaggedRays = w[8]; s.PacketPackets = w[9]; s.PacketNodeSteps = w[10]; s.PacketLiveLanes = w[11]; s.PacketRaysEntered = w[12]; s.PacketTriangleRounds = w[13]; }
    s.InstUnifiedLaunches = ctx->uniLaunches; s.InstUnifiedEntries = ctx->uniValid ? (uint32_t)ctx->uniEntries : 0u; s.InstUnifiedTopDepth = ctx->uniValid ? (uint32_t)ctx->uniDepth : 0u;
    *out = s;
    return IDKPT_OK;
}

static int32_t dev_ResetStats(dev_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    HIPC(hipStreamSynchronize(ctx->stream));
    memset(&ctx->stats, 0, sizeof(ctx->stats)); ctx->uniLaunches = 0;
    if (ctx->hPkStats) memset(ctx->hPkStats, 0, 64);
    for (int i = 0; i < 6; i++) ctx->pkSeen[i] = 0;
    ctx->evUsed = 0; ctx->traceMsAcc = 0.0; ctx->traceLaunchesAcc = 0;
    memset(ctx->hCounts, 0, (MAX_DEPTH_SLOTS - 1) * 4);     // (the last word, the length of the primary active list, is also the grid hint of the next batch: idkptGetStats reports it only once frames were rendered)
    HIPC(hipMemsetAsync(ctx->counters64.p, 0, 128, ctx->stream)); if (ctx->wtotals.p) HIPC(hipMemsetAsync(ctx->wtotals.p, 0, TOTALS_BYTES, ctx->stream)); HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

static int32_t dev_EnableCounters(dev_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->counters = enable != 0; return IDKPT_OK; }
static int32_t dev_EnableTiming(dev_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->timing = enable != 0; return IDKPT_OK; }

static int32_t dev_GetImageDevicePtr(dev_ctx* ctx, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!ctx || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3 && ctx->W > 0, "idkptGetImageDevicePtr: bad image / no size");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();                                        // launches what is still deferred
    { int rc = check_overflow(ctx); if (rc) return rc; }   // (no wait: see idkptGetFrameDevicePtr)
    *outPtr = image_ptr(ctx, image, ctx->curSlot);
    if (outBytes) *outBytes = (size_t)ctx->W * ctx->rows * 16;
    return IDKPT_OK;
}

static int32_t dev_SetStream(dev_ctx* ctx, void* hipStream)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    if (hipStream) { if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream); ctx->stream = (hipStream_t)hipStream; ctx->ownStream = false; }
    else if (!ctx->ownStream) { HIPC(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->ownStream = true; }
    return IDKPT_OK;
}
static int32_t dev_GetStream(dev_ctx* ctx, void** out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = (void*)ctx->stream; return IDKPT_OK; }
