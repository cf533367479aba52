// host_queries.hpp — idkptTraceRays / idkptTraceShadows (kernels_query.hpp) on the traversal kernels' scheduler.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

static DScene make_dscene(dev_ctx* ctx);
// frame constants for the ray-query / shadow kernels: only the traversal-related fields are read
static int query_frame(dev_ctx* ctx, Frame& f, size_t& ldsBytes, uint32_t& grid)
{
    memset(&f, 0, sizeof(f));
    f.g = ctx->st.Gpu; f.useTlas = ctx->st.UseTlas;
    f.stackCap = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack);
    f.tlasCap = std::min(TLAS_STACK_SIZE, std::max(1, ctx->tlasNeed));
    ldsBytes = (size_t)(f.stackCap + 2 + (f.useTlas ? f.tlasCap : 0)) * WAVE * 4;   // + the dummy and the spare row of k_trace2's stack (kernels_trace.hpp)
    if (ldsBytes > 64 * 1024) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack");
    int wavesPerCU = (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
    grid = (uint32_t)(ctx->numCUs * std::max(1, wavesPerCU));
    return IDKPT_OK;
}

// issue only (H2D, kernel, D2H on the context's stream); the caller synchronises.  hits must stay valid until then.
static int32_t dev_TraceRaysIssue(dev_ctx* ctx, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits, bool devicePtrs = false /* rays / hits live on this context's device: no copies, nothing to wait for */)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRays: no scene uploaded");
    REQUIRE(count == 0 || (rays && hits), "idkptTraceRays: null rays/hits");
    REQUIRE(count < (1ull << 31), "idkptTraceRays: too many rays in one call");
    REQUIRE((flags & ~3u) == 0, "idkptTraceRays: unknown flags");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRays: UseTlas set but no TLAS nodes uploaded");
    if (count == 0) return IDKPT_OK;
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    Frame f; size_t ldsBytes; uint32_t grid;
    int rc = query_frame(ctx, f, ldsBytes, grid); if (rc) return rc;
    DScene s = make_dscene(ctx);
    hipStream_t st = ctx->stream;
    const idkpt_ray* dIn = rays; idkpt_hit* dOut = hits;
    if (!devicePtrs) {
        HIPC(ctx->queryIn.ensure(count * sizeof(idkpt_ray))); HIPC(ctx->queryOut.ensure(count * sizeof(idkpt_hit)));
        HIPC(hipMemcpyAsync(ctx->queryIn.p, rays, count * sizeof(idkpt_ray), hipMemcpyHostToDevice, st));
        dIn = ctx->queryIn.as<idkpt_ray>(); dOut = ctx->queryOut.as<idkpt_hit>();
    }
    HIPC(ctx->qwork.ensure((WORK_WORDS + 128) * 4));                       // its own work-list counters: the frame's are reset by the frame's last kernel, not per batch
    uint32_t* work = ctx->qwork.as<uint32_t>();
    const int lights = (flags & IDKPT_TRACE_LIGHTS) ? 1 : 0;
    if (ctx->opt.queryScheduler && !f.g.DoDebugBVHTraversal) {
        const bool anyHit = (flags & IDKPT_TRACE_ANY_HIT) != 0;
        if (!anyHit && ctx->instanceCount > 1 && !f.useTlas) {   // closest hits of a several-instance scene without UseTlas: the exact loop with its instance sieve where a frame would use it or the own TLAS (kernels_trace_inst.hpp)
            bool useT = false, useS = false;
            rc = inst_tlas_prepare(ctx, &useT, &useS); if (rc) return rc;
            if ((useT || useS) && ctx->instRecValid) {
                f.instSieve = 1; s.instRec = (const float4*)ctx->instRec.as<float4>();
                ldsBytes = (size_t)(f.stackCap + 2 + inst_tlas_rows(ctx)) * WAVE * 4;
                if (ldsBytes > 64 * 1024) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack");
                grid = (uint32_t)(ctx->numCUs * std::max(1, (int)std::min<size_t>(32, (160 * 1024) / ldsBytes)));
            }
        }
        // closest hit / any hit: k_trace2's persistent-wave scheduler (kernels_query.hpp): prepare (lights, root test, trace-ready records) -> k_trace2 -> Hit flags
        HIPC(ctx->queryRec.ensure(count * 64)); HIPC(ctx->queryList.ensure(count * 4));
        HIPC(hipMemsetAsync(work, 0, (WORK_WORDS + 128) * 4, st));
        uint32_t* listCount = work + WORK_WORDS;
        f.queryMode = 1; f.g.DoTraceLights = 0;                            // (the lights are folded into the records)
        f.grabUnitLog2 = std::min(24, std::max(6, ctx->opt.grabUnitLog2)); f.grabFixed = std::max(0, ctx->opt.grabFixed); f.leafMin = ctx->opt.leafMin > 0 ? ctx->opt.leafMin : 16;
        f.poolMin = ctx->opt.poolMin; f.advMin = ctx->opt.advMin > 0 ? ctx->opt.advMin : 8; f.batch = 1; f.Npad = (uint32_t)count;
        TraceBufs tr = {ctx->queryRec.as<float4>(), nullptr, nullptr};
        const uint32_t blocks = (uint32_t)((count + 255) / 256);
        hipLaunchKernelGGL(k_query_prepare, dim3(blocks), dim3(256), 0, st, s, f, dIn, dOut, (uint32_t)count, lights, anyHit ? 1 : 0, tr, ctx->queryList.as<uint32_t>(), listCount);
        RayBufs noRays = {nullptr, nullptr, nullptr, nullptr, nullptr};
        HitBufs qhits = {(float4*)dOut, ctx->hitCost.as<float>()};
        const uint32_t g2 = std::min<uint32_t>(grid, std::max<uint32_t>(1u, (uint32_t)((count + 63) / 64)));
        launch_trace2<true>(ctx, g2, ldsBytes, st, s, f, noRays, tr, qhits, (const uint32_t*)ctx->queryList.as<uint32_t>(), (const uint32_t*)listCount, work, (uint64_t*)(work + WORK_WORDS + 64) /* visit counters of queries do not count as the frame's */, false, 0, anyHit);
        hipLaunchKernelGGL(k_query_finish, dim3(blocks), dim3(256), 0, st, dIn, dOut, (const uint32_t*)ctx->queryList.as<uint32_t>(), (const uint32_t*)listCount);
        HIPC(hipGetLastError());
        if (!devicePtrs) HIPC(hipMemcpyAsync(hits, ctx->queryOut.p, count * sizeof(idkpt_hit), hipMemcpyDeviceToHost, st));
        return IDKPT_OK;
    }
    HIPC(hipMemsetAsync(work, 0, 4, st));
    if (flags & IDKPT_TRACE_ANY_HIT) hipLaunchKernelGGL((k_trace_query<true>), dim3(grid), dim3(WAVE), ldsBytes, st, s, f, dIn, dOut, (uint32_t)count, lights, work);
    else hipLaunchKernelGGL((k_trace_query<false>), dim3(grid), dim3(WAVE), ldsBytes, st, s, f, dIn, dOut, (uint32_t)count, lights, work);
    HIPC(hipGetLastError());
    if (!devicePtrs) HIPC(hipMemcpyAsync(hits, ctx->queryOut.p, count * sizeof(idkpt_hit), hipMemcpyDeviceToHost, st));
    return IDKPT_OK;
}
static int32_t dev_TraceRaysDevice(dev_ctx* ctx, const idkpt_ray* dRays, size_t count, uint32_t flags, idkpt_hit* dHits) { return dev_TraceRaysIssue(ctx, dRays, count, flags, dHits, true); }
static int32_t dev_TraceRays(dev_ctx* ctx, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits)
{
    int rc = dev_TraceRaysIssue(ctx, rays, count, flags, hits); if (rc) return rc;
    if (count == 0) return IDKPT_OK;
    SYNC_CHECKED();
    return IDKPT_OK;
}

static int32_t dev_TraceShadows(dev_ctx* ctx, const idkpt_shadow_params* p, const float* depth, const float* normalOct, float* visibility, bool devicePtrs = false)
{
    if (!ctx || !p) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadows: no scene uploaded");
    REQUIRE(depth && normalOct && visibility, "idkptTraceShadows: null image");
    REQUIRE(p->Width > 0 && p->Height > 0 && (size_t)p->Width * p->Height < (1ull << 30), "idkptTraceShadows: bad image size");
    REQUIRE(p->RayTracingSamples >= 1, "idkptTraceShadows: RayTracingSamples must be >= 1");
    REQUIRE(p->LightIndex >= 0 && p->LightIndex < ctx->lightCount, "idkptTraceShadows: LightIndex out of range");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadows: UseTlas set but no TLAS nodes uploaded");
    HIPC(hipSetDevice(ctx->device));
    FLUSH_KEEP();
    Frame f; size_t ldsBytes; uint32_t grid;
    int rc = query_frame(ctx, f, ldsBytes, grid); if (rc) return rc;
    DScene s = make_dscene(ctx);
    const size_t N = (size_t)p->Width * p->Height;
    hipStream_t st = ctx->stream;
    const float* dDepth = depth; const float2* dNormal = (const float2*)normalOct; float* dVis = visibility;
    if (!devicePtrs) {
        HIPC(ctx->queryIn.ensure(N * 12)); HIPC(ctx->queryOut.ensure(N * 4));
        float* in = ctx->queryIn.as<float>();
        HIPC(hipMemcpyAsync(in, depth, N * 4, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(in + N, normalOct, N * 8, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(ctx->queryOut.p, visibility, N * 4, hipMemcpyHostToDevice, st));
        dDepth = in; dNormal = (const float2*)(in + N); dVis = ctx->queryOut.as<float>();
    }
    const uint32_t tiles = (uint32_t)(((p->Width + 7) / 8) * ((p->Height + 7) / 8));
    hipLaunchKernelGGL(k_shadows, dim3(tiles), dim3(WAVE), ldsBytes, st, s, f, *p, (const float*)dDepth, (const float2*)dNormal, dVis);
    HIPC(hipGetLastError());
    if (devicePtrs) return IDKPT_OK;                                       // (asynchronous, in stream order: idkptSynchronize or the host's own stream wait completes it)
    HIPC(hipMemcpyAsync(visibility, dVis, N * 4, hipMemcpyDeviceToHost, st));
    SYNC_CHECKED();
    return IDKPT_OK;
}
static int32_t dev_TraceShadowsDevice(dev_ctx* ctx, const idkpt_shadow_params* p, const float* dDepth, const float* dNormalOct, float* dVisibility) { return dev_TraceShadows(ctx, p, dDepth, dNormalOct, dVisibility, true); }
