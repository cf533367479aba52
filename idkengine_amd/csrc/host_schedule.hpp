// host_schedule.hpp — PathTracer.Compute: the launch schedule of a batch of samples (flush_batch), its deferred last bounce, idkptRender and the batching knobs.
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

// slots: which state of every versioned buffer the kernels read (null: the current one); multi: the batch's samples saw different states -> the pointers are the
// arena bases and DScene::ver holds every sample's offsets (VER kernels)
static DScene make_dscene(dev_ctx* ctx, const uint8_t* slots, bool multi)
{
    DScene s;
    auto at = [&](int b) -> char* { return multi ? (char*)vb_buf(ctx, b).p : vb_ptr(ctx, b, slots ? slots[b] : ctx->vcur[b]); };
    s.nodes = (const float4*)at(VB_NODES); s.tris = ctx->tris.as<uint4>(); s.triVerts = (const float4*)at(VB_TRIVERTS);
    s.descs = ctx->descs.as<GpuBlasDesc>(); s.instances = ctx->instances.as<GpuBlasInstance>(); s.instanceCount = ctx->instanceCount;
    s.tlas = (const float4*)at(VB_TLAS); s.tlasCount = ctx->tlasCount; s.vertices = (const uint4*)at(VB_VERTICES);
    s.instRec = nullptr;        // (set for the batches that walk the library's own TLAS, flush_batch)
    s.pairNodes = nullptr;      // (set for the batches whose one-BLAS launches take k_trace2's FAST node step, flush_batch)
    s.meshes = ctx->meshes.as<GpuMesh>(); s.materials = ctx->materials.as<GpuMaterial>(); s.xforms = (const float4*)at(VB_XFORMS);
    s.lights = ctx->lights.as<GpuLight>(); s.lightCount = ctx->lightCount; s.sky = ctx->sky.as<float4>(); s.skySize = ctx->skySize;
    s.textures = ctx->texDescs.as<TexDesc>(); s.textureCount = ctx->textureCount; s.srgbLut = ctx->srgbLut.as<float>();
    s.overflow = ctx->dOverflow;
    s.ver = multi ? ctx->verTab.as<uint32_t>() : nullptr;
    return s;
}
static DScene make_dscene(dev_ctx* ctx) { return make_dscene(ctx, nullptr, false); }
static DScene make_dscene_last(dev_ctx* ctx) { return make_dscene(ctx, ctx->lastSlots, ctx->lastMulti); }   // what the last launched batch read (finish_deferred, regeneration of culled rays)

static float4* image_ptr(dev_ctx* ctx, int i, int slot) { return ctx->img[i].as<float4>() + (size_t)slot * ((size_t)ctx->W * ctx->rows); }

// fast path = persistent while-while traversal (one BLAS, instance list or TLAS); only the debug traversal-cost view uses the general kernel
static bool fast_path(dev_ctx* ctx) { return ctx->instanceCount >= 1 && !ctx->st.Gpu.DoDebugBVHTraversal && !ctx->opt.forceGeneric; }

// One batch of B deferred samples: FirstHit -> [sort ->] NHit x (RayDepth-1) -> FinalDraw (PathTracer.cs:218-270), every
// stage launched once for all B samples.  Sample k owns ray ids [k*Npad, k*Npad+N); alive queues are batch-wide but stay
// grouped by sample (stable compaction / sort with the sample index above the key), and every ray's NHit slot is its
// position inside its own sample's queue, so each sample gets exactly the RNG streams of a stand-alone frame.
// Grid of a traversal launch whose ray count is only known on the device: `prev` = the count the same launch had in the previous batch (host-mapped mirror,
// possibly one batch stale; 0 = unknown -> full grid).  Any grid >= 1 is correct (the waves are persistent); the size only costs or saves time:
//   * never more than hintMul x the waves that hold all rays at once (tiny frames would otherwise spend their time dispatching idle workgroups), at least 256;
//   * launches below ~1.5 rays per lane of the full grid run faster on FEWER, fuller waves — every wave instruction costs the same whatever its exec mask, and a
//     launch this small lasts as long as its longest rays, whose steps get faster when fewer waves share a SIMD: raysX4 / 4 rays per lane, but not below
//     1024 waves where the first rule allows them (round 3, same box: headline one frame at a time +6 %, Cornell 1080p RayDepth 5 +12 %; profiles/r03_trace_experiments.md 7).
//   * launches of up to GRID_MID_RAYS rays (the headline frame with up to ~24 samples in flight, one rank's share of an N-GPU frame) run 2-4 % faster on 20 than on 24
//     waves per CU for the same reason, and views whose launches are that small only with a few samples in flight (every pixel traversing) lose nothing measurable;
//     above it 24 is never worse (profiles/r03_trace_experiments.md 9).
#define GRID_MID_RAYS 14000000u
// Which traversal kernel a launch gets: k_trace2s (kernels_trace_split.hpp: long rays split across the idle lanes of their wave once the work list is empty) pays
// where a launch ends with a few long rays on an otherwise idle chip — launches of up to SPLIT_MAX_RAYS rays (a frame traced alone, the bounce launches of small
// batches, one rank's share of an N-GPU frame); its extra registers (one wave per SIMD less) cost large launches more than their tails are worth.
// Measured (profiles/r04_small_launch_experiments.md): headline view one frame at a time (0.36 M + 0.28 M rays per launch) +8 %, three samples in flight +4 %, one
// rank's share of an 8 / 4-GPU frame +12 % / +5 %; the atrium and the interior view one frame at a time (1.9-2.1 M rays per launch, every pixel traverses) -10 % / -1.5 %.
// So: launches of fewer than SPLIT_MAX_RAYS rays, and only on views where most pixels miss the scene's root box (fewer than half of the primary rays entered the
// traversal in the previous batch) — there the launch time is the dependent chain of the rays that cross the whole scene without hitting anything.
#define SPLIT_MAX_RAYS 1500000u
static bool want_split(const dev_ctx* ctx, uint32_t prev, bool known, int samples)
{
    if (ctx->opt.split == 0) return false;
    if (ctx->opt.split >= 2) return true;
    const uint64_t pixels = (uint64_t)ctx->W * ctx->rows * (uint64_t)std::max(1, samples);
    const bool sparse = ctx->lastFast && ctx->lastBatch == samples && (uint64_t)ctx->hCounts[MAX_DEPTH_SLOTS - 1] * 2u < pixels;
    return known && sparse && prev > 0u && prev < SPLIT_MAX_RAYS;
}
// k_trace_fused: where a batch's two traversal launches are bound by their longest rays, not by their ray count (the same regime as the split)
// (measured: it saves launches, not chain length — +5 % where one sparse frame is traced alone, a loss everywhere else: kernels_trace_fused.hpp)
#define FUSED_MAX_RAYS 600000u
static bool want_fused(const dev_ctx* ctx, uint32_t prev, bool known, int samples)
{
    if (ctx->opt.fused == 0) return false;
    if (ctx->opt.fused >= 2) return true;
    const uint64_t pixels = (uint64_t)ctx->W * ctx->rows * (uint64_t)std::max(1, samples);
    return known && prev > 0u && prev < FUSED_MAX_RAYS && (uint64_t)prev * 2u < pixels;
}
static uint32_t small_launch_grid(uint32_t fullGrid, uint32_t prev, int hintMul, int raysX4, uint32_t midGrid)
{
    if (prev == 0u || hintMul <= 0) return fullGrid;
    const uint32_t cap = std::max<uint32_t>(256u, (uint32_t)(((uint64_t)hintMul * prev + 63) / 64));
    uint32_t g = cap;
    if (raysX4 > 0) g = std::max<uint32_t>((uint32_t)(((uint64_t)prev * 4u / (uint32_t)raysX4 + 63) / 64), std::min<uint32_t>(cap, 1024u));
    if (midGrid > 0u && prev < GRID_MID_RAYS) g = std::min(g, midGrid);
    return std::min(fullGrid, std::min(g, cap));
}

// The continuation of a deferred last bounce (k_shade_last): the radiance k_shade_last replaced goes back, then the ordinary kernels of the bounce run — shading,
// scan, scatter — on the inputs the batch left untouched (hit records, ray state, the queue entering the bounce, its per-sample bases).  Afterwards ray state, alive
// queue and counts are what the eager path leaves, bit for bit; the frame was complete before.
static int finish_deferred(dev_ctx* ctx)
{
    if (!ctx->defer.valid) return IDKPT_OK;
    ctx->defer.valid = false;
    using namespace ptd;
    hipStream_t st = ctx->stream;
    const int j = ctx->defer.j, side = ctx->defer.side, B = ctx->defer.B, BS = MAX_BATCH + 1;
    const uint32_t total = ctx->defer.total, Npad = ctx->defer.Npad, gridTotal = (total + 255) / 256;
    const Frame f = ctx->lastFrame;
    DScene s = make_dscene_last(ctx);                                // the scene states the deferred batch was traced with (its slots are pinned until now: ver_writable)
    const bool multiVer = ctx->lastMulti;
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    HitBufs hits = {ctx->hit.as<float4>(), ctx->hitCost.as<float>()};
    TraceBufs tr = {ctx->trRec.as<float4>(), nullptr, nullptr};
    uint32_t* counts = ctx->counts.as<uint32_t>(); uint32_t* bases = ctx->bases.as<uint32_t>();
    unsigned long long* contMask = ctx->contMask.as<unsigned long long>();
    uint32_t* waveLocal = ctx->waveCounts.as<uint32_t>(); uint32_t* blockSums = ctx->blockSums.as<uint32_t>(); uint32_t* keysTmp = ctx->keysTmp.as<uint32_t>();
    const uint32_t scanBlocks = ((total + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK;
    const uint32_t* q = ctx->queue[side].as<uint32_t>();
    const uint32_t* cnt = ctx->deferCount.as<uint32_t>();            // (counts[j] itself was reset by the batch's last kernel)
    if (ctx->defer.allHits) hipLaunchKernelGGL((k_restore_last<true>), dim3(gridTotal), dim3(256), 0, st, rays, hits, q, cnt, (const float4*)ctx->radSave.as<float4>(), f.hitsByRid);
    else hipLaunchKernelGGL((k_restore_last<false>), dim3(gridTotal), dim3(256), 0, st, rays, hits, q, cnt, (const float4*)ctx->radSave.as<float4>(), f.hitsByRid);
    if (multiVer) hipLaunchKernelGGL((k_shade<false, true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, q, cnt, 0u, (const uint32_t*)(bases + j * BS), (const uint32_t*)nullptr, contMask, waveLocal, keysTmp);
    else hipLaunchKernelGGL((k_shade<false, false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, q, cnt, 0u, (const uint32_t*)(bases + j * BS), (const uint32_t*)nullptr, contMask, waveLocal, keysTmp);
    hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, cnt, 0u, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, cnt, 0u, blockSums, (const uint32_t*)waveLocal, counts + j + 1, (unsigned long long*)nullptr,
                       (const unsigned long long*)contMask, (const uint32_t*)(bases + j * BS), Npad, B, bases + (j + 1) * BS,
                       ctx->dCountsMirror + j + 1, ctx->dBasesMirror + (size_t)(j + 1) * BS, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL((k_compact<false>), dim3(gridTotal), dim3(256), 0, st, q, cnt, 0u, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                       (const uint32_t*)keysTmp, ctx->queue[1 - side].as<uint32_t>(), ctx->keys[1 - side].as<uint32_t>());
    HIPC(hipGetLastError());
    ctx->lastQueueSide = 1 - side;
    ctx->countersDirty = true;                                       // (counts[j + 1] was written after the batch's reset: the next batch clears its counters itself)
    return IDKPT_OK;
}

static int flush_batch(dev_ctx* ctx)
{
    const int B = (int)ctx->pending.size();
    if (B == 0) return IDKPT_OK;
    if (ctx->grouped && !ctx->inGroupFlush) return fail(ctx, IDKPT_ERR_UNKNOWN, "internal: a member of a multi-device context was flushed on its own");
    const uint32_t N = (uint32_t)((size_t)ctx->W * ctx->rows);
    const uint32_t Npad = ctx->Npad;
    const uint32_t total = (uint32_t)B * Npad;
    // which state of the geometry every sample sees (scene versions): one for all -> plain pointers; else a per-sample table of offsets into the arenas (VER kernels)
    bool multiVer = false;
    for (int k = 1; k < B && !multiVer; k++) multiVer = memcmp(ctx->pending[k].vs, ctx->pending[0].vs, VB_COUNT) != 0;
    for (int b = 0; b < VB_COUNT; b++) { uint64_t m = 0; for (int k = 0; k < B; k++) m |= 1ull << ctx->pending[k].vs[b]; ctx->lastMask[b] = m; ctx->lastSlots[b] = ctx->pending[0].vs[b]; }
    ctx->lastMulti = multiVer;
    if (multiVer) {
        if (!ctx->hVerTab) {
            HIPC(hipHostMalloc((void**)&ctx->hVerTab, (size_t)2 * MAX_BATCH * SCENE_VER_WORDS * 4, hipHostMallocDefault));
            for (int i = 0; i < 2; i++) HIPC(hipEventCreateWithFlags(&ctx->evVer[i], hipEventDisableTiming));
            ctx->verHalf = 0;
        } else HIPC(hipEventSynchronize(ctx->evVer[ctx->verHalf]));          // the copy that last read this half has finished
        uint32_t* stage = ctx->hVerTab + (size_t)ctx->verHalf * MAX_BATCH * SCENE_VER_WORDS;
        for (int k = 0; k < B; k++) {
            const uint8_t* vs = ctx->pending[k].vs;
            uint32_t* row = stage + (size_t)k * SCENE_VER_WORDS;
            for (int b = 0; b < VB_COUNT; b++) row[b] = (uint32_t)(((size_t)vs[b] * ctx->vstride[b]) / 16);   // 16-byte units
            for (int b = VB_COUNT; b < SCENE_VER_WORDS; b++) row[b] = 0u;
        }
        HIPC(ctx->verTab.ensure((size_t)MAX_BATCH * SCENE_VER_WORDS * 4));
        HIPC(hipMemcpyAsync(ctx->verTab.p, stage, (size_t)B * SCENE_VER_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipEventRecord(ctx->evVer[ctx->verHalf], ctx->stream));
        ctx->verHalf ^= 1;
    }
    DScene s = make_dscene_last(ctx);
    Frame f; memset(&f, 0, sizeof(f));                                 // (every field a kernel variant may look at has a defined value: queryMode, hitsByRid, ...)
    memcpy(f.invProj, ctx->pending[0].cam, 64); memcpy(f.invView, ctx->pending[0].cam + 16, 64); memcpy(f.viewPos, ctx->pending[0].cam + 32, 12);   // the camera the samples were queued with
    f.W = ctx->W; f.H = ctx->H; f.rowMod = ctx->rowMod; f.rowRem = ctx->rowRem; f.rows = ctx->rows; f.rowBandLog2 = ctx->rowBandLog2;
    f.g = ctx->st.Gpu; f.useTlas = ctx->st.UseTlas;
    f.stackCap = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack);
    f.outputAovs = ctx->st.OutputAOVs;
    f.batch = B; f.Npad = Npad;
    HIPC(ctx->trRec.ensure((size_t)ctx->maxBatch * Npad * 64));
    for (int k = 0; k < MAX_BATCH; k++) { f.accum[k] = k < B ? ctx->pending[k].accum : 0u; f.slotOf[k] = (uint32_t)(k < B ? ctx->pending[k].slot : 0); }
    f.seqFirst = ctx->seqFirst; f.seqStride = ctx->seqStride;
    f.accumulated = f.seqFirst + f.accum[0] * f.seqStride;
    f.cams = nullptr;
    if (ctx->ringSize > 1) {   // frame ring: every sample renders with the camera it was queued with
        // pinned double-buffered staging: no stream synchronisation per batch (the host may run ahead of the GPU)
        if (!ctx->hCams) {
            HIPC(hipHostMalloc((void**)&ctx->hCams, (size_t)2 * MAX_BATCH * 36 * 4, hipHostMallocDefault));
            for (int i = 0; i < 2; i++) HIPC(hipEventCreateWithFlags(&ctx->evCams[i], hipEventDisableTiming));
            ctx->camHalf = 0;
        } else HIPC(hipEventSynchronize(ctx->evCams[ctx->camHalf]));          // the copy that last read this half has finished
        float* stage = ctx->hCams + (size_t)ctx->camHalf * MAX_BATCH * 36;
        for (int k = 0; k < B; k++) memcpy(stage + (size_t)k * 36, ctx->pending[k].cam, 36 * 4);
        HIPC(ctx->camTab.ensure((size_t)MAX_BATCH * 36 * 4));
        HIPC(hipMemcpyAsync(ctx->camTab.p, stage, (size_t)B * 36 * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPC(hipEventRecord(ctx->evCams[ctx->camHalf], ctx->stream));
        ctx->camHalf ^= 1;
        f.cams = ctx->camTab.as<float>();
    }
    f.tilePerSample = (f.cams != nullptr || multiVer) ? 1 : 0;
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    HitBufs hits = {ctx->hit.as<float4>(), ctx->hitCost.as<float>()};
    uint32_t* counts = ctx->counts.as<uint32_t>();
    uint32_t* bases = ctx->bases.as<uint32_t>();             // [MAX_DEPTH_SLOTS][MAX_BATCH+1]
    uint32_t* work = ctx->work.as<uint32_t>();
    uint64_t* counters = ctx->counters64.as<uint64_t>();
    const int depth = ctx->st.RayDepth;
    hipStream_t st = ctx->stream;
#ifdef IDKPT_DEVELOPER
    // "graph_probe" (developer build): is a hipGraph of a batch's launches faster than the launches?  The batch is captured instead of executed, then
    // executed once as a graph and replayed graph_probe times between two events (the replays re-accumulate the same sample: timing only).
    bool capturing = false;
    if (ctx->opt.graphProbe > 0 && !ctx->timing) capturing = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
#endif
    if (ctx->timing) HIPC(hipEventRecord(ctx->evFrame[0], st));
    ctx->defer.valid = false;                                      // (a deferred last bounce of the previous batch that nobody asked for: its buffers are reused now)
    if (ctx->countersDirty) { HIPC(hipMemsetAsync(work, 0, WORK_WORDS * 4, st)); HIPC(hipMemsetAsync(counts, 0, MAX_DEPTH_SLOTS * 4, st)); }   // (otherwise the previous batch's k_final_draw has reset them)
    ctx->countersDirty = true;
    uint32_t* hostCounts = ctx->dCountsMirror; uint32_t* hostBases = ctx->dBasesMirror;

    f.tlasCap = std::min(TLAS_STACK_SIZE, std::max(1, ctx->tlasNeed));
    f.grabUnitLog2 = std::min(24, std::max(6, ctx->opt.grabUnitLog2)); f.grabFixed = std::max(0, ctx->opt.grabFixed);   // work-list hand-out (kernels_trace.hpp)
    f.leafMin = ctx->opt.leafMin > 0 ? ctx->opt.leafMin : (B >= 4 ? 16 : 12);        // (measured: 16-20 with many samples in flight, 12 for a frame traced alone; tools/sweep_sched.py)
    f.instTlas = 0;                                                           // the instance loop through the library's own TLAS (kernels_trace_inst.hpp): decided per batch, the rays' producers look at it too
    if (fast_path(ctx)) { bool useT = false, useS = false; int rc = inst_tlas_prepare(ctx, &useT, &useS); if (rc) { ctx->pending.clear(); return rc; } f.instTlas = useT ? 1 : 0; f.instSieve = useS ? 1 : 0; }
    if (fast_path(ctx) && !multiVer && pair_nodes_wanted(ctx)) { int rc = pair_nodes_prepare(ctx); if (rc) { ctx->pending.clear(); return rc; } if (ctx->pairValid) s.pairNodes = (const float4*)ctx->pairNodes.as<float4>(); }
    if (f.instSieve) s.instRec = (const float4*)ctx->instRec.as<float4>();
    if (f.instTlas) { s.tlas = (const float4*)ctx->itlas.as<float4>(); s.tlasCount = 2 * ctx->instanceCount - 1; s.instRec = (const float4*)ctx->instRec.as<float4>(); }   // (what the kernels of this batch see as "the TLAS": only the primary rays' pre-cull and k_trace_inst look at it)
    size_t ldsBytes = (size_t)(f.stackCap + 2 + (f.useTlas ? f.tlasCap : ((f.instTlas || f.instSieve) ? inst_tlas_rows(ctx) : 0))) * WAVE * 4;   // + the dummy and the spare row of k_trace2's stack (kernels_trace.hpp)
    ldsBytes += (size_t)std::max(0, ctx->opt.ldsPad);   // option "lds_pad": caps the waves per CU (occupancy experiments)
    if (ldsBytes > 64 * 1024) { ctx->pending.clear(); return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack"); }
    // persistent trace grid: as many 1-wave workgroups as the chip holds (32 waves/CU, limited by LDS)
    int wavesPerCU = (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
    wavesPerCU = std::max(1, wavesPerCU);
    if (ctx->opt.traceWaves > 0) wavesPerCU = ctx->opt.traceWaves;   // option "trace_waves": one-wave workgroups per CU in the persistent grid
    // (a launch never needs more waves than it can have rays: small frames would otherwise spend their time dispatching idle workgroups)
    const uint32_t traceGrid = std::min<uint32_t>((uint32_t)(ctx->numCUs * wavesPerCU), std::max<uint32_t>(1u, (uint32_t)(((size_t)B * N + 63) / 64)));
    const uint32_t midGrid = (ctx->opt.gridMidWaves > 0 && ctx->opt.traceWaves == 0) ? (uint32_t)(ctx->numCUs * std::min(wavesPerCU, ctx->opt.gridMidWaves)) : 0u;   // (an explicit trace_waves wins)
    f.gridRaysX4 = (uint32_t)std::max(0, ctx->opt.gridRaysX4); f.gridMid = midGrid; f.gridMidRays = GRID_MID_RAYS; f.splitMode = (ctx->opt.split == 3 ? 2 : 1) | (ctx->opt.splitDonor ? 4 : 0); f.poolMin = ctx->opt.poolMin; f.advMin = ctx->opt.advMin > 0 ? ctx->opt.advMin : (B >= 4 ? 8 : 1); f.splitPeek = ctx->opt.splitPeek;   // the same rules inside k_trace2, on the launch's actual ray count
    const bool debug = f.g.DoDebugBVHTraversal != 0;
    const uint32_t gridTotal = (total + 255) / 256;
    const bool fast = fast_path(ctx);
    if (fast && wide_wanted(ctx)) { int rc = wide_prepare(ctx); if (rc) { ctx->pending.clear(); return rc; } }
    // (not where the samples of a batch are different frames — their own cameras or scene versions: the same pixel is then not the same ray, and under scene versions not even the
    // same node addresses; measured on the animated bench, 8 / 32 frames in flight: 3 291 / 3 716 -> 3 011 / 3 117 Mray/s)
    f.genPixelMajor = (fast && ctx->opt.genPixelMajor > 0 && B >= ctx->opt.genPixelMajor && !f.tilePerSample) ? 1 : 0;
    // the primary launch as a packet launch (kernels_packet.hpp): one-BLAS scenes; by the kernel's own counters unless forced (host_launch.hpp packet_decide)
    const bool packet = fast && !multiVer && !f.useTlas && packet_decide(ctx, f);
    // one launch for FirstHit + the last NHit (kernels_trace_fused.hpp): RayDepth 2, one BLAS instance, the last bounce deferred (no AOVs, no debug view), nothing that looks at
    // the primary hits or the visit counters, no per-bounce exchange with other contexts — and a launch small enough to be bound by its longest rays
    const bool fused = fast && ctx->st.RayDepth == 2 && ctx->opt.deferLast != 0 && !f.outputAovs && !f.g.DoDebugBVHTraversal && !f.useTlas && ctx->instanceCount == 1 && !multiVer && !ctx->counters
                       && !ctx->capturePrimary && !ctx->groupExchange && !ctx->exchangeFn && !ctx->bandExchangeFn && !ctx->bandExchangeDevFn && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100)
                       && !(packet && ctx->opt.packet >= 2)             // (a forced packet walk is the tests' setting: nothing else takes the primary launch)
                       && !(wide_wanted(ctx) && ctx->opt.fused < 2)    // (the wide-node walk shortens the dependent chains the fused launch only stops paying launches for)
                       && want_fused(ctx, ctx->hCounts[MAX_DEPTH_SLOTS - 1], ctx->lastFast && ctx->lastBatch == B, B);
    f.packet = (packet && !fused) ? 1 : 0;
    if (f.packet) { int rc = packet_prepare(ctx); if (rc) { ctx->pending.clear(); return rc; } }
    f.hitsByRid = fused ? 1 : 0; f.shadeMin = ctx->opt.fusedShadeMin; f.scatterLog2 = ctx->opt.splitScatter;
    if (!fast && (B != 1 || multiVer)) { ctx->pending.clear(); return fail(ctx, IDKPT_ERR_UNKNOWN, "internal: generic path is never batched"); }
    unsigned long long* contMask = ctx->contMask.as<unsigned long long>();
    uint32_t* waveCounts = ctx->waveCounts.as<uint32_t>();
    const int BS = MAX_BATCH + 1;

    const uint8_t* tileClass = nullptr;                   // per-tile pre-classification of this batch (fast path with pre-cull only)
    // ---- FirstHit
    uint32_t* activeList = ctx->sortVals.as<uint32_t>(); // scratch (capacity uints), free until the first sort
    uint32_t* activeCount = counts + (MAX_DEPTH_SLOTS - 1);
    uint32_t* pmList = nullptr; uint32_t* const pmCount = counts + (MAX_DEPTH_SLOTS - 2); bool pmBounce = false;   // (RayDepth < MAX_DEPTH_SLOTS - 2, idkptSetSettings: the word is nobody's queue length — k_scan_blocks writes counts[1 .. RayDepth]; reset with the others)
    TraceBufs tr = {ctx->trRec.as<float4>(), nullptr, nullptr};
    TraceBufs trNone = {nullptr, nullptr, nullptr};
    uint32_t* waveLocal = waveCounts;                     // per-wave exclusive offset inside its 256-wave scan block
    uint32_t* blockSums = ctx->blockSums.as<uint32_t>();
    const uint32_t scanBlocks = ((total + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK;
    uint32_t* keysTmp = ctx->keysTmp.as<uint32_t>();
    {
        if (fast) {
            const uint32_t tilesX = ((uint32_t)f.W + 7) / 8, tilesY = ((uint32_t)f.rows + 7) / 8;
            const uint32_t genWaves = tilesX * tilesY;
            const int cull = f.g.DoTraceLights ? 0 : 1;
            // single instance without lights: k_trace2 reads nothing but the trace-ready record, so the planes of a surviving primary ray need not exist before k_shade_first
            const bool noLean = ctx->opt.noLeanPrimary != 0;
            const int lean = (cull && !f.useTlas && s.instanceCount == 1 && !noLean) ? 1 : 0;
            if (ctx->capturePrimary) hipLaunchKernelGGL(k_fill_miss, dim3((N + 255) / 256), dim3(256), 0, st, hits, (size_t)(B - 1) * Npad, N);
            tileClass = nullptr;
            if (cull && !ctx->opt.noTileCull) {   // sample-independent pre-classification of the 8x8 tiles (conservative whole-tile miss test)
                const uint32_t classSets = f.tilePerSample ? (uint32_t)B : 1u;       // one classification per camera / scene version
                HIPC(ctx->tileClass.ensure((size_t)genWaves * classSets));
                if (multiVer) hipLaunchKernelGGL((k_classify_tiles<true>), dim3((genWaves + 255) / 256, classSets), dim3(256), 0, st, s, f, ctx->tileClass.as<uint8_t>(), tilesX, tilesY);
                else hipLaunchKernelGGL((k_classify_tiles<false>), dim3((genWaves + 255) / 256, classSets), dim3(256), 0, st, s, f, ctx->tileClass.as<uint8_t>(), tilesX, tilesY);
                tileClass = ctx->tileClass.as<uint8_t>();
            }
            const int genMax = std::min(16, std::max(1, ctx->opt.genGroupMax)), genGroups = (B + genMax - 1) / genMax, genPer = (B + genGroups - 1) / genGroups;     // pixel-major: the batch in equal groups of at most 16 samples, one wave per sample
            const dim3 genGrid = f.genPixelMajor ? dim3(genGroups, genWaves) : dim3(B, (genWaves + 15) / 16), genBlock = f.genPixelMajor ? dim3(64 * genPer) : dim3(1024);
            if (multiVer) hipLaunchKernelGGL((k_gen_primary<true>), genGrid, genBlock, 0, st, s, f, rays, tr, cull, activeList, activeCount, keysTmp, ctx->contFlag.as<uint8_t>(), tileClass, lean);
            else hipLaunchKernelGGL((k_gen_primary<false>), genGrid, genBlock, 0, st, s, f, rays, tr, cull, activeList, activeCount, keysTmp, ctx->contFlag.as<uint8_t>(), tileClass, lean);
            TRACE_T0();
            uint32_t grid0 = traceGrid;
            if (ctx->opt.gridRaysX4 > 0 && ctx->lastFast && ctx->lastBatch == B) grid0 = small_launch_grid(traceGrid, ctx->hCounts[MAX_DEPTH_SLOTS - 1], 2, ctx->opt.gridRaysX4, midGrid);
            if (fused) {
                // FirstHit's traversal, its shading and the bounce's traversal in one persistent launch (kernels_trace_fused.hpp); the bounce's hits are stored per ray id
                hipLaunchKernelGGL((k_trace_fused<32>), dim3(grid0), dim3(WAVE), ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, work + 0, ctx->contFlag.as<uint8_t>(), keysTmp, lean);
            } else
            launch_trace2<true>(ctx, grid0, ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, work + 0, counters,
                                want_split(ctx, ctx->hCounts[MAX_DEPTH_SLOTS - 1], ctx->lastFast && ctx->lastBatch == B, B), 0);
            TRACE_T1();
            if (ctx->capturePrimary) { HIPC(ctx->primHit.ensure((size_t)N * 16)); hipLaunchKernelGGL(k_capture_primary, dim3((N + 255) / 256), dim3(256), 0, st, hits, (size_t)(B - 1) * Npad, N, ctx->primHit.as<float4>()); }
            if (fused) {}
            else {
                // the first bounce in the primary list's pixel-major order (kernels_shade.hpp k_shade_first): its own work list, hits per ray id
                // (only on sparse views — fewer than half of the pixels entered the traversal in the previous batch of this shape, the pooled leaf phase's rule: measured +1.1 % / +2.2 % on
                // the headline view with 32 / 20 samples in flight, and -1.9 % / -2.3 % where every pixel traverses: the alive queue's runs of 64 neighbouring pixels are coherent there already,
                // and this list is in the order the workgroups happened to append; option bounce_pixel_major: 0 never, 1 sparse views, 2 always)
                const bool sparseView = ctx->lastFast && ctx->lastBatch >= 1 && (uint64_t)ctx->hCounts[MAX_DEPTH_SLOTS - 1] * 2u < (uint64_t)ctx->W * ctx->rows * (uint64_t)ctx->lastBatch;   // (per sample: the previous batch may have had another size — a warm-up)
                pmBounce = f.genPixelMajor && (ctx->opt.bouncePixelMajor >= 2 || (ctx->opt.bouncePixelMajor == 1 && sparseView)) && depth >= 2 && !multiVer;
                if (pmBounce) { HIPC(ctx->pmList.ensure((size_t)total * 4)); pmList = ctx->pmList.as<uint32_t>(); }
                if (multiVer) hipLaunchKernelGGL((k_shade_first<true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, ctx->contFlag.as<uint8_t>(), keysTmp, lean, (uint32_t*)nullptr, (uint32_t*)nullptr);
                else hipLaunchKernelGGL((k_shade_first<false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, ctx->contFlag.as<uint8_t>(), keysTmp, lean, pmList, pmCount);
            }
            hipLaunchKernelGGL((k_scan_local<true>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, (const uint32_t*)nullptr, total, (const uint8_t*)ctx->contFlag.as<uint8_t>(), contMask, waveLocal, blockSums);
        } else {
            TRACE_T0();
            uint32_t g = std::min<uint32_t>(traceGrid, (N + 63) / 64);
            if (ctx->counters) { if (debug) hipLaunchKernelGGL((k_trace_primary<true, true>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); else hipLaunchKernelGGL((k_trace_primary<true, false>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); }
            else { if (debug) hipLaunchKernelGGL((k_trace_primary<false, true>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); else hipLaunchKernelGGL((k_trace_primary<false, false>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); }
            TRACE_T1();
            if (ctx->capturePrimary) { HIPC(ctx->primHit.ensure((size_t)N * 16)); hipLaunchKernelGGL(k_capture_primary, dim3((N + 255) / 256), dim3(256), 0, st, hits, (size_t)0, N, ctx->primHit.as<float4>()); }
            hipLaunchKernelGGL((k_shade<true, false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, trNone, hits, (const uint32_t*)nullptr, (const uint32_t*)nullptr, total, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                               contMask, waveCounts, keysTmp);
            hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, (const uint32_t*)nullptr, total, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
        }
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, (const uint32_t*)nullptr, total, blockSums, (const uint32_t*)waveLocal, counts + 1, (unsigned long long*)(1 < depth ? counters + 2 : nullptr),
                           (const unsigned long long*)contMask, (const uint32_t*)nullptr, Npad, B, bases + 1 * BS,
                           hostCounts + 1, hostBases + 1 * BS, (const uint32_t*)(counts + MAX_DEPTH_SLOTS - 1), hostCounts + MAX_DEPTH_SLOTS - 1);
        if (ctx->evBounce) HIPC(hipEventRecord(ctx->evBounce[1], st));      // bases[1] (alive counts entering bounce 1) are final
        hipLaunchKernelGGL((k_compact<true>), dim3(gridTotal), dim3(256), 0, st, (const uint32_t*)nullptr, (const uint32_t*)nullptr, total, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                           (const uint32_t*)keysTmp, ctx->queue[1].as<uint32_t>(), ctx->keys[1].as<uint32_t>());
    }
    int side = 1; // queue[side] holds the rays entering bounce j, its length is counts[j], sample k starts at bases[j][k]
    // may the last bounce's continuation wait until somebody asks for it?  (k_shade_last: only where a hit of that bounce cannot change the radiance and nothing else of it reaches the frame)
    const bool deferLast = fast && depth >= 2 && ctx->opt.deferLast != 0 && !f.outputAovs && !debug;
    const bool deferAllHits = !(ctx->sceneNoEmission && !(f.g.DoTraceLights && s.lightCount > 0));   // a hit of the last bounce may add radiance: emission somewhere in the scene, or light hits
    for (int j = 1; j < depth; j++) {
        uint32_t* q = ctx->queue[side].as<uint32_t>(); uint32_t* k = ctx->keys[side].as<uint32_t>();
        const uint32_t* cnt = counts + j;
        f.hitsByRid = (fused || (pmBounce && j == 1)) ? 1 : 0;                  // how this bounce's hit records are indexed (what its shading kernels are told)
        // exact multi-GPU deep paths: the host tells every sample how many alive rays the contexts above this strip hold (idkpt.h)
        const uint32_t* gbase = nullptr;
        const bool bandExchange = (ctx->bandExchangeFn || ctx->bandExchangeDevFn) && ctx->rowMod > 1 && !(ctx->st.DoRaySorting && j > 1);
        if (bandExchange) {}                                               // (below; a member of a multi-device context with interleaved rows takes this route as well)
        else if (ctx->groupExchange) {   // member of a multi-device context: the group sums the counts of the members that own earlier rows, on the device (idkpt_api.hpp)
            int rc = ctx->groupExchange(ctx->groupUser, ctx, j, B, &gbase); if (rc) { ctx->pending.clear(); return rc; }
        } else if (ctx->exchangeFn) {
            std::vector<uint32_t> hb(B + 1), local(B), outBases(B, 0u);
            HIPC(hipMemcpyAsync(hb.data(), bases + j * BS, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            for (int b2 = 0; b2 < B; b2++) local[b2] = hb[b2 + 1] - hb[b2];
            ctx->exchangeFn(ctx->exchangeUser, j, B, local.data(), outBases.data());
            HIPC(ctx->gbases.ensure((size_t)MAX_BATCH * 4));
            HIPC(hipMemcpyAsync(ctx->gbases.p, outBases.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
            HIPC(hipStreamSynchronize(st));                      // outBases is a stack vector
            gbase = ctx->gbases.as<uint32_t>();
        }
        if (bandExchange) {
            // interleaved rows / bands (idkpt.h idkptSetBandExchange): the rays of one local band are a contiguous run of a sample's queue segment (ordered compaction);
            // the host returns, per (sample, band), the alive rays of all contexts in the image bands before it; k_shade adds the position inside the run
            const int bandRows = 1 << ctx->rowBandLog2, LB = (ctx->rows + bandRows - 1) / bandRows;
            HIPC(ctx->bandTab.ensure((size_t)4 * MAX_BATCH * ((size_t)LB + 1) * 4));
            uint32_t* dStarts = ctx->bandTab.as<uint32_t>(); uint32_t* dTab = dStarts + (size_t)MAX_BATCH * (LB + 1);
            hipLaunchKernelGGL(k_band_starts, dim3((uint32_t)(((size_t)B * (LB + 1) + 255) / 256)), dim3(256), 0, st, (const uint32_t*)q, (const uint32_t*)(bases + j * BS), B, LB, (uint32_t)ctx->W * (uint32_t)bandRows, Npad, dStarts);
            if (ctx->bandExchangeDevFn) {
                // device-side variant: counts -> (the host enqueues its exchange on this stream) -> bases -> table; nothing waits on the host
                uint32_t* dCounts = dTab + (size_t)MAX_BATCH * (LB + 1); uint32_t* dBases = dCounts + (size_t)MAX_BATCH * (LB + 1);
                const uint32_t nb = (uint32_t)(((size_t)B * LB + 255) / 256);
                hipLaunchKernelGGL(k_band_counts, dim3(nb), dim3(256), 0, st, (const uint32_t*)dStarts, B, LB, dCounts);
                HIPC(hipGetLastError());
                ctx->bandExchangeDevFn(ctx->bandExchangeDevUser, j, B, LB, dCounts, dBases, (void*)st);
                hipLaunchKernelGGL(k_band_tab, dim3(nb), dim3(256), 0, st, (const uint32_t*)dStarts, (const uint32_t*)dBases, B, LB, dTab);
                gbase = dTab; f.gbStride = LB; f.gbBands = 1;
            } else {
            std::vector<uint32_t> starts((size_t)B * (LB + 1)), local((size_t)B * LB), outBases((size_t)B * LB, 0u), tab((size_t)B * LB);
            HIPC(hipMemcpyAsync(starts.data(), dStarts, starts.size() * 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            for (int k2 = 0; k2 < B; k2++) for (int b2 = 0; b2 < LB; b2++) local[(size_t)k2 * LB + b2] = starts[(size_t)k2 * (LB + 1) + b2 + 1] - starts[(size_t)k2 * (LB + 1) + b2];
            ctx->bandExchangeFn(ctx->bandExchangeUser, j, B, LB, local.data(), outBases.data());
            for (int k2 = 0; k2 < B; k2++) for (int b2 = 0; b2 < LB; b2++) tab[(size_t)k2 * LB + b2] = outBases[(size_t)k2 * LB + b2] - starts[(size_t)k2 * (LB + 1) + b2];   // (mod 2^32: + position inside the sample's segment = global slot)
            HIPC(hipMemcpyAsync(dTab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
            HIPC(hipStreamSynchronize(st));                      // tab is a stack vector
            gbase = dTab; f.gbStride = LB; f.gbBands = 1;
            }
        }
        if (!bandExchange) { f.gbStride = 1; f.gbBands = 0; }
        if (ctx->st.DoRaySorting && j > 1) {
            // RaySorting() (PathTracer.cs:232-237): stable sort of (key, rayIndex); key = 21-bit triangle id with the batch's
            // sample index above it, so one sort orders every sample's queue exactly like a stand-alone counting sort
            const uint32_t nTiles = (total + SORT_TILE - 1) / SORT_TILE;
            int sampleBits = 0; while ((1 << sampleBits) < B) sampleBits++;
            const int passes = (IDKPT_SORT_KEY_BITS + sampleBits + 6) / 7;    // 7-bit digits over key + sample index: 3 passes alone, 4 up to 128 samples, 5 up to 256
            uint32_t* digitTotals = ctx->sortHist.as<uint32_t>() + (size_t)SORT_RADIX * nTiles;   // 128 words behind the [digit][tile] table
            uint32_t* ka = k; uint32_t* va = q; uint32_t* kb = ctx->sortKeys.as<uint32_t>(); uint32_t* vb = ctx->sortVals.as<uint32_t>();
            for (int pass = 0; pass < passes; pass++) {
                hipLaunchKernelGGL(k_sort_hist, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)ka, cnt, (uint32_t)(7 * pass), ctx->sortHist.as<uint32_t>(), nTiles);
                hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(1024), 0, st, cnt, ctx->sortHist.as<uint32_t>(), nTiles, digitTotals);
                hipLaunchKernelGGL(k_sort_scatter, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)ka, (const uint32_t*)va, cnt, (uint32_t)(7 * pass), (const uint32_t*)ctx->sortHist.as<uint32_t>(), nTiles, (const uint32_t*)digitTotals, kb, vb);
                std::swap(ka, kb); std::swap(va, vb);
            }
            // odd pass count: the sorted data sits in (sortKeys, sortVals) -> copy the indices back (the reference copies W*H*4 B too, PathTracer.cs:296)
            if (va != q) HIPC(hipMemcpyAsync(q, va, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
        }
        if (!fused) {
        TRACE_T0();
        // grid of the bounce launch: its queue length is only known on the device; the length the same bounce had in the previous batch (pinned copy,
        // possibly one batch stale) is a good predictor, and a grid that is too small or too large only costs time (the waves are persistent)
        uint32_t gridj = traceGrid;
        const int hintMul = ctx->opt.gridHint;
        if (hintMul > 0 && ctx->lastBatch == B && ctx->hBases) gridj = small_launch_grid(traceGrid, ctx->hBases[(size_t)j * BS + B], hintMul, ctx->opt.gridRaysX4, midGrid);
        if (fast && pmBounce && j == 1) {   // the list k_shade_first wrote: ray ids, hits stored per ray id (the PRIMARY instantiations read exactly that)
            Frame ft = f; ft.hitsByRid = 0;
            launch_trace2<true>(ctx, gridj, ldsBytes, st, s, ft, rays, tr, hits, (const uint32_t*)pmList, (const uint32_t*)pmCount, work + j, counters,
                                want_split(ctx, ctx->hBases ? ctx->hBases[(size_t)j * BS + B] : 0u, ctx->lastFast && ctx->lastBatch == B && ctx->hBases != nullptr, B), j);
        } else
        if (fast) launch_trace2<false>(ctx, gridj, ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)q, cnt, work + j, counters,
                                       want_split(ctx, ctx->hBases ? ctx->hBases[(size_t)j * BS + B] : 0u, ctx->lastFast && ctx->lastBatch == B && ctx->hBases != nullptr, B), j);
        else {
            if (ctx->counters) hipLaunchKernelGGL((k_trace_queue<true>), dim3(traceGrid), dim3(WAVE), ldsBytes, st, s, f, rays, hits, (const uint32_t*)q, cnt, work + j, counters);
            else hipLaunchKernelGGL((k_trace_queue<false>), dim3(traceGrid), dim3(WAVE), ldsBytes, st, s, f, rays, hits, (const uint32_t*)q, cnt, work + j, counters);
        }
        TRACE_T1();
        }
        if (deferLast && j == depth - 1 && gbase == nullptr) {
            // the last bounce: only its radiance is visible in the frame (kernels_shade.hpp k_shade_last); state, queue and counts follow on demand (finish_deferred)
            HIPC(ctx->radSave.ensure((size_t)ctx->maxBatch * ctx->Npad * 16)); HIPC(ctx->deferCount.ensure(64));
#define SHADE_LAST(A, V) hipLaunchKernelGGL((k_shade_last<A, V>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, hits, (const uint32_t*)q, cnt, (const uint32_t*)(bases + j * BS), ctx->radSave.as<float4>(), ctx->deferCount.as<uint32_t>())
            if (deferAllHits) { if (multiVer) SHADE_LAST(true, true); else SHADE_LAST(true, false); }
            else SHADE_LAST(false, false);                              // (misses only: the sky is not versioned)
#undef SHADE_LAST
            ctx->defer.allHits = deferAllHits; ctx->defer.valid = true; ctx->defer.j = j; ctx->defer.side = side; ctx->defer.B = B; ctx->defer.total = total; ctx->defer.Npad = Npad;
            break;
        }
        if (multiVer) hipLaunchKernelGGL((k_shade<false, true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, fast ? tr : trNone, hits, (const uint32_t*)q, cnt, 0u, (const uint32_t*)(bases + j * BS), gbase, contMask, waveCounts, keysTmp);
        else hipLaunchKernelGGL((k_shade<false, false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, fast ? tr : trNone, hits, (const uint32_t*)q, cnt, 0u, (const uint32_t*)(bases + j * BS), gbase, contMask, waveCounts, keysTmp);
        hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, cnt, 0u, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, cnt, 0u, blockSums, (const uint32_t*)waveLocal, counts + j + 1, (unsigned long long*)(j + 1 < depth ? counters + 2 : nullptr),
                           (const unsigned long long*)contMask, (const uint32_t*)(bases + j * BS), Npad, B, bases + (j + 1) * BS,
                           hostCounts + j + 1, hostBases + (size_t)(j + 1) * BS, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        if (ctx->evBounce) HIPC(hipEventRecord(ctx->evBounce[j + 1], st));
        hipLaunchKernelGGL((k_compact<false>), dim3(gridTotal), dim3(256), 0, st, (const uint32_t*)q, cnt, 0u, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                           (const uint32_t*)keysTmp, ctx->queue[1 - side].as<uint32_t>(), ctx->keys[1 - side].as<uint32_t>());
        side = 1 - side;
    }
    ctx->lastQueueSide = side; ctx->lastQueueCountSlot = depth; ctx->lastFast = fast; ctx->lastNeedsRegen = fast; ctx->lastBatch = B; ctx->lastFrame = f;
    hipLaunchKernelGGL(k_final_draw, dim3((N + 255) / 256), dim3(256), 0, st, s, f, rays, image_ptr(ctx, 0, 0), image_ptr(ctx, 1, 0), image_ptr(ctx, 2, 0), N, tileClass,
                       work, (uint32_t)WORK_WORDS, counts, (uint32_t)MAX_DEPTH_SLOTS);
    HIPC(hipGetLastError());
    ctx->countersDirty = false;
#ifdef IDKPT_DEVELOPER
    if (capturing) {
        hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
        const int K = ctx->opt.graphProbe; ctx->opt.graphProbe = 0;
        if (hipStreamEndCapture(st, &g) == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
            hipEvent_t e0 = nullptr, e1 = nullptr; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipGraphLaunch(ex, st); (void)hipStreamSynchronize(st);                       // the batch itself (a capture does not execute)
            (void)hipEventRecord(e0, st);
            for (int k = 0; k < K; k++) (void)hipGraphLaunch(ex, st);
            (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
            float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1);
            size_t nodes = 0; (void)hipGraphGetNodes(g, nullptr, &nodes);
            fprintf(stderr, "[idkpt graph] batch of %d sample(s) captured: %zu graph nodes; %d replays: %.1f us per batch\n", B, nodes, K, ms * 1000.0f / (float)K);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipGraphExecDestroy(ex);
        } else fprintf(stderr, "[idkpt graph] capture or instantiation failed: %s\n", hipGetErrorString(hipGetLastError()));
        if (g) (void)hipGraphDestroy(g);
    }
#endif
    // queue lengths stay on the GPU during the batch; k_scan_blocks mirrors them into host-mapped memory for GetStats and the queue downloads (no copy, no sync here)
    if (ctx->timing) HIPC(hipEventRecord(ctx->evFrame[1], st));
    ctx->stats.Frames += (uint64_t)B;
    ctx->stats.PrimaryRays += (uint64_t)N * (uint64_t)B;
    ctx->pending.clear();
    return IDKPT_OK;
}

static int32_t dev_Render(dev_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: no scene uploaded");
    if (ctx->W <= 0 || !ctx->frameOk) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: no frame buffers (idkptSetSize not called, or its allocation failed)");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: UseTlas set but no TLAS nodes uploaded");
    HIPC(hipSetDevice(ctx->device));
    if (ctx->timing && ctx->evUsed > 4096) { HIPC(hipStreamSynchronize(ctx->stream)); resolve_trace_events(ctx); }
    // every sample is deferred; a batch is launched as soon as maxBatch samples are pending (or on any call that needs
    // results).  The general path (multi-instance / TLAS / debug cost) is launched sample by sample.
    const int limit = fast_path(ctx) ? ctx->maxBatch : 1;
    for (int i = 0; i < ctx->st.SamplesPerPixel; i++) {
        PendingSample ps; ps.accum = ctx->accum[ctx->curSlot]++; ps.slot = ctx->curSlot;
        memcpy(ps.cam, ctx->invProj, 64); memcpy(ps.cam + 16, ctx->invView, 64); memcpy(ps.cam + 32, ctx->viewPos, 12); ps.cam[35] = 0.0f;
        for (int b = 0; b < VB_COUNT; b++) ps.vs[b] = (uint8_t)ctx->vcur[b];           // the state of the geometry this sample sees
        ctx->pending.push_back(ps);
        if (!ctx->grouped && (int)ctx->pending.size() >= limit) { int rc = flush_batch(ctx); if (rc) return rc; }   // (members of a multi-device context: the group launches)
    }
    return IDKPT_OK;
}

static int32_t dev_Synchronize(dev_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; HIPC(hipSetDevice(ctx->device)); FLUSH_KEEP(); SYNC_CHECKED(); return IDKPT_OK; }

// Launches whatever is pending without waiting for it (lets a host overlap its own work with the GPU).
static int32_t dev_Flush(dev_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; HIPC(hipSetDevice(ctx->device)); FLUSH_KEEP(); return IDKPT_OK; }

static int32_t dev_SetMaxBatch(dev_ctx* ctx, int32_t maxBatch)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(maxBatch >= 1 && maxBatch <= MAX_BATCH, "idkptSetMaxBatch: 1..256");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();   // (the wavefront buffers are about to be reallocated: a deferred last bounce is completed first)
    HIPC(hipStreamSynchronize(ctx->stream));
    if (maxBatch == ctx->maxBatch) return IDKPT_OK;
    const int previous = ctx->maxBatch;
    ctx->maxBatch = maxBatch;
    if (ctx->W > 0) {
        std::vector<uint32_t> acc = ctx->accum; int slot = ctx->curSlot; const bool started = ctx->ringStarted;
        int rc = alloc_frame_keep_images(ctx);
        if (rc) {   // e.g. out of device memory: fall back to the previous (smaller) buffer set; the accumulation restarts
            const std::string why = ctx->lastError;
            ctx->maxBatch = previous;
            (void)alloc_frame(ctx);
            return fail(ctx, rc, "idkptSetMaxBatch: could not allocate the wavefront buffers for " + std::to_string(maxBatch) + " samples in flight (" + why + "); kept " + std::to_string(previous));
        }
        ctx->accum = acc; ctx->curSlot = slot; ctx->ringStarted = started;
    }
    return IDKPT_OK;
}

// idkptSetSceneVersions: how many states of the geometry may be in flight (1: a scene update launches every queued sample first, as the reference's frame loop does)
static int32_t dev_SetSceneVersions(dev_ctx* ctx, int32_t versions)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(versions >= 1 && versions <= 64, "idkptSetSceneVersions: 1..64 versions");
    HIPC(hipSetDevice(ctx->device));
    if (versions == ctx->verSlots) return IDKPT_OK;
    FLUSH();                                                           // nothing queued or deferred: every buffer has exactly one live state, its current one
    if (versions < ctx->verSlots) {
        for (int b = 0; b < VB_COUNT; b++) {
            if (ctx->vcur[b] >= versions && ctx->vbytes[b] > 0) { HIPC(hipMemcpyAsync(vb_ptr(ctx, b, 0), vb_ptr(ctx, b, ctx->vcur[b]), ctx->vbytes[b], hipMemcpyDeviceToDevice, ctx->stream)); ctx->vcur[b] = 0; }
            ctx->valloc[b] = std::min(ctx->valloc[b], versions);
        }
        HIPC(hipStreamSynchronize(ctx->stream));
    }
    ctx->verSlots = versions;
    return IDKPT_OK;
}

static int32_t dev_SetFrameRing(dev_ctx* ctx, int32_t frames)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(frames >= 1 && frames <= 128, "idkptSetFrameRing: 1..128 frames");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();   // (the wavefront buffers are about to be reallocated: a deferred last bounce is completed first)
    HIPC(hipStreamSynchronize(ctx->stream));
    if (frames == ctx->ringSize) return IDKPT_OK;
    ctx->ringSize = frames;
    if (ctx->W > 0) return alloc_frame(ctx);       // images are re-created (cleared); accumulation restarts in slot 0
    ctx->accum.assign(frames, 0u); ctx->curSlot = 0; ctx->ringStarted = false;
    return IDKPT_OK;
}

static int32_t dev_BeginFrame(dev_ctx* ctx, int32_t* outSlot)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (ctx->ringStarted) ctx->curSlot = (ctx->curSlot + 1) % ctx->ringSize;   // the first frame after idkptSetFrameRing / idkptSetSize uses slot 0
    ctx->ringStarted = true;
    ctx->accum[ctx->curSlot] = 0;                   // a new frame: its first sample overwrites whatever the slot held
    if (outSlot) *outSlot = ctx->curSlot;
    return IDKPT_OK;
}
