// host_launch.hpp — derivation of the wide nodes and the one place that picks a traversal kernel for a launch (launch_trace2).
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

// ---- wide nodes (kernels_wide.hpp): which launches use them, and their derivation on the device ----------------------------------------------------------------
// One-BLAS scenes, closest hit, the reference's counters not asked for, one scene version: everything else keeps k_trace2.
static bool wide_wanted(const dev_ctx* ctx) { return ctx->opt.wide != 0 && ctx->instanceCount == 1 && !ctx->st.UseTlas && ctx->verSlots == 1 && !ctx->counters && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100 || ctx->opt.traceVariant == 213); }
static int wide_stack_rows(const dev_ctx* ctx) { return std::min(96, std::max(4, ctx->opt.wideCap > 0 ? ctx->opt.wideCap : 24)); }
// (re-)derives what is stale: the children lists after an upload / a node patch (k_wide_topo, one workgroup per BLAS), box bytes and leaf records after anything that
// moved boxes or positions (k_wide_fill).  Stream-ordered in front of the batch that is about to be launched; with one scene version every update launches the queued samples first.
static char* vb_ptr(dev_ctx* ctx, int b, int slot);
// totals of the optional walks since idkptResetStats (idkpt_stats.Wide*, InstTlasFlaggedRays, Packet*): sixteen 64-bit words ([0..3] wide, [4] own TLAS, [8..13] packets), zeroed when first needed
// LDS rows behind the BLAS stack: the own TLAS's stack, or an instance mask of 32 x that many bits.  A tree over n leaves is never deeper than n; once k_tlas_build has reported the depth of the
// tree it built (host-mapped, possibly one rebuild stale: a ray that needs more rows than it gets is flagged and traced by the exact loop) the rows are that depth, and what the mask needs
static int inst_tlas_rows(const dev_ctx* ctx)
{
    if (ctx->itlasDepth > 0 && ctx->itlasBuilt) return std::min(TLAS_STACK_SIZE, std::max((ctx->instanceCount + 31) / 32, ctx->itlasDepth));
    return std::min(TLAS_STACK_SIZE, std::max(1, ctx->instanceCount));
}
#define TOTALS_BYTES 128
static int totals_ensure(dev_ctx* ctx) { if (!ctx->wtotals.p) { HIPC(ctx->wtotals.ensure(TOTALS_BYTES)); HIPC(hipMemsetAsync(ctx->wtotals.p, 0, TOTALS_BYTES, ctx->stream)); } return IDKPT_OK; }
static int wide_prepare(dev_ctx* ctx)
{
    if (ctx->wideTopoValid && ctx->wideFillValid) return IDKPT_OK;
    { int rc = totals_ensure(ctx); if (rc) return rc; }
    const size_t nb = ctx->hDescs.size();
    hipStream_t st = ctx->stream;
    const float4* nodes = (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]);
    const float4* triVerts = (const float4*)vb_ptr(ctx, VB_TRIVERTS, ctx->vcur[VB_TRIVERTS]);
    if (!ctx->wideTopoValid) {
        ctx->wNodeOff.assign(nb + 1, 0u); ctx->wLeafOff.assign(nb + 1, 0u);
        for (size_t b = 0; b < nb; b++) {
            const GpuBlasDesc& d = ctx->hDescs[b];
            const uint32_t pairs = (uint32_t)std::max(0, d.NodeCount) / 2u + 1u, leavesMax = pairs + 1u;                       // a wide node stands for at least one pair; a tree of L leaves has L - 1 internal nodes
            ctx->wNodeOff[b + 1] = ctx->wNodeOff[b] + pairs;
            ctx->wLeafOff[b + 1] = ctx->wLeafOff[b] + 5u * leavesMax + 3u * (uint32_t)std::max(0, d.TriangleCount) + 4u;   // 2 + 3 per leaf-range triangle (a leaf pair may share one triangle)
        }
        HIPC(ctx->wnodes.ensure((size_t)ctx->wNodeOff[nb] * 64 + 64)); HIPC(ctx->wids.ensure((size_t)ctx->wNodeOff[nb] * 16 + 16)); HIPC(ctx->wpair.ensure((size_t)ctx->wNodeOff[nb] * 4 + 16));
        HIPC(ctx->wleaf.ensure((size_t)ctx->wLeafOff[nb] * 16 + 80)); HIPC(ctx->wcounts.ensure(nb * 8 + 8));
        for (size_t b = 0; b < nb; b++) {
            const GpuBlasDesc& d = ctx->hDescs[b];
            hipLaunchKernelGGL(k_wide_topo, dim3(1), dim3(WIDE_TOPO_THREADS), 0, st, nodes + 2 * (size_t)d.NodeOffset, (uint32_t)d.NodeCount, ctx->wpair.as<uint32_t>() + ctx->wNodeOff[b],
                               ctx->wids.as<uint4>() + ctx->wNodeOff[b], ctx->wnodes.as<uint4>() + 4 * (size_t)ctx->wNodeOff[b], ctx->wcounts.as<uint32_t>() + 2 * b);
        }
        HIPC(hipGetLastError());
        ctx->wideTopoValid = true; ctx->wideFillValid = false;
    }
    for (size_t b = 0; b < nb; b++) {
        const GpuBlasDesc& d = ctx->hDescs[b];
        const uint32_t pairs = ctx->wNodeOff[b + 1] - ctx->wNodeOff[b];
        hipLaunchKernelGGL(k_wide_fill, dim3((pairs + 255) / 256), dim3(256), 0, st, nodes + 2 * (size_t)d.NodeOffset, triVerts + 3 * (size_t)d.TriangleOffset, (const uint4*)(ctx->wids.as<uint4>() + ctx->wNodeOff[b]),
                           ctx->wnodes.as<uint4>() + 4 * (size_t)ctx->wNodeOff[b], ctx->wleaf.as<float4>() + ctx->wLeafOff[b], (const uint32_t*)(ctx->wcounts.as<uint32_t>() + 2 * b));
    }
    HIPC(hipGetLastError());
    ctx->wideFillValid = true;
    return IDKPT_OK;
}

// ---- instance records (DScene::instRec, k_inst_records) for the walk through the library's own TLAS: stream-ordered in front of the launch that reads them -----------
static int inst_records_prepare(dev_ctx* ctx)
{
    if (ctx->instRecValid || ctx->instanceCount < 2 || ctx->verSlots != 1) return IDKPT_OK;
    HIPC(ctx->instRec.ensure((size_t)ctx->instanceCount * 96));
    hipLaunchKernelGGL(k_inst_records, dim3((ctx->instanceCount + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]), ctx->descs.as<GpuBlasDesc>(),
                       ctx->instances.as<GpuBlasInstance>(), (const float4*)vb_ptr(ctx, VB_XFORMS, ctx->vcur[VB_XFORMS]), ctx->instanceCount, ctx->instRec.as<float4>());
    HIPC(hipGetLastError());
    ctx->instRecValid = true;
    return IDKPT_OK;
}

// ---- the per-triangle marks "not contained in its leaf box" (k_mark_triangles) of the walks that re-order a ray's candidates (k_trace_inst, k_trace_packet): derived on the
// device before the first launch that wants them and after everything that moved boxes or positions (one scene version only)
static int chunks_ensure(dev_ctx* ctx)
{
    if (ctx->ichunkCount != 0) return IDKPT_OK;   // the flattened (BLAS, chunk of 256 nodes) table: once per upload
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> tab;
    for (size_t b = 0; b < ctx->hDescs.size(); b++) for (int first = 0; first < ctx->hDescs[b].NodeCount; first += 256) { tab.push_back((uint32_t)b); tab.push_back((uint32_t)first); }
    ctx->ichunkCount = (uint32_t)(tab.size() / 2);
    HIPC(ctx->ichunks.ensure(tab.size() * 4 + 8));
    if (!tab.empty()) { HIPC(hipMemcpyAsync(ctx->ichunks.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }   // (tab is a stack vector)
    return IDKPT_OK;
}
static int marks_prepare(dev_ctx* ctx)
{
    if (ctx->imarksValid) return IDKPT_OK;
    hipStream_t st = ctx->stream;
    { int rc = chunks_ensure(ctx); if (rc) return rc; }
    HIPC(ctx->imarks.ensure((size_t)std::max(1, ctx->triCount)));
    HIPC(hipMemsetAsync(ctx->imarks.p, 0, (size_t)std::max(1, ctx->triCount), st));
    if (ctx->ichunkCount) hipLaunchKernelGGL(k_mark_triangles, dim3(ctx->ichunkCount), dim3(256), 0, st, (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]), (const float4*)vb_ptr(ctx, VB_TRIVERTS, ctx->vcur[VB_TRIVERTS]),
                                             ctx->descs.as<GpuBlasDesc>(), (const uint2*)ctx->ichunks.as<uint2>(), ctx->imarks.as<uint8_t>());
    HIPC(hipGetLastError());
    ctx->imarksValid = true;
    return IDKPT_OK;
}

// ---- DScene::pairNodes (k_pair_nodes) for k_trace2's FAST node step: one-BLAS scenes, one scene version; stream-ordered in front of the batch that is about to be launched ----------
static bool pair_nodes_wanted(const dev_ctx* ctx)
{
    return ctx->opt.pairNodes != 0 && ctx->instanceCount == 1 && !ctx->st.UseTlas && !ctx->st.Gpu.DoDebugBVHTraversal && ctx->verSlots == 1 && !ctx->counters && !ctx->opt.forceGeneric
           && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100);
}
static int pair_nodes_prepare(dev_ctx* ctx)
{
    if (ctx->pairValid || !pair_nodes_wanted(ctx)) return IDKPT_OK;
    { int rc = chunks_ensure(ctx); if (rc) return rc; }
    HIPC(ctx->pairNodes.ensure((size_t)std::max(1, ctx->nodeCount) * 32 + 64));
    if (ctx->ichunkCount) hipLaunchKernelGGL(k_pair_nodes, dim3(ctx->ichunkCount), dim3(256), 0, ctx->stream, (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]), ctx->descs.as<GpuBlasDesc>(),
                                             (const uint2*)ctx->ichunks.as<uint2>(), ctx->pairNodes.as<float4>());
    HIPC(hipGetLastError());
    ctx->pairValid = true;
    return IDKPT_OK;
}

// ---- the packet walk (kernels_packet.hpp): which primary launches use it ---------------------------------------------------------------------------------------------
// One BLAS instance whose boxes nest (the walk's lenient inner-box test needs it), closest hit, one scene version, the reference's counters not asked for, stock kernels; option packet = 1 additionally wants a pixel-major list and the
// kernel's own counters in its favour (packet_decide).
static bool packet_possible(const dev_ctx* ctx)
{
    return ctx->opt.packet != 0 && (ctx->instanceCount == 1 || (ctx->uniValid && ctx->uniMode == 1 && ctx->itlasValid && ctx->itlasBuilt)) && ctx->sceneNested && !ctx->st.UseTlas && !ctx->st.Gpu.DoDebugBVHTraversal && ctx->verSlots == 1 && !ctx->counters && !ctx->opt.forceGeneric
           && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100);      // (with option "wide" as well: the packet walk takes the primary launch, the wide-node walk the bounces)
}
// Called once per batch whose primary launch could be a packet launch (flush_batch): reads what the kernel's counters said so far (host-mapped, written by k_packet_mirror behind every
// packet launch: possibly a batch or two stale) and moves the decision: probing -> on / off by the live lanes per node step; on -> off when the view's coherence drops; off -> probing
// when the camera, the frame size or the batch size changed (not more often than every 16 batches) or after 512 batches.  The decision only moves time, never a result.
static bool packet_decide(dev_ctx* ctx, const Frame& f)
{
    if (!packet_possible(ctx)) return false;
    if (ctx->opt.packet >= 2) return true;
    if (!f.genPixelMajor) return false;
    if (!ctx->hPkStats) return true;                                     // first packet launch of this context: probing
    const volatile unsigned long long* h = ctx->hPkStats;
    const unsigned long long steps = h[2], live = h[3];
    if (steps < ctx->pkSeen[2]) { for (int i = 0; i < 6; i++) ctx->pkSeen[i] = 0; }                     // (idkptResetStats zeroed the totals)
    const unsigned long long dSteps = steps - ctx->pkSeen[2], dLive = live - ctx->pkSeen[3];
    const bool camSame = memcmp(ctx->pkCam, ctx->pending[0].cam, sizeof(ctx->pkCam)) == 0 && ctx->pkW == ctx->W && ctx->pkRows == ctx->rows && ctx->pkBatch == f.batch;
    ctx->pkBatchesSinceProbe++;
    if (ctx->pkState == 2) {
        if ((!camSame && ctx->pkBatchesSinceProbe >= 16) || ctx->pkBatchesSinceProbe >= 512) { ctx->pkState = 0; ctx->pkBatchesSinceProbe = 0; ctx->pkSeen[2] = steps; ctx->pkSeen[3] = live; }
    } else if (dSteps >= 4096ull) {                                       // enough node steps since the last look
        const float frac = (float)((double)dLive / (64.0 * (double)dSteps));
        ctx->pkLastLive = frac;
        ctx->pkState = frac * 100.0f >= (float)ctx->opt.packetMinLive ? 1 : 2;
        if (ctx->pkState == 2) ctx->pkBatchesSinceProbe = 0;
        ctx->pkSeen[2] = steps; ctx->pkSeen[3] = live;
    }
    memcpy(ctx->pkCam, ctx->pending[0].cam, sizeof(ctx->pkCam)); ctx->pkW = ctx->W; ctx->pkRows = ctx->rows; ctx->pkBatch = f.batch;
    return ctx->pkState != 2;
}
static int packet_prepare(dev_ctx* ctx)
{
    { int rc = totals_ensure(ctx); if (rc) return rc; }
    if (!ctx->hPkStats) { HIPC(hipHostMalloc((void**)&ctx->hPkStats, 64, hipHostMallocMapped)); memset(ctx->hPkStats, 0, 64); HIPC(hipHostGetDevicePointer((void**)&ctx->dPkStats, ctx->hPkStats, 0)); }
    return marks_prepare(ctx);
}

// ---- the library's own TLAS for the instance loop (kernels_trace_inst.hpp) ---------------------------------------------------------------------------------------
// Several instances, no UseTlas, closest hit, one scene version, the reference's counters not asked for: everything else keeps the exact loop (k_trace2 MODE 1).
static bool inst_tlas_wanted(const dev_ctx* ctx, bool sieve = false /* the same question for the exact loop with the instance sieve (option inst_sieve) */)
{
    const int from = sieve ? ctx->opt.instSieve : ctx->opt.instTlas;
    if (ctx->instanceCount > (sieve ? 1024 : 4096)) return false;       // (a lane's mask has 32 LDS rows; k_tlas_build is one workgroup: beyond a few thousand instances its cost per transform update is not the loop's business)
    return from > 0 && ctx->instanceCount >= std::max(2, from) && !ctx->st.UseTlas && !ctx->st.Gpu.DoDebugBVHTraversal && ctx->verSlots == 1 && !ctx->counters
           && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100);
}
// ---- the unified tree (kernels_trace_inst.hpp UNI; k_braid / k_unify_* in kernels_scene.hpp) -----------------------------------------------------------------------------
// Same conditions as the own TLAS, from two instances on; additionally the BLAS boxes nest, every BLAS is used by at most one instance (a scene-wide triangle index then names its
// instance), and every instance carries the same InvModel — all decided on the host (instances and transforms only ever arrive through it).
static bool inst_unify_wanted(const dev_ctx* ctx)
{
    if (ctx->opt.instUnify <= 0 || ctx->instanceCount < 2 || ctx->instanceCount > 1024 || !ctx->sceneNested) return false;
    return !ctx->st.UseTlas && !ctx->st.Gpu.DoDebugBVHTraversal && ctx->verSlots == 1 && !ctx->counters && (ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100);
}
static int inst_unify_prepare(dev_ctx* ctx)
{
    ctx->uniValid = false; ctx->uniMode = 0;
    if (!inst_unify_wanted(ctx)) return IDKPT_OK;
    const int n = ctx->instanceCount, nb = (int)ctx->hDescs.size();
    hipStream_t st = ctx->stream;
    if (!ctx->uniTabsValid) {                           // once per upload: each BLAS used at most once?  The per-BLAS tables in ascending order of TriangleOffset
        std::vector<int> user(nb, -1);
        bool ok = (int)ctx->hInstances.size() == n;
        for (int i = 0; i < n && ok; i++) { const int b = (int)ctx->hInstances[i].BlasId; if (b < 0 || b >= nb || user[b] >= 0) ok = false; else user[b] = i; }
        ctx->uniEligible = ok; ctx->uniTabsValid = true;
        if (ok) {
            std::vector<int> order(nb); for (int b = 0; b < nb; b++) order[b] = b;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return ctx->hDescs[a].TriangleOffset < ctx->hDescs[b].TriangleOffset; });
            std::vector<uint32_t> tab(2 * (size_t)nb);
            for (int k = 0; k < nb; k++) { tab[k] = (uint32_t)ctx->hDescs[order[k]].TriangleOffset; tab[nb + k] = user[order[k]] >= 0 ? (uint32_t)ctx->hInstances[user[order[k]]].MeshTransformId : 0u; }
            HIPC(ctx->uTabs.ensure(tab.size() * 4 + 16));
            HIPC(hipMemcpyAsync(ctx->uTabs.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st));   // (tab is a stack vector)
        }
    }
    bool same = ctx->uniEligible && (int)ctx->hInstances.size() == n;
    if (same) {   // every instance's InvModel (rows 3-5 of its GpuMeshTransform) is instance 0's, bit for bit?  On the host's copy: transforms only ever arrive through the host
        const size_t xb = sizeof(GpuMeshTransform);
        const size_t t0 = (size_t)ctx->hInstances[0].MeshTransformId;
        if ((t0 + 1) * xb > ctx->hXforms.size()) same = false;
        for (int i = 1; i < n && same; i++) {
            const size_t ti = (size_t)ctx->hInstances[i].MeshTransformId;
            if ((ti + 1) * xb > ctx->hXforms.size() || memcmp(ctx->hXforms.data() + ti * xb + 48, ctx->hXforms.data() + t0 * xb + 48, 48) != 0) same = false;
        }
    }
    // one space: the unified tree (TREE 1).  Otherwise — different transforms, or a BLAS instanced several times — the general array (TREE 2, option inst_general): a world-space top
    // whose entries take the ray into their instance's space
    const int mode = same ? 1 : (ctx->opt.instGeneral > 0 && n >= ctx->opt.instGeneral ? 2 : 0);
    if (mode == 0) return IDKPT_OK;
    if (!ctx->hUni) { HIPC(hipHostMalloc((void**)&ctx->hUni, 64, hipHostMallocMapped)); memset(ctx->hUni, 0, 64); HIPC(hipHostGetDevicePointer((void**)&ctx->dUni, ctx->hUni, 0)); }
    { int rc = chunks_ensure(ctx); if (rc) return rc; }
    const float4* nodes = (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]);
    const float4* xf = (const float4*)vb_ptr(ctx, VB_XFORMS, ctx->vcur[VB_XFORMS]);
    const int cap = std::max(n, std::min(ctx->opt.instUnify, 16384)), nodeCount = 2 * cap - 1;
    // the array: [0, 2 cap) the top; TREE 2: [2 cap, 4 cap) one stub pair per entry, then the RESTORE pair; then every BLAS's nodes
    const uint32_t stubBase = 2u * (uint32_t)cap, restoreIdx = 4u * (uint32_t)cap, baseB = mode == 2 ? 4u * (uint32_t)cap + 2u : 2u * (uint32_t)cap;
    // k_braid's entries, areas, leaf boxes, count; k_tlas_build's scratch; the PLOC top; the unified nodes
    const size_t entOff = 0, areaOff = entOff + (size_t)cap * 8, bleafOff = (areaOff + (size_t)cap * 4 + 15) & ~(size_t)15, cntOff = bleafOff + (size_t)cap * 32;
    const size_t ubOff = cntOff + 16;
    const size_t scOff = (ubOff + (size_t)cap * 4 + 255) & ~(size_t)255, leafOff = (size_t)nodeCount * 32, keyOff = leafOff + (size_t)cap * 32, prefOff = keyOff + (size_t)cap * 4;
    HIPC(ctx->uniBuf.ensure(scOff + prefOff + (size_t)nodeCount * 4)); HIPC(ctx->utlas.ensure((size_t)nodeCount * 32));      // (its own buffers: the caller's launches on tlasScratch / braidBuf are in flight)
    HIPC(ctx->unodes.ensure(((size_t)baseB + (size_t)ctx->nodeCount) * 32 + 64));
    if (mode == 2) HIPC(ctx->uniEntRec.ensure((size_t)cap * 96));
    char* bb = ctx->uniBuf.as<char>(); char* sc = bb + scOff;
    hipLaunchKernelGGL(k_braid, dim3(1), dim3(TLAS_BUILD_THREADS), 0, st, nodes, ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(), xf, n, cap,
                       (uint2*)(bb + entOff), (float*)(bb + areaOff), mode == 2 ? ctx->uniEntRec.as<float4>() : (float4*)nullptr, (float4*)(bb + bleafOff), (int*)(bb + cntOff), mode == 1 ? 1 : 0, (int*)(bb + ubOff));
    BraidOut bo{(const float4*)(bb + bleafOff), (const int*)(bb + cntOff), (const int*)(bb + ubOff)};
    hipLaunchKernelGGL(k_tlas_build, dim3(1), dim3(TLAS_BUILD_THREADS), 0, st, nodes, ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(), xf, n, ctx->opt.instUnifyRadius,
                       ctx->utlas.as<float4>(), (float4*)sc, (float4*)(sc + leafOff), (uint32_t*)(sc + keyOff), (int*)(sc + prefOff), 1, ctx->dUni, 0, bo);
    hipLaunchKernelGGL(k_unify_top, dim3(1), dim3(TLAS_BUILD_THREADS), 0, st, (const float4*)ctx->utlas.as<float4>(), (const int*)(bb + cntOff), (const uint2*)(bb + entOff), nodes, ctx->descs.as<GpuBlasDesc>(),
                       ctx->instances.as<GpuBlasInstance>(), baseB, ctx->unodes.as<float4>(), mode == 2 ? 1 : 0, stubBase, restoreIdx);
    if (ctx->ichunkCount) hipLaunchKernelGGL(k_unify_blas, dim3(ctx->ichunkCount), dim3(256), 0, st, nodes, ctx->descs.as<GpuBlasDesc>(), (const uint2*)ctx->ichunks.as<uint2>(), baseB, ctx->unodes.as<float4>());
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(st));                       // (waited for: the entries and the top's depth size this walk's stack; a re-derivation — transforms, refits — is rare on the scenes that use it)
    const volatile float* h = ctx->hUni;
    ctx->uniEntries = (int)h[1]; ctx->uniDepth = (int)h[2];
    ctx->uniValid = ctx->uniDepth > 0 && ctx->uniEntries >= n;
    ctx->uniMode = ctx->uniValid ? mode : 0; ctx->uniBaseB = baseB; ctx->uniRestoreIdx = restoreIdx;
    // rows of this walk's stack: what the device derived from the BLASes' RequiredStackSize (validated >= the trees' real need at upload) and the top above them, never more than
    // "the deepest BLAS under the whole top" (a ray that needs more rows is flagged and traced by the exact loop).  TREE 2: an entry parks the node the lane was about to visit and the
    // RESTORE pair under its subtree
    const int extra = mode == 2 ? 3 : 1;
    const int loose = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack) + std::max(1, ctx->uniDepth) + extra;
    ctx->uniCap = std::min(96, std::max(4, std::min(loose, h[3] > 0.0f ? (int)h[3] + extra : loose)));
    return IDKPT_OK;
}

// derives what is stale and decides how the next batch of a several-instance scene without UseTlas is traced: through the library's own tree (*useTlas), by the exact loop with the
// instance sieve (*useSieve), or by k_trace2 MODE 1 (neither).  The instances' overlap is measured on the device into host-mapped memory; the first measurement after an upload is
// waited for, later ones (animated transforms) are read whenever they have arrived — the decision only moves time, never a result
static int inst_tlas_prepare(dev_ctx* ctx, bool* useTlas, bool* useSieve)
{
    *useTlas = false; *useSieve = false;
    const bool wantU = inst_unify_wanted(ctx);
    const bool wantT = inst_tlas_wanted(ctx) || wantU, wantS = inst_tlas_wanted(ctx, true);     // (the unified tree's launches use the own TLAS's top boxes for the primary rays' pre-cull, k_gen_primary)
    if (!wantT && !wantS) return IDKPT_OK;
    { int rc = totals_ensure(ctx); if (rc) return rc; }
    hipStream_t st = ctx->stream;
    const float4* nodes = (const float4*)vb_ptr(ctx, VB_NODES, ctx->vcur[VB_NODES]);
    const int n = ctx->instanceCount;
    if (!ctx->hInstOverlap) { HIPC(hipHostMalloc((void**)&ctx->hInstOverlap, 64, hipHostMallocMapped)); memset(ctx->hInstOverlap, 0, 64); HIPC(hipHostGetDevicePointer((void**)&ctx->dInstOverlap, ctx->hInstOverlap, 0)); }
    if (!ctx->itlasValid) {
        // partial re-braiding (k_braid): the tree's leaves are subtrees of the instances' BLASes, at most `cap` of them
        const bool braid = wantT && ctx->opt.instBraid > 0 && ctx->sceneNested;
        const int cap = braid ? std::max(n, std::min(ctx->opt.instBraid, 8192)) : n, nodeCount = 2 * cap - 1;
        const size_t leafOff = (size_t)nodeCount * 32, keyOff = leafOff + (size_t)cap * 32, prefOff = keyOff + (size_t)cap * 4;
        HIPC(ctx->tlasScratch.ensure(prefOff + (size_t)nodeCount * 4)); HIPC(ctx->itlas.ensure((size_t)nodeCount * 32));
        char* sc = ctx->tlasScratch.as<char>();
        const float4* xf = (const float4*)vb_ptr(ctx, VB_XFORMS, ctx->vcur[VB_XFORMS]);
        BraidOut bo{nullptr, nullptr};
        if (braid) {
            const size_t entOff = 0, areaOff = entOff + (size_t)cap * 8, bleafOff = (areaOff + (size_t)cap * 4 + 15) & ~(size_t)15, cntOff = bleafOff + (size_t)cap * 32;
            HIPC(ctx->braidBuf.ensure(cntOff + 16)); HIPC(ctx->entRec.ensure((size_t)cap * 96));
            char* bb = ctx->braidBuf.as<char>();
            hipLaunchKernelGGL(k_braid, dim3(1), dim3(TLAS_BUILD_THREADS), 0, st, nodes, ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(), xf, n, cap,
                               (uint2*)(bb + entOff), (float*)(bb + areaOff), ctx->entRec.as<float4>(), (float4*)(bb + bleafOff), (int*)(bb + cntOff));
            HIPC(hipGetLastError());
            bo.leaf = (const float4*)(bb + bleafOff); bo.count = (const int*)(bb + cntOff);
        }
#define ITLAS_BUILD(leavesOnly) hipLaunchKernelGGL(k_tlas_build, dim3(1), dim3(TLAS_BUILD_THREADS), 0, st, nodes, ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(), xf, n, 15 /* TLAS.cs: SearchRadius */, \
                                                   ctx->itlas.as<float4>(), (float4*)sc, (float4*)(sc + leafOff), (uint32_t*)(sc + keyOff), (int*)(sc + prefOff), 1, ctx->dInstOverlap, leavesOnly, bo)
        ITLAS_BUILD(1);                                                       // the overlap of the boxes as they are now
        HIPC(hipGetLastError());
        const bool first = !ctx->instOverlapKnown;
        if (first) { HIPC(hipStreamSynchronize(st)); ctx->instOverlapKnown = true; }
        { int rc = inst_unify_prepare(ctx); if (rc) return rc; }
        const float met = *(volatile float*)ctx->hInstOverlap * 100.0f;      // percent of the leaves' boxes a random line meets, x their number
        const float leavesRead = ((volatile float*)ctx->hInstOverlap)[1], leaves = braid ? std::max((float)n, leavesRead) : (float)n;
        const bool worth = ctx->uniValid || (inst_tlas_wanted(ctx) && (ctx->opt.instTlasOverlap >= 100 || met <= (float)ctx->opt.instTlasOverlap * leaves));
        if (worth) { ITLAS_BUILD(0); HIPC(hipGetLastError()); if (first) HIPC(hipStreamSynchronize(st)); }   // (the first build's depth is waited for; later ones are read when they have arrived)
        ctx->itlasBuilt = worth; ctx->itlasBraided = worth && braid; ctx->itlasEntries = (int)leaves;
        ctx->itlasDepth = worth ? (int)((volatile float*)ctx->hInstOverlap)[2] : 0;
        float metInst = met;                                                 // the sieve's question is about whole instances
        if (braid && !worth && wantS) { bo = BraidOut{nullptr, nullptr}; ITLAS_BUILD(1); HIPC(hipGetLastError()); if (first) HIPC(hipStreamSynchronize(st)); metInst = *(volatile float*)ctx->hInstOverlap * 100.0f; }
#undef ITLAS_BUILD
        ctx->isieveWorth = !worth && wantS && (ctx->opt.instSieveOverlap >= 100 || metInst <= (float)ctx->opt.instSieveOverlap * (float)n);
        ctx->itlasNeed = inst_tlas_rows(ctx);                                // (a ray that needs more rows is traced by the exact loop)
        ctx->itlasValid = true;
    }
    const bool tree = ctx->itlasBuilt && (ctx->uniValid ? wantU : inst_tlas_wanted(ctx)), sieve = !tree && wantS && (ctx->isieveWorth || ctx->itlasBuilt);   // (the options that change what is wanted invalidate the decision: host_options.hpp)
    if (!tree && !sieve) return IDKPT_OK;
    { int rc = inst_records_prepare(ctx); if (rc) return rc; }
    if (!tree) { *useSieve = true; return IDKPT_OK; }
    { int rc = marks_prepare(ctx); if (rc) return rc; }
    *useTlas = true;
    return IDKPT_OK;
}

// One traversal launch over `list` (cnt entries, on the device): the instantiation of k_trace2 — or k_trace2s / k_trace_wide — that serves this scene, these settings and this launch.
// The shipped instantiations (everything else the template can express is unreachable from here):
//   k_trace2<P, C, 32, 1, false, 24, 0, 0 | 16, V>   one BLAS instance (MODE 0), plain or pooled leaf phase       P: primary / bounce launch, C: counting build, V: scene versions
//   k_trace2<P, C, 16, 1, false, 24, 1 | 2, 0, V>    instance loop / TLAS walk inside the kernel (MODE 1 / 2)
//   k_trace2<true, false, 32, 1, false, 24, M, 0, false, true>   any-hit queries (idkptTraceRays with IDKPT_TRACE_ANY_HIT), M = 0 / 1 / 2
//   k_trace2s<P>                                      small launches of sparse views: long rays split across idle lanes (kernels_trace_split.hpp)
//   k_trace_wide<P, C'> + k_trace2<P, false>          option "wide": the wide-node walk and the exact re-trace of the rays it does not vouch for (kernels_wide.hpp)
//   k_trace_packet + k_trace2<true, false>            option "packet" (default: by measurement): primary launches as wave-uniform packets + the exact re-trace of the rays it does not vouch for (kernels_packet.hpp)
//   k_trace_inst<P> + k_trace2<P, false, 16, .., 1>   option "inst_tlas" (default: from 8 instances on): the instance loop through the library's own TLAS + the exact loop for flagged rays
// Developer builds (-DIDKPT_DEVELOPER, option "trace_variant") add the s_memtime-instrumented and the scheduling-probe instantiations.
template <bool PRIMARY>
static void launch_trace2(dev_ctx* ctx, uint32_t grid, size_t lds, hipStream_t st, const DScene& s, const Frame& f, const RayBufs& rays, const TraceBufs& tr, const HitBufs& hits,
                          const uint32_t* list, const uint32_t* cnt, uint32_t* work, uint64_t* counters, bool split = false, int bounce = 0, bool anyHit = false)
{
    const bool stock = ctx->opt.traceVariant == 0 || ctx->opt.traceVariant == 100;
    if (anyHit) {   // idkptTraceRays with IDKPT_TRACE_ANY_HIT (kernels_query.hpp): TraceRayAny's walk on the same scheduler
#define T2A(M) hipLaunchKernelGGL((k_trace2<true, false, 32, 1, false, 24, M, 0, false, true>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
        if (f.useTlas) T2A(2); else if (ctx->instanceCount > 1) T2A(1); else T2A(0);
#undef T2A
        return;
    }
    if (PRIMARY && bounce == 0 && f.packet && f.instTlas && ctx->uniValid && ctx->uniMode == 1 && ctx->itlasValid && ctx->imarksValid && ctx->hPkStats && s.instRec && !s.ver && !f.useTlas && !f.queryMode && !f.hitsByRid && !ctx->counters) {
        // the packet walk over the unified tree of a same-space multi-instance scene, then — on the launch's own list of the rays it does not vouch for — the exact loop (sieved)
        PacketBufs pb;
        pb.marks = (const uint8_t*)ctx->imarks.as<uint8_t>(); pb.flagCount = work + 128; pb.flagA = ctx->sortKeys.as<uint32_t>(); pb.totals = ctx->wtotals.as<unsigned long long>() + 8;
        pb.unodes = (const float4*)ctx->unodes.as<float4>(); pb.uniXformId = (uint32_t)ctx->hInstances[0].MeshTransformId; pb.blasCount = (int)ctx->hDescs.size();
        pb.blasTriStart = (const uint32_t*)ctx->uTabs.as<uint32_t>(); pb.blasXform = pb.blasTriStart + pb.blasCount;
        const uint32_t pw = (uint32_t)std::min(32, std::max(1, ctx->opt.packetWaves > 0 ? ctx->opt.packetWaves : 28));
        const uint32_t gp = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)ctx->numCUs * pw, (uint32_t)(((size_t)f.batch * ctx->W * ctx->rows + 63) / 64)));
        ctx->uniLaunches++;
        hipLaunchKernelGGL((k_trace_packet<true, true>), dim3(gp), dim3(WAVE), 0, st, s, f, rays, tr, hits, list, cnt, work, pb);
        hipLaunchKernelGGL(k_packet_mirror, dim3(1), dim3(64), 0, st, (const unsigned long long*)pb.totals, ctx->dPkStats);
        InstTlasBufs ib; memset(&ib, 0, sizeof(ib)); ib.maskWords = (ctx->instanceCount + 31) / 32; ib.totals = ctx->wtotals.as<unsigned long long>() + 4;
        TraceBufs trf = tr; trf.order = nullptr; trf.orderIdx = nullptr;
        Frame ff = f; ff.gridRaysX4 = 6u; ff.gridMid = 0u;
        const uint32_t g2 = std::min<uint32_t>(grid, 2048u);
        if (ib.maskWords <= inst_tlas_rows(ctx)) hipLaunchKernelGGL((k_trace_inst<true, true>), dim3(g2), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)pb.flagA, (const uint32_t*)pb.flagCount, work + 64, ib);
        else hipLaunchKernelGGL((k_trace2<true, false, 16, 1, false, 24, 1, 0, false>), dim3(g2), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)pb.flagA, (const uint32_t*)pb.flagCount, work + 64, counters);
        return;
    }
    if (PRIMARY && bounce == 0 && f.packet && ctx->imarksValid && ctx->hPkStats && !s.ver && !f.useTlas && !f.queryMode && ctx->instanceCount == 1) {
        // packet walk (kernels_packet.hpp), then — on the launch's own list of the rays it does not vouch for — the exact kernel
        PacketBufs pb;
        pb.marks = (const uint8_t*)ctx->imarks.as<uint8_t>(); pb.flagCount = work + 128; pb.flagA = ctx->sortKeys.as<uint32_t>(); pb.totals = ctx->wtotals.as<unsigned long long>() + 8;
        const uint32_t pw = (uint32_t)std::min(32, std::max(1, ctx->opt.packetWaves > 0 ? ctx->opt.packetWaves : 28));
        const uint32_t gp = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)ctx->numCUs * pw, (uint32_t)(((size_t)f.batch * ctx->W * ctx->rows + 63) / 64)));
        hipLaunchKernelGGL((k_trace_packet<true>), dim3(gp), dim3(WAVE), 0, st, s, f, rays, tr, hits, list, cnt, work, pb);
        hipLaunchKernelGGL(k_packet_mirror, dim3(1), dim3(64), 0, st, (const unsigned long long*)pb.totals, ctx->dPkStats);
        TraceBufs trf = tr; trf.order = nullptr; trf.orderIdx = nullptr;
        Frame ff = f; ff.gridRaysX4 = 6u; ff.gridMid = 0u;                       // (the device sizes the launch from its actual count: k_trace2's own rule)
        hipLaunchKernelGGL((k_trace2<true, false>), dim3(std::min<uint32_t>(grid, 2048u)), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)pb.flagA, (const uint32_t*)pb.flagCount, work + 64, counters);
        return;
    }
    if (ctx->wideFillValid && ctx->wideTopoValid && wide_wanted(ctx) && !s.ver && !f.useTlas && !f.queryMode && !f.hitsByRid) {
        // wide-node walk (kernels_wide.hpp), then — on the launch's own list of the rays it does not vouch for, almost always empty — the exact kernel
        const int b0 = ctx->hInst0Blas;
        WideBufs wb;
        wb.nodes = (const uint4*)(ctx->wnodes.as<uint4>() + 4 * (size_t)ctx->wNodeOff[b0]); wb.leaves = (const float4*)(ctx->wleaf.as<float4>() + ctx->wLeafOff[b0]);
        wb.flagCount = work + 128; wb.flagA = ctx->sortKeys.as<uint32_t>(); wb.flagB = ctx->sortVals.as<uint32_t>(); wb.totals = ctx->wtotals.as<unsigned long long>(); wb.cap = wide_stack_rows(ctx);
        const size_t ldsW = (size_t)(wb.cap + 2) * WAVE * 4 + (size_t)std::max(0, ctx->opt.ldsPad);     // + the dummy and the spare row
#ifdef IDKPT_DEVELOPER
        if (ctx->opt.traceVariant == 213) hipLaunchKernelGGL((k_trace_wide<PRIMARY, false, 32, true>), dim3(grid), dim3(WAVE), ldsW, st, s, f, rays, tr, hits, list, cnt, work, wb, counters);   // s_memtime-instrumented
        else
#endif
        if (ctx->opt.wideCount) hipLaunchKernelGGL((k_trace_wide<PRIMARY, true>), dim3(grid), dim3(WAVE), ldsW, st, s, f, rays, tr, hits, list, cnt, work, wb, counters);
        else hipLaunchKernelGGL((k_trace_wide<PRIMARY, false>), dim3(grid), dim3(WAVE), ldsW, st, s, f, rays, tr, hits, list, cnt, work, wb, counters);
        TraceBufs trf = tr; trf.order = nullptr; trf.orderIdx = nullptr;
        if (!PRIMARY) { trf.order = wb.flagA; trf.orderIdx = wb.flagB; }        // position -> queue slot and ray id (the hit is stored at the slot, as always)
        Frame ff = f; ff.gridRaysX4 = 6u; ff.gridMid = 0u;                       // (the device sizes the launch from its actual count: k_trace2's own rule)
        hipLaunchKernelGGL((k_trace2<PRIMARY, false>), dim3(std::min<uint32_t>(grid, 2048u)), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)wb.flagA, (const uint32_t*)wb.flagCount, work + 64, counters);
        return;
    }
    if (f.instTlas && ctx->itlasValid && ctx->itlasBuilt && ctx->imarksValid && s.instRec && !s.ver && !f.useTlas && !f.queryMode && !f.hitsByRid && !ctx->counters) {
        // the instance loop through the library's own TLAS (kernels_trace_inst.hpp), then — on the launch's own list of flagged rays — the exact loop
        InstTlasBufs ib;
        ib.tlas = (const float4*)ctx->itlas.as<float4>(); ib.marks = (const uint8_t*)ctx->imarks.as<uint8_t>(); ib.tlasCap = ctx->itlasNeed;
        ib.entRec = ctx->itlasBraided ? (const float4*)ctx->entRec.as<float4>() : s.instRec;
        ib.flagCount = work + 128; ib.flagA = ctx->sortKeys.as<uint32_t>(); ib.flagB = ctx->sortVals.as<uint32_t>(); ib.totals = ctx->wtotals.as<unsigned long long>() + 4;
        ib.maskWords = (ctx->instanceCount + 31) / 32;
        if (ctx->uniValid) {
            // every instance in one BLAS space: the unified tree (no TLAS phase, no TLAS rows: its stack is the top's depth on the deepest BLAS's)
            ib.unodes = (const float4*)ctx->unodes.as<float4>(); ib.uniCap = ctx->uniCap; ib.blasCount = (int)ctx->hDescs.size();
            ib.blasTriStart = (const uint32_t*)ctx->uTabs.as<uint32_t>(); ib.blasXform = ib.blasTriStart + ib.blasCount;
            ib.uniXformId = (uint32_t)ctx->hInstances[0].MeshTransformId;
            ctx->uniLaunches++;
            ib.baseB = ctx->uniBaseB; ib.restoreIdx = ctx->uniRestoreIdx; if (ctx->uniMode == 2) ib.entRec = (const float4*)ctx->uniEntRec.as<float4>();
            const size_t ldsU = (size_t)(ctx->uniCap + 2) * WAVE * 4 + (size_t)std::max(0, ctx->opt.ldsPad);
            if (ctx->uniMode == 2) hipLaunchKernelGGL((k_trace_inst<PRIMARY, false, 16, 2>), dim3(grid), dim3(WAVE), ldsU, st, s, f, rays, tr, hits, list, cnt, work, ib);
            else if (ctx->opt.uniRefill == 32) hipLaunchKernelGGL((k_trace_inst<PRIMARY, false, 32, 1>), dim3(grid), dim3(WAVE), ldsU, st, s, f, rays, tr, hits, list, cnt, work, ib);
            else hipLaunchKernelGGL((k_trace_inst<PRIMARY, false, 16, 1>), dim3(grid), dim3(WAVE), ldsU, st, s, f, rays, tr, hits, list, cnt, work, ib);
        } else
        hipLaunchKernelGGL((k_trace_inst<PRIMARY>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, ib);
        TraceBufs trf = tr; trf.order = nullptr; trf.orderIdx = nullptr;
        if (!PRIMARY) { trf.order = ib.flagA; trf.orderIdx = ib.flagB; }
        Frame ff = f; ff.gridRaysX4 = 6u; ff.gridMid = 0u;
        const uint32_t g2 = std::min<uint32_t>(grid, 2048u);
        // the flagged rays: the exact loop, with the instances a ray cannot meet sieved out up front (k_trace_inst<P, true>) while a lane's mask fits the rows LDS has for it
        if (ib.maskWords <= inst_tlas_rows(ctx)) hipLaunchKernelGGL((k_trace_inst<PRIMARY, true>), dim3(g2), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)ib.flagA, (const uint32_t*)ib.flagCount, work + 64, ib);
        else hipLaunchKernelGGL((k_trace2<PRIMARY, false, 16, 1, false, 24, 1, 0, false>), dim3(g2), dim3(WAVE), lds, st, s, ff, rays, trf, hits, (const uint32_t*)ib.flagA, (const uint32_t*)ib.flagCount, work + 64, counters);
        return;
    }
    if (f.instSieve && s.instRec && !s.ver && !f.useTlas && !f.hitsByRid && !ctx->counters) {   // the exact loop with the per-ray instance sieve (kernels_trace_inst.hpp, EXACT)
        InstTlasBufs ib; memset(&ib, 0, sizeof(ib)); ib.maskWords = (ctx->instanceCount + 31) / 32;
        hipLaunchKernelGGL((k_trace_inst<PRIMARY, true>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, ib);
        return;
    }
    if (split && !s.ver && !f.useTlas && ctx->instanceCount == 1 && !ctx->counters && ctx->sceneNested && stock) {   // small launch: long rays are split across idle lanes (kernels_trace_split.hpp)
        hipLaunchKernelGGL((k_trace2s<PRIMARY>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work);
        return;
    }
    // pooled leaf phase (kernels_trace.hpp, DBG 16): per launch kind — option leaf_pool, a mask: 1 = primary launches, 2 = the first bounce, 4 = later bounces; -1 = automatic
    int poolMask = ctx->opt.leafPool;
    if (poolMask < 0) {   // automatic: by view class, known from the previous batch of the same shape (unknown: the dense-view choice)
        const uint64_t pixels = (uint64_t)ctx->W * ctx->rows * (uint64_t)std::max(1, f.batch);
        const bool sparse = ctx->lastFast && ctx->lastBatch == f.batch && (uint64_t)ctx->hCounts[MAX_DEPTH_SLOTS - 1] * 2u < pixels;
        poolMask = sparse ? 1 : 3;
    }
    const bool pool = ((poolMask >> std::min(bounce, 2)) & 1) && stock;
#define T2X(C, M, D, V) hipLaunchKernelGGL((k_trace2<PRIMARY, C, 32, 1, false, 24, M, D, V>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
#define T2P(C, V) do { if (pool) T2X(C, 0, 16, V); else T2X(C, 0, 0, V); } while (0)
// the instance-loop / TLAS kernels: never pooled (they gain nothing from it), and idle lanes are refilled from 16 on instead of 32 — their rays live two to three BLAS walks,
// so a refill is rarer per step than in MODE 0 and lanes are what these modes lack (profiles/r04_multi_blas.md; MODE 0 keeps 32: 24 measured -4.5 % there in round 1)
#define T2M(C, M, V) hipLaunchKernelGGL((k_trace2<PRIMARY, C, 16, 1, false, 24, M, 0, V>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
#define T2N(M, V) do { if (ctx->counters) T2M(true, M, V); else T2M(false, M, V); } while (0)
#ifdef IDKPT_DEVELOPER
    if (!s.ver && !ctx->counters && !stock) {
        const int v = ctx->opt.traceVariant;
        if ((f.useTlas || ctx->instanceCount > 1) && (v == 24 || v == 48 || v == 16 || v == 8 || v == 12)) {   // the refill threshold of the instance-loop / TLAS kernels
#define T2R(R, M) hipLaunchKernelGGL((k_trace2<PRIMARY, false, R, 1, false, 24, M, 0, false>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
            const int m = f.useTlas ? 2 : 1;
            if (v == 24) { if (m == 2) T2R(24, 2); else T2R(24, 1); } else if (v == 48) { if (m == 2) T2R(48, 2); else T2R(48, 1); } else if (v == 16) { if (m == 2) T2R(16, 2); else T2R(16, 1); }
            else if (v == 12) { if (m == 2) T2R(12, 2); else T2R(12, 1); } else { if (m == 2) T2R(8, 2); else T2R(8, 1); }
#undef T2R
            return;
        }
        if (!f.useTlas && ctx->instanceCount == 1) switch (v) {   // s_memtime-instrumented and scheduling-probe instantiations of MODE 0; results are bit-identical
            case 107: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 16, 1, true, 65>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;   // instrumented, round 1's policy
            case 113: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, true, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;   // instrumented, default policy
            case 116: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, true, 24, 0, 16>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;   // instrumented, pooled leaf phase
#define T2V(R, L) hipLaunchKernelGGL((k_trace2<PRIMARY, false, R, 1, false, L>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters)
            case 901: T2V(32, 20); return; case 902: T2V(40, 16); return; case 903: T2V(24, 24); return; case 904: T2V(16, 24); return;   // refill threshold / parked-leaf threshold probes
#undef T2V
#define T2Q(R) do { if (PRIMARY) hipLaunchKernelGGL((k_trace2<PRIMARY, false, R, 1, false, 24, 0, 16>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); else hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); } while (0)
            case 911: T2Q(16); return; case 912: T2Q(24); return; case 913: T2Q(48); return; case 914: T2Q(8); return;   // refill threshold of the PRIMARY launch (pooled leaf phase) under the pixel-major list; bounces as shipped
#undef T2Q
            case 961: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 7, false, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;   // occupancy probes: 7 / 8 waves per SIMD forced (launch bounds)
            case 962: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 8, false, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return;
            default: break;
        }
    }
#endif
    if (s.ver) {                    // scene versions: the samples of this batch see different states of the geometry (VER instantiations, kernels_trace.hpp)
        if (f.useTlas) T2N(2, true); else if (ctx->instanceCount > 1) T2N(1, true); else if (ctx->counters) T2P(true, true); else T2P(false, true);
        return;
    }
    if (f.useTlas) { T2N(2, false); return; }                   // TLAS walk inside the kernel
    if (ctx->instanceCount > 1) { T2N(1, false); return; }      // instance loop inside the kernel
    if (ctx->counters) T2P(true, false);
    else if (s.pairNodes && ctx->pairValid && stock) {   // the FAST node step on the regrouped pairs (one stack row more: kernels_trace.hpp)
        const size_t ldsF = lds + (size_t)WAVE * 4;
        if (pool) hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 0, 16, false, false, true>), dim3(grid), dim3(WAVE), ldsF, st, s, f, rays, tr, hits, list, cnt, work, counters);
        else hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 0, 0, false, false, true>), dim3(grid), dim3(WAVE), ldsF, st, s, f, rays, tr, hits, list, cnt, work, counters);
    }
    else T2P(false, false);
#undef T2N
#undef T2M
#undef T2P
#undef T2X
}
