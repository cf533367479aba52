// host_builder.hpp — idkptBuildBlasCore / idkptBuildBlas: the host side of the BLAS builder on the device (kernels: bvh_gpu.hpp, bvh_gpu_full.hpp).
// Part of the single translation unit idkpt.hip (included there, in this order).
#pragma once

// ---- the BLAS builder on the device: idkptBuildBlasCore (bvh_gpu.hpp) and idkptBuildBlas (bvh_gpu_full.hpp)
struct BuilderScratch {
    // core (per fragment / per node / per chunk)
    DevBuf fb, ids[3][2], keys[2], vals[2], hist, nodes, act[2], cnt, nodeChunk0, chunkNode, chunkBegin, cboxL, cboxR, carryL, carryR, rc, cbestCost, cbestPos, dec, sideL, sideR, freshOf,
           swapOf, leftCountOf, startOf, countOf, leftTable, pcnt, poff, smallList, aux;
    // whole build
    DevBuf pos, tris, prio, splitCnt, first, scanTmp[3], gboxPart, gbox, origTri, parent, jump[2], dist[2], arrived, need, aggStart, aggCount, bins, maxDepth, used, rank, outNodes, leafCnt, at, uniq, ucount,
           outTris, parents, leafFlag, leaves, sahPart, status;
    // results of the last idkptBuildBlas, for idkptBuildBlasFetch
    int outNodeCount = 0, outTriCount = 0, outParentCount = 0, outLeafCount = 0; bool haveResult = false;
    void release()
    {
        DevBuf* all[] = {&fb, &ids[0][0], &ids[0][1], &ids[1][0], &ids[1][1], &ids[2][0], &ids[2][1], &keys[0], &keys[1], &vals[0], &vals[1], &hist, &nodes, &act[0], &act[1], &cnt, &nodeChunk0, &chunkNode, &chunkBegin,
                         &cboxL, &cboxR, &carryL, &carryR, &rc, &cbestCost, &cbestPos, &dec, &sideL, &sideR, &freshOf, &swapOf, &leftCountOf, &startOf, &countOf, &leftTable, &pcnt, &poff, &smallList, &aux,
                         &pos, &tris, &prio, &splitCnt, &first, &scanTmp[0], &scanTmp[1], &scanTmp[2], &gboxPart, &gbox, &origTri, &parent, &jump[0], &jump[1], &dist[0], &dist[1], &arrived, &need, &aggStart, &aggCount,
                         &bins, &maxDepth, &used, &rank, &outNodes, &leafCnt, &at, &uniq, &ucount, &outTris, &parents, &leafFlag, &leaves, &sahPart, &status};
        for (DevBuf* b : all) b->release();
    }
};
static void builder_scratch_free(dev_ctx* ctx) { if (ctx->bscratch) { ctx->bscratch->release(); delete ctx->bscratch; ctx->bscratch = nullptr; } }
static BuilderScratch& builder_scratch(dev_ctx* ctx) { if (!ctx->bscratch) ctx->bscratch = new BuilderScratch(); return *ctx->bscratch; }

// BLAS.GetBuildData + the SweepSAH recursion over the n fragment boxes in B.fb (device): leaves the node array (2 * n entries, reference id scheme,
// not compacted) in B.nodes and the final x-sorted id order in B.ids[0][*outPP].  Issues on the context's stream; one host sync per level.
static int bvh_core(dev_ctx* ctx, BuilderScratch& B, int n, int* outPP, int* outLevels)
{
    using namespace bvhgpu;
    hipStream_t st = ctx->stream;
    const bool timing = ctx->opt.bvhTiming != 0;   // option "bvh_timing": host-side phase times on stderr
    auto tq = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; (void)hipStreamSynchronize(st); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[idkpt bvh] %-12s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tq).count()); tq = t; };
    const size_t nodeCount = (size_t)std::max(2 * n, 4);
    const int Cmax = n / CH + n + 2;                             // chunks of a level: at most one per CH positions plus one per active node
    for (int a = 0; a < 3; a++) for (int k = 0; k < 2; k++) HIPC(B.ids[a][k].ensure((size_t)n * 4));
    for (int k = 0; k < 2; k++) { HIPC(B.keys[k].ensure((size_t)n * 4)); HIPC(B.vals[k].ensure((size_t)n * 4)); HIPC(B.act[k].ensure(nodeCount * 4)); }
    const uint32_t nTiles = ((uint32_t)n + SORT_TILE - 1) / SORT_TILE;
    HIPC(B.hist.ensure(((size_t)SORT_RADIX * nTiles + SORT_RADIX) * 4));
    HIPC(B.nodes.ensure(nodeCount * 32)); HIPC(B.cnt.ensure(64));
    HIPC(B.nodeChunk0.ensure((nodeCount + 1) * 4)); HIPC(B.chunkNode.ensure((size_t)Cmax * 4)); HIPC(B.chunkBegin.ensure((size_t)Cmax * 4));
    HIPC(B.cboxL.ensure((size_t)3 * Cmax * sizeof(BBox))); HIPC(B.cboxR.ensure((size_t)3 * Cmax * sizeof(BBox))); HIPC(B.carryL.ensure((size_t)3 * Cmax * sizeof(BBox))); HIPC(B.carryR.ensure((size_t)3 * Cmax * sizeof(BBox)));
    HIPC(B.rc.ensure((size_t)3 * n * 4)); HIPC(B.cbestCost.ensure((size_t)3 * Cmax * 4)); HIPC(B.cbestPos.ensure((size_t)3 * Cmax * 4));
    HIPC(B.dec.ensure(nodeCount * sizeof(Decision))); HIPC(B.sideL.ensure((size_t)Cmax * sizeof(BBox))); HIPC(B.sideR.ensure((size_t)Cmax * sizeof(BBox)));
    HIPC(B.freshOf.ensure(nodeCount * 4)); HIPC(B.swapOf.ensure(nodeCount * 4)); HIPC(B.leftCountOf.ensure(nodeCount * 4)); HIPC(B.startOf.ensure(nodeCount * 4)); HIPC(B.countOf.ensure(nodeCount * 4));
    HIPC(B.leftTable.ensure((size_t)n)); HIPC(B.pcnt.ensure((size_t)3 * Cmax * 4)); HIPC(B.poff.ensure((size_t)3 * Cmax * 4));
    HIPC(B.smallList.ensure(nodeCount * 4)); HIPC(B.aux.ensure((size_t)n * 4));
    const int smallMax = std::max(0, std::min(128, ctx->opt.bvhSmall));   // subtrees of at most this many fragments are finished by one thread each (option "bvh_small")
    lap("alloc");
    // device words: [0] = n (count for the sort kernels), [1] = chunk count of the level, [2] = next level's node count, [3] = small subtrees
    uint32_t* cnt = B.cnt.as<uint32_t>();
    { uint32_t h[4] = {(uint32_t)n, 0u, 0u, 0u}; HIPC(hipMemcpyAsync(cnt, h, 16, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }
    // ---- BLAS.GetBuildData: per axis a stable sort of the ids by FloatToKey(min + max) (five 7-bit LSD passes over the 32-bit key)
    uint32_t* digitTotals = B.hist.as<uint32_t>() + (size_t)SORT_RADIX * nTiles;
    for (int axis = 0; axis < 3; axis++) {
        hipLaunchKernelGGL(k_keys, dim3((n + 255) / 256), dim3(256), 0, st, (const float4*)B.fb.as<float4>(), n, axis, B.keys[0].as<uint32_t>(), B.vals[0].as<uint32_t>());
        int cur = 0;
        for (int pass = 0; pass < 5; pass++) {
            hipLaunchKernelGGL(k_sort_hist, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)B.keys[cur].as<uint32_t>(), (const uint32_t*)cnt, (uint32_t)(7 * pass), B.hist.as<uint32_t>(), nTiles);
            hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(1024), 0, st, (const uint32_t*)cnt, B.hist.as<uint32_t>(), nTiles, digitTotals);
            hipLaunchKernelGGL(k_sort_scatter, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)B.keys[cur].as<uint32_t>(), (const uint32_t*)B.vals[cur].as<uint32_t>(), (const uint32_t*)cnt, (uint32_t)(7 * pass),
                               (const uint32_t*)B.hist.as<uint32_t>(), nTiles, (const uint32_t*)digitTotals, B.keys[1 - cur].as<uint32_t>(), B.vals[1 - cur].as<uint32_t>());
            cur = 1 - cur;
        }
        HIPC(hipMemcpyAsync(B.ids[axis][0].p, B.vals[cur].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    }
    lap("sort");
    // ---- the recursion, one level at a time
    HIPC(hipMemsetAsync(B.nodes.p, 0, nodeCount * 32, st));
    { HNodeG root = {}; root.startOrChild = 0; root.count = n; HIPC(hipMemcpyAsync(B.nodes.as<HNodeG>() + 1, &root, 32, hipMemcpyHostToDevice, st));
      int one = 1, two = 2; HIPC(hipMemcpyAsync(B.act[0].p, &one, 4, hipMemcpyHostToDevice, st)); HIPC(hipMemcpyAsync(B.freshOf.as<int>() + 1, &two, 4, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }
    HNodeG* nodes = B.nodes.as<HNodeG>();
    int A = 1, curAct = 0, pp = 0, levels = 0;
    if (n <= smallMax) { A = 0; int one = 1; HIPC(hipMemcpyAsync(B.smallList.p, &one, 4, hipMemcpyHostToDevice, st)); HIPC(hipMemcpyAsync(cnt + 3, &one, 4, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st)); }
    while (A > 0) {
        Level L; L.act = B.act[curAct].as<int>(); L.A = A; L.nodeChunk0 = B.nodeChunk0.as<int>(); L.chunkNode = B.chunkNode.as<int>(); L.chunkBegin = B.chunkBegin.as<int>(); L.chunkCount = (int*)(cnt + 1);
        const int Cl = std::min(Cmax, n / CH + A + 1);            // grid bound for this level's chunks (workgroups beyond the real count exit)
        const int gA = (A + 255) / 256;
        const int* i0 = B.ids[0][pp].as<int>(); const int* i1 = B.ids[1][pp].as<int>(); const int* i2 = B.ids[2][pp].as<int>();
        int* o0 = B.ids[0][1 - pp].as<int>(); int* o1 = B.ids[1][1 - pp].as<int>(); int* o2 = B.ids[2][1 - pp].as<int>();
        const float4* fb = B.fb.as<float4>();
        HIPC(hipMemsetAsync(cnt + 2, 0, 4, st));
        hipLaunchKernelGGL(k_chunks, dim3(1), dim3(1024), 0, st, (const HNodeG*)nodes, L);
        hipLaunchKernelGGL(k_snapshot_ranges, dim3(gA), dim3(256), 0, st, (const HNodeG*)nodes, L, B.startOf.as<int>(), B.countOf.as<int>());
        hipLaunchKernelGGL(k_chunk_box, dim3(Cl, 3), dim3(CH), 0, st, (const HNodeG*)nodes, L, fb, i0, i1, i2, B.cboxL.as<BBox>(), B.cboxR.as<BBox>(), Cmax);
        hipLaunchKernelGGL(k_node_carry, dim3(A), dim3(CH), 0, st, nodes, L, (const BBox*)B.cboxL.as<BBox>(), (const BBox*)B.cboxR.as<BBox>(), B.carryL.as<BBox>(), B.carryR.as<BBox>(), Cmax);
        hipLaunchKernelGGL(k_chunk_rc, dim3(Cl, 3), dim3(CH), 0, st, (const HNodeG*)nodes, L, fb, i0, i1, i2, (const BBox*)B.carryR.as<BBox>(), B.rc.as<float>(), n, Cmax);
        hipLaunchKernelGGL(k_chunk_cost, dim3(Cl, 3), dim3(CH), 0, st, (const HNodeG*)nodes, L, fb, i0, i1, i2, (const BBox*)B.carryL.as<BBox>(), (const float*)B.rc.as<float>(), B.cbestCost.as<float>(), B.cbestPos.as<int>(), n, Cmax);
        hipLaunchKernelGGL(k_node_decide, dim3(A), dim3(CH), 0, st, (const HNodeG*)nodes, L, (const float*)B.cbestCost.as<float>(), (const int*)B.cbestPos.as<int>(), B.dec.as<Decision>(), Cmax);
        hipLaunchKernelGGL(k_chunk_sides, dim3(Cl), dim3(CH), 0, st, (const HNodeG*)nodes, L, (const Decision*)B.dec.as<Decision>(), fb, i0, i1, i2, B.sideL.as<BBox>(), B.sideR.as<BBox>());
        hipLaunchKernelGGL(k_node_finalize, dim3(A), dim3(CH), 0, st, nodes, L, B.dec.as<Decision>(), (const BBox*)B.sideL.as<BBox>(), (const BBox*)B.sideR.as<BBox>(), B.freshOf.as<int>(), B.swapOf.as<int>(), B.leftCountOf.as<int>(),
                           B.act[1 - curAct].as<int>(), (int*)(cnt + 2), B.smallList.as<int>(), (int*)(cnt + 3), smallMax);
        hipLaunchKernelGGL(k_mark, dim3(Cl), dim3(CH), 0, st, (const HNodeG*)nodes, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.swapOf.as<int>(), (const int*)B.startOf.as<int>(), (const int*)B.countOf.as<int>(), i0, i1, i2, B.leftTable.as<uint8_t>());
        for (int a = 0; a < 3; a++) HIPC(hipMemcpyAsync(B.ids[a][1 - pp].p, B.ids[a][pp].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_part_count, dim3(Cl, 3), dim3(CH), 0, st, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.startOf.as<int>(), (const int*)B.countOf.as<int>(), i0, i1, i2, (const uint8_t*)B.leftTable.as<uint8_t>(), B.pcnt.as<int>(), Cmax);
        hipLaunchKernelGGL(k_part_offsets, dim3(A), dim3(CH), 0, st, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.pcnt.as<int>(), B.poff.as<int>(), Cmax);
        hipLaunchKernelGGL(k_part_scatter, dim3(Cl, 3), dim3(CH), 0, st, L, (const Decision*)B.dec.as<Decision>(), (const int*)B.startOf.as<int>(), (const int*)B.countOf.as<int>(), (const int*)B.leftCountOf.as<int>(), i0, i1, i2, o0, o1, o2,
                           (const uint8_t*)B.leftTable.as<uint8_t>(), (const int*)B.poff.as<int>(), Cmax);
        HIPC(hipGetLastError());
        uint32_t next = 0;
        HIPC(hipMemcpyAsync(&next, cnt + 2, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        A = (int)next; curAct = 1 - curAct; pp = 1 - pp; levels++;
        if (levels > 4096) return fail(ctx, IDKPT_ERR_UNKNOWN, "idkptBuildBlasCore: recursion does not terminate");
    }
    lap("levels");
    hipLaunchKernelGGL(k_small_subtrees, dim3((unsigned)((nodeCount + 63) / 64)), dim3(64), 0, st, nodes, (const int*)B.smallList.as<int>(), (const int*)(cnt + 3), (const float4*)B.fb.as<float4>(), B.ids[0][pp].as<int>(), B.ids[1][pp].as<int>(), B.ids[2][pp].as<int>(),
                       B.rc.as<float>(), B.aux.as<int>(), B.leftTable.as<uint8_t>(), (const int*)B.freshOf.as<int>());
    HIPC(hipGetLastError());
    lap("subtrees");
    *outPP = pp; if (outLevels) *outLevels = levels;
    return IDKPT_OK;
}

// idkptBuildBlasCore: fragment boxes in (host), node array + final x-sorted id order out (host).  Stateless apart from the device, the stream and
// the cached scratch buffers of the context: no scene is needed and none is touched.
static int32_t dev_BuildBlasCore(dev_ctx* ctx, const float* fragBoxes, int32_t n, GpuBlasNode* outNodes, int32_t* outSortedX, int32_t* outLevels)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(fragBoxes && outNodes && outSortedX && n >= 1, "idkptBuildBlasCore: null argument or no fragments");
    REQUIRE(n <= (1 << 27), "idkptBuildBlasCore: too many fragments");
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = builder_scratch(ctx);
    B.haveResult = false;
    hipStream_t st = ctx->stream;
    HIPC(B.fb.ensure((size_t)n * 32));
    HIPC(hipMemcpyAsync(B.fb.p, fragBoxes, (size_t)n * 32, hipMemcpyHostToDevice, st));
    int pp = 0;
    int rc = bvh_core(ctx, B, n, &pp, outLevels); if (rc) return rc;
    HIPC(hipMemcpyAsync(outNodes, B.nodes.p, (size_t)std::max(2 * n, 4) * 32, hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(outSortedX, B.ids[0][pp].p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return IDKPT_OK;
}

// exclusive scan of n uint32 on the device (bvh_gpu_full.hpp k_scan_*); in and out may alias; tmp[level] hold the block totals
static int scan_u32(dev_ctx* ctx, BuilderScratch& B, const uint32_t* in, uint32_t* out, uint32_t n, int level = 0)
{
    using namespace bvhgpu;
    if (n == 0) return IDKPT_OK;
    if (level > 2) return fail(ctx, IDKPT_ERR_UNKNOWN, "scan_u32: too many elements");
    const uint32_t per = SCAN_BLOCK * SCAN_ITEMS, blocks = (n + per - 1) / per;
    HIPC(B.scanTmp[level].ensure((size_t)blocks * 4 + 16));
    uint32_t* tot = B.scanTmp[level].as<uint32_t>();
    hipLaunchKernelGGL(k_scan_block, dim3(blocks), dim3(SCAN_BLOCK), 0, ctx->stream, in, out, n, tot);
    if (blocks > 1) {
        int rc = scan_u32(ctx, B, tot, tot, blocks, level + 1); if (rc) return rc;
        hipLaunchKernelGGL(k_scan_add, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, out, n, (const uint32_t*)tot);
    }
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

// OptimizeStackSize in the reference's own order (BLAS.cs:875-936) on a host copy of the uncompacted tree: the fallback of dev_BuildBlas for a
// decision its error bound cannot settle (and for trees deeper than its per-depth table).  Serial, recursive, binary64 like the reference.
namespace stackopt_host {
struct HN { float mn[3]; int32_t startOrChild; float mx[3]; int32_t count; };
static inline bool leaf(const HN& n) { return n.count > 0; }
static inline float half_area(const HN& n) { float x = n.mx[0] - n.mn[0], y = n.mx[1] - n.mn[1], z = n.mx[2] - n.mn[2]; return fmaf(x + y, z, x * y); }
static double global_sah(const std::vector<HN>& nodes)
{
    double cost = 0.0; const double rootArea = 1.0 / (double)half_area(nodes[1]);
    std::vector<int> st; st.push_back(1);
    while (!st.empty()) {
        const HN& n = nodes[st.back()]; st.pop_back();
        const double prob = (double)half_area(n) * rootArea;
        if (leaf(n)) cost += (double)(1.1f * (float)n.count) * prob;
        else { cost += (double)1.0f * prob; st.push_back(n.startOrChild + 1); st.push_back(n.startOrChild); }
    }
    return cost;
}
static void collapse(std::vector<HN>& nodes, int newStackSize, bool firstPass, double& nextCost, double rootHalfArea)
{
    // post-order without recursion: (node, depth, state)
    struct F { int p, depth, state; };
    std::vector<F> st; st.push_back({1, 0, 0});
    while (!st.empty()) {
        F& f = st.back();
        const int c = nodes[f.p].startOrChild;
        if (f.state == 0) { f.state = 1; if (!leaf(nodes[c])) { st.push_back({c, f.depth + 1, 0}); continue; } }
        if (f.state == 1) { f.state = 2; if (!leaf(nodes[c + 1])) { st.push_back({c + 1, f.depth + 1, 0}); continue; } }
        HN& p = nodes[f.p]; const HN l = nodes[c], r = nodes[c + 1]; const int depth = f.depth;
        if (leaf(l) && leaf(r)) {
            if (depth > newStackSize && !firstPass) { p.startOrChild = l.startOrChild; p.count = l.count + r.count; }
            if ((depth == newStackSize && !firstPass) || (depth > newStackSize && firstPass)) {
                const double leavesCost = (double)1.1f * ((double)l.count * (double)half_area(l) + (double)r.count * (double)half_area(r));
                const double newParentLeafCost = (double)1.1f * (double)(l.count + r.count);
                nextCost += ((double)half_area(nodes[f.p]) * (newParentLeafCost - (double)1.0f) - leavesCost) / rootHalfArea;
            }
        }
        st.pop_back();
    }
}
// returns the final RequiredStackSize; `nodes` is modified like the reference modifies its array
static int optimize(std::vector<HN>& nodes, int requiredStack)
{
    if (requiredStack < 16) return requiredStack;
    const double current = global_sah(nodes); double added = 0.0;
    const double rootHalfArea = (double)half_area(nodes[1]);
    collapse(nodes, requiredStack - 1, true, added, rootHalfArea);
    double inc = added / current;
    while (inc <= (double)0.0009745f && requiredStack > 0) { collapse(nodes, --requiredStack, false, added, rootHalfArea); inc = added / current; }
    return requiredStack;
}
}

// idkptBuildBlas: the whole BLAS build of one geometry on the device (bvh_gpu_full.hpp); results stay on the device until idkptBuildBlasFetch.
// Byte-identical to idkbvhBuildBlas (libidkbvh.so): nodes, triangles, parent / leaf indices, RequiredStackSize (tests/test_gpu_builder.py).
static int32_t dev_BuildBlas(dev_ctx* ctx, const float* positions, int32_t vertexCount, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable, float preSplitFactor, idkpt_blas_build_info* info)
{
    using namespace bvhgpu;
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(positions && tris && info && vertexCount > 0 && triCount > 0, "idkptBuildBlas: null argument or empty geometry");
    REQUIRE(triCount <= (1 << 26), "idkptBuildBlas: too many triangles");
    // One walk over the triangles: index range, the vertex range they reference, and — like idkbvhBuildBlas (bvh_builder.cpp), which this call mirrors — finiteness
    // of the REFERENCED positions only (the reference's builder has no defined result for NaN boxes / integer conversions of NaN; vertices no triangle of this BLAS
    // uses belong to other meshes of the host's global vertex array and are none of this call's business).
    uint32_t vmin = 0xffffffffu, vmax = 0u;
    {
        const uint32_t* pb = reinterpret_cast<const uint32_t*>(positions); uint32_t bad = 0;
        for (int i = 0; i < triCount; i++) {
            const uint32_t v[3] = {tris[i].X, tris[i].Y, tris[i].Z};
            REQUIRE(v[0] < (uint32_t)vertexCount && v[1] < (uint32_t)vertexCount && v[2] < (uint32_t)vertexCount, "idkptBuildBlas: triangle index out of range");
            for (int k = 0; k < 3; k++) {
                vmin = std::min(vmin, v[k]); vmax = std::max(vmax, v[k]);
                const uint32_t* q = pb + 3 * (size_t)v[k];
                bad |= (uint32_t)((q[0] & 0x7f800000u) == 0x7f800000u) | (uint32_t)((q[1] & 0x7f800000u) == 0x7f800000u) | (uint32_t)((q[2] & 0x7f800000u) == 0x7f800000u);
            }
        }
        REQUIRE(!bad, "idkptBuildBlas: a vertex position is not finite");
    }
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = builder_scratch(ctx);
    B.haveResult = false;
    hipStream_t st = ctx->stream;
    const auto t0 = std::chrono::steady_clock::now();
    const bool timing = ctx->opt.bvhTiming != 0;
    auto tq = t0;
    auto lap = [&](const char* what) { if (!timing) return; (void)hipStreamSynchronize(st); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[idkpt blas] %-14s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tq).count()); tq = t; };
    const int nT = triCount;
    const unsigned gT = (unsigned)((nT + 255) / 256);
    // only the vertex range this BLAS references crosses PCIe (a host that passes its GLOBAL vertex array for every BLAS pays for its own vertices, not for
    // O(#BLAS x V)); the kernels keep indexing with the host's ids through a pointer moved back by the range's first vertex
    const size_t vRange = (size_t)vmax - vmin + 1;
    HIPC(B.pos.ensure(vRange * 12)); HIPC(B.tris.ensure((size_t)nT * 16));
    HIPC(hipMemcpyAsync(B.pos.p, positions + 3 * (size_t)vmin, vRange * 12, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(B.tris.p, tris, (size_t)nT * 16, hipMemcpyHostToDevice, st));
    const float* dPos = B.pos.as<float>() - 3 * (ptrdiff_t)vmin; const uint4* dTris = B.tris.as<uint4>();
    lap("upload");
    // ---- fragments
    int F = nT;
    if (isRefittable) {
        HIPC(B.fb.ensure((size_t)nT * 32));
        hipLaunchKernelGGL(k_tri_boxes, dim3(gT), dim3(256), 0, st, dPos, dTris, nT, B.fb.as<float4>());
    } else {
        HIPC(B.prio.ensure((size_t)nT * 4)); HIPC(B.splitCnt.ensure(((size_t)nT + 1) * 4)); HIPC(B.first.ensure(((size_t)nT + 1) * 4));
        hipLaunchKernelGGL(k_tri_prio, dim3(gT), dim3(256), 0, st, dPos, dTris, nT, B.prio.as<float>());
        // the binary32 running sum of the priorities in index order (PreSplitting.cs:32-37): one dependent chain -> on the host
        std::vector<float> hp((size_t)nT);
        HIPC(hipMemcpyAsync(hp.data(), B.prio.p, (size_t)nT * 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        float total = 0.0f;
        for (int i = 0; i < nT; i++) total += hp[i];
        HIPC(B.status.ensure(16)); HIPC(hipMemsetAsync(B.status.p, 0, 16, st));      // [0..7] exact fragment count, [8..11] PreSplit stack flag
        hipLaunchKernelGGL(k_split_count, dim3(gT), dim3(256), 0, st, (const float*)B.prio.as<float>(), total, nT, preSplitFactor, B.splitCnt.as<uint32_t>(), B.status.as<unsigned long long>());
        HIPC(hipMemsetAsync(B.splitCnt.as<uint32_t>() + nT, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.splitCnt.as<uint32_t>(), B.first.as<uint32_t>(), (uint32_t)nT + 1u); if (rc) return rc; }
        uint32_t hF = 0; unsigned long long hSum = 0;
        HIPC(hipMemcpyAsync(&hF, B.first.as<uint32_t>() + nT, 4, hipMemcpyDeviceToHost, st)); HIPC(hipMemcpyAsync(&hSum, B.status.p, 8, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        REQUIRE(hSum <= (1ull << 27), "idkptBuildBlas: PreSplit asks for more than 2^27 fragments (split factor / priorities)");
        REQUIRE(hF == (uint32_t)hSum && hF >= (uint32_t)nT, "idkptBuildBlas: PreSplit produced an implausible fragment count");
        F = (int)hF;
        const int parts = (nT + CH - 1) / CH;
        HIPC(B.gboxPart.ensure((size_t)parts * sizeof(BBox))); HIPC(B.gbox.ensure(sizeof(BBox)));
        hipLaunchKernelGGL(k_global_box_partial, dim3(parts), dim3(CH), 0, st, dPos, dTris, nT, B.gboxPart.as<BBox>());
        hipLaunchKernelGGL(k_global_box_final, dim3(1), dim3(CH), 0, st, (const BBox*)B.gboxPart.as<BBox>(), parts, B.gbox.as<BBox>());
        HIPC(B.fb.ensure((size_t)F * 32)); HIPC(B.origTri.ensure((size_t)F * 4));
        hipLaunchKernelGGL(k_presplit, dim3((unsigned)((nT + 63) / 64)), dim3(64), 0, st, dPos, dTris, nT, (const uint32_t*)B.splitCnt.as<uint32_t>(), (const uint32_t*)B.first.as<uint32_t>(), (const BBox*)B.gbox.as<BBox>(), B.fb.as<float4>(), B.origTri.as<int>(), B.status.as<uint32_t>() + 2);
        uint32_t hOvf = 0;
        HIPC(hipMemcpyAsync(&hOvf, B.status.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        REQUIRE(!hOvf, "idkptBuildBlas: a triangle's PreSplit recursion needs more than 64 stack entries (the reference throws here, PreSplitting.cs:57)");
    }
    HIPC(hipGetLastError());
    lap("fragments");
    // ---- SweepSAH core
    int pp = 0, levels = 0;
    { int rc = bvh_core(ctx, B, F, &pp, &levels); if (rc) return rc; }
    lap("core");
    // ---- tail
    const int nodeCount = std::max(2 * F, 4);
    const unsigned gN = (unsigned)((nodeCount + 255) / 256);
    HNodeG* nodes = B.nodes.as<HNodeG>();
    const int* sorted0 = B.ids[0][pp].as<int>();
    HIPC(B.parent.ensure((size_t)nodeCount * 4)); HIPC(B.arrived.ensure((size_t)nodeCount * 4)); HIPC(B.need.ensure((size_t)nodeCount * 4)); HIPC(B.aggStart.ensure((size_t)nodeCount * 4)); HIPC(B.aggCount.ensure((size_t)nodeCount * 4));
    for (int k = 0; k < 2; k++) { HIPC(B.jump[k].ensure((size_t)nodeCount * 4)); HIPC(B.dist[k].ensure((size_t)nodeCount * 4)); }
    hipLaunchKernelGGL(k_fix_root, dim3(1), dim3(64), 0, st, nodes);
    hipLaunchKernelGGL(k_tree_init, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, B.parent.as<int>(), B.jump[0].as<int>(), B.dist[0].as<int>(), B.arrived.as<int>());
    // depths by pointer jumping (after r rounds dist = min(depth, 2^r)), then the bottom-up pass: required stack rows + fragment range of every subtree
    int curJ = 0;
    for (int span = 1; span < nodeCount; span <<= 1) {
        hipLaunchKernelGGL(k_depth_jump, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, (const int*)B.jump[curJ].as<int>(), (const int*)B.dist[curJ].as<int>(), B.jump[1 - curJ].as<int>(), B.dist[1 - curJ].as<int>());
        curJ = 1 - curJ;
    }
    const int* depthAll = B.dist[curJ].as<int>();
    HIPC(B.maxDepth.ensure(16)); HIPC(hipMemsetAsync(B.maxDepth.p, 0, 16, st));
    hipLaunchKernelGGL(k_leaf_init, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depthAll, B.need.as<int>(), B.aggStart.as<int>(), B.aggCount.as<int>(), B.maxDepth.as<int>());
    int treeDepth = 0;
    HIPC(hipMemcpyAsync(&treeDepth, B.maxDepth.p, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    if (treeDepth <= 512) {
        for (int d = treeDepth - 1; d >= 0; d--)
            hipLaunchKernelGGL(k_level_up, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depthAll, d, B.need.as<int>(), B.aggStart.as<int>(), B.aggCount.as<int>());
    } else   // a chain-like tree (identical fragments): one launch per level would be one per fragment -> the fenced climb
        hipLaunchKernelGGL(k_climb, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, (const int*)B.parent.as<int>(), B.arrived.as<int>(), B.need.as<int>(), B.aggStart.as<int>(), B.aggCount.as<int>());
    int hNeed = 0;
    HIPC(hipMemcpyAsync(&hNeed, B.need.as<int>() + 1, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    int requiredStack = hNeed;
    lap("bottom-up");
    int S = 0x7fffffff;                                              // depth threshold of the final collapse (none)
    const int* depth = nullptr;
    if (requiredStack >= 16) {                                       // StackOptThreshold (BLAS.cs:41, :880)
        depth = depthAll;
        const size_t nBins = 4 + 3 * (size_t)OPT_MAX_DEPTH;
        HIPC(B.bins.ensure(nBins * 8));
        HIPC(hipMemsetAsync(B.bins.p, 0, nBins * 8, st)); HIPC(hipMemsetAsync(B.maxDepth.p, 0, 16, st));
        hipLaunchKernelGGL(k_opt_sums, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depth, (const int*)B.aggCount.as<int>(), requiredStack, B.bins.as<double>(), B.maxDepth.as<int>());
        std::vector<double> bins(nBins); int maxDepth = 0;
        HIPC(hipMemcpyAsync(bins.data(), B.bins.p, nBins * 8, hipMemcpyDeviceToHost, st)); HIPC(hipMemcpyAsync(&maxDepth, B.maxDepth.p, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        lap("opt sums");
        // replay of the loop of BLAS.cs:882-894 on the sums.  The reference adds the same terms one by one in tree order; two binary64 summation
        // orders of N terms differ by at most ~2 N u sum|t| (u = 2^-53), so `inc <= acceptance` is decided here only outside that margin (x 8).
        const double accept = (double)0.0009745f, u = 1.1102230246251565e-16;
        bool certain = maxDepth < OPT_MAX_DEPTH && requiredStack - 1 < OPT_MAX_DEPTH && !ctx->opt.bvhStackOptHost;
        const double current = bins[0]; double added = bins[1], addedAbs = bins[2], nTerms = (double)nodeCount;
        int rs = requiredStack, sLast = 0x7fffffff;
        auto decide = [&](bool& le) {
            const double inc = added / current;
            const double margin = 8.0 * (2.0 * nTerms * u * addedAbs / current + 2.0 * nTerms * u * std::fabs(inc) + 4.0 * u * std::fabs(inc));
            if (std::fabs(inc - accept) <= margin || !(current > 0.0)) return false;
            le = inc <= accept; return true;
        };
        while (certain) {
            bool le = false;
            if (!decide(le)) { certain = false; break; }
            if (!(le && rs > 0)) break;
            rs--; sLast = rs;
            added += bins[4 + 3 * (size_t)rs]; addedAbs += bins[4 + 3 * (size_t)rs + 1];
        }
        if (certain) {
            requiredStack = rs; S = sLast;
            if (S != 0x7fffffff) hipLaunchKernelGGL(k_collapse, dim3(gN), dim3(256), 0, st, nodes, nodeCount, depth, (const int*)B.aggStart.as<int>(), (const int*)B.aggCount.as<int>(), S);
        } else {
            // never observed: a decision inside the rounding margin (or a tree deeper than the table) -> the reference's own walk on a host copy
            std::vector<stackopt_host::HN> hn((size_t)nodeCount);
            HIPC(hipMemcpyAsync(hn.data(), nodes, (size_t)nodeCount * 32, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
            requiredStack = stackopt_host::optimize(hn, requiredStack);
            HIPC(hipMemcpyAsync(nodes, hn.data(), (size_t)nodeCount * 32, hipMemcpyHostToDevice, st)); HIPC(hipStreamSynchronize(st));
            depth = nullptr; S = 0x7fffffff;                         // the host walk turned every collapsed node into a leaf: "internal" means live again
        }
    }
    lap("stack-opt");
    // RemoveEmptySubtrees: used child pairs in id order = the reference's pre-order numbering
    const int pairCount = nodeCount / 2;
    HIPC(B.used.ensure(((size_t)pairCount + 1) * 4)); HIPC(B.rank.ensure(((size_t)pairCount + 1) * 4));
    HIPC(hipMemsetAsync(B.used.p, 0, ((size_t)pairCount + 1) * 4, st));
    hipLaunchKernelGGL(k_mark_pairs, dim3(gN), dim3(256), 0, st, (const HNodeG*)nodes, nodeCount, depthAll, depth ? S : 0x7fffffff, B.used.as<uint32_t>());   // (no collapse on the device: every internal node is live)
    { int rc = scan_u32(ctx, B, B.used.as<uint32_t>(), B.rank.as<uint32_t>(), (uint32_t)pairCount + 1u); if (rc) return rc; }
    uint32_t usedPairs = 0;
    HIPC(hipMemcpyAsync(&usedPairs, B.rank.as<uint32_t>() + pairCount, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    const int outNodeCount = 2 + 2 * (int)usedPairs;
    HIPC(B.outNodes.ensure((size_t)outNodeCount * 32));
    hipLaunchKernelGGL(k_compact_nodes, dim3((unsigned)((pairCount + 255) / 256)), dim3(256), 0, st, (const HNodeG*)nodes, pairCount, (const uint32_t*)B.used.as<uint32_t>(), (const uint32_t*)B.rank.as<uint32_t>(), B.outNodes.as<HNodeG>());
    HNodeG* on = B.outNodes.as<HNodeG>();
    const unsigned gO = (unsigned)((outNodeCount + 255) / 256);
    lap("compact");
    // un-indexing
    int outTriCount = 0;
    if (isRefittable) {
        HIPC(B.leafCnt.ensure(((size_t)outNodeCount + 1) * 4)); HIPC(B.at.ensure(((size_t)outNodeCount + 1) * 4));
        hipLaunchKernelGGL(k_leaf_counts, dim3((unsigned)((outNodeCount + 1 + 255) / 256)), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.leafCnt.as<uint32_t>());
        HIPC(hipMemsetAsync(B.leafCnt.as<uint32_t>() + outNodeCount, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.leafCnt.as<uint32_t>(), B.at.as<uint32_t>(), (uint32_t)outNodeCount + 1u); if (rc) return rc; }
        uint32_t tot = 0; HIPC(hipMemcpyAsync(&tot, B.at.as<uint32_t>() + outNodeCount, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        outTriCount = (int)tot;
        HIPC(B.outTris.ensure(std::max<size_t>((size_t)outTriCount * 16, 16)));
        hipLaunchKernelGGL(k_unindex_plain, dim3(gO), dim3(256), 0, st, on, outNodeCount, (const uint32_t*)B.at.as<uint32_t>(), sorted0, dTris, B.outTris.as<uint4>());
    } else {
        const int pairs = (outNodeCount - 2) / 2;
        HIPC(B.uniq.ensure((size_t)2 * F * 4)); HIPC(B.ucount.ensure((size_t)outNodeCount * 4)); HIPC(B.leafCnt.ensure(((size_t)pairs + 1) * 4)); HIPC(B.at.ensure(((size_t)pairs + 1) * 4));
        HIPC(hipMemsetAsync(B.ucount.p, 0, (size_t)outNodeCount * 4, st));
        const unsigned gP = (unsigned)((pairs + 63) / 64);
        hipLaunchKernelGGL(k_unindex_ps_count, dim3(gP), dim3(64), 0, st, (const HNodeG*)on, pairs, sorted0, (const int*)B.origTri.as<int>(), F, B.uniq.as<int>(), B.ucount.as<int>(), B.leafCnt.as<uint32_t>());
        HIPC(hipMemsetAsync(B.leafCnt.as<uint32_t>() + pairs, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.leafCnt.as<uint32_t>(), B.at.as<uint32_t>(), (uint32_t)pairs + 1u); if (rc) return rc; }
        uint32_t tot = 0; HIPC(hipMemcpyAsync(&tot, B.at.as<uint32_t>() + pairs, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        outTriCount = (int)tot;
        HIPC(B.outTris.ensure(std::max<size_t>((size_t)outTriCount * 16, 16)));
        hipLaunchKernelGGL(k_unindex_ps_write, dim3(gP), dim3(64), 0, st, on, pairs, F, (const int*)B.uniq.as<int>(), (const int*)B.ucount.as<int>(), (const uint32_t*)B.at.as<uint32_t>(), dTris, B.outTris.as<uint4>());
    }
    lap("unindex");
    // parent / leaf indices (refittable BLASes only, BVH.cs:357-358)
    int leafCount = 0, parentCount = 0;
    if (isRefittable) {
        parentCount = outNodeCount;
        HIPC(B.parents.ensure((size_t)outNodeCount * 4)); HIPC(B.leafFlag.ensure(((size_t)outNodeCount + 1) * 4)); HIPC(B.rank.ensure(((size_t)outNodeCount + 1) * 4));
        hipLaunchKernelGGL(k_parents, dim3(gO), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.parents.as<int>());
        hipLaunchKernelGGL(k_leaf_flags, dim3(gO), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.leafFlag.as<uint32_t>());
        HIPC(hipMemsetAsync(B.leafFlag.as<uint32_t>() + outNodeCount, 0, 4, st));
        { int rc = scan_u32(ctx, B, B.leafFlag.as<uint32_t>(), B.rank.as<uint32_t>(), (uint32_t)outNodeCount + 1u); if (rc) return rc; }
        uint32_t tot = 0; HIPC(hipMemcpyAsync(&tot, B.rank.as<uint32_t>() + outNodeCount, 4, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
        leafCount = (int)tot;
        HIPC(B.leaves.ensure(std::max<size_t>((size_t)leafCount * 4, 16)));
        hipLaunchKernelGGL(k_leaf_list, dim3(gO), dim3(256), 0, st, (const uint32_t*)B.leafFlag.as<uint32_t>(), (const uint32_t*)B.rank.as<uint32_t>(), outNodeCount, B.leaves.as<int>());
    }
    // ComputeGlobalSAH of the finished tree
    const int sahBlocks = (int)gO;
    HIPC(B.sahPart.ensure((size_t)sahBlocks * 8));
    hipLaunchKernelGGL(k_sah_partial, dim3(gO), dim3(256), 0, st, (const HNodeG*)on, outNodeCount, B.sahPart.as<double>());
    HIPC(hipGetLastError());
    std::vector<double> sp((size_t)sahBlocks);
    HIPC(hipMemcpyAsync(sp.data(), B.sahPart.p, (size_t)sahBlocks * 8, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st));
    double sah = 0.0; for (double v : sp) sah += v;
    lap("indices+sah");
    B.outNodeCount = outNodeCount; B.outTriCount = outTriCount; B.outParentCount = parentCount; B.outLeafCount = leafCount; B.haveResult = true;
    info->NodeCount = outNodeCount; info->TriangleCount = outTriCount; info->RequiredStackSize = requiredStack; info->ParentIndexCount = parentCount; info->LeafIndexCount = leafCount;
    info->FragmentCount = F; info->Levels = levels; info->Sah = sah;
    info->BuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return IDKPT_OK;
}

static int32_t dev_BuildBlasFetch(dev_ctx* ctx, GpuBlasNode* nodes, GpuBlasTriangle* triangles, int32_t* parents, int32_t* leaves)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->bscratch || !ctx->bscratch->haveResult) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildBlasFetch: no finished idkptBuildBlas on this context");
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = *ctx->bscratch;
    hipStream_t st = ctx->stream;
    if (nodes) HIPC(hipMemcpyAsync(nodes, B.outNodes.p, (size_t)B.outNodeCount * 32, hipMemcpyDeviceToHost, st));
    if (triangles && B.outTriCount) HIPC(hipMemcpyAsync(triangles, B.outTris.p, (size_t)B.outTriCount * 16, hipMemcpyDeviceToHost, st));
    if (parents && B.outParentCount) HIPC(hipMemcpyAsync(parents, B.parents.p, (size_t)B.outParentCount * 4, hipMemcpyDeviceToHost, st));
    if (leaves && B.outLeafCount) HIPC(hipMemcpyAsync(leaves, B.leaves.p, (size_t)B.outLeafCount * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return IDKPT_OK;
}

// developer / test hook: the device cbrtf of the PreSplit priorities on an array of inputs (compared with the host's cbrtf by the tests)
static int32_t dev_CbrtProbe(dev_ctx* ctx, const float* in, float* out, int32_t n)
{
    if (!ctx || !in || !out || n <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    BuilderScratch& B = builder_scratch(ctx);
    HIPC(B.prio.ensure((size_t)n * 4)); HIPC(B.splitCnt.ensure((size_t)n * 4));
    HIPC(hipMemcpyAsync(B.prio.p, in, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(bvhgpu::k_cbrt_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)B.prio.as<float>(), B.splitCnt.as<float>(), n);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(out, B.splitCnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}
