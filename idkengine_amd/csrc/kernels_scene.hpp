// kernels_scene.hpp — what runs before the path on animated frames: TLAS rebuild (PLOC), BLAS refit, skinning.
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// TLAS rebuild on the device: BVH.TlasBuild (Bvh/BVH.cs:278-298) + TLAS.Build (Bvh/TLAS.cs:28-141).
// One 1024-thread workgroup (instance counts are small; the reference does this serially on the CPU every animated frame and
// re-uploads): world bounds of every instance's BLAS root (Box.Transformed, Shapes/Box.cs:177-187), Morton-30 order (stable
// rank = the reference's stable LSD radix sort), then PLOC rounds: every node picks its best partner inside +-searchRadius
// (FindBestMatch, TLAS.cs:271-301, strict '<' keeps the first best), mutual pairs merge, output positions come from an ordered
// block scan so the node array is identical to the serial build, bit for bit.
// Instance records (DScene::instRec): everything an instance entry of the traversal kernels reads, gathered per instance.  Re-derived after uploads, transform updates, refits, node patches.
__global__ void k_inst_records(const float4* blasNodes, const GpuBlasDesc* descs, const GpuBlasInstance* instances, const float4* xforms, int n, float4* out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const GpuBlasInstance in = instances[i];
    const GpuBlasDesc d = descs[in.BlasId];
    const float4* root = blasNodes + 2 * ((size_t)d.NodeOffset + 1);
    const float4* x = xforms + 9 * (size_t)in.MeshTransformId;
    float4* o = out + 6 * (size_t)i;
    o[0] = x[3]; o[1] = x[4]; o[2] = x[5];
    const float4 rmin = root[0], rmax = root[1];
    o[3] = make_float4(rmin.x, rmin.y, rmin.z, __uint_as_float((uint32_t)d.NodeOffset));
    o[4] = make_float4(rmax.x, rmax.y, rmax.z, __uint_as_float((uint32_t)d.TriangleOffset));
    o[5] = make_float4(__uint_as_float((uint32_t)in.MeshTransformId), __uint_as_float((uint32_t)in.BlasId), __uint_as_float(2u) /* the pair a walk entered here starts at: the root's children */, __uint_as_float(1u) /* the node this record's box is of */);
}

#define TLAS_BUILD_THREADS 1024
DEV uint32_t tlas_insert_two_zeros(uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; }
DEV uint32_t tlas_to_uint_sat(float f) { if (f != f || f <= 0.0f) return 0u; if (f >= 4294967296.0f) return 0xffffffffu; return (uint32_t)f; }
DEV float tlas_minN(float a, float b) { return a < b ? a : b; }   // minps / float.MinNative (Shapes/Box.cs:40-50)
DEV float tlas_maxN(float a, float b) { return a > b ? a : b; }
DEV float tlas_half_area(float4 mn, float4 mx) { float sx = mx.x - mn.x, sy = mx.y - mn.y, sz = mx.z - mn.z; return __fmaf_rn(sx + sy, sz, sx * sy); }   // MyMath.cs:222-229

// The library's own TLAS for the instance loop (kernels_trace_inst.hpp): the loop takes a ray into BLAS space with InvModel and nothing else (BVHIntersect.glsl:281-282), so the
// world-space region whose rays can pass the test of a BLAS-space box is the image of that box under the inverse OF InvModel — inverted here, in double, whatever the host's Model
// says — padded by 2^-10 of its size and of its distance from the origin (thousands of fp32 roundings of either transform).  A singular or non-finite InvModel: all of space.
// x: the instance's GpuMeshTransform rows (x[3..5] = InvModel); rmin / rmax: the box in BLAS space.
DEV void tlas_padded_world_box(const float4* x, const float4 rmin, const float4 rmax, float bmn[3], float bmx[3])
{
    const float4 i0 = x[3], i1 = x[4], i2 = x[5];
    const double A[3][3] = {{i0.x, i0.y, i0.z}, {i1.x, i1.y, i1.z}, {i2.x, i2.y, i2.z}}, tb[3] = {i0.w, i1.w, i2.w};
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    const double inv[3][3] = {{c00 / det, (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det, (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det},
                              {c01 / det, (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det, (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det},
                              {c02 / det, (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det, (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det}};
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool ok = det == det && fabs(det) > 1e-300;
    for (int c = 0; c < 8; c++) {
        const double q[3] = {(double)((c & 1) ? rmax.x : rmin.x) - tb[0], (double)((c & 2) ? rmax.y : rmin.y) - tb[1], (double)((c & 4) ? rmax.z : rmin.z) - tb[2]};
        for (int k = 0; k < 3; k++) { const double w = inv[k][0] * q[0] + inv[k][1] * q[1] + inv[k][2] * q[2]; ok = ok && w == w && fabs(w) < 1e30; lo[k] = w < lo[k] ? w : lo[k]; hi[k] = w > hi[k] ? w : hi[k]; }
    }
    for (int k = 0; k < 3; k++) {
        const double ext = hi[k] - lo[k], mag = fabs(lo[k]) > fabs(hi[k]) ? fabs(lo[k]) : fabs(hi[k]);
        const double pad = (ext > mag ? ext : mag) * (1.0 / 1024.0) + 1e-30;
        bmn[k] = ok ? (float)(lo[k] - pad) : -PT_FLOAT_MAX; bmx[k] = ok ? (float)(hi[k] + pad) : PT_FLOAT_MAX;
    }
}

// ordered exclusive scan of two per-thread counters over the workgroup; returns this thread's bases and the totals
DEV void block_scan2(uint32_t a, uint32_t b, uint32_t* sa, uint32_t* sb, uint32_t& baseA, uint32_t& baseB, uint32_t& totA, uint32_t& totB)
{
    const uint32_t t = threadIdx.x;
    sa[t] = a; sb[t] = b;
    __syncthreads();
    for (uint32_t off = 1; off < TLAS_BUILD_THREADS; off <<= 1) {
        uint32_t va = t >= off ? sa[t - off] : 0u, vb = t >= off ? sb[t - off] : 0u;
        __syncthreads();
        sa[t] += va; sb[t] += vb;
        __syncthreads();
    }
    baseA = sa[t] - a; baseB = sb[t] - b; totA = sa[TLAS_BUILD_THREADS - 1]; totB = sb[TLAS_BUILD_THREADS - 1];
    __syncthreads();
}

// (braid != nullptr — the library's own TLAS over the BRAIDED entries of k_braid below: the leaf boxes are taken from braid->leaf as they are, their number from braid->count;
//  overlapOut[1] = that number.  padded, after a full build: overlapOut[2] = the depth of the tree = the rows a front-to-back walk's stack can need)
struct BraidOut { const float4* leaf; const int* count; const int* ub = nullptr; };   // ub (unified tree only): per entry, an upper bound of the stack rows a walk of its subtree needs
__global__ __launch_bounds__(TLAS_BUILD_THREADS) void k_tlas_build(const float4* blasNodes, const GpuBlasDesc* descs, const GpuBlasInstance* instances, const float4* xforms,
                                                                    int n, int searchRadius, float4* nodes /* 2n-1 */, float4* temp /* 2n-1 */, float4* leaf /* n */, uint32_t* keys /* n */, int* pref /* n */,
                                                                    int padded = 0, float* overlapOut = nullptr, int leavesOnly = 0, BraidOut braid = BraidOut{nullptr, nullptr})
{
    __shared__ float red[6][TLAS_BUILD_THREADS / 64];
    __shared__ float gbox[6];
    __shared__ uint32_t sa[TLAS_BUILD_THREADS], sb[TLAS_BUILD_THREADS];
    const int t = (int)threadIdx.x, T = TLAS_BUILD_THREADS;
    if (braid.count) n = *braid.count;
    const int nodeCount = 2 * n - 1;
    // ---- leaves: world-space bounds of every instance
    float mn[3] = {PT_FLOAT_MAX, PT_FLOAT_MAX, PT_FLOAT_MAX}, mx[3] = {-PT_FLOAT_MAX, -PT_FLOAT_MAX, -PT_FLOAT_MAX};
    if (braid.leaf) {
        for (int i = t; i < n; i += T) {
            const float4 a = braid.leaf[2 * (size_t)i], b = braid.leaf[2 * (size_t)i + 1];
            leaf[2 * (size_t)i] = a; leaf[2 * (size_t)i + 1] = b;
            mn[0] = tlas_minN(mn[0], a.x); mn[1] = tlas_minN(mn[1], a.y); mn[2] = tlas_minN(mn[2], a.z); mx[0] = tlas_maxN(mx[0], b.x); mx[1] = tlas_maxN(mx[1], b.y); mx[2] = tlas_maxN(mx[2], b.z);
        }
    } else
    for (int i = t; i < n; i += T) {
        const GpuBlasInstance in = instances[i];
        const float4* root = blasNodes + 2 * ((size_t)descs[in.BlasId].NodeOffset + 1);
        const float4 rmin = root[0], rmax = root[1];
        const float4* x = xforms + 9 * (size_t)in.MeshTransformId;
        const float4 m0 = x[0], m1 = x[1], m2 = x[2];
        float bmn[3] = {PT_FLOAT_MAX, PT_FLOAT_MAX, PT_FLOAT_MAX}, bmx[3] = {-PT_FLOAT_MAX, -PT_FLOAT_MAX, -PT_FLOAT_MAX};
        if (padded) tlas_padded_world_box(x, rmin, rmax, bmn, bmx);     // (the library's own TLAS for the instance loop, kernels_trace_inst.hpp)
        else
        for (int c = 0; c < 8; c++) {
            const float cx = (c & 1) ? rmax.x : rmin.x, cy = (c & 2) ? rmax.y : rmin.y, cz = (c & 4) ? rmax.z : rmin.z;
            const float w[3] = {(cx * m0.x) + (cy * m0.y) + (cz * m0.z) + (1.0f * m0.w), (cx * m1.x) + (cy * m1.y) + (cz * m1.z) + (1.0f * m1.w), (cx * m2.x) + (cy * m2.y) + (cz * m2.z) + (1.0f * m2.w)};
            for (int k = 0; k < 3; k++) { bmn[k] = tlas_minN(bmn[k], w[k]); bmx[k] = tlas_maxN(bmx[k], w[k]); }
        }
        leaf[2 * (size_t)i] = make_float4(bmn[0], bmn[1], bmn[2], __uint_as_float((1u << 31) | (uint32_t)i));
        leaf[2 * (size_t)i + 1] = make_float4(bmx[0], bmx[1], bmx[2], 0.0f);
        for (int k = 0; k < 3; k++) { mn[k] = tlas_minN(mn[k], bmn[k]); mx[k] = tlas_maxN(mx[k], bmx[k]); }
    }
    // global box (min/max: order independent)
    for (int k = 0; k < 3; k++) {
        float a = mn[k], b = mx[k];
        for (int off = 32; off > 0; off >>= 1) { a = tlas_minN(a, __shfl_xor(a, off)); b = tlas_maxN(b, __shfl_xor(b, off)); }
        if ((t & 63) == 0) { red[k][t >> 6] = a; red[3 + k][t >> 6] = b; }
    }
    __syncthreads();
    if (t < 3) { float a = PT_FLOAT_MAX, b = -PT_FLOAT_MAX; for (int w = 0; w < T / 64; w++) { a = tlas_minN(a, red[t][w]); b = tlas_maxN(b, red[3 + t][w]); } gbox[t] = a; gbox[3 + t] = b; }
    __syncthreads();
    if (overlapOut) {
        // how many instance boxes a random line through the scene's box meets (sum of the boxes' half areas over the half area of their union): what the instance loop
        // walks per ray against the n root tests it makes — the host's measure of whether a tree over the instances can pay (host_launch.hpp inst_tlas_prepare)
        float sum = 0.0f;
        for (int i = t; i < n; i += T) sum += tlas_half_area(leaf[2 * (size_t)i], leaf[2 * (size_t)i + 1]);
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        __syncthreads();
        if ((t & 63) == 0) red[0][t >> 6] = sum;
        __syncthreads();
        if (t == 0) {
            float tot = 0.0f; for (int w = 0; w < T / 64; w++) tot += red[0][w];
            const float all = tlas_half_area(make_float4(gbox[0], gbox[1], gbox[2], 0.0f), make_float4(gbox[3], gbox[4], gbox[5], 0.0f));
            const float e = tot / all;
            *overlapOut = (e == e && e >= 0.0f && e < 3.0e38f) ? e : (float)n;      // (degenerate or unbounded boxes: "every instance")
            if (braid.count) overlapOut[1] = (float)n;
        }
        __syncthreads();
    }
    if (leavesOnly) return;
    // ---- Morton-30 keys of the box centres (MyMath.cs:241-257, 283-299)
    for (int i = t; i < n; i += T) {
        const float4 a = leaf[2 * (size_t)i], b = leaf[2 * (size_t)i + 1];
        const float c[3] = {(b.x + a.x) * 0.5f, (b.y + a.y) * 0.5f, (b.z + a.z) * 0.5f};
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            const float ext = gbox[3 + k] - gbox[k];
            float r = (c[k] - gbox[k]) / ext * (1.0f - 0.0f) + 0.0f;
            if (ext == 0.0f) r = 0.0f;
            const uint32_t u = tlas_to_uint_sat(r * 1024.0f);
            q[k] = u < 1023u ? u : 1023u;
        }
        keys[i] = (tlas_insert_two_zeros(q[0]) << 2) | (tlas_insert_two_zeros(q[1]) << 1) | tlas_insert_two_zeros(q[2]);
    }
    __syncthreads();
    // ---- stable sort by key (rank counting) into the tail of the node array
    for (int i = t; i < n; i += T) {
        const uint32_t ki = keys[i];
        int rank = 0;
        for (int j = 0; j < n; j++) { const uint32_t kj = keys[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
        const size_t d = (size_t)(nodeCount - n + rank);
        nodes[2 * d] = leaf[2 * (size_t)i]; nodes[2 * d + 1] = leaf[2 * (size_t)i + 1];
    }
    __syncthreads();
    // ---- PLOC rounds
    int activeCount = n, activeEnd = nodeCount;
    while (activeCount > 1) {
        const int start = activeEnd - activeCount;
        for (int i = t; i < activeCount; i += T) {
            const int a = start + i;
            const int s0 = max(a - searchRadius, start), s1 = min(a + searchRadius + 1, activeEnd);
            const float4 amn = nodes[2 * (size_t)a], amx = nodes[2 * (size_t)a + 1];
            float smallest = PT_FLOAT_MAX; int best = -1;
            for (int k = s0; k < s1; k++) {
                if (k == a) continue;
                const float4 omn = nodes[2 * (size_t)k], omx = nodes[2 * (size_t)k + 1];
                const float4 un = make_float4(tlas_minN(amn.x, omn.x), tlas_minN(amn.y, omn.y), tlas_minN(amn.z, omn.z), 0.0f);
                const float4 ux = make_float4(tlas_maxN(amx.x, omx.x), tlas_maxN(amx.y, omx.y), tlas_maxN(amx.z, omx.z), 0.0f);
                const float area = tlas_half_area(un, ux);
                if (area < smallest) { smallest = area; best = k; }
            }
            pref[i] = best - start;
        }
        __syncthreads();
        // contiguous chunk per thread so that the scan order is the serial loop's order
        const int chunk = (activeCount + T - 1) / T, c0 = min(t * chunk, activeCount), c1 = min(c0 + chunk, activeCount);
        uint32_t nPairs = 0, nOut = 0;
        uint32_t basePairs, baseOut, totPairs, totOut;
        for (int attempt = 0; attempt < 2; attempt++) {
            nPairs = 0; nOut = 0;
            for (int i = c0; i < c1; i++) { const int b = pref[i]; const bool mutual = b >= 0 && pref[b] == i; if (mutual && i < b) nPairs++; if (!mutual || i < b) nOut++; }
            block_scan2(nPairs, nOut, sa, sb, basePairs, baseOut, totPairs, totOut);
            if (totPairs != 0u || attempt == 1) break;
            // No mutual pair at all: every union area was infinite or NaN (an instance under a singular transform has "all of space" as its padded box; FindBestMatch's strict '<'
            // from FLT_MAX then never picks anybody, and the serial loop of TLAS.cs would not end either).  Merge the first two so that the build terminates.
            if (t == 0) { pref[0] = 1; pref[1] = 0; }
            __syncthreads();
        }
        const int merged = 2 * (int)totPairs, unmerged = activeCount - merged, newNodes = merged / 2;
        const int mergedHead0 = activeEnd - merged, newBegin = mergedHead0 - unmerged - newNodes;
        int mergedHead = mergedHead0 + 2 * (int)basePairs, unmergedHead = newBegin + (int)baseOut;
        for (int i = c0; i < c1; i++) {
            const int b = pref[i]; const bool mutual = b >= 0 && pref[b] == i; const size_t aId = (size_t)(i + start);
            if (mutual) {
                if (i < b) {
                    const size_t bId = (size_t)(b + start);
                    const float4 amn = nodes[2 * aId], amx = nodes[2 * aId + 1], bmn = nodes[2 * bId], bmx = nodes[2 * bId + 1];
                    temp[2 * (size_t)mergedHead] = amn; temp[2 * (size_t)mergedHead + 1] = amx; temp[2 * (size_t)mergedHead + 2] = bmn; temp[2 * (size_t)mergedHead + 3] = bmx;
                    temp[2 * (size_t)unmergedHead] = make_float4(tlas_minN(amn.x, bmn.x), tlas_minN(amn.y, bmn.y), tlas_minN(amn.z, bmn.z), __uint_as_float((uint32_t)mergedHead));
                    temp[2 * (size_t)unmergedHead + 1] = make_float4(tlas_maxN(amx.x, bmx.x), tlas_maxN(amx.y, bmx.y), tlas_maxN(amx.z, bmx.z), 0.0f);
                    unmergedHead++; mergedHead += 2;
                }
            } else { temp[2 * (size_t)unmergedHead] = nodes[2 * aId]; temp[2 * (size_t)unmergedHead + 1] = nodes[2 * aId + 1]; unmergedHead++; }
        }
        __syncthreads();
        for (int i = newBegin + t; i < activeEnd; i += T) { nodes[2 * (size_t)i] = temp[2 * (size_t)i]; nodes[2 * (size_t)i + 1] = temp[2 * (size_t)i + 1]; }
        __syncthreads();
        activeCount -= merged / 2; activeEnd -= merged;
    }
    if (padded && overlapOut) {
        // depth of the finished tree (children sit behind their parent in the array; level by level from the root, `pref` reused as the per-node depth)
        int* const depth = pref;                                 // (2n - 1 words: the own-TLAS caller sizes `pref` for that)
        for (int i = t; i < nodeCount; i += T) depth[i] = i == 0 ? 1 : 0;
        if (t == 0) overlapOut[2] = (float)(4 * TLAS_STACK_SIZE);
        __syncthreads();
        for (int level = 1; level <= 4 * TLAS_STACK_SIZE; level++) {
            uint32_t any = 0;
            for (int i = t; i < nodeCount; i += T) {
                if (depth[i] != level) continue;
                const uint32_t packed = __float_as_uint(nodes[2 * (size_t)i].w);
                if ((packed >> 31) == 0u) { depth[packed] = level + 1; depth[packed + 1] = level + 1; any = 1; }
            }
            sa[t] = any;
            __syncthreads();
            for (int off = T / 2; off > 0; off >>= 1) { if (t < off) sa[t] |= sa[t + off]; __syncthreads(); }
            const uint32_t more = sa[0];
            __syncthreads();
            if (!more) { if (t == 0) overlapOut[2] = (float)level; break; }
        }
        if (braid.ub) {
            // rows a walk of the whole tree needs: over the leaves, the inner nodes above one (the step at each can leave one entry on the stack) + its subtree's bound
            uint32_t need = 0;
            for (int i = t; i < nodeCount; i += T) {
                const uint32_t packed = __float_as_uint(nodes[2 * (size_t)i].w);
                if ((packed >> 31) == 1u) need = max(need, (uint32_t)max(0, depth[i] - 1 + braid.ub[packed & 0x7fffffffu]));
            }
            __syncthreads();
            sa[t] = need;
            __syncthreads();
            for (int off = T / 2; off > 0; off >>= 1) { if (t < off) sa[t] = max(sa[t], sa[t + off]); __syncthreads(); }
            if (t == 0) overlapOut[3] = (float)sa[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Partial re-braiding for the library's own TLAS (kernels_trace_inst.hpp): the tree's leaves need not be whole instances.  The reference holds one BLAS per mesh (Bvh/BVH.cs:156) and a
// building's meshes overlap: the atrium's 87 root boxes make a ray of the own-TLAS walk enter 2.2 x the nodes of the same triangles in one BLAS.  An ENTRY is (instance, BLAS node); the
// list starts as (i, root) for every instance, and the entries with the largest world-space boxes are OPENED — replaced by the node's two children — until the list has `budget` entries
// or nothing large is left to open.  A walk that enters an entry transforms the ray exactly as the loop does (the instance's InvModel), tests the entry's own BLAS box with the loop's
// RayBoxIntersect and continues at the node's child pair: every number is still the loop's arithmetic on the loop's data, only more of the order is the tree's (the header of
// kernels_trace_inst.hpp: the loop reaches the winning leaf because the boxes of its ancestors — now including the ones above the entry, never tested by this walk — contain the leaf
// box, so their exact tests pass whenever the leaf's does; the host braids only scenes whose BLAS boxes nest, dev_ctx::sceneNested).  Opened are internal nodes whose two children are
// internal (an entry's walk starts at a child PAIR).
// One 1024-thread workgroup, rounds: area of the new entries, block maximum A over the openable ones, every openable entry with area >= A / 2 is opened while the budget lasts (in list
// order: an ordered scan), the left child takes the parent's place, the right one is appended.  Deterministic; a few dozen rounds.
// Output: ent[e] = (instance, node); rec = an instance record per ENTRY (k_inst_records' layout: InvModel rows, {node box, NodeOffset / TriangleOffset}, {MeshTransformId, BlasId,
// child pair, node}); leaf = the entries' padded world boxes in k_tlas_build's leaf layout; *countOut = entries.
// unified != 0 (k_unify_*, below: every instance has the same InvModel, so one BLAS space holds the whole scene): an entry's box is its BLAS box itself, no image, no padding, no
// records.
__global__ __launch_bounds__(TLAS_BUILD_THREADS) void k_braid(const float4* blasNodes, const GpuBlasDesc* descs, const GpuBlasInstance* instances, const float4* xforms, int n0, int budget,
                                                               uint2* ent /* budget */, float* area /* budget */, float4* rec /* 6 x budget */, float4* leaf /* 2 x budget */, int* countOut,
                                                               int unified = 0, int* ub = nullptr /* budget: per entry, rows its subtree's walk can need (from BlasDesc.RequiredStackSize, one less per opening) */)
{
    __shared__ uint32_t sa[TLAS_BUILD_THREADS], sb[TLAS_BUILD_THREADS];
    __shared__ float red[TLAS_BUILD_THREADS / 64];
    const int t = (int)threadIdx.x, T = TLAS_BUILD_THREADS;
    int count = n0;
    for (int i = t; i < n0; i += T) { ent[i] = make_uint2((uint32_t)i, 1u); area[i] = -1.0f; if (ub) ub[i] = descs[instances[i].BlasId].RequiredStackSize; }
    __syncthreads();
    for (int round = 0; round < 96 && count < budget; round++) {
        // areas of the entries made by the last round (0 = not openable)
        float best = 0.0f;
        for (int e = t; e < count; e += T) {
            float a = area[e];
            if (a < 0.0f) {
                const uint2 en = ent[e];
                const GpuBlasInstance in = instances[en.x];
                const float4* nd = blasNodes + 2 * ((size_t)descs[in.BlasId].NodeOffset + en.y);
                const float4 bmin = nd[0], bmax = nd[1];
                a = 0.0f;
                if (__float_as_uint(bmax.w) == 0u) {                                  // internal
                    const float4* ch = blasNodes + 2 * ((size_t)descs[in.BlasId].NodeOffset + __float_as_uint(bmin.w));
                    if (__float_as_uint(ch[1].w) == 0u && __float_as_uint(ch[3].w) == 0u) {   // both children internal
                        float wmn[3] = {bmin.x, bmin.y, bmin.z}, wmx[3] = {bmax.x, bmax.y, bmax.z};
                        if (!unified) tlas_padded_world_box(xforms + 9 * (size_t)in.MeshTransformId, bmin, bmax, wmn, wmx);
                        const float h = tlas_half_area(make_float4(wmn[0], wmn[1], wmn[2], 0.0f), make_float4(wmx[0], wmx[1], wmx[2], 0.0f));
                        a = (h == h && h > 0.0f && h < 3.0e38f) ? h : 0.0f;
                    }
                }
                area[e] = a;
            }
            best = tlas_maxN(best, a);
        }
        for (int off = 32; off > 0; off >>= 1) best = tlas_maxN(best, __shfl_xor(best, off));
        if ((t & 63) == 0) red[t >> 6] = best;
        __syncthreads();
        float A = 0.0f;
        for (int w = 0; w < T / 64; w++) A = tlas_maxN(A, red[w]);
        __syncthreads();
        if (!(A > 0.0f)) break;
        const float thresh = A * 0.5f;
        const int chunk = (count + T - 1) / T, c0 = min(t * chunk, count), c1 = min(c0 + chunk, count);
        uint32_t nOpen = 0;
        for (int e = c0; e < c1; e++) nOpen += area[e] >= thresh ? 1u : 0u;
        uint32_t base, dummy, total, dummyT;
        block_scan2(nOpen, 0u, sa, sb, base, dummy, total, dummyT);
        const uint32_t allowed = min(total, (uint32_t)(budget - count));
        uint32_t rank = base;
        for (int e = c0; e < c1; e++) {
            if (!(area[e] >= thresh)) continue;
            if (rank < allowed) {
                const uint2 en = ent[e];
                const uint32_t child = __float_as_uint(blasNodes[2 * ((size_t)descs[instances[en.x].BlasId].NodeOffset + en.y)].w);
                ent[e] = make_uint2(en.x, child); area[e] = -1.0f;
                ent[count + rank] = make_uint2(en.x, child + 1u); area[count + rank] = -1.0f;
                if (ub) { const int u = max(0, ub[e] - 1); ub[e] = u; ub[count + rank] = u; }   // (need(parent) >= 1 + need(child))
            }
            rank++;
        }
        count += (int)allowed;
        __syncthreads();
    }
    for (int e = t; e < count; e += T) {
        const uint2 en = ent[e];
        const GpuBlasInstance in = instances[en.x];
        const GpuBlasDesc d = descs[in.BlasId];
        const float4* nd = blasNodes + 2 * ((size_t)d.NodeOffset + en.y);
        const float4 bmin = nd[0], bmax = nd[1];
        const float4* x = xforms + 9 * (size_t)in.MeshTransformId;
        if (unified) {
            leaf[2 * (size_t)e] = make_float4(bmin.x, bmin.y, bmin.z, __uint_as_float((1u << 31) | (uint32_t)e));
            leaf[2 * (size_t)e + 1] = make_float4(bmax.x, bmax.y, bmax.z, 0.0f);
            continue;
        }
        float4* o = rec + 6 * (size_t)e;
        o[0] = x[3]; o[1] = x[4]; o[2] = x[5];
        o[3] = make_float4(bmin.x, bmin.y, bmin.z, __uint_as_float((uint32_t)d.NodeOffset));
        o[4] = make_float4(bmax.x, bmax.y, bmax.z, __uint_as_float((uint32_t)d.TriangleOffset));
        o[5] = make_float4(__uint_as_float((uint32_t)in.MeshTransformId), __uint_as_float((uint32_t)in.BlasId), en.y == 1u ? __uint_as_float(2u) : bmin.w /* the pair a walk entered here starts at: the root's children (k_inst_records), or the child pair of an opened node's internal child */, __uint_as_float(en.y));
        float wmn[3], wmx[3];
        tlas_padded_world_box(x, bmin, bmax, wmn, wmx);
        leaf[2 * (size_t)e] = make_float4(wmn[0], wmn[1], wmn[2], __uint_as_float((1u << 31) | (uint32_t)e));
        leaf[2 * (size_t)e + 1] = make_float4(wmx[0], wmx[1], wmx[2], 0.0f);
    }
    if (t == 0) *countOut = count;
}

// ---------------------------------------------------------------------------------------------------------------
// The UNIFIED tree (kernels_trace_inst.hpp, UNI): when every instance of the scene carries the same InvModel — the reference's usual static scene, one BLAS per mesh of a model and
// one node transform (Bvh/BVH.cs:156) — the loop takes every ray into the same BLAS space, and the instances' BLASes are subtrees of ONE BVH2 in that space.  The derived array:
//   [0, 2 x cap)            the top: k_tlas_build's PLOC tree over k_braid's entries (their exact BLAS boxes), in GpuBlasNode's layout (root = node 1, its children = the pair at 2;
//                           TLAS node i -> node i + 1); a leaf slot holds the entry's own BLAS node with its child rebased, so the walk runs through it into the BLAS below;
//   [2 x cap, + all nodes)  every BLAS's nodes as uploaded, children rebased to this array, leaf ranges to scene-wide triangle indices.
// Boxes are copied, never recomputed: every test is the loop's test on the loop's box; the top's inner boxes are exact unions (min / max) of the entries' boxes and only order the walk.
// general != 0 (kernels_trace_inst.hpp TREE 2: the instances carry DIFFERENT transforms — the top lives in world space, every BLAS in its instance's space): the top's boxes are
// the entries' padded world boxes (k_braid, unified = 0), and a leaf slot is an inner node over a two-node STUB at stubBase + 2 x entry: {always-hit box (-inf .. +inf), entry id,
// count = UNIFY_MARK_ENTRY} and a never-hit sibling (NaN bounds: t2 = NaN, `t1 <= t2` false) — the walk's node step parks the lane on it as on a leaf, and its leaf phase, seeing the
// mark instead of a triangle count, takes the ray into the entry's instance (RayTransform with the entry record's InvModel), pushes the RESTORE pair and continues at the entry's own
// child pair in the BLAS region.  The RESTORE pair at restoreIdx = the same trick with UNIFY_MARK_RESTORE: popped when the instance's subtree is done, it puts the world ray back.
__global__ __launch_bounds__(TLAS_BUILD_THREADS) void k_unify_top(const float4* tlasNodes, const int* countPtr, const uint2* ent, const float4* blasNodes, const GpuBlasDesc* descs,
                                                                   const GpuBlasInstance* instances, uint32_t baseB, float4* unodes, int general = 0, uint32_t stubBase = 0, uint32_t restoreIdx = 0)
{
    const int n = *countPtr, nodeCount = 2 * n - 1;
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    for (int i = (int)threadIdx.x; i < nodeCount; i += TLAS_BUILD_THREADS) {
        const float4 mn = tlasNodes[2 * (size_t)i], mx = tlasNodes[2 * (size_t)i + 1];
        const uint32_t packed = __float_as_uint(mn.w);
        float4 o0, o1;
        if ((packed >> 31) == 1u && general) {
            const uint32_t e = packed & 0x7fffffffu;
            o0 = make_float4(mn.x, mn.y, mn.z, __uint_as_float(stubBase + 2u * e));
            o1 = make_float4(mx.x, mx.y, mx.z, __uint_as_float(0u));
            float4* st = unodes + 2 * (size_t)(stubBase + 2u * e);
            st[0] = make_float4(-inf, -inf, -inf, __uint_as_float(e)); st[1] = make_float4(inf, inf, inf, __uint_as_float(UNIFY_MARK_ENTRY));
            st[2] = make_float4(nan, nan, nan, __uint_as_float(0u)); st[3] = make_float4(nan, nan, nan, __uint_as_float(0u));
        } else if ((packed >> 31) == 1u) {
            const uint2 en = ent[packed & 0x7fffffffu];
            const GpuBlasDesc d = descs[instances[en.x].BlasId];
            const float4 bmin = blasNodes[2 * ((size_t)d.NodeOffset + en.y)], bmax = blasNodes[2 * ((size_t)d.NodeOffset + en.y) + 1];
            const bool isLeaf = __float_as_uint(bmax.w) != 0u;
            o0 = make_float4(bmin.x, bmin.y, bmin.z, __uint_as_float(__float_as_uint(bmin.w) + (isLeaf ? (uint32_t)d.TriangleOffset : baseB + (uint32_t)d.NodeOffset)));
            o1 = bmax;
        } else {
            o0 = make_float4(mn.x, mn.y, mn.z, __uint_as_float(packed + 1u));
            o1 = make_float4(mx.x, mx.y, mx.z, __uint_as_float(0u));
        }
        unodes[2 * (size_t)(i + 1)] = o0; unodes[2 * (size_t)(i + 1) + 1] = o1;
    }
    if (threadIdx.x == 0) {
        unodes[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); unodes[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (general) {
            float4* st = unodes + 2 * (size_t)restoreIdx;
            st[0] = make_float4(-inf, -inf, -inf, __uint_as_float(0u)); st[1] = make_float4(inf, inf, inf, __uint_as_float(UNIFY_MARK_RESTORE));
            st[2] = make_float4(nan, nan, nan, __uint_as_float(0u)); st[3] = make_float4(nan, nan, nan, __uint_as_float(0u));
        }
    }
}
// (chunk k of 256 nodes belongs to BLAS chunks[k].x and starts at its node chunks[k].y: k_mark_triangles' table)
__global__ __launch_bounds__(256) void k_unify_blas(const float4* blasNodes, const GpuBlasDesc* descs, const uint2* chunks, uint32_t baseB, float4* unodes)
{
    const uint2 ch = chunks[blockIdx.x];
    const GpuBlasDesc d = descs[ch.x];
    const uint32_t k = ch.y + threadIdx.x;
    if (k >= (uint32_t)d.NodeCount) return;
    const size_t src = (size_t)d.NodeOffset + k;
    const float4 bmin = blasNodes[2 * src], bmax = blasNodes[2 * src + 1];
    const bool isLeaf = __float_as_uint(bmax.w) != 0u;
    unodes[2 * (baseB + src)] = make_float4(bmin.x, bmin.y, bmin.z, __uint_as_float(__float_as_uint(bmin.w) + (isLeaf ? (uint32_t)d.TriangleOffset : baseB + (uint32_t)d.NodeOffset)));
    unodes[2 * (baseB + src) + 1] = bmax;
}

// DScene::pairNodes: every sibling pair of every BLAS with its fields regrouped for 2-wide arithmetic (k_trace2 FAST).  Chunk k of 256 nodes belongs to BLAS chunks[k].x and starts
// at its node chunks[k].y (k_mark_triangles' table); a thread converts the pair that starts at its (even) node.  Plain copies: no arithmetic.
__global__ __launch_bounds__(256) void k_pair_nodes(const float4* blasNodes, const GpuBlasDesc* descs, const uint2* chunks, float4* out)
{
    const uint2 ch = chunks[blockIdx.x];
    const GpuBlasDesc d = descs[ch.x];
    const uint32_t k = ch.y + threadIdx.x;
    if ((k & 1u) != 0u || k + 1u >= (uint32_t)d.NodeCount) return;
    const size_t at = 2 * ((size_t)d.NodeOffset + k);
    const float4 lmin = blasNodes[at], lmax = blasNodes[at + 1], rmin = blasNodes[at + 2], rmax = blasNodes[at + 3];
    out[at] = make_float4(lmin.x, lmin.y, rmin.x, rmin.y);
    out[at + 1] = make_float4(lmax.x, lmax.y, rmax.x, rmax.y);
    out[at + 2] = make_float4(lmin.z, rmin.z, lmax.z, rmax.z);
    out[at + 3] = make_float4(lmin.w, lmax.w, rmin.w, rmax.w);
}

// BLAS refit (Shaders/BLASRefit/compute.glsl).  The reference walks leaf->root inside one dispatch behind an
// atomicExchange "second arrival" lock; here the same unions are evaluated level by level (deepest first), one launch
// per level, so no workgroup ever consumes another workgroup's stores inside a launch (per-XCD L2s are not coherent).
// (src: the node array being replaced — topology words; nodes: the array being written — the same array unless the refit goes to another scene-version slot)
__global__ void k_refit_leaves(const float4* src, float4* nodes, const uint4* tris, const float4* triVerts, const int32_t* leafIds, uint32_t leafCount, uint32_t nodeOffset, uint32_t triOffset)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= leafCount) return;
    uint32_t id = nodeOffset + (uint32_t)leafIds[i];
    float4 mn = src[2 * (size_t)id], mx = src[2 * (size_t)id + 1];
    uint32_t start = triOffset + __float_as_uint(mn.w), count = __float_as_uint(mx.w);
    f3 bmin = splat3(PT_FLOAT_MAX), bmax = splat3(-PT_FLOAT_MAX);
    for (uint32_t k = start; k < start + count; k++) {
        for (int v = 0; v < 3; v++) { float4 p = triVerts[3 * (size_t)k + v]; bmin = mk3(gmin(bmin.x, p.x), gmin(bmin.y, p.y), gmin(bmin.z, p.z)); bmax = mk3(gmax(bmax.x, p.x), gmax(bmax.y, p.y), gmax(bmax.z, p.z)); }
    }
    nodes[2 * (size_t)id] = make_float4(bmin.x, bmin.y, bmin.z, mn.w); nodes[2 * (size_t)id + 1] = make_float4(bmax.x, bmax.y, bmax.z, mx.w);
}
__global__ void k_refit_level(const float4* src, float4* nodes, const int32_t* levelNodes, uint32_t count, uint32_t nodeOffset)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t id = nodeOffset + (uint32_t)levelNodes[i];
    float4 mn = src[2 * (size_t)id], mx = src[2 * (size_t)id + 1];
    uint32_t child = nodeOffset + __float_as_uint(mn.w);
    float4 lmn = nodes[2 * (size_t)child], lmx = nodes[2 * (size_t)child + 1], rmn = nodes[2 * (size_t)child + 2], rmx = nodes[2 * (size_t)child + 3];
    nodes[2 * (size_t)id] = make_float4(gmin(lmn.x, rmn.x), gmin(lmn.y, rmn.y), gmin(lmn.z, rmn.z), mn.w);
    nodes[2 * (size_t)id + 1] = make_float4(gmax(lmx.x, rmx.x), gmax(lmx.y, rmx.y), gmax(lmx.z, rmx.z), mx.w);
}

// Skinning (Shaders/Skinning/compute.glsl:14-47): 4-weight linear blend; joint matrices are row_major mat4x3 (3 x float4)
// (verticesIn: the vertex array being replaced — the uv words; vertices: the one being written; the same array unless the skinning goes to another scene-version slot)
__global__ void k_skin(const GpuUnskinnedVertex* unskinned, const float4* joints, float* positions, float* prevPositions, const uint4* verticesIn, uint4* vertices,
                       uint32_t inOff, uint32_t outOff, uint32_t jointOff, uint32_t count)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    GpuUnskinnedVertex u = unskinned[inOff + i];
    float4 m[3];
    for (int r = 0; r < 3; r++) {
        float4 a = joints[3 * (size_t)(jointOff + u.JointIndices[0]) + r], b = joints[3 * (size_t)(jointOff + u.JointIndices[1]) + r];
        float4 c = joints[3 * (size_t)(jointOff + u.JointIndices[2]) + r], d = joints[3 * (size_t)(jointOff + u.JointIndices[3]) + r];
        float w0 = u.JointWeights[0], w1 = u.JointWeights[1], w2 = u.JointWeights[2], w3 = u.JointWeights[3];
        m[r] = make_float4(((w0 * a.x + w1 * b.x) + w2 * c.x) + w3 * d.x, ((w0 * a.y + w1 * b.y) + w2 * c.y) + w3 * d.y,
                           ((w0 * a.z + w1 * b.z) + w2 * c.z) + w3 * d.z, ((w0 * a.w + w1 * b.w) + w2 * c.w) + w3 * d.w);
    }
    M34 M; M.r0 = m[0]; M.r1 = m[1]; M.r2 = m[2];
    f3 p = mk3(u.Position[0], u.Position[1], u.Position[2]);
    f3 n = DecompressSR11G11B10(u.Normal), t = DecompressSR11G11B10(u.Tangent);
    f3 np = xform34(M, p, 1.0f);
    // mat3(skinMatrix) * v : out_i = (R[i][0]*v.x + R[i][1]*v.y) + R[i][2]*v.z
    f3 nn = normalize(mk3((M.r0.x * n.x + M.r0.y * n.y) + M.r0.z * n.z, (M.r1.x * n.x + M.r1.y * n.y) + M.r1.z * n.z, (M.r2.x * n.x + M.r2.y * n.y) + M.r2.z * n.z));
    f3 nt = normalize(mk3((M.r0.x * t.x + M.r0.y * t.y) + M.r0.z * t.z, (M.r1.x * t.x + M.r1.y * t.y) + M.r1.z * t.z, (M.r2.x * t.x + M.r2.y * t.y) + M.r2.z * t.z));
    size_t o = (size_t)(outOff + i);
    if (prevPositions) { prevPositions[3 * o] = positions[3 * o]; prevPositions[3 * o + 1] = positions[3 * o + 1]; prevPositions[3 * o + 2] = positions[3 * o + 2]; }
    positions[3 * o] = np.x; positions[3 * o + 1] = np.y; positions[3 * o + 2] = np.z;
    uint4 v = verticesIn[o]; v.w = CompressSR11G11B10(nn); v.z = CompressSR11G11B10(nt); vertices[o] = v;
}
