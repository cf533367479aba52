// bvh_gpu.hpp — idkptBuildBlasCore: the SweepSAH recursion of the reference's BLAS builder on the GPU.
// Part of the single translation unit idkpt.hip.  Host-side companion: libidkbvh.so's idkbvhBlasBegin / idkbvhBlasCoreSet / idkbvhBlasFinish.
//
// What is computed: BLAS.GetBuildData (Bvh/BLAS.cs:128-157: fragment ids sorted per axis by FloatToKey(min + max), stable) and
// BLAS.Build's recursion (Bvh/BLAS.cs:159-243, TrySplit :730-873) — for every node the bounds, the best SAH split over three axes, the
// swap rule (smaller half-area first), the stable partition of the three id arrays, and the children at the ids the reference reserves
// from the fragment counts (:221-241).  The output is the same bytes libidkbvh's CPU core produces (tests/test_gpu_builder.py).
//
// How.  Level-synchronous: all nodes of a level are processed by the same launches; a node's fragment range is cut into chunks of 256
// positions, one workgroup per chunk, so the grid does not depend on the tree shape.
//   * TrySplit prunes its sweeps against the best cost found so far; because the partial costs HalfArea(box) * count are monotone along a
//     sweep, every position it skips costs at least the running best, so its result is the first minimum of the cost over (axis, index)
//     in that order.  That is evaluated here in full: chunk boxes -> per-node carries (boxes of the chunks before / after) -> in-chunk
//     suffix scan = right costs, in-chunk prefix scan = left costs, cost = left + right exactly as BLAS.cs:760-821 (binary32, fused
//     HalfArea, counts as exact floats) -> per-chunk first minimum -> per-node first minimum in (axis, chunk) order.
//   * Box unions follow Vector128.MinNative/MaxNative (SSE minps/maxps: on equal operands the SECOND one is returned), applied in the
//     reference's order: the bounds of a node are accumulated along its x-sorted range, so signed zeros come out identically.
//   * The three stable partitions are flag counts per chunk, offsets per node, scatter.
// Everything else of the build (PreSplit priorities use cbrt, whose last bit is the C library's; the stack-size optimisation and the
// un-indexing are order-dependent tree walks) stays in libidkbvh.
#pragma once

namespace bvhgpu {

constexpr int CH = 256;                 // positions per chunk = threads per workgroup
struct BBox { float mn[3], mx[3]; };
struct HNodeG { float mn[3]; int32_t startOrChild; float mx[3]; int32_t count; };
static_assert(sizeof(HNodeG) == 32, "node layout");

DEV BBox box_empty() { BBox b; for (int k = 0; k < 3; k++) { b.mn[k] = 3.402823466e+38f; b.mx[k] = -3.402823466e+38f; } return b; }
// minps / maxps semantics: (a < b) ? a : b — equal (and unordered) operands give the second
DEV float sse_min(float a, float b) { return a < b ? a : b; }
DEV float sse_max(float a, float b) { return a > b ? a : b; }
// prefix direction (BLAS.cs left sweep, boundsOf): acc = min(acc, next) — ties go to the later (right) operand
DEV BBox join_lr(const BBox& l, const BBox& r) { BBox o; for (int k = 0; k < 3; k++) { o.mn[k] = sse_min(l.mn[k], r.mn[k]); o.mx[k] = sse_max(l.mx[k], r.mx[k]); } return o; }
// suffix direction (right sweep runs from the end towards the start): acc = min(acc, previous) — ties go to the earlier (left) operand
DEV BBox join_rl(const BBox& l, const BBox& r) { BBox o; for (int k = 0; k < 3; k++) { o.mn[k] = sse_min(r.mn[k], l.mn[k]); o.mx[k] = sse_max(r.mx[k], l.mx[k]); } return o; }
DEV float half_area(const BBox& b) { float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2]; return __builtin_fmaf(x + y, z, x * y); }   // MyMath.HalfArea (Utils/MyMath.cs:222-229)
DEV BBox frag_box(const float4* fb, int id) { float4 a = fb[2 * (size_t)id], b = fb[2 * (size_t)id + 1];   // libidkbvh's fragment: min.xyz, pad, max.xyz, pad
    BBox o; o.mn[0] = a.x; o.mn[1] = a.y; o.mn[2] = a.z; o.mx[0] = b.x; o.mx[1] = b.y; o.mx[2] = b.z; return o; }
DEV uint32_t float_to_key(float v) { uint32_t f = __float_as_uint(v); return f ^ (uint32_t)(((int32_t)f >> 31) | (int32_t)0x80000000); }   // Algorithms.cs:15-34

struct Level {                          // per-level tables (device pointers)
    const int* act; int A;              // active node ids
    int* nodeChunk0;                    // [A + 1] first chunk of every active node
    int* chunkNode; int* chunkBegin;    // [C] active index / first position of every chunk
    int* chunkCount;                    // C
};

__global__ void k_keys(const float4* fb, int n, int axis, uint32_t* keys, uint32_t* vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = fb[2 * (size_t)i], b = fb[2 * (size_t)i + 1];
    const float lo = axis == 0 ? a.x : axis == 1 ? a.y : a.z, hi = axis == 0 ? b.x : axis == 1 ? b.y : b.z;
    keys[i] = float_to_key(lo + hi); vals[i] = (uint32_t)i;
}

// chunk table of a level: node a (count c) owns ceil(c / CH) chunks (at least one).  One workgroup; A can be large, so a two-pass block scan.
__global__ __launch_bounds__(1024) void k_chunks(const HNodeG* nodes, Level L)
{
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (L.A + 1023) / 1024;
    const int b = min(t * per, L.A), e = min(b + per, L.A);
    int sum = 0;
    for (int a = b; a < e; a++) sum += max(1, (nodes[L.act[a]].count + CH - 1) / CH);
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { int v = t >= off ? part[t - off] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - sum;
    for (int a = b; a < e; a++) {
        const HNodeG nd = nodes[L.act[a]];
        const int c = max(1, (nd.count + CH - 1) / CH);
        L.nodeChunk0[a] = run;
        for (int k = 0; k < c; k++) { L.chunkNode[run + k] = a; L.chunkBegin[run + k] = nd.startOrChild + k * CH; }
        run += c;
    }
    if (t == 1023) { L.nodeChunk0[L.A] = part[1023]; *L.chunkCount = part[1023]; }
}

// in-order reduction of the boxes of one chunk (256 positions of ids[axis]); `rl` selects the tie rule of the direction
template <bool RL>
DEV BBox block_reduce(BBox v, bool valid, BBox* sh)
{
    const int t = threadIdx.x;
    sh[t] = valid ? v : box_empty();
    __syncthreads();
    for (int s = 1; s < CH; s <<= 1) {
        BBox mine = sh[t];
        const bool doit = (t % (2 * s)) == 0 && t + s < CH;
        BBox other = doit ? sh[t + s] : mine;
        __syncthreads();
        if (doit) sh[t] = RL ? join_rl(mine, other) : join_lr(mine, other);
        __syncthreads();
    }
    BBox r = sh[0];
    __syncthreads();
    return r;
}

// union of every chunk along every axis (grid.y = axis).  Empty padding never wins a comparison, so it cannot disturb the tie rules.
__global__ __launch_bounds__(CH) void k_chunk_box(const HNodeG* nodes, Level L, const float4* fb, const int* ids0, const int* ids1, const int* ids2, BBox* cboxL, BBox* cboxR, int Cmax)
{
    __shared__ BBox sh[CH];
    const int k = blockIdx.x, axis = blockIdx.y;
    if (k >= *L.chunkCount) return;
    const HNodeG nd = nodes[L.act[L.chunkNode[k]]];
    const int end = nd.startOrChild + nd.count, pos = L.chunkBegin[k] + threadIdx.x;
    const bool valid = pos < end;
    const int* ids = axis == 0 ? ids0 : axis == 1 ? ids1 : ids2;
    BBox v = valid ? frag_box(fb, ids[pos]) : box_empty();
    BBox l = block_reduce<false>(v, valid, sh);
    BBox r = block_reduce<true>(v, valid, sh);
    if (threadIdx.x == 0) { cboxL[(size_t)axis * Cmax + k] = l; cboxR[(size_t)axis * Cmax + k] = r; }
}

// inclusive in-order scan of the chunk's boxes (Hillis-Steele over shared memory).  RL: from the right end (suffix), else from the left (prefix).
template <bool RL>
DEV BBox block_scan(BBox v, BBox* sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = 1; s < CH; s <<= 1) {
        BBox mine = sh[t];
        BBox other = RL ? (t + s < CH ? sh[t + s] : box_empty()) : (t >= s ? sh[t - s] : box_empty());
        const bool has = RL ? (t + s < CH) : (t >= s);
        __syncthreads();
        if (has) sh[t] = RL ? join_rl(mine, other) : join_lr(other, mine);
        __syncthreads();
    }
    BBox r = sh[t];
    __syncthreads();
    return r;
}

// per node (one workgroup): boxes of the chunks before (prefix rule) / after (suffix rule) every chunk, per axis; the node's bounds (x-sorted order)
__global__ __launch_bounds__(CH) void k_node_carry(HNodeG* nodes, Level L, const BBox* cboxL, const BBox* cboxR, BBox* carryL, BBox* carryR, int Cmax)
{
    __shared__ BBox sh[CH];
    __shared__ BBox shPrev[CH];
    const int a = blockIdx.x, t = threadIdx.x;
    if (a >= L.A) return;
    const int c0 = L.nodeChunk0[a], c1 = L.nodeChunk0[a + 1];
    for (int axis = 0; axis < 3; axis++) {
        const size_t o = (size_t)axis * Cmax;
        BBox run = box_empty();                                                  // union of the tiles already done
        for (int base = c0; base < c1; base += CH) {                             // prefix direction
            const int k = base + t; const bool valid = k < c1;
            BBox incl = block_scan<false>(valid ? cboxL[o + k] : box_empty(), sh);
            shPrev[t] = incl;
            __syncthreads();
            if (valid) carryL[o + k] = t == 0 ? run : join_lr(run, shPrev[t - 1]);
            const BBox tileAll = shPrev[CH - 1];
            __syncthreads();
            run = join_lr(run, tileAll);
        }
        if (axis == 0 && t == 0) { HNodeG& nd = nodes[L.act[a]]; for (int d = 0; d < 3; d++) { nd.mn[d] = run.mn[d]; nd.mx[d] = run.mx[d]; } }   // setBounds(boundsOf(.., axis 0)), BLAS.cs:206
        run = box_empty();
        const int tiles = (c1 - c0 + CH - 1) / CH;
        for (int tile = tiles - 1; tile >= 0; tile--) {                          // suffix direction
            const int base = c0 + tile * CH, k = base + t; const bool valid = k < c1;
            BBox incl = block_scan<true>(valid ? cboxR[o + k] : box_empty(), sh);
            shPrev[t] = incl;
            __syncthreads();
            if (valid) carryR[o + k] = t == CH - 1 ? run : join_rl(shPrev[t + 1], run);
            const BBox tileAll = shPrev[0];
            __syncthreads();
            run = join_rl(tileAll, run);
        }
    }
}

// right costs: rc[axis][pos] = HalfArea(union of [pos, end)) * (end - pos)   (BLAS.cs:768-782)
__global__ __launch_bounds__(CH) void k_chunk_rc(const HNodeG* nodes, Level L, const float4* fb, const int* ids0, const int* ids1, const int* ids2, const BBox* carryR, float* rc, int n, int Cmax)
{
    __shared__ BBox sh[CH];
    const int k = blockIdx.x, axis = blockIdx.y;
    if (k >= *L.chunkCount) return;
    const HNodeG nd = nodes[L.act[L.chunkNode[k]]];
    const int end = nd.startOrChild + nd.count, pos = L.chunkBegin[k] + threadIdx.x;
    const bool valid = pos < end;
    const int* ids = axis == 0 ? ids0 : axis == 1 ? ids1 : ids2;
    BBox v = valid ? frag_box(fb, ids[pos]) : box_empty();
    BBox s = block_scan<true>(v, sh);
    if (valid) {
        s = join_rl(s, carryR[(size_t)axis * Cmax + k]);
        rc[(size_t)axis * n + pos] = half_area(s) * (float)(end - pos);
    }
}

// left costs + total cost + first minimum of the chunk   (BLAS.cs:784-806)
__global__ __launch_bounds__(CH) void k_chunk_cost(const HNodeG* nodes, Level L, const float4* fb, const int* ids0, const int* ids1, const int* ids2, const BBox* carryL, const float* rc, float* cbestCost, int* cbestPos, int n, int Cmax)
{
    __shared__ BBox sh[CH];
    __shared__ float shc[CH]; __shared__ int shp[CH];
    const int k = blockIdx.x, axis = blockIdx.y, t = threadIdx.x;
    if (k >= *L.chunkCount) return;
    const HNodeG nd = nodes[L.act[L.chunkNode[k]]];
    const int start = nd.startOrChild, end = start + nd.count, pos = L.chunkBegin[k] + t;
    const bool valid = pos < end;
    const int* ids = axis == 0 ? ids0 : axis == 1 ? ids1 : ids2;
    BBox v = valid ? frag_box(fb, ids[pos]) : box_empty();
    BBox p = block_scan<false>(v, sh);
    float cost = 3.402823466e+38f; int cpos = 0x7fffffff;
    if (valid && pos < end - 1) {
        p = join_lr(carryL[(size_t)axis * Cmax + k], p);
        const float lcost = half_area(p) * (float)(pos - start + 1);
        cost = lcost + rc[(size_t)axis * n + pos + 1];
        cpos = pos + 1;                                        // the split index: first position of the right part
        // the CPU sweep only ever accepts `cost < bestCost` with bestCost <= FLT_MAX: a NaN (inf - inf of non-finite fragment boxes), +inf or FLT_MAX
        // cost is never chosen there.  Here a NaN sitting in a reduction slot would never be displaced (c2 < NaN is false): map all of them to "invalid".
        if (!(cost < 3.402823466e+38f)) { cost = 3.402823466e+38f; cpos = 0x7fffffff; }
    }
    shc[t] = cost; shp[t] = cpos;
    __syncthreads();
    for (int s = CH / 2; s > 0; s >>= 1) {                     // first minimum: on equal costs the lower position wins (a slot may already hold a higher position)
        if (t < s) { const float c2 = shc[t + s]; const int p2 = shp[t + s]; if (c2 < shc[t] || (c2 == shc[t] && p2 < shp[t])) { shc[t] = c2; shp[t] = p2; } }
        __syncthreads();
    }
    if (t == 0) { cbestCost[(size_t)axis * Cmax + k] = shc[0]; cbestPos[(size_t)axis * Cmax + k] = shp[0]; }
}

struct Decision { int split; int axis; int index; };           // split: 0 = leaf

// TrySplit's verdict per node (BLAS.cs:733-736, 806-829); one workgroup per node reduces the chunk minima in (axis, position) order
__global__ __launch_bounds__(CH) void k_node_decide(const HNodeG* nodes, Level L, const float* cbestCost, const int* cbestPos, Decision* dec, int Cmax)
{
    __shared__ float shc[CH]; __shared__ int shp[CH];
    const int a = blockIdx.x, t = threadIdx.x;
    if (a >= L.A) return;
    const HNodeG nd = nodes[L.act[a]];
    Decision d = {0, 0, 0};
    if (nd.count > 1) {                                        // kStopSplittingThreshold = 1
        float best = 3.402823466e+38f; int bi = 0, ba = 0;
        const int c0 = L.nodeChunk0[a], c1 = L.nodeChunk0[a + 1];
        for (int axis = 0; axis < 3; axis++) {
            float c = 3.402823466e+38f; int p = 0x7fffffff;
            for (int k = c0 + t; k < c1; k += CH) { const float c2 = cbestCost[(size_t)axis * Cmax + k]; const int p2 = cbestPos[(size_t)axis * Cmax + k]; if (c2 < c || (c2 == c && p2 < p)) { c = c2; p = p2; } }
            shc[t] = c; shp[t] = p;
            __syncthreads();
            for (int s2 = CH / 2; s2 > 0; s2 >>= 1) {
                if (t < s2) { const float c2 = shc[t + s2]; const int p2 = shp[t + s2]; if (c2 < shc[t] || (c2 == shc[t] && p2 < shp[t])) { shc[t] = c2; shp[t] = p2; } }
                __syncthreads();
            }
            if (shc[0] < best) { best = shc[0]; bi = shp[0]; ba = axis; }
            __syncthreads();
        }
        d.split = 1; d.axis = ba; d.index = bi;
        if (nd.count <= 2) {                                   // kMaxLeafTriangleCount = 2: a leaf unless splitting is cheaper
            const float x = nd.mx[0] - nd.mn[0], y = nd.mx[1] - nd.mn[1], z = nd.mx[2] - nd.mn[2];
            const float parentHalfArea = __builtin_fmaf(x + y, z, x * y);
            const float notSplit = 1.1f * (float)nd.count;
            const float newCost = 1.0f + (1.1f * best / parentHalfArea);
            if (newCost >= notSplit) d.split = 0;
        }
    }
    if (t == 0) dec[a] = d;
}

// boxes of the two sides of the chosen split, per chunk of the chosen axis (BLAS.cs:831-833: boundsOf both halves)
__global__ __launch_bounds__(CH) void k_chunk_sides(const HNodeG* nodes, Level L, const Decision* dec, const float4* fb, const int* ids0, const int* ids1, const int* ids2, BBox* sideL, BBox* sideR)
{
    __shared__ BBox sh[CH];
    const int k = blockIdx.x;
    if (k >= *L.chunkCount) return;
    const int a = L.chunkNode[k];
    const Decision d = dec[a];
    if (!d.split) return;
    const HNodeG nd = nodes[L.act[a]];
    const int end = nd.startOrChild + nd.count, pos = L.chunkBegin[k] + threadIdx.x;
    const bool valid = pos < end;
    const int* ids = d.axis == 0 ? ids0 : d.axis == 1 ? ids1 : ids2;
    BBox v = valid ? frag_box(fb, ids[pos]) : box_empty();
    BBox l = block_reduce<false>(v, valid && pos < d.index, sh);
    BBox r = block_reduce<false>(v, valid && pos >= d.index, sh);
    if (threadIdx.x == 0) { sideL[k] = l; sideR[k] = r; }
}

// swap rule, children, next level's list (BLAS.cs:208-241, 833-835)
__global__ __launch_bounds__(CH) void k_node_finalize(HNodeG* nodes, Level L, Decision* dec, const BBox* sideL, const BBox* sideR, int* freshOf, int* swapOf, int* leftCountOf, int* nextAct, int* nextCount, int* smallList, int* smallCount, int smallMax)
{
    __shared__ BBox sh[CH];
    const int a = blockIdx.x, t = threadIdx.x;
    if (a >= L.A) return;
    const Decision d = dec[a];
    if (!d.split) return;
    const int v = L.act[a];
    HNodeG nd = nodes[v];
    BBox lb = box_empty(), rb = box_empty();
    {   // in-order union over the node's chunks: every thread folds a contiguous run, the runs are then folded in order
        const int c0 = L.nodeChunk0[a], c1 = L.nodeChunk0[a + 1], per = (c1 - c0 + CH - 1) / CH;
        const int b0 = min(c0 + t * per, c1), b1 = min(b0 + per, c1);
        BBox l = box_empty(), r = box_empty();
        for (int k = b0; k < b1; k++) { l = join_lr(l, sideL[k]); r = join_lr(r, sideR[k]); }
        lb = block_reduce<false>(l, true, sh); rb = block_reduce<false>(r, true, sh);
    }
    if (t != 0) return;
    const int start = nd.startOrChild, end = start + nd.count;
    const int swap = half_area(lb) < half_area(rb) ? 1 : 0;
    const int lcount = swap ? end - d.index : d.index - start;
    swapOf[a] = swap; leftCountOf[a] = lcount;
    const int lid = freshOf[v], rid = lid + 1;
    HNodeG l = {}, r = {};
    l.startOrChild = start; l.count = lcount;
    r.startOrChild = start + lcount; r.count = nd.count - lcount;
    nodes[lid] = l; nodes[rid] = r;
    freshOf[lid] = rid + 1; freshOf[rid] = rid + (2 * lcount - 1);
    nodes[v].startOrChild = lid; nodes[v].count = 0;
    // children with few fragments leave the level-synchronous scheme: one thread each finishes their subtree (k_small_subtrees)
    if (l.count <= smallMax) smallList[atomicAdd(smallCount, 1)] = lid; else nextAct[atomicAdd(nextCount, 1)] = lid;
    if (r.count <= smallMax) smallList[atomicAdd(smallCount, 1)] = rid; else nextAct[atomicAdd(nextCount, 1)] = rid;
}

// leftTable (BLAS.cs:837-846) along the chosen axis
__global__ __launch_bounds__(CH) void k_mark(const HNodeG* nodesBefore, Level L, const Decision* dec, const int* swapOf, const int* startOf, const int* countOf, const int* ids0, const int* ids1, const int* ids2, uint8_t* leftTable)
{
    const int k = blockIdx.x;
    if (k >= *L.chunkCount) return;
    const int a = L.chunkNode[k];
    const Decision d = dec[a];
    if (!d.split) return;
    const int end = startOf[a] + countOf[a], pos = L.chunkBegin[k] + threadIdx.x;
    if (pos >= end) return;
    const int* ids = d.axis == 0 ? ids0 : d.axis == 1 ? ids1 : ids2;
    leftTable[ids[pos]] = (uint8_t)((pos < d.index) != (swapOf[a] != 0));
    (void)nodesBefore;
}

// stable partition, step 1: flagged positions per chunk and axis
__global__ __launch_bounds__(CH) void k_part_count(Level L, const Decision* dec, const int* startOf, const int* countOf, const int* ids0, const int* ids1, const int* ids2, const uint8_t* leftTable, int* pcnt, int Cmax)
{
    __shared__ int sh[CH];
    const int k = blockIdx.x, axis = blockIdx.y, t = threadIdx.x;
    if (k >= *L.chunkCount) return;
    const int a = L.chunkNode[k];
    if (!dec[a].split) return;
    const int end = startOf[a] + countOf[a], pos = L.chunkBegin[k] + t;
    const int* ids = axis == 0 ? ids0 : axis == 1 ? ids1 : ids2;
    sh[t] = (pos < end && leftTable[ids[pos]]) ? 1 : 0;
    __syncthreads();
    for (int s = CH / 2; s > 0; s >>= 1) { if (t < s) sh[t] += sh[t + s]; __syncthreads(); }
    if (t == 0) pcnt[(size_t)axis * Cmax + k] = sh[0];
}
// step 2: flagged positions before every chunk, per node and axis (one workgroup per node)
__global__ __launch_bounds__(CH) void k_part_offsets(Level L, const Decision* dec, const int* pcnt, int* poff, int Cmax)
{
    __shared__ int sh[CH];
    const int a = blockIdx.x, t = threadIdx.x;
    if (a >= L.A || !dec[a].split) return;
    const int c0 = L.nodeChunk0[a], c1 = L.nodeChunk0[a + 1];
    for (int axis = 0; axis < 3; axis++) {
        int run = 0;
        for (int base = c0; base < c1; base += CH) {
            const int k = base + t; const int v = k < c1 ? pcnt[(size_t)axis * Cmax + k] : 0;
            sh[t] = v;
            __syncthreads();
            for (int s2 = 1; s2 < CH; s2 <<= 1) { int w = t >= s2 ? sh[t - s2] : 0; __syncthreads(); sh[t] += w; __syncthreads(); }
            if (k < c1) poff[(size_t)axis * Cmax + k] = run + sh[t] - v;
            const int tileAll = sh[CH - 1];
            __syncthreads();
            run += tileAll;
        }
    }
}
// step 3: scatter (stable on both sides).  `out` was pre-filled with a copy of `in`, so ranges of nodes that do not split stay as they are.
__global__ __launch_bounds__(CH) void k_part_scatter(Level L, const Decision* dec, const int* startOf, const int* countOf, const int* leftCountOf, const int* in0, const int* in1, const int* in2, int* out0, int* out1, int* out2,
                                                    const uint8_t* leftTable, const int* poff, int Cmax)
{
    __shared__ int sh[CH];
    const int k = blockIdx.x, axis = blockIdx.y, t = threadIdx.x;
    if (k >= *L.chunkCount) return;
    const int a = L.chunkNode[k];
    if (!dec[a].split) return;
    const int start = startOf[a], end = start + countOf[a], cb = L.chunkBegin[k], pos = cb + t;
    const int* in = axis == 0 ? in0 : axis == 1 ? in1 : in2;
    int* out = axis == 0 ? out0 : axis == 1 ? out1 : out2;
    const bool valid = pos < end;
    const int id = valid ? in[pos] : 0;
    const int flag = (valid && leftTable[id]) ? 1 : 0;
    sh[t] = flag;
    __syncthreads();
    for (int s = 1; s < CH; s <<= 1) { int v = t >= s ? sh[t - s] : 0; __syncthreads(); sh[t] += v; __syncthreads(); }
    const int incl = sh[t], rank = incl - flag;                 // flagged positions of this chunk before t
    if (!valid) return;
    const int before = poff[(size_t)axis * Cmax + k];           // flagged positions of the node before this chunk
    const int dst = flag ? start + before + rank : start + leftCountOf[a] + ((cb - start) - before) + (t - rank);
    out[dst] = id;
}

__global__ void k_snapshot_ranges(const HNodeG* nodes, Level L, int* startOf, int* countOf)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= L.A) return;
    const HNodeG nd = nodes[L.act[a]];
    startOf[a] = nd.startOrChild; countOf[a] = nd.count;
}

// ---- subtrees of at most `smallMax` fragments: BLAS.Build / TrySplit as the reference runs them (Bvh/BLAS.cs:193-243, 730-873), one thread per subtree.
// The id arrays are partitioned in place (positions of different subtrees are disjoint), `rc` / `aux` are per-position scratch.
DEV void grow(BBox& a, const BBox& x) { for (int k = 0; k < 3; k++) { a.mn[k] = sse_min(a.mn[k], x.mn[k]); a.mx[k] = sse_max(a.mx[k], x.mx[k]); } }   // Box.GrowToFit: minps / maxps (acc, x)
DEV BBox bounds_of(const float4* fb, const int* ids, int start, int count) { BBox b = box_empty(); for (int i = 0; i < count; i++) grow(b, frag_box(fb, ids[start + i])); return b; }
DEV int stable_partition(int* src, int n, int* aux, const uint8_t* table)
{
    int l = 0, r = 0;
    for (int i = 0; i < n; i++) { const int id = src[i]; if (table[id]) src[l++] = id; else aux[r++] = id; }
    for (int i = 0; i < r; i++) src[l + i] = aux[i];
    return l;
}
__global__ void k_small_subtrees(HNodeG* nodes, const int* smallList, const int* smallCount, const float4* fb, int* ids0, int* ids1, int* ids2, float* rc, int* aux, uint8_t* leftTable, const int* freshOf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *smallCount) return;
    int* idsA[3] = {ids0, ids1, ids2};
    int stackNode[136], stackFresh[136]; int sp = 0;              // a subtree over <= smallMax (<= 128) fragments never holds more entries than it has fragments
    stackNode[0] = smallList[i]; stackFresh[0] = freshOf[smallList[i]]; sp = 1;
    while (sp > 0) {
        sp--;
        const int pid = stackNode[sp], fresh = stackFresh[sp];
        HNodeG p = nodes[pid];
        const int start = p.startOrChild, count = p.count, end = start + count;
        { const BBox b = bounds_of(fb, ids0, start, count); for (int d = 0; d < 3; d++) { p.mn[d] = b.mn[d]; p.mx[d] = b.mx[d]; } nodes[pid] = p; }
        if (count <= 1) continue;
        int bestAxis = 0, bestIndex = 0; float bestCost = 3.402823466e+38f;
        for (int axis = 0; axis < 3; axis++) {
            const int* ids = idsA[axis];
            int firstRight = start + 1;
            BBox acc = box_empty(); float cnt = 0.0f;
            for (int k = end - 1; k >= firstRight; k--) {
                cnt += 1.0f; grow(acc, frag_box(fb, ids[k]));
                const float c = half_area(acc) * cnt;
                rc[k] = c;
                if (c >= bestCost) { firstRight = k + 1; break; }
            }
            BBox lacc = box_empty(); float lcnt = (float)(firstRight - start) - 1.0f;
            for (int k = start; k < firstRight - 1; k++) grow(lacc, frag_box(fb, ids[k]));
            for (int k = firstRight - 1; k < end - 1; k++) {
                lcnt += 1.0f; grow(lacc, frag_box(fb, ids[k]));
                const float lcost = half_area(lacc) * lcnt;
                const float cost = lcost + rc[k + 1];
                if (cost < bestCost) { bestIndex = k + 1; bestAxis = axis; bestCost = cost; }
                else if (lcost >= bestCost) break;
            }
        }
        if (count <= 2) {
            const float x = p.mx[0] - p.mn[0], y = p.mx[1] - p.mn[1], z = p.mx[2] - p.mn[2];
            const float notSplit = 1.1f * (float)count;
            const float newCost = 1.0f + (1.1f * bestCost / __builtin_fmaf(x + y, z, x * y));
            if (newCost >= notSplit) continue;
        }
        const BBox lb = bounds_of(fb, idsA[bestAxis], start, bestIndex - start), rb = bounds_of(fb, idsA[bestAxis], bestIndex, end - bestIndex);
        const bool swap = half_area(lb) < half_area(rb);
        int* ids = idsA[bestAxis];
        for (int k = start; k < bestIndex; k++) leftTable[ids[k]] = (uint8_t)!swap;
        for (int k = bestIndex; k < end; k++) leftTable[ids[k]] = (uint8_t)swap;
        if (swap) bestIndex = start + stable_partition(ids + start, count, aux + start, leftTable);
        stable_partition(idsA[(bestAxis + 1) % 3] + start, count, aux + start, leftTable);
        stable_partition(idsA[(bestAxis + 2) % 3] + start, count, aux + start, leftTable);
        HNodeG l = {}, r = {};
        l.startOrChild = start; l.count = bestIndex - start;
        r.startOrChild = bestIndex; r.count = count - l.count;
        const int lid = fresh, rid = lid + 1;
        nodes[lid] = l; nodes[rid] = r;
        p.startOrChild = lid; p.count = 0; nodes[pid] = p;
        stackNode[sp] = rid; stackFresh[sp] = rid + (2 * l.count - 1); sp++;
        stackNode[sp] = lid; stackFresh[sp] = rid + 1; sp++;
    }
}

} // namespace bvhgpu
