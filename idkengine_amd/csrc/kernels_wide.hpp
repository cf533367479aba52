// kernels_wide.hpp — the wide-node closest-hit traversal (k_trace_wide) and the kernels that derive its structure on the device (k_wide_topo, k_wide_fill).
// Layout, build rules and the exactness argument: wide_nodes.hpp.  Part of the single translation unit idkpt.hip (included there, in this order).
//
// Why: k_trace2 walks the reference's BVH2 one 64-byte sibling pair per dependent memory round trip, ~63 of them per ray on the headline scene at ~1 700 cycles each
// with nothing on the chip saturated (profiles/r04_phase_profile.txt): the launch time is round trips x latency.  A wide node is one 64-byte block for up to four
// BVH2 nodes: 0.57x the round trips at 0.60x the bytes (tools/wide_sim.cpp, every view), a third of the node footprint.
// What stays exact: wide_nodes.hpp.  Rays the walk cannot vouch for are appended to a list and traced by k_trace2 right behind this launch.
#pragma once
#include "wide_nodes.hpp"

struct WideBufs {
    const uint4* nodes;          // wide::Node[] of the BLAS being traversed (4 x uint4 each)
    const float4* leaves;        // its leaf records (16-byte units: BVH2 leaf node (2) + 3 per triangle)
    uint32_t* flagCount;         // this launch's count of flagged rays ...
    uint32_t* flagA;             // ... their ray ids (PRIMARY: = hit slots) / queue slots (bounce launches)
    uint32_t* flagB;             // ... bounce launches: their ray ids
    unsigned long long* totals;  // [0] flagged rays since idkptResetStats; COUNT: [1] wide node visits, [2] leaf records fetched, [3] triangle tests
    int cap;                     // rows of the per-lane stack
};

// ---- derivation -----------------------------------------------------------------------------------------------------------------------------------------------
// k_wide_topo: which BVH2 nodes make up every wide node of one BLAS, breadth first (one workgroup per BLAS; the order wide::build_host produces).  Writes, per wide node,
// the BVH2 ids of its children (ids), its child words and child count; box bytes and leaf records are k_wide_fill's (they change with a refit, the topology does not).
#define WIDE_TOPO_THREADS 1024
__global__ __launch_bounds__(WIDE_TOPO_THREADS) void k_wide_topo(const float4* nodes4, uint32_t nodeCount, uint32_t* pairOf, uint4* ids, uint4* wn, uint32_t* outCounts)
{
    const wide::Bvh2Node* nodes = (const wide::Bvh2Node*)nodes4;
    __shared__ uint32_t waveC[WIDE_TOPO_THREADS / 64], waveL[WIDE_TOPO_THREADS / 64];
    __shared__ uint32_t next, runLeaf, lo, hi;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (nodeCount < 4u) { if (tid == 0) { outCounts[0] = 0u; outCounts[1] = 0u; } return; }
    if (tid == 0) { pairOf[0] = 2u; lo = 0u; hi = 1u; next = 1u; runLeaf = 0u; }
    __syncthreads();
    while (lo < hi) {
        const uint32_t levelLo = lo, levelHi = hi;
        for (uint32_t base = levelLo; base < levelHi; base += WIDE_TOPO_THREADS) {
            const uint32_t i = base + tid; const bool valid = i < levelHi;
            uint32_t id[4] = {0u, 0u, 0u, 0u}; int n = 0; uint32_t ci = 0, li = 0;
            if (valid) {
                n = wide::expand_pair(nodes, pairOf[i], id);
                for (int k = 0; k < n; k++) { const uint32_t tc = nodes[id[k]].triCount; if (tc == 0u) ci++; else li += 2u + 3u * tc; }
            }
            // exclusive prefix sums of (ci, li) over the workgroup, in thread order
            uint32_t pc = ci, pl = li;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t a = __shfl_up(pc, off), b = __shfl_up(pl, off); if ((int)lane >= off) { pc += a; pl += b; } }
            if (lane == 63u) { waveC[wv] = pc; waveL[wv] = pl; }
            __syncthreads();
            uint32_t offC = 0, offL = 0, totC = 0, totL = 0;
            for (uint32_t w = 0; w < WIDE_TOPO_THREADS / 64; w++) { if (w < wv) { offC += waveC[w]; offL += waveL[w]; } totC += waveC[w]; totL += waveL[w]; }
            uint32_t childAt = next + offC + pc - ci, leafAt = runLeaf + offL + pl - li;
            if (valid) {
                uint32_t cw[4] = {0u, 0u, 0u, 0u};
                for (int k = 0; k < n; k++) {
                    const wide::Bvh2Node& c = nodes[id[k]];
                    if (c.triCount == 0u) { cw[k] = childAt; pairOf[childAt] = c.startOrChild; childAt++; }
                    else { cw[k] = wide::LEAF_BIT | leafAt; leafAt += 2u + 3u * c.triCount; }
                }
                ids[i] = make_uint4(id[0], id[1], id[2], id[3]);
                wn[4 * (size_t)i] = make_uint4(0u, 0u, 0u, (uint32_t)n << 24);
                wn[4 * (size_t)i + 2] = make_uint4(0u, 0u, cw[0], cw[1]);
                wn[4 * (size_t)i + 3] = make_uint4(cw[2], cw[3], 0u, 0u);
            }
            __syncthreads();
            if (tid == 0) { next += totC; runLeaf += totL; }
            __syncthreads();
        }
        if (tid == 0) { lo = levelHi; hi = next; }
        __syncthreads();
    }
    if (tid == 0) { outCounts[0] = hi; outCounts[1] = runLeaf; }
}

// k_wide_fill: the grid and the children's box bytes of every wide node from the CURRENT BVH2 boxes, and the leaf records (BVH2 leaf node + its triangles' positions)
// from the current nodes / triVerts.  One thread per wide node.  Runs after k_wide_topo and after everything that rewrites node boxes or positions (refit, skinning, patches).
__global__ __launch_bounds__(256) void k_wide_fill(const float4* nodes4, const float4* triVerts /* of this BLAS */, const uint4* ids, uint4* wn, float4* leaves, const uint32_t* counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counts[0]) return;
    const wide::Bvh2Node* nodes = (const wide::Bvh2Node*)nodes4;
    const uint4 idv = ids[i]; const uint32_t id[4] = {idv.x, idv.y, idv.z, idv.w};
    const uint4 c2 = wn[4 * (size_t)i + 2], c3 = wn[4 * (size_t)i + 3];
    const uint32_t cw[4] = {c2.z, c2.w, c3.x, c3.y};
    const int n = (int)(wn[4 * (size_t)i].w >> 24);
    wide::Node q;
    wide::quantize_node(nodes, id, n, &q);
    wn[4 * (size_t)i] = make_uint4(__float_as_uint(q.ox), __float_as_uint(q.oy), __float_as_uint(q.oz), q.exps);
    wn[4 * (size_t)i + 1] = make_uint4(q.qlo[0], q.qlo[1], q.qlo[2], q.qhi[0]);
    wn[4 * (size_t)i + 2] = make_uint4(q.qhi[1], q.qhi[2], cw[0], cw[1]);
    for (int k = 0; k < n; k++) {
        if (!(cw[k] & wide::LEAF_BIT)) continue;
        float4* dst = leaves + (cw[k] & ~wide::LEAF_BIT);
        const float4* src = nodes4 + 2 * (size_t)id[k];
        const float4 h0 = src[0], h1 = src[1];
        dst[0] = h0; dst[1] = h1;
        const uint32_t start = __float_as_uint(h0.w), cnt = __float_as_uint(h1.w);
        const float4* tv = triVerts + 3 * (size_t)start;
        const wide::Bvh2Node& leaf = nodes[id[k]];
        for (uint32_t t = 0; t < cnt; t++) {
            float4 a = tv[3 * t], b = tv[3 * t + 1], c = tv[3 * t + 2];
            const float v[12] = {a.x, a.y, a.z, 0.0f, b.x, b.y, b.z, 0.0f, c.x, c.y, c.z, 0.0f};
            a.w = __uint_as_float(wide::outside_leaf_box(leaf, v) ? 1u : 0u);          // MARKED: a PreSplit fragment (wide_nodes.hpp)
            dst[2 + 3 * t] = a; dst[3 + 3 * t] = b; dst[4 + 3 * t] = c;
        }
    }
}

// ---- traversal ------------------------------------------------------------------------------------------------------------------------------------------------
// One BLAS instance, closest hit (the MODE 0 work of k_trace2): persistent waves, every lane owns one ray and is refilled from the sliced work list exactly like k_trace2's.
// A lane holds its current entry — a wide node or a leaf record — in a register and the entries still to visit on its LDS stack ([depth][lane], one word each).
//   node phase: all lanes whose entry is a wide node fetch it (4 x 16 B), test its four children (wide::test_node's arithmetic, two children per packed instruction),
//               sort the hits by entry distance, continue with the nearest and push the others, far ones first; a lane whose entry became a leaf record waits;
//   leaf phase: lanes parked on a leaf record fetch its header and first triangle (5 x 16 B), run the reference's box test on the exact leaf box and, if it passes,
//               the reference's triangle tests in order; then they pop;
//   retire:     a lane without entries stores its hit — or, if the walk cannot vouch for it (wide_nodes.hpp: FLAGGED), appends the ray to the launch's re-trace list.
// COUNT: also counts node visits, leaf records and triangle tests (developer option wide_count: bench.py's algorithmic bytes).
#define WIDE_NONE 0x7fffffffu
template <bool PRIMARY, bool COUNT = false, int REFILL_MIN = 32, bool PROF = false>
__global__ __launch_bounds__(WAVE) void k_trace_wide(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter, WideBufs wb, uint64_t* counters)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // LDS rows of this wave, one word per lane: row 0 = dummy (holds WIDE_NONE: what a pop of the empty stack reads), rows 1 .. cap = entries, row cap + 1 = spare (takes the
    // stores of pushes that do not happen).  sp is the LDS address of the lane's top entry (== stkBase: empty): pop = read [sp], push = write above it — no index arithmetic, no branches.
    lds_u32* const stkBase = (lds_u32*)lds + lane;
    lds_u32* const stkTop = stkBase + wb.cap * WAVE;        // the last row an entry may occupy
    lds_u32* const spare = stkTop + WAVE;
    stkBase[0] = WIDE_NONE;
    const uint32_t N = *countPtr;
    {   // (the launch's grid may be larger than its ray count wants: k_trace2's rule)
        uint32_t want = gridDim.x;
        if (f.gridRaysX4 > 0u) want = max((uint32_t)(((unsigned long long)N * 4ull / f.gridRaysX4 + 63ull) / 64ull), min(want, 1024u));
        if (f.gridMid > 0u && N < f.gridMidRays) want = min(want, f.gridMid);
        if (blockIdx.x >= max(want, 1u)) return;
    }
    const GpuBlasInstance inst = s.instances[0];
    const uint32_t triOffset = (uint32_t)s.descs[inst.BlasId].TriangleOffset;
    bool active = false, workLeft = N != 0u;
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    uint32_t cur = WIDE_NONE, slot = 0, rayIdx = 0, flags = 0;
    lds_u32* sp = stkBase;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(1.0f);
    float hitT = 0.0f, cullT = 0.0f, hbx = 0.0f, hby = 0.0f, second = 0.0f, leafT1 = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;
    uint32_t nNodes = 0, nLeaves = 0, nTris = 0;
    unsigned long long pc[4] = {0, 0, 0, 0}, pn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tPrev = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
#define WPROF_MARK(b) do { if (PROF) { unsigned long long _t = __builtin_amdgcn_s_memtime(); pc[b] += _t - tPrev; tPrev = _t; } } while (0)

    while (true) {
        WPROF_MARK(3);
        // ---- retire finished rays: store the hit, or hand the ray to the exact kernel
        {
            const bool done = active && cur == WIDE_NONE;
            if (__builtin_amdgcn_ballot_w64(done) != 0ull) {
                bool flagged = false;
                if (done) {
                    const float win = hitT * wide::WINDOW;
                    flagged = flags != 0u || (hitTri != ~0u && (second <= win || leafT1 > win));
                    if (!flagged) store_hit(hits, slot, hitT, hbx, hby, hitTri, hitXform);
                    active = false;
                }
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(flagged);
                if (fm != 0ull) {
                    const uint32_t cntF = (uint32_t)__builtin_popcountll(fm);
                    uint32_t base = 0;
                    if (lane == (uint32_t)__builtin_ctzll(fm)) { base = atomicAdd(wb.flagCount, cntF); atomicAdd(wb.totals, (unsigned long long)cntF); }
                    base = (uint32_t)__shfl((int)base, __builtin_ctzll(fm));
                    if (flagged) {
                        const uint32_t at = base + (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1ull));
                        if (PRIMARY) wb.flagA[at] = rayIdx; else { wb.flagA[at] = slot; wb.flagB[at] = rayIdx; }
                    }
                }
            }
        }
        // ---- refill idle lanes (k_trace2's sliced work list, kernels_trace.hpp)
        unsigned long long idle = __ballot(!active);
        if (workLeft && ((uint32_t)__popcll(idle) >= REFILL_MIN || idle == ~0ull)) {
            const uint32_t n = (uint32_t)__popcll(idle);
            if (PROF) pn[1] += n;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t avail = chunkEnd - chunkNext;
            uint32_t q, sl; bool valid = true;
            if (avail >= n) { q = chunkNext + rank; sl = chunkSlice; chunkNext += n; }
            else {
                const uint32_t need = n - avail, want = grabChunk > need ? grabChunk : need;
                uint32_t fresh = 0, len = 0; bool got = false;
                while (slicesDone < GRAB_SLICES) {
                    len = ((nBlocks + GRAB_SLICES - 1u - slice) / GRAB_SLICES) << unitLog2;
                    fresh = wave_grab(workCounter + GRAB_STRIDE * slice, want);
                    if (PROF) pn[0]++;
                    if (fresh < len) { got = true; break; }
                    slice = (slice + 1u) & (GRAB_SLICES - 1u); slicesDone++;
                }
                q = rank < avail ? chunkNext + rank : fresh + (rank - avail); sl = rank < avail ? chunkSlice : slice;
                valid = rank < avail || (got && q < len);
                const uint32_t end = got ? (fresh + want < len ? fresh + want : len) : 0u;
                chunkNext = got ? (fresh + need < end ? fresh + need : end) : 0u; chunkEnd = end; chunkSlice = slice;
                if (got && fresh + want >= len) { slice = (slice + 1u) & (GRAB_SLICES - 1u); slicesDone++; }
            }
            const uint32_t item = valid ? ((((q >> unitLog2) * GRAB_SLICES + sl) << unitLog2) | (q & ((1u << unitLog2) - 1u))) : N;
            if (slicesDone >= GRAB_SLICES && chunkNext >= chunkEnd) workLeft = false;
            if (!active && item < N) {
                const uint32_t idx = list[item];
                rayIdx = idx; slot = PRIMARY ? idx : item;
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f; flags = 0u; leafT1 = 0.0f;
                if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                    float4 o = rays.o_ior[idx];
                    f3 wd = DecodeUnitVec(rays.thr_px[idx].w, rays.rad_py[idx].w), wo = mk3(o.x, o.y, o.z);
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                float rootT;
                { float4 a = tr.rec[4 * (size_t)idx], b = tr.rec[4 * (size_t)idx + 1], c = tr.rec[4 * (size_t)idx + 2]; ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z); rootT = a.w; }
                second = hitT; cullT = hitT * wide::CULL;
                const bool enter = rootT < hitT;                 // root test (:32-39): its arithmetic ran in the kernel that produced the ray
                const bool finite = gabs(invDir.x) < __builtin_inff() && gabs(invDir.y) < __builtin_inff() && gabs(invDir.z) < __builtin_inff();
                if (enter && !finite) flags = 1u;                // a slab of this ray can be NaN: the monotonicity argument does not cover it
                active = true; sp = stkBase; cur = (enter && finite) ? 0u : WIDE_NONE;
            }
        }
        WPROF_MARK(0);
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        // ---- node phase
        while (true) {
            const bool canStep = active && !(cur & wide::LEAF_BIT) && cur != WIDE_NONE;
            const unsigned long long stepMask = __builtin_amdgcn_ballot_w64(canStep);
            if (stepMask == 0ull) break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && (cur & wide::LEAF_BIT) != 0u)) >= f.leafMin) break;
            if (PROF) { pn[2]++; pn[3] += (unsigned long long)__builtin_popcountll(stepMask); }
            if (canStep) {
                if (COUNT) nNodes++;
                const uint4* p = wb.nodes + 4 * (size_t)cur;
                const uint32_t popped = sp[0];                          // what a pop would return (in flight with the node; the dummy row for an empty stack)
                const uint4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3];
                const float org[3] = {__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z)};
                const uint32_t qlo[3] = {n1.x, n1.y, n1.z}, qhi[3] = {n1.w, n2.x, n2.y};
                uint32_t cw[4] = {n2.z, n2.w, n3.x, n3.y};
                const float roA[3] = {ro.x, ro.y, ro.z}, inA[3] = {invDir.x, invDir.y, invDir.z};
                v2f near01 = {0.0f, 0.0f}, near23 = {0.0f, 0.0f}, far01 = {__builtin_inff(), __builtin_inff()}, far23 = far01;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float cell = __uint_as_float(((n0.w >> (8 * a)) & 255u) << 23);
                    const bool neg = inA[a] < 0.0f;
                    const uint32_t qn = neg ? qhi[a] : qlo[a], qf = neg ? qlo[a] : qhi[a];
                    const v2f c2 = {cell, cell}, o2 = {org[a], org[a]}, r2 = {roA[a], roA[a]}, i2 = {inA[a], inA[a]};
                    const v2f qn01 = {(float)(qn & 255u), (float)((qn >> 8) & 255u)}, qn23 = {(float)((qn >> 16) & 255u), (float)(qn >> 24)};
                    const v2f qf01 = {(float)(qf & 255u), (float)((qf >> 8) & 255u)}, qf23 = {(float)((qf >> 16) & 255u), (float)(qf >> 24)};
                    const v2f tn01 = (__builtin_elementwise_fma(qn01, c2, o2) - r2) * i2, tn23 = (__builtin_elementwise_fma(qn23, c2, o2) - r2) * i2;
                    const v2f tf01 = (__builtin_elementwise_fma(qf01, c2, o2) - r2) * i2, tf23 = (__builtin_elementwise_fma(qf23, c2, o2) - r2) * i2;
                    near01 = __builtin_elementwise_max(near01, tn01); near23 = __builtin_elementwise_max(near23, tn23);
                    far01 = __builtin_elementwise_min(far01, tf01); far23 = __builtin_elementwise_min(far23, tf23);
                }
                const float t1[4] = {near01.x, near01.y, near23.x, near23.y}, t2[4] = {far01.x, far01.y, far23.x, far23.y};
                uint32_t key[4]; int nHit = 0;
                bool nearMiss = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool in = t1[k] <= t2[k], h = (cw[k] != 0u) & in & (t1[k] <= cullT);
                    nearMiss |= (cw[k] != 0u) & !in & (t1[k] <= t2[k] * wide::NEAR_MISS);
                    key[k] = h ? __float_as_uint(t1[k]) : 0xffffffffu; nHit += h ? 1 : 0;      // (t1 >= 0: the bit pattern orders like the value)
                }
                flags |= nearMiss ? 8u : 0u;
                // sort (key, word) ascending: five compare-exchanges; misses (key ~0) end up last
#define WCAS(i, j) do { const bool sw = key[i] > key[j]; const uint32_t ka = sw ? key[j] : key[i], kb = sw ? key[i] : key[j], wa = sw ? cw[j] : cw[i], wbb = sw ? cw[i] : cw[j]; key[i] = ka; key[j] = kb; cw[i] = wa; cw[j] = wbb; } while (0)
                WCAS(0, 1); WCAS(2, 3); WCAS(0, 2); WCAS(1, 3); WCAS(1, 2);
#undef WCAS
                // continue with the nearest, push the others far ones first: entry j (1 <= j < nHit) lands nHit - j rows above the old top; pushes that do not happen write the spare row
                const int more = nHit > 0 ? nHit - 1 : 0;
                const bool ovf = sp + more * (int)WAVE > stkTop;
                lds_u32* const a1 = (nHit > 1 && !ovf) ? sp + (nHit - 1) * (int)WAVE : spare;
                lds_u32* const a2 = (nHit > 2 && !ovf) ? sp + (nHit - 2) * (int)WAVE : spare;
                lds_u32* const a3 = (nHit > 3 && !ovf) ? sp + (int)WAVE : spare;
                a3[0] = cw[3]; a2[0] = cw[2]; a1[0] = cw[1];
                flags |= ovf ? 2u : 0u;                                  // this ray needs a deeper stack than the launch has: the exact kernel traces it
                cur = ovf ? WIDE_NONE : (nHit > 0 ? cw[0] : popped);
                sp = ovf ? stkBase : (nHit > 0 ? sp + more * (int)WAVE : (sp != stkBase ? sp - (int)WAVE : sp));
            }
        }
        WPROF_MARK(1);
        // ---- leaf phase: every lane parked on a leaf record
        {
            const bool atLeaf = active && (cur & wide::LEAF_BIT) != 0u;
            if (PROF) { const unsigned long long lm = __ballot(atLeaf); if (lm) { pn[4]++; pn[5] += (unsigned long long)__popcll(lm); } }
            if (atLeaf) {
                if (COUNT) nLeaves++;
                const float4* r = wb.leaves + (cur & ~wide::LEAF_BIT);
                const float4 h0 = r[0], h1 = r[1];
                float4 a = r[2], b = r[3], c = r[4];
                float t1, t2;
                const bool boxHit = RayBoxIntersect(ro, invDir, h0, h1, &t1, &t2);
                if (!boxHit && t1 <= t2 * wide::NEAR_MISS) flags |= 8u;
                if (boxHit && t1 <= cullT) {
                    const uint32_t first = __float_as_uint(h0.w) + triOffset, cnt = __float_as_uint(h1.w);
                    for (uint32_t i = 0; i < cnt; i++) {
                        if (COUNT) nTris++;
                        if (i > 0u) { const float4* tv = r + 2 + 3 * (size_t)i; a = tv[0]; b = tv[1]; c = tv[2]; }
                        float by, bz, t;
                        if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t)) {
                            const uint32_t id = first + i;
                            const bool marked = __float_as_uint(a.w) != 0u;
                            if (!marked && t1 > t * wide::ASSUME) flags |= 4u;     // the one assumption of the argument (wide_nodes.hpp) does not hold for this triangle: not vouched for
                            if (t < hitT) { if (id != hitTri) second = gmin(second, hitT); hitT = t; cullT = t * wide::CULL; hbx = 1.0f - by - bz; hby = by; hitTri = id; hitXform = inst.MeshTransformId; leafT1 = marked ? __builtin_inff() : t1; }   // (a marked best hit is never vouched for: +inf fails the window test)
                            else if (id != hitTri) second = gmin(second, t);
                        }
                    }
                }
                cur = sp[0];
                sp = sp != stkBase ? sp - (int)WAVE : sp;
            }
        }
        WPROF_MARK(2);
    }
    if (COUNT && (nNodes | nLeaves | nTris)) { atomicAdd(wb.totals + 1, (unsigned long long)nNodes); atomicAdd(wb.totals + 2, (unsigned long long)nLeaves); atomicAdd(wb.totals + 3, (unsigned long long)nTris); }
    if (PROF && lane == 0) { for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)&counters[4 + i], pc[i]); for (int i = 0; i < 8; i++) atomicAdd((unsigned long long*)&counters[8 + i], pn[i]); }
#undef WPROF_MARK
}
