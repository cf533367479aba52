// kernels_trace_quad.hpp — k_trace2q: k_trace2 (MODE 0) on a derived "quad" node layout that resolves up to TWO binary levels per memory round trip, for launches
// that are bound by the dependent chain of their longest rays.  Part of the single translation unit idkpt.hip (included after kernels_trace.hpp).
//
// VERDICT r3 item 3 ("change bytes or round trips per ray while replaying the reference's order exactly").  A record of the derived array holds a node pair AND the
// pairs of its two children where those are internal (192 B, k_derive_quads).  A lane fetches the record of its current pair, runs the ordinary node step on the pair —
// and, if that step descends into an internal child and found no leaf (a parked leaf must be tested before the lane's next box test: T may shrink), runs the NEXT
// ordinary node step right away on the child's pair, which is already in registers.  Every ray executes exactly the reference's sequence of box tests, pushes, pops and
// leaf tests (BVHIntersect.glsl:43-101); only the fetch of a pair that is reached by descending from the step before is gone.  About two thirds of a ray's pairs are
// reached that way, in runs; a run of d descents costs ceil(d / 2) round trips instead of d.
// What it costs: 3x the bytes per round trip, 12 instead of 4 loads per lane and step, ~48 instead of 16 registers of node data, and 178 MB for a 1 M-triangle BLAS.
// Where it was measured to pay and where not: profiles/r04_quad_records.md.  The host selects it (want_quad, idkpt.hip) for small launches only; static scenes only
// (the array is re-derived lazily after any change of the nodes; with scene versions in use the plain kernels run).
#pragma once

// q[6 * top .. 6 * top + 11] for every even node index top >= 2 of one BLAS (nodes = the BLAS's node array in the order the traversal reads, float4 pairs per node)
__global__ __launch_bounds__(256) void k_derive_quads(const float4* nodes, float4* q, uint32_t nodeCount)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, top = 2u * p;
    if (p == 0u || top + 1u >= nodeCount) return;
    const float4 a0 = nodes[2 * (size_t)top], a1 = nodes[2 * (size_t)top + 1], a2 = nodes[2 * (size_t)top + 2], a3 = nodes[2 * (size_t)top + 3];
    float4* o = q + 6 * (size_t)top;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    const uint32_t l = __float_as_uint(a0.w), lc = __float_as_uint(a1.w), r = __float_as_uint(a2.w), rc = __float_as_uint(a3.w);
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool li = lc == 0u && l >= 2u && l + 1u < nodeCount, ri = rc == 0u && r >= 2u && r + 1u < nodeCount;
    float4 b0 = z, b1 = z, b2 = z, b3 = z, c0 = z, c1 = z, c2 = z, c3 = z;
    if (li) { const float4* pl = nodes + 2 * (size_t)l; b0 = pl[0]; b1 = pl[1]; b2 = pl[2]; b3 = pl[3]; }
    if (ri) { const float4* pr = nodes + 2 * (size_t)r; c0 = pr[0]; c1 = pr[1]; c2 = pr[2]; c3 = pr[3]; }
    o[4] = b0; o[5] = b1; o[6] = b2; o[7] = b3; o[8] = c0; o[9] = c1; o[10] = c2; o[11] = c3;
}

template <bool PRIMARY, int REFILL_MIN = 32>
__global__ __launch_bounds__(WAVE, 1) void k_trace2q(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter, const float4* quads)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const stkBase = (lds_u32*)lds + lane;
    const int cap = f.stackCap;
    lds_u32* const stkFull = stkBase + cap * WAVE;
    const uint32_t N = *countPtr;
    {   // (the grid rules of k_trace2)
        uint32_t want = gridDim.x;
        if (f.gridRaysX4 > 0u) want = max((uint32_t)(((unsigned long long)N * 4ull / f.gridRaysX4 + 63ull) / 64ull), min(want, 1024u));
        if (f.gridMid > 0u && N < f.gridMidRays) want = min(want, f.gridMid);
        if (blockIdx.x >= max(want, 1u)) return;
    }
    const GpuBlasInstance inst = s.instances[0];
    const uint32_t triOffset = (uint32_t)s.descs[inst.BlasId].TriangleOffset;

    bool active = false, leafPending = false, workLeft = true;
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    if (N == 0u) workLeft = false;
    uint32_t top = 0, slot = 0, leafFirst = 0, leafEnd = 0;
    lds_u32* sp = stkBase;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, hbx = 0.0f, hby = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;
    bool ovf = false;

    while (true) {
        // ---- refill idle lanes (k_trace2's hand-out)
        unsigned long long idle = __ballot(!active);
        if (workLeft && ((uint32_t)__popcll(idle) >= REFILL_MIN || idle == ~0ull)) {
            const uint32_t n = (uint32_t)__popcll(idle);
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
                const bool ordered = !PRIMARY && tr.order != nullptr;
                const uint32_t idx = ordered ? tr.orderIdx[item] : list[item];
                slot = PRIMARY ? idx : (ordered ? tr.order[item] : item);
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f;
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
                const bool enter = rootT < hitT;
                active = true; leafPending = false; sp = stkBase; top = enter ? 2u : 0u;
            }
        }
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        // ---- node phase: one record per round trip, up to two of k_trace2's (branch-free) node steps on it
        // (NODE_STEP: the step of kernels_trace.hpp on the pair (lmin, lmax, rmin, rmax); leaves `descended` = the step went down into an internal child, `wentLeft` = into the left one)
#define NODE_STEP(lmin, lmax, rmin, rmax)                                                                                                                                    \
        {                                                                                                                                                                    \
            const uint32_t popped = sp[0];                                                                                                                                   \
            const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);           \
            float tMinLeft, tMinRight;                                                                                                                                       \
            const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hitT;                                                                     \
            const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hitT;                                                                  \
            const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;                                                                       \
            leafFirst = intersectLeft ? lStart : rStart; leafEnd = !intersectRight ? lStart + lCount : rStart + rCount; leafPending = intersectLeft || intersectRight;       \
            const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;                                                                       \
            const bool both = traverseLeft && traverseRight, none = !(traverseLeft || traverseRight);                                                                        \
            const bool leftCloser = tMinLeft < tMinRight;                                                                                                                    \
            wentLeft = both ? leftCloser : traverseLeft;                                                                                                                     \
            const uint32_t nearChild = wentLeft ? lStart : rStart;                                                                                                           \
            sp[WAVE] = leftCloser ? rStart : lStart;                                                                                                                         \
            const bool full = sp == stkFull, nonEmpty = sp != stkBase;                                                                                                       \
            ovf = ovf || (both && full);                                                                                                                                     \
            top = none ? (nonEmpty ? popped : 0u) : nearChild;                                                                                                               \
            sp += (both && !full) ? (int)WAVE : ((none && nonEmpty) ? -(int)WAVE : 0);                                                                                       \
            descended = !none;                                                                                                                                               \
        }
        while (true) {
            const bool canStep = active && !leafPending && top != 0u;
            if (__builtin_amdgcn_ballot_w64(canStep) == 0ull) break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && leafPending)) >= f.leafMin) break;
            if (canStep) {
                const float4* q = quads + 6 * (size_t)top;
                const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
                const float4 l0 = q[4], l1 = q[5], l2 = q[6], l3 = q[7];
                const float4 r0 = q[8], r1 = q[9], r2 = q[10], r3 = q[11];
                // (all twelve loads leave together and are waited for HERE: left alone, the compiler sinks the eight child loads into the second step's block — a second,
                // dependent round trip, which is what the record exists to avoid; checked in the ISA)
                asm volatile("" :: "v"(l0.x), "v"(l0.y), "v"(l0.z), "v"(l0.w), "v"(l1.x), "v"(l1.y), "v"(l1.z), "v"(l1.w), "v"(l2.x), "v"(l2.y), "v"(l2.z), "v"(l2.w), "v"(l3.x), "v"(l3.y), "v"(l3.z), "v"(l3.w));
                asm volatile("" :: "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r0.w), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w), "v"(r2.x), "v"(r2.y), "v"(r2.z), "v"(r2.w), "v"(r3.x), "v"(r3.y), "v"(r3.z), "v"(r3.w));
                bool descended, wentLeft;
                NODE_STEP(a0, a1, a2, a3)
                if (descended && !leafPending) {
                    // the next node step of this lane, on the child pair that came with the record (no fetch); T has not changed since (no leaf in between)
                    const float4 b0 = wentLeft ? l0 : r0, b1 = wentLeft ? l1 : r1, b2 = wentLeft ? l2 : r2, b3 = wentLeft ? l3 : r3;
                    NODE_STEP(b0, b1, b2, b3)
                }
            }
        }
#undef NODE_STEP
        // ---- leaf phase (BVHIntersect.glsl:54-79)
        if (leafPending) {
            for (uint32_t i = leafFirst + triOffset, e = leafEnd + triOffset; i < e; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)i;
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hitT) {
                    hitTri = i; hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = inst.MeshTransformId;
                }
            }
            leafPending = false;
        }
        if (active && top == 0u) {
            store_hit(hits, slot, hitT, hbx, hby, hitTri, hitXform);
            active = false;
        }
    }
    if (ovf) *s.overflow = 1u;
}
