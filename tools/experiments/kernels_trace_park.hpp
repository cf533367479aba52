// kernels_trace_park.hpp — k_trace2p: k_trace2 (MODE 0) whose lanes may carry TWO parked leaves, for scenes whose boxes are nested.
// Part of the single translation unit idkpt.hip (included after kernels_trace.hpp).
//
// In k_trace2 a lane that finds a leaf parks until the wave's next leaf phase: with 32 samples in flight 39 of 64 lanes take a node step and 18-22 lanes sit in a leaf phase
// (profiles/r04_phase_profile.txt).  Here a lane with one parked leaf keeps stepping — with a T that may still shrink when that leaf is tested — until it finds a second
// one; the leaf phase then tests the first leaf as before and the second one after RE-VALIDATING it against the T the sequential walk has at that point
// (`tMin <= T` of the two leaf boxes, the test of BVHIntersect.glsl:49-53, with the tMin values kept from the step that found them).
// Why the hits are the reference's: box tests and the near / far order depend on T only through `tMin <= T`, so a walk with a stale (larger) T visits a SUPERSET of the
// sequential walk's nodes in the same relative order.  A leaf the sequential walk would not reach lies below a culled box (or is culled itself); boxes are nested
// (checked at upload: sceneNested) and tMin grows monotonically from a box to a box inside it for finite 1/dir, so that leaf's own tMin exceeds the sequential T and
// the re-validation drops it before any of its triangles is tested; a leaf the sequential walk does reach passes with exactly the flags (left / right / both -> the
// triangle range of :54-56) the sequential walk computes.  Triangles are tested in the sequential order with the sequential T.  Surplus stack entries pushed under the
// stale T are popped later and find nothing (their children fail `tMin <= T`).  Rays whose 1/dir is not finite never carry a second leaf.  Visit counters differ, so the
// counting build keeps k_trace2.  The stack bound (ComputeRequiredStackSize) is structural and holds for any subset of visits.
#pragma once

template <bool PRIMARY, int REFILL_MIN = 32>
__global__ __launch_bounds__(WAVE, 1) void k_trace2p(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter)
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
    const int nodeOffset = s.descs[inst.BlasId].NodeOffset;
    const uint32_t triOffset = (uint32_t)s.descs[inst.BlasId].TriangleOffset;
    const float4* nodes = s.nodes + 2 * (size_t)nodeOffset;
    const float INF = __builtin_inff();

    bool active = false, leafPending = false, workLeft = true;
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    if (N == 0u) workLeft = false;
    uint32_t top = 0, slot = 0, leafFirst = 0, leafEnd = 0;
    bool leaf2 = false, specOK = false;                              // a second parked leaf (found with a T that may still shrink); this ray may carry one (finite 1/dir)
    uint32_t p1A = 0, p1B = 0, p1Cnt = 0; float p1tL = 0.0f, p1tR = 0.0f;   // ... its left / right triangle ranges (start, start, counts packed) and the tMin of the two leaf boxes (+inf: not a hit leaf)
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
                active = true; leafPending = false; leaf2 = false; sp = stkBase; top = enter ? 2u : 0u;
                specOK = __builtin_isfinite(invDir.x) && __builtin_isfinite(invDir.y) && __builtin_isfinite(invDir.z);
            }
        }
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        // ---- node phase (k_trace2's branch-free step; a lane with ONE parked leaf keeps stepping)
        while (true) {
            const bool canStep = active && top != 0u && !leaf2 && (!leafPending || specOK);
            if (__builtin_amdgcn_ballot_w64(canStep) == 0ull) break;
            const bool blocked = active && leafPending && !canStep;     // needs the leaf phase before it can go on
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(blocked)) >= f.leafMin) break;
            if (canStep) {
                const float4* p = nodes + 2 * (size_t)top;
                const uint32_t popped = sp[0];
                float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
                const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
                float tMinLeft, tMinRight;
                const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hitT;
                const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hitT;
                const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
                const bool found = intersectLeft || intersectRight, second = leafPending;
                // first parked leaf: its range is final (T is exact here); second one: both children's ranges and tMin, decided again when it is tested
                leafFirst = second ? leafFirst : (intersectLeft ? lStart : rStart); leafEnd = second ? leafEnd : (!intersectRight ? lStart + lCount : rStart + rCount);
                p1A = lStart; p1B = rStart; p1Cnt = lCount | (rCount << 16); p1tL = intersectLeft ? tMinLeft : INF; p1tR = intersectRight ? tMinRight : INF;   // (a stepping lane has no second leaf: free registers)
                leaf2 = second && found; leafPending = second || found;
                const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
                const bool both = traverseLeft && traverseRight, none = !(traverseLeft || traverseRight);
                const bool leftCloser = tMinLeft < tMinRight;
                const uint32_t nearChild = both ? (leftCloser ? lStart : rStart) : (traverseLeft ? lStart : rStart);
                sp[WAVE] = leftCloser ? rStart : lStart;
                const bool full = sp == stkFull, nonEmpty = sp != stkBase;
                ovf = ovf || (both && full);
                top = none ? (nonEmpty ? popped : 0u) : nearChild;
                sp += (both && !full) ? (int)WAVE : ((none && nonEmpty) ? -(int)WAVE : 0);
            }
        }
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
            if (leaf2) {
                // the two leaf tests of that node with the T the sequential walk has here (BVHIntersect.glsl:49-53), then its triangle range (:54-56)
                const bool iL = p1tL <= hitT, iR = p1tR <= hitT;
                if (iL || iR) {
                    const uint32_t first = (iL ? p1A : p1B) + triOffset, end = (!iR ? p1A + (p1Cnt & 0xffffu) : p1B + (p1Cnt >> 16)) + triOffset;
                    for (uint32_t i = first; i < end; i++) {
                        const float4* tv = s.triVerts + 3 * (size_t)i;
                        float4 a = tv[0], b = tv[1], c = tv[2];
                        float by, bz, t;
                        if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hitT) {
                            hitTri = i; hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = inst.MeshTransformId;
                        }
                    }
                }
                leaf2 = false;
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
