// kernels_trace_fused.hpp — FirstHit and the last NHit of a RayDepth-2 frame in ONE persistent launch, for launches of few rays per lane.
// Part of the single translation unit idkpt.hip (included there, after kernels_shade.hpp); see DESIGN.md §4 for the kernel table.
#pragma once

// Why: a frame traced alone (or a rank's rows of one) is bound by the dependent chain of its longest rays, once per traversal launch
// (profiles/r04_small_launch_experiments.md §0: 633 node steps x 0.8-0.9 us = the 0.5 ms BOTH launches of a lone headline frame take, whatever their ray counts).  The
// reference's wavefront schedule (PathTracer.cs:214-271) puts a full barrier between FirstHit and NHit, so a frame pays that chain twice, plus the shading kernel
// between them.  Nothing in the arithmetic needs the barrier at RayDepth 2: FirstHit seeds its RNG per PIXEL (FirstHit/compute.glsl:53), and of the last NHit only the
// radiance a ray picks up reaches the image (k_shade_last, kernels_shade.hpp) — the slot-seeded rest (NHit/compute.glsl:54) is produced on demand from the stored
// hits, after the ordered compaction has run.  So here a lane that finishes a primary ray shades it (the body of k_shade_first: same functions, same operands, same
// bits), and traces the bounce ray right away.  (The hope — that rays with long primary walks are grazing misses without a bounce, so that the frame's chain
// becomes max(primary + bounce) over its rays instead of max(primary) + max(bounce) — did not hold: see "Measured" below.)
// What leaves the kernel is what the two launches and k_shade_first leave: the ray state after FirstHit, the continue flags, the sort keys — and the bounce's hit
// records, stored per RAY ID (Frame::hitsByRid: k_shade_last / k_restore_last / the on-demand k_shade<false> look them up through the queue entry instead of the slot).
// Shading runs for parked lanes together (like leaves): when Frame::shadeMin lanes wait, or as many as are still tracing.
// Selected by the host (want_fused, idkpt.hip) for RayDepth 2, one BLAS instance, no TLAS, no AOVs, last bounce deferred, small launches on sparse views.
// Measured (profiles/r04_small_launch_experiments.md §5): the chain does NOT get shorter — the rays with the longest primary walks are the ones that end deep inside the
// scene, and their bounce rays start there: the frame's critical path is primary -> shading -> bounce of the SAME pixels, serial in any schedule.  What the kernel saves is
// two launches and the shading kernel between them: +5 % for the headline view traced one frame at a time (2 160 -> 2 277-2 305 Mray/s), and it costs 161 VGPRs
// (3 waves per SIMD): -20 to -35 % wherever most pixels traverse the scene or several samples are in flight.  Hence the narrow rule.
template <int REFILL_MIN = 32>
__global__ __launch_bounds__(WAVE, 1) void k_trace_fused(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter,
                                                        uint8_t* contFlag, uint32_t* seedsAndKeys, int lean)
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

    bool active = false, leafPending = false, workLeft = true;
    bool bounce = false, shadePend = false;                          // this lane's ray is the bounce ray of its pixel; its primary ray is finished and waits for the shading phase
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    if (N == 0u) workLeft = false;
    uint32_t top = 0, rid = 0, leafFirst = 0, leafEnd = 0;
    lds_u32* sp = stkBase;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, hbx = 0.0f, hby = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;
    bool ovf = false;

    while (true) {
        // ---- refill idle lanes from the primary work list (k_trace2's hand-out)
        unsigned long long idle = __ballot(!active && !shadePend);
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
            if (!active && !shadePend && item < N) {
                rid = list[item];
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f;
                if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                    float4 o = rays.o_ior[rid];
                    f3 wd = DecodeUnitVec(rays.thr_px[rid].w, rays.rad_py[rid].w), wo = mk3(o.x, o.y, o.z);
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                float rootT;
                { float4 a = tr.rec[4 * (size_t)rid], b = tr.rec[4 * (size_t)rid + 1], c = tr.rec[4 * (size_t)rid + 2]; ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z); rootT = a.w; }
                const bool enter = rootT < hitT;
                active = true; bounce = false; leafPending = false; sp = stkBase; top = enter ? 2u : 0u;
            }
        }
        if (__ballot(active || shadePend) == 0ull) { if (!workLeft) break; continue; }

        // ---- node phase (k_trace2's branch-free step)
        while (true) {
            const bool canStep = active && !leafPending && top != 0u;
            const unsigned long long stepMask = __builtin_amdgcn_ballot_w64(canStep);
            if (stepMask == 0ull) break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && leafPending)) >= f.leafMin) break;
            if (canStep) {
                const float4* p = nodes + 2 * (size_t)top;
                const uint32_t popped = sp[0];
                float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
                const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
                float tMinLeft, tMinRight;
                const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hitT;
                const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hitT;
                const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
                leafFirst = intersectLeft ? lStart : rStart; leafEnd = !intersectRight ? lStart + lCount : rStart + rCount; leafPending = intersectLeft || intersectRight;
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
            leafPending = false;
        }
        // ---- finished rays: a bounce ray stores its hit (per ray id) and frees the lane; a primary ray waits for the shading phase with its hit in registers
        if (active && top == 0u) {
            if (bounce) store_hit(hits, rid, hitT, hbx, hby, hitTri, hitXform);
            else shadePend = true;
            active = false;
        }
        // ---- shading phase (the body of k_shade_first, kernels_shade.hpp) for the parked lanes together
        {
            const uint32_t nS = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(shadePend)), nA = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(active));
            if (nS == 0u || !(nS >= (uint32_t)f.shadeMin || nS >= nA)) continue;
        }
        if (shadePend) {
            shadePend = false;
            const uint32_t smp = rid / f.Npad, pix = rid - smp * f.Npad;
            const uint32_t acc = sample_index(f, smp);
            HitRec hit; hit.T = hitT; hit.bx = hbx; hit.by = hby; hit.tri = hitTri; hit.xform = hitXform;
            RayState r; uint32_t rng, key = 0;
            if (lean) {
                f2 pd; gen_primary(f, smp, pix, acc, r.origin, pd, rng);
                r.prevIor = 1.0f; r.throughput = splat3(1.0f); r.pdx = pd.x; r.radiance = splat3(0.0f); r.pdy = pd.y;
            } else {
                float4 a = rays.o_ior[rid], b = rays.thr_px[rid], c = rays.rad_py[rid];
                r.origin = mk3(a.x, a.y, a.z); r.prevIor = a.w; r.throughput = mk3(b.x, b.y, b.z); r.pdx = b.w; r.radiance = mk3(c.x, c.y, c.z); r.pdy = c.w;
                rng = seedsAndKeys[rid];
            }
            AovState aov; aov.albedo = splat3(0.0f); aov.normal = splat3(0.0f); aov.newWeight = 1.0f;
            const int lx = (int)(pix % (uint32_t)f.W), ly = (int)(pix / (uint32_t)f.W);
            const uint32_t gidSeed = first_hit_gid_seed(f.W, f.H, lx, global_row(f, ly));
            const f3 rd0 = DecodeUnitVec(r.pdx, r.pdy);
            const bool cont = ShadeHit<true>(s, f, acc, hit, hit.T != PT_FLOAT_MAX, rd0, r, aov, rng, gidSeed, key);
            rays.o_ior[rid] = make_float4(r.origin.x, r.origin.y, r.origin.z, r.prevIor);
            rays.thr_px[rid] = make_float4(r.throughput.x, r.throughput.y, r.throughput.z, r.pdx);
            rays.rad_py[rid] = make_float4(r.radiance.x, r.radiance.y, r.radiance.z, r.pdy);
            if (f.outputAovs) { rays.aovA[rid] = make_float4(aov.albedo.x, aov.albedo.y, aov.albedo.z, aov.newWeight); rays.aovN[rid] = make_float4(aov.normal.x, aov.normal.y, aov.normal.z, 0.0f); }
            seedsAndKeys[rid] = (key & ((1u << IDKPT_SORT_KEY_BITS) - 1u)) | (smp << IDKPT_SORT_KEY_BITS);
            if (cont) {
                contFlag[rid] = 1;
                // the bounce ray, prepared as write_trace_ready does (NHit:93, BVHIntersect.glsl:281-282, IntersectionRoutines.glsl:29) — into registers instead of the record
                const f3 wd = DecodeUnitVec(r.pdx, r.pdy);
                const M34 inv = load_inv_model(s, inst.MeshTransformId);
                ro = xform34(inv, r.origin, 1.0f); rd = xform34(inv, wd, 0.0f);
                invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                const float4* root = s.nodes + 2 * (size_t)nodeOffset + 2;
                float t1;
                const float rootT = RayBoxIntersect(ro, invDir, root[0], root[1], &t1) ? t1 : __builtin_inff();
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f;
                if (f.g.DoTraceLights) {
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(r.origin, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                const bool enter = rootT < hitT;
                active = true; bounce = true; leafPending = false; sp = stkBase; top = enter ? 2u : 0u;
            }
        }
    }
    if (ovf) *s.overflow = 1u;
}
