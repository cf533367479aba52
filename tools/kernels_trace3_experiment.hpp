// kernels_trace3.hpp — k_trace3: the production traversal kernel (BVHIntersect.glsl:27-105, 183-291), second generation of the persistent
// while-while kernel k_trace2 (kernels_trace.hpp, kept as the counting build and as the cross-check in the tests).
// Part of the single translation unit idkpt.hip; see DESIGN.md §4.
//
// What changed against k_trace2, and why each step keeps the hits bit-identical:
//
//  1. Speculative stepping (SPEC).  A lane that found a leaf keeps its leaf range parked, as before, but goes on taking node steps with
//     its now possibly stale T instead of idling until enough other lanes are parked too.  If such a step finds ANOTHER leaf the step is
//     discarded (top / stack untouched) and the lane is "blocked": it repeats that step after its parked leaf has been tested.  So a leaf
//     range is only ever recorded by a step that ran with an up-to-date T, every ray tests exactly the reference's sequence of
//     triangles in the reference's order, and T evolves identically.  Steps taken with a stale T can only pass MORE children (T never
//     grows): extra far children get pushed and extra near children entered; when they are reached again T is current and — child boxes
//     are subsets of their parents under the monotone slab arithmetic — everything below them is rejected.  The depth-first order of
//     the nodes the reference visits is unchanged (near/far is decided by tMin alone).  Only the visit counters differ, hence the
//     counting build stays on k_trace2.
//  2. Ride-along refill.  k_trace2 refilled idle lanes in a separate wave-wide step (atomic -> list entry -> ray -> root test: three
//     dependent memory latencies during which the whole wave made no progress), which is why it only paid off once 32 lanes were idle.
//     Here every wave reserves the work list in chunks of 64 entries, one chunk ahead (the atomic of chunk k+2 and the coalesced list
//     load of chunk k+1 are in flight while chunk k is consumed), a refill hands list entries to the idle lanes with one cross-lane
//     move, and FETCHING A RAY IS A NODE STEP: the trace-ready ray is a 64-B record (TraceBufs), the size of a node pair, so a refilled
//     lane simply points the four 16-B loads of the node step at its ray record instead of a node pair.  The refill costs no load
//     instruction, no register and no latency of its own, and can run as soon as FETCH_MIN lanes are idle.  The root-box test of the
//     new ray was done by the kernel that produced it (record[0].w = the box's tMin); here it is one compare.
//  3. Wave votes use the ballot builtin directly (no bool -> int -> compare round trip through VGPRs).
#pragma once
#include <type_traits>

DEV unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEV uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

template <bool PRIMARY, int MODE, bool SPEC, int FETCH_MIN, int LEAF_MIN>
__global__ __launch_bounds__(WAVE) void k_trace3(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter)
{
    constexpr bool MULTI = MODE != 0, TLAS = MODE == 2;
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t* stk = lds + lane;
    const int cap = f.stackCap;
    uint32_t* tstk = stk + cap * WAVE;     // TLAS only
    const uint32_t N = *countPtr;
    // wave-uniform scene constants (MODE 0)
    const GpuBlasInstance inst = s.instances[0];
    const int nodeOffset = s.descs[inst.BlasId].NodeOffset;
    const uint32_t triOffset = (uint32_t)s.descs[inst.BlasId].TriangleOffset;
    const float4* nodes = s.nodes + 2 * (size_t)nodeOffset;

    // ---- work list.  Large lists (f.aheadMinRays: several chunks per resident wave) are reserved in chunks of 64 entries, one chunk ahead;
    //      short lists (one frame alone: fewer rays than lanes on the chip) are handed out on demand, exactly as many entries as lanes are
    //      idle, so that every wave gets its share — looking ahead would park two chunks on the first waves and leave the others empty.
    const bool ahead = N >= f.aheadMinRays;
    uint32_t curBase = 0, curLen = 0, curUsed = 0, curIdx = 0;     // entries being handed out: [curBase + curUsed, curBase + curLen); curIdx = list[curBase + lane]
    uint32_t nextBase = 0, nextLen = 0, nextIdx = 0;               // chunk whose list entries are being loaded
    uint32_t resvBase = 0, resvV = 0; bool haveResv = false;       // reservation behind it (resvV: landing register of its atomic)
    bool listDone = false;                                         // on-demand mode: a reservation reached the end of the list
    if (ahead) {
        // start-up: three dependent steps, exposed once per wave
        uint32_t b0 = wave_grab(workCounter, WAVE);
        if (b0 < N) { curBase = b0; curLen = min((uint32_t)WAVE, N - b0); if (lane < curLen) curIdx = list[b0 + lane]; }
        if (b0 + WAVE < N) {
            uint32_t b1 = wave_grab(workCounter, WAVE);
            if (b1 < N) { nextBase = b1; nextLen = min((uint32_t)WAVE, N - b1); if (lane < nextLen) nextIdx = list[b1 + lane]; }
            if (b1 + WAVE < N) { resvBase = wave_grab(workCounter, WAVE); haveResv = true; }
        }
        asm volatile("" : "+v"(curIdx), "+v"(nextIdx));           // both loads have landed before the loop starts (no pending state is carried into it)
    }

    bool active = false, leafPending = false, blocked = false;
    uint32_t top = 0, slot = 0, leafFirst = 0, leafEnd = 0;
    uint32_t instIdx = 0, rayId = 0, nodeOff = 0, triOff = 0, xformId = 0;   // MULTI only
    int tsp = 0; bool moreInst = false;
    int sp = 0;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, hbx = 0.0f, hby = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;

    // One step of the wave.  Lanes in `canStep` take a node step (BVHIntersect.glsl:43-53,81-101); lanes in `take` (MODE 0) start the ray
    // `takeIdx`: the same four loads fetch its 64-B record instead of a node pair.
    auto node_step = [&](auto withTake, const bool canStep, const bool takeArg, const uint32_t takeIdx) {
        const bool take = decltype(withTake)::value && !MULTI && takeArg;
        if (canStep || take) {
            const float4* p = MULTI ? s.nodes + 2 * ((size_t)nodeOff + top) : (take ? (const float4*)tr.rec + 4 * (size_t)takeIdx : nodes + 2 * (size_t)top);
            float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
            if (decltype(withTake)::value && !MULTI) {
                // the two kinds of lanes read different parts of the 64 bytes: keep the four 16-B loads whole and in front of the branch
                asm volatile("" : "+v"(lmin.x), "+v"(lmin.y), "+v"(lmin.z), "+v"(lmin.w), "+v"(lmax.x), "+v"(lmax.y), "+v"(lmax.z), "+v"(lmax.w),
                                  "+v"(rmin.x), "+v"(rmin.y), "+v"(rmin.z), "+v"(rmin.w), "+v"(rmax.x), "+v"(rmax.y), "+v"(rmax.z), "+v"(rmax.w));
            }
            if (take) {
                ro = mk3(lmin.x, lmin.y, lmin.z); rd = mk3(lmax.x, lmax.y, lmax.z); invDir = mk3(rmin.x, rmin.y, rmin.z);
                top = lmin.w < hitT ? 2u : 0u;       // root test (:32-39): the producer of the ray stored the box's tMin (+inf = miss) in record[0].w
            } else {
                const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
                float tMinLeft, tMinRight;
                const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hitT;
                const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hitT;
                const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
                const bool foundLeaf = intersectLeft || intersectRight;
                if (SPEC && foundLeaf && leafPending) {
                    blocked = true;                  // the step ran with a stale T and found a leaf: discard it, repeat it after the parked leaf was tested
                } else {
                    if (foundLeaf) {
                        const uint32_t tOff = MULTI ? triOff : triOffset;
                        leafFirst = (intersectLeft ? lStart : rStart) + tOff;
                        leafEnd = (!intersectRight ? (lStart + lCount) : (rStart + rCount)) + tOff;
                        leafPending = true;
                    }
                    const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
                    if (traverseLeft || traverseRight) {
                        if (traverseLeft && traverseRight) {
                            const bool leftCloser = tMinLeft < tMinRight;
                            top = leftCloser ? lStart : rStart;
                            if (sp < cap) stk[sp * WAVE] = leftCloser ? rStart : lStart; else *s.overflow = 1u;
                            sp++;
                        } else top = traverseLeft ? lStart : rStart;
                    } else {
                        if (sp == 0) top = 0u;
                        else { sp--; top = sp < cap ? stk[sp * WAVE] : 0u; }
                    }
                }
            }
        }
    };
    // lane state predicates
    auto can_step = [&]() { return active && top != 0u && !blocked && (SPEC || !leafPending); };
    auto instances_left = [&]() { return MULTI && (TLAS ? moreInst : instIdx < (uint32_t)s.instanceCount); };
    auto finished = [&]() { return active && top == 0u && !leafPending && !instances_left(); };
    // lanes that cannot go on before the wave leaves the node loop: a parked leaf has to be tested / the next instance has to be set up
    auto waiting = [&]() { return active && ((leafPending && (!SPEC || blocked || top == 0u)) || (top == 0u && !leafPending && instances_left())); };

    while (true) {
        // ---- F. retire finished rays (MULTI: only after the last instance)
        if (finished()) {
            hits.hit[slot] = make_float4(hitT, hbx, hby, __uint_as_float(hitTri));
            hits.xformId[slot] = hitXform;
            active = false;
        }
        // ---- A. chunk hand-over: the current chunk is used up -> the prefetched one becomes current, and the look-ahead moves on: the list
        //      entries of the chunk reserved earlier start loading, the atomic of the chunk behind it is issued.  Neither is waited for
        //      here: both are picked up behind this iteration's node loads (D'), whose wait they share (VMEM returns in order).
        bool issuedAtomic = false;
        if (curUsed == curLen && nextLen != 0u) {
            curBase = nextBase; curLen = nextLen; curUsed = 0; curIdx = nextIdx;
            nextLen = 0;
            if (haveResv) {
                const uint32_t b = resvBase;
                haveResv = false;
                if (b < N) {
                    nextBase = b; nextLen = min((uint32_t)WAVE, N - b); if (lane < nextLen) nextIdx = list[b + lane];
                    if (b + WAVE < N) { if (lane == 0) resvV = atomicAdd(workCounter, (uint32_t)WAVE); issuedAtomic = true; }
                }
            }
        }
        const unsigned long long idle = wballot(!active);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (!ahead && curUsed == curLen && !listDone && (nIdle >= 32u || idle == ~0ull)) {
            // on-demand reservation (short lists): one entry per idle lane, like k_trace2's refill
            const uint32_t b = wave_grab(workCounter, nIdle);
            curBase = b; curUsed = 0; curLen = b < N ? min(nIdle, N - b) : 0u;
            curIdx = (lane < curLen) ? list[b + lane] : 0u;
            asm volatile("" : "+v"(curIdx));          // needed right below; waiting here keeps the look-ahead path free of waits
            if (b + nIdle >= N) listDone = true;
        }
        // ---- A'. refill: idle lanes take list entries; their rays are fetched by the node step below (D)
        const uint32_t avail = curLen - curUsed;
        bool take = false; uint32_t takeIdx = 0;
        if (avail != 0u && (nIdle >= (uint32_t)FETCH_MIN || idle == ~0ull)) {
            const uint32_t rank = lanes_below(idle);
            take = !active && rank < avail;
            const uint32_t src = curUsed + rank;
            takeIdx = (uint32_t)__shfl((int)curIdx, (int)(src & 63u));
            curUsed += min(nIdle, avail);
            if (take) {
                slot = PRIMARY ? takeIdx : curBase + src;
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f;
                if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                    float4 o = rays.o_ior[takeIdx];
                    f3 wd = DecodeUnitVec(rays.thr_px[takeIdx].w, rays.rad_py[takeIdx].w), wo = mk3(o.x, o.y, o.z);
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                active = true; leafPending = false; blocked = false; sp = 0; top = 0u;
                if (MULTI) { rayId = takeIdx; instIdx = 0; tsp = 0; moreInst = TLAS ? s.tlasCount > 0 : true; }
            }
        } else if (idle == ~0ull && (ahead ? nextLen == 0u : listDone)) break;       // nothing running, nothing left to hand out

        // ---- B. instance list / TLAS walk of lanes whose current BLAS is exhausted (a parked leaf is tested first: it belongs to the old instance's ray)
        if (TLAS) {
            bool adv = active && !leafPending && top == 0u && moreInst;
            while (wballot(adv) != 0ull) {
                if (adv) {
                    const float4 pmin = s.tlas[2 * (size_t)instIdx];
                    const uint32_t packed = __float_as_uint(pmin.w), id = packed & 0x7fffffffu;
                    if ((packed >> 31) == 1u) {                                             // leaf: BVHIntersect.glsl:223-240
                        const GpuBlasInstance in2 = s.instances[id];
                        const M34 inv = load_inv_model(s, in2.MeshTransformId);
                        float4 a = tr.rec[4 * (size_t)rayId], b = tr.rec[4 * (size_t)rayId + 1];
                        ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                        invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        nodeOff = (uint32_t)s.descs[in2.BlasId].NodeOffset; triOff = (uint32_t)s.descs[in2.BlasId].TriangleOffset; xformId = in2.MeshTransformId;
                        sp = 0; top = 2u;                                                   // no root test under USE_TLAS (:32)
                        if (tsp == 0 || tsp > f.tlasCap) moreInst = false; else instIdx = tstk[--tsp * WAVE];
                    } else {
                        const uint32_t l = id, r = id + 1;
                        float4 a = tr.rec[4 * (size_t)rayId], c = tr.rec[4 * (size_t)rayId + 2];                         // world-space origin and 1/dir
                        const f3 wo = mk3(a.x, a.y, a.z), winv = mk3(c.x, c.y, c.z);
                        float4 lmin = s.tlas[2 * (size_t)l], lmax = s.tlas[2 * (size_t)l + 1], rmin = s.tlas[2 * (size_t)r], rmax = s.tlas[2 * (size_t)r + 1];
                        float tMinLeft, tMinRight;
                        const bool tl = RayBoxIntersect(wo, winv, lmin, lmax, &tMinLeft) && tMinLeft < hitT;
                        const bool tr2 = RayBoxIntersect(wo, winv, rmin, rmax, &tMinRight) && tMinRight < hitT;
                        if (tl || tr2) {
                            if (tl && tr2) { const bool lc = tMinLeft < tMinRight; instIdx = lc ? l : r; if (tsp < f.tlasCap) tstk[tsp * WAVE] = lc ? r : l; else *s.overflow = 1u; tsp++; }
                            else instIdx = tl ? l : r;
                        } else { if (tsp == 0 || tsp > f.tlasCap) moreInst = false; else instIdx = tstk[--tsp * WAVE]; }
                    }
                }
                adv = active && !leafPending && top == 0u && moreInst;
            }
        } else if (MULTI) {
            bool adv = active && !leafPending && top == 0u && instIdx < (uint32_t)s.instanceCount;
            while (wballot(adv) != 0ull) {
                if (adv) {
                    const GpuBlasInstance in2 = s.instances[instIdx];
                    const M34 inv = load_inv_model(s, in2.MeshTransformId);
                    float4 a = tr.rec[4 * (size_t)rayId], b = tr.rec[4 * (size_t)rayId + 1];                         // world-space origin / direction
                    ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                    invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                    nodeOff = (uint32_t)s.descs[in2.BlasId].NodeOffset; triOff = (uint32_t)s.descs[in2.BlasId].TriangleOffset; xformId = in2.MeshTransformId;
                    const float4* root = s.nodes + 2 * (size_t)nodeOff + 2;
                    float t1;
                    const bool enter = RayBoxIntersect(ro, invDir, root[0], root[1], &t1) && t1 < hitT;
                    sp = 0; top = enter ? 2u : 0u;
                    instIdx++;
                }
                adv = active && !leafPending && top == 0u && instIdx < (uint32_t)s.instanceCount;
            }
        }

        // ---- C. leaf phase, when enough lanes cannot go on without it (or nobody can step)
        {
            const unsigned long long need = wballot(leafPending && (!SPEC || blocked || top == 0u));
            if (need != 0ull && ((int)__popcll(need) >= LEAF_MIN || wballot(can_step()) == 0ull)) {
                if (leafPending) {
                    for (uint32_t i = leafFirst; i < leafEnd; i++) {
                        const float4* tv = s.triVerts + 3 * (size_t)i;
                        float4 a = tv[0], b = tv[1], c = tv[2];
                        float by, bz, t;
                        if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hitT) {
                            hitTri = i; hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = MULTI ? xformId : inst.MeshTransformId;
                        }
                    }
                    leafPending = false; blocked = false;
                }
            }
        }

        // ---- D. one step: node steps + the ray fetches of the refilled lanes; the look-ahead loads issued in A are in flight in front of it
        node_step(std::true_type{}, can_step() && !take, take, takeIdx);

        // ---- D'. look-ahead loads issued in A have landed by now (they were issued before the step's loads)
        asm volatile("" : "+v"(nextIdx), "+v"(resvV));   // (unconditional: no pending load is carried around the loop, so nothing at its top has to wait)
        if (issuedAtomic) { resvBase = __builtin_amdgcn_readfirstlane(resvV); haveResv = true; }

        // ---- G. node loop: plain node steps until an event is due — enough lanes wait for the leaf phase / the next instance, enough
        //      lanes are free for a refill that can be served, or nobody can step
        const uint32_t availNow = curLen - curUsed;
        while (true) {
            const bool cs = can_step();
            if (wballot(cs) == 0ull) break;
            if ((int)__popcll(wballot(waiting())) >= LEAF_MIN) break;
            if (availNow != 0u && (int)__popcll(wballot(!active || finished())) >= FETCH_MIN) break;
            node_step(std::false_type{}, cs, false, 0u);
        }
    }
}
