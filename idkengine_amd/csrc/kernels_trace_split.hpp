// kernels_trace_split.hpp — k_trace2s: k_trace2 (MODE 0: one BLAS instance, no visit counters) + intra-wave SPLITTING of long rays, for launches that cannot
// fill the chip.  Part of the single translation unit idkpt.hip (included after kernels_trace.hpp).
//
// Why.  A launch with fewer rays than the chip has lanes (one frame at a time: 0.36 M + 0.28 M rays against 0.39 M lanes; one rank's share of an N-GPU frame)
// ends when its LONGEST ray ends: a dependent chain of up to ~630 node fetches at ~1 us each, run by one lane while the other 63 lanes of its wave (and most
// of the chip) idle.  Here, once the work list is exhausted, an idle lane takes over the BOTTOM entry of a busy lane's traversal stack — the subtree that lane
// would have visited last — and traverses it as an independent PIECE of the same ray, with the donor's current T as its bound.  Pieces report back to the lane
// they split from; the root lane of a ray combines them and writes the hit.  (Round 2 measured the idea on the kernel of that time: +16 % on the headline view
// one frame at a time, nothing where every pixel traverses; this is that schedule on round 3's kernel — eight work-list counters, branch-free node step, the
// stack pointer an LDS address — selected by the host for small launches of sparse views only, idkpt.hip want_split: +8 % on the headline view one frame at a
// time, +12 % on one rank's share of an 8-GPU frame, but 123 VGPRs (4 instead of 6 waves per SIMD) and nothing to gain where every pixel traverses connected
// geometry: -10 % on the atrium.  A variant with the rarely used state in LDS and a fast path for waves without pieces (108 / 96 VGPRs) was measured as well: it
// halves the loss on large launches and the gain on small ones, profiles/r04_small_launch_experiments.md.)
//
// Why the result is the reference's (BVHIntersect.glsl:27-105).  Sequentially the donated subtree is traversed AFTER everything the donor still holds, with bound
// T_seq = the donor's T at that later time <= the bound T0 the piece starts with.  Box tests and the near/far order do not depend on T except through
// `tMin <= T` / `t < T`, so by induction the piece's bound stays >= the sequential bound at every corresponding step and the piece visits a SUPERSET of what the
// sequential traversal visits in that subtree, in the same order.  Hence (1) if the piece finds nothing better than what is ahead of it in order, neither does
// the sequential traversal; (2) if the piece's best hit (t_w, the first in order among equals) wins the combination, the sequential traversal finds the same
// triangle provided every box on the path from the piece's root to that leaf passes `tMin <= bound_seq`.  tMin grows monotonically from a box to a box nested
// in it when 1/dir is finite (every operation of RayBoxIntersect is monotone in the box coordinates), so the largest tMin on the path is the leaf box's own,
// tMinLeaf.  The sequential bound differs from the piece's only through hits of OTHER pieces, so the one case where the two can disagree is
//        some other piece's best t  <  tMinLeaf of the winner     (two hits within rounding distance of each other: a shared edge, coincident surfaces),
// or two pieces with exactly equal t (order between sibling pieces is not tracked).  The root lane detects exactly that (`second < leafTmin || tie`) and then
// simply traces the ray again by itself, sequentially.  Rays whose 1/dir is not finite never split, and the host only selects this kernel when every BLAS of
// the scene is nested (children inside parents; checked at upload and after node patches, refits keep it by construction).  Own hits of the root piece are first
// in order and need no check.  Visit counters differ from the sequential ones, so the counting build (k_trace2<.., COUNT = true>) never splits.
#pragma once

template <bool PRIMARY, int REFILL_MIN = 32>
__global__ __launch_bounds__(WAVE, 1) void k_trace2s(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const stkBase = (lds_u32*)lds + lane;                   // rows as in k_trace2: row 0 dummy, rows 1 .. cap the entries, row cap + 1 spare
    const int cap = f.stackCap;
    lds_u32* const stkFull = stkBase + cap * WAVE;
    const uint32_t N = *countPtr;
    {   // waves beyond what the launch's actual ray count wants retire before they touch the work list (k_trace2)
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

    bool active = false, leafPending = false, workLeft = true, needLoad = false;
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0, peekTick = 0, peekHead = 0;
    bool peekPending = false;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    // Scattered hand-out (Frame::scatterLog2 < 6): the list is in tile order (k_gen_primary: 8x8 pixels are 64 consecutive entries), so the rays that graze the scene for
    // hundreds of steps come in runs — a wave that gets 64 of them has no idle lane to split onto while the waves next to it sit idle.  Viewed as a matrix of
    // (64 >> scatterLog2) rows of groups of 2^scatterLog2 entries, the list is handed out column by column: a wave's 64 positions are 64 >> scatterLog2 groups from
    // places far apart.  Positions run over the padded matrix (Nh >= N); which lane traces which entry is free, so nothing of the result changes.
    const uint32_t scat = (uint32_t)f.scatterLog2 < 6u ? (uint32_t)f.scatterLog2 : 6u;
    const uint32_t scatCols = (((N + (1u << scat) - 1u) >> scat) + (64u >> scat) - 1u) / (64u >> scat);      // groups per row
    const uint32_t Nh = scat < 6u ? (scatCols * (64u >> scat)) << scat : N;
    const uint32_t nBlocks = (Nh + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    if (N == 0u) workLeft = false;
    uint32_t top = 0, slot = 0, rayIdx = 0, leafFirst = 0, leafEnd = 0, leafSplit = 0;
    lds_u32* sp = stkBase;                      // this lane's stack entries are the rows (lo, sp]; a donation raises lo
    lds_u32* lo = stkBase;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, hbx = 0.0f, hby = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;
    float leafTminA = 0.0f, leafTminB = 0.0f;   // tMin of the parked leaf boxes: triangles below leafSplit / from leafSplit on
    float hitLeafTmin = 0.0f;                   // tMin of the leaf box the current own hit came from
    bool ovf = false;
    // splitting state
    int parent = -1;                            // lane this piece reports to (-1: root piece of its ray)
    int kids = 0;                               // pieces split off this lane that have not reported yet
    bool ownDone = false, splittable = false;
    float kT = INF, kbx = 0.0f, kby = 0.0f, kLeafTmin = 0.0f; uint32_t kTri = ~0u;   // best hit reported by the pieces below this lane
    float second = INF; bool tie = false;       // smallest best-t of any piece other than the one holding kT; two pieces with equal best t

    while (true) {
        // ---- refill idle lanes from the work list (k_trace2's hand-out over GRAB_SLICES counters)
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
            uint32_t item = valid ? ((((q >> unitLog2) * GRAB_SLICES + sl) << unitLog2) | (q & ((1u << unitLog2) - 1u))) : N;
            if (scat < 6u && item < Nh) { const uint32_t gi = item >> scat, per = 64u >> scat; item = ((((gi & (per - 1u)) * scatCols) + gi / per) << scat) | (item & ((1u << scat) - 1u)); }
            if (slicesDone >= GRAB_SLICES && chunkNext >= chunkEnd) workLeft = false;
            if (!active && item < N) {
                const bool ordered = !PRIMARY && tr.order != nullptr;
                rayIdx = ordered ? tr.orderIdx[item] : list[item];
                slot = PRIMARY ? rayIdx : (ordered ? tr.order[item] : item);
                parent = -1; splittable = true; needLoad = true; active = true;
            }
        }
        // a wave that is too busy to refill still has to learn that the list is exhausted before it may split: its first eight lanes peek at the slices' head words.
        // The words are requested in one iteration and looked at in the next (the wave does not wait for the round trip), every f.splitPeek-th iteration (64) while at
        // least four lanes are idle.  Looking more often was measured and is much worse (every 8th iteration: -19 % on the headline view one frame at a time, -29 % with
        // three samples in flight): waves that start splitting while most of their lanes still carry rays of their own spend their time on pieces and bookkeeping.
        else if (workLeft && chunkNext >= chunkEnd && (uint32_t)__popcll(idle) >= 4u) {
            if (peekPending) {
                const uint32_t len = ((nBlocks + GRAB_SLICES - 1u - (lane & (GRAB_SLICES - 1u))) / GRAB_SLICES) << unitLog2;
                if (__ballot(lane >= GRAB_SLICES || peekHead >= len) == ~0ull) workLeft = false;
                peekPending = false;
            } else if (++peekTick >= (uint32_t)f.splitPeek) {
                peekTick = 0;
                if (lane < GRAB_SLICES) peekHead = __atomic_load_n(workCounter + GRAB_STRIDE * lane, __ATOMIC_RELAXED);
                peekPending = true;
            }
        }
        // ---- (re)start of a root piece: the ray's record -> registers (new rays, and the sequential re-trace of an inconclusive ray)
        if (active && needLoad) {
            needLoad = false;
            hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f; hitLeafTmin = 0.0f;
            if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                float4 o = rays.o_ior[rayIdx];
                f3 wd = DecodeUnitVec(rays.thr_px[rayIdx].w, rays.rad_py[rayIdx].w), wo = mk3(o.x, o.y, o.z);
                for (int i = 0; i < s.lightCount; i++) {
                    const GpuLight& l = s.lights[i];
                    float tMin, tMax;
                    if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                }
            }
            float rootT;
            { float4 a = tr.rec[4 * (size_t)rayIdx], b = tr.rec[4 * (size_t)rayIdx + 1], c = tr.rec[4 * (size_t)rayIdx + 2]; ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z); rootT = a.w; }
            const bool enter = rootT < hitT;   // root test (:32-39), see k_trace2
            leafPending = false; sp = stkBase; lo = stkBase; top = enter ? 2u : 0u;
            kids = 0; ownDone = false; kT = INF; kTri = ~0u; second = INF; tie = false;
            // pieces are only split off rays whose slab arithmetic is monotone in the box (finite 1/dir): see the header
            const float big = 3.0e38f;
            splittable = splittable && gabs(invDir.x) <= big && gabs(invDir.y) <= big && gabs(invDir.z) <= big;
        }
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        // ---- a piece keeps tightening its bound to the CURRENT T of the lane it split from.  Sequentially its subtree is traversed after everything that lane still
        // holds, i.e. with a bound <= that lane's T at any earlier time (T never grows), so the piece still visits a superset of the sequential visits — but no longer
        // everything a ray without a hit (T = FLOAT_MAX at the split) could reach: without this the pieces of such rays traverse whole subtrees the sequential walk
        // culls after its first near hit, and the root waits for them (measured: no gain at all without it, +8 % on the headline view one frame at a time with it).
        // A hit of the piece that the donor's T already beats (or equals: the donor's is first in order) can never win the combination and is dropped.
        if (!workLeft) {
            const float pT = __int_as_float(__builtin_amdgcn_ds_bpermute((parent < 0 ? (int)lane : parent) << 2, __float_as_int(hitT)));
            if (active && parent >= 0 && (pT < hitT || (pT == hitT && hitTri != ~0u))) { hitT = pT; hitTri = ~0u; }
        }

        // ---- split: idle lanes adopt the bottom stack entry of busy lanes (only once the work list is exhausted)
        idle = __ballot(!active);
        if (!workLeft && idle != 0ull) {
            // (splitMode bit 2: only rays that have not hit anything yet donate — the rays that cross the whole scene, whose far subtrees no later hit will cull)
            unsigned long long donors = __ballot(active && !ownDone && splittable && sp != lo && (!(f.splitMode & 4) || hitT == PT_FLOAT_MAX));
            int pairs = 0;
            while (donors != 0ull && idle != 0ull && pairs < 16) {
                const int d = (int)__builtin_ctzll(donors), i = (int)__builtin_ctzll(idle);
                donors &= donors - 1ull; idle &= idle - 1ull; pairs++;
                const uint32_t dlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)lo, d);          // the donor's bottom pointer (an LDS address)
                const float rox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ro.x), d)), roy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ro.y), d)), roz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ro.z), d));
                const float rdx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rd.x), d)), rdy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rd.y), d)), rdz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rd.z), d));
                const float ivx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invDir.x), d)), ivy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invDir.y), d)), ivz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invDir.z), d));
                const float bound = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hitT), d));
                if ((int)lane == i) {
                    ro = mk3(rox, roy, roz); rd = mk3(rdx, rdy, rdz); invDir = mk3(ivx, ivy, ivz);
                    hitT = bound; hitTri = ~0u; hbx = 0.0f; hby = 0.0f; hitXform = inst.MeshTransformId; hitLeafTmin = 0.0f;
                    top = ((lds_u32*)(uintptr_t)dlo)[WAVE];                       // the donor's bottom entry (the row above its bottom pointer)
                    sp = stkBase; lo = stkBase; leafPending = false; needLoad = false;
                    parent = d; kids = 0; ownDone = false; kT = INF; kTri = ~0u; second = INF; tie = false; splittable = true;
                    active = true;
                }
                if ((int)lane == d) { lo += WAVE; kids++; }
            }
        }

        // ---- node phase (k_trace2's branch-free step; "empty" is this lane's own bottom pointer)
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
                // (both leaves hit: the reference tests [lStart, rStart + rCount) in one loop, BVHIntersect.glsl:57-61)
                leafSplit = (intersectLeft && intersectRight) ? rStart : leafEnd;
                leafTminA = intersectLeft ? tMinLeft : tMinRight; leafTminB = tMinRight;
                const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
                const bool both = traverseLeft && traverseRight, none = !(traverseLeft || traverseRight);
                const bool leftCloser = tMinLeft < tMinRight;
                const uint32_t nearChild = both ? (leftCloser ? lStart : rStart) : (traverseLeft ? lStart : rStart);
                sp[WAVE] = leftCloser ? rStart : lStart;
                const bool full = sp == stkFull, nonEmpty = sp != lo;
                ovf = ovf || (both && full);
                top = none ? (nonEmpty ? popped : 0u) : nearChild;
                sp += (both && !full) ? (int)WAVE : ((none && nonEmpty) ? -(int)WAVE : 0);
            }
        }
        // ---- leaf phase
        if (leafPending) {
            for (uint32_t i = leafFirst + triOffset, e = leafEnd + triOffset, sL = leafSplit + triOffset; i < e; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)i;
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hitT) {
                    hitTri = i; hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = inst.MeshTransformId;
                    hitLeafTmin = i < sL ? leafTminA : leafTminB;
                }
            }
            leafPending = false;
        }
        // ---- a piece whose own traversal is over waits for the pieces split off it
        if (active && !ownDone && top == 0u && !leafPending && !needLoad) ownDone = true;

        // ---- pieces report to the lane they were split from (one at a time; only ever a few per iteration)
        {
            unsigned long long rep = __ballot(active && ownDone && kids == 0 && parent >= 0);
            while (rep != 0ull) {
                const int r = (int)__builtin_ctzll(rep); rep &= rep - 1ull;
                // a piece's result: its own hit is first in order among itself and its sub-pieces; a sub-piece's hit wins only if strictly nearer
                const float ownT = hitTri != ~0u ? hitT : INF;
                const bool kidsWin = kT < ownT;
                const float myT = kidsWin ? kT : ownT, mySecond = gmin(second, kidsWin ? ownT : kT);
                const float cT = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myT), r));
                const float cbx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kidsWin ? kbx : hbx), r)), cby = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kidsWin ? kby : hby), r));
                const uint32_t cTri = (uint32_t)__builtin_amdgcn_readlane((int)(kidsWin ? kTri : hitTri), r);
                const float cLeaf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kidsWin ? kLeafTmin : hitLeafTmin), r));
                const float cSecond = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mySecond), r));
                const int cTie = __builtin_amdgcn_readlane((int)tie, r);
                const int p = __builtin_amdgcn_readlane(parent, r);
                if ((int)lane == p) {
                    second = gmin(second, cSecond); tie = tie || cTie != 0;
                    if (cT < kT) { second = gmin(second, kT); kT = cT; kbx = cbx; kby = cby; kTri = cTri; kLeafTmin = cLeaf; }
                    else if (cT == kT && cT != INF) tie = true;
                    else second = gmin(second, cT);
                    kids--;
                }
                if ((int)lane == r) active = false;
            }
        }
        // ---- root pieces: combine and retire, or trace the ray again sequentially when the pieces are inconclusive (header comment)
        if (active && ownDone && kids == 0 && parent < 0) {
            const bool kidsWin = kT < hitT;                       // the own hit (also a light hit: T < MAX, TriangleId ~0) is first in order: ties stay with it
            const bool retrace = kidsWin && (tie || gmin(second, hitT) < kLeafTmin || (f.splitMode & 3) == 2);
            if (retrace) { splittable = false; needLoad = true; ownDone = false; }
            else {
                if (kidsWin) { hitT = kT; hbx = kbx; hby = kby; hitTri = kTri; hitXform = inst.MeshTransformId; }
                store_hit(hits, slot, hitT, hbx, hby, hitTri, hitXform);
                active = false;
            }
        }
    }
    if (ovf) *s.overflow = 1u;
}
