// kernels_trace_inst.hpp — the reference's instance LOOP (several BLAS instances, no USE_TLAS: BVHIntersect.glsl:275-287, its default mode, Bvh/BVH.cs:156) walked through a
// TLAS the library builds for itself, with the loop's results.
// Part of the single translation unit idkpt.hip (included there, in this order).
//
// WHY.  The loop hands every ray to every instance in list order: a ray pays one root-box test per instance and walks each BLAS it touches with whatever T the instances
// before it happened to leave.  With the atrium as 87 BLASes (one per mesh) that is 454 Mray/s against 1 641 through a TLAS (profiles/r05_bench_line.json) — but a host that
// has not set UseTlas gets the loop's hits, and they are not the TLAS walk's: `t < T` keeps the FIRST of two equal hits, and first means list order in one, distance order
// in the other.
// WHAT.  k_tlas_build's PLOC tree (the reference's own builder, Bvh/TLAS.cs:28-141) over boxes that are the images of the BLAS root boxes under the inverse of InvModel —
// the matrix the loop itself transforms rays with — padded (kernels_scene.hpp, padded = 1).  k_trace_inst walks it front to back and enters an instance with the loop's own
// RayTransform and root test; inside a BLAS it is k_trace2's node / leaf phases.  Everything that produces a number is the loop's arithmetic on the loop's data; what the
// tree changes is the ORDER in which a ray meets its candidates and the T they are culled with.
// WHY THE RESULTS STAY THE LOOP'S (the argument of wide_nodes.hpp, whose constants these are):
//   * every cull — TLAS box, instance root box, BLAS child box — keeps a slack: a box is entered while t1 <= T * CULL (1 + 2^-14); T only falls, so a box culled at some
//     moment has t1 above the final T * CULL;
//   * the walk remembers the second-best hit distance among other triangles (starting from the ray's initial T) and t1 of the leaf box its best hit was found in;
//   * a ray is FLAGGED, and traced again by the exact loop (k_trace2 MODE 1 over the launch's list of flagged rays), if that second-best distance or that t1 lies inside
//     T * WINDOW (1 + 2^-16), if its best hit is a MARKED triangle (a PreSplit fragment, not contained in its leaf box: which copy of the triangle the loop reports depends on
//     its order), if a component of 1/dir is not finite for the world ray or for an instance it enters (NaN slabs), or if a stack overflowed.
//   For an unflagged ray with winner f at T: no other candidate lies within T * WINDOW (one would have been met: its boxes start at most t * (1 + a) <= T * CULL in front of
//   it — the ASSUMPTION of wide_nodes.hpp, `a` <= 3 * 2^-16), so whatever T the loop holds when it comes to an ancestor box of f is above T * WINDOW >= t1(leaf of f) >=
//   t1(every ancestor, by the monotonicity of IEEE subtraction and multiplication: ancestors contain the leaf box) — also its strict root test (:33-39) passes — it reaches f,
//   accepts it, and nothing it meets later has t < T: same TriangleId, T, barycentrics, MeshTransformId, bit for bit.  A ray without any hit met a superset of what the loop
//   meets.  Not covered, as there: the visit counters (DoDebugBVHTraversal, the counting build), any-hit queries (first found = list order) — those keep the loop.
//
// WHERE THE TREE'S LEAVES NEED NOT BE WHOLE INSTANCES (round 6, k_braid in kernels_scene.hpp; profiles/r06_braid.md).  A leaf may stand for a SUBTREE of an instance's BLAS: its record
// (InstTlasBufs::entRec) carries the node's own BLAS box and the child pair a walk entered there starts at.  The argument above does not care where a BLAS is entered as long as every
// box above the entry contains it (the loop's tests of those boxes pass whenever the entry's does): the host braids only scenes whose BLAS boxes nest.  Under this kernel's TLAS phase an
// entry costs a RayTransform, so more entries lose (option inst_braid, default 0).  Under UNI (below) — every instance carries the same InvModel, the whole scene is ONE tree in one
// BLAS space and an entry is just a node — it is the default for such scenes (option inst_unify): the atrium as 87 BLASes 1 756 -> 3 477 Mray/s.
//
// THE FLAGGED RAYS' LOOP (k_trace_inst<P, EXACT = true>).  The exact loop over 87 instances is 87 x three dependent fetches per ray before anything is walked, and a launch of a
// few thousand flagged rays lasts as long as that chain — which a frame traced alone pays twice.  The EXACT instantiation is the loop itself with one thing hoisted: when a wave
// takes new rays it runs over the instance records once, wave-uniformly (scalar loads, no dependent chain), and every lane notes in a bit mask (LDS rows) which root boxes its ray
// meets at all (RayBoxIntersect's boolean, the loop's own expression on the loop's own operands: an instance whose box the ray does not meet is never entered by the loop, whatever T).
// Then the lane walks the set bits in list order with the loop's own entry — RayTransform, 1/dir, root test with the strict `t1 < T` — and k_trace2's node / leaf phases with the
// exact culls.  Nothing is approximated and nothing flagged: the same instances are entered in the same order with the same T.
#pragma once

// the general unified array's marked leaves (kernels_scene.hpp k_unify_top, general): a leaf's count word >= these is no triangle count
#define UNIFY_MARK_ENTRY 0x80000000u
#define UNIFY_MARK_RESTORE 0xC0000000u

struct InstTlasBufs {
    const float4* tlas;                  // the library's own TLAS (GpuTlasNode layout: root = 0, children adjacent, bit 31 of .w = leaf, the rest = child / ENTRY)
    const float4* entRec;                // the records of the tree's leaves, 6 x float4 each in DScene::instRec's layout: the instance records themselves (an entry = a whole instance), or
                                         // k_braid's (kernels_scene.hpp: an entry = a subtree of an instance's BLAS — its box, and in [5].z the node pair a walk entered through it starts at)
    const uint8_t* marks;                // per BLAS triangle (leaf order, scene-wide index): 1 = not contained in (one of) its leaf box(es)
    int tlasCap;                         // rows of the per-lane TLAS stack
    int maskWords;                       // EXACT: 32-bit words of a lane's instance mask (they use the TLAS rows: the host sizes LDS for the larger of the two)
    uint32_t* flagCount; uint32_t* flagA; uint32_t* flagB;    // this launch's list of flagged rays (kernels_wide.hpp has the same hand-over)
    unsigned long long* totals;          // [0] flagged rays since idkptResetStats
    // UNI (every instance has the same InvModel: one BLAS space, one tree — k_unify_top / k_unify_blas in kernels_scene.hpp)
    const float4* unodes;                // the unified node array (GpuBlasNode layout, root = node 1, children and leaf ranges scene-wide)
    uint32_t uniXformId;                 // a MeshTransformId whose InvModel is the instances' common one (instance 0's)
    int uniCap;                          // rows of the per-lane stack of this walk (the top's depth on top of the deepest BLAS's)
    uint32_t baseB, restoreIdx;          // TREE 2: where the BLAS region of the array starts; the RESTORE pair
    const uint32_t* blasTriStart; const uint32_t* blasXform; int blasCount;   // per BLAS in ascending order of TriangleOffset: that offset, and the MeshTransformId of the one instance that uses it
};

// marks[t] = 1 for every triangle of a leaf that the leaf's box does not contain.  One thread per BLAS node; chunk k of 256 nodes belongs to BLAS chunks[k].x and starts at
// its node chunks[k].y (BLAS sizes differ by orders of magnitude: the table is the flattened (BLAS, chunk) list, made once per upload).  marks is zeroed before.
__global__ __launch_bounds__(256) void k_mark_triangles(const float4* nodes, const float4* triVerts, const GpuBlasDesc* descs, const uint2* chunks, uint8_t* marks)
{
    const uint2 ch = chunks[blockIdx.x];
    const GpuBlasDesc d = descs[ch.x];
    const uint32_t n = ch.y + threadIdx.x;
    if (n < 2u || n >= (uint32_t)d.NodeCount) return;                        // (node 0 is padding, node 1 the root: never a leaf)
    const float4 bmin = nodes[2 * ((size_t)d.NodeOffset + n)], bmax = nodes[2 * ((size_t)d.NodeOffset + n) + 1];
    const uint32_t start = __float_as_uint(bmin.w), cnt = __float_as_uint(bmax.w);
    for (uint32_t t = 0; t < cnt; t++) {
        const size_t g = (size_t)d.TriangleOffset + start + t;
        bool inside = true;
        for (int v = 0; v < 3; v++) { const float4 p = triVerts[3 * g + v]; inside = inside && p.x >= bmin.x && p.x <= bmax.x && p.y >= bmin.y && p.y <= bmax.y && p.z >= bmin.z && p.z <= bmax.z; }
        if (!inside) marks[g] = 1;                                           // (NaN positions: marked)
    }
}

// UNI = true (EXACT = false): the same walk without a TLAS phase — every instance has the same InvModel, so a ray is taken into the one BLAS space ONCE, with the loop's own
// RayTransform, and walks the unified tree (kernels_scene.hpp k_unify_*: a PLOC top over subtrees of the BLASes, then the BLASes' own nodes) with k_trace2's node / leaf phases and the
// slack culls.  An entry costs nothing: it is a node.  Each BLAS belongs to one instance, so a scene-wide triangle index names its instance (looked up once, when the ray retires).
// TREE = 2 (EXACT = false): the instances carry DIFFERENT transforms, and the scene is still walked as one array (k_unify_top, general): the top over the entries' padded world
// boxes is walked with the WORLD ray; an entry is a marked leaf (kernels_scene.hpp UNIFY_MARK_ENTRY) whose "leaf phase" is the loop's entry — RayTransform with the entry record's
// InvModel, 1 / dir, the entry's own BLAS box with the slack cull — after which the lane saves the node it was about to visit, pushes the RESTORE pair and continues in the entry's
// BLAS (rebased copy) with the instance-space ray; popping the RESTORE pair (UNIFY_MARK_RESTORE) puts the world ray back.  No TLAS phase, no second stack, one dependent fetch per
// step everywhere; an entry costs one stub step and the transform.  A BLAS may be used by several instances (MeshTransformId is per lane, as in the own-TLAS walk).
template <bool PRIMARY, bool EXACT = false, int REFILL_MIN = 16, int TREE = 0 /* 0: the own TLAS + the BLASes; 1: the unified tree of a same-space scene; 2: the general unified array */>
__global__ __launch_bounds__(WAVE, 1) void k_trace_inst(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter, InstTlasBufs ib)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    // LDS rows as in k_trace2: row 0 = dummy, rows 1 .. cap = BLAS stack, row cap + 1 = spare, then the TLAS rows
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const stkBase = (lds_u32*)lds + lane;
    constexpr bool UNI = TREE == 1, GEN = TREE == 2, ONE = TREE != 0;
    static_assert(!(ONE && EXACT), "k_trace_inst: the unified trees are walks that flag; the exact loop is the loop");
    const int cap = ONE ? ib.uniCap : f.stackCap;
    lds_u32* const stkFull = stkBase + cap * WAVE;
    uint32_t* const tstk = lds + lane + (cap + 2) * WAVE;
    const float4* const walkNodes = ONE ? ib.unodes : s.nodes;
    M34 uniInv; uniInv.r0 = uniInv.r1 = uniInv.r2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (UNI) uniInv = load_inv_model_at(s.xforms, ib.uniXformId);          // (wave-uniform: the instances' common InvModel)
    const uint32_t N = *countPtr;
    {
        uint32_t want = gridDim.x;
        if (f.gridRaysX4 > 0u) want = max((uint32_t)(((unsigned long long)N * 4ull / f.gridRaysX4 + 63ull) / 64ull), min(want, 1024u));
        if (f.gridMid > 0u && N < f.gridMidRays) want = min(want, f.gridMid);
        if (blockIdx.x >= max(want, 1u)) return;
    }
    bool active = false, leafPending = false, workLeft = N != 0u;
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    uint32_t top = 0, slot = 0, rayIdx = 0, leafFirst = 0, leafEnd = 0, leftEnd = 0, rightStart = 0;
    uint32_t tnode = 0, nodeOff = 0, triOff = 0, xformId = 0;
    int tsp = 0; bool moreInst = false;
    lds_u32* sp = stkBase;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, cullT = 0.0f, second = 0.0f, bestLeafT1 = 0.0f, tL = 0.0f, tR = 0.0f, hbx = 0.0f, hby = 0.0f;
    uint32_t hitTri = ~0u, hitXform = 0, flags = 0;

    while (true) {
        // ---- retire finished rays: store the hit, or hand the ray to the exact loop
        {
            const bool done = active && top == 0u && !leafPending && !moreInst;
            if (__builtin_amdgcn_ballot_w64(done) != 0ull) {
                bool flagged = false;
                if (done) {
                    const float win = hitT * wide::WINDOW;
                    flagged = !EXACT && (flags != 0u || (hitTri != ~0u && (second <= win || bestLeafT1 > win || ib.marks[hitTri] != 0)));
                    if (UNI && !flagged && hitTri != ~0u) {          // whose triangle it is: the last BLAS whose triangles start at or before it
                        int lo = 0, hi = ib.blasCount;
                        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ib.blasTriStart[mid] <= hitTri) lo = mid; else hi = mid; }
                        hitXform = ib.blasXform[lo];
                    }
                    if (!flagged) store_hit(hits, slot, hitT, hbx, hby, hitTri, hitXform);
                    if (EXACT && flags != 0u) *s.overflow = 1u;          // (a dropped push: reported like k_trace2's, the upload-time validation makes it unreachable)
                    active = false;
                }
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(flagged);
                if (fm != 0ull) {
                    const uint32_t cntF = (uint32_t)__builtin_popcountll(fm);
                    uint32_t base = 0;
                    if (lane == (uint32_t)__builtin_ctzll(fm)) { base = atomicAdd(ib.flagCount, cntF); atomicAdd(ib.totals, (unsigned long long)cntF); }
                    base = (uint32_t)__shfl((int)base, __builtin_ctzll(fm));
                    if (flagged) {
                        const uint32_t at = base + (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1ull));
                        if (PRIMARY) ib.flagA[at] = rayIdx; else { ib.flagA[at] = slot; ib.flagB[at] = rayIdx; }
                    }
                }
            }
        }
        // ---- refill idle lanes (k_trace2's sliced work list, kernels_trace.hpp)
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
            const bool fresh = !active && item < N;
            if (fresh) {
                const bool ordered = !PRIMARY && tr.order != nullptr;          // (the launch over a wide / own-TLAS launch's flagged rays: position -> queue slot and ray id, pt_kernels.hpp TraceBufs)
                const uint32_t idx = ordered ? tr.orderIdx[item] : list[item];
                rayIdx = idx; slot = PRIMARY ? idx : (ordered ? tr.order[item] : item);
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f; flags = 0u; bestLeafT1 = 0.0f;
                if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                    float4 o = rays.o_ior[idx];
                    f3 wd = DecodeUnitVec(rays.thr_px[idx].w, rays.rad_py[idx].w), wo = mk3(o.x, o.y, o.z);
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                if (EXACT && f.queryMode) { hitT = tr.rec[4 * (size_t)idx + 1].w; hitXform = __float_as_uint(tr.rec[4 * (size_t)idx + 2].w); }   // idkptTraceRays (kernels_query.hpp k_query_prepare): T = maxDist or the nearest light, and that light
                second = hitT; cullT = EXACT ? hitT : hitT * wide::CULL;
                const float4 c = tr.rec[4 * (size_t)idx + 2];                  // world 1/dir (written for this walk: Frame::instTlas)
                const bool finite = EXACT || (gabs(c.x) < __builtin_inff() && gabs(c.y) < __builtin_inff() && gabs(c.z) < __builtin_inff());
                if (!finite) flags = 1u;
                active = true; leafPending = false; sp = stkBase; top = 0u; tsp = 0; tnode = 0u; moreInst = finite;
                if (GEN) {
                    // the top is walked with the world ray (its 1 / dir was written by the ray's producer: Frame::instTlas); entries switch to their instance's space
                    const float4 a = tr.rec[4 * (size_t)idx], b = tr.rec[4 * (size_t)idx + 1];
                    ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z);
                    nodeOff = 0u; triOff = 0u; xformId = 0u; moreInst = false; top = finite ? 2u : 0u;
                }
                if (UNI) {
                    // the loop's entry (BVHIntersect.glsl:281-282), once: every instance's InvModel is this one.  Root tests are skipped (a superset of what the loop enters)
                    const float4 a = tr.rec[4 * (size_t)idx], b = tr.rec[4 * (size_t)idx + 1];
                    ro = xform34(uniInv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(uniInv, mk3(b.x, b.y, b.z), 0.0f);
                    invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                    const bool fin = gabs(invDir.x) < __builtin_inff() && gabs(invDir.y) < __builtin_inff() && gabs(invDir.z) < __builtin_inff();
                    if (!fin) flags |= 1u;
                    nodeOff = 0u; triOff = 0u; xformId = 0u; moreInst = false; top = (finite && fin) ? 2u : 0u;
                }
            }
            if (EXACT && __builtin_amdgcn_ballot_w64(fresh) != 0ull) {
                // which root boxes the new rays meet at all: one wave-uniform pass over the instance records (see the header)
                f3 wo = splat3(0.0f), wd = splat3(0.0f);
                if (fresh) { const float4 a = tr.rec[4 * (size_t)rayIdx], b = tr.rec[4 * (size_t)rayIdx + 1]; wo = mk3(a.x, a.y, a.z); wd = mk3(b.x, b.y, b.z); }
                const int n = s.instanceCount;
                for (int w = 0; w < ib.maskWords; w++) {
                    uint32_t m = 0u;
                    const int last = min(n, 32 * (w + 1));
                    for (int i = 32 * w; i < last; i++) {
                        const float4* ir = s.instRec + 6 * (size_t)i;
                        M34 inv; inv.r0 = ir[0]; inv.r1 = ir[1]; inv.r2 = ir[2];
                        const float4 rootMin = ir[3], rootMax = ir[4];
                        const f3 lo = xform34(inv, wo, 1.0f), ld = xform34(inv, wd, 0.0f);
                        float t1;
                        if (RayBoxIntersect(lo, mk3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z), rootMin, rootMax, &t1)) m |= 1u << (i & 31);
                    }
                    if (fresh) tstk[w * WAVE] = m;
                }
            }
        }
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        // ---- EXACT: lanes whose current BLAS is exhausted take the next instance of their mask, in list order, with the loop's own entry (BVHIntersect.glsl:277-286, :32-39)
        if (EXACT) {
            bool adv = active && !leafPending && top == 0u && moreInst;
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(adv)) < (uint32_t)f.advMin && __builtin_amdgcn_ballot_w64(active && (leafPending || top != 0u)) != 0ull) adv = false;
            while (__any(adv)) {
                if (adv) {
                    uint32_t w = tnode >> 5;                                               // tnode: the first instance index not yet looked at
                    uint32_t m = w < (uint32_t)ib.maskWords ? tstk[w * WAVE] & (0xffffffffu << (tnode & 31u)) : 0u;
                    while (m == 0u && ++w < (uint32_t)ib.maskWords) m = tstk[w * WAVE];
                    if (m == 0u) moreInst = false;
                    else {
                        const uint32_t id = (w << 5) + (uint32_t)__builtin_ctz(m);
                        tnode = id + 1u;
                        const float4* ir = s.instRec + 6 * (size_t)id;
                        M34 inv; inv.r0 = ir[0]; inv.r1 = ir[1]; inv.r2 = ir[2];
                        const float4 rootMin = ir[3], rootMax = ir[4];
                        nodeOff = __float_as_uint(rootMin.w); triOff = __float_as_uint(rootMax.w); xformId = __float_as_uint(ir[5].x);
                        float4 a = tr.rec[4 * (size_t)rayIdx], b = tr.rec[4 * (size_t)rayIdx + 1];
                        ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                        invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        float t1;
                        const bool enter = RayBoxIntersect(ro, invDir, rootMin, rootMax, &t1) && t1 < hitT;
                        sp = stkBase; top = enter ? 2u : 0u;
                    }
                }
                adv = active && !leafPending && top == 0u && moreInst;
            }
        } else if (!ONE)
        // ---- TLAS walk: lanes whose current BLAS is exhausted go on until they reach the next instance they enter, or the end
        {
            bool adv = active && !leafPending && top == 0u && moreInst;
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(adv)) < (uint32_t)f.advMin && __builtin_amdgcn_ballot_w64(active && (leafPending || top != 0u)) != 0ull) adv = false;
            while (__any(adv)) {
                if (adv) {
                    const float4 pmin = ib.tlas[2 * (size_t)tnode];
                    const uint32_t packed = __float_as_uint(pmin.w), id = packed & 0x7fffffffu;
                    if ((packed >> 31) == 1u) {                                             // an instance: the loop's body (BVHIntersect.glsl:277-286, :32-39)
                        const float4* ir = ib.entRec + 6 * (size_t)id;                      // the entry's record: an instance (pt_kernels.hpp DScene::instRec) or one of its subtrees (k_braid)
                        M34 inv; inv.r0 = ir[0]; inv.r1 = ir[1]; inv.r2 = ir[2];
                        const float4 rootMin = ir[3], rootMax = ir[4], ids = ir[5];
                        nodeOff = __float_as_uint(rootMin.w); triOff = __float_as_uint(rootMax.w); xformId = __float_as_uint(ids.x);
                        float4 a = tr.rec[4 * (size_t)rayIdx], b = tr.rec[4 * (size_t)rayIdx + 1];
                        ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                        invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        float t1;
                        const bool enter = RayBoxIntersect(ro, invDir, rootMin, rootMax, &t1) && t1 <= cullT;
                        const bool finite = gabs(invDir.x) < __builtin_inff() && gabs(invDir.y) < __builtin_inff() && gabs(invDir.z) < __builtin_inff();
                        sp = stkBase; top = (enter && finite) ? __float_as_uint(ids.z) : 0u;      // (the root's children, or the children of the subtree's node)
                        if (tsp == 0) moreInst = false; else tnode = tstk[--tsp * WAVE];
                        if (!finite) { flags |= 1u; moreInst = false; }                     // (the exact loop traces this ray: nothing more to do for it here)
                    } else {
                        const uint32_t l = id, r = id + 1;
                        const float4* w4 = tr.rec + 4 * (size_t)rayIdx;
                        float4 a = w4[0], c = w4[2];
                        const f3 wo = mk3(a.x, a.y, a.z), winv = mk3(c.x, c.y, c.z);
                        float4 lmin = ib.tlas[2 * (size_t)l], lmax = ib.tlas[2 * (size_t)l + 1], rmin = ib.tlas[2 * (size_t)r], rmax = ib.tlas[2 * (size_t)r + 1];
                        float tMinLeft, tMinRight;
                        const bool tl = RayBoxIntersect(wo, winv, lmin, lmax, &tMinLeft) && tMinLeft <= cullT;
                        const bool tr2 = RayBoxIntersect(wo, winv, rmin, rmax, &tMinRight) && tMinRight <= cullT;
                        if (tl || tr2) {
                            if (tl && tr2) {
                                const bool lc = tMinLeft < tMinRight; tnode = lc ? l : r;
                                if (tsp < ib.tlasCap) { tstk[tsp * WAVE] = lc ? r : l; tsp++; } else { flags |= 2u; moreInst = false; }
                            } else tnode = tl ? l : r;
                        } else { if (tsp == 0) moreInst = false; else tnode = tstk[--tsp * WAVE]; }
                    }
                }
                adv = active && !leafPending && top == 0u && moreInst;
            }
        }

        // ---- node phase (k_trace2's branch-free step, culled with the slack)
        while (true) {
            const bool canStep = active && !leafPending && top != 0u;
            const unsigned long long stepMask = __builtin_amdgcn_ballot_w64(canStep);
            if (stepMask == 0ull) break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && leafPending)) >= f.leafMin) break;
            if (canStep) {
                const float4* p = walkNodes + 2 * ((size_t)nodeOff + top);
                const uint32_t popped = sp[0];
                float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
                const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
                float tMinLeft, tMinRight;
                const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= cullT;
                const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= cullT;
                const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
                leafFirst = intersectLeft ? lStart : rStart; leafEnd = !intersectRight ? lStart + lCount : rStart + rCount; leafPending = intersectLeft || intersectRight;
                // which leaf box(es) a triangle of the range [leafFirst, leafEnd) was reached through: the left one below leftEnd, the right one from rightStart on (a pair may share one)
                leftEnd = intersectLeft ? lStart + lCount : 0u; rightStart = intersectRight ? rStart : 0xffffffffu; tL = tMinLeft; tR = tMinRight;
                const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
                const bool both = traverseLeft && traverseRight, none = !(traverseLeft || traverseRight);
                const bool leftCloser = tMinLeft < tMinRight;
                const uint32_t nearChild = both ? (leftCloser ? lStart : rStart) : (traverseLeft ? lStart : rStart);
                sp[WAVE] = leftCloser ? rStart : lStart;
                const bool full = sp == stkFull, nonEmpty = sp != stkBase;
                flags |= (both && full) ? 2u : 0u;                       // (the push is dropped: the exact loop traces this ray)
                top = none ? (nonEmpty ? popped : 0u) : nearChild;
                sp += (both && !full) ? (int)WAVE : ((none && nonEmpty) ? -(int)WAVE : 0);
            }
        }
        // ---- leaf phase
        if (GEN && leafPending && leafEnd - leafFirst >= UNIFY_MARK_ENTRY) {
            if (leafEnd - leafFirst >= UNIFY_MARK_RESTORE) {
                // the instance's subtree is done: back to the world ray (the nodes below this stack entry are the top's)
                const float4 a = tr.rec[4 * (size_t)rayIdx], b = tr.rec[4 * (size_t)rayIdx + 1], c = tr.rec[4 * (size_t)rayIdx + 2];
                ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z);
            } else {
                // an entry: the loop's body for its instance (BVHIntersect.glsl:277-286, :32-39 on the entry's own box)
                const float4* ir = ib.entRec + 6 * (size_t)leafFirst;
                M34 inv; inv.r0 = ir[0]; inv.r1 = ir[1]; inv.r2 = ir[2];
                const float4 boxMin = ir[3], boxMax = ir[4], ids = ir[5];
                const f3 lo = xform34(inv, ro, 1.0f), ld = xform34(inv, rd, 0.0f);                 // (ro / rd hold the world ray here: entries are met from the top only)
                const f3 li = mk3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
                float t1;
                const bool enter = RayBoxIntersect(lo, li, boxMin, boxMax, &t1) && t1 <= cullT;
                const bool fin = gabs(li.x) < __builtin_inff() && gabs(li.y) < __builtin_inff() && gabs(li.z) < __builtin_inff();
                if (!fin) flags |= 1u;                                                          // (the exact loop traces this ray)
                if (enter && fin) {
                    // what the lane was about to visit in the top (its node step already chose it) waits under the RESTORE pair
                    const bool later = top != 0u || sp != stkBase;
                    const int need = (top != 0u ? 1 : 0) + (later ? 1 : 0);
                    if (sp + need * (int)WAVE > stkFull) flags |= 2u;                            // (no room: the exact loop traces this ray)
                    else {
                        if (top != 0u) { sp[WAVE] = top; sp += WAVE; }
                        if (later) { sp[WAVE] = ib.restoreIdx; sp += WAVE; }
                        ro = lo; rd = ld; invDir = li; xformId = __float_as_uint(ids.x);
                        top = ib.baseB + __float_as_uint(boxMin.w) + __float_as_uint(ids.z);      // the entry node's child pair, in the BLAS region of the array
                    }
                }
            }
            leafPending = false;
        }
        if (leafPending) {
            for (uint32_t i = leafFirst; i < leafEnd; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)(i + triOff);
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t)) {
                    const uint32_t id = i + triOff;
                    const bool other = id != hitTri || xformId != hitXform;     // (two instances may share a BLAS: the same triangle index under another transform is another candidate)
                    const float leafT1 = gmin(i < leftEnd ? tL : __builtin_inff(), i >= rightStart ? tR : __builtin_inff());
                    // the argument's one assumption, checked on every met hit as kernels_wide.hpp does: a triangle its leaf box contains is never hit more than 3 * 2^-16 in front of the box's
                    // entry.  (A marked triangle may be — and is flagged when it wins; an unmarked one that is — host-patched nodes with loose boxes — takes the ray to the exact loop.)
                    if (!EXACT && leafT1 > t * wide::ASSUME && ib.marks[id] == 0) flags |= 4u;
                    if (t < hitT) {
                        if (other) second = gmin(second, hitT);
                        hitT = t; cullT = EXACT ? t : t * wide::CULL; hbx = 1.0f - by - bz; hby = by; hitTri = id; hitXform = xformId;
                        bestLeafT1 = leafT1;
                    } else if (other) second = gmin(second, t);
                }
            }
            leafPending = false;
        }
    }
}
