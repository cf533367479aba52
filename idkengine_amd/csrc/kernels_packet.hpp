// kernels_packet.hpp — k_trace_packet: a wave-uniform PACKET walk of the reference's BVH2 for the primary launches whose work list is pixel-major (k_gen_primary, kernels_trace.hpp:
// 64 consecutive entries = a few neighbouring pixels under all their samples — rays that differ by their sub-pixel jitter).
// Part of the single translation unit idkpt.hip (included there, in this order).
//
// WHY.  k_trace2 gives every lane its own ray, its own LDS stack and its own 64-byte gather per node step; measured (profiles/r05_wide_nodes.md §4) it is bound by
// instruction issue at 39 of 64 live lanes per step with 45 scalar instructions of bookkeeping on top of the 62 vector ones.  Where the 64 rays of a wave want the SAME nodes that is
// 64 copies of one walk.  tools/packet_sim.cpp (CPU model, the bench scenes at 1920x1080, 4 pixels x 16 samples per wave): a shared walk takes 0.66-0.70 x the wave steps of k_trace2
// on the views where every pixel traverses (interior 153 vs 230 per 64 rays, atrium 35 vs 53) with 0.87-0.88 of the wave's rays live in a step — and 1.14-1.57 x on the headline
// view, whose pixels are wider than its triangles (0.37-0.54 live): the host decides per view from what the kernel itself counts (host_launch.hpp: packet_decide).
//
// WHAT.  One wave = one packet of 64 consecutive list entries; no refill inside a packet.
//   * the walk's state is wave-uniform: the current node pair and the 64-bit mask of lanes it is live for in SGPRs, the stack in three VGPRs (lane k holds entry k: node, mask lo / hi;
//     push = a select on lane == sp, pop = v_readlane — no LDS, no per-lane stack pointer);
//   * the pair is fetched ONCE per wave through the scalar cache (s_load_dwordx16 from the constant address space: BLAS nodes are read-only inside a launch), a leaf's triangles
//     likewise (3 x s_load_dwordx4 each): a node step issues no vector-memory instruction at all;
//   * every live lane tests both boxes with its own ray and its own T — RayBoxIntersect's expression (IntersectionRoutines.glsl:25-40) on the reference's own boxes; a leaf child is
//     tested right away (BVHIntersect.glsl:54-79: leaves before the descent) by the lanes whose box test passed, with RayTriangleIntersect's expression (:6-23) in the leaf's order;
//   * both children internal and wanted: the side more lanes find nearer is walked first, the other is pushed with the mask of the lanes that hit it.
//
// WHY THE RESULTS STAY THE REFERENCE'S.  Every number is the reference's arithmetic on the reference's data; the packet only changes the ORDER in which a ray meets its candidates and
// the T they are culled with — the situation of the wide-node walk and of k_trace_inst, and the same machinery answers it (wide_nodes.hpp, whose constants and argument these are, with
// "wide child box" = the exact BVH2 box):
//   * every cull keeps a slack: a box is entered while t1 <= T * CULL (1 + 2^-14); T only falls, so a box culled at some moment has t1 above the final T * CULL;
//   * a lane remembers the second-best hit distance among other triangle ids (starting from its initial T) and t1 of the leaf box its best hit was found in;
//   * a ray is FLAGGED — appended to the launch's list and traced again by the exact kernel, k_trace2<PRIMARY>, right behind this launch — if that second-best distance or that t1 lies
//     inside T * WINDOW (1 + 2^-16), if its best hit is a MARKED triangle (a PreSplit fragment: not contained in its leaf box; which copy the reference reports depends on its order;
//     InstTlasBufs::marks, k_mark_triangles), if a tested triangle violates the argument's one ASSUMPTION (a hit more than 3 * 2^-16 in front of its own leaf box's entry), if one of
//     its box tests failed by less than 2^-20 relative (NEAR_MISS), if a component of 1/dir is not finite (NaN slabs), or if the packet's stack overflowed (64 entries).
//   Not covered, as there: the reference's visit counters (DoDebugBVHTraversal, the counting build), any-hit queries, scene versions, several instances — those keep k_trace2.
#pragma once

struct PacketBufs {
    const uint8_t* marks;                // per BLAS triangle (leaf order, scene-wide index): 1 = not contained in its leaf box (k_mark_triangles, kernels_trace_inst.hpp)
    uint32_t* flagCount; uint32_t* flagA;   // this launch's list of flagged rays (ray ids; kernels_wide.hpp has the same hand-over)
    unsigned long long* totals;          // [0] flagged rays since idkptResetStats, [1] packets, [2] node steps, [3] live lanes summed over the node steps, [4] rays that entered, [5] triangle rounds
    // UNI: the unified tree of a same-space multi-instance scene (kernels_trace_inst.hpp InstTlasBufs, kernels_scene.hpp k_unify_*) instead of instance 0's BLAS
    const float4* unodes; uint32_t uniXformId; const uint32_t* blasTriStart; const uint32_t* blasXform; int blasCount;
};

typedef uint32_t pk_u16v __attribute__((ext_vector_type(16)));
typedef uint32_t pk_u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const pk_u16v pk_c16;
typedef __attribute__((address_space(4))) const pk_u4v pk_c4;

// RayBoxIntersect (pt_device.hpp) on a box whose bounds are wave-uniform: the same operations in the same order on the same operands (IEEE, no contraction) — bit-identical t1 / t2
// (x / y slabs through 2-wide vectors: v_pk_add_f32 / v_pk_mul_f32 with the box's adjacent SGPR pair as one operand — two IEEE operations per instruction, same roundings)
DEV void pk_box(const f3& o, const f3& inv, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float* t1, float* t2)
{
    const v2f oxy = {o.x, o.y}, ixy = {inv.x, inv.y};
    const v2f a2 = (v2f{mnx, mny} - oxy) * ixy, b2 = (v2f{mxx, mxy} - oxy) * ixy;
    const float az = (mnz - o.z) * inv.z, bz = (mxz - o.z) * inv.z;
    const float sx = gmin(a2.x, b2.x), sy = gmin(a2.y, b2.y), sz = gmin(az, bz);
    const float gx = gmax(a2.x, b2.x), gy = gmax(a2.y, b2.y), gz = gmax(az, bz);
    *t1 = gmax(sx, gmax(sy, gmax(sz, 0.0f)));
    *t2 = gmin(gx, gmin(gy, gz));
}

// Both boxes of a pair at once: x / y of each box as above, the z slabs of the LEFT and the RIGHT box through one 2-wide vector each (the pair's two z bounds are not adjacent in
// the node — two scalar moves make them a register pair; scalar issue has room, vector issue is what binds this walk).  Same operations on the same operands: bit-identical.
DEV void pk_box2(const f3& o, const f3& inv, const pk_u16v& P, float* t1L, float* t2L, float* t1R, float* t2R)
{
    const v2f oxy = {o.x, o.y}, ixy = {inv.x, inv.y}, ozz = {o.z, o.z}, izz = {inv.z, inv.z};
#define PKF(i) __uint_as_float(P[i])
    const v2f aL = (v2f{PKF(0), PKF(1)} - oxy) * ixy, bL = (v2f{PKF(4), PKF(5)} - oxy) * ixy;
    const v2f aR = (v2f{PKF(8), PKF(9)} - oxy) * ixy, bR = (v2f{PKF(12), PKF(13)} - oxy) * ixy;
    const v2f az = (v2f{PKF(2), PKF(10)} - ozz) * izz, bz = (v2f{PKF(6), PKF(14)} - ozz) * izz;
#undef PKF
    *t1L = gmax(gmin(aL.x, bL.x), gmax(gmin(aL.y, bL.y), gmax(gmin(az.x, bz.x), 0.0f)));
    *t2L = gmin(gmax(aL.x, bL.x), gmin(gmax(aL.y, bL.y), gmax(az.x, bz.x)));
    *t1R = gmax(gmin(aR.x, bR.x), gmax(gmin(aR.y, bR.y), gmax(gmin(az.y, bz.y), 0.0f)));
    *t2R = gmin(gmax(aR.x, bR.x), gmin(gmax(aR.y, bR.y), gmax(az.y, bz.y)));
}

// The walk is written in the MASK domain: which lanes a decision holds for is a wave-uniform 64-bit word (one v_cmp writes it, s_and / s_or combine it, __builtin_amdgcn_inverse_ballot_w64
// hands it back to a select as its lane mask), never a per-lane bool that the compiler would have to carry through exec-mask regions; everything per lane is a select under full exec.
#define PK_BALLOT(c) __builtin_amdgcn_ballot_w64(c)
#define PK_LANES(m) __builtin_amdgcn_inverse_ballot_w64(m)
// UNI: the walk over the unified tree of a scene whose instances all carry the same InvModel (one BVH2 in their common BLAS space: k_trace_inst's header, k_unify_* in kernels_scene.hpp).
// A ray is taken into that space here, with the instance loop's own RayTransform arithmetic (BVHIntersect.glsl:281-282; its producers left the world ray in the record), every root
// test is skipped (a superset of what the loop enters), triangle indices are scene-wide and name their instance (looked up when a hit is stored); the rays this walk does not vouch
// for go to the exact LOOP (k_trace_inst<true, EXACT>), whose hits these are.
template <bool STATS, bool UNI = false>
__global__ __launch_bounds__(WAVE) void k_trace_packet(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter, PacketBufs pb)
{
    typedef unsigned long long u64;
    const uint32_t lane = threadIdx.x;
    const uint32_t N = *countPtr;
    const GpuBlasInstance inst = s.instances[0];
    const uint32_t triOffset = UNI ? 0u : (uint32_t)s.descs[inst.BlasId].TriangleOffset;
    const float4* const nodes = UNI ? pb.unodes : s.nodes + 2 * (size_t)s.descs[inst.BlasId].NodeOffset;
    M34 uniInv; uniInv.r0 = uniInv.r1 = uniInv.r2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (UNI) uniInv = load_inv_model_at(s.xforms, pb.uniXformId);
    // the work list in packets of 64 entries, dealt over k_trace2's GRAB_SLICES counters (a slice owns runs of 2^grabUnitLog2 >= 64 consecutive entries: a packet never straddles two runs)
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    uint32_t nPackets = 0, nSteps = 0, nLive = 0, nEnter = 0, nRounds = 0;     // STATS (wave-uniform; a wave's share of a launch fits 32 bits)

    while (slicesDone < GRAB_SLICES) {
        const uint32_t len = ((nBlocks + GRAB_SLICES - 1u - slice) / GRAB_SLICES) << unitLog2;     // entries of this slice (the list's last run may be partial: item < N below)
        const uint32_t q = wave_grab(workCounter + GRAB_STRIDE * slice, WAVE);
        if (q >= len) { slice = (slice + 1u) & (GRAB_SLICES - 1u); slicesDone++; continue; }
        const uint32_t item = ((((q >> unitLog2) * GRAB_SLICES + slice) << unitLog2) | (q & ((1u << unitLog2) - 1u))) + lane;
        const bool valid = item < N;
        // ---- the packet's rays
        uint32_t rayIdx = 0, hitTri = ~0u, hitXform = 0, flags = 0;
        f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(1.0f);
        float hitT = PT_FLOAT_MAX, hbx = 0.0f, hby = 0.0f, second, cullT, leafT1 = 0.0f;
        bool enters = false;
        if (valid) {
            rayIdx = list[item];
            if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                const float4 o = rays.o_ior[rayIdx];
                const f3 wd = DecodeUnitVec(rays.thr_px[rayIdx].w, rays.rad_py[rayIdx].w), wo = mk3(o.x, o.y, o.z);
                for (int i = 0; i < s.lightCount; i++) {
                    const GpuLight& l = s.lights[i];
                    float tMin, tMax;
                    if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                }
            }
            const float4 a = tr.rec[4 * (size_t)rayIdx], b = tr.rec[4 * (size_t)rayIdx + 1], c = tr.rec[4 * (size_t)rayIdx + 2];
            ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z);
            if (UNI) { ro = xform34(uniInv, ro, 1.0f); rd = xform34(uniInv, rd, 0.0f); invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z); }
            const bool enter = UNI || a.w < hitT;               // root test (:32-39): its arithmetic ran in the kernel that produced the ray (record[0].w = tMin, +inf = miss); UNI: no root tests
            const bool finite = gabs(invDir.x) < __builtin_inff() && gabs(invDir.y) < __builtin_inff() && gabs(invDir.z) < __builtin_inff();
            if (enter && !finite) flags = 1u;                   // a slab of this ray can be NaN: the monotonicity argument does not cover it
            enters = enter && finite;
        }
        second = hitT; cullT = hitT * wide::CULL;
        const u64 entered = PK_BALLOT(enters);
        if (STATS) { nPackets++; nEnter += (uint32_t)__builtin_popcountll(entered); }
        // ---- the shared walk
        uint32_t top = 2u, maskLo = (uint32_t)entered, maskHi = (uint32_t)(entered >> 32), ovf = 0u;      // the current node pair, the lanes it is live for; "the stack overflowed"
        int stkNode = 0, stkLo = 0, stkHi = 0, sp = 0;          // lane k of the three registers = stack entry k
        const char* const nodeBytes = (const char*)nodes;
        if (entered != 0ull)
        do {
            const u64 mask = (u64)maskLo | ((u64)maskHi << 32);
            if (STATS) { nSteps++; nLive += (uint32_t)__builtin_popcountll(mask); }
            const pk_u16v P = *(pk_c16*)(uintptr_t)(nodeBytes + ((size_t)top << 5));          // {lmin.xyz, lStart}, {lmax.xyz, lCount}, {rmin.xyz, rStart}, {rmax.xyz, rCount}
            const uint32_t lStart = P[3], lCount = P[7], rStart = P[11], rCount = P[15];
            float t1L, t2L, t1R, t2R;
            pk_box2(ro, invDir, P, &t1L, &t2L, &t1R, &t2R);
            // A child is WANTED while t1 <= min(t2 * NEAR_MISS, T * CULL): one compare per box.  For an inner child that is the box test made lenient by 2^-20 — it enters a superset of
            // what the exact test enters, which only orders the walk (the BLAS boxes nest — packet_possible asks for it — so a leaf whose exact test passes has every ancestor's exact
            // test pass: nothing is met that the reference cannot meet) and spares the near-miss bookkeeping: the narrowly failing boxes are simply entered.  A LEAF child is tested
            // with the exact `t1 <= t2` below, and a lane that wanted it but fails that test by less than 2^-20 is flagged (NEAR_MISS, wide_nodes.hpp).
            u64 hL = mask & PK_BALLOT(t1L <= gmin(t2L * wide::NEAR_MISS, cullT)), hR = mask & PK_BALLOT(t1R <= gmin(t2R * wide::NEAR_MISS, cullT));
            if ((lCount | rCount) != 0u) {                      // (one test for the common case: both children internal)
                // leaf children first, left then right (BVHIntersect.glsl:54-79), by the lanes whose box test passed
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const uint32_t cnt = side ? rCount : lCount, first = (side ? rStart : lStart) + triOffset;
                    if (cnt == 0u || (side ? hR : hL) == 0ull) continue;
                    const u64 in = side ? PK_BALLOT(t1R <= t2R) : PK_BALLOT(t1L <= t2L), h = (side ? hR : hL) & in, nm = (side ? hR : hL) & ~in;
                    if (nm != 0ull) flags = PK_LANES(nm) ? (flags | 8u) : flags;
                    if (h == 0ull) continue;
                    const float t1 = side ? t1R : t1L;
                    for (uint32_t k = 0; k < cnt; k++) {
                        if (STATS) nRounds++;
                        pk_c4* tv = (pk_c4*)(uintptr_t)(s.triVerts + 3 * (size_t)(first + k));
                        const pk_u4v A = tv[0], B = tv[1], C = tv[2];
                        float by, bz, t;
                        (void)RayTriangleIntersect(ro, rd, mk3(__uint_as_float(A[0]), __uint_as_float(A[1]), __uint_as_float(A[2])), mk3(__uint_as_float(B[0]), __uint_as_float(B[1]), __uint_as_float(B[2])),
                                                   mk3(__uint_as_float(C[0]), __uint_as_float(C[1]), __uint_as_float(C[2])), &by, &bz, &t);
                        const float bx = 1.0f - by - bz;
                        const u64 hm = h & PK_BALLOT(bx >= 0.0f) & PK_BALLOT(by >= 0.0f) & PK_BALLOT(bz >= 0.0f) & PK_BALLOT(t >= 0.0f);      // RayTriangleIntersect's result (:22)
                        if (hm == 0ull) continue;
                        const uint32_t id = first + k;
                        const u64 closer = hm & PK_BALLOT(t < hitT), other = PK_BALLOT(id != hitTri);
                        const u64 assume = hm & PK_BALLOT(t1 > t * wide::ASSUME);      // the argument's assumption does not hold for this (ray, triangle) ...
                        if (assume != 0ull && pb.marks[id] == 0) flags = PK_LANES(assume) ? (flags | 4u) : flags;      // ... and the triangle is not a marked one (those are flagged when they win): not vouched for
                        // second-best distance among other triangle ids: the old best when this one takes over, this one's when it does not
                        const float cand = PK_LANES(closer) ? hitT : t;
                        second = PK_LANES(hm & other) ? gmin(second, cand) : second;
                        const bool c = PK_LANES(closer);
                        hitT = c ? t : hitT; hbx = c ? bx : hbx; hby = c ? by : hby; hitTri = c ? id : hitTri; leafT1 = c ? t1 : leafT1;
                        cullT = hitT * wide::CULL;
                    }
                }
                // the descent: internal children only, and only where the lanes' (possibly shrunken) T still wants them
                hL = lCount == 0u ? (hL & PK_BALLOT(t1L <= cullT)) : 0ull; hR = rCount == 0u ? (hR & PK_BALLOT(t1R <= cullT)) : 0ull;
            }
            // ---- where to go next.  Hand-written: left to the compiler this three-way decision over 64-bit masks, the push and the pop cost 30-35 scalar instructions of materialised
            // booleans per step, and the walk is bound by scalar issue (profiles/r06_packet.md: 52 scalar against 44 vector instructions per step, the SIMDs' scalar issue 90 % busy).
            //   both children wanted: the side more lanes find nearer goes first, the other is pushed with the mask of the lanes that hit it (stack entry sp = lane sp of the three
            //   registers: v_writelane_b32, the lane through m0); one child: go there; none: pop (v_readlane_b32); nothing to pop, or the stack full (64 entries; ovf = 1): mask = 0.
            {
                uint32_t pkA, pkB, pkC; u64 pkT;
                asm volatile(
                    "s_cmp_eq_u64 %[hL], 0\n\t"
                    "s_cbranch_scc1 1f\n\t"
                    "s_cmp_eq_u64 %[hR], 0\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "v_cmp_lt_f32 vcc, %[t1L], %[t1R]\n\t"            // nearL
                    "s_orn2_b64 %[tt], vcc, %[hR]\n\t"
                    "s_and_b64 %[tt], %[tt], %[hL]\n\t"
                    "s_bcnt1_i32_b64 %[a], %[tt]\n\t"                  // lanes that want the left child first: hL & (~hR | nearL)
                    "s_and_b64 %[tt], %[hL], vcc\n\t"
                    "s_andn2_b64 %[tt], %[hR], %[tt]\n\t"
                    "s_bcnt1_i32_b64 %[b], %[tt]\n\t"                  // ... the right one first: hR & ~(hL & nearL)
                    "s_cmp_gt_i32 %[sp], 63\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_mov_b32 m0, %[sp]\n\t"
                    "s_cmp_ge_i32 %[a], %[b]\n\t"                      // scc = left first
                    "s_cselect_b32 %[top], %[sL], %[sR]\n\t"
                    "s_cselect_b32 %[mlo], %[hLlo], %[hRlo]\n\t"
                    "s_cselect_b32 %[mhi], %[hLhi], %[hRhi]\n\t"
                    "s_cselect_b32 %[a], %[sR], %[sL]\n\t"             // the far child, the lanes that hit it
                    "s_cselect_b32 %[b], %[hRlo], %[hLlo]\n\t"
                    "s_cselect_b32 %[c], %[hRhi], %[hLhi]\n\t"
                    "s_add_i32 %[sp], %[sp], 1\n\t"
                    "v_writelane_b32 %[kn], %[a], m0\n\t"
                    "v_writelane_b32 %[kl], %[b], m0\n\t"
                    "v_writelane_b32 %[kh], %[c], m0\n\t"
                    "s_branch 9f\n"
                    "2:\n\t"                                           // left only
                    "s_mov_b32 %[top], %[sL]\n\t"
                    "s_mov_b32 %[mlo], %[hLlo]\n\t"
                    "s_mov_b32 %[mhi], %[hLhi]\n\t"
                    "s_branch 9f\n"
                    "1:\n\t"
                    "s_cmp_eq_u64 %[hR], 0\n\t"
                    "s_cbranch_scc1 4f\n\t"
                    "s_mov_b32 %[top], %[sR]\n\t"                     // right only
                    "s_mov_b32 %[mlo], %[hRlo]\n\t"
                    "s_mov_b32 %[mhi], %[hRhi]\n\t"
                    "s_branch 9f\n"
                    "4:\n\t"                                           // none: pop
                    "s_cmp_eq_u32 %[sp], 0\n\t"
                    "s_cbranch_scc1 5f\n\t"
                    "s_add_i32 %[sp], %[sp], -1\n\t"
                    "s_nop 0\n\t"
                    "v_readlane_b32 %[top], %[kn], %[sp]\n\t"
                    "v_readlane_b32 %[mlo], %[kl], %[sp]\n\t"
                    "v_readlane_b32 %[mhi], %[kh], %[sp]\n\t"
                    "s_branch 9f\n"
                    "3:\n\t"
                    "s_mov_b32 %[ovf], 1\n"
                    "5:\n\t"
                    "s_mov_b32 %[mlo], 0\n\t"
                    "s_mov_b32 %[mhi], 0\n"
                    "9:"
                    : [top] "=&s"(top), [mlo] "=&s"(maskLo), [mhi] "=&s"(maskHi), [sp] "+s"(sp), [ovf] "+s"(ovf), [kn] "+v"(stkNode), [kl] "+v"(stkLo), [kh] "+v"(stkHi), [a] "=&s"(pkA), [b] "=&s"(pkB), [c] "=&s"(pkC), [tt] "=&s"(pkT)
                    : [hL] "s"(hL), [hR] "s"(hR), [hLlo] "s"((uint32_t)hL), [hLhi] "s"((uint32_t)(hL >> 32)), [hRlo] "s"((uint32_t)hR), [hRhi] "s"((uint32_t)(hR >> 32)), [t1L] "v"(t1L), [t1R] "v"(t1R), [sL] "s"(lStart), [sR] "s"(rStart)
                    : "vcc", "scc", "m0");
            }
        } while ((maskLo | maskHi) != 0u);
        if (hitTri != ~0u) {
            hitXform = inst.MeshTransformId;
            if (UNI) {                                          // whose triangle it is: the last BLAS whose triangles start at or before it
                int lo = 0, hi = pb.blasCount;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pb.blasTriStart[mid] <= hitTri) lo = mid; else hi = mid; }
                hitXform = pb.blasXform[lo];
            }
        }
        // ---- retire: store the hit, or hand the ray to the exact kernel
        bool flagged = false;
        if (valid) {
            const float win = hitT * wide::WINDOW;
            flagged = flags != 0u || (ovf != 0u && PK_LANES(entered)) || (hitTri != ~0u && (second <= win || leafT1 > win || pb.marks[hitTri] != 0));
            if (!flagged) store_hit(hits, rayIdx, hitT, hbx, hby, hitTri, hitXform);
        }
        const u64 fm = PK_BALLOT(flagged);
        if (fm != 0ull) {
            const uint32_t cntF = (uint32_t)__builtin_popcountll(fm);
            uint32_t base = 0;
            if (lane == (uint32_t)__builtin_ctzll(fm)) { base = atomicAdd(pb.flagCount, cntF); atomicAdd(pb.totals, (unsigned long long)cntF); }
            base = (uint32_t)__shfl((int)base, __builtin_ctzll(fm));
            if (flagged) pb.flagA[base + (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1ull))] = rayIdx;
        }
    }
    if (STATS && lane == 0 && nPackets) { atomicAdd(pb.totals + 1, (unsigned long long)nPackets); atomicAdd(pb.totals + 2, (unsigned long long)nSteps); atomicAdd(pb.totals + 3, (unsigned long long)nLive); atomicAdd(pb.totals + 4, (unsigned long long)nEnter); atomicAdd(pb.totals + 5, (unsigned long long)nRounds); }
}
#undef PK_BALLOT
#undef PK_LANES

// the packet launches' counters, copied into host-mapped memory right behind every packet launch (stream order): what packet_decide reads, without a synchronisation
__global__ void k_packet_mirror(const unsigned long long* totals /* PacketBufs::totals */, unsigned long long* host)
{
    if (threadIdx.x >= 1u && threadIdx.x < 6u) host[threadIdx.x] = totals[threadIdx.x];
}
