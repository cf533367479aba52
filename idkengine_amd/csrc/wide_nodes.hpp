// wide_nodes.hpp — the derived 4-wide traversal structure ("wide nodes") of the closest-hit fast path: layout, the build rules and the per-node
// arithmetic, shared by the device kernels (kernels_wide.hpp), the host-side builder used for host-built scenes' validation and by tests/tools (plain C++: no HIP needed).
//
// WHAT IT IS.  The reference walks a BVH2 whose nodes come in 64-byte sibling pairs (Bvh/BLAS.cs:12-22; one pair per dependent memory round trip,
// BVHIntersect.glsl:43-101).  A wide node is ONE 64-byte block that stands for up to four BVH2 nodes found by expanding a pair's internal children (largest
// half-area first): a common grid origin, a power-of-two cell size per axis, 4 x 6 bytes of child boxes rounded OUTWARD to that grid, 4 child words.
// A ray takes about half as many dependent round trips at the same bytes per trip; the node array is a third of the BVH2's.
//
// WHY THE RESULTS STAY THE REFERENCE'S.  The structure only decides WHICH leaves are looked at; everything that produces a number is the reference's own
// arithmetic on the reference's own data:
//   * a wide child's box is dequantised to fp32 bounds lo' <= lo, hi' >= hi (checked at build time with the very expression the kernel evaluates) and then tested
//     with RayBoxIntersect's expression (IntersectionRoutines.glsl:25-40).  IEEE subtraction and multiplication are monotone, so t1' <= t1 and t2' >= t2 for every
//     BVH2 node inside that child: a box test the reference passes, the wide test passes too (rays with an infinite 1/dir component — NaN slabs — never use this path);
//   * a leaf is entered through its LEAF RECORD: the BVH2 leaf node itself (exact fp32 box, TriStartOrChild, TriCount) followed by its triangles' positions.
//     The triangles are tested only if the reference's box test on that exact box passes, with RayTriangleIntersect's expression (:6-23) in the reference's order;
//     by the same monotonicity that box test passing implies every ancestor's box test passes (ancestor boxes contain the leaf box: t1_anc <= t1_leaf, t2_anc >= t2_leaf);
//   * what can still differ is ORDER: the reference keeps the first of two hits with equal t, and culls a box whose t1 exceeds the T it has found so far.  The wide walk
//     therefore (a) culls with a slack, t1' <= T * CULL (1 + 2^-14), (b) remembers the second-best hit distance among different triangle ids (starting from the ray's
//     initial T), and (c) remembers t1 of the leaf box the best hit was found in.  A ray is FLAGGED — and traced again by the exact BVH2 kernel — if its second-best
//     distance or its winning leaf's t1 lies inside the window T * WINDOW (1 + 2^-16) of its final T, if a component of 1/dir is not finite, or if its stack overflowed.
//     For an unflagged ray: every triangle the reference can hit with t <= T * WINDOW sits in a leaf the wide walk tested (its exact box test passes and its t1 is at most
//     t * (1 + a) <= T * CULL — `a` <= 3 * 2^-16 is the one ASSUMPTION: the reference's triangle test never reports a hit that far in front of its own leaf box's entry; every
//     tested triangle is checked against it and the CPU model counts violations: none in 10^8 rays), so there is none but the winner; hence the reference's T stays above
//     T * WINDOW >= t1(winning leaf) >= t1(every ancestor) until it tests the winning leaf, it does reach that leaf, and it accepts the same triangle: same TriangleId, same
//     T, same barycentrics, bit for bit.  A ray the wide walk misses, the reference misses for the same reason.
//   * two more cases are flagged rather than argued: a ray whose best hit is a MARKED triangle (a PreSplit fragment: the leaf box holds only part of the triangle, the
//     assumption above does not apply to it and which of its copies the reference reports depends on the order of the reference's walk), and a ray for which any box test
//     of the wide walk failed by less than 2^-20 relative (NEAR_MISS: such a test can pass in exact arithmetic, and then the reference may still reach that triangle
//     through the box of another fragment).
//     (DESIGN.md §4 "Wide nodes" has the argument in full; tools/wide_sim.cpp and tests/test_wide_nodes.py check it against the BVH2 walk ray by ray on the CPU.)
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define WIDE_HD __host__ __device__ __forceinline__
#else
#define WIDE_HD inline
#endif

namespace wide {

// 64 bytes = 4 x uint4.  Children are packed from slot 0; an unused slot has child word 0 (wide node 0 is the root and nobody's child; leaf words have bit 31 set).
struct Node {
    float ox, oy, oz;          // grid origin: the componentwise minimum of the children's boxes (exact)
    uint32_t exps;             // byte a (a = 0, 1, 2): biased exponent of the cell size of axis a (cell = 2^(e - 127)); byte 3: number of children
    uint32_t qlo[3];           // per axis: byte k = lower bound of child k in cells from the origin, rounded down
    uint32_t qhi[3];           // per axis: byte k = upper bound of child k, rounded up
    uint32_t child[4];         // internal child: index of its wide node (BLAS-local); leaf child: LEAF_BIT | offset of its leaf record in 16-byte units (BLAS-local)
    uint32_t pad[2];           // zero
};
static_assert(sizeof(Node) == 64, "wide node is one 64-byte block");

constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr float CULL = 1.0f + 6.103515625e-05f;        // 1 + 2^-14: a child / leaf is kept while t1 <= T * CULL
constexpr float WINDOW = 1.0f + 1.52587890625e-05f;    // 1 + 2^-16: a second candidate, the initial T or the winning leaf's t1 inside T * WINDOW flags the ray
constexpr float NEAR_MISS = 1.0f + 9.5367431640625e-07f; // 1 + 2^-20 (16 roundings): a box test that fails with t2 < t1 <= t2 * NEAR_MISS may pass in exact arithmetic: it flags the ray
constexpr float ASSUME = 1.0f + 4.57763671875e-05f;      // 1 + 3 * 2^-16: the assumption's bound (see above); a tested triangle that violates it flags the ray

WIDE_HD float cell_of(uint32_t exps, int axis) { uint32_t b = ((exps >> (8 * axis)) & 255u) << 23; float f; memcpy(&f, &b, 4); return f; }
// THE dequantisation: one fused multiply-add, single rounding (q * cell is exact; the sum rounds once).  Build and traversal both call this.
WIDE_HD float dequant(float origin, float cell, uint32_t q) { return fmaf((float)q, cell, origin); }

}  // namespace wide

namespace wide {

// A BVH2 node as the reference stores it (GpuBlasNode, Source/GpuTypes/GpuBlasNode.cs): 32 bytes, siblings adjacent, TriCount > 0 = leaf.
struct Bvh2Node { float mn[3]; uint32_t startOrChild; float mx[3]; uint32_t triCount; };
static_assert(sizeof(Bvh2Node) == 32, "GpuBlasNode");

// A leaf record's triangle is MARKED (word 3 of its first vertex = 1) when its leaf box does not contain all of it: a PreSplit fragment (Bvh/PreSplitting.cs) — the same
// triangle sits, under other TriangleIds, in other leaves, a ray may meet it through a leaf whose box the hit point is not in, and which copy the reference reports
// depends on the order of its walk.  A ray whose best hit is a marked triangle is flagged.  tv: 3 x (x, y, z, w).
WIDE_HD bool outside_leaf_box(const Bvh2Node& leaf, const float* tv)
{
    bool out = false;
    for (int v = 0; v < 3; v++) for (int a = 0; a < 3; a++) { const float x = tv[4 * v + a]; out = out || !(x >= leaf.mn[a] && x <= leaf.mx[a]); }
    return out;
}

// half area of a box, one fixed expression (builder and device kernel agree on the expansion order through it)
WIDE_HD float half_area(const Bvh2Node& n) { const float x = n.mx[0] - n.mn[0], y = n.mx[1] - n.mn[1], z = n.mx[2] - n.mn[2]; return fmaf(x + y, z, x * y); }

// Which BVH2 nodes (BLAS-local ids) become the children of the wide node that stands for the sibling pair at `pair`: the pair itself, then — while fewer than four —
// the internal member with the largest half area (first slot on ties) is replaced by its two children (kept in its slot and appended).  Returns the count (2..4).
WIDE_HD int expand_pair(const Bvh2Node* nodes, uint32_t pair, uint32_t out[4])
{
    int n = 2; out[0] = pair; out[1] = pair + 1u;
    while (n < 4) {
        int best = -1; float bestA = -1.0f;
        for (int k = 0; k < n; k++) { const Bvh2Node& c = nodes[out[k]]; if (c.triCount == 0u) { const float a = half_area(c); if (best < 0 || a > bestA) { best = k; bestA = a; } } }
        if (best < 0) break;
        const uint32_t ch = nodes[out[best]].startOrChild;
        out[best] = ch; out[n++] = ch + 1u;
    }
    return n;
}

// The grid of one axis and the children's bytes on it.  lo[k] / hi[k]: the children's bounds on this axis.  Returns the exponent byte; qlo / qhi get byte k = child k.
WIDE_HD uint32_t quantize_axis(const float* lo, const float* hi, int n, float* originOut, uint32_t* qloOut, uint32_t* qhiOut)
{
    float origin = lo[0], top = hi[0];
    for (int k = 1; k < n; k++) { origin = fminf(origin, lo[k]); top = fmaxf(top, hi[k]); }
    const float extent = top - origin;
    int e = 1;                                                   // biased exponent of the cell (1..254)
    if (extent > 0.0f) { int ex; (void)frexpf(extent * (1.0f / 255.0f), &ex); e = ex + 126; if (e < 1) e = 1; if (e > 254) e = 254; }   // 2^(ex-1) <= extent/255 < 2^ex
    for (;; e++) {
        uint32_t eb = (uint32_t)e << 23; float cell; memcpy(&cell, &eb, 4);
        uint32_t ql = 0, qh = 0; bool fits = true;
        for (int k = 0; k < n && fits; k++) {
            float fl = floorf((lo[k] - origin) / cell); if (!(fl >= 0.0f)) fl = 0.0f; if (fl > 255.0f) fl = 255.0f;
            uint32_t a = (uint32_t)fl;
            while (a > 0u && dequant(origin, cell, a) > lo[k]) a--;               // (rounding of lo - origin may have pushed it one cell up)
            float fh = ceilf((hi[k] - origin) / cell); if (!(fh >= 0.0f)) fh = 0.0f;
            uint32_t b = fh > 255.0f ? 256u : (uint32_t)fh;
            while (b <= 255u && dequant(origin, cell, b) < hi[k]) b++;
            if (b > 255u) { fits = false; break; }
            ql |= a << (8 * k); qh |= b << (8 * k);
        }
        if (fits || e >= 254) { *originOut = origin; *qloOut = ql; *qhiOut = fits ? qh : 0xffffffffu; return (uint32_t)e; }
    }
}

// the box part of a wide node from its children's BVH2 boxes (topology words are the caller's)
WIDE_HD void quantize_node(const Bvh2Node* nodes, const uint32_t ids[4], int n, Node* out)
{
    float lo[4], hi[4], org[3]; uint32_t exps = (uint32_t)n << 24;
    for (int a = 0; a < 3; a++) {
        for (int k = 0; k < n; k++) { lo[k] = nodes[ids[k]].mn[a]; hi[k] = nodes[ids[k]].mx[a]; }
        const uint32_t e = quantize_axis(lo, hi, n, &org[a], &out->qlo[a], &out->qhi[a]);
        exps |= e << (8 * a);
    }
    out->ox = org[0]; out->oy = org[1]; out->oz = org[2]; out->exps = exps;
}

// The box tests of one wide node: RayBoxIntersect's expression (IntersectionRoutines.glsl:25-40) on the dequantised bounds.  inv must be finite and non-zero in every
// component (the caller flags other rays), so min(t0, t1) is the slab of the bound the ray meets first: the lower one where inv > 0, the upper one where inv < 0.
// t1[k] = entry distance of child k (max with 0 as in the reference); bit k of the result = child k exists, t1 <= t2 and t1 <= cullT.
WIDE_HD uint32_t test_node(const Node& w, const float ro[3], const float inv[3], float cullT, float t1[4], bool* nearMiss = nullptr)
{
    float nearT[4] = {0.0f, 0.0f, 0.0f, 0.0f}, farT[4];
    for (int k = 0; k < 4; k++) farT[k] = INFINITY;
    const float org[3] = {w.ox, w.oy, w.oz};
    for (int a = 0; a < 3; a++) {
        const float cell = cell_of(w.exps, a);
        const bool neg = inv[a] < 0.0f;
        const uint32_t qn = neg ? w.qhi[a] : w.qlo[a], qf = neg ? w.qlo[a] : w.qhi[a];
        for (int k = 0; k < 4; k++) {
            const float tn = (dequant(org[a], cell, (qn >> (8 * k)) & 255u) - ro[a]) * inv[a];
            const float tf = (dequant(org[a], cell, (qf >> (8 * k)) & 255u) - ro[a]) * inv[a];
            nearT[k] = fmaxf(nearT[k], tn); farT[k] = fminf(farT[k], tf);
        }
    }
    uint32_t mask = 0;
    for (int k = 0; k < 4; k++) {
        t1[k] = nearT[k];
        if (w.child[k] != 0u && nearT[k] <= farT[k] && nearT[k] <= cullT) mask |= 1u << k;
        if (nearMiss && w.child[k] != 0u && !(nearT[k] <= farT[k]) && nearT[k] <= farT[k] * NEAR_MISS) *nearMiss = true;
    }
    return mask;
}

}  // namespace wide

#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
namespace wide {

// Host-side build of one BLAS (the order the device build reproduces: breadth first, a node's internal children numbered in slot order, leaf records in the
// order their wide nodes are numbered and, inside a node, in slot order).  nodes: the BLAS's BVH2 nodes (index 0 = padding, 1 = root, 2 = the root's left child);
// triVerts: 3 x (x, y, z, w) floats per BLAS triangle in leaf order.  leafRecs: 16-byte units (4 floats each).
struct HostBuild { std::vector<Node> nodes; std::vector<float> leafRecs; std::vector<uint32_t> pairOf; /* BVH2 pair id of every wide node */ };
inline HostBuild build_host(const Bvh2Node* nodes, uint32_t nodeCount, const float* triVerts)
{
    HostBuild B;
    if (nodeCount < 4) return B;
    B.pairOf.push_back(2u);
    for (size_t i = 0; i < B.pairOf.size(); i++) {
        uint32_t ids[4]; const int n = expand_pair(nodes, B.pairOf[i], ids);
        Node w; memset(&w, 0, sizeof w);
        quantize_node(nodes, ids, n, &w);
        for (int k = 0; k < n; k++) {
            const Bvh2Node& c = nodes[ids[k]];
            if (c.triCount == 0u) { w.child[k] = (uint32_t)B.pairOf.size(); B.pairOf.push_back(c.startOrChild); }
            else {
                w.child[k] = LEAF_BIT | (uint32_t)(B.leafRecs.size() / 4);
                float h[8]; memcpy(h, &c, 32);
                B.leafRecs.insert(B.leafRecs.end(), h, h + 8);
                const float* tv = triVerts + 12 * (size_t)c.startOrChild;
                const size_t at = B.leafRecs.size();
                B.leafRecs.insert(B.leafRecs.end(), tv, tv + 12 * (size_t)c.triCount);
                for (uint32_t t = 0; t < c.triCount; t++) { const uint32_t m = outside_leaf_box(c, tv + 12 * (size_t)t) ? 1u : 0u; memcpy(&B.leafRecs[at + 12 * (size_t)t + 3], &m, 4); }
            }
        }
        B.nodes.push_back(w);
    }
    return B;
}

}  // namespace wide
#endif
