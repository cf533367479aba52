// bvh_gpu_full.hpp — idkptBuildBlas: the rest of the reference's BLAS build on the GPU, around the SweepSAH core of bvh_gpu.hpp.
// Part of the single translation unit idkpt.hip.  What runs here, and the reference code it must reproduce byte for byte (tests/test_gpu_builder.py
// compares with libidkbvh's CPU build, which tests/test_builder.py holds to the oracle's independent restatement and to tests/golden/bvh.json):
//
//   fragments   PreSplitting.PreSplit (Bvh/PreSplitting.cs:26-160): priority per triangle (cbrt of extent^2 x empty area), the binary32 running
//               sum of the priorities IN INDEX ORDER (one dependent chain: summed on the host from a 4-B-per-triangle download — a GPU has
//               nothing to offer a serial float sum), split counts, an integer scan for every triangle's first fragment, the scene box with
//               the minps/maxps tie rule in triangle order, and the recursive grid split of every triangle by one thread into its own range.
//               Refittable BLASes take one box per triangle (BLAS.GetTriangleBounds).
//   core        bvh_gpu.hpp, on the fragments where they lie.
//   tail        single-leaf root (BLAS.cs:173-183); ComputeRequiredStackSize (:672-702) as a bottom-up climb; OptimizeStackSize (:875-937);
//               RemoveEmptySubtrees (:245-273); GetUnindexedTriangles (BLAS.cs:441-466 / PreSplitting.cs:169-273); parent and leaf indices
//               (:481-514); ComputeGlobalSAH (:629-657).
//
// Two observations make the tail parallel without changing a bit:
//   * ids are reserved per subtree (BLAS.cs:221-241: the left subtree of a node owns the ids right behind the node's child pair, the right
//     subtree the ids behind those), so increasing child-pair id IS the pre-order in which RemoveEmptySubtrees renumbers the pairs: the
//     compaction is a stream compaction of the used pairs, a flag per pair and one integer scan.
//   * OptimizeStackSize collapses, pass after pass, every subtree below a depth threshold that moves up by one per pass; after the pass with
//     threshold S every node deeper than S is a leaf whose range is the union of its subtree's leaves.  So the whole loop is decided by sums of
//     per-node terms grouped by depth (terms of the nodes AT depth S, computed with the aggregated triangle counts of their children), and its
//     effect is one collapse at the final threshold.  The reference accumulates those terms in binary64 in tree order; here they are summed in
//     parallel, and the host replays the loop with a rigorous bound on the difference between the two summation orders: a decision is taken
//     only when `increase <= acceptance` holds or fails by more than the bound; otherwise (never observed) the host walks the downloaded tree
//     in the reference's order.  Decisions, hence bytes, are exactly the reference's either way.
#pragma once

namespace bvhgpu {

// cbrtf of glibc 2.35 (sysdeps/ieee754/flt-32/s_cbrtf.c: frexp, a quadratic seed, one Halley step in binary64, a factor table, ldexp) — what
// MathF.Cbrt / cbrtf return on this image's hosts, where libidkbvh calls it.  Compared with the host's cbrtf on all 2^32 inputs (CPU twin in
// tests/c_driver/cbrt_check.c: 0 mismatches) and on the device (tests/test_gpu_builder.py).
DEV float dev_cbrtf(float x)
{
    const uint32_t ax = __float_as_uint(x) & 0x7fffffffu;
    if (ax == 0u || ax >= 0x7f800000u) return x + x;
    int xe; uint32_t m = ax;
    if (m < 0x00800000u) { const int sh = __builtin_clz(m) - 8; m <<= sh; xe = (1 - sh) - 126; } else xe = (int)(m >> 23) - 126;   // frexpf: |x| = xm * 2^xe, xm in [0.5, 1)
    const float xm = __uint_as_float((m & 0x007fffffu) | 0x3f000000u);
    const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    const float t2 = u * u * u;
    const int r = xe % 3;
    const double f = r == -2 ? 1.0 / 1.5874010519681994748 : r == -1 ? 1.0 / 1.2599210498948731648 : r == 0 ? 1.0 : r == 1 ? 1.2599210498948731648 : 1.5874010519681994748;
    const float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * f);
    const float s = __uint_as_float((uint32_t)(127 + xe / 3) << 23);        // ldexpf(., xe / 3): exact, the result is a normal number
    return (x > 0.0f ? ym : -ym) * s;
}
__global__ void k_cbrt_probe(const float* in, float* out, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = dev_cbrtf(in[i]); }

DEV int sat_int(float f) { if (f != f) return 0; if (f >= 2147483648.0f) return 0x7fffffff; if (f <= -2147483648.0f) return (int)0x80000000; return (int)f; }   // C# (int)float saturates on x64 RyuJIT the way libidkbvh's satInt does
DEV void load_tri(const float* pos, const uint4 t, float a[3], float b[3], float c[3])
{
    for (int k = 0; k < 3; k++) { a[k] = pos[3 * (size_t)t.x + k]; b[k] = pos[3 * (size_t)t.y + k]; c[k] = pos[3 * (size_t)t.z + k]; }
}
DEV void grow_pt(BBox& bx, const float p[3]) { for (int k = 0; k < 3; k++) { bx.mn[k] = sse_min(bx.mn[k], p[k]); bx.mx[k] = sse_max(bx.mx[k], p[k]); } }   // Box.GrowToFit(point): minps / maxps (acc, p)
DEV BBox tri_box(const float a[3], const float b[3], const float c[3]) { BBox bx; for (int k = 0; k < 3; k++) { bx.mn[k] = a[k]; bx.mx[k] = a[k]; } grow_pt(bx, b); grow_pt(bx, c); return bx; }   // Box.From(triangle)
DEV float largest_extent(const BBox& b) { const float s0 = b.mx[0] - b.mn[0], s1 = b.mx[1] - b.mn[1], s2 = b.mx[2] - b.mn[2]; const float a = s1 > s2 ? s1 : s2; return s0 > a ? s0 : a; }
DEV int largest_axis(const BBox& b) { const float s[3] = {b.mx[0] - b.mn[0], b.mx[1] - b.mn[1], b.mx[2] - b.mn[2]}; int a = 0; if (s[0] < s[1]) a = 1; if (s[a] < s[2]) a = 2; return a; }
DEV void store_frag(float4* fb, size_t i, const BBox& b) { fb[2 * i] = make_float4(b.mn[0], b.mn[1], b.mn[2], 0.0f); fb[2 * i + 1] = make_float4(b.mx[0], b.mx[1], b.mx[2], 0.0f); }

// ---- fragments
__global__ void k_tri_boxes(const float* pos, const uint4* tris, int n, float4* fb)          // BLAS.GetTriangleBounds
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[3], b[3], c[3]; load_tri(pos, tris[i], a, b, c);
    store_frag(fb, (size_t)i, tri_box(a, b, c));
}
__global__ void k_tri_prio(const float* pos, const uint4* tris, int n, float* prio)          // Priority(), PreSplitting.cs:124-135
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[3], b[3], c[3]; load_tri(pos, tris[i], a, b, c);
    const BBox bx = tri_box(a, b, c);
    const float e = largest_extent(bx), area = half_area(bx) * 2.0f;
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
    const float triArea = __builtin_sqrtf((cx * cx) + (cy * cy) + (cz * cz)) * 0.5f;
    prio[i] = dev_cbrtf((e * e) * (area - triArea));
}
__global__ void k_split_count(const float* prio, float total, int n, float factor, uint32_t* cnt, unsigned long long* sum64)   // GetSplitCount, PreSplitting.cs:116-122
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0;
    if (i < n) {
        const float share = prio[i] / total * (float)n;
        c = 1u + (uint32_t)sat_int(share * factor);
        cnt[i] = c;
    }
    // the exact total (the 32-bit scan of the counts wraps for absurd split factors: the host refuses the build then)
    unsigned long long w = c;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(sum64, w);
}
// scene box (BLAS.ComputeBoundingBox over the geometry): grown point by point in triangle order with minps/maxps -> an in-order reduction
__global__ __launch_bounds__(CH) void k_global_box_partial(const float* pos, const uint4* tris, int n, BBox* part)
{
    __shared__ BBox sh[CH];
    const int i = blockIdx.x * CH + threadIdx.x;
    BBox v = box_empty();
    if (i < n) { float a[3], b[3], c[3]; load_tri(pos, tris[i], a, b, c); grow_pt(v, a); grow_pt(v, b); grow_pt(v, c); }
    const BBox r = block_reduce<false>(v, i < n, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ __launch_bounds__(CH) void k_global_box_final(const BBox* part, int m, BBox* out)
{
    __shared__ BBox sh[CH];
    const int t = threadIdx.x, per = (m + CH - 1) / CH, b0 = min(t * per, m), b1 = min(b0 + per, m);
    BBox v = box_empty();
    for (int k = b0; k < b1; k++) v = join_lr(v, part[k]);
    const BBox r = block_reduce<false>(v, true, sh);
    if (t == 0) *out = r;
}
// the recursive grid split of one triangle (PreSplitting.cs:57-112, Triangle.Split Shapes/Triangle.cs:48-97), fragments in the order the
// reference's stack emits them, into the triangle's own range
__global__ __launch_bounds__(64) void k_presplit(const float* pos, const uint4* tris, int n, const uint32_t* cnt, const uint32_t* first, const BBox* globalBox, float4* fb, int* origTri, uint32_t* stackOverflow)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float P[3][3]; load_tri(pos, tris[i], P[0], P[1], P[2]);
    const BBox global = *globalBox;
    const float gsz[3] = {global.mx[0] - global.mn[0], global.mx[1] - global.mn[1], global.mx[2] - global.mn[2]};
    size_t w = first[i];
    BBox stackBox[64]; int stackSplits[64]; int sp = 0;
    stackBox[0] = tri_box(P[0], P[1], P[2]); stackSplits[0] = (int)cnt[i]; sp = 1;
    while (sp > 0) {
        sp--;
        const BBox box = stackBox[sp]; const int splits = stackSplits[sp];
        if (splits == 1) { store_frag(fb, w, box); origTri[w] = i; w++; continue; }
        const int axis = largest_axis(box);
        const float ext = largest_extent(box);
        const float alpha = ext / gsz[axis];
        float nodeSize = __uint_as_float(__float_as_uint(alpha) & (255u << 23)) * gsz[axis];      // GetNodeSize: 2^floor(log2(alpha)) * globalSize
        if (nodeSize >= ext - 0.0001f) nodeSize *= 0.5f;
        const float mid = (box.mn[axis] + box.mx[axis]) * 0.5f;
        const float idx = __builtin_rintf((mid - global.mn[axis]) / nodeSize);                      // MathF.Round: to nearest, ties to even
        const float splitPos = global.mn[axis] + idx * nodeSize;
        BBox lb = box_empty(), rb = box_empty();
        bool q[3];
        for (int v = 0; v < 3; v++) { q[v] = P[v][axis] <= splitPos; if (q[v]) grow_pt(lb, P[v]); else grow_pt(rb, P[v]); }
        for (int e = 0; e < 3; e++) {
            const int a = e, b = (e + 1) % 3;
            if (q[a] != q[b]) {
                const float t = (splitPos - P[a][axis]) / (P[b][axis] - P[a][axis]);
                float m[3]; for (int k = 0; k < 3; k++) m[k] = P[a][k] + t * (P[b][k] - P[a][k]);
                grow_pt(lb, m); grow_pt(rb, m);
            }
        }
        for (int k = 0; k < 3; k++) {                                                            // Box.ClipAgainst: maxps(mn, parent.mn), minps(mx, parent.mx)
            lb.mn[k] = sse_max(lb.mn[k], box.mn[k]); lb.mx[k] = sse_min(lb.mx[k], box.mx[k]);
            rb.mn[k] = sse_max(rb.mn[k], box.mn[k]); rb.mx[k] = sse_min(rb.mx[k], box.mx[k]);
        }
        const float le = largest_extent(lb), re = largest_extent(rb);
        int lc = sat_int((float)splits * (le / (le + re)));
        lc = min(max(lc, 1), splits - 1);
        if (sp + 2 > 64) { *stackOverflow = 1u; sp = 0; break; }                                  // (the reference's stackalloc of 64 would have thrown: the host reports it)
        stackBox[sp] = rb; stackSplits[sp] = splits - lc; sp++;
        stackBox[sp] = lb; stackSplits[sp] = lc; sp++;
    }
}

// ---- generic exclusive scan of uint32 (integers: order-free).  One level: 1024 threads x 8 items per workgroup; block totals are scanned recursively.
#define SCAN_BLOCK 1024
#define SCAN_ITEMS 8
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_block(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* blockTotals)
{
    __shared__ uint32_t sh[SCAN_BLOCK];
    const uint32_t t = threadIdx.x, base = (blockIdx.x * SCAN_BLOCK + t) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    sh[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < SCAN_BLOCK; off <<= 1) { const uint32_t x = t >= off ? sh[t - off] : 0u; __syncthreads(); sh[t] += x; __syncthreads(); }
    uint32_t run = sh[t] - sum;
    for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (t == SCAN_BLOCK - 1) blockTotals[blockIdx.x] = sh[t];
}
__global__ void k_scan_add(uint32_t* out, uint32_t n, const uint32_t* blockBase)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += blockBase[i / (SCAN_BLOCK * SCAN_ITEMS)];
}

// ---- tail
// The node array after the core: zero-filled, then written by the recursion.  A slot holds a leaf (count > 0), an internal node (count == 0,
// child id >= 2) or nothing (all zero: reserved and not used).
DEV bool node_leaf(const HNodeG& n) { return n.count > 0; }
DEV bool node_internal(const HNodeG& n) { return n.count == 0 && n.startOrChild >= 2; }
DEV float node_half_area(const HNodeG& n) { const float x = n.mx[0] - n.mn[0], y = n.mx[1] - n.mn[1], z = n.mx[2] - n.mn[2]; return __builtin_fmaf(x + y, z, x * y); }

__global__ void k_fix_root(HNodeG* nodes)                       // BLAS.cs:173-183: a root that is a leaf gets two copies of itself as children
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && node_leaf(nodes[1])) { nodes[2] = nodes[1]; nodes[3] = nodes[1]; nodes[1].startOrChild = 2; nodes[1].count = 0; }
}
// parent of every node, pointer-jumping state for the depths, and the climb's arrival counters
__global__ void k_tree_init(const HNodeG* nodes, int nodeCount, int* parent, int* jump, int* dist, int* arrived)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount) return;
    if (i == 1) { parent[1] = 1; jump[1] = 1; dist[1] = 0; }
    arrived[i] = 0;
    const HNodeG nd = i >= 1 ? nodes[i] : HNodeG{};
    if (i >= 1 && node_internal(nd)) { const int c = nd.startOrChild; parent[c] = i; parent[c + 1] = i; jump[c] = i; jump[c + 1] = i; dist[c] = 1; dist[c + 1] = 1; }
}
__global__ void k_depth_jump(const HNodeG* nodes, int nodeCount, const int* jumpIn, const int* distIn, int* jumpOut, int* distOut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount || i < 1) return;
    const HNodeG nd = nodes[i];
    if (!(i == 1 || node_leaf(nd) || node_internal(nd))) return;
    const int j = jumpIn[i];
    distOut[i] = distIn[i] + distIn[j]; jumpOut[i] = jumpIn[j];
}
// Bottom-up climb: a leaf reports to its parent; the second child to arrive (its sibling's values are then published) computes the parent:
//   need[p]  = traversal-stack rows of the pair below p (ComputeRequiredStackSize, BLAS.cs:672-702)
//   agg*[p]  = fragment range of p's whole subtree (what p becomes when OptimizeStackSize collapses it)
// Cross-workgroup hand-off: values are published with agent-scope release (threadfence) before the arrival counter's atomic, read after it.
__global__ void k_climb(const HNodeG* nodes, int nodeCount, const int* parent, int* arrived, int* need, int* aggStart, int* aggCount)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount || i < 2) return;
    const HNodeG nd = nodes[i];
    if (!node_leaf(nd)) return;
    need[i] = 0; aggStart[i] = nd.startOrChild; aggCount[i] = nd.count;
    int cur = i;
    while (cur != 1) {
        const int p = parent[cur];
        __threadfence();
        if (atomicAdd(&arrived[p], 1) == 0) return;                  // first child: the sibling will continue
        __threadfence();
        const int c = nodes[p].startOrChild;
        const HNodeG l = nodes[c], r = nodes[c + 1];
        const bool tl = !node_leaf(l), tr = !node_leaf(r);
        const int nl = __atomic_load_n(&need[c], __ATOMIC_RELAXED), nr = __atomic_load_n(&need[c + 1], __ATOMIC_RELAXED);
        need[p] = (tl && tr) ? max(nl, nr) + 1 : (tl ? nl : (tr ? nr : 0));
        aggStart[p] = __atomic_load_n(&aggStart[c], __ATOMIC_RELAXED);
        aggCount[p] = __atomic_load_n(&aggCount[c], __ATOMIC_RELAXED) + __atomic_load_n(&aggCount[c + 1], __ATOMIC_RELAXED);
        cur = p;
    }
}
// The same bottom-up pass level by level (what the host uses unless the tree is a degenerate chain): leaves first, then one launch per depth from the
// deepest internal level up to the root — 40-60 small launches instead of a million fenced atomic hand-offs (8.6 -> 1 ms on soup-1M).
__global__ __launch_bounds__(256) void k_leaf_init(const HNodeG* nodes, int nodeCount, const int* depth, int* need, int* aggStart, int* aggCount, int* maxDepth)
{
    __shared__ int shMax;
    if (threadIdx.x == 0) shMax = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 1 && i < nodeCount) {
        const HNodeG nd = nodes[i];
        if (node_leaf(nd)) { need[i] = 0; aggStart[i] = nd.startOrChild; aggCount[i] = nd.count; }
        if (node_leaf(nd) || node_internal(nd)) atomicMax(&shMax, depth[i]);
    }
    __syncthreads();
    if (threadIdx.x == 0 && shMax > 0) atomicMax(maxDepth, shMax);
}
__global__ void k_level_up(const HNodeG* nodes, int nodeCount, const int* depth, int d, int* need, int* aggStart, int* aggCount)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount || i < 1) return;
    const HNodeG nd = nodes[i];
    if (!node_internal(nd) || depth[i] != d) return;
    const int c = nd.startOrChild;
    const bool tl = !node_leaf(nodes[c]), tr = !node_leaf(nodes[c + 1]);
    const int nl = need[c], nr = need[c + 1];
    need[i] = (tl && tr) ? max(nl, nr) + 1 : (tl ? nl : (tr ? nr : 0));
    aggStart[i] = aggStart[c]; aggCount[i] = aggCount[c] + aggCount[c + 1];
}
// Sums that decide OptimizeStackSize (BLAS.cs:875-936), binary64, grouped the way the host's replay of the loop needs them:
//   bins[0]            ComputeGlobalSAH of the tree                                       (all terms >= 0)
//   bins[1], bins[2]   first pass: sum / sum of |.| of the collapse terms of the nodes deeper than R - 1 whose children are both leaves
//   bins[4 + 3d ...]   per depth d: sum, sum of |.|, count of the collapse terms of ALL internal nodes at depth d, with the children's
//                      aggregated counts (what they hold once everything below d has been collapsed)
#define OPT_MAX_DEPTH 1024
DEV double collapse_term(const HNodeG& p, const HNodeG& l, const HNodeG& r, int lc, int rc, double rootHalfArea)
{
    const double leavesCost = (double)1.1f * ((double)lc * (double)node_half_area(l) + (double)rc * (double)node_half_area(r));
    const double newParentLeafCost = (double)1.1f * (double)(lc + rc);
    return ((double)node_half_area(p) * (newParentLeafCost - (double)1.0f) - leavesCost) / rootHalfArea;
}
#define OPT_LDS_DEPTHS 128
__global__ __launch_bounds__(256) void k_opt_sums(const HNodeG* nodes, int nodeCount, const int* depth, const int* aggCount, int R, double* bins, int* maxDepth)
{
    // per-depth sums are gathered in LDS first (one address per depth takes every internal node of the tree otherwise), then flushed once per workgroup
    __shared__ double shBin[3 * OPT_LDS_DEPTHS];
    __shared__ double sh[3][256];
    __shared__ int shMax;
    const int t = threadIdx.x;
    for (int k = t; k < 3 * OPT_LDS_DEPTHS; k += 256) shBin[k] = 0.0;
    if (t == 0) shMax = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double sah = 0.0, first = 0.0, firstAbs = 0.0;
    if (i >= 1 && i < nodeCount) {
        const HNodeG nd = nodes[i];
        const bool leaf = node_leaf(nd), inner = node_internal(nd);
        if (leaf || inner) {
            const double rootHalfArea = (double)node_half_area(nodes[1]);
            const double prob = (double)node_half_area(nd) * (1.0 / rootHalfArea);
            sah = leaf ? (double)(1.1f * (float)nd.count) * prob : (double)1.0f * prob;
            if (inner) {
                const int d = depth[i];
                atomicMax(&shMax, d);
                const int c = nd.startOrChild;
                const HNodeG l = nodes[c], r = nodes[c + 1];
                if (d < OPT_MAX_DEPTH) {
                    const double tm = collapse_term(nd, l, r, aggCount[c], aggCount[c + 1], rootHalfArea);
                    if (d < OPT_LDS_DEPTHS) { atomicAdd(&shBin[3 * d], tm); atomicAdd(&shBin[3 * d + 1], tm < 0.0 ? -tm : tm); atomicAdd(&shBin[3 * d + 2], 1.0); }
                    else { atomicAdd(&bins[4 + 3 * d], tm); atomicAdd(&bins[4 + 3 * d + 1], tm < 0.0 ? -tm : tm); atomicAdd(&bins[4 + 3 * d + 2], 1.0); }
                }
                if (d > R - 1 && node_leaf(l) && node_leaf(r)) { first = collapse_term(nd, l, r, l.count, r.count, rootHalfArea); firstAbs = first < 0.0 ? -first : first; }
            }
        }
    }
    sh[0][t] = sah; sh[1][t] = first; sh[2][t] = firstAbs;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) { sh[0][t] += sh[0][t + s]; sh[1][t] += sh[1][t + s]; sh[2][t] += sh[2][t + s]; } __syncthreads(); }
    if (t == 0) { atomicAdd(&bins[0], sh[0][0]); if (sh[2][0] != 0.0) { atomicAdd(&bins[1], sh[1][0]); atomicAdd(&bins[2], sh[2][0]); } atomicMax(maxDepth, shMax); }
    for (int k = t; k < 3 * OPT_LDS_DEPTHS; k += 256) if (shBin[k] != 0.0) atomicAdd(&bins[4 + k], shBin[k]);
}
// the loop's effect: every internal node at depth S + 1 becomes the leaf of its subtree's fragments (everything deeper is then unreachable)
__global__ void k_collapse(HNodeG* nodes, int nodeCount, const int* depth, const int* aggStart, const int* aggCount, int S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount || i < 2) return;
    HNodeG nd = nodes[i];
    if (node_internal(nd) && depth[i] == S + 1) { nd.startOrChild = aggStart[i]; nd.count = aggCount[i]; nodes[i] = nd; }
}
// RemoveEmptySubtrees: the child pairs of the live internal nodes, in id order
__global__ void k_mark_pairs(const HNodeG* nodes, int nodeCount, const int* depth, int S, uint32_t* used)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount || i < 1) return;
    const HNodeG nd = nodes[i];
    if (node_internal(nd) && depth[i] <= S) used[nd.startOrChild >> 1] = 1u;
}
__global__ void k_compact_nodes(const HNodeG* nodes, int pairCount, const uint32_t* used, const uint32_t* rank, HNodeG* out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= pairCount) return;
    if (k == 0) {                                                // nodes 0 (unused) and 1 (root)
        out[0] = nodes[0];
        HNodeG r = nodes[1]; if (node_internal(r)) r.startOrChild = 2 + 2 * (int)rank[r.startOrChild >> 1]; out[1] = r;
        return;
    }
    if (!used[k]) return;
    const int dst = 2 + 2 * (int)rank[k];
    for (int h = 0; h < 2; h++) { HNodeG n = nodes[2 * k + h]; if (node_internal(n)) n.startOrChild = 2 + 2 * (int)rank[n.startOrChild >> 1]; out[dst + h] = n; }
}
// un-indexing without PreSplit (BLAS.GetUnindexedTriangles): leaves in memory order
__global__ void k_leaf_counts(const HNodeG* nodes, int nodeCount, uint32_t* cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount) return;
    cnt[i] = (i >= 2 && node_leaf(nodes[i])) ? (uint32_t)nodes[i].count : 0u;
}
__global__ void k_unindex_plain(HNodeG* nodes, int nodeCount, const uint32_t* at, const int* sorted0, const uint4* tris, uint4* outTris)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount || i < 2) return;
    HNodeG n = nodes[i];
    if (!node_leaf(n)) return;
    const uint32_t w = at[i];
    for (int j = 0; j < n.count; j++) outTris[w + j] = tris[sorted0[n.startOrChild + j]];
    n.startOrChild = (int)w; nodes[i] = n;
}
// un-indexing after PreSplit (PreSplitting.GetUnindexedTriangles): distinct original triangles per leaf, sibling leaves share their common ones
DEV int unique_ids(const HNodeG& leaf, const int* sorted0, const int* origTri, int* ids)
{
    const int n = leaf.count;
    for (int i = 0; i < n; i++) {                                 // insertion sort while loading (leaves are small; collapsed ones a few dozen)
        const int v = origTri[sorted0[leaf.startOrChild + i]];
        int j = i;
        while (j > 0 && ids[j - 1] > v) { ids[j] = ids[j - 1]; j--; }
        ids[j] = v;
    }
    int m = 0;
    for (int i = 0; i < n; i++) if (i == 0 || ids[i] != ids[i - 1]) ids[m++] = ids[i];
    return m;
}
DEV bool has_id(const int* a, int n, int v) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; } return lo < n && a[lo] == v; }
__global__ void k_unindex_ps_count(const HNodeG* nodes, int pairs, const int* sorted0, const int* origTri, int fragCount, int* uniq, int* ucount, uint32_t* adv)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    const int top = 2 + 2 * p;
    const HNodeG l = nodes[top], r = nodes[top + 1];
    int a = 0;
    int* lu = uniq + l.startOrChild; int* ru = uniq + fragCount + r.startOrChild;     // left and right leaves write disjoint halves (a single-leaf root's two copies share one range)
    if (node_leaf(l) && node_leaf(r)) {
        const int nl = unique_ids(l, sorted0, origTri, lu), nr = unique_ids(r, sorted0, origTri, ru);
        ucount[top] = nl; ucount[top + 1] = nr;
        int shared = 0;
        for (int i = 0; i < nl; i++) if (has_id(ru, nr, lu[i])) shared++;
        a = (nl - shared) + nr;
    } else if (node_leaf(l)) { a = ucount[top] = unique_ids(l, sorted0, origTri, lu); }
    else if (node_leaf(r)) { a = ucount[top + 1] = unique_ids(r, sorted0, origTri, ru); }
    adv[p] = (uint32_t)a;
}
__global__ void k_unindex_ps_write(HNodeG* nodes, int pairs, int fragCount, const int* uniq, const int* ucount, const uint32_t* at, const uint4* tris, uint4* outTris)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    const int top = 2 + 2 * p; const int g = (int)at[p];
    HNodeG l = nodes[top], r = nodes[top + 1];
    const int* lu = uniq + l.startOrChild; const int* ru = uniq + fragCount + r.startOrChild;
    if (node_leaf(l) && node_leaf(r)) {
        const int nl = ucount[top], nr = ucount[top + 1];
        int onlyLeft = 0, back = 0;
        for (int i = 0; i < nl; i++) { const int id = lu[i]; if (has_id(ru, nr, id)) { outTris[g + nl - back - 1] = tris[id]; back++; } else outTris[g + onlyLeft++] = tris[id]; }
        int onlyRight = 0;
        for (int i = 0; i < nr; i++) if (!has_id(lu, nl, ru[i])) outTris[g + nl + onlyRight++] = tris[ru[i]];
        l.startOrChild = g; l.count = nl; r.startOrChild = g + onlyLeft; r.count = nr;
        nodes[top] = l; nodes[top + 1] = r;
    } else if (node_leaf(l)) {
        const int n = ucount[top];
        for (int i = 0; i < n; i++) outTris[g + i] = tris[lu[i]];
        l.startOrChild = g; l.count = n; nodes[top] = l;
    } else if (node_leaf(r)) {
        const int n = ucount[top + 1];
        for (int i = 0; i < n; i++) outTris[g + i] = tris[ru[i]];
        r.startOrChild = g; r.count = n; nodes[top + 1] = r;
    }
}
// refit support (BLAS.GetParentIndices / GetLeafIndices, BLAS.cs:481-514)
__global__ void k_parents(const HNodeG* nodes, int nodeCount, int* parents)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount) return;
    if (i < 2) parents[i] = -1;
    if (i >= 1 && !node_leaf(nodes[i])) { parents[nodes[i].startOrChild] = i; parents[nodes[i].startOrChild + 1] = i; }
}
__global__ void k_leaf_flags(const HNodeG* nodes, int nodeCount, uint32_t* flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nodeCount) flag[i] = (i >= 2 && node_leaf(nodes[i])) ? 1u : 0u;
}
__global__ void k_leaf_list(const uint32_t* flag, const uint32_t* rank, int nodeCount, int* leaves)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nodeCount && flag[i]) leaves[rank[i]] = i;
}
// ComputeGlobalSAH of the finished tree: per-node terms, then a fixed-shape reduction (deterministic; differs from the reference's tree-order
// binary64 sum only in the rounding of the additions)
__global__ __launch_bounds__(256) void k_sah_partial(const HNodeG* nodes, int nodeCount, double* part)
{
    __shared__ double sh[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, t = threadIdx.x;
    double v = 0.0;
    if (i >= 1 && i < nodeCount) { const HNodeG nd = nodes[i]; const double prob = (double)node_half_area(nd) * (1.0 / (double)node_half_area(nodes[1])); v = node_leaf(nd) ? (double)(1.1f * (float)nd.count) * prob : (double)1.0f * prob; }
    sh[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sh[t] += sh[t + s]; __syncthreads(); }
    if (t == 0) part[blockIdx.x] = sh[0];
}

} // namespace bvhgpu
