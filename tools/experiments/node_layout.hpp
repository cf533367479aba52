// node_layout.hpp — host side: the DERIVED order of the BLAS node pairs that k_trace2 fetches (DESIGN.md §3 "derived node array").
//
// The reference stores a BLAS as GpuBlasNode[] in the order its builder produces (Bvh/BLAS.cs:245-273: after RemoveEmptySubtrees a
// depth-first order — a pair is followed by the whole subtree of its left node); the traversal (BVHIntersect.glsl:43-102) fetches one
// sibling PAIR (64 B) per step and follows child indices.  Nothing in the traversal depends on WHERE a pair lies, only on the indices
// that lead to it, so a copy of the array with the pairs permuted and the child indices rewritten visits the same boxes in the same
// order with the same arithmetic: hits, T, barycentrics and the visit counters stay bit-identical.  The reference array stays what the
// host uploads / downloads / refits; the derived copy is rebuilt from it on the device (k_derive_nodes) whenever it changes.
//
// What the permutation is for: the L2 fills 128-B lines, a pair is 64 B.  A pair and the child pair the traversal is most likely to
// fetch next should share one 128-B line, so that the second fetch finds the line the first one brought in.  Which parent/child
// couples share a line is a maximum-weight matching on the tree of pairs (weight of an edge = surface area of the node that owns
// the child pair: the SAH's hit probability), solved exactly by the usual two-state tree DP; couples are laid out on even slots,
// pairs that stay single fill the odd holes (the next single in layout order is pulled forward), so the array does not grow.
// Lines are emitted depth-first, larger area first (mode 1), or as treelets — breadth-first blocks of `treeletDepth` line levels,
// then the treelets below (mode 2; a van-Emde-Boas-like order for TLB / Infinity-Cache locality).
#pragma once
#include <stdint.h>
#include <vector>
#include <algorithm>

namespace nodelayout {

struct Node { float mn[3]; uint32_t startOrChild; float mx[3]; uint32_t count; };   // == GpuBlasNode (include/idkpt_types.h)

static inline double half_area(const Node& n) { const double x = (double)n.mx[0] - n.mn[0], y = (double)n.mx[1] - n.mn[1], z = (double)n.mx[2] - n.mn[2]; const double a = (x + y) * z + x * y; return a > 0.0 ? a : 0.0; }
static inline bool internal(const Node& n) { return n.count == 0 && n.startOrChild != 0; }

// newSlot[k] = slot of pair k (nodes 2k, 2k+1) in the derived array; slots 0 (unused node + root) and 1 (the root's children) stay.
// basePair = global pair index of this BLAS's pair 0 (NodeOffset / 2): 128-B alignment is a property of the whole array.
// mode 0: identity; 1: couples + depth-first; 2: couples + treelets.  Returns false (identity) when the nodes are not a tree of aligned
// pairs (an odd child index, a child pair with two parents: never produced by the reference's builder, legal for the traversal).
static bool compute(const Node* nodes, int nodeCount, uint32_t basePair, int mode, int treeletDepth, std::vector<uint32_t>& newSlot)
{
    const int P = nodeCount / 2;
    newSlot.resize((size_t)P);
    for (int k = 0; k < P; k++) newSlot[k] = (uint32_t)k;
    if (mode == 0 || P <= 2) return true;
    // child pairs and their weights
    std::vector<int> ch((size_t)2 * P, 0); std::vector<double> w((size_t)2 * P, 0.0);
    std::vector<char> referenced((size_t)P, 0);
    for (int k = 0; k < P; k++) for (int i = 0; i < 2; i++) {
        const int n = 2 * k + i;
        if (n == 0) continue;                                   // node 0 is unused (the root is node 1)
        const Node& nd = nodes[n];
        if (!internal(nd)) continue;
        const int c = (int)(nd.startOrChild / 2);
        // not a tree of aligned pairs -> identity: an odd child index, or a child pair that two nodes share (a DAG: legal for the traversal — the validation only asks
        // for children behind their parent — but the layout below would emit such a pair once per path to it and run past the BLAS's pair range)
        if ((nd.startOrChild & 1u) || c <= k || c >= P || referenced[c]) { for (int q = 0; q < P; q++) newSlot[q] = (uint32_t)q; return false; }
        referenced[c] = 1;
        ch[n] = c; w[n] = half_area(nd);
    }
    // maximum-weight matching of parent/child pairs (tree DP, children have larger indices than their parents)
    std::vector<double> f0((size_t)P, 0.0), f1((size_t)P, -1.0); std::vector<signed char> pick((size_t)P, -1);
    for (int k = P - 1; k >= 1; k--) {
        double s = 0.0;
        for (int i = 0; i < 2; i++) { const int c = ch[2 * k + i]; if (c) s += std::max(f0[c], f1[c]); }
        f0[k] = s;
        double best = -1.0; int bi = -1;
        for (int i = 0; i < 2; i++) { const int c = ch[2 * k + i]; if (!c) continue; const double v = s - std::max(f0[c], f1[c]) + f0[c] + w[2 * k + i]; if (v > best) { best = v; bi = i; } }
        f1[k] = best; pick[k] = (signed char)bi;
    }
    // top-down: partner[k] = the child pair k shares its line with (0: none); taken[k] = k is the second half of its parent's line
    std::vector<int> partner((size_t)P, 0); std::vector<char> taken((size_t)P, 0);
    for (int k = 1; k < P; k++) {
        const bool asParent = !taken[k] && k != 1 && pick[k] >= 0 && f1[k] > f0[k];   // (pair 1 sits on the odd slot 1: it stays single)
        if (asParent) { partner[k] = ch[2 * k + pick[k]]; taken[partner[k]] = 1; }
    }
    // layout order of the lines ("units": a couple or a single)
    struct Unit { int a, b; };
    std::vector<Unit> seq; seq.reserve((size_t)P);
    auto children_of_unit = [&](const Unit& u, int* out, double* ow) {     // the pairs below a unit, larger area first
        int m = 0;
        const int members[2] = {u.a, u.b};
        for (int q = 0; q < 2; q++) { const int k = members[q]; if (k <= 0) continue; for (int i = 0; i < 2; i++) { const int c = ch[2 * k + i]; if (c && c != u.b) { out[m] = c; ow[m] = w[2 * k + i]; m++; } } }
        for (int x = 1; x < m; x++) for (int y = x; y > 0 && ow[y] > ow[y - 1]; y--) { std::swap(ow[y], ow[y - 1]); std::swap(out[y], out[y - 1]); }
        return m;
    };
    auto unit_of = [&](int k) { Unit u; u.a = k; u.b = partner[k] ? partner[k] : -1; return u; };
    {
        std::vector<int> stack; stack.reserve(256);
        int c0[4]; double w0[4];
        { Unit r; r.a = 1; r.b = -1; const int m = children_of_unit(r, c0, w0); for (int x = m - 1; x >= 0; x--) stack.push_back(c0[x]); }
        if (mode == 2 && treeletDepth > 1) {
            std::vector<int> cur, nxt;
            while (!stack.empty()) {
                const int root = stack.back(); stack.pop_back();
                cur.assign(1, root);
                for (int lvl = 0; lvl < treeletDepth && !cur.empty(); lvl++) {
                    nxt.clear();
                    for (int k : cur) { const Unit u = unit_of(k); seq.push_back(u); const int m = children_of_unit(u, c0, w0); for (int x = 0; x < m; x++) nxt.push_back(c0[x]); }
                    cur.swap(nxt);
                }
                for (int x = (int)cur.size() - 1; x >= 0; x--) stack.push_back(cur[x]);     // the treelets below, first one on top
            }
        } else {
            while (!stack.empty()) {
                const int k = stack.back(); stack.pop_back();
                const Unit u = unit_of(k); seq.push_back(u);
                const int m = children_of_unit(u, c0, w0);
                for (int x = m - 1; x >= 0; x--) stack.push_back(c0[x]);
            }
        }
    }
    // slots: couples on even (128-B aligned) global slots; an odd slot in front of a couple takes the next single of the order
    const size_t n = seq.size();
    std::vector<char> placed(n, 0);
    uint32_t cursor = 2; size_t ns = 0;
    for (size_t i = 0; i < n; i++) {
        if (placed[i]) continue;
        const Unit u = seq[i];
        if (u.b < 0) { newSlot[u.a] = cursor++; placed[i] = 1; continue; }
        if ((basePair + cursor) & 1u) {
            if (ns <= i) ns = i + 1;
            while (ns < n && (placed[ns] || seq[ns].b >= 0)) ns++;
            if (ns < n) { newSlot[seq[ns].a] = cursor++; placed[ns] = 1; }
            else { newSlot[u.a] = cursor++; newSlot[u.b] = cursor++; placed[i] = 1; continue; }   // no single left: the couple straddles two lines
        }
        newSlot[u.a] = cursor; newSlot[u.b] = cursor + 1; cursor += 2; placed[i] = 1;
    }
    // (pairs that the root cannot reach keep no slot of their own above: give them the remaining ones so that the map stays a permutation)
    if (cursor < (uint32_t)P) {
        std::vector<char> used((size_t)P, 0), reached((size_t)P, 0);
        reached[0] = reached[1] = 1; for (const Unit& u : seq) { reached[u.a] = 1; if (u.b > 0) reached[u.b] = 1; }
        for (int k = 0; k < P; k++) if (reached[k]) used[newSlot[k]] = 1;
        uint32_t freeSlot = 0;
        for (int k = 0; k < P; k++) if (!reached[k]) { while (used[freeSlot]) freeSlot++; newSlot[k] = freeSlot; used[freeSlot] = 1; }
    }
    return true;
}

} // namespace nodelayout
