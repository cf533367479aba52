// libidkbvh.so — native SweepSAH (+PreSplit) BLAS builder, PLOC TLAS builder, CPU refit.  See include/idkbvh.h.
//
// Host-side product code (C++17, SSE, std::thread).  Must produce the same bytes as the reference's C# builder
// (Source/Bvh/{BLAS,PreSplitting,TLAS}.cs), so every float operation mirrors it:
//   * boxes are 4-lane SSE registers updated with minps/maxps (Vector128.MinNative/MaxNative, Shapes/Box.cs:40-70),
//   * HalfArea = fma(x+y, z, x*y) (Utils/MyMath.cs:222-229),
//   * no contraction anywhere else (-ffp-contract=off), cbrtf/rintf for MathF.Cbrt/MathF.Round.
// Node ids do not depend on scheduling: a subtree with L fragments owns the id range reserved for it up front
// (Bvh/BLAS.cs:221-241), which is what makes the threaded build deterministic.
#include <stdint.h>
#include <stdio.h>
#include <sys/resource.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <immintrin.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <cmath>
#include <memory>
#include <utility>
#include <sched.h>
#include <new>
#include "../../include/idkbvh.h"

namespace {
static std::atomic<int> g_phaseTiming{0};   // idkbvhSetPhaseTiming

constexpr int kThreadedRecursionThreshold = 1 << 13; // BLAS.cs:28
constexpr int kThreadedSortingThreshold = 1 << 16;   // BLAS.cs:29
constexpr int kParallelSweepThreshold = 1 << 15;     // ours: nodes this large sweep their three axes on three threads
constexpr float kTraversalCost = 1.0f;               // BLAS.cs:26
constexpr float kTriangleCost = 1.1f;                // BuildSettings defaults, BLAS.cs:31-48
constexpr int kStopSplittingThreshold = 1;
constexpr int kMaxLeafTriangleCount = 2;
// Threads a build may use by default: the hardware threads this process may run on (affinity mask), capped by the container's CPU quota
// (cgroup v2 cpu.max / v1 cfs quota).  More runnable threads than the quota get the whole process throttled for the rest of the scheduler period.
static int defaultThreadCount()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) n = std::min(n, a); }
    double quota = 0.0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[32] = {0}; long long per = 0; if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / (double)per; fclose(f); }
    else {
        long long q = -1, per = 0;
        if (FILE* a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%lld", &q) != 1) q = -1; fclose(a); }
        if (FILE* b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%lld", &per) != 1) per = 0; fclose(b); }
        if (q > 0 && per > 0) quota = (double)q / (double)per;
    }
    if (quota > 0.0) n = std::min(n, std::max(1, (int)(quota + 0.5)));
    return n;
}

constexpr int kStackOptThreshold = 16;
constexpr float kStackOptSahIncreaseAcceptance = 0.0009745f;

struct alignas(16) SBox {
    __m128 mn, mx;
    static SBox empty() { return {_mm_set1_ps(FLT_MAX), _mm_set1_ps(-FLT_MAX)}; }
    static SBox point(__m128 p) { return {p, p}; }
    void grow(__m128 p) { mn = _mm_min_ps(mn, p); mx = _mm_max_ps(mx, p); }
    void grow(const SBox& b) { mn = _mm_min_ps(mn, b.mn); mx = _mm_max_ps(mx, b.mx); }
    void clip(const SBox& b) { mn = _mm_max_ps(mn, b.mn); mx = _mm_min_ps(mx, b.mx); }
    void size(float s[4]) const { _mm_storeu_ps(s, _mm_sub_ps(mx, mn)); }
    float halfArea() const { float s[4]; size(s); return fmaf(s[0] + s[1], s[2], s[0] * s[1]); }
    float area() const { return halfArea() * 2.0f; }
    float largestExtent() const { float s[4]; size(s); float a = s[1] > s[2] ? s[1] : s[2]; return s[0] > a ? s[0] : a; }
    int largestAxis() const { float s[4]; size(s); int a = 0; if (s[0] < s[1]) a = 1; if (s[a] < s[2]) a = 2; return a; }
    float lo(int a) const { float v[4]; _mm_storeu_ps(v, mn); return v[a]; }
    float hi(int a) const { float v[4]; _mm_storeu_ps(v, mx); return v[a]; }
};
inline __m128 load3(const float* p) { return _mm_set_ps(0.0f, p[2], p[1], p[0]); }

struct HNode { float mn[3]; int32_t startOrChild; float mx[3]; int32_t count; };
// Allocator of the build's large arrays: with NoInit, resize() / the sized constructor leave new elements uninitialised (they are overwritten in
// full, by several threads or by a device copy: value-initialising 85 MB first would be a serial pass that also puts every page on the calling
// thread); assign(n, v) and copies behave as usual.  (Transparent huge pages were tried for these blocks — 43 faults instead of 21 000 for the
// node array — and dropped: with defrag = madvise a 2 MB fault can stall for milliseconds in compaction, 5 ms each on the development container.)
template <class T, bool NoInit = false> struct BigAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = BigAlloc<U, NoInit>; };
    template <class U, class... A> void construct(U* p, A&&... a)
    {
        if constexpr (NoInit && sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using BigVec = std::vector<T, BigAlloc<T>>;
using NodeVec = std::vector<HNode, BigAlloc<HNode, true>>;
static_assert(sizeof(HNode) == sizeof(GpuBlasNode), "node layout");
inline bool isLeaf(const HNode& n) { return n.count > 0; }
inline float nodeHalfArea(const HNode& n) { float x = n.mx[0] - n.mn[0], y = n.mx[1] - n.mn[1], z = n.mx[2] - n.mn[2]; return fmaf(x + y, z, x * y); }
inline void setBounds(HNode& n, const SBox& b) { float v[4]; _mm_storeu_ps(v, b.mn); n.mn[0] = v[0]; n.mn[1] = v[1]; n.mn[2] = v[2]; _mm_storeu_ps(v, b.mx); n.mx[0] = v[0]; n.mx[1] = v[1]; n.mx[2] = v[2]; }
inline SBox nodeBox(const HNode& n) { return {load3(n.mn), load3(n.mx)}; }

inline int satInt(float f) { if (f != f) return 0; if (f >= 2147483648.0f) return INT32_MAX; if (f <= -2147483648.0f) return INT32_MIN; return (int)f; }
inline uint32_t satUInt(float f) { if (f != f || f <= 0.0f) return 0u; if (f >= 4294967296.0f) return UINT32_MAX; return (uint32_t)f; }
inline uint32_t floatToKey(float v) { uint32_t f; memcpy(&f, &v, 4); return f ^ (uint32_t)(((int32_t)f >> 31) | (int32_t)0x80000000); } // Algorithms.cs:15-34

// 3 x 11-bit stable LSD radix sort of ids by key (Algorithms.cs:45-112); keys are precomputed once (same values the
// reference recomputes per pass)
void radixSortIds(const uint32_t* keys, int n, int* out)
{
    std::vector<int> tmp(n);
    std::vector<int> hist(3 * 2048, 0);
    for (int i = 0; i < n; i++) { uint32_t k = keys[i]; hist[k & 2047]++; hist[2048 + ((k >> 11) & 2047)]++; hist[4096 + (k >> 22)]++; }
    for (int p = 0; p < 3; p++) { int sum = 0; int* h = &hist[p * 2048]; for (int i = 0; i < 2048; i++) { int t = h[i]; h[i] = sum; sum += t; } }
    int* a = out; int* b = tmp.data();
    for (int i = 0; i < n; i++) a[i] = i;                          // Helper.FillIncreasing
    // pass 0: a -> b, pass 1: b -> a, pass 2: a -> b ; then copy to out (the reference ends in its `output` buffer too)
    for (int i = 0; i < n; i++) { int id = a[i]; b[hist[keys[id] & 2047]++] = id; }
    for (int i = 0; i < n; i++) { int id = b[i]; a[hist[2048 + ((keys[id] >> 11) & 2047)]++] = id; }
    for (int i = 0; i < n; i++) { int id = a[i]; b[hist[4096 + (keys[id] >> 22)]++] = id; }
    memcpy(out, b, sizeof(int) * (size_t)n);
}

struct Builder {
    // inputs
    const float* positions; const GpuBlasTriangle* tris; int triCount;
    // fragments
    BigVec<SBox> frag; BigVec<int> origTri;
    // build state
    BigVec<int> sorted[3]; BigVec<float> rightCosts; BigVec<uint8_t> leftTable;
    NodeVec nodes;
    int requiredStack = 0;
    std::atomic<int> liveThreads{0};   // workers inside buildSubtree (+2 while a node sweeps its axes on extra threads)
    int maxThreads = 1;
    std::atomic<int> error{0};   // 3: fragment count out of range, 4: PreSplit stack (the reference throws), 5: non-finite vertex position
    // outputs
    BigVec<GpuBlasTriangle> outTris; std::vector<int> parents, leaves;
    double sah = 0.0, buildMs = 0.0;

    void triPoints(int i, __m128& a, __m128& b, __m128& c) const { const GpuBlasTriangle& t = tris[i]; a = load3(positions + 3 * (size_t)t.X); b = load3(positions + 3 * (size_t)t.Y); c = load3(positions + 3 * (size_t)t.Z); }
    SBox triBox(int i) const { __m128 a, b, c; triPoints(i, a, b, c); SBox bx = SBox::point(a); bx.grow(b); bx.grow(c); return bx; } // Box.From(triangle)

    // ---------------- PreSplitting.PreSplit (PreSplitting.cs:26-160)
    static float triArea(__m128 a, __m128 b, __m128 c)
    {
        float p0[4], p1[4], p2[4]; _mm_storeu_ps(p0, a); _mm_storeu_ps(p1, b); _mm_storeu_ps(p2, c);
        float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        return sqrtf((cx * cx) + (cy * cy) + (cz * cz)) * 0.5f;
    }
    float priority(int i) const { __m128 a, b, c; triPoints(i, a, b, c); SBox bx = SBox::point(a); bx.grow(b); bx.grow(c); float e = bx.largestExtent(); return cbrtf((e * e) * (bx.area() - triArea(a, b, c))); }
    static int splitCount(float prio, float total, int n, float factor) { float share = prio / total * (float)n; return 1 + satInt(share * factor); }
    template <class F> void parallelFor(int n, F&& body) const   // body(begin, end) over contiguous chunks; chunks are independent
    {
        const int t = std::max(1, std::min(maxThreads, n / 4096));
        if (t == 1) { body(0, n); return; }
        std::vector<std::thread> pool;
        for (int k = 1; k < t; k++) pool.emplace_back([&, k] { body((int)((long long)n * k / t), (int)((long long)n * (k + 1) / t)); });
        body(0, (int)((long long)n / t));
        for (auto& th : pool) th.join();
    }
    void preSplit(float factor)
    {
        std::vector<float> prio(triCount);
        parallelFor(triCount, [&](int b, int e) { for (int i = b; i < e; i++) prio[i] = priority(i); });
        float total = 0.0f;
        for (int i = 0; i < triCount; i++) total += prio[i];                       // binary32 running sum in index order (PreSplitting.cs:40-48)
        std::vector<size_t> first((size_t)triCount + 1);
        size_t count = 0;
        for (int i = 0; i < triCount; i++) { first[i] = count; count += (size_t)(uint32_t)splitCount(prio[i], total, triCount, factor); }
        first[triCount] = count;
        if (count > ((size_t)1 << 27)) { error = 3; return; }                     // (a split factor or priorities that ask for more fragments than any array here can hold)
        frag.resize(count); origTri.resize(count);
        SBox global = SBox::empty();
        for (int i = 0; i < triCount; i++) { __m128 a, b, c; triPoints(i, a, b, c); global.grow(a); global.grow(b); global.grow(c); }
        float gsz[4]; global.size(gsz);
        struct Item { SBox box; int splits; };
        // every triangle writes its own fragment range [first[i], first[i+1]) in the order the serial loop emits them
        parallelFor(triCount, [&](int tb0, int te0) {
        Item stack[64];
        for (int i = tb0; i < te0; i++) {
            size_t w = first[i];
            __m128 pa, pb, pc; triPoints(i, pa, pb, pc);
            float P[3][4]; _mm_storeu_ps(P[0], pa); _mm_storeu_ps(P[1], pb); _mm_storeu_ps(P[2], pc);
            __m128 PV[3] = {pa, pb, pc};
            int sp = 0;
            SBox tb = SBox::point(pa); tb.grow(pb); tb.grow(pc);
            stack[sp++] = {tb, splitCount(prio[i], total, triCount, factor)};
            while (sp > 0) {
                Item it = stack[--sp];
                if (it.splits == 1) { frag[w] = it.box; origTri[w] = i; w++; continue; }
                int axis = it.box.largestAxis();
                float ext = it.box.largestExtent();
                float alpha = ext / gsz[axis];
                uint32_t bits; memcpy(&bits, &alpha, 4); bits &= (255u << 23); float p2; memcpy(&p2, &bits, 4);
                float nodeSize = p2 * gsz[axis];
                if (nodeSize >= ext - 0.0001f) nodeSize *= 0.5f;
                float mid = (it.box.lo(axis) + it.box.hi(axis)) * 0.5f;
                float idx = rintf((mid - global.lo(axis)) / nodeSize);
                float pos = global.lo(axis) + idx * nodeSize;
                // Triangle.Split (Shapes/Triangle.cs:48-97)
                SBox lb = SBox::empty(), rb = SBox::empty();
                bool q[3];
                for (int v = 0; v < 3; v++) { q[v] = P[v][axis] <= pos; if (q[v]) lb.grow(PV[v]); else rb.grow(PV[v]); }
                for (int e = 0; e < 3; e++) {
                    int a = e, b = (e + 1) % 3;
                    if (q[a] ^ q[b]) {
                        float t = (pos - P[a][axis]) / (P[b][axis] - P[a][axis]);
                        __m128 m = _mm_add_ps(PV[a], _mm_mul_ps(_mm_set1_ps(t), _mm_sub_ps(PV[b], PV[a])));
                        m = _mm_blend_ps(m, _mm_setzero_ps(), 8); // Vector3 -> Vector128(x,y,z,0)
                        lb.grow(m); rb.grow(m);
                    }
                }
                lb.clip(it.box); rb.clip(it.box);
                float le = lb.largestExtent(), re = rb.largestExtent();
                int lc = satInt((float)it.splits * (le / (le + re)));
                lc = std::min(std::max(lc, 1), it.splits - 1);
                if (sp + 2 > 64) { error = 4; sp = 0; break; }                    // the reference's stackalloc of 64 entries (PreSplitting.cs:57) would have thrown
                stack[sp++] = {rb, it.splits - lc};
                stack[sp++] = {lb, lc};
            }
        }
        });
    }

    // ---------------- build
    SBox boundsOf(int start, int count, int axis) const { SBox b = SBox::empty(); const int* ids = sorted[axis].data() + start; for (int i = 0; i < count; i++) b.grow(frag[ids[i]]); return b; }
    static int stablePartition(int* src, int n, int* aux, const uint8_t* table)
    {
        int l = 0, r = 0;
        for (int i = 0; i < n; i++) { int id = src[i]; if (table[id]) src[l++] = id; else aux[r++] = id; }
        memcpy(src + l, aux, sizeof(int) * (size_t)r);
        return l;
    }
    // One axis of the sweep of BLAS.TrySplit (BLAS.cs:760-821): right-to-left accumulation of the right costs, then the left sweep.
    // `bestCost` carries the best cost found so far (the reference prunes both sweeps against it); on return it, bestIndex and the
    // return value (true = this axis improved on it) are updated.
    bool sweepAxis(int axis, int start, int end, float* rc, float& bestCost, int& bestIndex) const
    {
        bool improved = false;
        const int* ids = sorted[axis].data();
        int firstRight = start + 1;
        SBox acc = SBox::empty(); float cnt = 0.0f;
        for (int i = end - 1; i >= firstRight; i--) {
            cnt++; acc.grow(frag[ids[i]]);
            float c = acc.halfArea() * cnt;
            rc[i] = c;
            if (c >= bestCost) { firstRight = i + 1; break; }
        }
        SBox lacc = SBox::empty(); float lcnt = (float)(firstRight - start) - 1.0f;
        for (int i = start; i < firstRight - 1; i++) lacc.grow(frag[ids[i]]);
        for (int i = firstRight - 1; i < end - 1; i++) {
            lcnt++; lacc.grow(frag[ids[i]]);
            float lcost = lacc.halfArea() * lcnt;
            float cost = lcost + rc[i + 1];
            if (cost < bestCost) { bestIndex = i + 1; bestCost = cost; improved = true; }
            else if (lcost >= bestCost) break;
        }
        return improved;
    }
    // BLAS.TrySplit (BLAS.cs:730-873). Returns split index or -1.
    // Large nodes sweep the three axes on three threads, each pruning only against its own running best.  The outcome is the one of the
    // serial sweep: both prune only positions whose cost cannot be strictly below the final best (the partial costs are monotone), so
    // each axis reports (its minimum, the first index reaching it) and the axes are then compared in order with the reference's strict '<'.
    int trySplit(const HNode& parent)
    {
        if (parent.count <= kStopSplittingThreshold) return -1;
        const int start = parent.startOrChild, end = start + parent.count;
        const int n = (int)frag.size();
        int bestAxis = 0, bestIndex = 0; float bestCost = FLT_MAX;
        float* rcAxis[3] = {rightCosts.data(), rightCosts.data() + n, rightCosts.data() + 2 * (size_t)n};
        bool wide = false;
        if (parent.count >= kParallelSweepThreshold && maxThreads > 2) {
            int cur = liveThreads.load();
            while (cur + 2 <= maxThreads && !wide) wide = liveThreads.compare_exchange_weak(cur, cur + 2);
        }
        if (wide) {
            float c[3] = {FLT_MAX, FLT_MAX, FLT_MAX}; int ix[3] = {0, 0, 0}; bool ok[3] = {false, false, false};
            std::thread t1([&] { ok[1] = sweepAxis(1, start, end, rcAxis[1], c[1], ix[1]); }), t2([&] { ok[2] = sweepAxis(2, start, end, rcAxis[2], c[2], ix[2]); });
            ok[0] = sweepAxis(0, start, end, rcAxis[0], c[0], ix[0]);
            t1.join(); t2.join();
            for (int axis = 0; axis < 3; axis++) if (ok[axis] && c[axis] < bestCost) { bestCost = c[axis]; bestIndex = ix[axis]; bestAxis = axis; }
        } else {
            for (int axis = 0; axis < 3; axis++) if (sweepAxis(axis, start, end, rcAxis[0], bestCost, bestIndex)) bestAxis = axis;
        }
        auto release = [&] { if (wide) liveThreads -= 2; };
        if (parent.count <= kMaxLeafTriangleCount) {
            float notSplit = kTriangleCost * (float)parent.count;
            float newCost = kTraversalCost + (kTriangleCost * bestCost / nodeHalfArea(parent));
            if (newCost >= notSplit) { release(); return -1; }
        }
        SBox lb, rb;
        if (wide) { std::thread t([&] { rb = boundsOf(bestIndex, end - bestIndex, bestAxis); }); lb = boundsOf(start, bestIndex - start, bestAxis); t.join(); }
        else { lb = boundsOf(start, bestIndex - start, bestAxis); rb = boundsOf(bestIndex, end - bestIndex, bestAxis); }
        const bool swap = lb.halfArea() < rb.halfArea();
        int* ids = sorted[bestAxis].data();
        for (int i = start; i < bestIndex; i++) leftTable[ids[i]] = !swap;
        for (int i = bestIndex; i < end; i++) leftTable[ids[i]] = swap;
        const int a1 = (bestAxis + 1) % 3, a2 = (bestAxis + 2) % 3;
        if (wide) {
            // the three id arrays are partitioned independently (each with its own scratch: the right-cost rows are free again)
            std::thread t1([&] { stablePartition(sorted[a1].data() + start, parent.count, reinterpret_cast<int*>(rcAxis[1] + start), leftTable.data()); });
            std::thread t2([&] { stablePartition(sorted[a2].data() + start, parent.count, reinterpret_cast<int*>(rcAxis[2] + start), leftTable.data()); });
            if (swap) bestIndex = start + stablePartition(ids + start, parent.count, reinterpret_cast<int*>(rcAxis[0] + start), leftTable.data());
            t1.join(); t2.join();
        } else {
            int* aux = reinterpret_cast<int*>(rcAxis[0] + start);
            if (swap) bestIndex = start + stablePartition(ids + start, parent.count, aux, leftTable.data());
            stablePartition(sorted[a1].data() + start, parent.count, aux, leftTable.data());
            stablePartition(sorted[a2].data() + start, parent.count, aux, leftTable.data());
        }
        release();
        return bestIndex;
    }
    // Subtrees are independent once their id range is reserved (BLAS.cs:221-241), so they are built by a pool of workers fed from one
    // shared list: a worker walks its subtree depth-first and hands every right child that is still large to the list.  Which worker
    // builds what does not matter for the result.
    struct Task { int parent, fresh; };
    std::vector<Task> shared; std::mutex mtx; std::condition_variable cv; int pendingTasks = 0;
    void pushShared(Task t) { { std::lock_guard<std::mutex> lk(mtx); shared.push_back(t); pendingTasks++; } cv.notify_one(); }
    void buildSubtree(Task first)
    {
        std::vector<Task> todo; todo.push_back(first);
        while (!todo.empty()) {
            Task t = todo.back(); todo.pop_back();
            HNode& p = nodes[t.parent];
            setBounds(p, boundsOf(p.startOrChild, p.count, 0));
            int split = trySplit(p);
            if (split < 0) continue;
            HNode l = {}, r = {};
            l.startOrChild = p.startOrChild; l.count = split - l.startOrChild;
            r.startOrChild = split; r.count = p.count - l.count;
            const int lid = t.fresh, rid = lid + 1;
            nodes[lid] = l; nodes[rid] = r;
            p.startOrChild = lid; p.count = 0;
            const int leftFresh = rid + 1, rightFresh = rid + (2 * l.count - 1);
            if (maxThreads > 1 && std::min(l.count, r.count) >= kThreadedRecursionThreshold) pushShared({rid, rightFresh});
            else todo.push_back({rid, rightFresh});
            todo.push_back({lid, leftFresh}); // left first (order is irrelevant for the result, ids are pre-reserved)
        }
    }
    void worker()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mtx);
                cv.wait(lk, [&] { return !shared.empty() || pendingTasks == 0; });
                if (shared.empty()) return;
                t = shared.back(); shared.pop_back();
            }
            liveThreads++;
            buildSubtree(t);
            liveThreads--;
            bool done;
            { std::lock_guard<std::mutex> lk(mtx); done = --pendingTasks == 0; }
            if (done) cv.notify_all();
        }
    }
    void buildAll()
    {
        liveThreads = 0;
        pushShared({1, 2});
        std::vector<std::thread> pool;
        for (int i = 1; i < maxThreads; i++) pool.emplace_back([this] { worker(); });
        worker();
        for (auto& th : pool) th.join();
    }
    // ---- whole-tree walks (RequiredStackSize, GlobalSAH, CollapseDeepest).  Each is a depth-first walk whose binary64 sums depend on the visiting
    // order.  They run in parallel without changing a bit: the tree is cut at depth kCutDepth, every subtree below the cut is walked by a task
    // that records its terms IN ORDER instead of adding them, and the part above the cut is then walked serially, splicing the recorded
    // sequences in where the serial walk would have produced them; the additions happen once, in the serial order.
    static constexpr int kCutDepth = 9;
    void cutTree(int parentId, int depth, std::vector<int>& roots) const   // internal nodes at depth kCutDepth, in left-first order
    {
        if (depth == kCutDepth) { roots.push_back(parentId); return; }
        const int c = nodes[parentId].startOrChild;
        if (!isLeaf(nodes[c])) cutTree(c, depth + 1, roots);
        if (!isLeaf(nodes[c + 1])) cutTree(c + 1, depth + 1, roots);
    }
    template <class F> void runTasks(int n, F&& task) const   // dynamic: subtrees of an SAH tree differ widely in size
    {
        const int t = std::max(1, std::min(maxThreads, n));
        if (t == 1) { for (int i = 0; i < n; i++) task(i); return; }
        std::atomic<int> next{0};
        auto loop = [&] { for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) task(i); };
        std::vector<std::thread> pool;
        for (int k = 1; k < t; k++) pool.emplace_back(loop);
        loop();
        for (auto& th : pool) th.join();
    }
    int requiredStackSize(int nodeId = 2) const // BLAS.cs:672-702
    {
        const HNode& l = nodes[nodeId]; const HNode& r = nodes[nodeId + 1];
        bool tl = !isLeaf(l), tr = !isLeaf(r);
        if (tl && tr) return std::max(requiredStackSize(l.startOrChild), requiredStackSize(r.startOrChild)) + 1;
        if (tl || tr) return requiredStackSize(tl ? l.startOrChild : r.startOrChild);
        return 0;
    }
    int requiredStackTop(int parentId, int depth, const std::vector<int>& below, size_t& next) const
    {
        if (depth == kCutDepth) return below[next++];
        const int c = nodes[parentId].startOrChild;
        const bool tl = !isLeaf(nodes[c]), tr = !isLeaf(nodes[c + 1]);
        const int a = tl ? requiredStackTop(c, depth + 1, below, next) : 0;
        const int b = tr ? requiredStackTop(c + 1, depth + 1, below, next) : 0;
        return (tl && tr) ? std::max(a, b) + 1 : (tl ? a : b);
    }
    int requiredStackSizeAll() const
    {
        std::vector<int> roots; cutTree(1, 0, roots);
        std::vector<int> below(roots.size());
        runTasks((int)roots.size(), [&](int k) { below[k] = requiredStackSize(nodes[roots[k]].startOrChild); });
        size_t next = 0;
        return requiredStackTop(1, 0, below, next);
    }
    template <class Add> void sahTerms(int rootId, double rootArea, Add&& add) const // BLAS.cs:629-657: pre-order, left subtree first
    {
        std::vector<int> st; st.push_back(rootId);
        while (!st.empty()) {
            const HNode& n = nodes[st.back()]; st.pop_back();
            double prob = (double)nodeHalfArea(n) * rootArea;
            if (isLeaf(n)) add((double)(kTriangleCost * (float)n.count) * prob);
            else { add((double)kTraversalCost * prob); st.push_back(n.startOrChild + 1); st.push_back(n.startOrChild); }
        }
    }
    void sahTop(int nodeId, int depth, double rootArea, const std::vector<std::vector<double>>& below, size_t& next, double& cost) const
    {
        const HNode& n = nodes[nodeId];
        if (!isLeaf(n) && depth == kCutDepth) { for (double t : below[next]) cost += t; next++; return; }
        double prob = (double)nodeHalfArea(n) * rootArea;
        if (isLeaf(n)) { cost += (double)(kTriangleCost * (float)n.count) * prob; return; }
        cost += (double)kTraversalCost * prob;
        sahTop(n.startOrChild, depth + 1, rootArea, below, next, cost);
        sahTop(n.startOrChild + 1, depth + 1, rootArea, below, next, cost);
    }
    double globalSAH() const
    {
        const double rootArea = 1.0 / (double)nodeHalfArea(nodes[1]);
        std::vector<int> roots; cutTree(1, 0, roots);
        std::vector<std::vector<double>> below(roots.size());
        runTasks((int)roots.size(), [&](int k) { sahTerms(roots[k], rootArea, [&](double t) { below[k].push_back(t); }); });
        double cost = 0.0; size_t next = 0;
        sahTop(1, 0, rootArea, below, next, cost);
        return cost;
    }
    template <class Add> void collapseDeepest(int newStackSize, bool firstPass, Add&& add, int parentId, int depth) // BLAS.cs:897-936
    {
        HNode& p = nodes[parentId];
        const int c = p.startOrChild;
        if (!isLeaf(nodes[c])) collapseDeepest(newStackSize, firstPass, add, c, depth + 1);
        if (!isLeaf(nodes[c + 1])) collapseDeepest(newStackSize, firstPass, add, c + 1, depth + 1);
        const HNode& l = nodes[c]; const HNode& r = nodes[c + 1];
        if (isLeaf(l) && isLeaf(r)) {
            if (depth > newStackSize && !firstPass) { p.startOrChild = l.startOrChild; p.count = l.count + r.count; }
            if ((depth == newStackSize && !firstPass) || (depth > newStackSize && firstPass)) {
                // StackOptMaxLeafTriangleCount = int.MaxValue: the guard at BLAS.cs:924-928 can never fire
                double leavesCost = (double)kTriangleCost * ((double)l.count * (double)nodeHalfArea(l) + (double)r.count * (double)nodeHalfArea(r));
                double newParentLeafCost = (double)kTriangleCost * (double)(l.count + r.count);
                add(((double)nodeHalfArea(p) * (newParentLeafCost - (double)kTraversalCost) - leavesCost) / rootHalfArea);
            }
        }
    }
    void collapseTop(int parentId, int depth, const std::vector<std::vector<double>>& below, size_t& next, double& nextCost) const
    {
        if (depth == kCutDepth) { for (double t : below[next]) nextCost += t; next++; return; }
        const int c = nodes[parentId].startOrChild;   // above the cut a node is never deep enough to be touched (newStackSize > kCutDepth): only the order matters
        if (!isLeaf(nodes[c])) collapseTop(c, depth + 1, below, next, nextCost);
        if (!isLeaf(nodes[c + 1])) collapseTop(c + 1, depth + 1, below, next, nextCost);
    }
    double rootHalfArea = 0.0;
    void collapseDeepestAll(int newStackSize, bool firstPass, double& nextCost)
    {
        rootHalfArea = (double)nodeHalfArea(nodes[1]);   // the root is never collapsed
        if (newStackSize <= kCutDepth) { collapseDeepest(newStackSize, firstPass, [&](double t) { nextCost += t; }, 1, 0); return; }
        std::vector<int> roots; cutTree(1, 0, roots);
        std::vector<std::vector<double>> below(roots.size());
        runTasks((int)roots.size(), [&](int k) { collapseDeepest(newStackSize, firstPass, [&](double t) { below[k].push_back(t); }, roots[k], kCutDepth); });
        size_t next = 0;
        collapseTop(1, 0, below, next, nextCost);
    }
    void optimizeStackSize() // BLAS.cs:875-895
    {
        requiredStack = requiredStackSizeAll();
        if (timing) fprintf(stderr, "[idkbvh] required stack before optimisation %d\n", requiredStack);
        if (requiredStack < kStackOptThreshold) return;
        double current = globalSAH(), added = 0.0;
        collapseDeepestAll(requiredStack - 1, true, added);
        double inc = added / current;
        while (inc <= (double)kStackOptSahIncreaseAcceptance && requiredStack > 0) { collapseDeepestAll(--requiredStack, false, added); inc = added / current; }
    }
    // RemoveEmptySubtrees, BLAS.cs:245-273: child pairs renumbered in pre-order of their parents (left first), pair k at slot 2 + 2k.
    // Out of place and by subtree: internal-node counts below the cut give every subtree its first slot, then the subtrees copy themselves.
    int countInternal(int rootId) const
    {
        int n = 0;
        std::vector<int> st; st.push_back(rootId);
        while (!st.empty()) {
            const int c = nodes[st.back()].startOrChild; st.pop_back(); n++;
            if (!isLeaf(nodes[c + 1])) st.push_back(c + 1);
            if (!isLeaf(nodes[c])) st.push_back(c);
        }
        return n;
    }
    void compactSubtree(NodeVec& out, int oldRoot, int newRoot, int counter) const
    {
        std::vector<std::pair<int, int>> st; st.push_back({oldRoot, newRoot});
        while (!st.empty()) {
            const int o = st.back().first, nw = st.back().second; st.pop_back();
            const int c = nodes[o].startOrChild;
            out[counter] = nodes[c]; out[counter + 1] = nodes[c + 1];
            out[nw].startOrChild = counter;
            if (!isLeaf(nodes[c + 1])) st.push_back({c + 1, counter + 1});
            if (!isLeaf(nodes[c])) st.push_back({c, counter});
            counter += 2;
        }
    }
    struct CutSlot { int newId, firstPair; };
    void compactTop(NodeVec& out, int oldId, int newId, int depth, const std::vector<int>& internal, std::vector<CutSlot>& slots, int& pre) const
    {
        if (depth == kCutDepth) { slots.push_back({newId, pre}); pre += internal[slots.size() - 1]; return; }
        const int c = nodes[oldId].startOrChild, slot = 2 + 2 * pre++;
        out[slot] = nodes[c]; out[slot + 1] = nodes[c + 1];
        out[newId].startOrChild = slot;
        if (!isLeaf(nodes[c])) compactTop(out, c, slot, depth + 1, internal, slots, pre);
        if (!isLeaf(nodes[c + 1])) compactTop(out, c + 1, slot + 1, depth + 1, internal, slots, pre);
    }
    int compactNodes()
    {
        std::vector<int> roots; cutTree(1, 0, roots);
        std::vector<int> internal(roots.size());
        runTasks((int)roots.size(), [&](int k) { internal[k] = countInternal(roots[k]); });
        int total = 0; for (int v : internal) total += v;
        int top = 0;   // internal nodes above the cut
        { std::vector<std::pair<int, int>> st; st.push_back({1, 0});
          while (!st.empty()) { const int o = st.back().first, d = st.back().second; st.pop_back(); if (d == kCutDepth) continue; top++;
                                const int c = nodes[o].startOrChild; if (!isLeaf(nodes[c])) st.push_back({c, d + 1}); if (!isLeaf(nodes[c + 1])) st.push_back({c + 1, d + 1}); } }
        NodeVec out((size_t)2 + 2 * (size_t)(total + top));
        out[0] = nodes[0]; out[1] = nodes[1];
        std::vector<CutSlot> slots; slots.reserve(roots.size());
        int pre = 0;
        compactTop(out, 1, 1, 0, internal, slots, pre);
        runTasks((int)roots.size(), [&](int k) { compactSubtree(out, roots[k], slots[k].newId, 2 + 2 * slots[k].firstPair); });
        nodes.swap(out);
        return (int)nodes.size();
    }
    // Both un-indexing passes walk the compacted tree in the order RemoveEmptySubtrees numbered it (pre-order over the parents, left first), so
    // "the next leaf / sibling pair of the walk" is simply the next one in memory: sizes per pair -> running offsets -> independent writes.
    void unindexPlain() // BLAS.GetUnindexedTriangles, BLAS.cs:441-466
    {
        // single-leaf root: its leaf is duplicated into nodes 2 and 3 (BLAS.cs:173-183; the reference throws here) ->
        // size by the sum of leaf counts so the duplicated triangles are simply stored twice
        const int nn = (int)nodes.size();
        std::vector<int> at((size_t)nn + 1, 0);
        for (int i = 2; i < nn; i++) at[i + 1] = at[i] + (isLeaf(nodes[i]) ? nodes[i].count : 0);
        outTris.resize((size_t)at[nn]);
        parallelFor(nn, [&](int b, int e) {
            for (int i = std::max(b, 2); i < e; i++) {
                HNode& n = nodes[i];
                if (!isLeaf(n)) continue;
                const int w = at[i];
                for (int j = 0; j < n.count; j++) outTris[w + j] = tris[sorted[0][n.startOrChild + j]];
                n.startOrChild = w;
            }
        });
    }
    // sorted distinct original-triangle ids of a leaf, written over the leaf's own (disjoint) range of `uniq`; returns how many
    int uniqueIds(const HNode& leaf, int* uniq) const
    {
        int* ids = uniq + leaf.startOrChild;
        for (int i = 0; i < leaf.count; i++) ids[i] = origTri[sorted[0][leaf.startOrChild + i]];
        std::sort(ids, ids + leaf.count);
        return (int)(std::unique(ids, ids + leaf.count) - ids);
    }
    void unindexPreSplit() // PreSplitting.GetUnindexedTriangles, PreSplitting.cs:169-273
    {
        const int pairs = ((int)nodes.size() - 2) / 2;
        std::vector<int, BigAlloc<int, true>> uniq(frag.size()); BigVec<int> ucount(nodes.size(), 0), at((size_t)pairs + 1, 0);
        const bool twins = pairs == 1 && isLeaf(nodes[2]) && isLeaf(nodes[3]) && nodes[2].startOrChild == nodes[3].startOrChild;   // single-leaf root: both leaves are the same range
        std::vector<int> twinIds;
        // pass 1: distinct ids per leaf and how far the pair advances the output cursor (PreSplitting.cs:200-262)
        parallelFor(pairs, [&](int b, int e) {
            for (int p = b; p < e; p++) {
                const int top = 2 + 2 * p;
                const HNode& l = nodes[top]; const HNode& r = nodes[top + 1];
                int adv = 0;
                if (isLeaf(l) && isLeaf(r)) {
                    const int nl = ucount[top] = uniqueIds(l, uniq.data());
                    if (twins) twinIds.assign(uniq.begin() + l.startOrChild, uniq.begin() + l.startOrChild + nl);
                    const int nr = ucount[top + 1] = twins ? nl : uniqueIds(r, uniq.data());
                    const int* lu = uniq.data() + l.startOrChild; const int* ru = twins ? twinIds.data() : uniq.data() + r.startOrChild;
                    int shared = 0;
                    for (int i = 0; i < nl; i++) if (std::binary_search(ru, ru + nr, lu[i])) shared++;
                    adv = (nl - shared) + nr;                       // l = [g, g + nl), r = [g + onlyLeft, g + onlyLeft + nr): they overlap in the shared ids
                } else if (isLeaf(l) || isLeaf(r)) {
                    const int leaf = isLeaf(l) ? top : top + 1;
                    adv = ucount[leaf] = uniqueIds(nodes[leaf], uniq.data());
                }
                at[p + 1] = adv;
            }
        });
        for (int p = 0; p < pairs; p++) at[p + 1] += at[p];
        outTris.assign(frag.size(), GpuBlasTriangle{});
        // pass 2: every pair writes its own output range
        parallelFor(pairs, [&](int b, int e) {
            for (int p = b; p < e; p++) {
                const int top = 2 + 2 * p, g = at[p];
                HNode& l = nodes[top]; HNode& r = nodes[top + 1];
                if (isLeaf(l) && isLeaf(r)) {
                    const int nl = ucount[top], nr = ucount[top + 1];
                    const int* lu = uniq.data() + l.startOrChild; const int* ru = twins ? twinIds.data() : uniq.data() + r.startOrChild;
                    int onlyLeft = 0, back = 0;
                    for (int i = 0; i < nl; i++) { const int id = lu[i]; if (std::binary_search(ru, ru + nr, id)) outTris[g + nl - back++ - 1] = tris[id]; else outTris[g + onlyLeft++] = tris[id]; }
                    int onlyRight = 0;
                    for (int i = 0; i < nr; i++) if (!std::binary_search(lu, lu + nl, ru[i])) outTris[g + nl + onlyRight++] = tris[ru[i]];
                    l.startOrChild = g; l.count = nl;
                    r.startOrChild = g + onlyLeft; r.count = nr;
                } else if (isLeaf(l) || isLeaf(r)) {
                    HNode& leaf = isLeaf(l) ? l : r;
                    const int n = ucount[isLeaf(l) ? top : top + 1];
                    const int* lu = uniq.data() + leaf.startOrChild;
                    for (int i = 0; i < n; i++) outTris[g + i] = tris[lu[i]];
                    leaf.startOrChild = g; leaf.count = n;
                }
            }
        });
        outTris.resize((size_t)at[pairs]);
    }

    // The build in three steps, so that a host can run the middle one elsewhere (libidkpt's idkptBuildBlasCore on the GPU):
    //   begin  — fragments (PreSplit or one box per triangle);
    //   core   — BLAS.GetBuildData + the SweepSAH recursion: the node array (ids reserved per subtree, no compaction yet) and the final
    //            order of the x-sorted id array, which is all the later steps read;
    //   finish — single-leaf root, OptimizeStackSize, RemoveEmptySubtrees, GetUnindexedTriangles, parent / leaf indices, SAH.
    bool refit = false; bool timing = false;
    std::chrono::steady_clock::time_point t0, tp; struct rusage ru0;
    void lap(const char* what)
    {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        struct rusage ru; getrusage(RUSAGE_SELF, &ru);
        auto tv = [](const timeval& a) { return a.tv_sec * 1e3 + a.tv_usec * 1e-3; };
        fprintf(stderr, "[idkbvh] %-18s %8.2f ms   cpu user %8.2f sys %8.2f ms   minor faults %ld\n", what, std::chrono::duration<double, std::milli>(t - tp).count(),
                tv(ru.ru_utime) - tv(ru0.ru_utime), tv(ru.ru_stime) - tv(ru0.ru_stime), ru.ru_minflt - ru0.ru_minflt);
        tp = t; ru0 = ru;
    }
    void begin(bool refittable, float factor, int threads)
    {
        t0 = tp = std::chrono::steady_clock::now();
        timing = g_phaseTiming.load(std::memory_order_relaxed) != 0;   // idkbvhSetPhaseTiming: phase times on stderr
        getrusage(RUSAGE_SELF, &ru0);
        maxThreads = threads <= 0 ? defaultThreadCount() : threads;
        refit = refittable;
        // non-finite positions: the reference's builder has no defined result for them (NaN boxes, int conversions of NaN); refused here
        parallelFor(triCount, [&](int b, int e) { for (int i = b; i < e; i++) { __m128 p[3]; triPoints(i, p[0], p[1], p[2]); for (int v = 0; v < 3; v++) { float f[4]; _mm_storeu_ps(f, p[v]); if (!std::isfinite(f[0]) || !std::isfinite(f[1]) || !std::isfinite(f[2])) error = 5; } } });
        if (error) return;
        if (!refit) { preSplit(factor); if (error) return; }
        else { frag.resize(triCount); for (int i = 0; i < triCount; i++) frag[i] = triBox(i); }
        lap("presplit");
    }
    void coreCpu()
    {
        const int n = (int)frag.size();
        nodes.assign((size_t)std::max(2 * n, 4), HNode{});
        leftTable.assign(n, 0); rightCosts.assign(3 * (size_t)n, 0.0f);   // one right-cost / scratch row per axis
        // BLAS.GetBuildData (BLAS.cs:128-157): ids sorted by FloatToKey(min+max) per axis
        {
            auto sortAxis = [&](int axis) {
                std::vector<uint32_t> keys(n);
                for (int i = 0; i < n; i++) keys[i] = floatToKey(frag[i].lo(axis) + frag[i].hi(axis));
                sorted[axis].resize(n);
                radixSortIds(keys.data(), n, sorted[axis].data());
            };
            if (n >= kThreadedSortingThreshold && maxThreads > 1) { std::thread a(sortAxis, 0), b(sortAxis, 1); sortAxis(2); a.join(); b.join(); }
            else for (int a = 0; a < 3; a++) sortAxis(a);
        }
        lap("sort");
        nodes[1].startOrChild = 0; nodes[1].count = n;
        buildAll();
        lap("build");
    }
    void coreSet(const HNode* coreNodes, const int32_t* sorted0)
    {
        const int n = (int)frag.size();
        nodes.assign(coreNodes, coreNodes + (size_t)std::max(2 * n, 4));
        sorted[0].assign(sorted0, sorted0 + n);
        lap("core (external)");
    }
    void finish()
    {
        if (isLeaf(nodes[1])) { nodes[2] = nodes[1]; nodes[3] = nodes[1]; nodes[1].startOrChild = 2; nodes[1].count = 0; } // BLAS.cs:173-183
        optimizeStackSize();
        lap("stack-opt");
        nodes.resize((size_t)compactNodes());
        lap("compact");
        if (!refit) unindexPreSplit(); else unindexPlain();
        lap("unindex");
        if (refit) {
            const int nn = (int)nodes.size();
            parents.assign(nn, -1);
            for (int i = 1; i < nn; i++) if (!isLeaf(nodes[i])) { parents[nodes[i].startOrChild] = i; parents[nodes[i].startOrChild + 1] = i; }
            for (int i = 2; i < nn; i++) if (isLeaf(nodes[i])) leaves.push_back(i);
        }
        sah = globalSAH();
        lap("parents+sah");
        buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void run(bool refittable, float factor, int threads) { begin(refittable, factor, threads); if (error) return; coreCpu(); finish(); }
};

uint32_t spread3(uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; }

} // namespace

struct idkbvh_blas { Builder b; int fragmentCount = 0; };

extern "C" {

int32_t idkbvhBuildBlas(const float* positions, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable, float preSplitFactor, int32_t threads, idkbvh_blas** out)
{
    if (!positions || !tris || triCount <= 0 || !out) return 2;
    idkbvh_blas* h = new idkbvh_blas();
    h->b.positions = positions; h->b.tris = tris; h->b.triCount = triCount;
    h->b.run(isRefittable != 0, preSplitFactor, threads);
    if (h->b.error) { const int32_t rc = h->b.error; delete h; return rc; }
    h->fragmentCount = (int)h->b.frag.size();
    h->b.positions = nullptr; h->b.tris = nullptr; // inputs are only borrowed during the call
    *out = h;
    return 0;
}
int32_t idkbvhBlasBegin(const float* positions, const GpuBlasTriangle* tris, int32_t triCount, int32_t isRefittable, float preSplitFactor, int32_t threads, idkbvh_blas** out)
{
    if (!positions || !tris || triCount <= 0 || !out) return 2;
    idkbvh_blas* h = new idkbvh_blas();
    h->b.positions = positions; h->b.tris = tris; h->b.triCount = triCount;
    h->b.begin(isRefittable != 0, preSplitFactor, threads);
    if (h->b.error) { const int32_t rc = h->b.error; delete h; return rc; }
    h->fragmentCount = (int)h->b.frag.size();
    h->b.positions = nullptr; h->b.tris = nullptr;
    *out = h;
    return 0;
}
int32_t idkbvhBlasFragments(const idkbvh_blas* h, const float** boxes, int32_t* count)
{
    if (!h || !boxes || !count) return 2;
    *boxes = reinterpret_cast<const float*>(h->b.frag.data()); *count = (int32_t)h->b.frag.size();
    return 0;
}
int32_t idkbvhBlasCoreCpu(idkbvh_blas* h) { if (!h || h->b.frag.empty()) return 2; h->b.coreCpu(); return 0; }
int32_t idkbvhBlasCoreGet(const idkbvh_blas* h, GpuBlasNode* nodes, int32_t* sorted0)
{
    if (!h || h->b.nodes.empty() || h->b.sorted[0].empty()) return 2;
    if (nodes) memcpy(nodes, h->b.nodes.data(), h->b.nodes.size() * sizeof(HNode));
    if (sorted0) memcpy(sorted0, h->b.sorted[0].data(), h->b.sorted[0].size() * 4);
    return 0;
}
int32_t idkbvhBlasCoreSet(idkbvh_blas* h, const GpuBlasNode* nodes, const int32_t* sorted0)
{
    if (!h || h->b.frag.empty() || !nodes || !sorted0) return 2;
    h->b.coreSet(reinterpret_cast<const HNode*>(nodes), sorted0);
    return 0;
}
int32_t idkbvhBlasCoreBuffers(idkbvh_blas* h, GpuBlasNode** nodes, int32_t** sorted0)
{
    if (!h || h->b.frag.empty() || !nodes || !sorted0) return 2;
    const size_t n = h->b.frag.size();
    h->b.nodes.resize(std::max<size_t>(2 * n, 4)); h->b.sorted[0].resize(n);      // written in full by the external core
    *nodes = reinterpret_cast<GpuBlasNode*>(h->b.nodes.data()); *sorted0 = h->b.sorted[0].data();
    return 0;
}
int32_t idkbvhBlasFinish(idkbvh_blas* h, const float* positions, const GpuBlasTriangle* tris)
{
    if (!h || !positions || !tris || h->b.nodes.empty() || h->b.sorted[0].empty()) return 2;
    h->b.positions = positions; h->b.tris = tris;
    h->b.lap("core");
    h->b.finish();
    h->b.positions = nullptr; h->b.tris = nullptr;
    return 0;
}
int32_t idkbvhBlasGetInfo(const idkbvh_blas* h, idkbvh_blas_info* o)
{
    if (!h || !o) return 2;
    o->NodeCount = (int)h->b.nodes.size(); o->TriangleCount = (int)h->b.outTris.size(); o->RequiredStackSize = h->b.requiredStack;
    o->ParentIndexCount = (int)h->b.parents.size(); o->LeafIndexCount = (int)h->b.leaves.size(); o->FragmentCount = h->fragmentCount; o->Sah = h->b.sah; o->BuildMs = h->b.buildMs;
    return 0;
}
int32_t idkbvhBlasCopy(const idkbvh_blas* h, GpuBlasNode* nodes, GpuBlasTriangle* triangles, int32_t* parents, int32_t* leaves)
{
    if (!h) return 2;
    if (nodes) memcpy(nodes, h->b.nodes.data(), h->b.nodes.size() * sizeof(HNode));
    if (triangles) memcpy(triangles, h->b.outTris.data(), h->b.outTris.size() * sizeof(GpuBlasTriangle));
    if (parents && !h->b.parents.empty()) memcpy(parents, h->b.parents.data(), h->b.parents.size() * 4);
    if (leaves && !h->b.leaves.empty()) memcpy(leaves, h->b.leaves.data(), h->b.leaves.size() * 4);
    return 0;
}
void idkbvhBlasFree(idkbvh_blas* h) { delete h; }
void idkbvhSetPhaseTiming(int32_t enabled) { g_phaseTiming.store(enabled ? 1 : 0, std::memory_order_relaxed); }

int32_t idkbvhInstanceWorldBounds(const GpuBlasNode* root, const GpuMeshTransform* xf, float out[6])
{
    if (!root || !xf || !out) return 2;
    SBox nb = SBox::empty();
    for (int i = 0; i < 8; i++) {
        float c[3] = {(i & 1) ? root->Max[0] : root->Min[0], (i & 2) ? root->Max[1] : root->Min[1], (i & 4) ? root->Max[2] : root->Min[2]};
        float w[3];
        for (int k = 0; k < 3; k++) w[k] = (c[0] * xf->Model[k][0]) + (c[1] * xf->Model[k][1]) + (c[2] * xf->Model[k][2]) + (1.0f * xf->Model[k][3]); // (Vector4(p,1) * M).Xyz
        nb.grow(load3(w));
    }
    float v[4]; _mm_storeu_ps(v, nb.mn); out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; _mm_storeu_ps(v, nb.mx); out[3] = v[0]; out[4] = v[1]; out[5] = v[2];
    return 0;
}

int32_t idkbvhBuildTlas(const float* leafBounds, int32_t count, int32_t searchRadius, GpuTlasNode* outNodes)
{
    if (!leafBounds || count <= 0 || !outNodes) return 2;
    struct TN { float mn[3]; uint32_t id; float mx[3]; float pad; };
    const int nodeCount = 2 * count - 1;
    TN* nodes = reinterpret_cast<TN*>(outNodes);
    memset(nodes, 0, sizeof(TN) * (size_t)nodeCount);
    std::vector<TN> temp(nodeCount); memset(temp.data(), 0, sizeof(TN) * (size_t)nodeCount);
    auto boxOf = [](const TN& n) { return SBox{load3(n.mn), load3(n.mx)}; };
    {   // leaves, Morton-sorted (stable radix, TLAS.cs:36-56)
        std::vector<TN> leaf(count); SBox global = SBox::empty();
        for (int i = 0; i < count; i++) { const float* b = leafBounds + 6 * (size_t)i; TN n = {{b[0], b[1], b[2]}, (1u << 31) | (uint32_t)i, {b[3], b[4], b[5]}, 0.0f}; leaf[i] = n; global.grow(boxOf(n)); }
        float gmn[4], gmx[4]; _mm_storeu_ps(gmn, global.mn); _mm_storeu_ps(gmx, global.mx);
        std::vector<uint32_t> keys(count);
        for (int i = 0; i < count; i++) {
            uint32_t q[3];
            for (int a = 0; a < 3; a++) {
                float c = (leaf[i].mx[a] + leaf[i].mn[a]) * 0.5f, ext = gmx[a] - gmn[a];
                float m = (c - gmn[a]) / ext * (1.0f - 0.0f) + 0.0f; if (ext == 0.0f) m = 0.0f;   // MyMath.MapToZeroOne
                q[a] = std::min(satUInt(m * 1024.0f), 1023u);
            }
            keys[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);                  // MyMath.GetMortonCode30
        }
        std::vector<int> order(count); radixSortIds(keys.data(), count, order.data());
        for (int i = 0; i < count; i++) nodes[nodeCount - count + i] = leaf[order[i]];
    }
    int activeCount = count, activeEnd = nodeCount;
    std::vector<int> pref(count);
    while (activeCount > 1) {
        const int activeStart = activeEnd - activeCount;
        for (int i = 0; i < activeCount; i++) {
            const int a = activeStart + i, s0 = std::max(a - searchRadius, activeStart), s1 = std::min(a + searchRadius + 1, activeEnd);
            float best = FLT_MAX; int bestIdx = -1; const SBox nb = boxOf(nodes[a]);
            for (int k = s0; k < s1; k++) { if (k == a) continue; SBox m = nb; m.grow(boxOf(nodes[k])); float area = m.halfArea(); if (area < best) { best = area; bestIdx = k; } }
            pref[i] = bestIdx - activeStart;
        }
        int merged = 0;
        for (int i = 0; i < activeCount; i++) { int b = pref[i]; if (pref[b] == i && i < b) merged += 2; }
        const int unmerged = activeCount - merged, fresh = merged / 2;
        int mergedHead = activeEnd - merged; const int newBegin = mergedHead - unmerged - fresh; int unmergedHead = newBegin;
        for (int i = 0; i < activeCount; i++) {
            const int b = pref[i], aId = i + activeStart;
            if (pref[b] == i) {
                if (i < b) {
                    temp[mergedHead] = nodes[aId]; temp[mergedHead + 1] = nodes[b + activeStart];
                    SBox m = boxOf(temp[mergedHead]); m.grow(boxOf(temp[mergedHead + 1]));
                    TN p; memset(&p, 0, sizeof(p)); float v[4]; _mm_storeu_ps(v, m.mn); p.mn[0] = v[0]; p.mn[1] = v[1]; p.mn[2] = v[2]; _mm_storeu_ps(v, m.mx); p.mx[0] = v[0]; p.mx[1] = v[1]; p.mx[2] = v[2];
                    p.id = (uint32_t)mergedHead;
                    temp[unmergedHead++] = p; mergedHead += 2;
                }
            } else temp[unmergedHead++] = nodes[aId];
        }
        memcpy(&nodes[newBegin], &temp[newBegin], sizeof(TN) * (size_t)(activeEnd - newBegin));
        activeCount -= merged / 2; activeEnd -= merged;
    }
    return 0;
}

int32_t idkbvhRefitBlas(GpuBlasNode* nodes_, int32_t nodeCount, const float* positions, const GpuBlasTriangle* tris)
{
    if (!nodes_ || nodeCount < 4 || !positions || !tris) return 2;
    HNode* nodes = reinterpret_cast<HNode*>(nodes_);
    for (int i = nodeCount - 1; i >= 1; i--) {
        HNode& p = nodes[i];
        if (isLeaf(p)) {
            SBox b = SBox::empty();
            for (int k = 0; k < p.count; k++) { const GpuBlasTriangle& t = tris[p.startOrChild + k]; b.grow(load3(positions + 3 * (size_t)t.X)); b.grow(load3(positions + 3 * (size_t)t.Y)); b.grow(load3(positions + 3 * (size_t)t.Z)); }
            setBounds(p, b);
        } else { SBox m = nodeBox(nodes[p.startOrChild]); m.grow(nodeBox(nodes[p.startOrChild + 1])); setBounds(p, m); }
    }
    return 0;
}

} // extern "C"
