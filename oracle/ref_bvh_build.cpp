// ORACLE (test infrastructure only; PARITY UNPINNED for this C# half, see ref_math.h header).
// Sequential restatement of the reference's CPU BVH builder.  All citations relative to
// /root/reference/IDKEngine/Source.  Arithmetic notes:
//   * Box min/max follow x86 minps/maxps (Vector128.MinNative/MaxNative, Shapes/Box.cs:40-50): r = a<b ? a : b.
//   * HalfArea = fma(x+y, z, x*y) (Utils/MyMath.cs:222-229, float.MultiplyAddEstimate on FMA hardware).
//   * C# never contracts a*b+c; compile with -ffp-contract=off.
//   * MathF.Cbrt / MathF.Round -> cbrtf / rintf (round-half-even).
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <vector>
#include <algorithm>
#include "../include/idkpt_types.h"

namespace {

struct Vec3 { float x, y, z; float operator[](int i) const { return (&x)[i]; } float& operator[](int i) { return (&x)[i]; } };
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator*(float t, Vec3 a) { return {t * a.x, t * a.y, t * a.z}; }
static inline float minN(float a, float b) { return a < b ? a : b; } // minps / float.MinNative
static inline float maxN(float a, float b) { return a > b ? a : b; } // maxps / float.MaxNative
static inline float HalfArea3(float sx, float sy, float sz) { return fmaf(sx + sy, sz, sx * sy); } // MyMath.cs:222-229

struct Box {                                              // Shapes/Box.cs
    Vec3 mn, mx;
    static Box Empty() { return {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; } // :206-212
    void Grow(Vec3 p) { mn = {minN(mn.x, p.x), minN(mn.y, p.y), minN(mn.z, p.z)}; mx = {maxN(mx.x, p.x), maxN(mx.y, p.y), maxN(mx.z, p.z)}; } // :40-44
    void Grow(const Box& b) { mn = {minN(mn.x, b.mn.x), minN(mn.y, b.mn.y), minN(mn.z, b.mn.z)}; mx = {maxN(mx.x, b.mx.x), maxN(mx.y, b.mx.y), maxN(mx.z, b.mx.z)}; } // :46-50
    void Clip(const Box& b) { mn = {maxN(mn.x, b.mn.x), maxN(mn.y, b.mn.y), maxN(mn.z, b.mn.z)}; mx = {minN(mx.x, b.mx.x), minN(mx.y, b.mx.y), minN(mx.z, b.mx.z)}; } // :66-70
    Vec3 Size() const { return mx - mn; }
    float HalfArea() const { Vec3 s = Size(); return HalfArea3(s.x, s.y, s.z); }       // :133-139
    float Area() const { return HalfArea() * 2.0f; }                                    // :126-129
    int LargestAxis() const { Vec3 s = Size(); int a = 0; if (s[0] < s[1]) a = 1; if (s[a] < s[2]) a = 2; return a; } // :108-115
    float LargestExtent() const { Vec3 s = Size(); return maxN(s[0], maxN(s[1], s[2])); } // :119-123
};

struct Tri { Vec3 p0, p1, p2; };
static inline Box BoxFromTri(const Tri& t) { Box b = {t.p0, t.p0}; b.Grow(t.p1); b.Grow(t.p2); return b; } // Box.cs:214-221
static inline Vec3 CrossOtk(Vec3 l, Vec3 r) { return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; } // OpenTK Vector3.Cross
static inline float TriArea(const Tri& t) { Vec3 c = CrossOtk(t.p1 - t.p0, t.p2 - t.p0); return sqrtf((c.x * c.x) + (c.y * c.y) + (c.z * c.z)) * 0.5f; } // Triangle.cs:10

// .NET (>=9) saturating float->int conversion, NaN -> 0
static inline int ToIntSat(float f) { if (f != f) return 0; if (f >= 2147483648.0f) return INT32_MAX; if (f <= -2147483648.0f) return INT32_MIN; return (int)f; }
static inline uint32_t ToUIntSat(float f) { if (f != f || f <= 0.0f) return 0u; if (f >= 4294967296.0f) return UINT32_MAX; return (uint32_t)f; }

static inline uint32_t FloatToKey(float v) // Utils/Algorithms.cs:15-34
{
    uint32_t f; memcpy(&f, &v, 4);
    uint32_t mask = (uint32_t)(((int32_t)f >> 31) | (int32_t)(1u << 31));
    return f ^ mask;
}

// Utils/Algorithms.cs:45-112: stable LSD radix sort, 3 passes x 11 bits; result ends up in `output`.
template <class T, class KeyFn>
static void RadixSort(std::vector<T>& input, std::vector<T>& output, KeyFn key)
{
    const int radixSize = 11, binSize = 1 << radixSize, mask = binSize - 1, passes = 3;
    std::vector<int> prefix(binSize * passes, 0);
    size_t n = input.size();
    for (size_t i = 0; i < n; i++) { uint32_t k = key(input[i]); for (int p = 0; p < passes; p++) prefix[((k >> (p * radixSize)) & mask) + p * binSize]++; }
    for (int p = 0; p < passes; p++) { int sum = 0; for (int i = 0; i < binSize; i++) { int t = prefix[i + p * binSize]; prefix[i + p * binSize] = sum; sum += t; } }
    T* in = input.data(); T* out = output.data();
    for (int p = 0; p < passes; p++) {
        for (size_t j = 0; j < n; j++) { T el = in[j]; uint32_t k = key(el); out[prefix[((k >> (p * radixSize)) & mask) + p * binSize]++] = el; }
        std::swap(in, out);
    }
    // after 3 passes the data sits in the buffer `in` points to, which is output.data()
}

struct Geometry {                     // Bvh/BLAS.cs:71-97
    const float* positions;           // packed float3, global vertex array
    const GpuBlasTriangle* tris;      // this BLAS' slice (global vertex ids)
    int triCount;
    Tri GetTri(int i) const { return GetTri(tris[i]); }
    Tri GetTri(const GpuBlasTriangle& t) const {
        const float* a = positions + 3 * (size_t)t.X; const float* b = positions + 3 * (size_t)t.Y; const float* c = positions + 3 * (size_t)t.Z;
        return {{a[0], a[1], a[2]}, {b[0], b[1], b[2]}, {c[0], c[1], c[2]}};
    }
};

struct BuildSettings {                // Bvh/BLAS.cs:31-48
    int StopSplittingThreshold = 1;
    int MaxLeafTriangleCount = 2;
    float TriangleCost = 1.1f;
    int StackOptThreshold = 16;
    float StackOptSahIncreaseAcceptance = 0.0009745f;
    float StackOptMaxLeafTriangleCount = (float)INT32_MAX;
};
const float TRAVERSAL_COST = 1.0f;    // :26

struct Node { Vec3 mn; int32_t TriStartOrChild; Vec3 mx; int32_t TriCount; bool IsLeaf() const { return TriCount > 0; } int TriEnd() const { return TriStartOrChild + TriCount; }
    float HalfArea() const { Vec3 s = mx - mn; return HalfArea3(s.x, s.y, s.z); } void SetBounds(const Box& b) { mn = b.mn; mx = b.mx; } }; // GpuTypes/GpuBlasNode.cs
static_assert(sizeof(Node) == 32, "node");

struct BuildData {                    // Bvh/BLAS.cs:106-119
    std::vector<float> RightCostsAccum;
    std::vector<uint8_t> FragLeftTable;
    std::vector<Box> Bounds;
    std::vector<int> OriginalTriIds;
    std::vector<int> Sorted[3];
};

struct Blas {
    std::vector<Node> nodes;
    int RequiredStackSize = 0;
    std::vector<GpuBlasTriangle> triangles;
    std::vector<int> parents, leaves;
    int fragmentCount = 0;
    double sah = 0.0;
};

// ---------------- PreSplitting.PreSplit (Bvh/PreSplitting.cs:26-160) ----------------
static float Priority(const Tri& t) // :124-135
{
    Box b = BoxFromTri(t);
    float extentPrio = b.LargestExtent() * b.LargestExtent();
    float emptyAreaPrio = b.Area() - TriArea(t);
    return cbrtf(extentPrio * emptyAreaPrio);
}
static int GetSplitCount(float priority, float totalPriority, int triangleCount, float splitFactor) // :116-122
{
    float shareOfTris = priority / totalPriority * (float)triangleCount;
    return 1 + ToIntSat(shareOfTris * splitFactor);
}
static float GetNodeSize(float extent, float globalSize) // :137-159
{
    float alpha = extent / globalSize;
    uint32_t bits; memcpy(&bits, &alpha, 4); bits &= (255u << 23);
    float p2; memcpy(&p2, &bits, 4);
    return p2 * globalSize;
}
static void TriSplit(const Tri& t, int axis, float position, Box* l, Box* r) // Shapes/Triangle.cs:48-97
{
    Box lBox = Box::Empty(), rBox = Box::Empty();
    bool q0 = t.p0[axis] <= position, q1 = t.p1[axis] <= position, q2 = t.p2[axis] <= position;
    if (q0) lBox.Grow(t.p0); else rBox.Grow(t.p0);
    if (q1) lBox.Grow(t.p1); else rBox.Grow(t.p1);
    if (q2) lBox.Grow(t.p2); else rBox.Grow(t.p2);
    auto SplitEdge = [&](Vec3 a, Vec3 b) { float tt = (position - a[axis]) / (b[axis] - a[axis]); return a + tt * (b - a); };
    if (q0 ^ q1) { Vec3 m = SplitEdge(t.p0, t.p1); lBox.Grow(m); rBox.Grow(m); }
    if (q1 ^ q2) { Vec3 m = SplitEdge(t.p1, t.p2); lBox.Grow(m); rBox.Grow(m); }
    if (q2 ^ q0) { Vec3 m = SplitEdge(t.p2, t.p0); lBox.Grow(m); rBox.Grow(m); }
    *l = lBox; *r = rBox;
}
static void PreSplit(const Geometry& g, float splitFactor, std::vector<Box>& bounds, std::vector<int>& orig)
{
    float totalPriority = 0.0f;
    for (int i = 0; i < g.triCount; i++) totalPriority += Priority(g.GetTri(i));
    int counter = 0;
    for (int i = 0; i < g.triCount; i++) counter += GetSplitCount(Priority(g.GetTri(i)), totalPriority, g.triCount, splitFactor);
    bounds.resize(counter); orig.resize(counter);
    counter = 0;
    Box globalBox = Box::Empty();                        // BLAS.ComputeBoundingBox(0, n, geometry) (BLAS.cs:704-715)
    for (int i = 0; i < g.triCount; i++) { Tri t = g.GetTri(i); globalBox.Grow(t.p0); globalBox.Grow(t.p1); globalBox.Grow(t.p2); }
    Vec3 globalSize = globalBox.Size();
    struct Item { Box box; int splits; };
    Item stack[64];
    for (int i = 0; i < g.triCount; i++) {
        Tri tri = g.GetTri(i);
        int splitCount = GetSplitCount(Priority(tri), totalPriority, g.triCount, splitFactor);
        int sp = 0;
        stack[sp++] = {BoxFromTri(tri), splitCount};
        while (sp > 0) {
            Item it = stack[--sp];
            if (it.splits == 1) { bounds[counter] = it.box; orig[counter] = i; counter++; continue; }
            int axis = it.box.LargestAxis();
            float largestExtent = it.box.LargestExtent();
            float nodeSize = GetNodeSize(largestExtent, globalSize[axis]);
            if (nodeSize >= largestExtent - 0.0001f) nodeSize *= 0.5f;
            float midPos = (it.box.mn[axis] + it.box.mx[axis]) * 0.5f;
            float index = rintf((midPos - globalBox.mn[axis]) / nodeSize);
            float splitPos = globalBox.mn[axis] + index * nodeSize;
            Box lBox, rBox; TriSplit(tri, axis, splitPos, &lBox, &rBox);
            lBox.Clip(it.box); rBox.Clip(it.box);
            float le = lBox.LargestExtent(), re = rBox.LargestExtent();
            int leftCount = ToIntSat((float)it.splits * (le / (le + re)));
            leftCount = std::min(std::max(leftCount, 1), it.splits - 1);
            int rightCount = it.splits - leftCount;
            stack[sp++] = {rBox, rightCount};
            stack[sp++] = {lBox, leftCount};
        }
    }
}

// ---------------- BLAS build (Bvh/BLAS.cs) ----------------
static void GetBuildData(BuildData& bd) // :128-157, key :950-955
{
    int n = (int)bd.Bounds.size();
    bd.FragLeftTable.assign(n, 0);
    bd.RightCostsAccum.assign(n, 0.0f);
    for (int axis = 0; axis < 3; axis++) {
        std::vector<int> input(n); bd.Sorted[axis].assign(n, 0);
        for (int i = 0; i < n; i++) input[i] = i;
        RadixSort(input, bd.Sorted[axis], [&](int idx) { float p = bd.Bounds[idx].mn[axis] + bd.Bounds[idx].mx[axis]; return FloatToKey(p); });
    }
}
static Box ComputeBoundingBox(int start, int count, const BuildData& bd, int axis = 0) // :717-728
{
    Box b = Box::Empty();
    const int* ids = bd.Sorted[axis].data() + start;
    for (int i = 0; i < count; i++) b.Grow(bd.Bounds[ids[i]]);
    return b;
}
static int StablePartition(int* source, int n, int* aux, const uint8_t* table) // Utils/Algorithms.cs:282-303
{
    int l = 0, r = 0;
    for (int i = 0; i < n; i++) { int id = source[i]; if (table[id]) source[l++] = id; else aux[r++] = id; }
    memcpy(source + l, aux, sizeof(int) * (size_t)r);
    return l;
}
struct ObjectSplit { int Axis; int SplitIndex; float NewCost; };
static bool TrySplit(const Node& parent, BuildData& bd, const BuildSettings& s, ObjectSplit* out) // :730-873
{
    Box parentBox = {parent.mn, parent.mx};
    if (parent.TriCount <= s.StopSplittingThreshold) return false;
    int start = parent.TriStartOrChild, end = parent.TriEnd();
    ObjectSplit best = {0, 0, FLT_MAX};
    float* rightCostsAccum = bd.RightCostsAccum.data();
    const Box* fragBounds = bd.Bounds.data();
    for (int axis = 0; axis < 3; axis++) {
        const int* ids = bd.Sorted[axis].data();
        int firstRight = start + 1;
        Box rightBoxAccum = Box::Empty();
        float rightCounter = 0.0f;
        for (int i = end - 1; i >= firstRight; i--) {
            rightCounter++;
            rightBoxAccum.Grow(fragBounds[ids[i]]);
            float rightCost = rightBoxAccum.HalfArea() * rightCounter;
            rightCostsAccum[i] = rightCost;
            if (rightCost >= best.NewCost) { firstRight = i + 1; break; }
        }
        Box leftBoxAccum = Box::Empty();
        float leftCounter = (float)(firstRight - start) - 1.0f;
        for (int i = start; i < firstRight - 1; i++) leftBoxAccum.Grow(fragBounds[ids[i]]);
        for (int i = firstRight - 1; i < end - 1; i++) {
            int splitIndex = i + 1;
            leftCounter++;
            leftBoxAccum.Grow(fragBounds[ids[i]]);
            float leftCost = leftBoxAccum.HalfArea() * leftCounter;
            float rightCost = rightCostsAccum[splitIndex];
            float cost = leftCost + rightCost;
            if (cost < best.NewCost) { best.SplitIndex = splitIndex; best.Axis = axis; best.NewCost = cost; }
            else if (leftCost >= best.NewCost) break;
        }
    }
    if (parent.TriCount <= s.MaxLeafTriangleCount) {
        float notSplitCost = s.TriangleCost * (float)parent.TriCount;
        best.NewCost = TRAVERSAL_COST + (s.TriangleCost * best.NewCost / parentBox.HalfArea());
        if (best.NewCost >= notSplitCost) return false;
    }
    Box leftBox = ComputeBoundingBox(start, best.SplitIndex - start, bd, best.Axis);
    Box rightBox = ComputeBoundingBox(best.SplitIndex, end - best.SplitIndex, bd, best.Axis);
    bool leftSmaller = leftBox.HalfArea() < rightBox.HalfArea();
    bool swapSides = leftSmaller; // larger child goes left (:822-824)
    int* ids = bd.Sorted[best.Axis].data();
    for (int i = start; i < best.SplitIndex; i++) bd.FragLeftTable[ids[i]] = !swapSides;
    for (int i = best.SplitIndex; i < end; i++) bd.FragLeftTable[ids[i]] = swapSides;
    int* aux = reinterpret_cast<int*>(rightCostsAccum + start); // Helper.ReUseMemory (:836)
    if (swapSides) best.SplitIndex = start + StablePartition(ids + start, parent.TriCount, aux, bd.FragLeftTable.data());
    StablePartition(bd.Sorted[(best.Axis + 1) % 3].data() + start, parent.TriCount, aux, bd.FragLeftTable.data());
    StablePartition(bd.Sorted[(best.Axis + 2) % 3].data() + start, parent.TriCount, aux, bd.FragLeftTable.data());
    *out = best;
    return true;
}
static void ProcessBuildTask(Blas& blas, BuildData& bd, const BuildSettings& s, int parentNodeId, int newNodesId) // :197-243
{
    Node& p0 = blas.nodes[parentNodeId];
    p0.SetBounds(ComputeBoundingBox(p0.TriStartOrChild, p0.TriCount, bd));
    ObjectSplit split;
    if (TrySplit(p0, bd, s, &split)) {
        Node& parent = blas.nodes[parentNodeId];
        Node left = {}; left.TriStartOrChild = parent.TriStartOrChild; left.TriCount = split.SplitIndex - left.TriStartOrChild;
        Node right = {}; right.TriStartOrChild = split.SplitIndex; right.TriCount = parent.TriCount - left.TriCount;
        int leftId = newNodesId, rightId = leftId + 1;
        blas.nodes[leftId] = left; blas.nodes[rightId] = right;
        parent.TriStartOrChild = leftId; parent.TriCount = 0;
        ProcessBuildTask(blas, bd, s, leftId, rightId + 1);
        ProcessBuildTask(blas, bd, s, rightId, rightId + (2 * left.TriCount - 1));
    }
}
static int ComputeRequiredStackSize(const Blas& blas, int nodeId = 2) // :672-702
{
    const Node& l = blas.nodes[nodeId]; const Node& r = blas.nodes[nodeId + 1];
    bool tl = !l.IsLeaf(), tr = !r.IsLeaf();
    if (tl || tr) {
        if (tl && tr) return std::max(ComputeRequiredStackSize(blas, l.TriStartOrChild), ComputeRequiredStackSize(blas, r.TriStartOrChild)) + 1;
        return ComputeRequiredStackSize(blas, tl ? l.TriStartOrChild : r.TriStartOrChild);
    }
    return 0;
}
static double ComputeGlobalSAH(const Blas& blas, const BuildSettings& s) // :629-657
{
    double cost = 0.0;
    double rootArea = 1.0 / (double)blas.nodes[1].HalfArea();
    std::vector<int> stack; stack.push_back(1);
    while (!stack.empty()) {
        const Node& n = blas.nodes[stack.back()]; stack.pop_back();
        double prob = (double)n.HalfArea() * rootArea;
        if (n.IsLeaf()) cost += (double)(s.TriangleCost * (float)n.TriCount) * prob;
        else { cost += (double)TRAVERSAL_COST * prob; stack.push_back(n.TriStartOrChild + 1); stack.push_back(n.TriStartOrChild); }
    }
    return cost;
}
static void CollapseDeepestLevel(Blas& blas, const BuildSettings& s, int newStackSize, bool firstPass, double& nextCollapseCost, int parentId = 1, int stackSize = 0) // :897-936
{
    Node& parent = blas.nodes[parentId];
    const Node& left = blas.nodes[parent.TriStartOrChild];
    const Node& right = blas.nodes[parent.TriStartOrChild + 1];
    int childBase = parent.TriStartOrChild;
    if (!left.IsLeaf()) CollapseDeepestLevel(blas, s, newStackSize, firstPass, nextCollapseCost, childBase + 0, stackSize + 1);
    if (!right.IsLeaf()) CollapseDeepestLevel(blas, s, newStackSize, firstPass, nextCollapseCost, childBase + 1, stackSize + 1);
    if (left.IsLeaf() && right.IsLeaf()) {
        if (stackSize > newStackSize && !firstPass) {
            parent.TriStartOrChild = left.TriStartOrChild;
            parent.TriCount = left.TriCount + right.TriCount;
        }
        if ((stackSize == newStackSize && !firstPass) || (stackSize > newStackSize && firstPass)) {
            if ((float)(left.TriCount + right.TriCount) > s.StackOptMaxLeafTriangleCount) { nextCollapseCost = (double)FLT_MAX; return; }
            double leavesCost = (double)s.TriangleCost * ((double)left.TriCount * (double)left.HalfArea() + (double)right.TriCount * (double)right.HalfArea());
            double newParentLeafCost = (double)s.TriangleCost * (double)(left.TriCount + right.TriCount);
            nextCollapseCost += ((double)parent.HalfArea() * (newParentLeafCost - (double)TRAVERSAL_COST) - leavesCost) / (double)blas.nodes[1].HalfArea();
        }
    }
}
static void OptimizeStackSize(Blas& blas, const BuildSettings& s) // :875-895
{
    blas.RequiredStackSize = ComputeRequiredStackSize(blas);
    if (blas.RequiredStackSize < s.StackOptThreshold) return;
    double currentCost = ComputeGlobalSAH(blas, s);
    double addedCost = 0.0;
    CollapseDeepestLevel(blas, s, blas.RequiredStackSize - 1, true, addedCost);
    double increasePercent = addedCost / currentCost;
    while (increasePercent <= (double)s.StackOptSahIncreaseAcceptance && blas.RequiredStackSize > 0) {
        CollapseDeepestLevel(blas, s, --blas.RequiredStackSize, false, addedCost);
        increasePercent = addedCost / currentCost;
    }
}
static int RemoveEmptySubtrees(Blas& blas) // :245-273
{
    int nodeCounter = 2;
    std::vector<int> stack; stack.push_back(1);
    while (!stack.empty()) {
        int pid = stack.back(); stack.pop_back();
        Node& parent = blas.nodes[pid];
        Node left = blas.nodes[parent.TriStartOrChild];
        Node right = blas.nodes[parent.TriStartOrChild + 1];
        int leftId = nodeCounter, rightId = nodeCounter + 1;
        blas.nodes[leftId] = left; blas.nodes[rightId] = right;
        parent.TriStartOrChild = leftId;
        nodeCounter += 2;
        if (!right.IsLeaf()) stack.push_back(rightId);
        if (!left.IsLeaf()) stack.push_back(leftId);
    }
    return nodeCounter;
}
static int Build(Blas& blas, BuildData& bd, const BuildSettings& s) // :159-195
{
    blas.nodes[0] = Node{};
    Node& root = blas.nodes[1];
    root = Node{}; root.TriStartOrChild = 0; root.TriCount = (int)bd.Bounds.size();
    ProcessBuildTask(blas, bd, s, 1, 2);
    if (blas.nodes[1].IsLeaf()) {
        blas.nodes[2] = blas.nodes[1]; blas.nodes[3] = blas.nodes[1];
        blas.nodes[1].TriStartOrChild = 2; blas.nodes[1].TriCount = 0;
    }
    OptimizeStackSize(blas, s);
    return RemoveEmptySubtrees(blas);
}
static void GetUnindexedTriangles(Blas& blas, const BuildData& bd, const Geometry& g) // BLAS.cs:441-466
{
    // The reference sizes this array by the fragment count and throws for a single-leaf root, whose leaf is duplicated
    // into nodes 2 and 3 (BLAS.cs:173-183 "TODO ... causes a crash in GetUnindexedTriangles").  Defined behaviour here:
    // size by the sum of leaf counts, i.e. the duplicated leaf's triangles are stored twice.
    size_t total = 0;
    for (size_t i = 2; i < blas.nodes.size(); i++) if (blas.nodes[i].IsLeaf()) total += blas.nodes[i].TriCount;
    blas.triangles.assign(std::max(total, bd.Bounds.size()), GpuBlasTriangle{});
    int triCounter = 0;
    for (size_t i = 2; i < blas.nodes.size(); i++) {
        Node& n = blas.nodes[i];
        if (n.IsLeaf()) {
            for (int j = 0; j < n.TriCount; j++) blas.triangles[triCounter + j] = g.tris[bd.Sorted[0][n.TriStartOrChild + j]];
            n.TriStartOrChild = triCounter; triCounter += n.TriCount;
        }
    }
    blas.triangles.resize(triCounter);
}
static std::vector<int> GetUniqueTriIds(const Node& leaf, const BuildData& bd) // PreSplitting.cs:251-272
{
    std::vector<int> ids(leaf.TriCount);
    for (int i = 0; i < leaf.TriCount; i++) ids[i] = bd.OriginalTriIds[bd.Sorted[0][leaf.TriStartOrChild + i]];
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    return ids;
}
static void GetUnindexedTrianglesPreSplit(Blas& blas, const BuildData& bd, const Geometry& g) // PreSplitting.cs:169-249
{
    blas.triangles.assign(bd.Bounds.size(), GpuBlasTriangle{});
    int global = 0;
    std::vector<int> stack; stack.push_back(2);
    while (!stack.empty()) {
        int top = stack.back(); stack.pop_back();
        Node& l = blas.nodes[top]; Node& r = blas.nodes[top + 1];
        if (l.IsLeaf() && r.IsLeaf()) {
            std::vector<int> lu = GetUniqueTriIds(l, bd), ru = GetUniqueTriIds(r, bd);
            auto contains = [](const std::vector<int>& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
            int onlyLeft = 0, backwards = 0;
            for (size_t i = 0; i < lu.size(); i++) {
                int id = lu[i];
                if (contains(ru, id)) blas.triangles[global + (int)lu.size() - backwards++ - 1] = g.tris[id];
                else blas.triangles[global + onlyLeft++] = g.tris[id];
            }
            int onlyRight = 0;
            for (size_t i = 0; i < ru.size(); i++) { int id = ru[i]; if (!contains(lu, id)) blas.triangles[global + (int)lu.size() + onlyRight++] = g.tris[id]; }
            l.TriStartOrChild = global; l.TriCount = (int)lu.size();
            r.TriStartOrChild = global + onlyLeft; r.TriCount = (int)ru.size();
            global += r.TriEnd() - l.TriStartOrChild;
        } else if (l.IsLeaf() || r.IsLeaf()) {
            Node& leaf = l.IsLeaf() ? l : r;
            std::vector<int> u = GetUniqueTriIds(leaf, bd);
            for (size_t i = 0; i < u.size(); i++) blas.triangles[global + i] = g.tris[u[i]];
            leaf.TriStartOrChild = global; leaf.TriCount = (int)u.size(); global += (int)u.size();
        }
        if (!r.IsLeaf()) stack.push_back(r.TriStartOrChild);
        if (!l.IsLeaf()) stack.push_back(l.TriStartOrChild);
    }
    blas.triangles.resize(global);
}

} // namespace

extern "C" {

// BVH.BlasesBuild body for ONE blas (Bvh/BVH.cs:315-375).  tris = this BLAS' slice of BVH.BlasTriangles as produced
// by BVH.Add (global vertex ids + MeshId).  Returns an opaque handle.
void* ref_blas_build(const float* positions, const GpuBlasTriangle* tris, int triCount, int isRefittable, float preSplitFactor)
{
    Geometry g = {positions, tris, triCount};
    BuildSettings s;
    Blas* blas = new Blas();
    BuildData bd;
    bool doPresplit = !isRefittable; // BVH.cs:325
    if (doPresplit) PreSplit(g, preSplitFactor, bd.Bounds, bd.OriginalTriIds);
    else { bd.Bounds.resize(triCount); for (int i = 0; i < triCount; i++) bd.Bounds[i] = BoxFromTri(g.GetTri(i)); } // BLAS.GetTriangleBounds :468-479
    blas->fragmentCount = (int)bd.Bounds.size();
    blas->nodes.assign(std::max(2 * (int)bd.Bounds.size(), 4), Node{}); // GetUpperBoundNodes :531-534
    GetBuildData(bd);
    int used = Build(*blas, bd, s);
    blas->nodes.resize(used);
    if (doPresplit) GetUnindexedTrianglesPreSplit(*blas, bd, g); else GetUnindexedTriangles(*blas, bd, g);
    if (isRefittable) {
        int n = (int)blas->nodes.size();
        blas->parents.assign(n, -1);                               // GetParentIndices :481-498
        for (int i = 1; i < n; i++) { const Node& nd = blas->nodes[i]; if (!nd.IsLeaf()) { blas->parents[nd.TriStartOrChild] = i; blas->parents[nd.TriStartOrChild + 1] = i; } }
        for (int i = 2; i < n; i++) if (blas->nodes[i].IsLeaf()) blas->leaves.push_back(i); // GetLeafIndices :500-514
    }
    blas->sah = ComputeGlobalSAH(*blas, s);
    return blas;
}
int ref_blas_node_count(void* h) { return (int)((Blas*)h)->nodes.size(); }
int ref_blas_triangle_count(void* h) { return (int)((Blas*)h)->triangles.size(); }
int ref_blas_fragment_count(void* h) { return ((Blas*)h)->fragmentCount; }
int ref_blas_required_stack_size(void* h) { return ((Blas*)h)->RequiredStackSize; }
int ref_blas_parent_count(void* h) { return (int)((Blas*)h)->parents.size(); }
int ref_blas_leaf_count(void* h) { return (int)((Blas*)h)->leaves.size(); }
double ref_blas_sah(void* h) { return ((Blas*)h)->sah; }
void ref_blas_get(void* h, GpuBlasNode* nodes, GpuBlasTriangle* tris, int* parents, int* leaves)
{
    Blas* b = (Blas*)h;
    if (nodes) memcpy(nodes, b->nodes.data(), b->nodes.size() * sizeof(Node));
    if (tris) memcpy(tris, b->triangles.data(), b->triangles.size() * sizeof(GpuBlasTriangle));
    if (parents && !b->parents.empty()) memcpy(parents, b->parents.data(), b->parents.size() * sizeof(int));
    if (leaves && !b->leaves.empty()) memcpy(leaves, b->leaves.data(), b->leaves.size() * sizeof(int));
}
void ref_blas_free(void* h) { delete (Blas*)h; }

// BLAS.Refit (Bvh/BLAS.cs:276-293): bottom-up over the node array (children always have larger ids than parents)
void ref_blas_refit(GpuBlasNode* nodes_, int nodeCount, const float* positions, const GpuBlasTriangle* tris)
{
    Node* nodes = reinterpret_cast<Node*>(nodes_);
    Geometry g = {positions, tris, 0};
    for (int i = nodeCount - 1; i >= 1; i--) {
        Node& p = nodes[i];
        if (p.IsLeaf()) {
            Box b = Box::Empty();
            for (int k = 0; k < p.TriCount; k++) { Tri t = g.GetTri(tris[p.TriStartOrChild + k]); b.Grow(t.p0); b.Grow(t.p1); b.Grow(t.p2); }
            p.SetBounds(b); continue;
        }
        const Node& l = nodes[p.TriStartOrChild]; const Node& r = nodes[p.TriStartOrChild + 1];
        Box m = {l.mn, l.mx}; Box rb = {r.mn, r.mx}; m.Grow(rb);
        p.SetBounds(m);
    }
}

// TLAS.Build (Bvh/TLAS.cs:28-141) with BVH.TlasBuild's GetPrimitive (Bvh/BVH.cs:285-296).
// leafBounds: world-space box per instance (6 floats min,max), computed by ref_instance_world_bounds.
static uint32_t InsertTwoZeros(uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; }
static uint32_t Morton30(Vec3 n) // Utils/MyMath.cs:283-299
{
    uint32_t x = std::min(ToUIntSat(n.x * 1024.0f), 1023u), y = std::min(ToUIntSat(n.y * 1024.0f), 1023u), z = std::min(ToUIntSat(n.z * 1024.0f), 1023u);
    return (InsertTwoZeros(x) << 2) | (InsertTwoZeros(y) << 1) | InsertTwoZeros(z);
}
static Vec3 MapToZeroOne(Vec3 v, Vec3 mn, Vec3 mx) // MyMath.cs:241-257 (Remap with map range 0..1)
{
    Vec3 temp = mx - mn; Vec3 r;
    for (int i = 0; i < 3; i++) { r[i] = (v[i] - mn[i]) / temp[i] * (1.0f - 0.0f) + 0.0f; if (temp[i] == 0.0f) r[i] = 0.0f; }
    return r;
}
void ref_instance_world_bounds(const GpuBlasNode* blasRoot, const GpuMeshTransform* xf, float* outMinMax6) // Box.Transformed (Box.cs:177-187)
{
    Box nb = Box::Empty();
    for (int i = 0; i < 8; i++) {
        Vec3 c = {(i & 1) ? blasRoot->Max[0] : blasRoot->Min[0], (i & 2) ? blasRoot->Max[1] : blasRoot->Min[1], (i & 4) ? blasRoot->Max[2] : blasRoot->Min[2]};
        Vec3 w;
        for (int k = 0; k < 3; k++) w[k] = (c.x * xf->Model[k][0]) + (c.y * xf->Model[k][1]) + (c.z * xf->Model[k][2]) + (1.0f * xf->Model[k][3]);
        nb.Grow(w);
    }
    outMinMax6[0] = nb.mn.x; outMinMax6[1] = nb.mn.y; outMinMax6[2] = nb.mn.z; outMinMax6[3] = nb.mx.x; outMinMax6[4] = nb.mx.y; outMinMax6[5] = nb.mx.z;
}
struct TNode { Vec3 mn; uint32_t id; Vec3 mx; float pad; };
void ref_tlas_build(const float* leafBounds, int primitiveCount, GpuTlasNode* outNodes /* 2n-1 */, int searchRadius)
{
    int nodeCount = std::max(2 * primitiveCount - 1, 0);
    if (nodeCount == 0) return;
    TNode* nodes = reinterpret_cast<TNode*>(outNodes);
    memset(nodes, 0, sizeof(TNode) * (size_t)nodeCount);
    std::vector<TNode> temp(nodeCount);
    memset(temp.data(), 0, sizeof(TNode) * (size_t)nodeCount);
    {
        std::vector<TNode> leafNodes(primitiveCount), sorted(primitiveCount);
        Box global = Box::Empty();
        for (int i = 0; i < primitiveCount; i++) {
            Box b = {{leafBounds[6 * i], leafBounds[6 * i + 1], leafBounds[6 * i + 2]}, {leafBounds[6 * i + 3], leafBounds[6 * i + 4], leafBounds[6 * i + 5]}};
            global.Grow(b);
            leafNodes[i] = {b.mn, (1u << 31) | (uint32_t)i, b.mx, 0.0f};
        }
        RadixSort(leafNodes, sorted, [&](const TNode& n) { Vec3 c = {(n.mx.x + n.mn.x) * 0.5f, (n.mx.y + n.mn.y) * 0.5f, (n.mx.z + n.mn.z) * 0.5f}; return Morton30(MapToZeroOne(c, global.mn, global.mx)); });
        for (int i = 0; i < primitiveCount; i++) nodes[nodeCount - primitiveCount + i] = sorted[i];
    }
    int activeRangeCount = primitiveCount, activeRangeEnd = nodeCount;
    std::vector<int> pref(activeRangeCount);
    while (activeRangeCount > 1) {
        int activeRangeStart = activeRangeEnd - activeRangeCount;
        for (int i = 0; i < activeRangeCount; i++) {
            int a = activeRangeStart + i;
            int s0 = std::max(a - searchRadius, activeRangeStart), s1 = std::min(a + searchRadius + 1, activeRangeEnd);
            float smallest = FLT_MAX; int bestIdx = -1;                       // FindBestMatch :271-301
            Box nb = {nodes[a].mn, nodes[a].mx};
            for (int k = s0; k < s1; k++) { if (k == a) continue; Box m = nb; Box ob = {nodes[k].mn, nodes[k].mx}; m.Grow(ob); float area = m.HalfArea(); if (area < smallest) { smallest = area; bestIdx = k; } }
            pref[i] = bestIdx - activeRangeStart;
        }
        int merged = 0;
        for (int i = 0; i < activeRangeCount; i++) { int b = pref[i]; int c = pref[b]; if (i == c && i < b) merged += 2; }
        int unmerged = activeRangeCount - merged, newNodes = merged / 2;
        int mergedHead = activeRangeEnd - merged;
        int newBegin = mergedHead - unmerged - newNodes;
        int unmergedHead = newBegin;
        for (int i = 0; i < activeRangeCount; i++) {
            int b = pref[i]; int c = pref[b]; int aId = i + activeRangeStart;
            if (i == c) {
                if (i < b) {
                    int bId = b + activeRangeStart;
                    temp[mergedHead + 0] = nodes[aId]; temp[mergedHead + 1] = nodes[bId];
                    Box m = {temp[mergedHead].mn, temp[mergedHead].mx}; Box ob = {temp[mergedHead + 1].mn, temp[mergedHead + 1].mx}; m.Grow(ob);
                    temp[unmergedHead] = {m.mn, (uint32_t)mergedHead, m.mx, 0.0f};
                    unmergedHead++; mergedHead += 2;
                }
            } else temp[unmergedHead++] = nodes[aId];
        }
        memcpy(&nodes[newBegin], &temp[newBegin], sizeof(TNode) * (size_t)(activeRangeEnd - newBegin));
        activeRangeCount -= merged / 2;
        activeRangeEnd -= merged;
    }
}

// KAT helpers
uint32_t ref_float_to_key(float v) { return FloatToKey(v); }
uint32_t ref_morton30(float x, float y, float z) { return Morton30({x, y, z}); }
float ref_half_area(float x, float y, float z) { return HalfArea3(x, y, z); }

} // extern "C"
