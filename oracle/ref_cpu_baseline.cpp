// ORACLE (test infrastructure only; PARITY UNPINNED for this C# half, see ref_math.h header).
// C++ stand-in for the reference's "C# CPU BVH traverse path" (BASELINE.json configs[0]):
//   Gui.Test (Source/Render/Gui.cs:1484-1503) -> BVH.Intersect (Source/Bvh/BVH.cs:162-193, no TLAS)
//   -> BLAS.Intersect (Source/Bvh/BLAS.cs:313-386) -> Intersections.RayVsBox / RayVsTriangle
//   (Source/Shapes/Intersections.cs:363-396).
// C# semantics kept exactly: slab test by DIVISION per box, MinNative/MaxNative (minss/maxss: r = a<b ? a : b),
// triangle accept needs t > 0 and divides by the determinant, root test without the T bound, int[128] stack,
// rows distributed over threads like Parallel.For (Gui.cs:1490).  It is NOT bit-identical to the GLSL path
// (SURVEY.md §8a quirk 6); it is the throughput baseline and a TriangleId cross-check.
// .NET is not available in this environment, so this is a port, not the C# binary.
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../include/idkpt.h"

namespace {
struct Vec3 { float x, y, z; };
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator/(Vec3 a, Vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline Vec3 neg(Vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline float Dot(Vec3 l, Vec3 r) { return (l.x * r.x) + (l.y * r.y) + (l.z * r.z); }
static inline Vec3 Cross(Vec3 l, Vec3 r) { return {l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }
static inline float MinN(float a, float b) { return a < b ? a : b; }
static inline float MaxN(float a, float b) { return a > b ? a : b; }
struct Ray { Vec3 o, d; };

static inline bool RayVsBox(const Ray& ray, const float* bmin, const float* bmax, float* t1, float* t2) // Intersections.cs:363-377
{
    Vec3 t0s = (Vec3{bmin[0], bmin[1], bmin[2]} - ray.o) / ray.d;
    Vec3 t1s = (Vec3{bmax[0], bmax[1], bmax[2]} - ray.o) / ray.d;
    Vec3 ts = {MinN(t0s.x, t1s.x), MinN(t0s.y, t1s.y), MinN(t0s.z, t1s.z)};
    Vec3 tb = {MaxN(t0s.x, t1s.x), MaxN(t0s.y, t1s.y), MaxN(t0s.z, t1s.z)};
    *t1 = MaxN(ts.x, MaxN(ts.y, MaxN(ts.z, 0.0f)));
    *t2 = MinN(tb.x, MinN(tb.y, tb.z));
    return *t1 <= *t2;
}
static inline bool RayVsTriangle(const Ray& ray, Vec3 p0, Vec3 p1, Vec3 p2, Vec3* bary, float* t) // Intersections.cs:379-396
{
    Vec3 v1v0 = p1 - p0, v2v0 = p2 - p0, rov0 = ray.o - p0;
    Vec3 normal = Cross(v1v0, v2v0);
    Vec3 q = Cross(rov0, ray.d);
    float x = Dot(ray.d, normal);
    bary->y = Dot(neg(q), v2v0) / x; bary->z = Dot(q, v1v0) / x;
    bary->x = 1.0f - bary->y - bary->z;
    *t = Dot(neg(normal), rov0) / x;
    return bary->x >= 0.0f && bary->y >= 0.0f && bary->z >= 0.0f && *t > 0.0f;
}
struct Hit { float T; int TriangleId; Vec3 bary; };

template <bool COUNT>
static bool BlasIntersect(const GpuBlasNode* nodes, const GpuBlasTriangle* tris, const float* pos, const Ray& ray, Hit* hit, float tMaxDist, uint64_t* boxTests, uint64_t* triTests) // BLAS.cs:313-386
{
    hit->T = tMaxDist; hit->TriangleId = 0; hit->bary = {0, 0, 0};
    int stack[128]; int stackPtr = 0; int stackTop = 2;
    float a, b;
    if (!RayVsBox(ray, nodes[1].Min, nodes[1].Max, &a, &b)) return false;
    while (true) {
        const GpuBlasNode& L = nodes[stackTop]; const GpuBlasNode& R = nodes[stackTop + 1];
        float tMinLeft, tMinRight, dummy;
        bool hitLeft = RayVsBox(ray, L.Min, L.Max, &tMinLeft, &dummy) && tMinLeft <= hit->T;
        bool hitRight = RayVsBox(ray, R.Min, R.Max, &tMinRight, &dummy) && tMinRight <= hit->T;
        if (COUNT) *boxTests += 2;
        bool intersectLeft = hitLeft && L.TriCount > 0, intersectRight = hitRight && R.TriCount > 0;
        if (intersectLeft || intersectRight) {
            int first = (int)(intersectLeft ? L.TriStartOrChild : R.TriStartOrChild);
            int end = !intersectRight ? (first + (int)L.TriCount) : (int)(R.TriStartOrChild + R.TriCount);
            for (int i = first; i < end; i++) {
                const GpuBlasTriangle& t = tris[i];
                const float* p0 = pos + 3 * (size_t)t.X; const float* p1 = pos + 3 * (size_t)t.Y; const float* p2 = pos + 3 * (size_t)t.Z;
                Vec3 bary; float tt;
                if (RayVsTriangle(ray, {p0[0], p0[1], p0[2]}, {p1[0], p1[1], p1[2]}, {p2[0], p2[1], p2[2]}, &bary, &tt) && tt < hit->T) { hit->bary = bary; hit->T = tt; hit->TriangleId = i; }
            }
            if (COUNT) *triTests += (uint64_t)(end - first);
        }
        bool traverseLeft = hitLeft && L.TriCount == 0, traverseRight = hitRight && R.TriCount == 0;
        if (traverseLeft || traverseRight) {
            if (traverseLeft && traverseRight) {
                bool leftCloser = tMinLeft < tMinRight;
                stackTop = (int)(leftCloser ? L.TriStartOrChild : R.TriStartOrChild);
                stack[stackPtr++] = (int)(leftCloser ? R.TriStartOrChild : L.TriStartOrChild);
            } else stackTop = (int)(traverseLeft ? L.TriStartOrChild : R.TriStartOrChild);
        } else { if (stackPtr == 0) break; stackTop = stack[--stackPtr]; }
    }
    return hit->T != tMaxDist;
}
static inline Vec3 XformRow(const float M[3][4], Vec3 p, float w) // (Vector4(p,w) * Matrix4).Xyz, OpenTK left-to-right
{
    return {(p.x * M[0][0]) + (p.y * M[0][1]) + (p.z * M[0][2]) + (w * M[0][3]), (p.x * M[1][0]) + (p.y * M[1][1]) + (p.z * M[1][2]) + (w * M[1][3]), (p.x * M[2][0]) + (p.y * M[2][1]) + (p.z * M[2][2]) + (w * M[2][3])};
}
} // namespace

extern "C" {
// Gui.Test: one primary ray per pixel of rows [y0, y1), ndc = (x,y)/res*2-1 (no jitter).  outT/outTri may be NULL.
// Returns rays traced.  threads<=0 => hardware_concurrency.  counters!=NULL => also counts box/triangle tests (slower,
// like the Interlocked counters of BLAS.cs:338,359).
uint64_t ref_cpu_trace_primary(const idkpt_scene_desc* sc, int W, int H, int y0, int y1, const float* invProj, const float* invView, const float* viewPos,
                               int threads, float* outT, int32_t* outTri, uint64_t* counters2)
{
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    std::atomic<int> nextRow(y0);
    std::vector<uint64_t> boxC(threads, 0), triC(threads, 0);
    auto worker = [&](int tid) {
        uint64_t bt = 0, tt = 0;
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= y1) break;
            for (int x = 0; x < W; x++) {
                float nx = (float)x / (float)W * 2.0f - 1.0f, ny = (float)y / (float)H * 2.0f - 1.0f;
                float rx = nx * invProj[0] + ny * invProj[4], ry = nx * invProj[1] + ny * invProj[5]; // Ray.GetWorldSpaceRay (Shapes/Ray.cs:30-39)
                Vec3 rw = {(rx * invView[0]) + (ry * invView[4]) + (-1.0f * invView[8]) + (0.0f * invView[12]),
                           (rx * invView[1]) + (ry * invView[5]) + (-1.0f * invView[9]) + (0.0f * invView[13]),
                           (rx * invView[2]) + (ry * invView[6]) + (-1.0f * invView[10]) + (0.0f * invView[14])};
                float scale = 1.0f / sqrtf((rw.x * rw.x) + (rw.y * rw.y) + (rw.z * rw.z));
                Ray ray = {{viewPos[0], viewPos[1], viewPos[2]}, {rw.x * scale, rw.y * scale, rw.z * scale}};
                float bestT = FLT_MAX; int bestTri = -1;                               // BVH.Intersect (BVH.cs:170-192)
                for (int i = 0; i < sc->BlasInstanceCount; i++) {
                    const GpuBlasInstance& inst = sc->BlasInstances[i];
                    const GpuBlasDesc& d = sc->BlasDescs[inst.BlasId];
                    const GpuMeshTransform& xf = sc->MeshTransforms[inst.MeshTransformId];
                    Ray local = {XformRow(xf.InvModel, ray.o, 1.0f), XformRow(xf.InvModel, ray.d, 0.0f)};
                    Hit h; bool hit;
                    if (counters2) hit = BlasIntersect<true>(sc->BlasNodes + d.NodeOffset, sc->BlasTriangles + d.TriangleOffset, sc->VertexPositions, local, &h, bestT, &bt, &tt);
                    else hit = BlasIntersect<false>(sc->BlasNodes + d.NodeOffset, sc->BlasTriangles + d.TriangleOffset, sc->VertexPositions, local, &h, bestT, nullptr, nullptr);
                    if (hit) { bestT = h.T; bestTri = d.TriangleOffset + h.TriangleId; }
                }
                if (outT) { size_t idx = (size_t)(y - y0) * W + x; outT[idx] = bestT; outTri[idx] = bestTri; }
            }
        }
        boxC[tid] = bt; triC[tid] = tt;
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
    for (auto& t : pool) t.join();
    if (counters2) { counters2[0] = counters2[1] = 0; for (int t = 0; t < threads; t++) { counters2[0] += boxC[t]; counters2[1] += triC[t]; } }
    return (uint64_t)W * (uint64_t)(y1 - y0);
}
}
