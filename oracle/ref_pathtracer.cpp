// ORACLE (test infrastructure only).  PINNED against the reference's own shaders run on llvmpipe (oracle/glref/, tests/test_glref.py; see ref_math.h header).
// Sequential CPU restatement of the reference's wavefront path tracer in GLSL semantics, executed in the canonical
// order SURVEY.md §8c fixes (the reference itself is order-nondeterministic through atomicAdd slots):
//   FirstHit enqueues in increasing pixel index, NHit in increasing old slot, Reorder is a stable counting sort.
// Citations relative to /root/reference/IDKEngine/Resource/Shaders unless stated.
//   PathTracing/FirstHit/compute.glsl, PathTracing/NHit/compute.glsl, PathTracing/FinalDraw/compute.glsl,
//   PathTracing/include/{Shading,RussianRoulette}.glsl, PathTracing/CountingSort/**, include/BVHIntersect.glsl,
//   include/Surface.glsl; host schedule Source/Render/PathTracer.cs:214-297.
// OpenMP is used only over independent invocations; every order-dependent step (compaction, sort) is a serial loop.
#include <stdint.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <omp.h>
#include "ref_math.h"
#include "../include/idkpt.h"

using namespace ref;

namespace {

struct Scene {
    std::vector<GpuBlasNode> nodes; std::vector<GpuBlasTriangle> tris; std::vector<GpuBlasDesc> descs;
    std::vector<GpuBlasInstance> instances; std::vector<GpuTlasNode> tlas;
    std::vector<float> positions; std::vector<GpuVertex> vertices; std::vector<GpuMesh> meshes;
    std::vector<GpuMaterial> materials; std::vector<GpuMeshTransform> xforms; std::vector<GpuLight> lights;
    std::vector<float> sky; int skySize = 0;
    struct Tex { int w, h; std::vector<float> rgba; int wrapS = 0, wrapT = 0, magFilter = 0, format = 0; };   // rgba: DECODED texels (8-bit formats are decoded per texel before the filter, GL 4.6 8.24: the decode of the load below)
    std::vector<Tex> textures;
};

struct HitInfo { v2 bary; float T; uint32_t TriangleId; uint32_t MeshTransformId; };
struct Counters { uint64_t pairs = 0, tris = 0; };

static inline v3 P(const Scene& s, uint32_t i) { const float* p = &s.positions[3 * (size_t)i]; return V3(p[0], p[1], p[2]); }

// Traversal stack: the capacity is not part of the semantics (the reference sizes it from the BVH, Bvh/BVH.cs:559-567); fixed storage for the
// depths real BVHs reach (no allocation per ray, like the C# path's stackalloc, Bvh/BLAS.cs:318), heap beyond that.
struct TraversalStack {
    uint32_t fixed[128]; int n = 0; std::vector<uint32_t> spill;
    void push_back(uint32_t v) { if (n < 128) fixed[n] = v; else spill.push_back(v); n++; }
    uint32_t back() const { return n <= 128 ? fixed[n - 1] : spill.back(); }
    void pop_back() { if (n > 128) spill.pop_back(); n--; }
    bool empty() const { return n == 0; }
};

// include/BVHIntersect.glsl:27-105
static bool IntersectBlas(const Scene& s, const Ray& ray, const GpuBlasDesc& d, bool useTlas, HitInfo& hit, float& debugCost, Counters* cnt)
{
    bool anyHit = false;
    float tMinLeft, tMinRight;
    v3 invDir = V3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z); // IntersectionRoutines.glsl:29 (loop-invariant)
    const GpuBlasNode* nodes = s.nodes.data() + d.NodeOffset;
    if (!useTlas) { // :32-39
        const GpuBlasNode& root = nodes[1];
        if (!(RayBoxIntersect(ray.o, invDir, root.Min, root.Max, &tMinLeft) && tMinLeft < hit.T)) return false;
    }
    TraversalStack stack; // shared uint BlasTraversalStack[BLAS_STACK_SIZE][..] (:18-22)
    uint32_t stackTop = 2;
    while (true) {
        debugCost += 1.0f; if (cnt) cnt->pairs++;
        const GpuBlasNode& L = nodes[stackTop]; const GpuBlasNode& R = nodes[stackTop + 1];
        bool hitLeft = RayBoxIntersect(ray.o, invDir, L.Min, L.Max, &tMinLeft) && tMinLeft <= hit.T;
        bool hitRight = RayBoxIntersect(ray.o, invDir, R.Min, R.Max, &tMinRight) && tMinRight <= hit.T;
        bool intersectLeft = hitLeft && L.TriCount > 0, intersectRight = hitRight && R.TriCount > 0;
        if (intersectLeft || intersectRight) {
            uint32_t first = intersectLeft ? L.TriStartOrChild : R.TriStartOrChild;
            uint32_t end = !intersectRight ? (L.TriStartOrChild + L.TriCount) : (R.TriStartOrChild + R.TriCount);
            first += (uint32_t)d.TriangleOffset; end += (uint32_t)d.TriangleOffset;
            debugCost += (float)(end - first) * 1.1f; if (cnt) cnt->tris += end - first;
            for (uint32_t i = first; i < end; i++) {
                const GpuBlasTriangle& t = s.tris[i];
                v3 bary; float hitT;
                if (RayTriangleIntersect(ray, P(s, t.X), P(s, t.Y), P(s, t.Z), &bary, &hitT) && hitT < hit.T) {
                    anyHit = true; hit.TriangleId = i; hit.bary.x = bary.x; hit.bary.y = bary.y; hit.T = hitT;
                }
            }
        }
        bool traverseLeft = hitLeft && L.TriCount == 0, traverseRight = hitRight && R.TriCount == 0;
        if (traverseLeft || traverseRight) {
            if (traverseLeft && traverseRight) {
                bool leftCloser = tMinLeft < tMinRight;
                stackTop = leftCloser ? L.TriStartOrChild : R.TriStartOrChild;
                stack.push_back(leftCloser ? R.TriStartOrChild : L.TriStartOrChild);
            } else stackTop = traverseLeft ? L.TriStartOrChild : R.TriStartOrChild;
        } else {
            if (stack.empty()) break;
            stackTop = stack.back(); stack.pop_back();
        }
    }
    return anyHit;
}

static inline Ray RayTransform(const Ray& r, const float M[3][4]) { Ray o; o.o = xform34(M, r.o, 1.0f); o.d = xform34(M, r.d, 0.0f); return o; } // Ray.glsl:7-12

// include/BVHIntersect.glsl:183-291
static bool TraceRay(const Scene& s, const Ray& ray, HitInfo& hit, float& debugCost, bool traceLights, bool useTlas, float maxDist, Counters* cnt)
{
    hit.T = maxDist; hit.TriangleId = ~0u; hit.MeshTransformId = 0; hit.bary.x = hit.bary.y = 0.0f;
    debugCost = 0.0f;
    if (traceLights) { // :189-203
        for (int i = 0; i < (int)s.lights.size(); i++) {
            const GpuLight& l = s.lights[i]; float tMin, tMax;
            if (RaySphereIntersect(ray, V3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hit.T) {
                hit.T = tMin < 0.0f ? tMax : tMin; hit.MeshTransformId = (uint32_t)i; hit.TriangleId = ~0u;
            }
        }
    }
    if (useTlas) { // :205-272
        if (s.tlas.empty()) return hit.T != maxDist;
        float tMinLeft, tMinRight;
        v3 invDir = V3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        TraversalStack stack; uint32_t stackTop = 0;
        while (true) {
            const GpuTlasNode& parent = s.tlas[stackTop];
            bool isLeaf = (parent.IsLeafAndChildOrInstanceId >> 31) == 1;
            uint32_t id = parent.IsLeafAndChildOrInstanceId & ((1u << 31) - 1);
            if (isLeaf) {
                const GpuBlasInstance& inst = s.instances[id];
                Ray local = RayTransform(ray, s.xforms[inst.MeshTransformId].InvModel);
                if (IntersectBlas(s, local, s.descs[inst.BlasId], true, hit, debugCost, cnt)) hit.MeshTransformId = inst.MeshTransformId;
                if (stack.empty()) break;
                stackTop = stack.back(); stack.pop_back();
                continue;
            }
            uint32_t l = id, r = id + 1;
            const GpuTlasNode& ln = s.tlas[l]; const GpuTlasNode& rn = s.tlas[r];
            bool tl = RayBoxIntersect(ray.o, invDir, ln.Min, ln.Max, &tMinLeft) && tMinLeft < hit.T;
            bool tr = RayBoxIntersect(ray.o, invDir, rn.Min, rn.Max, &tMinRight) && tMinRight < hit.T;
            if (tl || tr) {
                if (tl && tr) { bool lc = tMinLeft < tMinRight; stackTop = lc ? l : r; stack.push_back(lc ? r : l); }
                else stackTop = tl ? l : r;
            } else { if (stack.empty()) break; stackTop = stack.back(); stack.pop_back(); }
        }
    } else { // :275-287
        for (size_t i = 0; i < s.instances.size(); i++) {
            const GpuBlasInstance& inst = s.instances[i];
            Ray local = RayTransform(ray, s.xforms[inst.MeshTransformId].InvModel);
            if (IntersectBlas(s, local, s.descs[inst.BlasId], false, hit, debugCost, cnt)) hit.MeshTransformId = inst.MeshTransformId;
        }
    }
    return hit.T != maxDist;
}

// include/BVHIntersect.glsl:107-181 — any-hit BLAS traversal: returns at the first triangle with hitT < T; children are always
// visited left first (no near/far ordering, :163-167).
static bool IntersectBlasAny(const Scene& s, const Ray& ray, const GpuBlasDesc& d, bool useTlas, HitInfo& hit)
{
    float tMinLeft, tMinRight;
    v3 invDir = V3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    const GpuBlasNode* nodes = s.nodes.data() + d.NodeOffset;
    if (!useTlas) { // :112-118
        const GpuBlasNode& root = nodes[1];
        if (!(RayBoxIntersect(ray.o, invDir, root.Min, root.Max, &tMinLeft) && tMinLeft < hit.T)) return false;
    }
    TraversalStack stack;
    uint32_t stackTop = 2;
    while (true) {
        const GpuBlasNode& L = nodes[stackTop]; const GpuBlasNode& R = nodes[stackTop + 1];
        bool hitLeft = RayBoxIntersect(ray.o, invDir, L.Min, L.Max, &tMinLeft) && tMinLeft <= hit.T;
        bool hitRight = RayBoxIntersect(ray.o, invDir, R.Min, R.Max, &tMinRight) && tMinRight <= hit.T;
        bool intersectLeft = hitLeft && L.TriCount > 0, intersectRight = hitRight && R.TriCount > 0;
        if (intersectLeft || intersectRight) {
            uint32_t first = intersectLeft ? L.TriStartOrChild : R.TriStartOrChild;
            uint32_t end = !intersectRight ? (L.TriStartOrChild + L.TriCount) : (R.TriStartOrChild + R.TriCount);
            first += (uint32_t)d.TriangleOffset; end += (uint32_t)d.TriangleOffset;
            for (uint32_t i = first; i < end; i++) {
                const GpuBlasTriangle& t = s.tris[i];
                v3 bary; float hitT;
                if (RayTriangleIntersect(ray, P(s, t.X), P(s, t.Y), P(s, t.Z), &bary, &hitT) && hitT < hit.T) {
                    hit.TriangleId = i; hit.bary.x = bary.x; hit.bary.y = bary.y; hit.T = hitT;
                    return true;
                }
            }
        }
        bool traverseLeft = hitLeft && L.TriCount == 0, traverseRight = hitRight && R.TriCount == 0;
        if (traverseLeft || traverseRight) {
            if (traverseLeft && traverseRight) { stackTop = L.TriStartOrChild; stack.push_back(R.TriStartOrChild); }
            else stackTop = traverseLeft ? L.TriStartOrChild : R.TriStartOrChild;
        } else {
            if (stack.empty()) break;
            stackTop = stack.back(); stack.pop_back();
        }
    }
    return false;
}

// include/BVHIntersect.glsl:299-411
static bool TraceRayAny(const Scene& s, const Ray& ray, HitInfo& hit, bool traceLights, bool useTlas, float maxDist)
{
    hit.T = maxDist; hit.TriangleId = ~0u; hit.MeshTransformId = 0; hit.bary.x = hit.bary.y = 0.0f;
    if (traceLights) { // :304-320
        for (int i = 0; i < (int)s.lights.size(); i++) {
            const GpuLight& l = s.lights[i]; float tMin, tMax;
            if (RaySphereIntersect(ray, V3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hit.T) {
                hit.T = tMin < 0.0f ? tMax : tMin; hit.MeshTransformId = (uint32_t)i;
                return true;
            }
        }
    }
    if (useTlas) { // :323-385
        if (s.tlas.empty()) return false;
        float tMinLeft, tMinRight;
        v3 invDir = V3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        TraversalStack stack; uint32_t stackTop = 0;
        while (true) {
            const GpuTlasNode& parent = s.tlas[stackTop];
            bool isLeaf = (parent.IsLeafAndChildOrInstanceId >> 31) == 1;
            uint32_t id = parent.IsLeafAndChildOrInstanceId & ((1u << 31) - 1);
            if (isLeaf) {
                const GpuBlasInstance& inst = s.instances[id];
                Ray local = RayTransform(ray, s.xforms[inst.MeshTransformId].InvModel);
                if (IntersectBlasAny(s, local, s.descs[inst.BlasId], true, hit)) { hit.MeshTransformId = inst.MeshTransformId; return true; }
                if (stack.empty()) break;
                stackTop = stack.back(); stack.pop_back();
                continue;
            }
            uint32_t l = id, r = id + 1;
            const GpuTlasNode& ln = s.tlas[l]; const GpuTlasNode& rn = s.tlas[r];
            bool tl = RayBoxIntersect(ray.o, invDir, ln.Min, ln.Max, &tMinLeft) && tMinLeft < hit.T;
            bool tr = RayBoxIntersect(ray.o, invDir, rn.Min, rn.Max, &tMinRight) && tMinRight < hit.T;
            if (tl || tr) {
                if (tl && tr) { bool lc = tMinLeft < tMinRight; stackTop = lc ? l : r; stack.push_back(lc ? r : l); }
                else stackTop = tl ? l : r;
            } else { if (stack.empty()) break; stackTop = stack.back(); stack.pop_back(); }
        }
    } else { // :388-408
        for (size_t i = 0; i < s.instances.size(); i++) {
            const GpuBlasInstance& inst = s.instances[i];
            Ray local = RayTransform(ray, s.xforms[inst.MeshTransformId].InvModel);
            if (IntersectBlasAny(s, local, s.descs[inst.BlasId], false, hit)) { hit.MeshTransformId = inst.MeshTransformId; return true; }
        }
    }
    return false;
}

// ---- texture / sky sampling (stand-in for GL bindless samplers; DESIGN.md "Textures") ----
// handle 0 = 1x1 white (Utils/ModelLoader.cs:1857-1877).  Bilinear, repeat wrap, LOD 0, texel centres at (i+0.5)/w.
struct RGBA { float r, g, b, a; };
// Sampler arithmetic of the checker (process-wide, ref_set_sampler_mode).  0 (default, what the HIP path implements): the filter as the GL specification writes it —
// u * size - 0.5 on the unreduced coordinate, weights = its fraction, GLSL mix for the two lerps.  1: the same filter in the arithmetic Mesa llvmpipe uses for a
// non-power-of-two REPEAT dimension of a float texture (lp_bld_sample_soa.c: the coordinate is reduced to [0, 1) FIRST — `coord = fract(coord)` — then unnormalised, and the
// lerps are v0 + w * (v1 - v0)).  GL leaves both choices to the implementation; mode 1 exists to show that the few rays of textured cases beyond the 1e-4 gate against
// llvmpipe are this freedom and nothing else (tests/test_glref.py, oracle/glref/fuzz_reference.py --sampler llvmpipe).
static int g_sampler_mode = 0;
extern "C" void ref_set_sampler_mode(int mode) { g_sampler_mode = mode; }
static bool is_pot(int n) { return n > 0 && (n & (n - 1)) == 0; }
static RGBA SampleTexLlvmpipe(const Scene::Tex& t, v2 uv)
{
    float w[2]; int i0[2], i1[2];
    const float u[2] = {uv.x, uv.y}; const int n[2] = {t.w, t.h};
    for (int a = 0; a < 2; a++) {
        float c;
        if (is_pot(n[a])) c = u[a] * (float)n[a] - 0.5f;                        // power of two: unnormalise, then wrap the integer coordinate with a mask
        else { float f = u[a] - gfloor(u[a]); c = f * (float)n[a] - 0.5f; }     // otherwise: fract first
        const float fl = gfloor(c);
        w[a] = c - fl;
        int k0 = (int)fl, k1 = k0 + 1;
        k0 %= n[a]; if (k0 < 0) k0 += n[a];
        k1 %= n[a]; if (k1 < 0) k1 += n[a];
        i0[a] = k0; i1[a] = k1;
    }
    float c[4];
    for (int k = 0; k < 4; k++) {
        const float v00 = t.rgba[4 * ((size_t)i0[1] * t.w + i0[0]) + k], v01 = t.rgba[4 * ((size_t)i0[1] * t.w + i1[0]) + k];
        const float v10 = t.rgba[4 * ((size_t)i1[1] * t.w + i0[0]) + k], v11 = t.rgba[4 * ((size_t)i1[1] * t.w + i1[0]) + k];
        const float r0 = v00 + w[0] * (v01 - v00), r1 = v10 + w[0] * (v11 - v10);
        c[k] = r0 + w[1] * (r1 - r0);
    }
    RGBA o = {c[0], c[1], c[2], c[3]}; return o;
}
// One image of the texture table as the sampler sees it: texels decoded to float (idkpt_texture, include/idkpt.h).  UNORM: c / (2^8 - 1) (GL 4.6 2.3.5.1); sRGB: the transfer
// function of 8.24 on R, G, B — evaluated in double, rounded once — alpha linear.
static Scene::Tex load_texture(const idkpt_texture& src)
{
    Scene::Tex t; t.w = src.width; t.h = src.height; t.wrapS = src.wrapS; t.wrapT = src.wrapT; t.magFilter = src.magFilter; t.format = src.format;
    const size_t n = 4 * (size_t)t.w * t.h;
    if (src.format == IDKPT_TEXFMT_RGBA32F) { const float* f = (const float*)src.rgba; t.rgba.assign(f, f + n); return t; }
    const uint8_t* b = (const uint8_t*)src.rgba; t.rgba.resize(n);
    for (size_t k = 0; k < n; k++) {
        if (src.format == IDKPT_TEXFMT_SRGB8_A8 && (k & 3) != 3) { const double cs = (double)b[k] / 255.0; t.rgba[k] = (float)(cs <= 0.04045 ? cs / 12.92 : pow((cs + 0.055) / 1.055, 2.4)); }
        else t.rgba[k] = (float)b[k] / 255.0f;
    }
    return t;
}
// wrap(coord) of GL 4.6 table 8.20 on an integer texel coordinate (8.14.2): what the texture's WrapModeS / WrapModeT (Utils/ModelLoader.cs:1166-1197) select
static int WrapTexel(int i, int n, int mode)
{
    if (mode == IDKPT_WRAP_CLAMP_TO_EDGE) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);                    // clamp(coord, 0, size - 1)
    if (mode == IDKPT_WRAP_MIRRORED_REPEAT) { int m = i % (2 * n); if (m < 0) m += 2 * n; const int a = m - n; return (n - 1) - (a >= 0 ? a : -(1 + a)); }   // (size - 1) - mirror((coord mod (2 size)) - size), mirror(a) = a >= 0 ? a : -(1 + a)
    int k = i % n; if (k < 0) k += n; return k;                                                          // coord mod size
}
static RGBA SampleTex(const Scene& s, uint64_t handle, v2 uv)
{
    if (handle == 0 || handle > s.textures.size()) { RGBA w = {1, 1, 1, 1}; return w; }
    const Scene::Tex& t = s.textures[handle - 1];
    if (t.w == 1 && t.h == 1) { RGBA c = {t.rgba[0], t.rgba[1], t.rgba[2], t.rgba[3]}; return c; }
    if (g_sampler_mode == 1 && t.wrapS == 0 && t.wrapT == 0 && t.magFilter == 0) return SampleTexLlvmpipe(t, uv);
    if (t.magFilter == IDKPT_FILTER_NEAREST) {                                                          // 8.14.2: i = wrap(floor(u')), no filter
        const int x = WrapTexel((int)gfloor(uv.x * (float)t.w), t.w, t.wrapS), y = WrapTexel((int)gfloor(uv.y * (float)t.h), t.h, t.wrapT);
        const float* p = &t.rgba[4 * ((size_t)y * t.w + x)];
        RGBA c = {p[0], p[1], p[2], p[3]}; return c;
    }
    float fx = uv.x * (float)t.w - 0.5f, fy = uv.y * (float)t.h - 0.5f;
    float x0f = gfloor(fx), y0f = gfloor(fy);
    float ax = fx - x0f, ay = fy - y0f;
    auto wrapS = [&](float v) { return WrapTexel((int)v, t.w, t.wrapS); }; auto wrapT = [&](float v) { return WrapTexel((int)v, t.h, t.wrapT); };
    int x0 = wrapS(x0f), x1 = wrapS(x0f + 1.0f), y0 = wrapT(y0f), y1 = wrapT(y0f + 1.0f);
    float c[4];
    for (int k = 0; k < 4; k++) {
        float a = t.rgba[4 * ((size_t)y0 * t.w + x0) + k], b = t.rgba[4 * ((size_t)y0 * t.w + x1) + k];
        float cc = t.rgba[4 * ((size_t)y1 * t.w + x0) + k], d = t.rgba[4 * ((size_t)y1 * t.w + x1) + k];
        c[k] = gmix(gmix(a, b, ax), gmix(cc, d, ax), ay);
    }
    RGBA o = {c[0], c[1], c[2], c[3]}; return o;
}
// Sky lookup = texture(skyBoxUBO.Albedo, dir) on a GL_LINEAR, seamless cube map (Render/SkyBoxManager.cs:44,74; FirstHit:227, NHit:208).
// Face selection and (s,t): GL 4.6 spec table 8.19.  Linear filter at LOD 0: texel centres at (i+0.5)/S, weights frac(u*S-0.5).
// Seamless: a texel one step beyond a face edge is the texel of the adjacent face it folds onto (SkyFold, exact integer lattice);
// the one texel beyond a face CORNER has no owner and is the mean of the three other texels of the footprint (the behaviour the GL
// spec recommends, section 8.14.1).  S = 1 is this library's "constant colour per face" stand-in and stays unfiltered.
static inline void SkyFold(int S, int& face, int& x, int& y)
{
    int sc = 2 * x + 1 - S, tc = 2 * y + 1 - S;          // lattice coordinates of the texel centre: the face spans (-S, S), centres 2 apart
    int p[3];
    switch (face) {                                       // inverse of table 8.19: (face, sc, tc) -> point on the cube of half-size S
        case 0: p[0] = S; p[1] = -tc; p[2] = -sc; break;
        case 1: p[0] = -S; p[1] = -tc; p[2] = sc; break;
        case 2: p[1] = S; p[0] = sc; p[2] = tc; break;
        case 3: p[1] = -S; p[0] = sc; p[2] = -tc; break;
        case 4: p[2] = S; p[0] = sc; p[1] = -tc; break;
        default: p[2] = -S; p[0] = -sc; p[1] = -tc; break;
    }
    const int m = face >> 1;
    int o = -1;
    for (int k = 0; k < 3; k++) if (k != m && (p[k] > S || p[k] < -S)) o = k;
    if (o < 0) return;                                     // inside the face
    p[m] = p[m] > 0 ? S - 1 : -(S - 1);                    // one step beyond the edge folds to the outermost row of the neighbour
    p[o] = p[o] > 0 ? S : -S;
    face = 2 * o + (p[o] > 0 ? 0 : 1);
    switch (face) {
        case 0: sc = -p[2]; tc = -p[1]; break;
        case 1: sc = p[2]; tc = -p[1]; break;
        case 2: sc = p[0]; tc = p[2]; break;
        case 3: sc = p[0]; tc = -p[2]; break;
        case 4: sc = p[0]; tc = -p[1]; break;
        default: sc = -p[0]; tc = -p[1]; break;
    }
    x = (sc + S - 1) / 2; y = (tc + S - 1) / 2;
}
static v3 SampleSky(const Scene& s, v3 d)
{
    if (s.skySize <= 0) return V3s(0.0f);
    float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
    int face; float sc, tc, ma;
    if (ax >= ay && ax >= az) { face = d.x >= 0.0f ? 0 : 1; sc = d.x >= 0.0f ? -d.z : d.z; tc = -d.y; ma = ax; }
    else if (ay >= az) { face = d.y >= 0.0f ? 2 : 3; sc = d.x; tc = d.y >= 0.0f ? d.z : -d.z; ma = ay; }
    else { face = d.z >= 0.0f ? 4 : 5; sc = d.z >= 0.0f ? d.x : -d.x; tc = -d.y; ma = az; }
    const int S = s.skySize;
    if (S == 1) { const float* p = &s.sky[4 * (size_t)face]; return V3(p[0], p[1], p[2]); }
    float u = 0.5f * (sc / ma + 1.0f), v = 0.5f * (tc / ma + 1.0f);
    float fx = u * (float)S - 0.5f, fy = v * (float)S - 0.5f;
    float x0f = gfloor(fx), y0f = gfloor(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)gmin(gmax(x0f, -1.0f), (float)(S - 1)), y0 = (int)gmin(gmax(y0f, -1.0f), (float)(S - 1));
    v3 t[4]; bool corner[4]; int nCorner = 0;
    for (int k = 0; k < 4; k++) {                          // footprint order: (x0,y0) (x1,y0) (x0,y1) (x1,y1)
        int x = x0 + (k & 1), y = y0 + (k >> 1), f = face;
        corner[k] = (x < 0 || x >= S) && (y < 0 || y >= S);
        if (corner[k]) { nCorner++; continue; }
        SkyFold(S, f, x, y);
        const float* p = &s.sky[4 * (((size_t)f * S + y) * S + x)];
        t[k] = V3(p[0], p[1], p[2]);
    }
    if (nCorner) {
        v3 sum = V3s(0.0f);
        for (int k = 0; k < 4; k++) if (!corner[k]) sum = sum + t[k];
        for (int k = 0; k < 4; k++) if (corner[k]) t[k] = sum / 3.0f;
    }
    return V3(gmix(gmix(t[0].x, t[1].x, wx), gmix(t[2].x, t[3].x, wx), wy),
              gmix(gmix(t[0].y, t[1].y, wx), gmix(t[2].y, t[3].y, wx), wy),
              gmix(gmix(t[0].z, t[1].z, wx), gmix(t[2].z, t[3].z, wx), wy));
}

// include/Surface.glsl
struct Surface { v3 Albedo; float Alpha; v3 Normal; v3 Emissive; v3 Absorbance; float Metallic, Roughness, Transmission, IOR, AlphaCutoff; bool IsVolumetric, TintOnTransmissive; };
static Surface GetDefaultSurface() // :25-47
{
    Surface s; s.Albedo = V3s(1.0f); s.Alpha = 1.0f; s.Normal = V3s(0.0f); s.Emissive = V3s(0.0f); s.Absorbance = V3s(0.0f);
    s.Metallic = 0.0f; s.Roughness = 0.0f; s.Transmission = 0.0f; s.IOR = 1.5f; s.AlphaCutoff = 0.5f; s.IsVolumetric = false; s.TintOnTransmissive = true;
    return s;
}
static Surface GetSurface(const Scene& sc, const GpuMaterial& m, v2 uv) // :49-77 (compute stage: no lod bias)
{
    Surface s;
    float f[4]; unpackUnorm4x8(m.BaseColorFactor, f);
    RGBA bc = SampleTex(sc, m.BaseColorTexture, uv);
    s.Albedo = V3(bc.r * f[0], bc.g * f[1], bc.b * f[2]); s.Alpha = bc.a * f[3];
    RGBA nm = SampleTex(sc, m.NormalTexture, uv); v2 nrg = {nm.r, nm.g};
    s.Normal = ReconstructPackedNormal(nrg);
    RGBA em = SampleTex(sc, m.EmissiveTexture, uv);
    s.Emissive = V3(em.r * m.EmissiveFactor[0], em.g * m.EmissiveFactor[1], em.b * m.EmissiveFactor[2]);
    s.Absorbance = V3(m.Absorbance[0], m.Absorbance[1], m.Absorbance[2]);
    RGBA mr = SampleTex(sc, m.MetallicRoughnessTexture, uv);
    s.Metallic = mr.r * m.MetallicFactor; s.Roughness = mr.g * m.RoughnessFactor;
    RGBA tr = SampleTex(sc, m.TransmissionTexture, uv);
    s.Transmission = tr.r * m.TransmissionFactor; s.IOR = m.IOR;
    s.AlphaCutoff = m.AlphaCutoff; s.IsVolumetric = m.IsVolumetric != 0; s.TintOnTransmissive = true;
    return s;
}
static void SurfaceApplyModificatons(Surface& s, const GpuMesh& mesh) // :79-91 (SURFACE_EMISSIVE_FACTOR 1.0)
{
    s.Emissive = s.Emissive * 1.0f + mesh.EmissiveBias * s.Albedo;
    s.Absorbance = V3(gmax(s.Absorbance.x + mesh.AbsorbanceBias[0], 0.0f), gmax(s.Absorbance.y + mesh.AbsorbanceBias[1], 0.0f), gmax(s.Absorbance.z + mesh.AbsorbanceBias[2], 0.0f));
    s.Metallic = gclamp(s.Metallic + mesh.SpecularBias, 0.0f, 1.0f);
    s.Roughness = gclamp(s.Roughness + mesh.RoughnessBias, 0.0f, 1.0f);
    s.Transmission = gclamp(s.Transmission + mesh.TransmissionBias, 0.0f, 1.0f);
    s.IOR = gmax(s.IOR + mesh.IORBias, 1.0f);
    s.TintOnTransmissive = mesh.TintOnTransmissive != 0;
}
static inline float GetSurfaceVariance(float spec, float trans, float rough) { float diffuse = 1.0f - spec - trans; return diffuse + spec * rough + trans * rough; } // :103-108

// PathTracing/include/Shading.glsl
enum { BSDF_DIFFUSE = 0, BSDF_SPECULAR = 1, BSDF_TRANSMISSIVE = 2 };
struct SampleMaterialResult { v3 RayDirection; uint32_t BsdfType; v3 Bsdf; float Pdf; float NewIor; };
static SampleMaterialResult SampleMaterial(v3 incomming, Surface surface, float prevIor, bool fromInside, Rng* rng, uint32_t gidSeed, uint32_t accumulatedSamples) // :59-150
{
    surface.Roughness *= surface.Roughness;
    float cosTheta = dot(-incomming, surface.Normal);
    {
        float diffuseChance = 1.0f - surface.Metallic - surface.Transmission;
        float f0 = BaseReflectivity(prevIor, surface.IOR);                                   // SpecularBasedOnViewAngle :21-29
        surface.Metallic = gmix(surface.Metallic, 1.0f, FresnelSchlick(f0, 1.0f, cosTheta));
        surface.Transmission = gmax(1.0f - diffuseChance - surface.Metallic, 0.0f);
    }
    SampleMaterialResult result; result.NewIor = 0.0f; result.Bsdf = V3s(0.0f); result.Pdf = 0.0f; result.RayDirection = V3s(0.0f);
    { // SelectBsdf :31-52
        float specularChance = surface.Metallic, transmissionChance = surface.Transmission;
        float rnd = rnd01(rng);
        if (specularChance > rnd) result.BsdfType = BSDF_SPECULAR;
        else if (specularChance + transmissionChance > rnd) result.BsdfType = BSDF_TRANSMISSIVE;
        else result.BsdfType = BSDF_DIFFUSE;
    }
    v3 diffuseRayDir;
    { // :73-79 — reseeds from gl_GlobalInvocationID, then restores
        Rng local; local.seed = gidSeed;
        v2 r2 = R2Sequence(accumulatedSamples);
        v2 pixelOffset; pixelOffset.x = rnd01(&local); pixelOffset.y = rnd01(&local);
        v2 uv = DecorrelateSequence(r2, pixelOffset);
        diffuseRayDir = CosineSampleHemisphere(surface.Normal, uv);
    }
    if (result.BsdfType == BSDF_DIFFUSE) {
        result.RayDirection = diffuseRayDir; result.NewIor = prevIor; result.Bsdf = surface.Albedo; result.Pdf = 1.0f;
    } else if (result.BsdfType == BSDF_SPECULAR) {
        v3 refl = reflect(incomming, surface.Normal);
        refl = normalize(gmix(refl, diffuseRayDir, surface.Roughness));
        result.RayDirection = refl; result.Bsdf = surface.Albedo; result.Pdf = 1.0f; result.NewIor = prevIor;
    } else {
        result.NewIor = fromInside ? 1.0f : surface.IOR;
        v3 refr; bool tir;
        if (!surface.IsVolumetric) { refr = incomming; tir = false; result.NewIor = 1.0f; }
        else {
            refr = refract(incomming, surface.Normal, prevIor / result.NewIor);
            tir = (refr.x == 0.0f && refr.y == 0.0f && refr.z == 0.0f);
            if (tir) { refr = reflect(incomming, surface.Normal); result.NewIor = prevIor; }
        }
        refr = normalize(gmix(refr, !tir ? -diffuseRayDir : diffuseRayDir, surface.Roughness));
        result.RayDirection = refr;
        bool gltfWantsTint = surface.IsVolumetric || !fromInside;
        result.Bsdf = (gltfWantsTint && surface.TintOnTransmissive) ? surface.Albedo : V3s(1.0f);
        result.Pdf = 1.0f;
    }
    result.Pdf = gmax(result.Pdf, 0.0001f);
    return result;
}
// PathTracing/include/RussianRoulette.glsl:3-12
static bool RussianRouletteTerminateRay(v3& throughput, Rng* rng)
{
    float p = gmax(throughput.x, gmax(throughput.y, throughput.z));
    if (rnd01(rng) > p) return true;
    throughput = throughput / p;
    return false;
}

struct PT {
    const Scene* scene;
    int W = 0, H = 0, rowMod = 1, rowRem = 0, rows = 0, rowBand = 1;   // rowBand: rows are dealt in bands of that many rows (idkptSetRowBands)
    idkpt_settings st;
    float invProj[16], invView[16], viewPos[3];
    uint32_t accumulated = 0;
    uint32_t seqFirst = 0, seqStride = 1;   // idkptSetSampleSequence: AccumulatedSamples the shaders see = seqFirst + accumulated * seqStride
    uint32_t sampleIndex() const { return seqFirst + accumulated * seqStride; }
    std::vector<GpuWavefrontRay> rays; std::vector<GpuAovRay> aov;
    std::vector<float> img[3];
    std::vector<uint32_t> alive; // queue (local pixel indices)
    std::vector<uint32_t> keys;
    std::vector<float> primT, primBary; std::vector<uint32_t> primTri;
    uint32_t aliveCounts[16];
    idkpt_bounce_exchange_fn exchangeFn = nullptr; void* exchangeUser = nullptr;   // multi-context exact mode (idkptSetBounceExchange)
    idkpt_band_exchange_fn bandExchangeFn = nullptr; void* bandExchangeUser = nullptr;   // ... for interleaved rows / bands (idkptSetBandExchange)
    Counters counters; bool countersOn = false;
    uint64_t raysTraced = 0;
    // test hooks (oracle/glref/fuzz_reference.py, the textured-stage check): in stage uvStage (0 = FirstHit, j = NHit j) the texture coordinate a ray interpolates is written to
    // uvDump[2 * rayIndex ..] and / or replaced by uvOverride[2 * rayIndex ..] where that is not NaN — the taps of the stage then read exactly the coordinates another execution read
    int uvStage = -1; const float* uvOverride = nullptr; float* uvDump = nullptr;
    double parallelSec = 0.0, totalSec = 0.0;   // cpu_baseline leg of bench.py: time inside the per-invocation (OpenMP) sections / whole RenderSample
};

// gl_GlobalInvocationID of the FirstHit invocation that shades pixel (px,py): inverse of ReorderInvocations(20)
// (FirstHit/compute.glsl:236-262) for an 8x8 group grid of ceil(W/8) x ceil(H/8).
static void FirstHitGid(int W, int H, int px, int py, uint32_t* gx, uint32_t* gy)
{
    const uint32_t n = 20;
    uint32_t numX = (uint32_t)(W + 7) / 8, numY = (uint32_t)(H + 7) / 8;
    uint32_t sx = (uint32_t)px / 8, sy = (uint32_t)py / 8;
    uint32_t columnSize = numY * n, fullColumnCount = numX / n, lastColumnWidth = numX % n;
    uint32_t columnIdx = sx / n;
    uint32_t columnWidth = (columnIdx == fullColumnCount) ? lastColumnWidth : n;
    uint32_t idxInColumn = sy * columnWidth + (sx - columnIdx * n);
    uint32_t idx = columnIdx * columnSize + idxInColumn;
    uint32_t wgY = idx / numX, wgX = idx % numX;
    *gx = wgX * 8 + (uint32_t)px % 8; *gy = wgY * 8 + (uint32_t)py % 8;
}

// FirstHit TraceRay (FirstHit/compute.glsl:100-234) and NHit TraceRay (NHit/compute.glsl:91-215) share this body.
static bool ShadeRay(PT& pt, bool first, GpuWavefrontRay& wr, GpuAovRay& ar, Rng* rng, uint32_t gidSeed, uint32_t* sortingKey, Counters* cnt, float* outT, uint32_t* outTri, v2* outBary, int stage = -2, size_t rayIndex = 0)
{
    const Scene& s = *pt.scene;
    const GpuSettings& g = pt.st.Gpu;
    v2 packed = {wr.PackedDirectionX, wr.PackedDirectionY};
    v3 rayDir = DecodeUnitVec(packed);
    v3 origin = V3(wr.Origin[0], wr.Origin[1], wr.Origin[2]);
    v3 throughput = V3(wr.Throughput[0], wr.Throughput[1], wr.Throughput[2]);
    v3 radiance = V3(wr.Radiance[0], wr.Radiance[1], wr.Radiance[2]);
    HitInfo hit; float debugCost = 0.0f;
    Ray ray = {origin, rayDir};
    bool hitScene = TraceRay(s, ray, hit, debugCost, g.DoTraceLights != 0, pt.st.UseTlas != 0, REF_FLOAT_MAX, cnt);
    if (outT) { *outT = hit.T; *outTri = hit.TriangleId; *outBary = hit.bary; }
    auto store = [&]() { wr.Origin[0] = origin.x; wr.Origin[1] = origin.y; wr.Origin[2] = origin.z; wr.Throughput[0] = throughput.x; wr.Throughput[1] = throughput.y; wr.Throughput[2] = throughput.z;
                         wr.Radiance[0] = radiance.x; wr.Radiance[1] = radiance.y; wr.Radiance[2] = radiance.z; };
    if (first && g.DoDebugBVHTraversal) { wr.PreviousIOROrTraverseCost = debugCost; return false; } // FirstHit:108-112
    if (hitScene) {
        origin = origin + rayDir * hit.T;
        Surface surface = GetDefaultSurface();
        v3 geometricNormal = V3s(0.0f);
        bool hitLight = hit.TriangleId == ~0u;
        if (!hitLight) {
            if (sortingKey) *sortingKey = hit.TriangleId;
            const GpuBlasTriangle& tri = s.tris[hit.TriangleId];
            const GpuVertex& v0 = s.vertices[tri.X]; const GpuVertex& v1 = s.vertices[tri.Y]; const GpuVertex& v2_ = s.vertices[tri.Z];
            v3 bary = V3(hit.bary.x, hit.bary.y, 1.0f - hit.bary.x - hit.bary.y);
            v2 t0 = {v0.TexCoord[0], v0.TexCoord[1]}, t1 = {v1.TexCoord[0], v1.TexCoord[1]}, t2 = {v2_.TexCoord[0], v2_.TexCoord[1]};
            v2 uv = Interpolate2(t0, t1, t2, bary);
            if (stage == pt.uvStage) {                                                   // test hooks (struct PT)
                if (pt.uvDump) { pt.uvDump[2 * rayIndex] = uv.x; pt.uvDump[2 * rayIndex + 1] = uv.y; }
                if (pt.uvOverride && pt.uvOverride[2 * rayIndex] == pt.uvOverride[2 * rayIndex]) { uv.x = pt.uvOverride[2 * rayIndex]; uv.y = pt.uvOverride[2 * rayIndex + 1]; }
            }
            v3 interpNormal = normalize(Interpolate(DecompressSR11G11B10(v0.Normal), DecompressSR11G11B10(v1.Normal), DecompressSR11G11B10(v2_.Normal), bary));
            v3 interpTangent = normalize(Interpolate(DecompressSR11G11B10(v0.Tangent), DecompressSR11G11B10(v1.Tangent), DecompressSR11G11B10(v2_.Tangent), bary));
            const GpuMeshTransform& xf = s.xforms[hit.MeshTransformId];
            const GpuMesh& mesh = s.meshes[tri.MeshId];
            const GpuMaterial& mat = s.materials[mesh.MaterialId];
            surface = GetSurface(s, mat, uv);
            SurfaceApplyModificatons(surface, mesh);
            float alphaCutoff = (surface.AlphaCutoff == 2.0f) ? rnd01(rng) : surface.AlphaCutoff; // SurfaceHasAlphaBlending
            if (surface.Alpha < alphaCutoff) { origin = origin + rayDir * 0.001f; store(); return true; }
            v3 worldNormal = normalize(xform34_transposed3(xf.InvModel, interpNormal));
            v3 worldTangent = normalize(xform34_transposed3(xf.InvModel, interpTangent));
            v3 N = normalize(worldNormal), T = normalize(worldTangent), B = normalize(cross(N, T)); // Math.glsl:130-137 GetTBN
            v3 sn = surface.Normal;
            v3 tn = V3((T.x * sn.x + B.x * sn.y) + N.x * sn.z, (T.y * sn.x + B.y * sn.y) + N.y * sn.z, (T.z * sn.x + B.z * sn.y) + N.z * sn.z); // mat3(T,B,N) * n
            surface.Normal = normalize(gmix(worldNormal, tn, mesh.NormalMapStrength));
            geometricNormal = GetTriangleNormal(P(s, tri.X), P(s, tri.Y), P(s, tri.Z));
            geometricNormal = normalize(xform34_transposed3(xf.InvModel, geometricNormal));
        } else if (g.DoTraceLights) {
            if (sortingKey) *sortingKey = hit.MeshTransformId;
            const GpuLight& l = s.lights[hit.MeshTransformId];
            surface.Emissive = V3(l.Color[0], l.Color[1], l.Color[2]); surface.Albedo = surface.Emissive;
            surface.Normal = (origin - V3(l.Position[0], l.Position[1], l.Position[2])) / l.Radius;
            geometricNormal = surface.Normal;
        }
        float prevIor = first ? 1.0f : wr.PreviousIOROrTraverseCost;
        bool fromInside = dot(-rayDir, geometricNormal) < 0.0f;
        if (fromInside) {
            if (first) prevIor = surface.IOR;
            geometricNormal = geometricNormal * -1.0f;
            if (surface.IsVolumetric) throughput = throughput * gexp3(-surface.Absorbance * hit.T);
        }
        float cosTheta = dot(-rayDir, surface.Normal);
        if (cosTheta < 0.0f) { surface.Normal = surface.Normal * -1.0f; cosTheta *= -1.0f; }
        radiance = radiance + surface.Emissive * throughput;
        SampleMaterialResult result = SampleMaterial(rayDir, surface, prevIor, fromInside, rng, gidSeed, pt.sampleIndex());
        throughput = throughput * (result.Bsdf / result.Pdf);
        {
            float weight = GetSurfaceVariance(surface.Metallic, surface.Transmission, surface.Roughness);
            if (first) { v3 a = surface.Albedo * weight, n = surface.Normal * weight; ar.Albedo[0] = a.x; ar.Albedo[1] = a.y; ar.Albedo[2] = a.z; ar.Normal[0] = n.x; ar.Normal[1] = n.y; ar.Normal[2] = n.z; ar.NewWeight = 1.0f - weight; }
            else {
                v3 a = V3(ar.Albedo[0], ar.Albedo[1], ar.Albedo[2]) + ar.NewWeight * surface.Albedo * weight;
                v3 n = V3(ar.Normal[0], ar.Normal[1], ar.Normal[2]) + ar.NewWeight * surface.Normal * weight;
                ar.Albedo[0] = a.x; ar.Albedo[1] = a.y; ar.Albedo[2] = a.z; ar.Normal[0] = n.x; ar.Normal[1] = n.y; ar.Normal[2] = n.z; ar.NewWeight *= (1.0f - weight);
            }
        }
        if (!first) { // NHit:188-192
            bool terminate = g.DoRussianRoulette && RussianRouletteTerminateRay(throughput, rng);
            if (terminate) { store(); return false; }
        }
        if (result.BsdfType == BSDF_TRANSMISSIVE) geometricNormal = geometricNormal * -1.0f;
        origin = origin + geometricNormal * 0.001f;
        wr.PreviousIOROrTraverseCost = result.NewIor;
        v2 pd = EncodeUnitVec(result.RayDirection);
        wr.PackedDirectionX = pd.x; wr.PackedDirectionY = pd.y;
        store();
        return true;
    } else {
        v3 albedo = SampleSky(s, rayDir);
        v3 fn = CubemapFaceNormal(rayDir);
        if (first) { ar.Albedo[0] = albedo.x; ar.Albedo[1] = albedo.y; ar.Albedo[2] = albedo.z; ar.Normal[0] = fn.x; ar.Normal[1] = fn.y; ar.Normal[2] = fn.z; }
        else {
            v3 a = V3(ar.Albedo[0], ar.Albedo[1], ar.Albedo[2]) + ar.NewWeight * albedo; v3 n = V3(ar.Normal[0], ar.Normal[1], ar.Normal[2]) + ar.NewWeight * fn;
            ar.Albedo[0] = a.x; ar.Albedo[1] = a.y; ar.Albedo[2] = a.z; ar.Normal[0] = n.x; ar.Normal[1] = n.y; ar.Normal[2] = n.z;
        }
        ar.NewWeight = 0.0f;
        radiance = radiance + albedo * throughput;
        store();
        return false;
    }
}

static v3 TurboColormap(float x) // FinalDraw/compute.glsl:64-82
{
    x = gclamp(x, 0.0f, 1.0f);
    float v4[4] = {1.0f, x, x * x, x * x * x};
    float v2_[2] = {v4[2] * v4[2], v4[3] * v4[2]};
    auto dot4 = [&](const float* k) { return ((v4[0] * k[0] + v4[1] * k[1]) + v4[2] * k[2]) + v4[3] * k[3]; };
    auto dot2_ = [&](const float* k) { return v2_[0] * k[0] + v2_[1] * k[1]; };
    const float kR4[4] = {0.13572138f, 4.61539260f, -42.66032258f, 132.13108234f}, kG4[4] = {0.09140261f, 2.19418839f, 4.84296658f, -14.18503333f}, kB4[4] = {0.10667330f, 12.64194608f, -60.58204836f, 110.36276771f};
    const float kR2[2] = {-152.94239396f, 59.28637943f}, kG2[2] = {4.27729857f, 2.82956604f}, kB2[2] = {-89.90310912f, 27.34824973f};
    return V3(dot4(kR4) + dot2_(kR2), dot4(kG4) + dot2_(kG2), dot4(kB4) + dot2_(kB2));
}

static void RenderSample(PT& pt)
{
    const int W = pt.W, rows = pt.rows; const size_t N = (size_t)W * rows;
    const GpuSettings& g = pt.st.Gpu;
    std::vector<uint8_t> cont(N);
    std::vector<Counters> rowCnt(rows);
    memset(pt.aliveCounts, 0, sizeof(pt.aliveCounts));
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point tStart = clk::now();
    clk::time_point tp = tStart;
    // ---- FirstHit main (FirstHit/compute.glsl:44-98), one invocation per pixel ----
    #pragma omp parallel for schedule(dynamic, 1)
    for (int ly = 0; ly < rows; ly++) {
        int y = ((ly / pt.rowBand) * pt.rowMod + pt.rowRem) * pt.rowBand + ly % pt.rowBand;   // idkptSetRowBands (rowBand 1: y % rowMod == rowRem, or a strip)
        for (int x = 0; x < W; x++) {
            size_t rayIndex = (size_t)ly * W + x;
            Rng rng; rng.seed = (uint32_t)(y * 4096 + x) * (pt.sampleIndex() + 1u);
            float ox = rnd01(&rng), oy = rnd01(&rng);
            v2 ndc = {((float)x + ox) / (float)W * 2.0f - 1.0f, ((float)y + oy) / (float)pt.H * 2.0f - 1.0f};
            v3 camDir = GetWorldSpaceDirection(pt.invProj, pt.invView, ndc);
            v3 viewPos = V3(pt.viewPos[0], pt.viewPos[1], pt.viewPos[2]);
            v3 focalPoint = viewPos + camDir * g.FocalLength;
            v2 disk = SampleDisk(&rng);
            v3 pointOnLense = mat4_mul_xyz(pt.invView, g.LenseRadius * disk.x, g.LenseRadius * disk.y, 0.0f, 1.0f);
            camDir = normalize(focalPoint - pointOnLense);
            GpuWavefrontRay wr; GpuAovRay ar;
            wr.Origin[0] = pointOnLense.x; wr.Origin[1] = pointOnLense.y; wr.Origin[2] = pointOnLense.z;
            v2 pd = EncodeUnitVec(camDir); wr.PackedDirectionX = pd.x; wr.PackedDirectionY = pd.y;
            wr.Throughput[0] = wr.Throughput[1] = wr.Throughput[2] = 1.0f; wr.Radiance[0] = wr.Radiance[1] = wr.Radiance[2] = 0.0f; wr.PreviousIOROrTraverseCost = 1.0f;
            ar.Albedo[0] = ar.Albedo[1] = ar.Albedo[2] = 0.0f; ar.Normal[0] = ar.Normal[1] = ar.Normal[2] = 0.0f; ar.NewWeight = 1.0f; ar._pad0 = 0.0f;
            uint32_t gx, gy; FirstHitGid(W, pt.H, x, y, &gx, &gy);
            v2 pb;
            bool c = ShadeRay(pt, true, wr, ar, &rng, gy * 4096u + gx, nullptr, pt.countersOn ? &rowCnt[ly] : nullptr, &pt.primT[rayIndex], &pt.primTri[rayIndex], &pb, 0, rayIndex);
            pt.primBary[2 * rayIndex] = pb.x; pt.primBary[2 * rayIndex + 1] = pb.y;
            pt.rays[rayIndex] = wr;
            if (pt.st.OutputAOVs) pt.aov[rayIndex] = ar;
            cont[rayIndex] = c;
        }
    }
    pt.raysTraced += N;
    pt.parallelSec += secs(tp, clk::now());
    // canonical enqueue: increasing pixel index (FirstHit:88-97)
    pt.alive.clear();
    for (size_t i = 0; i < N; i++) if (cont[i]) pt.alive.push_back((uint32_t)i);
    for (int j = 1; j < pt.st.RayDepth; j++) {
        size_t A = pt.alive.size();
        if (j < 16) pt.aliveCounts[j] = (uint32_t)A;
        // RaySorting() (PathTracer.cs:232-237,273-297): stable counting sort on the 21-bit key cached by NHit j-1
        if (pt.st.DoRaySorting && j > 1) {
            std::vector<uint32_t> order(A);
            for (size_t i = 0; i < A; i++) order[i] = (uint32_t)i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pt.keys[a] < pt.keys[b]; });
            std::vector<uint32_t> sorted(A);
            for (size_t i = 0; i < A; i++) sorted[i] = pt.alive[order[i]];
            pt.alive.swap(sorted);
        }
        // exact multi-context mode: global slot = local slot + alive rays of the contexts that own earlier rows (include/idkpt.h)
        uint32_t slotBase = 0;
        if (pt.exchangeFn) { uint32_t localCount = (uint32_t)A; pt.exchangeFn(pt.exchangeUser, j, 1, &localCount, &slotBase); }
        // interleaved rows / bands (idkptSetBandExchange): the queue is in local pixel order, so the rays of one local band are a contiguous run of slots; the host returns,
        // per band, the alive rays of ALL contexts in the image bands before it: global slot = that + the position inside the run
        std::vector<uint32_t> bandStart, bandBase;
        const bool banded = pt.bandExchangeFn && pt.rowMod > 1 && !(pt.st.DoRaySorting && j > 1);
        if (banded) {
            const int LB = (pt.rows + pt.rowBand - 1) / pt.rowBand;
            std::vector<uint32_t> cnt((size_t)LB, 0u); bandStart.assign((size_t)LB + 1, 0u); bandBase.assign((size_t)LB, 0u);
            for (size_t i = 0; i < A; i++) cnt[(size_t)((pt.alive[i] / (uint32_t)pt.W) / (uint32_t)pt.rowBand)]++;
            for (int b = 0; b < LB; b++) bandStart[(size_t)b + 1] = bandStart[(size_t)b] + cnt[(size_t)b];
            pt.bandExchangeFn(pt.bandExchangeUser, j, 1, LB, cnt.data(), bandBase.data());
        }
        std::vector<uint8_t> cont2(A); std::vector<uint32_t> keyOut(A, 0u);
        std::vector<Counters> chunkCnt((A + 255) / 256);
        // ---- NHit main (NHit/compute.glsl:40-89), one invocation per queue slot ----
        tp = clk::now();
        #pragma omp parallel for schedule(dynamic, 1)
        for (long long chunk = 0; chunk < (long long)((A + 255) / 256); chunk++) {
            for (size_t slot = (size_t)chunk * 256; slot < std::min(A, (size_t)(chunk + 1) * 256); slot++) {
                uint32_t rayIndex = pt.alive[slot];
                uint32_t gslot = slotBase + (uint32_t)slot;
                if (banded) { const size_t lb = (size_t)((rayIndex / (uint32_t)pt.W) / (uint32_t)pt.rowBand); gslot = bandBase[lb] + ((uint32_t)slot - bandStart[lb]); }
                Rng rng; rng.seed = gslot * 4096u + pt.sampleIndex();
                GpuWavefrontRay wr = pt.rays[rayIndex]; GpuAovRay ar = pt.aov[rayIndex];
                uint32_t key = 0;
                bool c = ShadeRay(pt, false, wr, ar, &rng, gslot, &key, pt.countersOn ? &chunkCnt[chunk] : nullptr, nullptr, nullptr, nullptr, j, (size_t)rayIndex);
                pt.rays[rayIndex] = wr;
                if (pt.st.OutputAOVs) pt.aov[rayIndex] = ar;
                cont2[slot] = c; keyOut[slot] = key & ((1u << IDKPT_SORT_KEY_BITS) - 1u);
            }
        }
        pt.parallelSec += secs(tp, clk::now());
        for (auto& c : chunkCnt) { pt.counters.pairs += c.pairs; pt.counters.tris += c.tris; }
        pt.raysTraced += A;
        // canonical enqueue: increasing old slot (NHit:69-88)
        std::vector<uint32_t> next; next.reserve(A); pt.keys.clear();
        for (size_t slot = 0; slot < A; slot++) if (cont2[slot]) { next.push_back(pt.alive[slot]); pt.keys.push_back(keyOut[slot]); }
        pt.alive.swap(next);
    }
    for (auto& c : rowCnt) { pt.counters.pairs += c.pairs; pt.counters.tris += c.tris; }
    // ---- FinalDraw (FinalDraw/compute.glsl:24-62) ----
    float w = 1.0f / ((float)pt.accumulated + 1.0f);
    tp = clk::now();
    #pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)N; i++) {          // one invocation per pixel, independent
        const GpuWavefrontRay& wr = pt.rays[i];
        v3 nr = V3(wr.Radiance[0], wr.Radiance[1], wr.Radiance[2]);
        if (g.DoDebugBVHTraversal) nr = TurboColormap(wr.PreviousIOROrTraverseCost / 150.0f);
        float* o = &pt.img[0][4 * i];
        v3 r = gmix(V3(o[0], o[1], o[2]), nr, w); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = 1.0f;
        if (pt.st.OutputAOVs) {
            const GpuAovRay& a = pt.aov[i];
            float* oa = &pt.img[1][4 * i]; v3 ra = gmix(V3(oa[0], oa[1], oa[2]), V3(a.Albedo[0], a.Albedo[1], a.Albedo[2]), w); oa[0] = ra.x; oa[1] = ra.y; oa[2] = ra.z; oa[3] = 1.0f;
            float* on = &pt.img[2][4 * i]; v3 rn = gmix(V3(on[0], on[1], on[2]), V3(a.Normal[0], a.Normal[1], a.Normal[2]), w); on[0] = rn.x; on[1] = rn.y; on[2] = rn.z; on[3] = 1.0f;
        }
    }
    pt.parallelSec += secs(tp, clk::now());
    pt.totalSec += secs(tStart, clk::now());
    pt.accumulated++;
}

} // namespace

// ---- Shaders/ShadowsRayTraced/compute.glsl:19-127 (one point shadow) --------------------------------------------------------
static inline float InterleavedGradientNoise(float cx, float cy, uint32_t index) // Random.glsl:35-41
{
    float add = (float)index * 5.588238f;
    cx = cx + add; cy = cy + add;
    return gfract(52.9829189f * gfract(0.06711056f * cx + 0.00583715f * cy));
}
static inline v3 ConstructBasisMul(v3 normal, v3 local) // Math.glsl:112-127: mat3(tangent, normal, bitangent) * local
{
    v3 up = gabs(normal.z) < 0.999f ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
    v3 tangent = normalize(cross(up, normal));
    v3 bitangent = cross(normal, tangent);
    return V3((tangent.x * local.x + normal.x * local.y) + bitangent.x * local.z,
              (tangent.y * local.x + normal.y * local.y) + bitangent.y * local.z,
              (tangent.z * local.x + normal.z * local.y) + bitangent.z * local.z);
}
static inline v3 SampleSphereCone(v3 toSphere, float sphereRadius, float rnd0, float rnd1, float* distanceToSphere) // Sampling.glsl:21-52
{
    const float PI = 3.14159265f;
    float radiusSq = sphereRadius * sphereRadius;
    float distanceSq = dot(toSphere, toSphere);
    float sinThetaMaxSq = radiusSq / distanceSq;
    float cosThetaMax = gsqrt(gmax(1.0f - sinThetaMaxSq, 0.0f));
    float phiMax = 2.0f * PI;
    float phi = phiMax * rnd0;
    float cosTheta = gmix(cosThetaMax, 1.0f, gmax(rnd1, 0.001f));
    float sinTheta = gsqrt(gmax(1.0f - cosTheta * cosTheta, 0.0f));
    *distanceToSphere = length(toSphere) * cosTheta - gsqrt(radiusSq - distanceSq * sinTheta * sinTheta);
    float sp, cp; gsincos(phi, &sp, &cp);
    v3 local = V3(cp * sinTheta, cosTheta, sp * sinTheta);
    return ConstructBasisMul(normalize(toSphere), local);
}
static void ShadowPixel(const Scene& s, bool useTlas, const idkpt_shadow_params& p, int x, int y, const float* depthImg, const float* normalImg, float* vis)
{
    size_t pix = (size_t)y * p.Width + x;
    uint32_t noiseIndex = p.NoiseIndex;
    Rng rng = {0u};                                        // un-seeded in the reference (:24 commented out); defined as 0 here
    float depth = depthImg[pix];
    if (depth == 1.0f) return;                             // :28-32 (bounds are the launch bounds here)
    const GpuLight& light = s.lights[p.LightIndex];
    v3 lightPos = V3(light.Position[0], light.Position[1], light.Position[2]);
    float u = ((float)x + 0.5f) / (float)p.Width, v = ((float)y + 0.5f) / (float)p.Height;
    float nx = (u * 2.0f - 1.0f) - p.TaaJitter[0], ny = (v * 2.0f - 1.0f) - p.TaaJitter[1];
    const float* m = p.InvProjView;                        // PerspectiveTransform (Math.glsl:75-79)
    v3 wp = mat4_mul_xyz(m, nx, ny, depth, 1.0f);
    float ww = ((m[3] * nx + m[7] * ny) + m[11] * depth) + m[15] * 1.0f;
    v3 fragPos = wp / ww;
    v2 nrg = {normalImg[2 * pix], normalImg[2 * pix + 1]};
    v3 normal = DecodeUnitVec(nrg);
    float cosTheta = dot(normal, normalize(lightPos - fragPos));
    if (cosTheta <= 0.0f) { vis[pix] = 0.0f; return; }    // :44-49
    float visibility = 0.0f;
    for (int i = 0; i < p.RayTracingSamples; i++) {
        v3 biased = fragPos + normal * 0.01f;
        float rnd0 = InterleavedGradientNoise((float)x, (float)y, noiseIndex + 0u);
        float rnd1 = InterleavedGradientNoise((float)x, (float)y, noiseIndex + 1u);
        noiseIndex++;
        v3 fragToLight = lightPos - biased;
        float distanceToLight;
        v3 direction = SampleSphereCone(fragToLight, light.Radius, rnd0, rnd1, &distanceToLight);
        Ray ray; ray.o = biased; ray.d = direction;
        HitInfo hit; float cost;
        float thisVisibility = 1.0f;
        while (TraceRay(s, ray, hit, cost, true, useTlas, distanceToLight - 0.001f, nullptr)) { // :73
            if (hit.TriangleId == ~0u) {                   // :75-87
                if (hit.MeshTransformId != (uint32_t)p.LightIndex) thisVisibility = 0.0f;
                break;
            }
            const GpuBlasTriangle& t = s.tris[hit.TriangleId];
            v3 bary = V3(hit.bary.x, hit.bary.y, 1.0f - hit.bary.x - hit.bary.y);
            const GpuVertex& v0 = s.vertices[t.X]; const GpuVertex& v1 = s.vertices[t.Y]; const GpuVertex& v2_ = s.vertices[t.Z];
            v2 t0 = {v0.TexCoord[0], v0.TexCoord[1]}, t1 = {v1.TexCoord[0], v1.TexCoord[1]}, t2 = {v2_.TexCoord[0], v2_.TexCoord[1]};
            v2 uv = Interpolate2(t0, t1, t2, bary);
            const GpuMesh& mesh = s.meshes[t.MeshId];
            const GpuMaterial& mat = s.materials[mesh.MaterialId];
            Surface surface = GetSurface(s, mat, uv);
            SurfaceApplyModificatons(surface, mesh);
            bool blend = surface.AlphaCutoff == 2.0f;      // SurfaceHasAlphaBlending (Surface.glsl:93-96)
            float alphaCutoff = blend ? rnd01(&rng) : surface.AlphaCutoff;
            if (blend) thisVisibility *= 1.0f - surface.Alpha;
            else if (surface.Alpha > alphaCutoff) thisVisibility = 0.0f;
            if (thisVisibility < 0.01f) break;
            float dist = hit.T + 0.001f;
            ray.o = ray.o + ray.d * dist;
            distanceToLight -= dist;
        }
        visibility += thisVisibility;
    }
    visibility /= (float)p.RayTracingSamples;
    vis[pix] = visibility;
}

extern "C" {

void* ref_scene_create(const idkpt_scene_desc* d)
{
    Scene* s = new Scene();
    s->nodes.assign(d->BlasNodes, d->BlasNodes + d->BlasNodeCount);
    s->tris.assign(d->BlasTriangles, d->BlasTriangles + d->BlasTriangleCount);
    s->descs.assign(d->BlasDescs, d->BlasDescs + d->BlasDescCount);
    s->instances.assign(d->BlasInstances, d->BlasInstances + d->BlasInstanceCount);
    if (d->TlasNodes) s->tlas.assign(d->TlasNodes, d->TlasNodes + d->TlasNodeCount);
    s->positions.assign(d->VertexPositions, d->VertexPositions + 3 * (size_t)d->VertexCount);
    s->vertices.assign(d->Vertices, d->Vertices + d->VertexCount);
    s->meshes.assign(d->Meshes, d->Meshes + d->MeshCount);
    s->materials.assign(d->Materials, d->Materials + d->MaterialCount);
    s->xforms.assign(d->MeshTransforms, d->MeshTransforms + d->MeshTransformCount);
    if (d->Lights) s->lights.assign(d->Lights, d->Lights + d->LightCount);
    if (d->SkyFaces && d->SkyFaceSize > 0) { s->skySize = d->SkyFaceSize; s->sky.assign(d->SkyFaces, d->SkyFaces + 6 * 4 * (size_t)d->SkyFaceSize * d->SkyFaceSize); }
    for (int i = 0; i < d->TextureCount; i++) s->textures.push_back(load_texture(d->Textures[i]));
    return s;
}
// test hook: one tap of the texture unit, outside any scene (tests/test_oracle_kats.py: wrap modes / filters / decodes against an independent statement of GL 4.6 8.14.2)
void ref_sample_texture(const idkpt_texture* t, const float* uv, int count, float* outRgba)
{
    Scene s; s.textures.push_back(load_texture(*t));
    for (int i = 0; i < count; i++) { v2 q = {uv[2 * i], uv[2 * i + 1]}; const RGBA c = SampleTex(s, 1, q); outRgba[4 * i] = c.r; outRgba[4 * i + 1] = c.g; outRgba[4 * i + 2] = c.b; outRgba[4 * i + 3] = c.a; }
}
void ref_scene_set_texture(void* s, int index, const idkpt_texture* t) { ((Scene*)s)->textures[(size_t)index] = load_texture(*t); }    // checker for idkptUpdateTexture
void ref_scene_destroy(void* s) { delete (Scene*)s; }
void ref_scene_set_positions(void* s, const float* positions, int vertexCount) { ((Scene*)s)->positions.assign(positions, positions + 3 * (size_t)vertexCount); }
void ref_scene_set_blas_nodes(void* s, const GpuBlasNode* nodes, int count) { ((Scene*)s)->nodes.assign(nodes, nodes + count); }

// Batched ray queries (TraceRay / TraceRayAny with explicit maxDist and traceLights) — checker for idkptTraceRays
void ref_trace_rays(void* scene, int useTlas, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits)
{
    const Scene& s = *(Scene*)scene;
    #pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)count; i++) {
        Ray r; r.o = V3(rays[i].Origin[0], rays[i].Origin[1], rays[i].Origin[2]); r.d = V3(rays[i].Direction[0], rays[i].Direction[1], rays[i].Direction[2]);
        HitInfo h; float cost; bool hit;
        if (flags & IDKPT_TRACE_ANY_HIT) hit = TraceRayAny(s, r, h, (flags & IDKPT_TRACE_LIGHTS) != 0, useTlas != 0, rays[i].MaxDist);
        else hit = TraceRay(s, r, h, cost, (flags & IDKPT_TRACE_LIGHTS) != 0, useTlas != 0, rays[i].MaxDist, nullptr);
        idkpt_hit o; memset(&o, 0, sizeof(o));
        o.T = h.T; o.BaryX = h.bary.x; o.BaryY = h.bary.y; o.TriangleId = h.TriangleId; o.MeshTransformId = h.MeshTransformId; o.Hit = hit ? 1u : 0u;
        hits[i] = o;
    }
}
// Checker for idkptTraceShadows
void ref_trace_shadows(void* scene, int useTlas, const idkpt_shadow_params* p, const float* depth, const float* normalOct, float* visibility)
{
    const Scene& s = *(Scene*)scene;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < p->Height; y++) for (int x = 0; x < p->Width; x++) ShadowPixel(s, useTlas != 0, *p, x, y, depth, normalOct, visibility);
}

void* ref_pt_create(void* scene, int w, int h, int rowMod, int rowRem)
{
    PT* pt = new PT(); pt->scene = (Scene*)scene; pt->W = w; pt->H = h; pt->rowMod = rowMod; pt->rowRem = rowRem;
    pt->rows = 0; for (int y = rowRem; y < h; y += rowMod) pt->rows++;
    size_t N = (size_t)w * pt->rows;
    pt->rays.assign(N, GpuWavefrontRay{}); pt->aov.assign(N, GpuAovRay{});
    for (int i = 0; i < 3; i++) pt->img[i].assign(4 * N, 0.0f);
    pt->primT.assign(N, 0.0f); pt->primTri.assign(N, 0u); pt->primBary.assign(2 * N, 0.0f);
    memset(&pt->st, 0, sizeof(pt->st));
    pt->st.Gpu.FocalLength = 8.0f; pt->st.Gpu.DoRussianRoulette = 1; pt->st.RayDepth = 7; pt->st.SamplesPerPixel = 1;
    return pt;
}
void ref_pt_destroy(void* p) { delete (PT*)p; }
// idkptSetRowRange / idkptSetBounceExchange counterparts
void ref_pt_set_row_range(void* p, int firstRow, int rowCount)
{
    PT* pt = (PT*)p; pt->rowMod = 1; pt->rowRem = firstRow; pt->rowBand = 1; pt->rows = std::min(rowCount, pt->H - firstRow);
    size_t N = (size_t)pt->W * pt->rows;
    pt->rays.assign(N, GpuWavefrontRay{}); pt->aov.assign(N, GpuAovRay{});
    for (int i = 0; i < 3; i++) pt->img[i].assign(4 * N, 0.0f);
    pt->primT.assign(N, 0.0f); pt->primTri.assign(N, 0u); pt->primBary.assign(2 * N, 0.0f);
    pt->accumulated = 0;
}
// idkptSetRowBands counterpart: rows y with (y / bandRows) % rowMod == rowRem
void ref_pt_set_row_bands(void* p, int bandRows, int rowMod, int rowRem)
{
    PT* pt = (PT*)p; pt->rowBand = rowMod > 1 ? bandRows : 1; pt->rowMod = rowMod; pt->rowRem = rowRem;
    pt->rows = 0; for (int y = 0; y < pt->H; y++) if ((y / pt->rowBand) % rowMod == rowRem) pt->rows++;
    size_t N = (size_t)pt->W * pt->rows;
    pt->rays.assign(N, GpuWavefrontRay{}); pt->aov.assign(N, GpuAovRay{});
    for (int i = 0; i < 3; i++) pt->img[i].assign(4 * N, 0.0f);
    pt->primT.assign(N, 0.0f); pt->primTri.assign(N, 0u); pt->primBary.assign(2 * N, 0.0f);
    pt->accumulated = 0;
}
void ref_pt_set_bounce_exchange(void* p, idkpt_bounce_exchange_fn fn, void* user) { PT* pt = (PT*)p; pt->exchangeFn = fn; pt->exchangeUser = user; }
void ref_pt_set_band_exchange(void* p, idkpt_band_exchange_fn fn, void* user) { PT* pt = (PT*)p; pt->bandExchangeFn = fn; pt->bandExchangeUser = user; }
void ref_pt_set_settings(void* p, const idkpt_settings* s) { ((PT*)p)->st = *s; }
void ref_pt_set_perframe(void* p, const float* invProj, const float* invView, const float* viewPos) { PT* pt = (PT*)p; memcpy(pt->invProj, invProj, 64); memcpy(pt->invView, invView, 64); memcpy(pt->viewPos, viewPos, 12); }
void ref_pt_reset_accumulation(void* p) { ((PT*)p)->accumulated = 0; }
void ref_pt_set_uv_hooks(void* p, int stage, const float* overrideUv, float* dumpUv) { PT* pt = (PT*)p; pt->uvStage = stage; pt->uvOverride = overrideUv; pt->uvDump = dumpUv; }   // test hooks (struct PT)
void ref_pt_set_sample_sequence(void* p, uint32_t first, uint32_t stride) { PT* pt = (PT*)p; pt->seqFirst = first; pt->seqStride = stride; pt->accumulated = 0; }
void ref_pt_enable_counters(void* p, int on) { ((PT*)p)->countersOn = on != 0; }
void ref_pt_render(void* p) { PT* pt = (PT*)p; for (int i = 0; i < pt->st.SamplesPerPixel; i++) RenderSample(*pt); } // PathTracer.cs:218
void ref_pt_get_image(void* p, int which, float* out) { PT* pt = (PT*)p; memcpy(out, pt->img[which].data(), pt->img[which].size() * 4); }
void ref_pt_get_rays(void* p, GpuWavefrontRay* out) { PT* pt = (PT*)p; memcpy(out, pt->rays.data(), pt->rays.size() * sizeof(GpuWavefrontRay)); }
void ref_pt_get_primary_hits(void* p, float* t, uint32_t* tri, float* bary) { PT* pt = (PT*)p; size_t N = pt->primT.size(); memcpy(t, pt->primT.data(), 4 * N); memcpy(tri, pt->primTri.data(), 4 * N); memcpy(bary, pt->primBary.data(), 8 * N); }
uint32_t ref_pt_get_alive(void* p, uint32_t* out, uint32_t cap) { PT* pt = (PT*)p; uint32_t n = (uint32_t)pt->alive.size(); if (out) memcpy(out, pt->alive.data(), 4 * (size_t)std::min(n, cap)); return n; }
// sort keys cached by the last NHit for the alive queue, entry by entry (NHit/compute.glsl:80-85; empty after FirstHit) — the checker forces them as part of a stage's input state
uint32_t ref_pt_get_alive_keys(void* p, uint32_t* out, uint32_t cap) { PT* pt = (PT*)p; uint32_t n = (uint32_t)pt->keys.size(); if (out) memcpy(out, pt->keys.data(), 4 * (size_t)std::min(n, cap)); return n; }
void ref_pt_get_stats(void* p, uint64_t* raysTraced, uint64_t* pairs, uint64_t* tris, uint32_t* aliveCounts16) { PT* pt = (PT*)p; *raysTraced = pt->raysTraced; *pairs = pt->counters.pairs; *tris = pt->counters.tris; memcpy(aliveCounts16, pt->aliveCounts, 64); }
uint32_t ref_pt_accumulated(void* p) { return ((PT*)p)->accumulated; }
void ref_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }   // cpu_baseline leg of bench.py: pick the thread count the host actually scales to
int ref_get_max_threads(void) { return omp_get_max_threads(); }
void ref_pt_get_timing(void* p, double* parallelSec, double* totalSec, int reset) { PT* pt = (PT*)p; *parallelSec = pt->parallelSec; *totalSec = pt->totalSec; if (reset) { pt->parallelSec = 0.0; pt->totalSec = 0.0; } }

// ---- KAT helpers (tests/test_oracle_kats.py) ----
uint32_t ref_pcg_hash(uint32_t* seed) { return pcg_hash(seed); }
void ref_encode_unit_vec(const float* n, float* out2) { v2 r = EncodeUnitVec(V3(n[0], n[1], n[2])); out2[0] = r.x; out2[1] = r.y; }
void ref_decode_unit_vec(const float* f, float* out3) { v2 a = {f[0], f[1]}; v3 r = DecodeUnitVec(a); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; }
uint32_t ref_compress_sr11g11b10(const float* v) { return CompressSR11G11B10(V3(v[0], v[1], v[2])); }
void ref_decompress_sr11g11b10(uint32_t d, float* out3) { v3 r = DecompressSR11G11B10(d); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; }
void ref_sincos(float x, float* s, float* c) { gsincos(x, s, c); }
float ref_exp(float x) { return gexp(x); }
void ref_turbo(float x, float* out3) { v3 r = TurboColormap(x); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; }
void ref_first_hit_gid(int W, int H, int px, int py, uint32_t* gx, uint32_t* gy) { FirstHitGid(W, H, px, py, gx, gy); }
void ref_r2_sequence(uint32_t id, float* out2) { v2 r = R2Sequence(id); out2[0] = r.x; out2[1] = r.y; }
int ref_ray_triangle(const float* o, const float* d, const float* p0, const float* p1, const float* p2, float* bary3, float* t)
{ Ray r = {V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])}; v3 b; bool h = RayTriangleIntersect(r, V3(p0[0], p0[1], p0[2]), V3(p1[0], p1[1], p1[2]), V3(p2[0], p2[1], p2[2]), &b, t); bary3[0] = b.x; bary3[1] = b.y; bary3[2] = b.z; return h; }
int ref_ray_box(const float* o, const float* d, const float* bmin, const float* bmax, float* t1)
{ v3 inv = V3(1.0f / d[0], 1.0f / d[1], 1.0f / d[2]); return RayBoxIntersect(V3(o[0], o[1], o[2]), inv, bmin, bmax, t1); }

} // extern "C"
