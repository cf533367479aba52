// pt_kernels.hpp — gfx950 kernels of the wavefront path tracer (device code only; host API in idkpt.hip).
//
// Replaces the reference's GLSL compute pipeline (paths relative to /root/reference/IDKEngine/Resource/Shaders):
//   PathTracing/FirstHit/compute.glsl, PathTracing/NHit/compute.glsl, PathTracing/FinalDraw/compute.glsl,
//   PathTracing/CountingSort/**, include/BVHIntersect.glsl.
// Structure (DESIGN.md "Kernels"): trace and shade are separate kernels so that the pointer-chasing traversal runs at
// the occupancy its ~48 VGPRs allow and can be scheduled as persistent waves pulling 64-ray packets; the alive queue is
// compacted with wave64 ballots + an ordered scan, which reproduces the canonical (sequential) enqueue order the
// oracle defines for the reference's atomicAdd slots.
#pragma once
#include "pt_device.hpp"

namespace ptd {

// one image of the texture table: state = wrapS | wrapT << 2 | magFilter << 4 | format << 5 (enum idkpt_wrap / idkpt_filter / idkpt_texture_format, include/idkpt.h)
struct TexDesc { const void* data; int w, h; uint32_t state; uint32_t pad; };

struct DScene {
    const float4* nodes;        // 2 x float4 per GpuBlasNode: {Min.xyz, TriStartOrChild}, {Max.xyz, TriCount}
    const float4* pairNodes;    // derived (k_pair_nodes, kernels_scene.hpp), or null: the same sibling pairs, 64 bytes each at the same index, fields regrouped for 2-wide arithmetic —
                                // {lmin.xy, rmin.xy}, {lmax.xy, rmax.xy}, {lmin.z, rmin.z, lmax.z, rmax.z}, {lStart, lCount, rStart, rCount}: k_trace2's FAST node step (kernels_trace.hpp)
    const uint4* tris;          // GpuBlasTriangle
    const float4* triVerts;     // derived: 3 x float4 per BLAS triangle (leaf order): positions of X,Y,Z (w unused)
    const GpuBlasDesc* descs;
    const GpuBlasInstance* instances; int instanceCount;
    const float4* tlas; int tlasCount;
    const float4* instRec;      // derived (k_inst_records), or null: 6 x float4 per BLAS instance — InvModel rows 0-2, {root Min.xyz, NodeOffset}, {root Max.xyz, TriangleOffset}, {MeshTransformId, BlasId, 0, 0}:
                                // what a leaf of the library's own TLAS needs to enter an instance in ONE round trip instead of three dependent ones (instance -> desc + transform -> root node;
                                // kernels_trace_inst.hpp: +5 %.  The instance loop itself is 4-14 % slower with it — kernels_trace.hpp — and keeps its three fetches)
    const uint4* vertices;      // GpuVertex {uv.x, uv.y, tangent, normal} as raw dwords
    const GpuMesh* meshes;
    const GpuMaterial* materials;
    const float4* xforms;       // GpuMeshTransform as 9 x float4 (Model rows 0-2, InvModel rows 3-5, Prev rows 6-8)
    const GpuLight* lights; int lightCount;
    const float4* sky; int skySize;
    const TexDesc* textures; int textureCount;
    const float* srgbLut;       // 256 floats: the sRGB -> linear transfer function of GL 4.6 8.24 per byte value (IDKPT_TEXFMT_SRGB8_A8 texels are decoded before filtering)
    uint32_t* overflow;         // host-mapped word: set when a traversal-stack push had to be dropped (idkpt.hip turns it into an error at the next sync)
    // scene versions (idkptSetSceneVersions): samples of one batch may see different states of the geometry (animated frames in flight).  Null: every sample
    // reads the pointers above.  Else the pointers above are the bases of the version arenas and row smp of this table holds, in 16-byte units, where the
    // version of sample smp starts in each: [0] nodes [1] triVerts [2] vertices [3] tlas [4] xforms [5..7] unused (kernels instantiated with VER)
    const uint32_t* ver;
};
#define SCENE_VER_WORDS 8
// the scene as sample smp of the batch sees it (VER instantiations only; per-lane pointers where smp varies inside a wave)
DEV DScene scene_of_sample(const DScene& s, uint32_t smp)
{
    DScene v = s;
    const uint32_t* t = s.ver + SCENE_VER_WORDS * (size_t)smp;
    v.nodes = s.nodes + t[0]; v.triVerts = s.triVerts + t[1]; v.vertices = s.vertices + t[2]; v.tlas = s.tlas + t[3]; v.xforms = s.xforms + t[4];
    return v;
}

struct Frame {
    float invProj[16]; float invView[16]; float viewPos[3];
    int W, H, rowMod, rowRem, rows;
    int rowBandLog2;            // rows are dealt in bands of 2^rowBandLog2 rows (idkptSetRowBands; 0: single rows / a strip): see global_row
    GpuSettings g;
    uint32_t accumulated;       // AccumulatedSamples of sample 0 of the batch (== accum[0])
    int useTlas, stackCap, outputAovs;
    int tlasCap;                // rows of the per-lane TLAS stack (<= TLAS_STACK_SIZE; a TLAS over n instances is never deeper than n)
    int leafMin;                // k_trace2 leaves its node phase when this many lanes are parked on a leaf
    uint32_t gridRaysX4, gridMid, gridMidRays;   // k_trace2: the waves of a launch beyond what its ray count wants retire at once (the rules of small_launch_grid, idkpt.hip, applied on the device to the launch's actual count)
    int grabUnitLog2, grabFixed;   // k_trace2's work-list hand-out: a slice owns runs of 2^grabUnitLog2 entries; developer knob: entries reserved per atomic (0: what the refill needs)
    // batch of independent samples traced together (DESIGN.md "Batching"): sample s owns ray ids [s*Npad, s*Npad+N)
    int batch; uint32_t Npad; uint32_t accum[256];   // [MAX_BATCH]
    uint32_t seqFirst, seqStride;   // idkptSetSampleSequence: sample i of an accumulation draws the RNG streams of AccumulatedSamples = seqFirst + i * seqStride (reference: 0, 1)
    // frame ring (idkptSetFrameRing): sample k renders with camera cams[36*k ..] (null: the one camera above) into result-image slot slotOf[k]
    const float* cams; uint32_t slotOf[256];   // (dwords: scalar loads from the kernel-argument segment; gfx9 has no scalar byte load)
    int advMin;                 // k_trace2 MODE 1-4: lanes whose BLAS is exhausted wait for this many of their kind before they enter the next instance / walk the TLAS (1: at once)
    int poolMin;                // pooled leaf phase (k_trace2 DBG 16): (ray, triangle) pairs a wave must have parked before they are tested by all lanes together
    int splitPeek;              // k_trace2s: iterations between two looks at the work-list heads of a wave that is too busy to refill
    int splitMode;              // k_trace2s (kernels_trace_split.hpp): bits 0-1: 2 = every ray whose pieces found a hit is traced again sequentially (test hook for the re-trace path); bit 2: only rays without a hit donate subtrees
    int scatterLog2;            // k_trace2s: the work list is handed out in groups of 2^scatterLog2 entries taken from places far apart (6 = in list order)
    int gbStride, gbBands;      // per-bounce slot bases (k_shade's gbase): one per sample (strips: stride 1, gbBands 0) or one per (sample, local band) for interleaved rows (idkptSetBandExchange)
    int queryMode;              // k_trace2 serves idkptTraceRays: a ray starts from the T / light its trace-ready record carries (record[1].w, record[2].w) instead of FLOAT_MAX / 0
    int shadeMin;               // k_trace_fused: lanes that wait for the shading phase before it runs
    int hitsByRid;              // the last bounce's hit records are indexed by ray id instead of queue slot (k_trace_fused, kernels_trace_fused.hpp)
    int genPixelMajor;          // k_gen_primary appends a tile's survivors pixel by pixel over 16 samples (batches of >= 8 samples): kernels_trace.hpp
    int instSieve;              // several instances without USE_TLAS: the exact loop with the per-ray instance sieve (kernels_trace_inst.hpp, EXACT) instead of k_trace2 MODE 1
    int instTlas;               // several instances without USE_TLAS walked through the library's own TLAS (kernels_trace_inst.hpp): the producers of a ray also write its world 1/dir
    int packet;                 // this batch's primary launch is a packet launch (kernels_packet.hpp; decided by the host per batch: host_launch.hpp packet_decide)
    int tilePerSample;          // k_classify_tiles ran once per sample of the batch (per-sample cameras or scene versions): tile classes are indexed [sample][tile]
};
#define MAX_BATCH 256

struct RayBufs {                // SoA planes of the reference's GpuWavefrontRay / GpuAovRay, indexed by local pixel
    float4* o_ior;              // Origin.xyz, PreviousIOROrTraverseCost
    float4* thr_px;             // Throughput.xyz, PackedDirectionX
    float4* rad_py;             // Radiance.xyz, PackedDirectionY
    float4* aovA;               // Albedo.xyz, NewWeight
    float4* aovN;               // Normal.xyz, pad
};
struct TraceBufs {              // derived, per ray id: the ray ready for the traversal kernel, one 64-B record (the size and alignment of a node pair, so
    float4* rec;                // that a refill touches one cache line per ray instead of three): [0] RayTransform(origin) (Ray.glsl:7-12), .w = tMin of
                                // the root-box test (+inf = miss), so that the traversal kernel's root test is one compare; [1] RayTransform(direction), not renormalised; [2] 1 / [1] (IntersectionRoutines.glsl:29); [3] unused.
                                // Several instances / TLAS: [0],[1] hold the WORLD-space ray, [2] the world 1/dir (TLAS walks only: USE_TLAS or Frame::instTlas).
    // a bounce launch over a SUBSET of the queue (null: the whole queue in order): the launch hands out positions of `order`; order[i] = queue slot, orderIdx[i] = the ray id
    // in that slot — the exact re-trace behind a wide-node launch (kernels_wide.hpp).  The slot a hit is stored at — which seeds NHit's RNG (NHit/compute.glsl:54) — stays what it is.
    const uint32_t* order; const uint32_t* orderIdx;
};
struct HitBufs {                // indexed by queue slot: one 32-B record per slot (one aligned store sector instead of a 16-B and a 4-B partial write)
    float4* hit;                // [2*slot] = T, BaryXY.x, BaryXY.y, TriangleId (bits); [2*slot+1].x = MeshTransformId or light index (bits)
    float* cost;                // debugCost (only written in DoDebugBVHTraversal mode)
};

struct HitRec { float T, bx, by; uint32_t tri, xform; };
DEV void store_hit(const HitBufs& h, size_t slot, float T, float bx, float by, uint32_t tri, uint32_t xform)
{
    h.hit[2 * slot] = make_float4(T, bx, by, __uint_as_float(tri));
    h.hit[2 * slot + 1] = make_float4(__uint_as_float(xform), 0.0f, 0.0f, 0.0f);
}
DEV HitRec load_hit(const HitBufs& h, size_t slot)
{
    const float4 a = h.hit[2 * slot]; const float x = h.hit[2 * slot + 1].x;
    HitRec r; r.T = a.x; r.bx = a.y; r.by = a.z; r.tri = __float_as_uint(a.w); r.xform = __float_as_uint(x);
    return r;
}

#define TLAS_STACK_SIZE 32

// ---------------------------------------------------------------------------------------------------------------
// BVH traversal (include/BVHIntersect.glsl:27-105, 183-291).  One ray per lane; the per-lane traversal stack lives in
// LDS as stack[depth][lane] (bank-conflict free: lane l and l+32 never share a cycle on ds_*_b32).
template <bool COUNT, bool COST>
DEV bool IntersectBlas(const DScene& s, f3 ro, f3 rd, const GpuBlasDesc& dref, bool useTlas, HitRec& hit, float& debugCost,
                       uint32_t* stk, int stride, int cap, uint32_t& nPairs, uint32_t& nTris)
{
    bool anyHit = false;
    float tMinLeft, tMinRight;
    f3 invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    struct { int NodeOffset, TriangleOffset; } d = {dref.NodeOffset, dref.TriangleOffset}; // keep in registers (no reload in the leaf path)
    const float4* nodes = s.nodes + 2 * (size_t)d.NodeOffset;
    if (!useTlas) {
        float4 rmin = nodes[2], rmax = nodes[3];
        if (!(RayBoxIntersect(ro, invDir, rmin, rmax, &tMinLeft) && tMinLeft < hit.T)) return false;
    }
    int sp = 0;
    uint32_t top = 2;
    while (true) {
        if (COST) debugCost += 1.0f;
        if (COUNT) nPairs++;
        const float4* p = nodes + 2 * (size_t)top;
        float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
        uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
        bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hit.T;
        bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hit.T;
        bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
        if (intersectLeft || intersectRight) {
            uint32_t first = intersectLeft ? lStart : rStart;
            uint32_t end = !intersectRight ? (lStart + lCount) : (rStart + rCount);
            first += (uint32_t)d.TriangleOffset; end += (uint32_t)d.TriangleOffset;
            if (COST) debugCost += (float)(end - first) * 1.1f;
            if (COUNT) nTris += end - first;
            for (uint32_t i = first; i < end; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)i;
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hit.T) {
                    anyHit = true; hit.tri = i; hit.bx = 1.0f - by - bz; hit.by = by; hit.T = t;
                }
            }
        }
        bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
        if (traverseLeft || traverseRight) {
            if (traverseLeft && traverseRight) {
                bool leftCloser = tMinLeft < tMinRight;
                top = leftCloser ? lStart : rStart;
                if (sp < cap) stk[sp * stride] = leftCloser ? rStart : lStart; else *s.overflow = 1u;
                sp++;
            } else top = traverseLeft ? lStart : rStart;
        } else {
            if (sp == 0) break;
            sp--;
            if (sp >= cap) break;          // the matching push was dropped (flagged): stop instead of following a garbage index
            top = stk[sp * stride];
        }
    }
    return anyHit;
}

DEV M34 load_inv_model_at(const float4* xforms, uint32_t xformId) { const float4* x = xforms + 9 * (size_t)xformId; M34 m; m.r0 = x[3]; m.r1 = x[4]; m.r2 = x[5]; return m; }
DEV M34 load_inv_model(const DScene& s, uint32_t xformId) { return load_inv_model_at(s.xforms, xformId); }

// traceLights / maxDist: the path tracer passes (settings.DoTraceLights, FLOAT_MAX) (FirstHit:106, NHit:96); ray queries and the
// shadow kernel pass their own (BVHIntersect.glsl:183, ShadowsRayTraced/compute.glsl:73)
template <bool COUNT, bool COST>
DEV bool TraceRay(const DScene& s, const Frame& f, f3 ro, f3 rd, HitRec& hit, float& debugCost, uint32_t* stk, int stride, uint32_t& nPairs, uint32_t& nTris,
                  bool traceLights, float maxDist)
{
    hit.T = maxDist; hit.tri = ~0u; hit.xform = 0; hit.bx = 0.0f; hit.by = 0.0f;
    debugCost = 0.0f;
    if (traceLights) {
        for (int i = 0; i < s.lightCount; i++) {
            const GpuLight& l = s.lights[i];
            float tMin, tMax;
            if (RaySphereIntersect(ro, rd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hit.T) {
                hit.T = tMin < 0.0f ? tMax : tMin; hit.xform = (uint32_t)i; hit.tri = ~0u;
            }
        }
    }
    if (f.useTlas) {
        if (s.tlasCount == 0) return hit.T != maxDist;
        uint32_t* tstk = stk + f.stackCap * stride; // TLAS stack rows follow the BLAS rows in the same LDS column
        float tMinLeft, tMinRight;
        f3 invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        int sp = 0; uint32_t top = 0;
        while (true) {
            float4 pmin = s.tlas[2 * (size_t)top];
            uint32_t packed = __float_as_uint(pmin.w);
            bool isLeaf = (packed >> 31) == 1;
            uint32_t id = packed & 0x7fffffffu;
            if (isLeaf) {
                GpuBlasInstance inst = s.instances[id];
                M34 inv = load_inv_model(s, inst.MeshTransformId);
                f3 lo = xform34(inv, ro, 1.0f), ld = xform34(inv, rd, 0.0f);
                if (IntersectBlas<COUNT, COST>(s, lo, ld, s.descs[inst.BlasId], true, hit, debugCost, stk, stride, f.stackCap, nPairs, nTris)) hit.xform = inst.MeshTransformId;
                if (sp == 0 || sp > f.tlasCap) break;
                top = tstk[--sp * stride];
                continue;
            }
            uint32_t l = id, r = id + 1;
            float4 lmin = s.tlas[2 * (size_t)l], lmax = s.tlas[2 * (size_t)l + 1], rmin = s.tlas[2 * (size_t)r], rmax = s.tlas[2 * (size_t)r + 1];
            bool tl = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft < hit.T;
            bool tr = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight < hit.T;
            if (tl || tr) {
                if (tl && tr) { bool lc = tMinLeft < tMinRight; top = lc ? l : r; if (sp < f.tlasCap) tstk[sp * stride] = lc ? r : l; else *s.overflow = 1u; sp++; }
                else top = tl ? l : r;
            } else { if (sp == 0 || sp > f.tlasCap) break; top = tstk[--sp * stride]; }
        }
    } else {
        for (int i = 0; i < s.instanceCount; i++) {
            GpuBlasInstance inst = s.instances[i];
            M34 inv = load_inv_model(s, inst.MeshTransformId);
            f3 lo = xform34(inv, ro, 1.0f), ld = xform34(inv, rd, 0.0f);
            if (IntersectBlas<COUNT, COST>(s, lo, ld, s.descs[inst.BlasId], false, hit, debugCost, stk, stride, f.stackCap, nPairs, nTris)) hit.xform = inst.MeshTransformId;
        }
    }
    return hit.T != maxDist;
}

template <bool COUNT, bool COST>
DEV bool TraceRay(const DScene& s, const Frame& f, f3 ro, f3 rd, HitRec& hit, float& debugCost, uint32_t* stk, int stride, uint32_t& nPairs, uint32_t& nTris)
{
    return TraceRay<COUNT, COST>(s, f, ro, rd, hit, debugCost, stk, stride, nPairs, nTris, f.g.DoTraceLights != 0, PT_FLOAT_MAX);
}

// Any-hit variants (BVHIntersect.glsl:107-181, 299-411): first intersection found wins; children visited left first.
DEV bool IntersectBlasAny(const DScene& s, f3 ro, f3 rd, const GpuBlasDesc& dref, bool useTlas, HitRec& hit, uint32_t* stk, int stride, int cap)
{
    float tMinLeft, tMinRight;
    f3 invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    const int triOffset = dref.TriangleOffset;
    const float4* nodes = s.nodes + 2 * (size_t)dref.NodeOffset;
    if (!useTlas) {
        float4 rmin = nodes[2], rmax = nodes[3];
        if (!(RayBoxIntersect(ro, invDir, rmin, rmax, &tMinLeft) && tMinLeft < hit.T)) return false;
    }
    int sp = 0;
    uint32_t top = 2;
    while (true) {
        const float4* p = nodes + 2 * (size_t)top;
        float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
        uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
        bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hit.T;
        bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hit.T;
        bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
        if (intersectLeft || intersectRight) {
            uint32_t first = intersectLeft ? lStart : rStart;
            uint32_t end = !intersectRight ? (lStart + lCount) : (rStart + rCount);
            first += (uint32_t)triOffset; end += (uint32_t)triOffset;
            for (uint32_t i = first; i < end; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)i;
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hit.T) {
                    hit.tri = i; hit.bx = 1.0f - by - bz; hit.by = by; hit.T = t;
                    return true;
                }
            }
        }
        bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
        if (traverseLeft || traverseRight) {
            if (traverseLeft && traverseRight) { top = lStart; if (sp < cap) stk[sp * stride] = rStart; else *s.overflow = 1u; sp++; }
            else top = traverseLeft ? lStart : rStart;
        } else {
            if (sp == 0) break;
            sp--;
            if (sp >= cap) break;
            top = stk[sp * stride];
        }
    }
    return false;
}

DEV bool TraceRayAny(const DScene& s, const Frame& f, f3 ro, f3 rd, HitRec& hit, uint32_t* stk, int stride, bool traceLights, float maxDist)
{
    hit.T = maxDist; hit.tri = ~0u; hit.xform = 0; hit.bx = 0.0f; hit.by = 0.0f;
    if (traceLights) {
        for (int i = 0; i < s.lightCount; i++) {
            const GpuLight& l = s.lights[i];
            float tMin, tMax;
            if (RaySphereIntersect(ro, rd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hit.T) {
                hit.T = tMin < 0.0f ? tMax : tMin; hit.xform = (uint32_t)i;
                return true;
            }
        }
    }
    if (f.useTlas) {
        if (s.tlasCount == 0) return false;
        uint32_t* tstk = stk + f.stackCap * stride;
        float tMinLeft, tMinRight;
        f3 invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        int sp = 0; uint32_t top = 0;
        while (true) {
            float4 pmin = s.tlas[2 * (size_t)top];
            uint32_t packed = __float_as_uint(pmin.w);
            bool isLeaf = (packed >> 31) == 1;
            uint32_t id = packed & 0x7fffffffu;
            if (isLeaf) {
                GpuBlasInstance inst = s.instances[id];
                M34 inv = load_inv_model(s, inst.MeshTransformId);
                f3 lo = xform34(inv, ro, 1.0f), ld = xform34(inv, rd, 0.0f);
                if (IntersectBlasAny(s, lo, ld, s.descs[inst.BlasId], true, hit, stk, stride, f.stackCap)) { hit.xform = inst.MeshTransformId; return true; }
                if (sp == 0 || sp > f.tlasCap) break;
                top = tstk[--sp * stride];
                continue;
            }
            uint32_t l = id, r = id + 1;
            float4 lmin = s.tlas[2 * (size_t)l], lmax = s.tlas[2 * (size_t)l + 1], rmin = s.tlas[2 * (size_t)r], rmax = s.tlas[2 * (size_t)r + 1];
            bool tl = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft < hit.T;
            bool tr = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight < hit.T;
            if (tl || tr) {
                if (tl && tr) { bool lc = tMinLeft < tMinRight; top = lc ? l : r; if (sp < f.tlasCap) tstk[sp * stride] = lc ? r : l; else *s.overflow = 1u; sp++; }
                else top = tl ? l : r;
            } else { if (sp == 0 || sp > f.tlasCap) break; top = tstk[--sp * stride]; }
        }
    } else {
        for (int i = 0; i < s.instanceCount; i++) {
            GpuBlasInstance inst = s.instances[i];
            M34 inv = load_inv_model(s, inst.MeshTransformId);
            f3 lo = xform34(inv, ro, 1.0f), ld = xform34(inv, rd, 0.0f);
            if (IntersectBlasAny(s, lo, ld, s.descs[inst.BlasId], false, hit, stk, stride, f.stackCap)) { hit.xform = inst.MeshTransformId; return true; }
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------
// Image row of a context's local row: band (ly >> b) of the context is band (ly >> b) * rowMod + rowRem of the image (b = 0: rows y % rowMod == rowRem,
// or the strip that starts at rowRem when rowMod == 1).  Bands of 8 rows keep a wave's 8x8 pixel tile one 8x8 block of the image on every rank.
DEV int global_row(const Frame& f, int ly) { return ((((ly >> f.rowBandLog2) * f.rowMod + f.rowRem) << f.rowBandLog2) | (ly & ((1 << f.rowBandLog2) - 1))); }

// Primary ray generation (FirstHit/compute.glsl:44-77).  `pix` = local pixel index.
// the value the reference's shaders would read as wavefrontPTSSBO.AccumulatedSamples for sample `smp` of the batch (FinalDraw's weight keeps the plain count)
DEV uint32_t sample_index(const Frame& f, uint32_t smp) { return f.seqFirst + f.accum[smp] * f.seqStride; }
DEV void gen_primary(const Frame& f, uint32_t smp, uint32_t pix, uint32_t acc, f3& origin, f2& packedDir, uint32_t& rngSeed)
{
    const float* cam = f.cams ? f.cams + 36u * smp : f.invProj;      // invProj[16] invView[16] viewPos[3], contiguous in both places
    const float* invProj = cam; const float* invView = cam + 16; const float* vp = cam + 32;
    int lx = (int)(pix % (uint32_t)f.W), ly = (int)(pix / (uint32_t)f.W);
    int y = global_row(f, ly), x = lx;
    uint32_t seed = (uint32_t)(y * 4096 + x) * (acc + 1u);
    float ox = rnd01(seed), oy = rnd01(seed);
    float nx = ((float)x + ox) / (float)f.W * 2.0f - 1.0f, ny = ((float)y + oy) / (float)f.H * 2.0f - 1.0f;
    f3 camDir = GetWorldSpaceDirection(invProj, invView, nx, ny);
    f3 viewPos = mk3(vp[0], vp[1], vp[2]);
    f3 focalPoint = viewPos + camDir * f.g.FocalLength;
    f2 disk = SampleDisk(seed);
    f3 pointOnLense = mat4_mul_xyz(invView, f.g.LenseRadius * disk.x, f.g.LenseRadius * disk.y, 0.0f, 1.0f);
    camDir = normalize(focalPoint - pointOnLense);
    origin = pointOnLense;
    packedDir = EncodeUnitVec(camDir);
    rngSeed = seed;
}

// gl_GlobalInvocationID of the FirstHit invocation for pixel (px,py): inverse of ReorderInvocations(20) (FirstHit:236-262)
DEV uint32_t first_hit_gid_seed(int W, int H, int px, int py)
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
    uint32_t gx = wgX * 8 + (uint32_t)px % 8, gy = wgY * 8 + (uint32_t)py % 8;
    return gy * 4096u + gx;
}

// ---------------------------------------------------------------------------------------------------------------
// texture / sky stand-ins for the GL bindless samplers (DESIGN.md "Textures")
// Texel selection of one axis (GL 4.6 8.14.2, table 8.20).  REPEAT keeps round 5's expression ((int)c % n on the floored coordinate and on floor + 1: the fixtures' bits).
DEV int tex_wrap(int i, int n, uint32_t mode)
{
    if (mode == 1u) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);                                   // CLAMP_TO_EDGE: clamp(coord, 0, size - 1)
    if (mode == 2u) { int m = i % (2 * n); if (m < 0) m += 2 * n; const int a = m - n; return (n - 1) - (a >= 0 ? a : -(1 + a)); }   // MIRRORED_REPEAT: (size - 1) - mirror((coord mod (2 size)) - size)
    int k = i % n; if (k < 0) k += n; return k;                                                   // REPEAT: coord mod size
}
DEV float4 tex_fetch(const DScene& s, const TexDesc& t, int x, int y)
{
    const uint32_t fmt = t.state >> 5;
    const size_t at = (size_t)y * (size_t)t.w + (size_t)x;
    if (fmt == 0u) return ((const float4*)t.data)[at];
    const uint32_t p = ((const uint32_t*)t.data)[at];                                              // bytes R, G, B, A
    const float a = (float)(p >> 24) / 255.0f;
    if (fmt == 1u) return make_float4((float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f, a);
    return make_float4(s.srgbLut[p & 255u], s.srgbLut[(p >> 8) & 255u], s.srgbLut[(p >> 16) & 255u], a);   // sRGB -> linear per texel, before the filter (GL 4.6 8.24): a 256-entry table made on the host
}
DEV float4 SampleTex(const DScene& s, uint64_t handle, float u, float v)
{
    if (handle == 0 || handle > (uint64_t)s.textureCount) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const TexDesc t = s.textures[handle - 1];
    if (t.w == 1 && t.h == 1) return tex_fetch(s, t, 0, 0);
    const uint32_t ws = t.state & 3u, wt = (t.state >> 2) & 3u;
    if ((t.state >> 4) & 1u) {                                                                     // NEAREST: i = wrap(floor(u)) (8.14.2)
        const int x = tex_wrap((int)gfloor(u * (float)t.w), t.w, ws), y = tex_wrap((int)gfloor(v * (float)t.h), t.h, wt);
        return tex_fetch(s, t, x, y);
    }
    float fx = u * (float)t.w - 0.5f, fy = v * (float)t.h - 0.5f;
    float x0f = gfloor(fx), y0f = gfloor(fy);
    float ax = fx - x0f, ay = fy - y0f;
    const int x0 = tex_wrap((int)x0f, t.w, ws), x1 = tex_wrap((int)(x0f + 1.0f), t.w, ws), y0 = tex_wrap((int)y0f, t.h, wt), y1 = tex_wrap((int)(y0f + 1.0f), t.h, wt);
    const float4 a = tex_fetch(s, t, x0, y0), b = tex_fetch(s, t, x1, y0), c = tex_fetch(s, t, x0, y1), d = tex_fetch(s, t, x1, y1);
    return make_float4(gmix(gmix(a.x, b.x, ax), gmix(c.x, d.x, ax), ay), gmix(gmix(a.y, b.y, ax), gmix(c.y, d.y, ax), ay),
                       gmix(gmix(a.z, b.z, ax), gmix(c.z, d.z, ax), ay), gmix(gmix(a.w, b.w, ax), gmix(c.w, d.w, ax), ay));
}
// texture(skyBoxUBO.Albedo, dir) on the reference's GL_LINEAR, seamless cube map (Render/SkyBoxManager.cs:44,74): table 8.19 face
// selection, linear filter at LOD 0; a texel one step beyond a face edge is the adjacent face's texel it folds onto (integer lattice,
// exact), the ownerless texel beyond a corner is the mean of the footprint's other three.  S == 1 = constant colour per face, unfiltered.
DEV void SkyFold(int S, int& face, int& x, int& y)
{
    int sc = 2 * x + 1 - S, tc = 2 * y + 1 - S;
    int px, py, pz;
    switch (face) {
        case 0: px = S; py = -tc; pz = -sc; break;
        case 1: px = -S; py = -tc; pz = sc; break;
        case 2: py = S; px = sc; pz = tc; break;
        case 3: py = -S; px = sc; pz = -tc; break;
        case 4: pz = S; px = sc; py = -tc; break;
        default: pz = -S; px = -sc; py = -tc; break;
    }
    const int m = face >> 1;
    const bool ox = m != 0 && (px > S || px < -S), oy = m != 1 && (py > S || py < -S), oz = m != 2 && (pz > S || pz < -S);
    if (!(ox || oy || oz)) return;
    const int in = S - 1;
    if (m == 0) px = px > 0 ? in : -in; else if (m == 1) py = py > 0 ? in : -in; else pz = pz > 0 ? in : -in;
    if (oz) { pz = pz > 0 ? S : -S; face = pz > 0 ? 4 : 5; }          // (at most one axis overflows: corners never get here)
    else if (oy) { py = py > 0 ? S : -S; face = py > 0 ? 2 : 3; }
    else { px = px > 0 ? S : -S; face = px > 0 ? 0 : 1; }
    switch (face) {
        case 0: sc = -pz; tc = -py; break;
        case 1: sc = pz; tc = -py; break;
        case 2: sc = px; tc = pz; break;
        case 3: sc = px; tc = -pz; break;
        case 4: sc = px; tc = -py; break;
        default: sc = -px; tc = -py; break;
    }
    x = (sc + S - 1) / 2; y = (tc + S - 1) / 2;
}
DEV f3 SampleSky(const DScene& s, f3 d)
{
    if (s.skySize <= 0) return splat3(0.0f);
    float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
    int face; float sc, tc, ma;
    if (ax >= ay && ax >= az) { face = d.x >= 0.0f ? 0 : 1; sc = d.x >= 0.0f ? -d.z : d.z; tc = -d.y; ma = ax; }
    else if (ay >= az) { face = d.y >= 0.0f ? 2 : 3; sc = d.x; tc = d.y >= 0.0f ? d.z : -d.z; ma = ay; }
    else { face = d.z >= 0.0f ? 4 : 5; sc = d.z >= 0.0f ? d.x : -d.x; tc = -d.y; ma = az; }
    const int S = s.skySize;
    if (S == 1) { float4 p = s.sky[face]; return mk3(p.x, p.y, p.z); }
    float u = 0.5f * (sc / ma + 1.0f), v = 0.5f * (tc / ma + 1.0f);
    float fx = u * (float)S - 0.5f, fy = v * (float)S - 0.5f;
    float x0f = gfloor(fx), y0f = gfloor(fy);
    float wx = fx - x0f, wy = fy - y0f;
    const int x0 = (int)gmin(gmax(x0f, -1.0f), (float)(S - 1)), y0 = (int)gmin(gmax(y0f, -1.0f), (float)(S - 1));
    f3 t[4]; bool corner[4]; bool anyCorner = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int x = x0 + (k & 1), y = y0 + (k >> 1), f2 = face;
        corner[k] = (x < 0 || x >= S) && (y < 0 || y >= S);
        t[k] = splat3(0.0f);
        if (corner[k]) { anyCorner = true; continue; }
        SkyFold(S, f2, x, y);
        float4 p = s.sky[((size_t)f2 * S + y) * S + x];
        t[k] = mk3(p.x, p.y, p.z);
    }
    if (anyCorner) {
        f3 sum = splat3(0.0f);
#pragma unroll
        for (int k = 0; k < 4; k++) if (!corner[k]) sum = sum + t[k];
        const f3 mean = mk3(sum.x / 3.0f, sum.y / 3.0f, sum.z / 3.0f);
#pragma unroll
        for (int k = 0; k < 4; k++) if (corner[k]) t[k] = mean;
    }
    return mk3(gmix(gmix(t[0].x, t[1].x, wx), gmix(t[2].x, t[3].x, wx), wy),
               gmix(gmix(t[0].y, t[1].y, wx), gmix(t[2].y, t[3].y, wx), wy),
               gmix(gmix(t[0].z, t[1].z, wx), gmix(t[2].z, t[3].z, wx), wy));
}

// ---------------------------------------------------------------------------------------------------------------
// Shading: FirstHit TraceRay after the trace (FirstHit/compute.glsl:114-233) / NHit TraceRay (NHit/compute.glsl:98-214)
struct RayState { f3 origin; float prevIor; f3 throughput; float pdx; f3 radiance; float pdy; };
struct AovState { f3 albedo; float newWeight; f3 normal; };

// RAD_ONLY (k_shade_last): stop once the radiance of this hit is final — the continuation (SampleMaterial, the new ray, Russian roulette) is not wanted; everything the
// radiance depends on (alpha test incl. its random number, absorption, emission) is the code below, unchanged.
template <bool FIRST, bool RAD_ONLY = false>
DEV bool ShadeHit(const DScene& s, const Frame& f, uint32_t acc, const HitRec& hit, bool hitScene, f3 rayDir, RayState& r, AovState& aov, uint32_t& rng, uint32_t gidSeed, uint32_t& sortingKey)
{
    if (hitScene) {
        r.origin = r.origin + rayDir * hit.T;
        // GetDefaultSurface (Surface.glsl:25-47)
        f3 sAlbedo = splat3(1.0f), sNormal = splat3(0.0f), sEmissive = splat3(0.0f), sAbsorbance = splat3(0.0f);
        float sAlpha = 1.0f, sMetallic = 0.0f, sRoughness = 0.0f, sTransmission = 0.0f, sIOR = 1.5f, sAlphaCutoff = 0.5f;
        bool sIsVolumetric = false, sTint = true;
        f3 geometricNormal = splat3(0.0f);
        bool hitLight = hit.tri == ~0u;
        if (!hitLight) {
            sortingKey = hit.tri;
            uint4 tri = s.tris[hit.tri];
            uint4 v0 = s.vertices[tri.x], v1 = s.vertices[tri.y], v2 = s.vertices[tri.z];
            f3 bary = mk3(hit.bx, hit.by, 1.0f - hit.bx - hit.by);
            float u = __uint_as_float(v0.x) * bary.x + __uint_as_float(v1.x) * bary.y + __uint_as_float(v2.x) * bary.z;
            float v = __uint_as_float(v0.y) * bary.x + __uint_as_float(v1.y) * bary.y + __uint_as_float(v2.y) * bary.z;
            f3 interpNormal = normalize(Interpolate(DecompressSR11G11B10(v0.w), DecompressSR11G11B10(v1.w), DecompressSR11G11B10(v2.w), bary));
            f3 interpTangent = normalize(Interpolate(DecompressSR11G11B10(v0.z), DecompressSR11G11B10(v1.z), DecompressSR11G11B10(v2.z), bary));
            M34 inv = load_inv_model(s, hit.xform);
            const GpuMesh& mesh = s.meshes[tri.w];
            const GpuMaterial& mat = s.materials[mesh.MaterialId];
            { // GetSurface (Surface.glsl:49-77)
                uint32_t bcf = mat.BaseColorFactor;
                float4 bc = SampleTex(s, mat.BaseColorTexture, u, v);
                sAlbedo = mk3(bc.x * ((float)(bcf & 255u) / 255.0f), bc.y * ((float)((bcf >> 8) & 255u) / 255.0f), bc.z * ((float)((bcf >> 16) & 255u) / 255.0f));
                sAlpha = bc.w * ((float)((bcf >> 24) & 255u) / 255.0f);
                float4 nm = SampleTex(s, mat.NormalTexture, u, v);
                sNormal = mk3(nm.x * 2.0f - 1.0f, nm.y * 2.0f - 1.0f, gsqrt(gmax(1.0f - (nm.x * nm.x + nm.y * nm.y), 0.0f)));
                float4 em = SampleTex(s, mat.EmissiveTexture, u, v);
                sEmissive = mk3(em.x * mat.EmissiveFactor[0], em.y * mat.EmissiveFactor[1], em.z * mat.EmissiveFactor[2]);
                sAbsorbance = mk3(mat.Absorbance[0], mat.Absorbance[1], mat.Absorbance[2]);
                float4 mr = SampleTex(s, mat.MetallicRoughnessTexture, u, v);
                sMetallic = mr.x * mat.MetallicFactor; sRoughness = mr.y * mat.RoughnessFactor;
                float4 tr = SampleTex(s, mat.TransmissionTexture, u, v);
                sTransmission = tr.x * mat.TransmissionFactor; sIOR = mat.IOR;
                sAlphaCutoff = mat.AlphaCutoff; sIsVolumetric = mat.IsVolumetric != 0;
            }
            { // SurfaceApplyModificatons (Surface.glsl:79-91)
                sEmissive = sEmissive * 1.0f + mesh.EmissiveBias * sAlbedo;
                sAbsorbance = mk3(gmax(sAbsorbance.x + mesh.AbsorbanceBias[0], 0.0f), gmax(sAbsorbance.y + mesh.AbsorbanceBias[1], 0.0f), gmax(sAbsorbance.z + mesh.AbsorbanceBias[2], 0.0f));
                sMetallic = gclamp(sMetallic + mesh.SpecularBias, 0.0f, 1.0f);
                sRoughness = gclamp(sRoughness + mesh.RoughnessBias, 0.0f, 1.0f);
                sTransmission = gclamp(sTransmission + mesh.TransmissionBias, 0.0f, 1.0f);
                sIOR = gmax(sIOR + mesh.IORBias, 1.0f);
                sTint = mesh.TintOnTransmissive != 0;
            }
            float alphaCutoff = (sAlphaCutoff == 2.0f) ? rnd01(rng) : sAlphaCutoff;
            if (sAlpha < alphaCutoff) { r.origin = r.origin + rayDir * 0.001f; return true; }
            f3 worldNormal = normalize(xform34_transposed3(inv, interpNormal));
            f3 worldTangent = normalize(xform34_transposed3(inv, interpTangent));
            f3 N = normalize(worldNormal), T = normalize(worldTangent), B = normalize(cross(N, T));
            f3 tn = mk3((T.x * sNormal.x + B.x * sNormal.y) + N.x * sNormal.z, (T.y * sNormal.x + B.y * sNormal.y) + N.y * sNormal.z, (T.z * sNormal.x + B.z * sNormal.y) + N.z * sNormal.z);
            sNormal = normalize(gmix(worldNormal, tn, mesh.NormalMapStrength));
            const float4* tv = s.triVerts + 3 * (size_t)hit.tri;
            float4 a = tv[0], b = tv[1], c = tv[2];
            f3 p0 = mk3(a.x, a.y, a.z), p1 = mk3(b.x, b.y, b.z), p2 = mk3(c.x, c.y, c.z);
            geometricNormal = normalize(cross(p1 - p0, p2 - p0));
            geometricNormal = normalize(xform34_transposed3(inv, geometricNormal));
        } else if (f.g.DoTraceLights) {
            sortingKey = hit.xform;
            const GpuLight& l = s.lights[hit.xform];
            sEmissive = mk3(l.Color[0], l.Color[1], l.Color[2]); sAlbedo = sEmissive;
            sNormal = (r.origin - mk3(l.Position[0], l.Position[1], l.Position[2])) / l.Radius;
            geometricNormal = sNormal;
        }
        float prevIor = FIRST ? 1.0f : r.prevIor;
        bool fromInside = dot(-rayDir, geometricNormal) < 0.0f;
        if (fromInside) {
            if (FIRST) prevIor = sIOR;
            geometricNormal = geometricNormal * -1.0f;
            if (sIsVolumetric) {
                f3 e = (-sAbsorbance) * hit.T;
                r.throughput = r.throughput * mk3(gexp(e.x), gexp(e.y), gexp(e.z));
            }
        }
        float cosTheta = dot(-rayDir, sNormal);
        if (cosTheta < 0.0f) { sNormal = sNormal * -1.0f; }
        r.radiance = r.radiance + sEmissive * r.throughput;
        if (RAD_ONLY) return false;

        // SampleMaterial (Shading.glsl:59-150)
        float roughness2 = sRoughness * sRoughness;
        float metallic, transmission;
        {
            float ct = dot(-rayDir, sNormal);
            float diffuseChance = 1.0f - sMetallic - sTransmission;
            float f0 = BaseReflectivity(prevIor, sIOR);
            metallic = gmix(sMetallic, 1.0f, FresnelSchlick(f0, 1.0f, ct));
            transmission = gmax(1.0f - diffuseChance - metallic, 0.0f);
        }
        uint32_t bsdfType;
        {
            float rnd = rnd01(rng);
            if (metallic > rnd) bsdfType = 1u;
            else if (metallic + transmission > rnd) bsdfType = 2u;
            else bsdfType = 0u;
        }
        f3 diffuseRayDir;
        {
            uint32_t local = gidSeed;
            f2 r2 = R2Sequence(acc);
            float px = rnd01(local), py = rnd01(local);
            f2 uv; uv.x = gfract(r2.x + px); uv.y = gfract(r2.y + py);
            diffuseRayDir = CosineSampleHemisphere(sNormal, uv);
        }
        f3 newDir, bsdf; float pdf, newIor;
        if (bsdfType == 0u) { newDir = diffuseRayDir; newIor = prevIor; bsdf = sAlbedo; pdf = 1.0f; }
        else if (bsdfType == 1u) {
            f3 refl = reflect(rayDir, sNormal);
            newDir = normalize(gmix(refl, diffuseRayDir, roughness2));
            bsdf = sAlbedo; pdf = 1.0f; newIor = prevIor;
        } else {
            newIor = fromInside ? 1.0f : sIOR;
            f3 refr; bool tir;
            if (!sIsVolumetric) { refr = rayDir; tir = false; newIor = 1.0f; }
            else {
                refr = refract(rayDir, sNormal, prevIor / newIor);
                tir = (refr.x == 0.0f && refr.y == 0.0f && refr.z == 0.0f);
                if (tir) { refr = reflect(rayDir, sNormal); newIor = prevIor; }
            }
            newDir = normalize(gmix(refr, !tir ? -diffuseRayDir : diffuseRayDir, roughness2));
            bool gltfWantsTint = sIsVolumetric || !fromInside;
            bsdf = (gltfWantsTint && sTint) ? sAlbedo : splat3(1.0f);
            pdf = 1.0f;
        }
        pdf = gmax(pdf, 0.0001f);
        r.throughput = r.throughput * (bsdf / pdf);
        {
            // GetSurfaceVariance(surface.Metallic, surface.Transmission, surface.Roughness) uses the caller's `surface`
            // (SampleMaterial takes it by value), i.e. the un-squared roughness and the un-adjusted chances.
            float diffuseC = 1.0f - sMetallic - sTransmission;
            float weight = diffuseC + sMetallic * sRoughness + sTransmission * sRoughness;
            if (FIRST) { aov.albedo = sAlbedo * weight; aov.normal = sNormal * weight; aov.newWeight = 1.0f - weight; }
            else { aov.albedo = aov.albedo + aov.newWeight * sAlbedo * weight; aov.normal = aov.normal + aov.newWeight * sNormal * weight; aov.newWeight *= (1.0f - weight); }
        }
        if (!FIRST) {
            if (f.g.DoRussianRoulette) { // RussianRoulette.glsl:3-12
                float p = gmax(r.throughput.x, gmax(r.throughput.y, r.throughput.z));
                if (rnd01(rng) > p) return false;
                r.throughput = r.throughput / p;
            }
        }
        if (bsdfType == 2u) geometricNormal = geometricNormal * -1.0f;
        r.origin = r.origin + geometricNormal * 0.001f;
        r.prevIor = newIor;
        f2 pd = EncodeUnitVec(newDir);
        r.pdx = pd.x; r.pdy = pd.y;
        return true;
    } else {
        f3 albedo = SampleSky(s, rayDir);
        f3 fn = CubemapFaceNormal(rayDir);
        if (FIRST) { aov.albedo = albedo; aov.normal = fn; }
        else { aov.albedo = aov.albedo + aov.newWeight * albedo; aov.normal = aov.normal + aov.newWeight * fn; }
        aov.newWeight = 0.0f;
        r.radiance = r.radiance + albedo * r.throughput;
        return false;
    }
}

} // namespace ptd
