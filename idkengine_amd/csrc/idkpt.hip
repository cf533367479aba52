// idkpt.hip — libidkpt.so: kernels' __global__ entry points and the C-ABI of include/idkpt.h.
// Host schedule mirrors PathTracer.Compute (Source/Render/PathTracer.cs:214-271); see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/idkpt.h"
#include "pt_kernels.hpp"

using namespace ptd;

#define WAVE 64
#define MAX_DEPTH_SLOTS 64

// =================================================================================================== kernels

DEV uint32_t wave_grab(uint32_t* counter, uint32_t amount)
{
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, amount);
    return __builtin_amdgcn_readfirstlane(base);
}
DEV void flush_counters(uint64_t* counters, uint32_t nPairs, uint32_t nTris)
{
    // wave reduction then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { nPairs += __shfl_down(nPairs, off); nTris += __shfl_down(nTris, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd((unsigned long long*)&counters[0], (unsigned long long)nPairs); atomicAdd((unsigned long long*)&counters[1], (unsigned long long)nTris); }
}

// FirstHit part 1: ray generation + closest-hit trace of the primary rays (FirstHit/compute.glsl:44-77,100-106).
// Persistent waves; each wave pulls packets of 64 consecutive pixels.
template <bool COUNT, bool COST>
__global__ __launch_bounds__(WAVE) void k_trace_primary(DScene s, Frame f, RayBufs rays, HitBufs hits, uint32_t N, uint32_t* workCounter, uint64_t* counters)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    uint32_t* stk = lds + lane;
    uint32_t nPairs = 0, nTris = 0;
    while (true) {
        uint32_t base = wave_grab(workCounter, WAVE);
        if (base >= N) break;
        uint32_t pix = base + lane;
        if (pix < N) {
            f3 origin; f2 pd; uint32_t seed;
            gen_primary(f, pix, f.accumulated, origin, pd, seed);
            rays.o_ior[pix] = make_float4(origin.x, origin.y, origin.z, 1.0f);
            rays.thr_px[pix] = make_float4(1.0f, 1.0f, 1.0f, pd.x);
            rays.rad_py[pix] = make_float4(0.0f, 0.0f, 0.0f, pd.y);
            f3 rd = DecodeUnitVec(pd.x, pd.y);
            HitRec hit; float cost;
            TraceRay<COUNT, COST>(s, f, origin, rd, hit, cost, stk, WAVE, nPairs, nTris);
            hits.hit[pix] = make_float4(hit.T, hit.bx, hit.by, __uint_as_float(hit.tri));
            hits.xformId[pix] = hit.xform;
            if (COST) hits.cost[pix] = cost;
        }
    }
    if (COUNT) flush_counters(counters, nPairs, nTris);
}

// NHit part 1: closest-hit trace of the alive queue (NHit/compute.glsl:56-58,93-98)
template <bool COUNT>
__global__ __launch_bounds__(WAVE) void k_trace_queue(DScene s, Frame f, RayBufs rays, HitBufs hits, const uint32_t* queue, const uint32_t* countPtr, uint32_t* workCounter, uint64_t* counters)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    uint32_t* stk = lds + lane;
    const uint32_t N = *countPtr;
    uint32_t nPairs = 0, nTris = 0;
    while (true) {
        uint32_t base = wave_grab(workCounter, WAVE);
        if (base >= N) break;
        uint32_t slot = base + lane;
        if (slot < N) {
            uint32_t idx = queue[slot];
            float4 o = rays.o_ior[idx];
            float pdx = rays.thr_px[idx].w, pdy = rays.rad_py[idx].w;
            f3 rd = DecodeUnitVec(pdx, pdy);
            HitRec hit; float cost;
            TraceRay<COUNT, false>(s, f, mk3(o.x, o.y, o.z), rd, hit, cost, stk, WAVE, nPairs, nTris);
            hits.hit[slot] = make_float4(hit.T, hit.bx, hit.by, __uint_as_float(hit.tri));
            hits.xformId[slot] = hit.xform;
        }
    }
    if (COUNT) flush_counters(counters, nPairs, nTris);
}

// ---------------------------------------------------------------------------------------------------------------
// Adjacent consumers of the traversal core.
// k_trace_query: batched TraceRay / TraceRayAny calls with explicit maxDist and traceLights (BVHIntersect.glsl:183-411).
template <bool ANY>
__global__ __launch_bounds__(WAVE) void k_trace_query(DScene s, Frame f, const idkpt_ray* rays, idkpt_hit* out, uint32_t N, int traceLights, uint32_t* workCounter)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    uint32_t* stk = lds + lane;
    uint32_t nPairs = 0, nTris = 0;
    while (true) {
        uint32_t base = wave_grab(workCounter, WAVE);
        if (base >= N) break;
        uint32_t i = base + lane;
        if (i < N) {
            const float4 a = ((const float4*)rays)[2 * (size_t)i], b = ((const float4*)rays)[2 * (size_t)i + 1];
            HitRec hit; float cost; bool h;
            if (ANY) h = TraceRayAny(s, f, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), hit, stk, WAVE, traceLights != 0, a.w);
            else h = TraceRay<false, false>(s, f, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), hit, cost, stk, WAVE, nPairs, nTris, traceLights != 0, a.w);
            ((float4*)out)[2 * (size_t)i] = make_float4(hit.T, hit.bx, hit.by, __uint_as_float(hit.tri));
            ((uint4*)out)[2 * (size_t)i + 1] = make_uint4(hit.xform, h ? 1u : 0u, 0u, 0u);
        }
    }
}

// k_shadows: Shaders/ShadowsRayTraced/compute.glsl:19-127 for one point shadow; one thread per pixel, 8x8 tiles per wave.
DEV float InterleavedGradientNoise(float cx, float cy, uint32_t index) // Random.glsl:35-41
{
    const float add = (float)index * 5.588238f;
    cx = cx + add; cy = cy + add;
    return gfract(52.9829189f * gfract(0.06711056f * cx + 0.00583715f * cy));
}
DEV f3 SampleSphereCone(f3 toSphere, float sphereRadius, float rnd0, float rnd1, float* distanceToSphere) // Sampling.glsl:21-52 + ConstructBasis (Math.glsl:112-127)
{
    const float radiusSq = sphereRadius * sphereRadius;
    const float distanceSq = dot(toSphere, toSphere);
    const float sinThetaMaxSq = radiusSq / distanceSq;
    const float cosThetaMax = gsqrt(gmax(1.0f - sinThetaMaxSq, 0.0f));
    const float phiMax = 2.0f * PT_PI;
    const float phi = phiMax * rnd0;
    const float cosTheta = gmix(cosThetaMax, 1.0f, gmax(rnd1, 0.001f));
    const float sinTheta = gsqrt(gmax(1.0f - cosTheta * cosTheta, 0.0f));
    *distanceToSphere = gsqrt(dot(toSphere, toSphere)) * cosTheta - gsqrt(radiusSq - distanceSq * sinTheta * sinTheta);
    float sp, cp; gsincos(phi, &sp, &cp);
    const f3 local = mk3(cp * sinTheta, cosTheta, sp * sinTheta);
    const f3 normal = normalize(toSphere);
    const f3 up = gabs(normal.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    const f3 tangent = normalize(cross(up, normal));
    const f3 bitangent = cross(normal, tangent);
    return mk3((tangent.x * local.x + normal.x * local.y) + bitangent.x * local.z,
               (tangent.y * local.x + normal.y * local.y) + bitangent.y * local.z,
               (tangent.z * local.x + normal.z * local.y) + bitangent.z * local.z);
}
__global__ __launch_bounds__(WAVE) void k_shadows(DScene s, Frame f, idkpt_shadow_params p, const float* depthImg, const float2* normalImg, float* vis)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    uint32_t* stk = lds + lane;
    const uint32_t tilesX = ((uint32_t)p.Width + 7) / 8;
    const int x = (int)((blockIdx.x % tilesX) * 8 + (lane & 7)), y = (int)((blockIdx.x / tilesX) * 8 + (lane >> 3));
    if (x >= p.Width || y >= p.Height) return;
    const size_t pix = (size_t)y * p.Width + x;
    uint32_t noiseIndex = p.NoiseIndex, rng = 0u, nPairs = 0, nTris = 0;
    const float depth = depthImg[pix];
    if (depth == 1.0f) return;
    const GpuLight& light = s.lights[p.LightIndex];
    const f3 lightPos = mk3(light.Position[0], light.Position[1], light.Position[2]);
    const float u = ((float)x + 0.5f) / (float)p.Width, v = ((float)y + 0.5f) / (float)p.Height;
    const float nx = (u * 2.0f - 1.0f) - p.TaaJitter[0], ny = (v * 2.0f - 1.0f) - p.TaaJitter[1];
    const float* m = p.InvProjView;
    const f3 wp = mat4_mul_xyz(m, nx, ny, depth, 1.0f);
    const float ww = ((m[3] * nx + m[7] * ny) + m[11] * depth) + m[15] * 1.0f;
    const f3 fragPos = wp / ww;
    const float2 nrg = normalImg[pix];
    const f3 normal = DecodeUnitVec(nrg.x, nrg.y);
    const float cosTheta = dot(normal, normalize(lightPos - fragPos));
    if (cosTheta <= 0.0f) { vis[pix] = 0.0f; return; }
    float visibility = 0.0f;
    for (int i = 0; i < p.RayTracingSamples; i++) {
        const f3 biased = fragPos + normal * 0.01f;
        const float rnd0 = InterleavedGradientNoise((float)x, (float)y, noiseIndex + 0u);
        const float rnd1 = InterleavedGradientNoise((float)x, (float)y, noiseIndex + 1u);
        noiseIndex++;
        const f3 fragToLight = lightPos - biased;
        float distanceToLight;
        const f3 direction = SampleSphereCone(fragToLight, light.Radius, rnd0, rnd1, &distanceToLight);
        f3 ro = biased;
        HitRec hit; float cost;
        float thisVisibility = 1.0f;
        while (TraceRay<false, false>(s, f, ro, direction, hit, cost, stk, WAVE, nPairs, nTris, true, distanceToLight - 0.001f)) {
            if (hit.tri == ~0u) { if (hit.xform != (uint32_t)p.LightIndex) thisVisibility = 0.0f; break; }
            const uint4 tri = s.tris[hit.tri];
            const uint4 v0 = s.vertices[tri.x], v1 = s.vertices[tri.y], v2 = s.vertices[tri.z];
            const f3 bary = mk3(hit.bx, hit.by, 1.0f - hit.bx - hit.by);
            const float tu = __uint_as_float(v0.x) * bary.x + __uint_as_float(v1.x) * bary.y + __uint_as_float(v2.x) * bary.z;
            const float tv = __uint_as_float(v0.y) * bary.x + __uint_as_float(v1.y) * bary.y + __uint_as_float(v2.y) * bary.z;
            const GpuMesh& mesh = s.meshes[tri.w];
            const GpuMaterial& mat = s.materials[mesh.MaterialId];
            const float4 bc = SampleTex(s, mat.BaseColorTexture, tu, tv);                     // GetSurface: only Alpha / AlphaCutoff matter here
            const float alpha = bc.w * ((float)((mat.BaseColorFactor >> 24) & 255u) / 255.0f);
            const bool blend = mat.AlphaCutoff == 2.0f;
            const float alphaCutoff = blend ? rnd01(rng) : mat.AlphaCutoff;
            if (blend) thisVisibility *= 1.0f - alpha;
            else if (alpha > alphaCutoff) thisVisibility = 0.0f;
            if (thisVisibility < 0.01f) break;
            const float dist = hit.T + 0.001f;
            ro = ro + direction * dist;
            distanceToLight -= dist;
        }
        visibility += thisVisibility;
    }
    visibility /= (float)p.RayTracingSamples;
    vis[pix] = visibility;
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path (single BLAS instance, no TLAS): coherent ray generation + persistent "while-while" traversal.
//
// k_gen_primary: one thread per pixel, 8x8 pixel tiles per wave.  Generates the primary ray (FirstHit:44-77), stores it,
// and pre-culls rays whose root-box test (BVHIntersect.glsl:32-39 with T = FLOAT_MAX) fails: those get their miss
// record written here and never reach the traversal kernel.  Survivors are appended (wave ballot + one atomic per
// wave) to an unordered active list; results are stored per pixel, so the list order is free.
__global__ __launch_bounds__(1024) void k_gen_primary(DScene s, Frame f, RayBufs rays, TraceBufs tr, int cull, uint32_t* activeList, uint32_t* activeCount, uint32_t* seedOut, uint8_t* contFlag)
{
    __shared__ uint32_t waveKeep[16]; __shared__ uint32_t blockBase;
    // grid = (samples, tile groups): the samples of one tile group are dispatched back to back, so the active list keeps
    // rays of the same screen region (all samples) together -> coherent waves in the traversal kernel
    const uint32_t smp = blockIdx.x;                                   // sample of the batch
    const uint32_t wave = (blockIdx.y * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t tilesX = ((uint32_t)f.W + 7) / 8;
    const uint32_t tx = wave % tilesX, ty = wave / tilesX;
    const uint32_t x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    const bool valid = x < (uint32_t)f.W && y < (uint32_t)f.rows;
    const uint32_t pix = y * (uint32_t)f.W + x;
    const uint32_t rid = smp * f.Npad + pix;                           // ray id inside the batch
    bool keep = false;
    if (valid) {
        f3 origin; f2 pd; uint32_t seed;
        gen_primary(f, pix, f.accum[smp], origin, pd, seed);
        f3 rd = DecodeUnitVec(pd.x, pd.y);
        f3 lo = origin, ld = rd, invDir = splat3(0.0f);   // several instances / TLAS: the traversal kernel transforms the world ray per instance
        keep = !cull;
        if (f.useTlas) {
            // first TLAS step (BVHIntersect.glsl:242-249) with T = FLOAT_MAX: a ray that misses both children of the root is a miss
            invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
            if (cull) {
                if (s.tlasCount == 0) keep = false;
                else {
                    const uint32_t packed = __float_as_uint(s.tlas[0].w), id = packed & 0x7fffffffu;
                    if ((packed >> 31) == 1u) keep = true;
                    else {
                        float t1, t2;
                        const bool tl = RayBoxIntersect(origin, invDir, s.tlas[2 * (size_t)id], s.tlas[2 * (size_t)id + 1], &t1) && t1 < PT_FLOAT_MAX;
                        const bool tr2 = RayBoxIntersect(origin, invDir, s.tlas[2 * (size_t)id + 2], s.tlas[2 * (size_t)id + 3], &t2) && t2 < PT_FLOAT_MAX;
                        keep = tl || tr2;
                    }
                }
            }
        } else
        // root-box test of BVHIntersect.glsl:32-39 with T = FLOAT_MAX (no lights): a ray that fails it for every instance is a miss
        for (int ii = 0; ii < s.instanceCount && (ii == 0 || cull); ii++) {
            GpuBlasInstance inst = s.instances[ii];
            M34 inv = load_inv_model(s, inst.MeshTransformId);
            f3 l0 = xform34(inv, origin, 1.0f), l1 = xform34(inv, rd, 0.0f);
            f3 iv = mk3(1.0f / l1.x, 1.0f / l1.y, 1.0f / l1.z);
            if (s.instanceCount == 1) { lo = l0; ld = l1; invDir = iv; }
            if (cull) {
                const float4* root = s.nodes + 2 * (size_t)s.descs[inst.BlasId].NodeOffset + 2;
                float t1;
                if (RayBoxIntersect(l0, iv, root[0], root[1], &t1) && t1 < PT_FLOAT_MAX) keep = true;
            }
        }
        f3 radiance = splat3(0.0f);
        if (keep) {
            tr.lo[rid] = make_float4(lo.x, lo.y, lo.z, 0.0f); tr.ld[rid] = make_float4(ld.x, ld.y, ld.z, 0.0f); tr.inv[rid] = make_float4(invDir.x, invDir.y, invDir.z, 0.0f);
            seedOut[rid] = seed;                                        // RNG state after ray generation, consumed by k_shade_first
        } else {
            // miss branch of FirstHit TraceRay (FirstHit/compute.glsl:225-233), evaluated right here
            f3 albedo = SampleSky(s, rd);
            radiance = radiance + albedo * splat3(1.0f);
            if (f.outputAovs) { f3 fn = CubemapFaceNormal(rd); rays.aovA[rid] = make_float4(albedo.x, albedo.y, albedo.z, 0.0f); rays.aovN[rid] = make_float4(fn.x, fn.y, fn.z, 0.0f); }
        }
        // A culled pixel's ray is finished: FinalDraw only needs its radiance.  Origin/throughput planes (32 of the 48 B) are not
        // written; the flag lets idkptDownloadRays regenerate them on demand (k_regen_culled).
        if (keep) { rays.o_ior[rid] = make_float4(origin.x, origin.y, origin.z, 1.0f); rays.thr_px[rid] = make_float4(1.0f, 1.0f, 1.0f, pd.x); }
        contFlag[rid] = keep ? 0 : 2;       // also resets the continue flag of this ray id (k_shade_first sets 1); pad ids stay 0 from allocation
        rays.rad_py[rid] = make_float4(radiance.x, radiance.y, radiance.z, pd.y);
    }
    // append the survivors: one atomic per 16-wave workgroup (a single counter word saturates at ~88 atomics/us)
    const unsigned long long m = __ballot(keep);
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 0) waveKeep[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t tot = 0; for (int i = 0; i < 16; i++) { uint32_t c = waveKeep[i]; waveKeep[i] = tot; tot += c; } blockBase = tot ? atomicAdd(activeCount, tot) : 0u; }
    __syncthreads();
    if (keep) activeList[blockBase + waveKeep[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = rid;
}

// k_trace2: persistent waves; every lane owns one ray at a time and is refilled from the work list as soon as enough
// lanes of the wave are idle.  Node steps (BVHIntersect.glsl:43-53,81-101) run for all lanes that can step until none
// can; leaves found on the way are parked per lane and tested together afterwards (BVHIntersect.glsl:54-79).  A lane
// never takes its next node step before its own pending leaf is tested, so every ray sees exactly the reference's
// sequence of T updates and pushes: results (T, TriangleId, bary, visit counts) are bit-identical, only the interleaving
// between different rays changes.
// MODE 0: one BLAS instance, the trace-ready planes hold the BLAS-local ray.
// MODE 1: several BLAS instances without a TLAS (the reference's default mode, BVHIntersect.glsl:275-287): every lane walks the
//         instance list itself; the trace-ready planes hold the WORLD-space ray and the per-instance RayTransform happens here.
// MODE 2: USE_TLAS (BVHIntersect.glsl:205-272): every lane walks the TLAS with its own stack (LDS rows after the BLAS rows); a
//         TLAS leaf hands its instance to the same node/leaf phases (no root test, :32), then the TLAS walk resumes.
template <bool PRIMARY, bool COUNT, int REFILL_MIN = 32, int OCC = 1, bool PROF = false, int LEAF_MIN = 24, int MODE = 0>
__global__ __launch_bounds__(WAVE, OCC) void k_trace2(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter, uint64_t* counters)
{
    constexpr bool MULTI = MODE != 0, TLAS = MODE == 2;
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t* stk = lds + lane;
    const int cap = f.stackCap;
    uint32_t* tstk = stk + cap * WAVE;     // TLAS only
    const uint32_t N = *countPtr;
    // wave-uniform scene constants
    const GpuBlasInstance inst = s.instances[0];
    const int nodeOffset = s.descs[inst.BlasId].NodeOffset;
    const uint32_t triOffset = (uint32_t)s.descs[inst.BlasId].TriangleOffset;
    const float4* nodes = s.nodes + 2 * (size_t)nodeOffset;

    bool active = false, leafPending = false, workLeft = true;
    uint32_t top = 0, slot = 0, leafFirst = 0, leafEnd = 0;
    uint32_t instIdx = 0, rayId = 0, nodeOff = 0, triOff = 0, xformId = 0;   // MULTI only: per-lane instance cursor (TLAS: next TLAS node) and BLAS offsets
    int tsp = 0; bool moreInst = false;                                       // TLAS stack pointer; "there are instances / TLAS nodes left for this ray"
    int sp = 0;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, hbx = 0.0f, hby = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;
    uint32_t nPairs = 0, nTris = 0;
    // PROF: per-wave cycle buckets [refill, node, leaf, other], step counts and active-lane sums (developer instrumentation)
    unsigned long long pc[4] = {0, 0, 0, 0}, pn[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tPrev = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
#define PROF_MARK(b) do { if (PROF) { unsigned long long _t = __builtin_amdgcn_s_memtime(); pc[b] += _t - tPrev; tPrev = _t; } } while (0)

    while (true) {
        PROF_MARK(3);
        // ---- refill idle lanes
        unsigned long long idle = __ballot(!active);
        if (workLeft && ((uint32_t)__popcll(idle) >= REFILL_MIN || idle == ~0ull)) {
            const uint32_t n = (uint32_t)__popcll(idle);
            if (PROF) { pn[0]++; pn[1] += n; }
            const uint32_t base = wave_grab(workCounter, n);      // (chunked grabbing was measured: no gain, worse balance at small N)
            const uint32_t item = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (base + n >= N) workLeft = false;
            if (!active && item < N) {
                const uint32_t idx = list[item];
                slot = PRIMARY ? idx : item;
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f;
                if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                    float4 o = rays.o_ior[idx];
                    f3 wd = DecodeUnitVec(rays.thr_px[idx].w, rays.rad_py[idx].w), wo = mk3(o.x, o.y, o.z);
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                if (MULTI) { rayId = idx; instIdx = 0; tsp = 0; moreInst = TLAS ? s.tlasCount > 0 : true; active = true; leafPending = false; sp = 0; top = 0u; }
                else {
                    // local-space ray and 1/dir were prepared by the (coherent, full-lane) kernel that produced this ray
                    { float4 a = tr.lo[idx], b = tr.ld[idx], c = tr.inv[idx]; ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z); }
                    float t1;
                    bool enter = RayBoxIntersect(ro, invDir, nodes[2], nodes[3], &t1) && t1 < hitT; // root test (:32-39)
                    active = true; leafPending = false; sp = 0; top = enter ? 2u : 0u;
                }
            }
        }
        PROF_MARK(0);
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        if (TLAS) {
            // lanes whose current BLAS is exhausted continue their TLAS walk until it reaches the next leaf (= instance) or ends
            bool adv = active && !leafPending && top == 0u && moreInst;
            while (__any(adv)) {
                if (adv) {
                    const float4 pmin = s.tlas[2 * (size_t)instIdx];
                    const uint32_t packed = __float_as_uint(pmin.w), id = packed & 0x7fffffffu;
                    if ((packed >> 31) == 1u) {                                             // leaf: BVHIntersect.glsl:223-240
                        const GpuBlasInstance in2 = s.instances[id];
                        const M34 inv = load_inv_model(s, in2.MeshTransformId);
                        float4 a = tr.lo[rayId], b = tr.ld[rayId];
                        ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                        invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        nodeOff = (uint32_t)s.descs[in2.BlasId].NodeOffset; triOff = (uint32_t)s.descs[in2.BlasId].TriangleOffset; xformId = in2.MeshTransformId;
                        sp = 0; top = 2u;                                                   // no root test under USE_TLAS (:32)
                        if (tsp == 0) moreInst = false; else instIdx = tstk[--tsp * WAVE];  // the pop the reference does after the BLAS; order-independent
                    } else {
                        const uint32_t l = id, r = id + 1;
                        float4 a = tr.lo[rayId], c = tr.inv[rayId];                         // world-space origin and 1/dir
                        const f3 wo = mk3(a.x, a.y, a.z), winv = mk3(c.x, c.y, c.z);
                        float4 lmin = s.tlas[2 * (size_t)l], lmax = s.tlas[2 * (size_t)l + 1], rmin = s.tlas[2 * (size_t)r], rmax = s.tlas[2 * (size_t)r + 1];
                        float tMinLeft, tMinRight;
                        const bool tl = RayBoxIntersect(wo, winv, lmin, lmax, &tMinLeft) && tMinLeft < hitT;
                        const bool tr2 = RayBoxIntersect(wo, winv, rmin, rmax, &tMinRight) && tMinRight < hitT;
                        if (tl || tr2) {
                            if (tl && tr2) { const bool lc = tMinLeft < tMinRight; instIdx = lc ? l : r; if (tsp < f.tlasCap) tstk[tsp * WAVE] = lc ? r : l; tsp++; }
                            else instIdx = tl ? l : r;
                        } else { if (tsp == 0) moreInst = false; else instIdx = tstk[--tsp * WAVE]; }
                    }
                }
                adv = active && !leafPending && top == 0u && moreInst;
            }
        } else if (MULTI) {
            // lanes whose current BLAS is exhausted move on to the next instance (loop: the root test may fail right away)
            bool adv = active && !leafPending && top == 0u && instIdx < (uint32_t)s.instanceCount;
            while (__any(adv)) {
                if (adv) {
                    const GpuBlasInstance in2 = s.instances[instIdx];
                    const M34 inv = load_inv_model(s, in2.MeshTransformId);
                    float4 a = tr.lo[rayId], b = tr.ld[rayId];                         // world-space origin / direction
                    ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                    invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                    nodeOff = (uint32_t)s.descs[in2.BlasId].NodeOffset; triOff = (uint32_t)s.descs[in2.BlasId].TriangleOffset; xformId = in2.MeshTransformId;
                    const float4* root = s.nodes + 2 * (size_t)nodeOff + 2;
                    float t1;
                    const bool enter = RayBoxIntersect(ro, invDir, root[0], root[1], &t1) && t1 < hitT;
                    sp = 0; top = enter ? 2u : 0u;
                    instIdx++;
                }
                adv = active && !leafPending && top == 0u && instIdx < (uint32_t)s.instanceCount;
            }
        }

        // ---- node phase
        while (true) {
            const bool canStep = active && !leafPending && top != 0u;
            if (!__any(canStep)) break;
            // enough lanes are parked on a leaf: test those leaves now instead of letting the stragglers run on alone
            if (LEAF_MIN <= 64 && (int)__popcll(__ballot(active && leafPending)) >= LEAF_MIN) break;
            if (PROF) { pn[2]++; pn[3] += (unsigned long long)__popcll(__ballot(canStep)); }
            if (canStep) {
                if (COUNT) nPairs++;
                const float4* p = MULTI ? s.nodes + 2 * ((size_t)nodeOff + top) : nodes + 2 * (size_t)top;
                float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
                const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
                float tMinLeft, tMinRight;
                const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hitT;
                const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hitT;
                const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
                if (intersectLeft || intersectRight) {
                    const uint32_t tOff = MULTI ? triOff : triOffset;
                    leafFirst = (intersectLeft ? lStart : rStart) + tOff;
                    leafEnd = (!intersectRight ? (lStart + lCount) : (rStart + rCount)) + tOff;
                    leafPending = true;
                    if (COUNT) nTris += leafEnd - leafFirst;
                }
                const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
                if (traverseLeft || traverseRight) {
                    if (traverseLeft && traverseRight) {
                        const bool leftCloser = tMinLeft < tMinRight;
                        top = leftCloser ? lStart : rStart;
                        if (sp < cap) stk[sp * WAVE] = leftCloser ? rStart : lStart;
                        sp++;
                    } else top = traverseLeft ? lStart : rStart;
                } else {
                    if (sp == 0) top = 0u;
                    else { sp--; top = stk[sp * WAVE]; }
                }
            }
        }
        PROF_MARK(1);
        if (PROF) { unsigned long long lm = __ballot(leafPending); if (lm) { pn[4]++; pn[5] += (unsigned long long)__popcll(lm); } }
        // ---- leaf phase
        if (leafPending) {
            for (uint32_t i = leafFirst; i < leafEnd; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)i;
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hitT) {
                    hitTri = i; hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = MULTI ? xformId : inst.MeshTransformId;
                }
            }
            leafPending = false;
        }
        PROF_MARK(2);
        // ---- retire finished rays (MULTI: only after the last instance)
        if (active && top == 0u && (!MULTI || (TLAS ? !moreInst : instIdx >= (uint32_t)s.instanceCount))) {
            hits.hit[slot] = make_float4(hitT, hbx, hby, __uint_as_float(hitTri));
            hits.xformId[slot] = hitXform;
            active = false;
        }
    }
    if (COUNT) flush_counters(counters, nPairs, nTris);
    if (PROF && lane == 0) { for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)&counters[4 + i], pc[i]); for (int i = 0; i < 6; i++) atomicAdd((unsigned long long*)&counters[8 + i], pn[i]); }
#undef PROF_MARK
}

// FirstHit / NHit part 2: shade + BSDF sample + continue decision.  One thread per queue slot; the continue bits of a
// wave are published as one 64-bit ballot + popcount for the ordered compaction that follows.
// local-space ray + 1/dir of a continuing ray, for the next traversal launch (single-instance fast path): exactly what
// NHit does first (decode the packed direction, NHit:93; RayTransform, BVHIntersect.glsl:281-282; 1/dir, IntersectionRoutines.glsl:29)
DEV void write_trace_ready(const DScene& s, const Frame& f, const TraceBufs& tr, uint32_t rid, const RayState& r)
{
    f3 rd = DecodeUnitVec(r.pdx, r.pdy);
    if (f.useTlas || s.instanceCount > 1) {   // the traversal kernel walks the TLAS / instance list and transforms the world ray itself
        tr.lo[rid] = make_float4(r.origin.x, r.origin.y, r.origin.z, 0.0f); tr.ld[rid] = make_float4(rd.x, rd.y, rd.z, 0.0f);
        if (f.useTlas) tr.inv[rid] = make_float4(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z, 0.0f);   // world 1/dir for the TLAS slab tests (:209)
        return;
    }
    GpuBlasInstance inst = s.instances[0];
    M34 inv = load_inv_model(s, inst.MeshTransformId);
    f3 lo = xform34(inv, r.origin, 1.0f), ld = xform34(inv, rd, 0.0f);
    tr.lo[rid] = make_float4(lo.x, lo.y, lo.z, 0.0f); tr.ld[rid] = make_float4(ld.x, ld.y, ld.z, 0.0f);
    tr.inv[rid] = make_float4(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z, 0.0f);
}

// Fast-path FirstHit shading: only the rays that entered the traversal (active list, any order).  The continue decision
// goes to a per-ray byte (pre-zeroed), which the ordered compaction turns back into pixel order.
__global__ __launch_bounds__(256) void k_shade_first(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* activeList, const uint32_t* activeCount,
                                                     uint8_t* contFlag, uint32_t* seedsAndKeys)
{
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= *activeCount) return;
    const uint32_t rid = activeList[item];
    const uint32_t smp = rid / f.Npad, pix = rid - smp * f.Npad;
    const uint32_t acc = f.accum[smp];
    float4 a = rays.o_ior[rid], b = rays.thr_px[rid], c = rays.rad_py[rid];
    float4 h = hits.hit[rid];
    HitRec hit; hit.T = h.x; hit.bx = h.y; hit.by = h.z; hit.tri = __float_as_uint(h.w); hit.xform = hits.xformId[rid];
    RayState r; r.origin = mk3(a.x, a.y, a.z); r.prevIor = a.w; r.throughput = mk3(b.x, b.y, b.z); r.pdx = b.w; r.radiance = mk3(c.x, c.y, c.z); r.pdy = c.w;
    AovState aov; aov.albedo = splat3(0.0f); aov.normal = splat3(0.0f); aov.newWeight = 1.0f;
    uint32_t rng = seedsAndKeys[rid], key = 0;
    int lx = (int)(pix % (uint32_t)f.W), ly = (int)(pix / (uint32_t)f.W);
    uint32_t gidSeed = first_hit_gid_seed(f.W, f.H, lx, ly * f.rowMod + f.rowRem);
    f3 rd = DecodeUnitVec(r.pdx, r.pdy);
    bool cont = ShadeHit<true>(s, f, acc, hit, hit.T != PT_FLOAT_MAX, rd, r, aov, rng, gidSeed, key);
    rays.o_ior[rid] = make_float4(r.origin.x, r.origin.y, r.origin.z, r.prevIor);
    rays.thr_px[rid] = make_float4(r.throughput.x, r.throughput.y, r.throughput.z, r.pdx);
    rays.rad_py[rid] = make_float4(r.radiance.x, r.radiance.y, r.radiance.z, r.pdy);
    if (f.outputAovs) { rays.aovA[rid] = make_float4(aov.albedo.x, aov.albedo.y, aov.albedo.z, aov.newWeight); rays.aovN[rid] = make_float4(aov.normal.x, aov.normal.y, aov.normal.z, 0.0f); }
    seedsAndKeys[rid] = (key & ((1u << IDKPT_SORT_KEY_BITS) - 1u)) | (smp << IDKPT_SORT_KEY_BITS);
    if (cont) { contFlag[rid] = 1; write_trace_ready(s, f, tr, rid, r); }
}

template <bool FIRST>
__global__ __launch_bounds__(256) void k_shade(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* queue, const uint32_t* countPtr, uint32_t countImm,
                                               const uint32_t* qbase, uint32_t slotBase, unsigned long long* contMask, uint32_t* waveCounts, uint32_t* keysTmp)
{
    // FIRST: slots are ray ids (sample-major, Npad per sample, Npad % 64 == 0).  Otherwise slots are positions of the
    // batch-wide alive queue, which is grouped by sample; qbase[k] = first slot of sample k.
    const uint32_t N = FIRST ? countImm : *countPtr;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if ((slot & ~63u) >= N) return; // whole wave out of range
    bool cont = false;
    uint32_t key = 0;
    bool inRange = slot < N;
    uint32_t smp = 0, pix = 0, idx = 0;
    if (inRange) {
        idx = FIRST ? slot : queue[slot];
        smp = idx / f.Npad; pix = idx - smp * f.Npad;
        if (FIRST && pix >= (uint32_t)f.W * (uint32_t)f.rows) inRange = false; // padding of the sample segment
    }
    if (inRange) {
        const uint32_t acc = f.accum[smp];
        float4 a = rays.o_ior[idx], b = rays.thr_px[idx], c = rays.rad_py[idx];
        float4 h = hits.hit[slot];
        HitRec hit; hit.T = h.x; hit.bx = h.y; hit.by = h.z; hit.tri = __float_as_uint(h.w); hit.xform = hits.xformId[slot];
        if (FIRST && f.g.DoDebugBVHTraversal) {
            rays.o_ior[idx] = make_float4(a.x, a.y, a.z, hits.cost[slot]); // FirstHit:108-112
        } else {
            RayState r; r.origin = mk3(a.x, a.y, a.z); r.prevIor = a.w; r.throughput = mk3(b.x, b.y, b.z); r.pdx = b.w; r.radiance = mk3(c.x, c.y, c.z); r.pdy = c.w;
            AovState aov; aov.albedo = splat3(0.0f); aov.normal = splat3(0.0f); aov.newWeight = 1.0f;
            if (!FIRST && f.outputAovs) { float4 aa = rays.aovA[idx], an = rays.aovN[idx]; aov.albedo = mk3(aa.x, aa.y, aa.z); aov.newWeight = aa.w; aov.normal = mk3(an.x, an.y, an.z); }
            uint32_t rng, gidSeed;
            if (FIRST) {
                f3 o2; f2 pd2; gen_primary(f, pix, acc, o2, pd2, rng); // re-derives the RNG state after ray generation (cheaper than 4 B/pixel of HBM)
                int lx = (int)(pix % (uint32_t)f.W), ly = (int)(pix / (uint32_t)f.W);
                gidSeed = first_hit_gid_seed(f.W, f.H, lx, ly * f.rowMod + f.rowRem);
            } else {
                uint32_t gslot = slotBase + (slot - qbase[smp]);  // slot inside this sample's own queue
                rng = gslot * 4096u + acc;            // NHit:54
                gidSeed = gslot;                      // Shading.glsl:74 with gl_GlobalInvocationID = (slot, 0)
            }
            f3 rd = DecodeUnitVec(r.pdx, r.pdy);
            bool hitScene = hit.T != PT_FLOAT_MAX;
            cont = ShadeHit<FIRST>(s, f, acc, hit, hitScene, rd, r, aov, rng, gidSeed, key);
            rays.o_ior[idx] = make_float4(r.origin.x, r.origin.y, r.origin.z, r.prevIor);
            rays.thr_px[idx] = make_float4(r.throughput.x, r.throughput.y, r.throughput.z, r.pdx);
            rays.rad_py[idx] = make_float4(r.radiance.x, r.radiance.y, r.radiance.z, r.pdy);
            if (f.outputAovs) { rays.aovA[idx] = make_float4(aov.albedo.x, aov.albedo.y, aov.albedo.z, aov.newWeight); rays.aovN[idx] = make_float4(aov.normal.x, aov.normal.y, aov.normal.z, 0.0f); }
            if (cont && tr.lo) write_trace_ready(s, f, tr, idx, r);
        }
        // NHit:81 masks the key to 21 bits; the sample index goes above it so that the batch-wide sort stays grouped by sample
        keysTmp[slot] = (key & ((1u << IDKPT_SORT_KEY_BITS) - 1u)) | (smp << IDKPT_SORT_KEY_BITS);
    }
    unsigned long long m = __ballot(cont);
    if ((threadIdx.x & 63) == 0) contMask[slot >> 6] = m;
    (void)waveCounts;
}

// Ordered (= sequential enqueue order) compaction in three launches: (1) per 256-wave block: continue masks (from the
// shade kernel's ballots, or rebuilt from per-ray bytes) -> exclusive offsets inside the block + block total,
// (2) one workgroup scans the block totals and derives the next queue length and every sample's first slot,
// (3) scatter (k_compact).
#define SCAN_WAVES_PER_BLOCK 256
template <bool FROM_FLAGS>
__global__ __launch_bounds__(SCAN_WAVES_PER_BLOCK) void k_scan_local(const uint32_t* countPtr, uint32_t countImm, const uint8_t* contFlag, unsigned long long* contMask, uint32_t* waveLocal, uint32_t* blockSums)
{
    __shared__ uint32_t part[SCAN_WAVES_PER_BLOCK];
    const uint32_t N = countPtr ? *countPtr : countImm;
    const uint32_t nW = (N + 63) / 64;
    const uint32_t w = blockIdx.x * SCAN_WAVES_PER_BLOCK + threadIdx.x;
    if (blockIdx.x * SCAN_WAVES_PER_BLOCK >= nW) return;
    uint32_t c = 0;
    if (w < nW) {
        unsigned long long m;
        if (FROM_FLAGS) {
            const uint4* p = reinterpret_cast<const uint4*>(contFlag + (size_t)w * 64);
            m = 0ull;
            for (int q = 0; q < 4; q++) {
                uint4 v = p[q]; uint32_t d[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 4; k++) for (int bb = 0; bb < 4; bb++) m |= (unsigned long long)((d[k] >> (8 * bb)) & 1u) << (q * 16 + k * 4 + bb);
            }
            contMask[w] = m;
        } else m = contMask[w];
        c = (uint32_t)__popcll(m);
    }
    part[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < SCAN_WAVES_PER_BLOCK; off <<= 1) { uint32_t v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0; __syncthreads(); part[threadIdx.x] += v; __syncthreads(); }
    if (w < nW) waveLocal[w] = part[threadIdx.x] - c;
    if (threadIdx.x == SCAN_WAVES_PER_BLOCK - 1) blockSums[blockIdx.x] = part[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* countPtr, uint32_t countImm, uint32_t* blockSums, const uint32_t* waveLocal, uint32_t* nextCount, unsigned long long* tracedRays,
                                                      const unsigned long long* contMask, const uint32_t* curBase /* null: FIRST (slots are ray ids) */, uint32_t Npad, int batch, uint32_t* nextBase)
{
    __shared__ uint32_t part[1024];
    const uint32_t N = countPtr ? *countPtr : countImm;
    const uint32_t nW = (N + 63) / 64, nB = (nW + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK;
    const uint32_t per = (nB + 1023) / 1024;
    const uint32_t t = threadIdx.x;
    uint32_t b = t * per, e = min(b + per, nB);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += blockSums[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { uint32_t v = (t >= off) ? part[t - off] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    uint32_t run = part[t] - sum;
    for (uint32_t i = b; i < e; i++) { uint32_t c = blockSums[i]; blockSums[i] = run; run += c; }   // blockSums becomes blockBase
    if (t == 1023) { *nextCount = part[1023]; if (tracedRays) atomicAdd(tracedRays, (unsigned long long)part[1023]); }
    __threadfence_block();
    __syncthreads();
    // first slot of every sample in the NEXT queue = number of survivors in front of the sample's first current slot
    if (t <= (uint32_t)batch) {
        uint32_t g = (t == (uint32_t)batch) ? N : (curBase ? curBase[t] : t * Npad);
        g = min(g, N);
        uint32_t w = g >> 6, l = g & 63;
        uint32_t v = part[1023];
        if (w < nW) v = blockSums[w / SCAN_WAVES_PER_BLOCK] + waveLocal[w] + (uint32_t)__popcll(contMask[w] & ((1ull << l) - 1ull));
        nextBase[t] = v;
    }
}

// Scatter of the surviving ray indices (and their sort keys) to their ordered slots.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_compact(const uint32_t* queue, const uint32_t* countPtr, uint32_t countImm, const unsigned long long* contMask, const uint32_t* waveOffsets, const uint32_t* blockBase,
                                                 const uint32_t* keysTmp, uint32_t* queueNext, uint32_t* keysNext)
{
    const uint32_t N = FIRST ? countImm : *countPtr;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= N) return;
    const uint32_t w = slot >> 6, lane = slot & 63;
    unsigned long long m = contMask[w];
    if ((m >> lane) & 1ull) {
        uint32_t dst = blockBase[w / SCAN_WAVES_PER_BLOCK] + waveOffsets[w] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        queueNext[dst] = FIRST ? slot : queue[slot];
        keysNext[dst] = keysTmp[slot];
    }
}

// ---- stable LSD radix sort of the alive queue by the 21-bit key (replaces CountingSort/**; PathTracer.cs:273-297).
// 3 passes x 7 bits.  Per pass: (1) per-block digit histogram, (2) exclusive scan over [digit][block], (3) stable scatter.
#define SORT_BLOCK 256
#define SORT_ITEMS 8                      // items per thread
#define SORT_TILE (SORT_BLOCK * SORT_ITEMS)
#define SORT_RADIX 128
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_hist(const uint32_t* keys, const uint32_t* countPtr, uint32_t shift, uint32_t* hist /*[RADIX][numTiles]*/, uint32_t numTilesMax)
{
    __shared__ uint32_t h[SORT_RADIX];
    const uint32_t N = *countPtr;
    const uint32_t tile = blockIdx.x;
    if (tile * SORT_TILE >= N) return;
    if (threadIdx.x < SORT_RADIX) h[threadIdx.x] = 0;
    __syncthreads();
    for (int k = 0; k < SORT_ITEMS; k++) { uint32_t i = tile * SORT_TILE + k * SORT_BLOCK + threadIdx.x; if (i < N) atomicAdd(&h[(keys[i] >> shift) & (SORT_RADIX - 1)], 1u); }
    __syncthreads();
    if (threadIdx.x < SORT_RADIX) hist[threadIdx.x * numTilesMax + tile] = h[threadIdx.x];
}
// one workgroup per digit: exclusive scan of that digit's per-tile counts (a contiguous row) + the digit's total.  The scatter
// kernel adds the exclusive prefix over the 128 digit totals itself, so the global offset of (digit, tile) is
// sum(totals[0..digit)) + row prefix — the same value a single serial scan over [digit][tile] would give.
__global__ __launch_bounds__(1024) void k_sort_scan(const uint32_t* countPtr, uint32_t* hist, uint32_t numTilesMax, uint32_t* digitTotals)
{
    __shared__ uint32_t part[1024];
    const uint32_t N = *countPtr;
    const uint32_t nT = (N + SORT_TILE - 1) / SORT_TILE;
    uint32_t* row = hist + (size_t)blockIdx.x * numTilesMax;
    const uint32_t per = (nT + 1023) / 1024;
    const uint32_t t = threadIdx.x;
    const uint32_t b = min(t * per, nT), e = min(b + per, nT);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += row[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { uint32_t v = (t >= off) ? part[t - off] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    uint32_t run = part[t] - sum;
    for (uint32_t i = b; i < e; i++) { uint32_t c = row[i]; row[i] = run; run += c; }
    if (t == 1023) digitTotals[blockIdx.x] = part[1023];
}
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scatter(const uint32_t* keys, const uint32_t* vals, const uint32_t* countPtr, uint32_t shift, const uint32_t* hist, uint32_t numTilesMax,
                                                             const uint32_t* digitTotals, uint32_t* keysOut, uint32_t* valsOut)
{
    // stable within the tile: items are visited in index order (k-major, then wave, then lane)
    __shared__ uint32_t digitBase[SORT_RADIX];
    __shared__ uint32_t waveDigit[SORT_BLOCK / 64][SORT_RADIX];
    const uint32_t N = *countPtr;
    const uint32_t tile = blockIdx.x;
    if (tile * SORT_TILE >= N) return;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < SORT_RADIX) {
        uint32_t base = 0;
        for (uint32_t d2 = 0; d2 < threadIdx.x; d2++) base += digitTotals[d2];       // exclusive prefix over the digit totals
        digitBase[threadIdx.x] = base + hist[(size_t)threadIdx.x * numTilesMax + tile];
    }
    for (int k = 0; k < SORT_ITEMS; k++) {
        uint32_t i = tile * SORT_TILE + k * SORT_BLOCK + threadIdx.x;
        bool valid = i < N;
        uint32_t key = valid ? keys[i] : 0, val = valid ? vals[i] : 0;
        uint32_t d = (key >> shift) & (SORT_RADIX - 1);
        // rank among lanes of this wave with the same digit (match via 7 ballots)
        unsigned long long same = __ballot(valid);
        for (int bit = 0; bit < 7; bit++) { unsigned long long bm = __ballot((d >> bit) & 1u); same &= ((d >> bit) & 1u) ? bm : ~bm; }
        uint32_t rankInWave = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        uint32_t cntInWave = (uint32_t)__popcll(same);
        for (uint32_t x = threadIdx.x; x < (SORT_BLOCK / 64) * SORT_RADIX; x += SORT_BLOCK) (&waveDigit[0][0])[x] = 0;
        __syncthreads();
        if (valid && rankInWave == 0) waveDigit[wv][d] = cntInWave;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w2 = 0; w2 < wv; w2++) before += waveDigit[w2][d];
        uint32_t dst = digitBase[d] + before + rankInWave;
        if (valid) { keysOut[dst] = key; valsOut[dst] = val; }
        __syncthreads();
        if (threadIdx.x < SORT_RADIX) { uint32_t tot = 0; for (uint32_t w2 = 0; w2 < SORT_BLOCK / 64; w2++) tot += waveDigit[w2][threadIdx.x]; digitBase[threadIdx.x] += tot; }
        __syncthreads();
    }
}

// FinalDraw/compute.glsl:24-62
DEV f3 TurboColormap(float x)
{
    x = gclamp(x, 0.0f, 1.0f);
    float v0 = 1.0f, v1 = x, v2 = x * x, v3 = x * x * x;
    float w0 = v2 * v2, w1 = v3 * v2;
    float r = (((v0 * 0.13572138f + v1 * 4.61539260f) + v2 * -42.66032258f) + v3 * 132.13108234f) + (w0 * -152.94239396f + w1 * 59.28637943f);
    float g = (((v0 * 0.09140261f + v1 * 2.19418839f) + v2 * 4.84296658f) + v3 * -14.18503333f) + (w0 * 4.27729857f + w1 * 2.82956604f);
    float b = (((v0 * 0.10667330f + v1 * 12.64194608f) + v2 * -60.58204836f) + v3 * 110.36276771f) + (w0 * -89.90310912f + w1 * 27.34824973f);
    return mk3(r, g, b);
}
__global__ __launch_bounds__(256) void k_final_draw(Frame f, RayBufs rays, float4* imgResult, float4* imgAlbedo, float4* imgNormal, uint32_t N)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 o = imgResult[i];
    f3 r = mk3(o.x, o.y, o.z), ra = splat3(0.0f), rn = splat3(0.0f);
    if (f.outputAovs) { float4 oa = imgAlbedo[i], on = imgNormal[i]; ra = mk3(oa.x, oa.y, oa.z); rn = mk3(on.x, on.y, on.z); }
    for (int k = 0; k < f.batch; k++) {                 // samples are accumulated in submission order, exactly like consecutive FinalDraw dispatches
        const size_t rid = (size_t)k * f.Npad + i;
        float w = 1.0f / ((float)f.accum[k] + 1.0f);
        float4 c = rays.rad_py[rid];
        f3 nr = mk3(c.x, c.y, c.z);
        if (f.g.DoDebugBVHTraversal) nr = TurboColormap(rays.o_ior[rid].w / 150.0f);
        r = gmix(r, nr, w);
        if (f.outputAovs) { float4 a = rays.aovA[rid], n = rays.aovN[rid]; ra = gmix(ra, mk3(a.x, a.y, a.z), w); rn = gmix(rn, mk3(n.x, n.y, n.z), w); }
    }
    imgResult[i] = make_float4(r.x, r.y, r.z, 1.0f);
    if (f.outputAovs) { imgAlbedo[i] = make_float4(ra.x, ra.y, ra.z, 1.0f); imgNormal[i] = make_float4(rn.x, rn.y, rn.z, 1.0f); }
}

// idkptDownloadRays support: the ray-state planes k_gen_primary skipped for culled pixels (flag 2) of one sample of the batch
__global__ __launch_bounds__(256) void k_regen_culled(Frame f, RayBufs rays, const uint8_t* contFlag, uint32_t smp, uint32_t N)
{
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= N) return;
    const size_t rid = (size_t)smp * f.Npad + pix;
    if (contFlag[rid] != 2) return;
    f3 origin; f2 pd; uint32_t seed;
    gen_primary(f, pix, f.accum[smp], origin, pd, seed);
    rays.o_ior[rid] = make_float4(origin.x, origin.y, origin.z, 1.0f);
    rays.thr_px[rid] = make_float4(1.0f, 1.0f, 1.0f, pd.x);
}

// test support (idkptEnablePrimaryHitCapture): miss records for the pixels the pre-cull removes before the traversal
__global__ void k_fill_miss(float4* hit, uint32_t* xform, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { hit[i] = make_float4(PT_FLOAT_MAX, 0.0f, 0.0f, __uint_as_float(~0u)); xform[i] = 0; }
}

// derived layout: positions of each BLAS triangle's vertices, in leaf order (48 B/triangle, one contiguous fetch in the leaf loop)
__global__ void k_gather_triverts(const uint4* tris, const float* positions, float4* triVerts, uint32_t first, uint32_t count)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint4 t = tris[first + i];
    const float* a = positions + 3 * (size_t)t.x; const float* b = positions + 3 * (size_t)t.y; const float* c = positions + 3 * (size_t)t.z;
    float4* o = triVerts + 3 * (size_t)(first + i);
    o[0] = make_float4(a[0], a[1], a[2], 0.0f); o[1] = make_float4(b[0], b[1], b[2], 0.0f); o[2] = make_float4(c[0], c[1], c[2], 0.0f);
}

// ---------------------------------------------------------------------------------------------------------------
// TLAS rebuild on the device: BVH.TlasBuild (Bvh/BVH.cs:278-298) + TLAS.Build (Bvh/TLAS.cs:28-141).
// One 1024-thread workgroup (instance counts are small; the reference does this serially on the CPU every animated frame and
// re-uploads): world bounds of every instance's BLAS root (Box.Transformed, Shapes/Box.cs:177-187), Morton-30 order (stable
// rank = the reference's stable LSD radix sort), then PLOC rounds: every node picks its best partner inside +-searchRadius
// (FindBestMatch, TLAS.cs:271-301, strict '<' keeps the first best), mutual pairs merge, output positions come from an ordered
// block scan so the node array is identical to the serial build, bit for bit.
#define TLAS_BUILD_THREADS 1024
DEV uint32_t tlas_insert_two_zeros(uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; }
DEV uint32_t tlas_to_uint_sat(float f) { if (f != f || f <= 0.0f) return 0u; if (f >= 4294967296.0f) return 0xffffffffu; return (uint32_t)f; }
DEV float tlas_minN(float a, float b) { return a < b ? a : b; }   // minps / float.MinNative (Shapes/Box.cs:40-50)
DEV float tlas_maxN(float a, float b) { return a > b ? a : b; }
DEV float tlas_half_area(float4 mn, float4 mx) { float sx = mx.x - mn.x, sy = mx.y - mn.y, sz = mx.z - mn.z; return __fmaf_rn(sx + sy, sz, sx * sy); }   // MyMath.cs:222-229

// ordered exclusive scan of two per-thread counters over the workgroup; returns this thread's bases and the totals
DEV void block_scan2(uint32_t a, uint32_t b, uint32_t* sa, uint32_t* sb, uint32_t& baseA, uint32_t& baseB, uint32_t& totA, uint32_t& totB)
{
    const uint32_t t = threadIdx.x;
    sa[t] = a; sb[t] = b;
    __syncthreads();
    for (uint32_t off = 1; off < TLAS_BUILD_THREADS; off <<= 1) {
        uint32_t va = t >= off ? sa[t - off] : 0u, vb = t >= off ? sb[t - off] : 0u;
        __syncthreads();
        sa[t] += va; sb[t] += vb;
        __syncthreads();
    }
    baseA = sa[t] - a; baseB = sb[t] - b; totA = sa[TLAS_BUILD_THREADS - 1]; totB = sb[TLAS_BUILD_THREADS - 1];
    __syncthreads();
}

__global__ __launch_bounds__(TLAS_BUILD_THREADS) void k_tlas_build(const float4* blasNodes, const GpuBlasDesc* descs, const GpuBlasInstance* instances, const float4* xforms,
                                                                    int n, int searchRadius, float4* nodes /* 2n-1 */, float4* temp /* 2n-1 */, float4* leaf /* n */, uint32_t* keys /* n */, int* pref /* n */)
{
    __shared__ float red[6][TLAS_BUILD_THREADS / 64];
    __shared__ float gbox[6];
    __shared__ uint32_t sa[TLAS_BUILD_THREADS], sb[TLAS_BUILD_THREADS];
    const int t = (int)threadIdx.x, T = TLAS_BUILD_THREADS;
    const int nodeCount = 2 * n - 1;
    // ---- leaves: world-space bounds of every instance
    float mn[3] = {PT_FLOAT_MAX, PT_FLOAT_MAX, PT_FLOAT_MAX}, mx[3] = {-PT_FLOAT_MAX, -PT_FLOAT_MAX, -PT_FLOAT_MAX};
    for (int i = t; i < n; i += T) {
        const GpuBlasInstance in = instances[i];
        const float4* root = blasNodes + 2 * ((size_t)descs[in.BlasId].NodeOffset + 1);
        const float4 rmin = root[0], rmax = root[1];
        const float4* x = xforms + 9 * (size_t)in.MeshTransformId;
        const float4 m0 = x[0], m1 = x[1], m2 = x[2];
        float bmn[3] = {PT_FLOAT_MAX, PT_FLOAT_MAX, PT_FLOAT_MAX}, bmx[3] = {-PT_FLOAT_MAX, -PT_FLOAT_MAX, -PT_FLOAT_MAX};
        for (int c = 0; c < 8; c++) {
            const float cx = (c & 1) ? rmax.x : rmin.x, cy = (c & 2) ? rmax.y : rmin.y, cz = (c & 4) ? rmax.z : rmin.z;
            const float w[3] = {(cx * m0.x) + (cy * m0.y) + (cz * m0.z) + (1.0f * m0.w), (cx * m1.x) + (cy * m1.y) + (cz * m1.z) + (1.0f * m1.w), (cx * m2.x) + (cy * m2.y) + (cz * m2.z) + (1.0f * m2.w)};
            for (int k = 0; k < 3; k++) { bmn[k] = tlas_minN(bmn[k], w[k]); bmx[k] = tlas_maxN(bmx[k], w[k]); }
        }
        leaf[2 * (size_t)i] = make_float4(bmn[0], bmn[1], bmn[2], __uint_as_float((1u << 31) | (uint32_t)i));
        leaf[2 * (size_t)i + 1] = make_float4(bmx[0], bmx[1], bmx[2], 0.0f);
        for (int k = 0; k < 3; k++) { mn[k] = tlas_minN(mn[k], bmn[k]); mx[k] = tlas_maxN(mx[k], bmx[k]); }
    }
    // global box (min/max: order independent)
    for (int k = 0; k < 3; k++) {
        float a = mn[k], b = mx[k];
        for (int off = 32; off > 0; off >>= 1) { a = tlas_minN(a, __shfl_xor(a, off)); b = tlas_maxN(b, __shfl_xor(b, off)); }
        if ((t & 63) == 0) { red[k][t >> 6] = a; red[3 + k][t >> 6] = b; }
    }
    __syncthreads();
    if (t < 3) { float a = PT_FLOAT_MAX, b = -PT_FLOAT_MAX; for (int w = 0; w < T / 64; w++) { a = tlas_minN(a, red[t][w]); b = tlas_maxN(b, red[3 + t][w]); } gbox[t] = a; gbox[3 + t] = b; }
    __syncthreads();
    // ---- Morton-30 keys of the box centres (MyMath.cs:241-257, 283-299)
    for (int i = t; i < n; i += T) {
        const float4 a = leaf[2 * (size_t)i], b = leaf[2 * (size_t)i + 1];
        const float c[3] = {(b.x + a.x) * 0.5f, (b.y + a.y) * 0.5f, (b.z + a.z) * 0.5f};
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            const float ext = gbox[3 + k] - gbox[k];
            float r = (c[k] - gbox[k]) / ext * (1.0f - 0.0f) + 0.0f;
            if (ext == 0.0f) r = 0.0f;
            const uint32_t u = tlas_to_uint_sat(r * 1024.0f);
            q[k] = u < 1023u ? u : 1023u;
        }
        keys[i] = (tlas_insert_two_zeros(q[0]) << 2) | (tlas_insert_two_zeros(q[1]) << 1) | tlas_insert_two_zeros(q[2]);
    }
    __syncthreads();
    // ---- stable sort by key (rank counting) into the tail of the node array
    for (int i = t; i < n; i += T) {
        const uint32_t ki = keys[i];
        int rank = 0;
        for (int j = 0; j < n; j++) { const uint32_t kj = keys[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
        const size_t d = (size_t)(nodeCount - n + rank);
        nodes[2 * d] = leaf[2 * (size_t)i]; nodes[2 * d + 1] = leaf[2 * (size_t)i + 1];
    }
    __syncthreads();
    // ---- PLOC rounds
    int activeCount = n, activeEnd = nodeCount;
    while (activeCount > 1) {
        const int start = activeEnd - activeCount;
        for (int i = t; i < activeCount; i += T) {
            const int a = start + i;
            const int s0 = max(a - searchRadius, start), s1 = min(a + searchRadius + 1, activeEnd);
            const float4 amn = nodes[2 * (size_t)a], amx = nodes[2 * (size_t)a + 1];
            float smallest = PT_FLOAT_MAX; int best = -1;
            for (int k = s0; k < s1; k++) {
                if (k == a) continue;
                const float4 omn = nodes[2 * (size_t)k], omx = nodes[2 * (size_t)k + 1];
                const float4 un = make_float4(tlas_minN(amn.x, omn.x), tlas_minN(amn.y, omn.y), tlas_minN(amn.z, omn.z), 0.0f);
                const float4 ux = make_float4(tlas_maxN(amx.x, omx.x), tlas_maxN(amx.y, omx.y), tlas_maxN(amx.z, omx.z), 0.0f);
                const float area = tlas_half_area(un, ux);
                if (area < smallest) { smallest = area; best = k; }
            }
            pref[i] = best - start;
        }
        __syncthreads();
        // contiguous chunk per thread so that the scan order is the serial loop's order
        const int chunk = (activeCount + T - 1) / T, c0 = min(t * chunk, activeCount), c1 = min(c0 + chunk, activeCount);
        uint32_t nPairs = 0, nOut = 0;
        for (int i = c0; i < c1; i++) { const int b = pref[i]; const bool mutual = b >= 0 && pref[b] == i; if (mutual && i < b) nPairs++; if (!mutual || i < b) nOut++; }
        uint32_t basePairs, baseOut, totPairs, totOut;
        block_scan2(nPairs, nOut, sa, sb, basePairs, baseOut, totPairs, totOut);
        const int merged = 2 * (int)totPairs, unmerged = activeCount - merged, newNodes = merged / 2;
        const int mergedHead0 = activeEnd - merged, newBegin = mergedHead0 - unmerged - newNodes;
        int mergedHead = mergedHead0 + 2 * (int)basePairs, unmergedHead = newBegin + (int)baseOut;
        for (int i = c0; i < c1; i++) {
            const int b = pref[i]; const bool mutual = b >= 0 && pref[b] == i; const size_t aId = (size_t)(i + start);
            if (mutual) {
                if (i < b) {
                    const size_t bId = (size_t)(b + start);
                    const float4 amn = nodes[2 * aId], amx = nodes[2 * aId + 1], bmn = nodes[2 * bId], bmx = nodes[2 * bId + 1];
                    temp[2 * (size_t)mergedHead] = amn; temp[2 * (size_t)mergedHead + 1] = amx; temp[2 * (size_t)mergedHead + 2] = bmn; temp[2 * (size_t)mergedHead + 3] = bmx;
                    temp[2 * (size_t)unmergedHead] = make_float4(tlas_minN(amn.x, bmn.x), tlas_minN(amn.y, bmn.y), tlas_minN(amn.z, bmn.z), __uint_as_float((uint32_t)mergedHead));
                    temp[2 * (size_t)unmergedHead + 1] = make_float4(tlas_maxN(amx.x, bmx.x), tlas_maxN(amx.y, bmx.y), tlas_maxN(amx.z, bmx.z), 0.0f);
                    unmergedHead++; mergedHead += 2;
                }
            } else { temp[2 * (size_t)unmergedHead] = nodes[2 * aId]; temp[2 * (size_t)unmergedHead + 1] = nodes[2 * aId + 1]; unmergedHead++; }
        }
        __syncthreads();
        for (int i = newBegin + t; i < activeEnd; i += T) { nodes[2 * (size_t)i] = temp[2 * (size_t)i]; nodes[2 * (size_t)i + 1] = temp[2 * (size_t)i + 1]; }
        __syncthreads();
        activeCount -= merged / 2; activeEnd -= merged;
    }
}

// BLAS refit (Shaders/BLASRefit/compute.glsl).  The reference walks leaf->root inside one dispatch behind an
// atomicExchange "second arrival" lock; here the same unions are evaluated level by level (deepest first), one launch
// per level, so no workgroup ever consumes another workgroup's stores inside a launch (per-XCD L2s are not coherent).
__global__ void k_refit_leaves(float4* nodes, const uint4* tris, const float4* triVerts, const int32_t* leafIds, uint32_t leafCount, uint32_t nodeOffset, uint32_t triOffset)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= leafCount) return;
    uint32_t id = nodeOffset + (uint32_t)leafIds[i];
    float4 mn = nodes[2 * (size_t)id], mx = nodes[2 * (size_t)id + 1];
    uint32_t start = triOffset + __float_as_uint(mn.w), count = __float_as_uint(mx.w);
    f3 bmin = splat3(PT_FLOAT_MAX), bmax = splat3(-PT_FLOAT_MAX);
    for (uint32_t k = start; k < start + count; k++) {
        for (int v = 0; v < 3; v++) { float4 p = triVerts[3 * (size_t)k + v]; bmin = mk3(gmin(bmin.x, p.x), gmin(bmin.y, p.y), gmin(bmin.z, p.z)); bmax = mk3(gmax(bmax.x, p.x), gmax(bmax.y, p.y), gmax(bmax.z, p.z)); }
    }
    nodes[2 * (size_t)id] = make_float4(bmin.x, bmin.y, bmin.z, mn.w); nodes[2 * (size_t)id + 1] = make_float4(bmax.x, bmax.y, bmax.z, mx.w);
}
__global__ void k_refit_level(float4* nodes, const int32_t* levelNodes, uint32_t count, uint32_t nodeOffset)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t id = nodeOffset + (uint32_t)levelNodes[i];
    float4 mn = nodes[2 * (size_t)id], mx = nodes[2 * (size_t)id + 1];
    uint32_t child = nodeOffset + __float_as_uint(mn.w);
    float4 lmn = nodes[2 * (size_t)child], lmx = nodes[2 * (size_t)child + 1], rmn = nodes[2 * (size_t)child + 2], rmx = nodes[2 * (size_t)child + 3];
    nodes[2 * (size_t)id] = make_float4(gmin(lmn.x, rmn.x), gmin(lmn.y, rmn.y), gmin(lmn.z, rmn.z), mn.w);
    nodes[2 * (size_t)id + 1] = make_float4(gmax(lmx.x, rmx.x), gmax(lmx.y, rmx.y), gmax(lmx.z, rmx.z), mx.w);
}

// Skinning (Shaders/Skinning/compute.glsl:14-47): 4-weight linear blend; joint matrices are row_major mat4x3 (3 x float4)
__global__ void k_skin(const GpuUnskinnedVertex* unskinned, const float4* joints, float* positions, float* prevPositions, uint4* vertices,
                       uint32_t inOff, uint32_t outOff, uint32_t jointOff, uint32_t count)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    GpuUnskinnedVertex u = unskinned[inOff + i];
    float4 m[3];
    for (int r = 0; r < 3; r++) {
        float4 a = joints[3 * (size_t)(jointOff + u.JointIndices[0]) + r], b = joints[3 * (size_t)(jointOff + u.JointIndices[1]) + r];
        float4 c = joints[3 * (size_t)(jointOff + u.JointIndices[2]) + r], d = joints[3 * (size_t)(jointOff + u.JointIndices[3]) + r];
        float w0 = u.JointWeights[0], w1 = u.JointWeights[1], w2 = u.JointWeights[2], w3 = u.JointWeights[3];
        m[r] = make_float4(((w0 * a.x + w1 * b.x) + w2 * c.x) + w3 * d.x, ((w0 * a.y + w1 * b.y) + w2 * c.y) + w3 * d.y,
                           ((w0 * a.z + w1 * b.z) + w2 * c.z) + w3 * d.z, ((w0 * a.w + w1 * b.w) + w2 * c.w) + w3 * d.w);
    }
    M34 M; M.r0 = m[0]; M.r1 = m[1]; M.r2 = m[2];
    f3 p = mk3(u.Position[0], u.Position[1], u.Position[2]);
    f3 n = DecompressSR11G11B10(u.Normal), t = DecompressSR11G11B10(u.Tangent);
    f3 np = xform34(M, p, 1.0f);
    // mat3(skinMatrix) * v : out_i = (R[i][0]*v.x + R[i][1]*v.y) + R[i][2]*v.z
    f3 nn = normalize(mk3((M.r0.x * n.x + M.r0.y * n.y) + M.r0.z * n.z, (M.r1.x * n.x + M.r1.y * n.y) + M.r1.z * n.z, (M.r2.x * n.x + M.r2.y * n.y) + M.r2.z * n.z));
    f3 nt = normalize(mk3((M.r0.x * t.x + M.r0.y * t.y) + M.r0.z * t.z, (M.r1.x * t.x + M.r1.y * t.y) + M.r1.z * t.z, (M.r2.x * t.x + M.r2.y * t.y) + M.r2.z * t.z));
    size_t o = (size_t)(outOff + i);
    if (prevPositions) { prevPositions[3 * o] = positions[3 * o]; prevPositions[3 * o + 1] = positions[3 * o + 1]; prevPositions[3 * o + 2] = positions[3 * o + 2]; }
    positions[3 * o] = np.x; positions[3 * o + 1] = np.y; positions[3 * o + 2] = np.z;
    uint4 v = vertices[o]; v.w = CompressSR11G11B10(nn); v.z = CompressSR11G11B10(nt); vertices[o] = v;
}

// =================================================================================================== host side

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) { if (n <= bytes && p) return hipSuccess; if (p) (void)hipFree(p); p = nullptr; bytes = 0; if (n == 0) return hipSuccess; hipError_t e = hipMalloc(&p, n); if (e == hipSuccess) bytes = n; return e; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct idkpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr; bool ownStream = true;
    std::string lastError;
    int numCUs = 256;
    // config
    idkpt_settings st;
    int W = 0, H = 0, rowMod = 1, rowRem = 0, rows = 0;
    float invProj[16], invView[16], viewPos[3];
    uint32_t accumulated = 0;
    bool counters = false, timing = false, capturePrimary = false, forceGeneric = false; int traceVariant = 0;
    // scene
    bool haveScene = false;
    DevBuf nodes, tris, triVerts, descs, instances, tlas, parents, leaves, positions, prevPositions, vertices, meshes, materials, xforms, lights, sky, texDescs, unskinned, joints, levelNodes, tlasScratch, queryIn, queryOut;
    std::vector<DevBuf> texData;
    std::vector<GpuBlasDesc> hDescs;
    std::vector<std::vector<uint32_t>> levelOffsets; // per BLAS: offsets into levelNodes (level l occupies [off[l], off[l+1]))
    std::vector<uint32_t> levelBase;                 // per BLAS base into levelNodes
    int nodeCount = 0, triCount = 0, instanceCount = 0, tlasCount = 0, vertexCount = 0, meshCount = 0, materialCount = 0, xformCount = 0, lightCount = 0, skySize = 0, textureCount = 0, unskinnedCount = 0;
    int sceneStack = 1;
    // wavefront state
    DevBuf trLo, trLd, trInv, contFlag, blockSums, rayO, rayT, rayR, aovA, aovN, hit, hitX, hitCost, primHit, queue[2], keys[2], keysTmp, sortKeys, sortVals, contMask, waveCounts, counts, work, sortHist, counters64;
    DevBuf img[3];
    float4* extImg[3] = {nullptr, nullptr, nullptr};
    uint32_t slotBases[MAX_DEPTH_SLOTS];
    // stats
    idkpt_stats stats;
    uint32_t* hCounts = nullptr; // pinned
    hipEvent_t evFrame[2] = {nullptr, nullptr};
    // trace-kernel timing (idkptEnableTiming): one event pair per trace launch, resolved lazily in idkptGetStats
    std::vector<hipEvent_t> evPool; size_t evUsed = 0;
    double traceMsAcc = 0.0; uint64_t traceLaunchesAcc = 0;
    int lastQueueSide = 0; int lastQueueCountSlot = 0; bool lastFast = false; int lastBatch = 1; Frame lastFrame;
    int maxBatch = 1; uint32_t Npad = 0; std::vector<uint32_t> pending; DevBuf bases; uint32_t* hBases = nullptr;
};

static hipEvent_t next_event(idkpt_ctx* ctx)
{
    if (ctx->evUsed == ctx->evPool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; ctx->evPool.push_back(e); }
    return ctx->evPool[ctx->evUsed++];
}
// folds all recorded (start, stop) pairs into the accumulators; requires the stream to be idle
static void resolve_trace_events(idkpt_ctx* ctx)
{
    for (size_t i = 0; i + 1 < ctx->evUsed; i += 2) { float ms = 0.0f; if (hipEventElapsedTime(&ms, ctx->evPool[i], ctx->evPool[i + 1]) == hipSuccess) { ctx->traceMsAcc += ms; ctx->traceLaunchesAcc++; } }
    ctx->evUsed = 0;
}
#define TRACE_T0() do { if (ctx->timing) { hipEvent_t _e = next_event(ctx); if (_e) (void)hipEventRecord(_e, st); } } while (0)
#define TRACE_T1() do { if (ctx->timing) { hipEvent_t _e = next_event(ctx); if (_e) (void)hipEventRecord(_e, st); } } while (0)

static int fail(idkpt_ctx* c, int code, const std::string& msg) { if (c) c->lastError = msg; return code; }
#define HIPC(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(ctx, IDKPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
#define REQUIRE(cond, msg) do { if (!(cond)) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, msg); } while (0)

static int local_rows(int H, int mod, int rem) { int n = 0; for (int y = rem; y < H; y += mod) n++; return n; }

template <bool PRIMARY>
static void launch_trace2(idkpt_ctx* ctx, uint32_t grid, size_t lds, hipStream_t st, const DScene& s, const Frame& f, const RayBufs& rays, const TraceBufs& tr, const HitBufs& hits,
                          const uint32_t* list, const uint32_t* cnt, uint32_t* work, uint64_t* counters)
{
    if (f.useTlas) {                // TLAS walk inside the kernel
        if (ctx->counters) hipLaunchKernelGGL((k_trace2<PRIMARY, true, 32, 1, false, 24, 2>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
        else hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 2>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
        return;
    }
    if (ctx->instanceCount > 1) {   // instance loop inside the kernel
        if (ctx->counters) hipLaunchKernelGGL((k_trace2<PRIMARY, true, 32, 1, false, 24, 1>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
        else hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, false, 24, 1>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters);
        return;
    }
    if (ctx->counters) { hipLaunchKernelGGL((k_trace2<PRIMARY, true>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); return; }
    switch (ctx->traceVariant) {   // developer knob (IDKPT_TRACE_VARIANT): s_memtime-instrumented builds; results are bit-identical
        case 7: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 16, 1, true, 65>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); break;   // instrumented, old policy
        case 13: hipLaunchKernelGGL((k_trace2<PRIMARY, false, 32, 1, true, 24>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); break;  // instrumented, default policy
        default: hipLaunchKernelGGL((k_trace2<PRIMARY, false>), dim3(grid), dim3(WAVE), lds, st, s, f, rays, tr, hits, list, cnt, work, counters); break;
    }
}

static int alloc_frame(idkpt_ctx* ctx)
{
    const size_t N = (size_t)ctx->W * ctx->rows;
    ctx->Npad = (uint32_t)((N + 63) / 64 * 64);
    const size_t cap = (size_t)ctx->maxBatch * ctx->Npad;   // ray ids of one batch
    ctx->pending.clear();
    ctx->lastFast = false; ctx->lastBatch = 1;   // nothing rendered into the new buffers yet
    HIPC(ctx->rayO.ensure(cap * 16)); HIPC(ctx->rayT.ensure(cap * 16)); HIPC(ctx->rayR.ensure(cap * 16));
    HIPC(ctx->aovA.ensure(cap * 16)); HIPC(ctx->aovN.ensure(cap * 16));
    HIPC(ctx->trLo.ensure(cap * 16)); HIPC(ctx->trLd.ensure(cap * 16)); HIPC(ctx->trInv.ensure(cap * 16)); HIPC(ctx->contFlag.ensure(cap));
    HIPC(ctx->blockSums.ensure(((cap + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK * 4 + 16));
    HIPC(ctx->hit.ensure(cap * 16)); HIPC(ctx->hitX.ensure(cap * 4)); HIPC(ctx->hitCost.ensure(cap * 4));
    for (int i = 0; i < 2; i++) { HIPC(ctx->queue[i].ensure(cap * 4)); HIPC(ctx->keys[i].ensure(cap * 4)); }
    HIPC(ctx->keysTmp.ensure(cap * 4)); HIPC(ctx->sortKeys.ensure(cap * 4)); HIPC(ctx->sortVals.ensure(cap * 4));
    size_t nW = (cap + 63) / 64;
    HIPC(ctx->contMask.ensure(nW * 8)); HIPC(ctx->waveCounts.ensure(nW * 4));
    HIPC(ctx->counts.ensure(MAX_DEPTH_SLOTS * 4)); HIPC(ctx->work.ensure(4 * MAX_DEPTH_SLOTS * 4)); HIPC(ctx->counters64.ensure(128));
    HIPC(ctx->bases.ensure((size_t)MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4));
    size_t nTiles = (cap + SORT_TILE - 1) / SORT_TILE;
    HIPC(ctx->sortHist.ensure((SORT_RADIX * nTiles + SORT_RADIX) * 4));
    for (int i = 0; i < 3; i++) { HIPC(ctx->img[i].ensure(N * 16)); HIPC(hipMemsetAsync(ctx->img[i].p, 0, N * 16, ctx->stream)); }
    HIPC(hipMemsetAsync(ctx->counters64.p, 0, 128, ctx->stream));
    HIPC(hipMemsetAsync(ctx->aovA.p, 0, cap * 16, ctx->stream)); HIPC(hipMemsetAsync(ctx->aovN.p, 0, cap * 16, ctx->stream));
    HIPC(hipMemsetAsync(ctx->contFlag.p, 0, cap, ctx->stream));   // per-batch values are written by k_gen_primary; the pad ids [N, Npad) must read 0
    ctx->accumulated = 0;
    return IDKPT_OK;
}

// maxBatch changed: the wavefront buffers grow, the accumulation images (and their contents) stay
static int alloc_frame_keep_images(idkpt_ctx* ctx)
{
    const size_t N = (size_t)ctx->W * ctx->rows;
    DevBuf saved[3];
    for (int i = 0; i < 3; i++) { saved[i] = ctx->img[i]; ctx->img[i] = DevBuf(); }
    int rc = alloc_frame(ctx);
    for (int i = 0; i < 3; i++) { if (rc == IDKPT_OK && saved[i].p) (void)hipMemcpy(ctx->img[i].p, saved[i].p, N * 16, hipMemcpyDeviceToDevice); saved[i].release(); }
    return rc;
}

static int flush_batch(idkpt_ctx* ctx);
#define FLUSH() do { int _rc = flush_batch(ctx); if (_rc) return _rc; } while (0)

extern "C" {

const char* idkptGetVersionString(void) { return "idkpt 0.1 (gfx950)"; }

int32_t idkptGetDeviceCount(int32_t* outCount)
{
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    if (outCount) *outCount = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? IDKPT_OK : IDKPT_ERR_NO_DEVICE;
}

int32_t idkptCreate(int32_t deviceCount, const int32_t* deviceIds, idkpt_ctx** outCtx)
{
    if (!outCtx) return IDKPT_ERR_INVALID_ARGUMENT;
    *outCtx = nullptr;
    if (deviceCount != 1) return IDKPT_ERR_INVALID_ARGUMENT; // one context per process per GPU
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return IDKPT_ERR_NO_DEVICE;
    int dev = deviceIds ? deviceIds[0] : 0;
    if (dev < 0 || dev >= n) return IDKPT_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(dev) != hipSuccess) return IDKPT_ERR_HIP;
    idkpt_ctx* ctx = new idkpt_ctx();
    ctx->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ctx->numCUs = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return IDKPT_ERR_HIP; }
    memset(&ctx->st, 0, sizeof(ctx->st));
    ctx->st.Gpu.FocalLength = 8.0f; ctx->st.Gpu.DoRussianRoulette = 1; ctx->st.RayDepth = 7; ctx->st.SamplesPerPixel = 1;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    memset(ctx->slotBases, 0, sizeof(ctx->slotBases));
    memset(ctx->invProj, 0, 64); memset(ctx->invView, 0, 64); memset(ctx->viewPos, 0, 12);
    if (hipHostMalloc((void**)&ctx->hCounts, MAX_DEPTH_SLOTS * 4 + 16, hipHostMallocDefault) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    memset(ctx->hCounts, 0, MAX_DEPTH_SLOTS * 4 + 16);
    if (hipHostMalloc((void**)&ctx->hBases, MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4, hipHostMallocDefault) != hipSuccess) { delete ctx; return IDKPT_ERR_OUT_OF_MEMORY; }
    memset(ctx->hBases, 0, MAX_DEPTH_SLOTS * (MAX_BATCH + 1) * 4);
    (void)hipEventCreate(&ctx->evFrame[0]); (void)hipEventCreate(&ctx->evFrame[1]);
    if (const char* e = getenv("IDKPT_FORCE_GENERIC")) ctx->forceGeneric = atoi(e) != 0;
    if (const char* e = getenv("IDKPT_TRACE_VARIANT")) ctx->traceVariant = atoi(e);
    *outCtx = ctx;
    return IDKPT_OK;
}

int32_t idkptDestroy(idkpt_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    ctx->pending.clear();
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* all[] = {&ctx->nodes, &ctx->tris, &ctx->triVerts, &ctx->descs, &ctx->instances, &ctx->tlas, &ctx->parents, &ctx->leaves, &ctx->positions, &ctx->prevPositions, &ctx->vertices, &ctx->meshes,
                     &ctx->materials, &ctx->xforms, &ctx->lights, &ctx->sky, &ctx->texDescs, &ctx->unskinned, &ctx->joints, &ctx->levelNodes, &ctx->tlasScratch, &ctx->queryIn, &ctx->queryOut, &ctx->trLo, &ctx->trLd, &ctx->trInv, &ctx->contFlag, &ctx->blockSums, &ctx->rayO, &ctx->rayT, &ctx->rayR, &ctx->aovA, &ctx->aovN, &ctx->hit, &ctx->hitX,
                     &ctx->hitCost, &ctx->primHit, &ctx->queue[0], &ctx->queue[1], &ctx->keys[0], &ctx->keys[1], &ctx->keysTmp, &ctx->sortKeys, &ctx->sortVals, &ctx->contMask, &ctx->waveCounts,
                     &ctx->counts, &ctx->work, &ctx->sortHist, &ctx->counters64, &ctx->bases, &ctx->img[0], &ctx->img[1], &ctx->img[2]};
    for (DevBuf* b : all) b->release();
    for (auto& t : ctx->texData) t.release();
    if (ctx->hCounts) (void)hipHostFree(ctx->hCounts);
    if (ctx->hBases) (void)hipHostFree(ctx->hBases);
    if (ctx->evFrame[0]) (void)hipEventDestroy(ctx->evFrame[0]);
    if (ctx->evFrame[1]) (void)hipEventDestroy(ctx->evFrame[1]);
    for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
    if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return IDKPT_OK;
}

int32_t idkptGetLastError(idkpt_ctx* ctx, const char** outMessage)
{
    if (!ctx || !outMessage) return IDKPT_ERR_INVALID_ARGUMENT;
    *outMessage = ctx->lastError.c_str();
    return IDKPT_OK;
}

int32_t idkptSetSize(idkpt_ctx* ctx, int32_t width, int32_t height)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(width > 0 && height > 0 && width <= 4096 && height <= 65536, "idkptSetSize: bad size (FirstHit seeds pack x into 12 bits: width <= 4096)");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    ctx->W = width; ctx->H = height; ctx->rows = local_rows(height, ctx->rowMod, ctx->rowRem);
    return alloc_frame(ctx);
}

int32_t idkptSetRowSharding(idkpt_ctx* ctx, int32_t rowModulo, int32_t rowRemainder)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(rowModulo >= 1 && rowRemainder >= 0 && rowRemainder < rowModulo, "idkptSetRowSharding: need 0 <= remainder < modulo");
    FLUSH();
    ctx->rowMod = rowModulo; ctx->rowRem = rowRemainder;
    if (ctx->W > 0) { HIPC(hipSetDevice(ctx->device)); ctx->rows = local_rows(ctx->H, rowModulo, rowRemainder); return alloc_frame(ctx); }
    return IDKPT_OK;
}

int32_t idkptSetSlotBases(idkpt_ctx* ctx, const uint32_t* slotBases, int32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(count >= 0 && count <= MAX_DEPTH_SLOTS, "idkptSetSlotBases: count out of range");
    FLUSH();
    memset(ctx->slotBases, 0, sizeof(ctx->slotBases));
    for (int i = 0; i < count; i++) ctx->slotBases[i] = slotBases[i];
    return IDKPT_OK;
}

int32_t idkptSetSettings(idkpt_ctx* ctx, const idkpt_settings* s)
{
    if (!ctx || !s) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(s->RayDepth >= 1 && s->RayDepth < MAX_DEPTH_SLOTS - 1, "idkptSetSettings: RayDepth out of range");
    REQUIRE(s->SamplesPerPixel >= 1, "idkptSetSettings: SamplesPerPixel must be >= 1");
    if (memcmp(&ctx->st, s, sizeof(*s)) != 0) FLUSH();   // pending samples were submitted under the old settings
    const idkpt_settings& o = ctx->st;
    // PathTracer setters that call ResetAccumulation (PathTracer.cs:17-98): RayDepth, FocalLength, LenseRadius, DoDebugBVHTraversal, DoTraceLights
    bool reset = o.RayDepth != s->RayDepth || o.Gpu.FocalLength != s->Gpu.FocalLength || o.Gpu.LenseRadius != s->Gpu.LenseRadius ||
                 o.Gpu.DoDebugBVHTraversal != s->Gpu.DoDebugBVHTraversal || o.Gpu.DoTraceLights != s->Gpu.DoTraceLights || o.UseTlas != s->UseTlas;
    ctx->st = *s;
    if (ctx->st.Gpu.DoDebugBVHTraversal) ctx->st.RayDepth = 1; // PathTracer.cs:67-71
    if (reset) ctx->accumulated = 0;
    return IDKPT_OK;
}
int32_t idkptGetSettings(idkpt_ctx* ctx, idkpt_settings* out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = ctx->st; return IDKPT_OK; }

int32_t idkptSetPerFrame(idkpt_ctx* ctx, const float invProjection[16], const float invView[16], const float viewPos[3])
{
    if (!ctx || !invProjection || !invView || !viewPos) return IDKPT_ERR_INVALID_ARGUMENT;
    if (memcmp(ctx->invProj, invProjection, 64) || memcmp(ctx->invView, invView, 64) || memcmp(ctx->viewPos, viewPos, 12)) FLUSH();
    memcpy(ctx->invProj, invProjection, 64); memcpy(ctx->invView, invView, 64); memcpy(ctx->viewPos, viewPos, 12);
    return IDKPT_OK;
}
int32_t idkptSetPerFrameData(idkpt_ctx* ctx, const GpuPerFrameData* p) { if (!ctx || !p) return IDKPT_ERR_INVALID_ARGUMENT; return idkptSetPerFrame(ctx, p->InvProjection, p->InvView, p->ViewPos); }

static int regather_triverts(idkpt_ctx* ctx, uint32_t first, uint32_t count)
{
    if (count == 0) return IDKPT_OK;
    hipLaunchKernelGGL(k_gather_triverts, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, ctx->tris.as<uint4>(), ctx->positions.as<float>(), ctx->triVerts.as<float4>(), first, count);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

static int upload(idkpt_ctx* ctx, DevBuf& b, const void* src, size_t bytes)
{
    HIPC(b.ensure(std::max<size_t>(bytes, 16)));
    if (bytes) HIPC(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return IDKPT_OK;
}

int32_t idkptUploadScene(idkpt_ctx* ctx, const idkpt_scene_desc* sc)
{
    if (!ctx || !sc) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(sc->BlasNodes && sc->BlasNodeCount >= 4, "idkptUploadScene: BlasNodes missing");
    REQUIRE(sc->BlasTriangles && sc->BlasTriangleCount > 0, "idkptUploadScene: BlasTriangles missing");
    REQUIRE(sc->BlasDescs && sc->BlasDescCount > 0 && sc->BlasInstances && sc->BlasInstanceCount > 0, "idkptUploadScene: BlasDescs/BlasInstances missing");
    REQUIRE(sc->VertexPositions && sc->Vertices && sc->VertexCount > 0, "idkptUploadScene: vertices missing");
    REQUIRE(sc->Meshes && sc->MeshCount > 0 && sc->Materials && sc->MaterialCount > 0 && sc->MeshTransforms && sc->MeshTransformCount > 0, "idkptUploadScene: meshes/materials/transforms missing");
    REQUIRE(sc->LightCount >= 0 && sc->LightCount <= IDKPT_MAX_LIGHTS, "idkptUploadScene: more than 256 lights");
    // validate indices so that a bad host array cannot fault the GPU
    for (int i = 0; i < sc->BlasTriangleCount; i++) { const GpuBlasTriangle& t = sc->BlasTriangles[i]; REQUIRE(t.X < (uint32_t)sc->VertexCount && t.Y < (uint32_t)sc->VertexCount && t.Z < (uint32_t)sc->VertexCount && t.MeshId < (uint32_t)sc->MeshCount, "idkptUploadScene: BlasTriangle index out of range"); }
    for (int i = 0; i < sc->MeshCount; i++) REQUIRE(sc->Meshes[i].MaterialId >= 0 && sc->Meshes[i].MaterialId < sc->MaterialCount, "idkptUploadScene: Mesh.MaterialId out of range");
    for (int i = 0; i < sc->BlasInstanceCount; i++) REQUIRE(sc->BlasInstances[i].BlasId < (uint32_t)sc->BlasDescCount && sc->BlasInstances[i].MeshTransformId < (uint32_t)sc->MeshTransformCount, "idkptUploadScene: BlasInstance out of range");
    int maxStack = 1;
    for (int i = 0; i < sc->BlasDescCount; i++) {
        const GpuBlasDesc& d = sc->BlasDescs[i];
        REQUIRE(d.NodeOffset >= 0 && d.NodeCount >= 4 && d.NodeOffset + d.NodeCount <= sc->BlasNodeCount && d.TriangleOffset >= 0 && d.TriangleOffset + d.TriangleCount <= sc->BlasTriangleCount, "idkptUploadScene: BlasDesc range out of bounds");
        maxStack = std::max(maxStack, d.RequiredStackSize);
        for (int n = 1; n < d.NodeCount; n++) {
            const GpuBlasNode& nd = sc->BlasNodes[d.NodeOffset + n];
            if (nd.TriCount > 0) REQUIRE((uint64_t)nd.TriStartOrChild + nd.TriCount <= (uint64_t)d.TriangleCount, "idkptUploadScene: leaf triangle range out of bounds");
            else if (n == 1 || nd.TriStartOrChild != 0) REQUIRE(nd.TriStartOrChild >= 2 && nd.TriStartOrChild + 1 < (uint32_t)d.NodeCount, "idkptUploadScene: child index out of bounds");
        }
    }
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    int rc;
    if ((rc = upload(ctx, ctx->nodes, sc->BlasNodes, (size_t)sc->BlasNodeCount * 32))) return rc;
    if ((rc = upload(ctx, ctx->tris, sc->BlasTriangles, (size_t)sc->BlasTriangleCount * 16))) return rc;
    if ((rc = upload(ctx, ctx->descs, sc->BlasDescs, (size_t)sc->BlasDescCount * sizeof(GpuBlasDesc)))) return rc;
    if ((rc = upload(ctx, ctx->instances, sc->BlasInstances, (size_t)sc->BlasInstanceCount * 8))) return rc;
    if ((rc = upload(ctx, ctx->tlas, sc->TlasNodes, (size_t)(sc->TlasNodes ? sc->TlasNodeCount : 0) * 32))) return rc;
    if ((rc = upload(ctx, ctx->parents, sc->BlasParentIndices, (size_t)(sc->BlasParentIndices ? sc->BlasParentIndexCount : 0) * 4))) return rc;
    if ((rc = upload(ctx, ctx->leaves, sc->BlasLeafIndices, (size_t)(sc->BlasLeafIndices ? sc->BlasLeafIndexCount : 0) * 4))) return rc;
    if ((rc = upload(ctx, ctx->positions, sc->VertexPositions, (size_t)sc->VertexCount * 12))) return rc;
    if ((rc = upload(ctx, ctx->vertices, sc->Vertices, (size_t)sc->VertexCount * 16))) return rc;
    if ((rc = upload(ctx, ctx->meshes, sc->Meshes, (size_t)sc->MeshCount * sizeof(GpuMesh)))) return rc;
    if ((rc = upload(ctx, ctx->materials, sc->Materials, (size_t)sc->MaterialCount * sizeof(GpuMaterial)))) return rc;
    if ((rc = upload(ctx, ctx->xforms, sc->MeshTransforms, (size_t)sc->MeshTransformCount * sizeof(GpuMeshTransform)))) return rc;
    HIPC(ctx->lights.ensure(IDKPT_MAX_LIGHTS * sizeof(GpuLight)));
    if (sc->Lights && sc->LightCount) HIPC(hipMemcpyAsync(ctx->lights.p, sc->Lights, (size_t)sc->LightCount * sizeof(GpuLight), hipMemcpyHostToDevice, ctx->stream));
    ctx->skySize = (sc->SkyFaces && sc->SkyFaceSize > 0) ? sc->SkyFaceSize : 0;
    if ((rc = upload(ctx, ctx->sky, sc->SkyFaces, (size_t)6 * ctx->skySize * ctx->skySize * 16))) return rc;
    for (auto& t : ctx->texData) t.release();
    ctx->texData.clear();
    std::vector<TexDesc> td;
    for (int i = 0; i < sc->TextureCount; i++) {
        const idkpt_texture& t = sc->Textures[i];
        REQUIRE(t.width > 0 && t.height > 0 && t.rgba, "idkptUploadScene: bad texture");
        ctx->texData.emplace_back();
        if ((rc = upload(ctx, ctx->texData.back(), t.rgba, (size_t)t.width * t.height * 16))) return rc;
        td.push_back({ctx->texData.back().as<float4>(), t.width, t.height});
    }
    if ((rc = upload(ctx, ctx->texDescs, td.data(), td.size() * sizeof(TexDesc)))) return rc;
    HIPC(ctx->triVerts.ensure((size_t)sc->BlasTriangleCount * 48));
    ctx->nodeCount = sc->BlasNodeCount; ctx->triCount = sc->BlasTriangleCount; ctx->instanceCount = sc->BlasInstanceCount; ctx->tlasCount = sc->TlasNodes ? sc->TlasNodeCount : 0;
    ctx->vertexCount = sc->VertexCount; ctx->meshCount = sc->MeshCount; ctx->materialCount = sc->MaterialCount; ctx->xformCount = sc->MeshTransformCount;
    ctx->lightCount = sc->Lights ? sc->LightCount : 0; ctx->textureCount = sc->TextureCount;
    ctx->hDescs.assign(sc->BlasDescs, sc->BlasDescs + sc->BlasDescCount);
    ctx->sceneStack = maxStack;
    // refit schedule: internal nodes of every refittable BLAS grouped by depth (children have larger ids than parents)
    ctx->levelOffsets.assign(sc->BlasDescCount, {}); ctx->levelBase.assign(sc->BlasDescCount, 0);
    std::vector<int32_t> allLevels;
    for (int bi = 0; bi < sc->BlasDescCount; bi++) {
        const GpuBlasDesc& d = sc->BlasDescs[bi];
        if (!d.IsRefittable) continue;
        std::vector<int> depth(d.NodeCount, 0); int maxD = 0;
        for (int n = 1; n < d.NodeCount; n++) { const GpuBlasNode& nd = sc->BlasNodes[d.NodeOffset + n]; if (nd.TriCount == 0) { int c = (int)nd.TriStartOrChild; depth[c] = depth[c + 1] = depth[n] + 1; maxD = std::max(maxD, depth[n]); } }
        std::vector<std::vector<int32_t>> lv(maxD + 1);
        for (int n = 1; n < d.NodeCount; n++) if (sc->BlasNodes[d.NodeOffset + n].TriCount == 0 && (n == 1 || n >= 2)) lv[depth[n]].push_back(n);
        ctx->levelBase[bi] = (uint32_t)allLevels.size();
        uint32_t off = 0;
        for (auto& l : lv) { ctx->levelOffsets[bi].push_back(off); off += (uint32_t)l.size(); allLevels.insert(allLevels.end(), l.begin(), l.end()); }
        ctx->levelOffsets[bi].push_back(off);
    }
    if ((rc = upload(ctx, ctx->levelNodes, allLevels.data(), allLevels.size() * 4))) return rc;
    if ((rc = regather_triverts(ctx, 0, (uint32_t)sc->BlasTriangleCount))) return rc;
    HIPC(hipStreamSynchronize(ctx->stream)); // host arrays are only borrowed for the duration of the call
    ctx->haveScene = true;
    ctx->accumulated = 0;
    return IDKPT_OK;
}

int32_t idkptSetLightCount(idkpt_ctx* ctx, int32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(count >= 0 && count <= IDKPT_MAX_LIGHTS, "idkptSetLightCount: out of range");
    FLUSH();
    ctx->lightCount = count; return IDKPT_OK;
}

static DevBuf* which_buffer(idkpt_ctx* ctx, int which, size_t* cap)
{
    switch (which) {
        case IDKPT_BUF_MESH_TRANSFORMS: *cap = (size_t)ctx->xformCount * sizeof(GpuMeshTransform); return &ctx->xforms;
        case IDKPT_BUF_VERTEX_POSITIONS: *cap = (size_t)ctx->vertexCount * 12; return &ctx->positions;
        case IDKPT_BUF_VERTICES: *cap = (size_t)ctx->vertexCount * 16; return &ctx->vertices;
        case IDKPT_BUF_MESHES: *cap = (size_t)ctx->meshCount * sizeof(GpuMesh); return &ctx->meshes;
        case IDKPT_BUF_MATERIALS: *cap = (size_t)ctx->materialCount * sizeof(GpuMaterial); return &ctx->materials;
        case IDKPT_BUF_LIGHTS: *cap = (size_t)IDKPT_MAX_LIGHTS * sizeof(GpuLight); return &ctx->lights;
        case IDKPT_BUF_BLAS_NODES: *cap = (size_t)ctx->nodeCount * 32; return &ctx->nodes;
        case IDKPT_BUF_TLAS_NODES: *cap = (size_t)ctx->tlasCount * 32; return &ctx->tlas;
        case IDKPT_BUF_JOINT_MATRICES: *cap = ctx->joints.bytes; return &ctx->joints;
        default: return nullptr;
    }
}

int32_t idkptUpdateBuffer(idkpt_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, const void* data)
{
    if (!ctx || !data) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptUpdateBuffer: no scene uploaded");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    if (which == IDKPT_BUF_JOINT_MATRICES) { size_t need = offsetBytes + bytes; if (need > ctx->joints.bytes) { DevBuf nb; HIPC(nb.ensure(need)); if (ctx->joints.p) HIPC(hipMemcpy(nb.p, ctx->joints.p, ctx->joints.bytes, hipMemcpyDeviceToDevice)); ctx->joints.release(); ctx->joints = nb; } }
    size_t cap = 0; DevBuf* b = which_buffer(ctx, which, &cap);
    REQUIRE(b != nullptr, "idkptUpdateBuffer: unknown buffer");
    REQUIRE(offsetBytes + bytes <= cap, "idkptUpdateBuffer: range exceeds buffer");
    HIPC(hipMemcpyAsync((char*)b->p + offsetBytes, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (which == IDKPT_BUF_VERTEX_POSITIONS) { int rc = regather_triverts(ctx, 0, (uint32_t)ctx->triCount); if (rc) return rc; }
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

int32_t idkptDownloadBuffer(idkpt_ctx* ctx, int32_t which, size_t offsetBytes, size_t bytes, void* dst)
{
    if (!ctx || !dst) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t cap = 0; DevBuf* b = which_buffer(ctx, which, &cap);
    REQUIRE(b != nullptr && offsetBytes + bytes <= cap, "idkptDownloadBuffer: bad buffer/range");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipMemcpyAsync(dst, (char*)b->p + offsetBytes, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

int32_t idkptBuildTlas(idkpt_ctx* ctx, const GpuTlasNode* nodes, int32_t nodeCount)
{
    if (!ctx || !nodes || nodeCount <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    int rc = upload(ctx, ctx->tlas, nodes, (size_t)nodeCount * 32); if (rc) return rc;
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->tlasCount = nodeCount;
    return IDKPT_OK;
}

int32_t idkptBuildTlasOnDevice(idkpt_ctx* ctx, int32_t searchRadius)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptBuildTlasOnDevice: no scene uploaded");
    REQUIRE(searchRadius >= 1, "idkptBuildTlasOnDevice: searchRadius must be >= 1 (reference: 15)");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    const int n = ctx->instanceCount, nodeCount = 2 * n - 1;
    HIPC(ctx->tlas.ensure((size_t)nodeCount * 32));
    // scratch: temp nodes (2n-1) + leaves (n) as float4 pairs, keys (n), pref (n)
    const size_t tempOff = 0, leafOff = (size_t)nodeCount * 32, keyOff = leafOff + (size_t)n * 32, prefOff = keyOff + (size_t)n * 4;
    HIPC(ctx->tlasScratch.ensure(prefOff + (size_t)n * 4));
    char* sc = ctx->tlasScratch.as<char>();
    hipLaunchKernelGGL(k_tlas_build, dim3(1), dim3(TLAS_BUILD_THREADS), 0, ctx->stream, ctx->nodes.as<float4>(), ctx->descs.as<GpuBlasDesc>(), ctx->instances.as<GpuBlasInstance>(),
                       ctx->xforms.as<float4>(), n, (int)searchRadius, ctx->tlas.as<float4>(), (float4*)(sc + tempOff), (float4*)(sc + leafOff), (uint32_t*)(sc + keyOff), (int*)(sc + prefOff));
    HIPC(hipGetLastError());
    ctx->tlasCount = nodeCount;
    return IDKPT_OK;
}

static DScene make_dscene(idkpt_ctx* ctx);
// frame constants for the ray-query / shadow kernels: only the traversal-related fields are read
static int query_frame(idkpt_ctx* ctx, Frame& f, size_t& ldsBytes, uint32_t& grid)
{
    memset(&f, 0, sizeof(f));
    f.g = ctx->st.Gpu; f.useTlas = ctx->st.UseTlas;
    f.stackCap = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack);
    f.tlasCap = std::min(TLAS_STACK_SIZE, std::max(1, ctx->instanceCount));
    ldsBytes = (size_t)(f.stackCap + (f.useTlas ? f.tlasCap : 0)) * WAVE * 4;
    if (ldsBytes > 64 * 1024) return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack");
    int wavesPerCU = (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
    grid = (uint32_t)(ctx->numCUs * std::max(1, wavesPerCU));
    return IDKPT_OK;
}

int32_t idkptTraceRays(idkpt_ctx* ctx, const idkpt_ray* rays, size_t count, uint32_t flags, idkpt_hit* hits)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRays: no scene uploaded");
    REQUIRE(count == 0 || (rays && hits), "idkptTraceRays: null rays/hits");
    REQUIRE(count < (1ull << 31), "idkptTraceRays: too many rays in one call");
    REQUIRE((flags & ~3u) == 0, "idkptTraceRays: unknown flags");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceRays: UseTlas set but no TLAS nodes uploaded");
    if (count == 0) return IDKPT_OK;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    Frame f; size_t ldsBytes; uint32_t grid;
    int rc = query_frame(ctx, f, ldsBytes, grid); if (rc) return rc;
    DScene s = make_dscene(ctx);
    HIPC(ctx->queryIn.ensure(count * sizeof(idkpt_ray))); HIPC(ctx->queryOut.ensure(count * sizeof(idkpt_hit)));
    hipStream_t st = ctx->stream;
    HIPC(hipMemcpyAsync(ctx->queryIn.p, rays, count * sizeof(idkpt_ray), hipMemcpyHostToDevice, st));
    uint32_t* work = ctx->work.as<uint32_t>();
    HIPC(hipMemsetAsync(work, 0, 4, st));
    const int lights = (flags & IDKPT_TRACE_LIGHTS) ? 1 : 0;
    if (flags & IDKPT_TRACE_ANY_HIT) hipLaunchKernelGGL((k_trace_query<true>), dim3(grid), dim3(WAVE), ldsBytes, st, s, f, ctx->queryIn.as<idkpt_ray>(), ctx->queryOut.as<idkpt_hit>(), (uint32_t)count, lights, work);
    else hipLaunchKernelGGL((k_trace_query<false>), dim3(grid), dim3(WAVE), ldsBytes, st, s, f, ctx->queryIn.as<idkpt_ray>(), ctx->queryOut.as<idkpt_hit>(), (uint32_t)count, lights, work);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(hits, ctx->queryOut.p, count * sizeof(idkpt_hit), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return IDKPT_OK;
}

int32_t idkptTraceShadows(idkpt_ctx* ctx, const idkpt_shadow_params* p, const float* depth, const float* normalOct, float* visibility)
{
    if (!ctx || !p) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadows: no scene uploaded");
    REQUIRE(depth && normalOct && visibility, "idkptTraceShadows: null image");
    REQUIRE(p->Width > 0 && p->Height > 0 && (size_t)p->Width * p->Height < (1ull << 30), "idkptTraceShadows: bad image size");
    REQUIRE(p->RayTracingSamples >= 1, "idkptTraceShadows: RayTracingSamples must be >= 1");
    REQUIRE(p->LightIndex >= 0 && p->LightIndex < ctx->lightCount, "idkptTraceShadows: LightIndex out of range");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptTraceShadows: UseTlas set but no TLAS nodes uploaded");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    Frame f; size_t ldsBytes; uint32_t grid;
    int rc = query_frame(ctx, f, ldsBytes, grid); if (rc) return rc;
    DScene s = make_dscene(ctx);
    const size_t N = (size_t)p->Width * p->Height;
    HIPC(ctx->queryIn.ensure(N * 12)); HIPC(ctx->queryOut.ensure(N * 4));
    hipStream_t st = ctx->stream;
    float* dDepth = ctx->queryIn.as<float>(); float2* dNormal = (float2*)(dDepth + N); float* dVis = ctx->queryOut.as<float>();
    HIPC(hipMemcpyAsync(dDepth, depth, N * 4, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(dNormal, normalOct, N * 8, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(dVis, visibility, N * 4, hipMemcpyHostToDevice, st));
    const uint32_t tiles = (uint32_t)(((p->Width + 7) / 8) * ((p->Height + 7) / 8));
    hipLaunchKernelGGL(k_shadows, dim3(tiles), dim3(WAVE), ldsBytes, st, s, f, *p, (const float*)dDepth, (const float2*)dNormal, dVis);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(visibility, dVis, N * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return IDKPT_OK;
}

int32_t idkptRefitBlas(idkpt_ctx* ctx, int32_t blasId)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRefitBlas: no scene uploaded");
    REQUIRE(blasId >= 0 && blasId < (int)ctx->hDescs.size(), "idkptRefitBlas: blasId out of range");
    const GpuBlasDesc& d = ctx->hDescs[blasId];
    if (!d.IsRefittable || d.LeafIndicesCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRefitBlas: BLAS is not refittable (no leaf/parent indices)");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    int rc = regather_triverts(ctx, (uint32_t)d.TriangleOffset, (uint32_t)d.TriangleCount); if (rc) return rc;
    hipLaunchKernelGGL(k_refit_leaves, dim3((d.LeafIndicesCount + 63) / 64), dim3(64), 0, ctx->stream, ctx->nodes.as<float4>(), ctx->tris.as<uint4>(), ctx->triVerts.as<float4>(),
                       ctx->leaves.as<int32_t>() + d.LeafIndicesOffset, (uint32_t)d.LeafIndicesCount, (uint32_t)d.NodeOffset, (uint32_t)d.TriangleOffset);
    const std::vector<uint32_t>& off = ctx->levelOffsets[blasId];
    for (int l = (int)off.size() - 2; l >= 0; l--) {
        uint32_t cnt = off[l + 1] - off[l];
        if (!cnt) continue;
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 63) / 64), dim3(64), 0, ctx->stream, ctx->nodes.as<float4>(), ctx->levelNodes.as<int32_t>() + ctx->levelBase[blasId] + off[l], cnt, (uint32_t)d.NodeOffset);
    }
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

int32_t idkptUploadUnskinnedVertices(idkpt_ctx* ctx, const GpuUnskinnedVertex* verts, int32_t count)
{
    if (!ctx || !verts || count <= 0) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    int rc = upload(ctx, ctx->unskinned, verts, (size_t)count * sizeof(GpuUnskinnedVertex)); if (rc) return rc;
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->unskinnedCount = count;
    return IDKPT_OK;
}

int32_t idkptSkin(idkpt_ctx* ctx, uint32_t inOff, uint32_t outOff, uint32_t jointOff, uint32_t count)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene || ctx->unskinnedCount == 0 || ctx->joints.bytes == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptSkin: needs scene, unskinned vertices and joint matrices");
    REQUIRE((uint64_t)inOff + count <= (uint64_t)ctx->unskinnedCount && (uint64_t)outOff + count <= (uint64_t)ctx->vertexCount, "idkptSkin: range out of bounds");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(ctx->prevPositions.ensure((size_t)ctx->vertexCount * 12));
    if (count) hipLaunchKernelGGL(k_skin, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, ctx->unskinned.as<GpuUnskinnedVertex>(), ctx->joints.as<float4>(), ctx->positions.as<float>(),
                                  ctx->prevPositions.as<float>(), ctx->vertices.as<uint4>(), inOff, outOff, jointOff, count);
    HIPC(hipGetLastError());
    return IDKPT_OK;
}

int32_t idkptResetAccumulation(idkpt_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->accumulated = 0; return IDKPT_OK; }
int32_t idkptGetAccumulatedSamples(idkpt_ctx* ctx, uint32_t* out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = ctx->accumulated; return IDKPT_OK; }

static DScene make_dscene(idkpt_ctx* ctx)
{
    DScene s;
    s.nodes = ctx->nodes.as<float4>(); s.tris = ctx->tris.as<uint4>(); s.triVerts = ctx->triVerts.as<float4>();
    s.descs = ctx->descs.as<GpuBlasDesc>(); s.instances = ctx->instances.as<GpuBlasInstance>(); s.instanceCount = ctx->instanceCount;
    s.tlas = ctx->tlas.as<float4>(); s.tlasCount = ctx->tlasCount; s.vertices = ctx->vertices.as<uint4>();
    s.meshes = ctx->meshes.as<GpuMesh>(); s.materials = ctx->materials.as<GpuMaterial>(); s.xforms = ctx->xforms.as<float4>();
    s.lights = ctx->lights.as<GpuLight>(); s.lightCount = ctx->lightCount; s.sky = ctx->sky.as<float4>(); s.skySize = ctx->skySize;
    s.textures = ctx->texDescs.as<TexDesc>(); s.textureCount = ctx->textureCount;
    return s;
}

static float4* image_ptr(idkpt_ctx* ctx, int i) { return ctx->extImg[i] ? ctx->extImg[i] : ctx->img[i].as<float4>(); }

// fast path = persistent while-while traversal (one BLAS, instance list or TLAS); only the debug traversal-cost view uses the general kernel
static bool fast_path(idkpt_ctx* ctx) { return ctx->instanceCount >= 1 && !ctx->st.Gpu.DoDebugBVHTraversal && !ctx->forceGeneric; }

// One batch of B deferred samples: FirstHit -> [sort ->] NHit x (RayDepth-1) -> FinalDraw (PathTracer.cs:218-270), every
// stage launched once for all B samples.  Sample k owns ray ids [k*Npad, k*Npad+N); alive queues are batch-wide but stay
// grouped by sample (stable compaction / sort with the sample index above the key), and every ray's NHit slot is its
// position inside its own sample's queue, so each sample gets exactly the RNG streams of a stand-alone frame.
static int flush_batch(idkpt_ctx* ctx)
{
    const int B = (int)ctx->pending.size();
    if (B == 0) return IDKPT_OK;
    const uint32_t N = (uint32_t)((size_t)ctx->W * ctx->rows);
    const uint32_t Npad = ctx->Npad;
    const uint32_t total = (uint32_t)B * Npad;
    DScene s = make_dscene(ctx);
    Frame f;
    memcpy(f.invProj, ctx->invProj, 64); memcpy(f.invView, ctx->invView, 64); memcpy(f.viewPos, ctx->viewPos, 12);
    f.W = ctx->W; f.H = ctx->H; f.rowMod = ctx->rowMod; f.rowRem = ctx->rowRem; f.rows = ctx->rows;
    f.g = ctx->st.Gpu; f.useTlas = ctx->st.UseTlas;
    f.stackCap = std::max(1, ctx->st.BlasStackSize > 0 ? ctx->st.BlasStackSize : ctx->sceneStack);
    f.outputAovs = ctx->st.OutputAOVs;
    f.batch = B; f.Npad = Npad;
    for (int k = 0; k < MAX_BATCH; k++) f.accum[k] = k < B ? ctx->pending[k] : 0u;
    f.accumulated = f.accum[0];
    RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
    HitBufs hits = {ctx->hit.as<float4>(), ctx->hitX.as<uint32_t>(), ctx->hitCost.as<float>()};
    uint32_t* counts = ctx->counts.as<uint32_t>();
    uint32_t* bases = ctx->bases.as<uint32_t>();             // [MAX_DEPTH_SLOTS][MAX_BATCH+1]
    uint32_t* work = ctx->work.as<uint32_t>();
    uint64_t* counters = ctx->counters64.as<uint64_t>();
    const int depth = ctx->st.RayDepth;
    hipStream_t st = ctx->stream;
    if (ctx->timing) HIPC(hipEventRecord(ctx->evFrame[0], st));
    HIPC(hipMemsetAsync(work, 0, 4 * MAX_DEPTH_SLOTS * 4, st));
    HIPC(hipMemsetAsync(counts, 0, MAX_DEPTH_SLOTS * 4, st));

    f.tlasCap = std::min(TLAS_STACK_SIZE, std::max(1, ctx->instanceCount));
    size_t ldsBytes = (size_t)(f.stackCap + (f.useTlas ? f.tlasCap : 0)) * WAVE * 4;
    if (const char* e = getenv("IDKPT_LDS_PAD")) ldsBytes += (size_t)atoi(e);   // developer knob: caps the waves per CU (occupancy experiments)
    if (ldsBytes > 64 * 1024) { ctx->pending.clear(); return fail(ctx, IDKPT_ERR_INVALID_ARGUMENT, "BlasStackSize too large for the LDS traversal stack"); }
    // persistent trace grid: as many 1-wave workgroups as the chip holds (32 waves/CU, limited by LDS)
    int wavesPerCU = (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(ldsBytes, 1));
    wavesPerCU = std::max(1, wavesPerCU);
    const uint32_t traceGrid = (uint32_t)(ctx->numCUs * wavesPerCU);
    const bool debug = f.g.DoDebugBVHTraversal != 0;
    const uint32_t gridTotal = (total + 255) / 256;
    const bool fast = fast_path(ctx);
    if (!fast && B != 1) { ctx->pending.clear(); return fail(ctx, IDKPT_ERR_UNKNOWN, "internal: generic path is never batched"); }
    unsigned long long* contMask = ctx->contMask.as<unsigned long long>();
    uint32_t* waveCounts = ctx->waveCounts.as<uint32_t>();
    const int BS = MAX_BATCH + 1;

    // ---- FirstHit
    uint32_t* activeList = ctx->sortVals.as<uint32_t>(); // scratch (capacity uints), free until the first sort
    uint32_t* activeCount = counts + (MAX_DEPTH_SLOTS - 1);
    TraceBufs tr = {ctx->trLo.as<float4>(), ctx->trLd.as<float4>(), ctx->trInv.as<float4>()};
    TraceBufs trNone = {nullptr, nullptr, nullptr};
    uint32_t* waveLocal = waveCounts;                     // per-wave exclusive offset inside its 256-wave scan block
    uint32_t* blockSums = ctx->blockSums.as<uint32_t>();
    const uint32_t scanBlocks = ((total + 63) / 64 + SCAN_WAVES_PER_BLOCK - 1) / SCAN_WAVES_PER_BLOCK;
    uint32_t* keysTmp = ctx->keysTmp.as<uint32_t>();
    {
        if (fast) {
            const uint32_t tilesX = ((uint32_t)f.W + 7) / 8, tilesY = ((uint32_t)f.rows + 7) / 8;
            const uint32_t genWaves = tilesX * tilesY;
            const int cull = f.g.DoTraceLights ? 0 : 1;
            if (ctx->capturePrimary) hipLaunchKernelGGL(k_fill_miss, dim3((N + 255) / 256), dim3(256), 0, st, hits.hit + (size_t)(B - 1) * Npad, hits.xformId + (size_t)(B - 1) * Npad, N);
            hipLaunchKernelGGL(k_gen_primary, dim3(B, (genWaves + 15) / 16), dim3(1024), 0, st, s, f, rays, tr, cull, activeList, activeCount, keysTmp, ctx->contFlag.as<uint8_t>());
            TRACE_T0();
            launch_trace2<true>(ctx, traceGrid, ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, work + 0, counters);
            TRACE_T1();
            if (ctx->capturePrimary) { HIPC(ctx->primHit.ensure((size_t)N * 16)); HIPC(hipMemcpyAsync(ctx->primHit.p, ctx->hit.as<float4>() + (size_t)(B - 1) * Npad, (size_t)N * 16, hipMemcpyDeviceToDevice, st)); }
            hipLaunchKernelGGL(k_shade_first, dim3(gridTotal), dim3(256), 0, st, s, f, rays, tr, hits, (const uint32_t*)activeList, (const uint32_t*)activeCount, ctx->contFlag.as<uint8_t>(), keysTmp);
            hipLaunchKernelGGL((k_scan_local<true>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, (const uint32_t*)nullptr, total, (const uint8_t*)ctx->contFlag.as<uint8_t>(), contMask, waveLocal, blockSums);
        } else {
            TRACE_T0();
            uint32_t g = std::min<uint32_t>(traceGrid, (N + 63) / 64);
            if (ctx->counters) { if (debug) hipLaunchKernelGGL((k_trace_primary<true, true>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); else hipLaunchKernelGGL((k_trace_primary<true, false>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); }
            else { if (debug) hipLaunchKernelGGL((k_trace_primary<false, true>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); else hipLaunchKernelGGL((k_trace_primary<false, false>), dim3(g), dim3(WAVE), ldsBytes, st, s, f, rays, hits, N, work + 0, counters); }
            TRACE_T1();
            if (ctx->capturePrimary) { HIPC(ctx->primHit.ensure((size_t)N * 16)); HIPC(hipMemcpyAsync(ctx->primHit.p, ctx->hit.p, (size_t)N * 16, hipMemcpyDeviceToDevice, st)); }
            hipLaunchKernelGGL((k_shade<true>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, trNone, hits, (const uint32_t*)nullptr, (const uint32_t*)nullptr, total, (const uint32_t*)nullptr, 0u,
                               contMask, waveCounts, keysTmp);
            hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, (const uint32_t*)nullptr, total, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
        }
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, (const uint32_t*)nullptr, total, blockSums, (const uint32_t*)waveLocal, counts + 1, (unsigned long long*)(1 < depth ? counters + 2 : nullptr),
                           (const unsigned long long*)contMask, (const uint32_t*)nullptr, Npad, B, bases + 1 * BS);
        hipLaunchKernelGGL((k_compact<true>), dim3(gridTotal), dim3(256), 0, st, (const uint32_t*)nullptr, (const uint32_t*)nullptr, total, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                           (const uint32_t*)keysTmp, ctx->queue[1].as<uint32_t>(), ctx->keys[1].as<uint32_t>());
    }
    int side = 1; // queue[side] holds the rays entering bounce j, its length is counts[j], sample k starts at bases[j][k]
    for (int j = 1; j < depth; j++) {
        uint32_t* q = ctx->queue[side].as<uint32_t>(); uint32_t* k = ctx->keys[side].as<uint32_t>();
        const uint32_t* cnt = counts + j;
        if (ctx->st.DoRaySorting && j > 1) {
            // RaySorting() (PathTracer.cs:232-237): stable sort of (key, rayIndex); key = 21-bit triangle id with the batch's
            // sample index above it, so one sort orders every sample's queue exactly like a stand-alone counting sort
            const uint32_t nTiles = (total + SORT_TILE - 1) / SORT_TILE;
            const int passes = B > 1 ? 4 : 3;
            uint32_t* digitTotals = ctx->sortHist.as<uint32_t>() + (size_t)SORT_RADIX * nTiles;   // 128 words behind the [digit][tile] table
            uint32_t* ka = k; uint32_t* va = q; uint32_t* kb = ctx->sortKeys.as<uint32_t>(); uint32_t* vb = ctx->sortVals.as<uint32_t>();
            for (int pass = 0; pass < passes; pass++) {
                hipLaunchKernelGGL(k_sort_hist, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)ka, cnt, (uint32_t)(7 * pass), ctx->sortHist.as<uint32_t>(), nTiles);
                hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(1024), 0, st, cnt, ctx->sortHist.as<uint32_t>(), nTiles, digitTotals);
                hipLaunchKernelGGL(k_sort_scatter, dim3(nTiles), dim3(SORT_BLOCK), 0, st, (const uint32_t*)ka, (const uint32_t*)va, cnt, (uint32_t)(7 * pass), (const uint32_t*)ctx->sortHist.as<uint32_t>(), nTiles, (const uint32_t*)digitTotals, kb, vb);
                std::swap(ka, kb); std::swap(va, vb);
            }
            // odd pass count: the sorted data sits in (sortKeys, sortVals) -> copy the indices back (the reference copies W*H*4 B too, PathTracer.cs:296)
            if (va != q) HIPC(hipMemcpyAsync(q, va, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
        }
        TRACE_T0();
        if (fast) launch_trace2<false>(ctx, traceGrid, ldsBytes, st, s, f, rays, tr, hits, (const uint32_t*)q, cnt, work + j, counters);
        else {
            if (ctx->counters) hipLaunchKernelGGL((k_trace_queue<true>), dim3(traceGrid), dim3(WAVE), ldsBytes, st, s, f, rays, hits, (const uint32_t*)q, cnt, work + j, counters);
            else hipLaunchKernelGGL((k_trace_queue<false>), dim3(traceGrid), dim3(WAVE), ldsBytes, st, s, f, rays, hits, (const uint32_t*)q, cnt, work + j, counters);
        }
        TRACE_T1();
        hipLaunchKernelGGL((k_shade<false>), dim3(gridTotal), dim3(256), 0, st, s, f, rays, fast ? tr : trNone, hits, (const uint32_t*)q, cnt, 0u, (const uint32_t*)(bases + j * BS), ctx->slotBases[j],
                           contMask, waveCounts, keysTmp);
        hipLaunchKernelGGL((k_scan_local<false>), dim3(scanBlocks), dim3(SCAN_WAVES_PER_BLOCK), 0, st, cnt, 0u, (const uint8_t*)nullptr, contMask, waveLocal, blockSums);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, cnt, 0u, blockSums, (const uint32_t*)waveLocal, counts + j + 1, (unsigned long long*)(j + 1 < depth ? counters + 2 : nullptr),
                           (const unsigned long long*)contMask, (const uint32_t*)(bases + j * BS), Npad, B, bases + (j + 1) * BS);
        hipLaunchKernelGGL((k_compact<false>), dim3(gridTotal), dim3(256), 0, st, (const uint32_t*)q, cnt, 0u, (const unsigned long long*)contMask, (const uint32_t*)waveLocal, (const uint32_t*)blockSums,
                           (const uint32_t*)keysTmp, ctx->queue[1 - side].as<uint32_t>(), ctx->keys[1 - side].as<uint32_t>());
        side = 1 - side;
    }
    ctx->lastQueueSide = side; ctx->lastQueueCountSlot = depth; ctx->lastFast = fast; ctx->lastBatch = B; ctx->lastFrame = f;
    hipLaunchKernelGGL(k_final_draw, dim3((N + 255) / 256), dim3(256), 0, st, f, rays, image_ptr(ctx, 0), image_ptr(ctx, 1), image_ptr(ctx, 2), N);
    HIPC(hipGetLastError());
    // queue lengths stay on the GPU during the batch; a copy goes to pinned memory for GetStats (no sync here)
    HIPC(hipMemcpyAsync(ctx->hCounts, counts, MAX_DEPTH_SLOTS * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(ctx->hBases, bases, MAX_DEPTH_SLOTS * BS * 4, hipMemcpyDeviceToHost, st));
    if (ctx->timing) HIPC(hipEventRecord(ctx->evFrame[1], st));
    ctx->stats.Frames += (uint64_t)B;
    ctx->stats.PrimaryRays += (uint64_t)N * (uint64_t)B;
    ctx->pending.clear();
    return IDKPT_OK;
}

int32_t idkptRender(idkpt_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    if (!ctx->haveScene) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: no scene uploaded");
    if (ctx->W <= 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: idkptSetSize not called");
    if (ctx->st.UseTlas && ctx->tlasCount == 0) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptRender: UseTlas set but no TLAS nodes uploaded");
    HIPC(hipSetDevice(ctx->device));
    if (ctx->timing && ctx->evUsed > 4096) { HIPC(hipStreamSynchronize(ctx->stream)); resolve_trace_events(ctx); }
    // every sample is deferred; a batch is launched as soon as maxBatch samples are pending (or on any call that needs
    // results).  The general path (multi-instance / TLAS / debug cost) is launched sample by sample.
    const int limit = fast_path(ctx) ? ctx->maxBatch : 1;
    for (int i = 0; i < ctx->st.SamplesPerPixel; i++) {
        ctx->pending.push_back(ctx->accumulated++);
        if ((int)ctx->pending.size() >= limit) { int rc = flush_batch(ctx); if (rc) return rc; }
    }
    return IDKPT_OK;
}

int32_t idkptSynchronize(idkpt_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; HIPC(hipSetDevice(ctx->device)); FLUSH(); HIPC(hipStreamSynchronize(ctx->stream)); return IDKPT_OK; }

// Launches whatever is pending without waiting for it (lets a host overlap its own work with the GPU).
int32_t idkptFlush(idkpt_ctx* ctx) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; HIPC(hipSetDevice(ctx->device)); FLUSH(); return IDKPT_OK; }

int32_t idkptSetMaxBatch(idkpt_ctx* ctx, int32_t maxBatch)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(maxBatch >= 1 && maxBatch <= MAX_BATCH, "idkptSetMaxBatch: 1..128");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    if (maxBatch == ctx->maxBatch) return IDKPT_OK;
    ctx->maxBatch = maxBatch;
    if (ctx->W > 0) { uint32_t acc = ctx->accumulated; int rc = alloc_frame_keep_images(ctx); if (rc) return rc; ctx->accumulated = acc; }
    return IDKPT_OK;
}

int32_t idkptDownload(idkpt_ctx* ctx, int32_t image, float* rgba, size_t bytes)
{
    if (!ctx || !rgba) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3, "idkptDownload: bad image id");
    size_t need = (size_t)ctx->W * ctx->rows * 16;
    REQUIRE(bytes == need && need > 0, "idkptDownload: bytes must equal localRows*width*16");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipMemcpyAsync(rgba, image_ptr(ctx, image), need, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    return IDKPT_OK;
}

int32_t idkptDownloadRays(idkpt_ctx* ctx, GpuWavefrontRay* out, size_t bytes)
{
    if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t N = (size_t)ctx->W * ctx->rows;
    REQUIRE(bytes == N * sizeof(GpuWavefrontRay) && N > 0, "idkptDownloadRays: bytes must equal pixelCount*48");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    const size_t off = (size_t)(ctx->lastBatch - 1) * ctx->Npad * 16; // the most recent sample of the last batch
    if (ctx->lastFast) {   // complete the planes the ray generation left out for pre-culled pixels
        RayBufs rays = {ctx->rayO.as<float4>(), ctx->rayT.as<float4>(), ctx->rayR.as<float4>(), ctx->aovA.as<float4>(), ctx->aovN.as<float4>()};
        hipLaunchKernelGGL(k_regen_culled, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, ctx->lastFrame, rays, (const uint8_t*)ctx->contFlag.as<uint8_t>(), (uint32_t)(ctx->lastBatch - 1), (uint32_t)N);
    }
    std::vector<float4> a(N), b(N), c(N);
    HIPC(hipMemcpyAsync(a.data(), (char*)ctx->rayO.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(b.data(), (char*)ctx->rayT.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(c.data(), (char*)ctx->rayR.p + off, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < N; i++) {
        GpuWavefrontRay& r = out[i];
        r.Origin[0] = a[i].x; r.Origin[1] = a[i].y; r.Origin[2] = a[i].z; r.PreviousIOROrTraverseCost = a[i].w;
        r.Throughput[0] = b[i].x; r.Throughput[1] = b[i].y; r.Throughput[2] = b[i].z; r.PackedDirectionX = b[i].w;
        r.Radiance[0] = c[i].x; r.Radiance[1] = c[i].y; r.Radiance[2] = c[i].z; r.PackedDirectionY = c[i].w;
    }
    return IDKPT_OK;
}

int32_t idkptDownloadAliveQueue(idkpt_ctx* ctx, uint32_t* indices, size_t capacity, uint32_t* outCount)
{
    if (!ctx || !outCount) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    // the most recent sample's segment of the batch-wide queue; entries are ray ids -> subtract the sample's id offset
    const uint32_t* hb = ctx->hBases + (size_t)ctx->lastQueueCountSlot * (MAX_BATCH + 1);
    const uint32_t first = hb[ctx->lastBatch - 1], n = hb[ctx->lastBatch] - first;
    *outCount = n;
    if (indices && n) {
        REQUIRE(capacity >= n, "idkptDownloadAliveQueue: capacity too small");
        HIPC(hipMemcpy(indices, ctx->queue[ctx->lastQueueSide].as<uint32_t>() + first, (size_t)n * 4, hipMemcpyDeviceToHost));
        const uint32_t sub = (uint32_t)(ctx->lastBatch - 1) * ctx->Npad;
        for (uint32_t i = 0; i < n; i++) indices[i] -= sub;
    }
    return IDKPT_OK;
}

int32_t idkptEnablePrimaryHitCapture(idkpt_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->capturePrimary = enable != 0; return IDKPT_OK; }

int32_t idkptDownloadPrimaryHits(idkpt_ctx* ctx, float* t, uint32_t* triangleId, float* baryXY, size_t pixelCount)
{
    if (!ctx || !t || !triangleId || !baryXY) return IDKPT_ERR_INVALID_ARGUMENT;
    size_t N = (size_t)ctx->W * ctx->rows;
    REQUIRE(pixelCount == N, "idkptDownloadPrimaryHits: pixelCount mismatch");
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    if (!ctx->capturePrimary || ctx->primHit.bytes < N * 16) return fail(ctx, IDKPT_ERR_INVALID_OPERATION, "idkptDownloadPrimaryHits: call idkptEnablePrimaryHitCapture(ctx,1) before idkptRender");
    std::vector<float4> h(N);
    HIPC(hipMemcpyAsync(h.data(), ctx->primHit.p, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < N; i++) { t[i] = h[i].x; baryXY[2 * i] = h[i].y; baryXY[2 * i + 1] = h[i].z; memcpy(&triangleId[i], &h[i].w, 4); }
    return IDKPT_OK;
}

int32_t idkptGetStats(idkpt_ctx* ctx, idkpt_stats* out)
{
    if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    idkpt_stats s = ctx->stats;
    for (int j = 0; j < 16; j++) { const uint32_t* hb = ctx->hBases + (size_t)j * (MAX_BATCH + 1); s.LastAliveCounts[j] = (j >= 1 && j < ctx->st.RayDepth) ? hb[ctx->lastBatch] - hb[ctx->lastBatch - 1] : 0; }
    // [0]: primary rays that entered the traversal kernel (all pixels, or the survivors of the root-box pre-cull on the fast path)
    s.LastAliveCounts[0] = s.Frames ? (ctx->lastFast ? ctx->hCounts[MAX_DEPTH_SLOTS - 1] : (uint32_t)((size_t)ctx->W * ctx->rows)) : 0;
    s.LastFrameMs = 0.0f; s.LastTraceMs = 0.0f;
    if (ctx->timing && s.Frames > 0) { float ms = 0.0f; if (hipEventElapsedTime(&ms, ctx->evFrame[0], ctx->evFrame[1]) == hipSuccess) s.LastFrameMs = ms; }
    resolve_trace_events(ctx);
    s.TraceMsTotal = ctx->traceMsAcc; s.TraceLaunches = ctx->traceLaunchesAcc;
    s.LastTraceMs = s.TraceLaunches ? (float)(s.TraceMsTotal / (double)s.TraceLaunches) : 0.0f;
    uint64_t c[4] = {0, 0, 0, 0};
    HIPC(hipMemcpy(c, ctx->counters64.p, 32, hipMemcpyDeviceToHost));
    s.NodePairVisits = c[0]; s.TriangleTests = c[1];
    if (ctx->traceVariant == 7 || ctx->traceVariant == 13) { uint64_t d[16]; HIPC(hipMemcpy(d, ctx->counters64.p, 128, hipMemcpyDeviceToHost)); fprintf(stderr, "[idkpt prof] cycles refill %llu node %llu leaf %llu other %llu | refills %llu lanes %llu | nodeSteps %llu lanes %llu | leafPhases %llu lanes %llu\n", (unsigned long long)d[4], (unsigned long long)d[5], (unsigned long long)d[6], (unsigned long long)d[7], (unsigned long long)d[8], (unsigned long long)d[9], (unsigned long long)d[10], (unsigned long long)d[11], (unsigned long long)d[12], (unsigned long long)d[13]); }
    s.RaysTraced = s.PrimaryRays + c[2]; // N per sample + every alive-queue entry that entered a bounce
    *out = s;
    return IDKPT_OK;
}

int32_t idkptResetStats(idkpt_ctx* ctx)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    ctx->evUsed = 0; ctx->traceMsAcc = 0.0; ctx->traceLaunchesAcc = 0;
    memset(ctx->hCounts, 0, MAX_DEPTH_SLOTS * 4);
    HIPC(hipMemset(ctx->counters64.p, 0, 128));
    return IDKPT_OK;
}

int32_t idkptEnableCounters(idkpt_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->counters = enable != 0; return IDKPT_OK; }
int32_t idkptEnableTiming(idkpt_ctx* ctx, int32_t enable) { if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT; ctx->timing = enable != 0; return IDKPT_OK; }

int32_t idkptGetImageDevicePtr(idkpt_ctx* ctx, int32_t image, void** outPtr, size_t* outBytes)
{
    if (!ctx || !outPtr) return IDKPT_ERR_INVALID_ARGUMENT;
    REQUIRE(image >= 0 && image < 3 && ctx->W > 0, "idkptGetImageDevicePtr: bad image / no size");
    *outPtr = image_ptr(ctx, image);
    if (outBytes) *outBytes = (size_t)ctx->W * ctx->rows * 16;
    return IDKPT_OK;
}

int32_t idkptSetStream(idkpt_ctx* ctx, void* hipStream)
{
    if (!ctx) return IDKPT_ERR_INVALID_ARGUMENT;
    HIPC(hipSetDevice(ctx->device));
    FLUSH();
    HIPC(hipStreamSynchronize(ctx->stream));
    if (hipStream) { if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream); ctx->stream = (hipStream_t)hipStream; ctx->ownStream = false; }
    else if (!ctx->ownStream) { HIPC(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->ownStream = true; }
    return IDKPT_OK;
}
int32_t idkptGetStream(idkpt_ctx* ctx, void** out) { if (!ctx || !out) return IDKPT_ERR_INVALID_ARGUMENT; *out = (void*)ctx->stream; return IDKPT_OK; }

} // extern "C"
