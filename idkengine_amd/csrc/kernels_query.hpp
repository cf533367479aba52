// kernels_query.hpp — adjacent consumers of the traversal core: batched ray queries (idkptTraceRays) and ray-traced point-light shadows (idkptTraceShadows).
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

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

// Closest-hit queries on k_trace2's scheduler (round 4).  The thread-per-ray kernel above runs TraceRay's nested loops with whatever lanes are still busy (1 559 Mray/s on one
// primary ray per pixel of the headline view, 575 where every ray traverses: 2.5x the time k_trace2 needs for the same rays).  k_query_prepare does what TraceRay does before
// its traversal (BVHIntersect.glsl:185-203: T = maxDist, the sphere lights; with one BLAS instance also RayTransform, 1/dir and the root-box test of :32-39) with all lanes
// busy and leaves a trace-ready record per ray — the initial T in record[1].w, the light it belongs to in record[2].w (Frame::queryMode: k_trace2 starts from them instead of
// FLOAT_MAX / 0) — rays that cannot enter the one BLAS get their final record right here; k_trace2 (MODE 0 / 1 / 2 by scene, as for frames) stores its 32-B hit records
// straight into the caller's idkpt_hit array (same layout), and k_query_finish sets `Hit` = T != maxDist (:290).  Same operands, same order per ray: bit-identical to the
// thread-per-ray kernel and the oracle (tests/test_gpu_queries.py).  TraceRayAny takes the same route through k_trace2's ANY instantiations (left child first inside a BLAS,
// the first intersection found ends the ray; a light in front of maxDist ends it right here).
__global__ __launch_bounds__(256) void k_query_prepare(DScene s, Frame f, const idkpt_ray* rays, idkpt_hit* out, uint32_t N, int traceLights, int anyHit, TraceBufs tr, uint32_t* list, uint32_t* listCount)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < N) {
        const float4 a = ((const float4*)rays)[2 * (size_t)i], b = ((const float4*)rays)[2 * (size_t)i + 1];
        const f3 ro = mk3(a.x, a.y, a.z), rd = mk3(b.x, b.y, b.z);
        const float maxDist = a.w;
        float T = maxDist; uint32_t xf = 0u;
        bool lightEnds = false;                                                    // TraceRayAny (:303-314): the first light in front of maxDist is the answer
        if (traceLights) {
            for (int k = 0; k < s.lightCount && !lightEnds; k++) {
                const GpuLight& l = s.lights[k];
                float tMin, tMax;
                if (RaySphereIntersect(ro, rd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < T) { T = tMin < 0.0f ? tMax : tMin; xf = (uint32_t)k; lightEnds = anyHit != 0; }
            }
        }
        if (lightEnds) {
            ((float4*)out)[2 * (size_t)i] = make_float4(T, 0.0f, 0.0f, __uint_as_float(~0u));
            ((uint4*)out)[2 * (size_t)i + 1] = make_uint4(xf, 1u, 0u, 0u);
        } else
        if (f.useTlas) {
            keep = s.tlasCount > 0;
            tr.rec[4 * (size_t)i] = make_float4(ro.x, ro.y, ro.z, 0.0f); tr.rec[4 * (size_t)i + 1] = make_float4(rd.x, rd.y, rd.z, T);
            tr.rec[4 * (size_t)i + 2] = make_float4(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z, __uint_as_float(xf));
        } else if (s.instanceCount > 1) {
            keep = true;
            tr.rec[4 * (size_t)i] = make_float4(ro.x, ro.y, ro.z, 0.0f); tr.rec[4 * (size_t)i + 1] = make_float4(rd.x, rd.y, rd.z, T);
            tr.rec[4 * (size_t)i + 2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(xf));
        } else {
            const GpuBlasInstance inst = s.instances[0];
            const M34 inv = load_inv_model(s, inst.MeshTransformId);
            const f3 lo = xform34(inv, ro, 1.0f), ld = xform34(inv, rd, 0.0f);
            const f3 iv = mk3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
            const float4* root = s.nodes + 2 * (size_t)s.descs[inst.BlasId].NodeOffset + 2;
            float t1;
            const float rootT = RayBoxIntersect(lo, iv, root[0], root[1], &t1) ? t1 : __builtin_inff();
            keep = rootT < T;                                                      // (:32-39: the test k_trace2 would do first)
            tr.rec[4 * (size_t)i] = make_float4(lo.x, lo.y, lo.z, rootT); tr.rec[4 * (size_t)i + 1] = make_float4(ld.x, ld.y, ld.z, T);
            tr.rec[4 * (size_t)i + 2] = make_float4(iv.x, iv.y, iv.z, __uint_as_float(xf));
        }
        if (!keep && !lightEnds) {
            ((float4*)out)[2 * (size_t)i] = make_float4(T, 0.0f, 0.0f, __uint_as_float(~0u));
            ((uint4*)out)[2 * (size_t)i + 1] = make_uint4(xf, T != maxDist ? 1u : 0u, 0u, 0u);
        }
    }
    // survivors -> work list (any order: results are stored per ray); one atomic per wave
    const unsigned long long m = __ballot(keep);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(listCount, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0);
    if (keep) list[base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = i;
}
__global__ __launch_bounds__(256) void k_query_finish(const idkpt_ray* rays, idkpt_hit* out, const uint32_t* list, const uint32_t* listCount)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *listCount) return;
    const uint32_t i = list[k];
    const float T = ((const float*)out)[8 * (size_t)i], maxDist = ((const float*)rays)[8 * (size_t)i + 3];
    ((uint32_t*)out)[8 * (size_t)i + 5] = T != maxDist ? 1u : 0u;
    ((uint32_t*)out)[8 * (size_t)i + 6] = 0u; ((uint32_t*)out)[8 * (size_t)i + 7] = 0u;
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
