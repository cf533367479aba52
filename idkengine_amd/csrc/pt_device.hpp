// pt_device.hpp — device-side math, intersectors and shading for the gfx950 path tracer.
//
// Arithmetic contract (DESIGN.md "Numerics"): IEEE-754 binary32, one rounding per written operation, compiled with
// -ffp-contract=off; `/` and sqrt are the correctly rounded forms hipcc emits by default
// (-fhip-fp32-correctly-rounded-divide-sqrt); min/max are v_min_f32/v_max_f32 (minNum/maxNum); sin/cos/exp are the
// explicit polynomial forms below, never the ocml library.  What is computed follows the reference shaders:
//   Shaders/include/{Random,Sampling,Compression,Math,Pbr,IntersectionRoutines,Ray,Surface}.glsl,
//   Shaders/PathTracing/include/{Shading,RussianRoulette}.glsl  (paths relative to /root/reference/IDKEngine/Resource).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/idkpt_types.h"

#define DEV __device__ __forceinline__

namespace ptd {

struct f2 { float x, y; };
struct f3 { float x, y, z; };

DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV f3 splat3(float s) { return mk3(s, s, s); }
DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

DEV float gmin(float a, float b) { return __builtin_fminf(a, b); }
DEV float gmax(float a, float b) { return __builtin_fmaxf(a, b); }
DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
DEV float gabs(float a) { return __builtin_fabsf(a); }
DEV float gsqrt(float a) { return __builtin_sqrtf(a); }
DEV float gfloor(float a) { return __builtin_floorf(a); }
DEV float gfract(float a) { return a - __builtin_floorf(a); }
DEV float gsign(float a) { return a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f); }
DEV float gmix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
DEV f3 gmix(f3 x, f3 y, float a) { float ia = 1.0f - a; return mk3(x.x * ia + y.x * a, x.y * ia + y.y * a, x.z * ia + y.z * a); }
DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
DEV f3 normalize(f3 v) { float inv = 1.0f / gsqrt(dot(v, v)); return v * inv; }
DEV f3 reflect(f3 I, f3 N) { return I - (2.0f * dot(N, I)) * N; }
DEV f3 refract(f3 I, f3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return splat3(0.0f);
    return eta * I - (eta * d + gsqrt(k)) * N;
}
DEV float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

DEV void gsincos(float x, float* s, float* c)
{
    float kf = gfloor(x * 0.636619772367581343f + 0.5f);
    int k = (int)kf;
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    int q = k & 3;
    float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = (q == 1 || q == 2) ? -cc : cc;
}
DEV float gexp(float x)
{
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return __uint_as_float(0x7f800000u);
    float n = gfloor(x * 1.44269504088896341f + 0.5f);
    float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float z = r * r;
    float p = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f) * z + r + 1.0f;
    int e = (int)n;
    return p * __uint_as_float((uint32_t)(e + 127) << 23);
}

// ---- Random.glsl:4-33 ----
DEV uint32_t pcg_hash(uint32_t& seed)
{
    seed = seed * 747796405u + 2891336453u;
    uint32_t word = ((seed >> ((seed >> 28u) + 4u)) ^ seed) * 277803737u;
    return (word >> 22u) ^ word;
}
DEV float rnd01(uint32_t& seed) { return (float)pcg_hash(seed) * 2.3283064365386962890625e-10f; }

// ---- Sampling.glsl ----
#define PT_PI 3.14159265f
DEV f2 R2Sequence(uint32_t id)
{
    const float g = 1.32471795724474602596f;
    const float a1 = 1.0f / g;
    const float a2 = 1.0f / (g * g);
    f2 r; r.x = gfract((float)id * a1); r.y = gfract((float)id * a2);
    return r;
}
DEV f3 SampleSphere(float rnd0, float rnd1)
{
    float cosTheta = rnd0 * 2.0f - 1.0f;
    float phi = rnd1 * 2.0f * PT_PI;
    float sinTheta = gsqrt(1.0f - cosTheta * cosTheta);
    float sinPhi, cosPhi;
    gsincos(phi, &sinPhi, &cosPhi);
    return mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
DEV f3 CosineSampleHemisphere(f3 n, f2 uv) { return normalize(n + SampleSphere(uv.x, uv.y)); }
DEV f2 SampleDisk(uint32_t& seed)
{
    f2 p; float dist;
    float lastRnd = rnd01(seed);
    do {
        float thisRnd = rnd01(seed);
        p.x = lastRnd; p.y = thisRnd;
        dist = p.x * p.x + p.y * p.y;
        lastRnd = thisRnd;
    } while (dist > 1.0f);
    f2 r; r.x = p.x * 2.0f - 1.0f; r.y = p.y * 2.0f - 1.0f;
    return r;
}

// ---- Compression.glsl ----
DEV f3 DecompressSR11G11B10(uint32_t d)
{
    float r = (float)(d & 2047u), g = (float)((d >> 11) & 2047u), b = (float)((d >> 22) & 1023u);
    r = r / 2047.0f; g = g / 2047.0f; b = b / 1023.0f;
    return mk3(r * 2.0f - 1.0f, g * 2.0f - 1.0f, b * 2.0f - 1.0f);
}
DEV uint32_t CompressSR11G11B10(f3 d)
{
    f3 u = mk3(d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f);
    uint32_t r = (uint32_t)__builtin_rintf(u.x * 2047.0f), g = (uint32_t)__builtin_rintf(u.y * 2047.0f), b = (uint32_t)__builtin_rintf(u.z * 1023.0f);
    return (b << 22) | (g << 11) | r;
}
DEV f2 EncodeUnitVec(f3 n)
{
    n = n / (gabs(n.x) + gabs(n.y) + gabs(n.z));
    f2 xy; xy.x = n.x; xy.y = n.y;
    if (!(n.z > 0.0f)) {
        f2 w; w.x = 1.0f - gabs(xy.y); w.y = 1.0f - gabs(xy.x);
        if (xy.x < 0.0f) w.x = -w.x;
        if (xy.y < 0.0f) w.y = -w.y;
        xy = w;
    }
    f2 r; r.x = xy.x * 0.5f + 0.5f; r.y = xy.y * 0.5f + 0.5f;
    return r;
}
DEV f3 DecodeUnitVec(float fx, float fy)
{
    fx = fx * 2.0f - 1.0f; fy = fy * 2.0f - 1.0f;
    f3 n = mk3(fx, fy, 1.0f - gabs(fx) - gabs(fy));
    float t = gmax(-n.z, 0.0f);
    n.x += n.x >= 0.0f ? -t : t;
    n.y += n.y >= 0.0f ? -t : t;
    return normalize(n);
}

// ---- Math.glsl ----
DEV f3 mat4_mul_xyz(const float* m, float x, float y, float z, float w)
{
    return mk3(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w,
               ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
               ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w);
}
DEV f3 GetWorldSpaceDirection(const float* invProj, const float* invView, float nx, float ny)
{
    float rx = invProj[0] * nx + invProj[4] * ny;
    float ry = invProj[1] * nx + invProj[5] * ny;
    return normalize(mat4_mul_xyz(invView, rx, ry, -1.0f, 0.0f));
}
DEV f3 CubemapFaceNormal(f3 d)
{
    f3 a = mk3(gabs(d.x), gabs(d.y), gabs(d.z));
    float mx = (a.x >= gmax(a.y, a.z)) ? 1.0f : 0.0f;
    float my = (a.y >= gmax(a.z, a.x)) ? 1.0f : 0.0f;
    float mz = (a.z >= gmax(a.x, a.y)) ? 1.0f : 0.0f;
    return mk3(mx * -gsign(d.x), my * -gsign(d.y), mz * -gsign(d.z));
}
DEV f3 Interpolate(f3 p0, f3 p1, f3 p2, f3 b) { return p0 * b.x + p1 * b.y + p2 * b.z; }

// rows of a row_major mat4x3 as three float4
struct M34 { float4 r0, r1, r2; };
DEV f3 xform34(const M34& R, f3 p, float w)
{
    return mk3(((R.r0.x * p.x + R.r0.y * p.y) + R.r0.z * p.z) + R.r0.w * w,
               ((R.r1.x * p.x + R.r1.y * p.y) + R.r1.z * p.z) + R.r1.w * w,
               ((R.r2.x * p.x + R.r2.y * p.y) + R.r2.z * p.z) + R.r2.w * w);
}
DEV f3 xform34_transposed3(const M34& R, f3 n)
{
    return mk3((n.x * R.r0.x + n.y * R.r1.x) + n.z * R.r2.x,
               (n.x * R.r0.y + n.y * R.r1.y) + n.z * R.r2.y,
               (n.x * R.r0.z + n.y * R.r1.z) + n.z * R.r2.z);
}

// ---- Pbr.glsl ----
DEV float BaseReflectivity(float n1, float n2) { float r0 = (n1 - n2) / (n1 + n2); r0 *= r0; return r0; }
DEV float FresnelSchlick(float f0, float f90, float cosTheta) { return f0 + (f90 - f0) * pow5(1.0f - cosTheta); }

// ---- IntersectionRoutines.glsl ----
#define PT_FLOAT_MAX 3.4028235e+38f

// :6-23.  Returns hit flag; bary.x is not needed by callers beyond the sign test.
DEV bool RayTriangleIntersect(f3 ro, f3 rd, f3 p0, f3 p1, f3 p2, float* by, float* bz, float* t)
{
    f3 p1p0 = p1 - p0;
    f3 p2p0 = p2 - p0;
    f3 rop0 = ro - p0;
    f3 normal = cross(p1p0, p2p0);
    f3 q = cross(rop0, rd);
    float invDet = 1.0f / dot(rd, normal);
    *t = dot(-normal, rop0) * invDet;
    *by = dot(-q, p2p0) * invDet;
    *bz = dot(q, p1p0) * invDet;
    float bx = 1.0f - *by - *bz;
    return bx >= 0.0f && *by >= 0.0f && *bz >= 0.0f && *t >= 0.0f;
}
// :25-46
// The x/y slabs go through 2-wide vectors so that the compiler emits v_pk_add_f32 / v_pk_mul_f32 (gfx950 packed FP32: two IEEE
// operations per instruction, same roundings); the traversal kernel is VALU-issue bound, this takes 8 of ~65 instructions off a node step.
typedef float v2f __attribute__((ext_vector_type(2)));
DEV bool RayBoxIntersect(f3 o, f3 invDir, float4 bmin, float4 bmax, float* t1, float* t2Out = nullptr)
{
    const v2f oxy = {o.x, o.y}, ixy = {invDir.x, invDir.y};
    const v2f lo2 = (v2f{bmin.x, bmin.y} - oxy) * ixy, hi2 = (v2f{bmax.x, bmax.y} - oxy) * ixy;
    f3 t0s = mk3(lo2.x, lo2.y, (bmin.z - o.z) * invDir.z);
    f3 t1s = mk3(hi2.x, hi2.y, (bmax.z - o.z) * invDir.z);
    f3 tsm = mk3(gmin(t0s.x, t1s.x), gmin(t0s.y, t1s.y), gmin(t0s.z, t1s.z));
    f3 tbg = mk3(gmax(t0s.x, t1s.x), gmax(t0s.y, t1s.y), gmax(t0s.z, t1s.z));
    *t1 = gmax(tsm.x, gmax(tsm.y, gmax(tsm.z, 0.0f)));
    float t2 = gmin(tbg.x, gmin(tbg.y, tbg.z));
    if (t2Out) *t2Out = t2;
    return *t1 <= t2;
}
// :48-69
DEV bool RaySphereIntersect(f3 ro, f3 rd, f3 position, float radius, float* t1, float* t2)
{
    *t1 = PT_FLOAT_MAX; *t2 = PT_FLOAT_MAX;
    f3 sphereToRay = ro - position;
    float b = dot(rd, sphereToRay);
    float c = dot(sphereToRay, sphereToRay) - radius * radius;
    float discriminant = b * b - c;
    if (discriminant < 0.0f) return false;
    float squareRoot = gsqrt(discriminant);
    *t1 = -b - squareRoot;
    *t2 = -b + squareRoot;
    return *t1 <= *t2 && *t2 > 0.0f;
}

} // namespace ptd
