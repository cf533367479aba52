// ORACLE (test infrastructure, never shipped, never on the product path).
// PARITY: the GLSL half of the path (FirstHit / NHit / FinalDraw / CountingSort and everything they include —
// ref_math.h, ref_pathtracer.cpp) is PINNED against the reference itself: the reference's own shaders, read from
// /root/reference and executed by Mesa llvmpipe (oracle/glref/), produce the vectors under tests/golden/glref/, and
// tests/test_glref.py holds this restatement to them (every bounce from identical inputs: no flipped decision, every
// value within 1e-4 relative, ~97 % of the words bit-identical; llvmpipe's own `/`, inversesqrt and transcendentals
// are the remainder).  The C# half (ref_bvh_build.cpp, ref_cpu_baseline.cpp) stays UNPINNED: no .NET here, and the
// reference has no tests or golden vectors; it is anchored by published known-answers of the
// third-party algorithms it contains (tests/test_oracle_kats.py), by the brute-force / metamorphic second opinion
// (tests/test_metamorphic.py) and by the fact that the reference's shaders traverse the trees it builds to the same hits.
//
// ref_math.h — the arithmetic contract shared by every oracle function: IEEE-754 binary32, one rounding per
// written operation, NO contraction (compile with -ffp-contract=off), left-to-right evaluation of GLSL
// built-ins.  GLSL leaves the precision of sin/cos/exp/pow/normalize/`/` to the driver; the choices made here
// (each documented at its definition) are one legal execution of the reference shaders and are what the HIP
// path is held to bit-for-bit.
//
// Reference files restated here (relative to /root/reference/IDKEngine/Resource/Shaders/include):
//   Random.glsl:4-33, Sampling.glsl:4-19,59-72,86-114, Compression.glsl:11-32,44-84, Math.glsl:6-15,41-57,104-137,
//   Pbr.glsl:19-27,64-67, IntersectionRoutines.glsl:6-69, Ray.glsl:7-12
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

namespace ref {

struct v2 { float x, y; };
struct v3 { float x, y, z; };

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 V3s(float s) { v3 r = {s, s, s}; return r; }
static inline v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 operator*(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); } // every GLSL `/` is an IEEE division
static inline v3 operator/(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// GLSL min/max as executed by AMD hardware (v_min_f32/v_max_f32 = IEEE-754 minNum/maxNum: a NaN operand is
// ignored).  IntersectionRoutines.glsl:25-46 relies on this for the 0*inf slab case.
// Equal operands (only +-0 can differ in bits): min returns -0, max returns +0, like the hardware.
static inline float gmin(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == b) return u2f(f2u(a) | f2u(b)); return b < a ? b : a; }
static inline float gmax(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == b) return u2f(f2u(a) & f2u(b)); return b > a ? b : a; }
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline float gabs(float a) { return u2f(f2u(a) & 0x7fffffffu); }
static inline float gsqrt(float a) { return sqrtf(a); } // correctly rounded
static inline float gfloor(float a) { return floorf(a); }
static inline float gfract(float a) { return a - floorf(a); } // GLSL fract
static inline float gsign(float a) { return a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f); }
static inline float gmix(float x, float y, float a) { return x * (1.0f - a) + y * a; } // GLSL spec formula
static inline v3 gmix(v3 x, v3 y, float a) { float ia = 1.0f - a; return V3(x.x * ia + y.x * a, x.y * ia + y.y * a, x.z * ia + y.z * a); }

static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // ((x+y)+z)
static inline float dot2(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline v3 cross(v3 a, v3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); } // GLSL spec form
// normalize(v) := v * (1 / sqrt(dot(v,v)))   [one IEEE sqrt, one IEEE div, three mul]
static inline v3 normalize(v3 v) { float inv = 1.0f / gsqrt(dot(v, v)); return v * inv; }
static inline float length(v3 v) { return gsqrt(dot(v, v)); }
static inline v3 reflect(v3 I, v3 N) { return I - (2.0f * dot(N, I)) * N; } // GLSL spec: I - 2*dot(N,I)*N
static inline v3 refract(v3 I, v3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3s(0.0f);
    return eta * I - (eta * d + gsqrt(k)) * N;
}
// pow(x, 5.0) in Pbr.glsl:64-67 := (x*x)*(x*x)*x
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// sin/cos: Cody-Waite reduction by pi/2 (three constants, exact products for |k| < 2^11), cephes sinf/cosf minimax
// polynomials on [-pi/4, pi/4].  Only called with phi in [0, 2*pi] (Sampling.glsl:59-69).
static inline void gsincos(float x, float* s, float* c)
{
    float kf = gfloor(x * 0.636619772367581343f + 0.5f);
    int k = (int)kf;
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    switch (k & 3) {
        case 0: *s = sp;  *c = cp;  break;
        case 1: *s = cp;  *c = -sp; break;
        case 2: *s = -sp; *c = -cp; break;
        default: *s = -cp; *c = sp; break;
    }
}
// exp(x): cephes expf (round-to-nearest reduction by ln2 in two parts, degree-5 polynomial); exp(x) := 0 for x < -87.
static inline float gexp(float x)
{
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return u2f(0x7f800000u);
    float n = gfloor(x * 1.44269504088896341f + 0.5f);
    float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float z = r * r;
    float p = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f) * z + r + 1.0f;
    int e = (int)n; // in [-126, 127]
    return p * u2f((uint32_t)(e + 127) << 23);
}
static inline v3 gexp3(v3 v) { return V3(gexp(v.x), gexp(v.y), gexp(v.z)); }

// ---------------- Random.glsl:4-33 (PCG hash, uint32 wraparound) ----------------
struct Rng { uint32_t seed; };
static inline uint32_t pcg_hash(uint32_t* seed)
{
    *seed = *seed * 747796405u + 2891336453u;
    uint32_t word = ((*seed >> ((*seed >> 28u) + 4u)) ^ *seed) * 277803737u;
    return (word >> 22u) ^ word;
}
// float(u)/4294967296.0: u->float rounds to nearest-even, so the result can be exactly 1.0
static inline float rnd01(Rng* r) { return (float)pcg_hash(&r->seed) * 2.3283064365386962890625e-10f; }

// ---------------- Sampling.glsl ----------------
#define REF_PI 3.14159265f
// Sampling.glsl:4-13. g, a1, a2 are folded in binary32 (const float expressions); id*a1 = float(id)*a1
static inline v2 R2Sequence(uint32_t id)
{
    const float g = 1.32471795724474602596f;
    const float a1 = 1.0f / g;
    const float a2 = 1.0f / (g * g);
    v2 r = {gfract((float)id * a1), gfract((float)id * a2)};
    return r;
}
static inline v2 DecorrelateSequence(v2 s, v2 noise) { v2 r = {gfract(s.x + noise.x), gfract(s.y + noise.y)}; return r; } // :15-19
static inline v3 SampleSphere(float rnd0, float rnd1) // :59-68
{
    float cosTheta = rnd0 * 2.0f - 1.0f;
    float phi = rnd1 * 2.0f * REF_PI;
    float sinTheta = gsqrt(1.0f - cosTheta * cosTheta);
    float sinPhi, cosPhi;
    gsincos(phi, &sinPhi, &cosPhi);
    return V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
static inline v3 CosineSampleHemisphere(v3 normal, v2 uv) { return normalize(normal + SampleSphere(uv.x, uv.y)); } // :86-89
static inline v2 SampleDisk(Rng* rng) // :98-114 (rejection on the unit-square quadrant, then *2-1)
{
    v2 p; float dist;
    float lastRnd = rnd01(rng);
    do {
        float thisRnd = rnd01(rng);
        p.x = lastRnd; p.y = thisRnd;
        dist = dot2(p, p);
        lastRnd = thisRnd;
    } while (dist > 1.0f);
    v2 r = {p.x * 2.0f - 1.0f, p.y * 2.0f - 1.0f};
    return r;
}

// ---------------- Compression.glsl ----------------
static inline v3 DecompressUR11G11B10(uint32_t d) // :11-22 (three IEEE divisions)
{
    float r = (float)((d >> 0) & 2047u), g = (float)((d >> 11) & 2047u), b = (float)((d >> 22) & 1023u);
    return V3(r / 2047.0f, g / 2047.0f, b / 1023.0f);
}
static inline v3 DecompressSR11G11B10(uint32_t d) { v3 u = DecompressUR11G11B10(d); return V3(u.x * 2.0f - 1.0f, u.y * 2.0f - 1.0f, u.z * 2.0f - 1.0f); } // :29-32
// round() := round-half-to-even (v_rndne_f32 / C# MathF.Round, Utils/Compression.cs:26-40)
static inline uint32_t CompressUR11G11B10(v3 d) // :1-10
{
    uint32_t r = (uint32_t)rintf(d.x * 2047.0f), g = (uint32_t)rintf(d.y * 2047.0f), b = (uint32_t)rintf(d.z * 1023.0f);
    return (b << 22) | (g << 11) | r;
}
static inline uint32_t CompressSR11G11B10(v3 d) { return CompressUR11G11B10(V3(d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f)); } // :24-28
static inline void unpackUnorm4x8(uint32_t p, float out[4]) // GLSL spec: f / 255.0
{
    for (int i = 0; i < 4; i++) out[i] = (float)((p >> (8 * i)) & 255u) / 255.0f;
}
static inline v2 OctWrap(v2 v) // :44-49
{
    v2 w = {1.0f - gabs(v.y), 1.0f - gabs(v.x)};
    if (v.x < 0.0f) w.x = -w.x;
    if (v.y < 0.0f) w.y = -w.y;
    return w;
}
static inline v2 EncodeUnitVec(v3 n) // :54-60
{
    n = n / (gabs(n.x) + gabs(n.y) + gabs(n.z));
    v2 xy = {n.x, n.y};
    if (!(n.z > 0.0f)) xy = OctWrap(xy);
    v2 r = {xy.x * 0.5f + 0.5f, xy.y * 0.5f + 0.5f};
    return r;
}
static inline v3 DecodeUnitVec(v2 f) // :63-73
{
    f.x = f.x * 2.0f - 1.0f; f.y = f.y * 2.0f - 1.0f;
    v3 n = V3(f.x, f.y, 1.0f - gabs(f.x) - gabs(f.y));
    float t = gmax(-n.z, 0.0f);
    n.x += n.x >= 0.0f ? -t : t;
    n.y += n.y >= 0.0f ? -t : t;
    return normalize(n);
}
static inline v3 ReconstructPackedNormal(v2 v) // :78-84 (note: dot uses the un-remapped v, as the reference does)
{
    v3 r;
    r.x = v.x * 2.0f - 1.0f; r.y = v.y * 2.0f - 1.0f;
    r.z = gsqrt(gmax(1.0f - dot2(v, v), 0.0f));
    return r;
}

// ---------------- Math.glsl ----------------
// GLSL mat4 * vec4 with the matrix given as the 16 floats the host uploaded (OpenTK memory order): GLSL column c is
// m[4c..4c+3]; result_i = ((m[i]*v.x + m[4+i]*v.y) + m[8+i]*v.z) + m[12+i]*v.w
static inline v3 mat4_mul_xyz(const float* m, float x, float y, float z, float w)
{
    return V3(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w,
              ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
              ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w);
}
static inline v3 GetWorldSpaceDirection(const float* invProj, const float* invView, v2 ndc) // Math.glsl:6-15
{
    float rx = invProj[0] * ndc.x + invProj[4] * ndc.y; // mat2(inverseProj) * ndc
    float ry = invProj[1] * ndc.x + invProj[5] * ndc.y;
    return normalize(mat4_mul_xyz(invView, rx, ry, -1.0f, 0.0f));
}
static inline v3 CubemapFaceNormal(v3 d) // :41-47
{
    v3 a = V3(gabs(d.x), gabs(d.y), gabs(d.z));
    float mx = (a.x >= gmax(a.y, a.z)) ? 1.0f : 0.0f;
    float my = (a.y >= gmax(a.z, a.x)) ? 1.0f : 0.0f;
    float mz = (a.z >= gmax(a.x, a.y)) ? 1.0f : 0.0f;
    return V3(mx * -gsign(d.x), my * -gsign(d.y), mz * -gsign(d.z));
}
static inline v3 Interpolate(v3 p0, v3 p1, v3 p2, v3 b) { return p0 * b.x + p1 * b.y + p2 * b.z; } // :49-52
static inline v2 Interpolate2(v2 p0, v2 p1, v2 p2, v3 b) { v2 r = {p0.x * b.x + p1.x * b.y + p2.x * b.z, p0.y * b.x + p1.y * b.y + p2.y * b.z}; return r; }
static inline v3 GetTriangleNormal(v3 p0, v3 p1, v3 p2) { return normalize(cross(p1 - p0, p2 - p0)); } // :104-110

// mat4x3 (row_major in memory: R[3][4]) * vec4(p, w):  out_i = ((R[i][0]*p.x + R[i][1]*p.y) + R[i][2]*p.z) + R[i][3]*w
static inline v3 xform34(const float R[3][4], v3 p, float w)
{
    return V3(((R[0][0] * p.x + R[0][1] * p.y) + R[0][2] * p.z) + R[0][3] * w,
              ((R[1][0] * p.x + R[1][1] * p.y) + R[1][2] * p.z) + R[1][3] * w,
              ((R[2][0] * p.x + R[2][1] * p.y) + R[2][2] * p.z) + R[2][3] * w);
}
// mat3(transpose(M)) * n for the row_major mat4x3 M: out_i = (n.x*R[0][i] + n.y*R[1][i]) + n.z*R[2][i]
static inline v3 xform34_transposed3(const float R[3][4], v3 n)
{
    return V3((n.x * R[0][0] + n.y * R[1][0]) + n.z * R[2][0],
              (n.x * R[0][1] + n.y * R[1][1]) + n.z * R[2][1],
              (n.x * R[0][2] + n.y * R[1][2]) + n.z * R[2][2]);
}

// ---------------- Pbr.glsl ----------------
static inline float BaseReflectivity(float n1, float n2) { float r0 = (n1 - n2) / (n1 + n2); r0 *= r0; return r0; } // :19-27
static inline float FresnelSchlick(float f0, float f90, float cosTheta) { return f0 + (f90 - f0) * pow5(1.0f - cosTheta); } // :64-67

// ---------------- IntersectionRoutines.glsl ----------------
struct Ray { v3 o, d; };
#define REF_FLOAT_MAX 3.4028235e+38f

// :6-23 (iq cross-product form; accept iff all(bary, t) >= 0)
static inline bool RayTriangleIntersect(const Ray& ray, v3 p0, v3 p1, v3 p2, v3* bary, float* t)
{
    v3 p1p0 = p1 - p0;
    v3 p2p0 = p2 - p0;
    v3 rop0 = ray.o - p0;
    v3 normal = cross(p1p0, p2p0);
    v3 q = cross(rop0, ray.d);
    float invDet = 1.0f / dot(ray.d, normal);
    *t = dot(-normal, rop0) * invDet;
    bary->y = dot(-q, p2p0) * invDet;
    bary->z = dot(q, p1p0) * invDet;
    bary->x = 1.0f - bary->y - bary->z;
    return bary->x >= 0.0f && bary->y >= 0.0f && bary->z >= 0.0f && *t >= 0.0f;
}
// :25-46. invDir = 1/dir is loop invariant and passed in.
static inline bool RayBoxIntersect(v3 o, v3 invDir, const float* bmin, const float* bmax, float* t1)
{
    v3 t0s = (V3(bmin[0], bmin[1], bmin[2]) - o) * invDir;
    v3 t1s = (V3(bmax[0], bmax[1], bmax[2]) - o) * invDir;
    v3 tsm = V3(gmin(t0s.x, t1s.x), gmin(t0s.y, t1s.y), gmin(t0s.z, t1s.z));
    v3 tbg = V3(gmax(t0s.x, t1s.x), gmax(t0s.y, t1s.y), gmax(t0s.z, t1s.z));
    *t1 = gmax(tsm.x, gmax(tsm.y, gmax(tsm.z, 0.0f)));
    float t2 = gmin(tbg.x, gmin(tbg.y, tbg.z));
    return *t1 <= t2;
}
// :48-69 (assumes unit direction)
static inline bool RaySphereIntersect(const Ray& ray, v3 position, float radius, float* t1, float* t2)
{
    *t1 = REF_FLOAT_MAX; *t2 = REF_FLOAT_MAX;
    v3 sphereToRay = ray.o - position;
    float b = dot(ray.d, sphereToRay);
    float c = dot(sphereToRay, sphereToRay) - radius * radius;
    float discriminant = b * b - c;
    if (discriminant < 0.0f) return false;
    float squareRoot = gsqrt(discriminant);
    *t1 = -b - squareRoot;
    *t2 = -b + squareRoot;
    return *t1 <= *t2 && *t2 > 0.0f;
}

} // namespace ref
