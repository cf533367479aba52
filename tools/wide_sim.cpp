// wide_sim.cpp — CPU model of the wide-node walk (idkengine_amd/csrc/wide_nodes.hpp) beside the reference's BVH2 walk (BVHIntersect.glsl:27-105), ray by ray.
// Developer / test tool: reports round trips per ray for both structures, how many rays the wide walk flags (and why), and checks that every UNFLAGGED ray
// gets the reference's hit bit for bit.  Scene files come from tools/dump_scene_for_sim.py.
//   g++ -O2 -std=c++17 -ffp-contract=off -fopenmp -I idkengine_amd/csrc tools/wide_sim.cpp -o /tmp/wide_sim
//   /tmp/wide_sim scene.bin <view: headline|interior|atrium|cornell|cam:ex,ey,ez,dx,dy,dz,fovy> [width height] [policy bits] [stack rows]
// Exit status 2 if a ray the wide walk vouches for differs from the BVH2 walk.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>
#include "wide_nodes.hpp"

using wide::Bvh2Node;
struct V3 { float x, y, z; };
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static inline V3 normalize(V3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
#define FLOAT_MAX 3.4028235e+38f

// IntersectionRoutines.glsl:6-23
static inline bool ray_tri(V3 ro, V3 rd, V3 p0, V3 p1, V3 p2, float* by, float* bz, float* t)
{
    V3 p1p0 = p1 - p0, p2p0 = p2 - p0, rop0 = ro - p0;
    V3 normal = cross(p1p0, p2p0), q = cross(rop0, rd);
    float invDet = 1.0f / dot(rd, normal);
    *t = dot(neg(normal), rop0) * invDet;
    *by = dot(neg(q), p2p0) * invDet;
    *bz = dot(q, p1p0) * invDet;
    float bx = 1.0f - *by - *bz;
    return bx >= 0.0f && *by >= 0.0f && *bz >= 0.0f && *t >= 0.0f;
}
// :25-40
static inline bool ray_box2(V3 o, V3 inv, const float* mn, const float* mx, float* t1, float* t2o)
{
    float t0x = (mn[0] - o.x) * inv.x, t0y = (mn[1] - o.y) * inv.y, t0z = (mn[2] - o.z) * inv.z;
    float t1x = (mx[0] - o.x) * inv.x, t1y = (mx[1] - o.y) * inv.y, t1z = (mx[2] - o.z) * inv.z;
    float sx = fminf(t0x, t1x), sy = fminf(t0y, t1y), sz = fminf(t0z, t1z);
    float bx = fmaxf(t0x, t1x), by = fmaxf(t0y, t1y), bz = fmaxf(t0z, t1z);
    *t1 = fmaxf(sx, fmaxf(sy, fmaxf(sz, 0.0f)));
    float t2 = fminf(bx, fminf(by, bz));
    *t2o = t2;
    return *t1 <= t2;
}
static inline bool ray_box(V3 o, V3 inv, const float* mn, const float* mx, float* t1) { float t2; return ray_box2(o, inv, mn, mx, t1, &t2); }

struct Scene { std::vector<Bvh2Node> nodes; std::vector<float> tv; int nTris; };
struct Hit { float T, by, bz; uint32_t tri; };
struct Cnt { uint64_t pairs = 0, tris = 0, wnodes = 0, wleaves = 0, wleafPass = 0, wtris = 0, popSkips = 0, spHist[64] = {0}, refSpHist[64] = {0}; };

static Hit ref_trace(const Scene& s, V3 ro, V3 rd, float T0, Cnt& c)
{
    Hit h = {T0, 0, 0, ~0u};
    V3 inv = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    float tl, tr;
    if (!(ray_box(ro, inv, s.nodes[1].mn, s.nodes[1].mx, &tl) && tl < h.T)) return h;
    uint32_t stack[128]; int sp = 0; uint32_t top = 2;
    while (true) {
        c.pairs++;
        const Bvh2Node& L = s.nodes[top]; const Bvh2Node& R = s.nodes[top + 1];
        bool hitL = ray_box(ro, inv, L.mn, L.mx, &tl) && tl <= h.T;
        bool hitR = ray_box(ro, inv, R.mn, R.mx, &tr) && tr <= h.T;
        bool iL = hitL && L.triCount > 0, iR = hitR && R.triCount > 0;
        if (iL || iR) {
            uint32_t first = iL ? L.startOrChild : R.startOrChild;
            uint32_t end = !iR ? (L.startOrChild + L.triCount) : (R.startOrChild + R.triCount);
            for (uint32_t i = first; i < end; i++) {
                c.tris++;
                const float* v = &s.tv[12 * (size_t)i];
                float by, bz, t;
                if (ray_tri(ro, rd, {v[0], v[1], v[2]}, {v[4], v[5], v[6]}, {v[8], v[9], v[10]}, &by, &bz, &t) && t < h.T) { h.T = t; h.by = by; h.bz = bz; h.tri = i; }
            }
        }
        bool tL = hitL && L.triCount == 0, tR = hitR && R.triCount == 0;
        if (tL || tR) {
            if (tL && tR) { bool lc = tl < tr; top = lc ? L.startOrChild : R.startOrChild; stack[sp++] = lc ? R.startOrChild : L.startOrChild; }
            else top = tL ? L.startOrChild : R.startOrChild;
        } else { if (sp == 0) break; top = stack[--sp]; }
    }
    return h;
}

// policy bits: 1 = stack entries carry t1 and are skipped when popped beyond T * CULL; 2 = only the nearest child is chosen, the others pushed in slot order (no sort)
struct WideResult { Hit h; uint32_t flags; };   // flags: 1 inv not finite, 2 overflow, 4 second candidate in window, 8 leaf t1 outside window, 16 assumption violated by a tested triangle, 32 near-miss of a box test, 64 best hit is a marked (PreSplit) triangle
static WideResult wide_trace(const Scene& s, const wide::HostBuild& W, V3 ro, V3 rd, float T0, int policy, int stackCap, Cnt& c)
{
    WideResult r; r.h = {T0, 0, 0, ~0u}; r.flags = 0;
    V3 inv = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    if (!(fabsf(inv.x) < INFINITY && fabsf(inv.y) < INFINITY && fabsf(inv.z) < INFINITY)) { r.flags = 1; return r; }
    float tl;
    if (!(ray_box(ro, inv, s.nodes[1].mn, s.nodes[1].mx, &tl) && tl < r.h.T)) return r;
    const float roA[3] = {ro.x, ro.y, ro.z}, invA[3] = {inv.x, inv.y, inv.z};
    uint32_t stk[256]; float stkT[256]; int sp = 0;
    uint32_t cur = 0; float curT = 0.0f; bool have = true;           // wide node 0 = the root pair
    float second = T0, leafT1 = 0.0f; int maxSp = 0; bool bestMarked = false;
    while (true) {
        if (!have) {
            if (sp == 0) break;
            sp--; cur = stk[sp]; curT = stkT[sp];
            if ((policy & 1) && curT > r.h.T * wide::CULL) { c.popSkips++; continue; }
        }
        have = false;
        if (cur & wide::LEAF_BIT) {
            c.wleaves++;
            const float* rec = &W.leafRecs[4 * (size_t)(cur & ~wide::LEAF_BIT)];
            Bvh2Node ln; memcpy(&ln, rec, 32);
            float t1, t2;
            const bool boxHit = ray_box2(ro, inv, ln.mn, ln.mx, &t1, &t2);
            if (!boxHit && t1 <= t2 * wide::NEAR_MISS) r.flags |= 32;
            if (!(boxHit && t1 <= r.h.T * wide::CULL)) continue;
            c.wleafPass++;
            for (uint32_t i = 0; i < ln.triCount; i++) {
                c.wtris++;
                const float* v = rec + 8 + 12 * (size_t)i;
                float by, bz, t;
                if (ray_tri(ro, rd, {v[0], v[1], v[2]}, {v[4], v[5], v[6]}, {v[8], v[9], v[10]}, &by, &bz, &t)) {
                    const uint32_t id = ln.startOrChild + i;
                    uint32_t mk; memcpy(&mk, v + 3, 4);
                    if (!mk && t1 > t * wide::ASSUME) r.flags |= 16;
                    if (t < r.h.T) { if (id != r.h.tri) second = fminf(second, r.h.T); r.h.T = t; r.h.by = by; r.h.bz = bz; r.h.tri = id; leafT1 = t1; bestMarked = mk != 0; }
                    else if (id != r.h.tri) second = fminf(second, t);
                }
            }
            continue;
        }
        c.wnodes++;
        const wide::Node& w = W.nodes[cur];
        float t1[4];
        bool nm = false;
        uint32_t mask = wide::test_node(w, roA, invA, r.h.T * wide::CULL, t1, &nm);
        if (nm) r.flags |= 32;
        if (!mask) continue;
        // order: nearest first
        int order[4], n = 0;
        for (int k = 0; k < 4; k++) if (mask & (1u << k)) order[n++] = k;
        if (policy & 2) { int b = 0; for (int i = 1; i < n; i++) if (t1[order[i]] < t1[order[b]]) b = i; std::swap(order[0], order[b]); }
        else if (policy & 4) std::sort(order, order + n, [&](int a, int b) { const bool la = (w.child[a] & wide::LEAF_BIT) != 0, lb = (w.child[b] & wide::LEAF_BIT) != 0; if (la != lb) return la; return t1[a] < t1[b] || (t1[a] == t1[b] && a < b); });
        else std::sort(order, order + n, [&](int a, int b) { return t1[a] < t1[b] || (t1[a] == t1[b] && a < b); });
        for (int i = n - 1; i >= 1; i--) {
            if (sp >= stackCap) { r.flags |= 2; return r; }
            stk[sp] = w.child[order[i]]; stkT[sp] = t1[order[i]]; sp++;
        }
        if (sp > maxSp) maxSp = sp;
        cur = w.child[order[0]]; curT = t1[order[0]]; have = true;
    }
    c.spHist[maxSp < 63 ? maxSp : 63]++;
    if (r.h.tri != ~0u) {
        const float win = r.h.T * wide::WINDOW;
        if (second <= win) r.flags |= 4;
        if (leafT1 > win) r.flags |= 8;
        if (bestMarked) r.flags |= 64;
    }
    return r;
}

static uint32_t rng_state = 12345;
static inline uint32_t pcg(uint32_t& st) { st = st * 747796405u + 2891336453u; uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u; return (w >> 22u) ^ w; }
static inline float rnd(uint32_t& st) { return (float)pcg(st) * 2.3283064365386962890625e-10f; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: wide_sim scene.bin view [w h] [policy] [stackCap]\n"); return 1; }
    const std::string view = argv[2];
    const int Wd = argc > 3 ? atoi(argv[3]) : 480, Ht = argc > 4 ? atoi(argv[4]) : 270;
    const int policy = argc > 5 ? atoi(argv[5]) : 0, stackCap = argc > 6 ? atoi(argv[6]) : 255;
    Scene s;
    { FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; } int32_t hd[2]; if (fread(hd, 4, 2, f) != 2) return 1; s.nodes.resize(hd[0]); s.nTris = hd[1]; s.tv.resize(12 * (size_t)hd[1]);
      if (fread(s.nodes.data(), 32, hd[0], f) != (size_t)hd[0] || fread(s.tv.data(), 48, hd[1], f) != (size_t)hd[1]) return 1; fclose(f); }
    wide::HostBuild WB = wide::build_host(s.nodes.data(), (uint32_t)s.nodes.size(), s.tv.data());
    { uint64_t ch[5] = {0, 0, 0, 0, 0}, leaves = 0; for (auto& w : WB.nodes) { ch[w.exps >> 24]++; for (int k = 0; k < 4; k++) if (w.child[k] & wide::LEAF_BIT) leaves++; }
      printf("scene %s: %zu BVH2 nodes (%.1f MB), %d triangles -> %zu wide nodes (%.1f MB; %llu/%llu/%llu with 2/3/4 children), %llu leaf records (%.1f MB; triVerts %.1f MB)\n", argv[1], s.nodes.size(), s.nodes.size() * 32e-6, s.nTris,
             WB.nodes.size(), WB.nodes.size() * 64e-6, (unsigned long long)ch[2], (unsigned long long)ch[3], (unsigned long long)ch[4], (unsigned long long)leaves, WB.leafRecs.size() * 4e-6, s.nTris * 48e-6); }
    // camera
    V3 eye = {0, 0, 25}, fwd = {0, 0, -1}; float fovy = 102.0f;
    if (view == "interior") eye = {0, 0, 0};
    else if (view == "atrium") { eye = {-15.5f, 2.2f, 0.6f}; fwd = normalize(V3{1.0f, 0.12f, -0.05f}); fovy = 70.0f; }
    else if (view == "cornell") { eye = {0, 0.0f, 3.4f}; fovy = 40.0f; }
    else if (view.rfind("cam:", 0) == 0) { float v[7]; if (sscanf(view.c_str() + 4, "%f,%f,%f,%f,%f,%f,%f", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6) != 7) { fprintf(stderr, "cam:ex,ey,ez,dx,dy,dz,fovy\n"); return 1; } eye = {v[0], v[1], v[2]}; fwd = normalize(V3{v[3], v[4], v[5]}); fovy = v[6]; }
    V3 up = {0, 1, 0}, right = normalize(cross(fwd, up)); up = cross(right, fwd);
    const float th = tanf(0.5f * fovy * 3.14159265f / 180.0f), aspect = (float)Wd / Ht;
    std::vector<V3> ro((size_t)Wd * Ht), rd((size_t)Wd * Ht);
    for (int y = 0; y < Ht; y++) for (int x = 0; x < Wd; x++) {
        uint32_t st = (uint32_t)(y * Wd + x) * 9781u + 7u;
        float u = ((x + rnd(st)) / Wd * 2.0f - 1.0f) * th * aspect, v = ((y + rnd(st)) / Ht * 2.0f - 1.0f) * th;
        ro[(size_t)y * Wd + x] = eye; rd[(size_t)y * Wd + x] = normalize(fwd + right * u + up * v);
    }
    uint64_t totalMismatch = 0;
    for (int bounce = 0; bounce < 3; bounce++) {
        const size_t N = ro.size();
        if (!N) break;
        std::vector<Hit> href(N); std::vector<WideResult> hw(N);
        Cnt tot; uint64_t flagged[8] = {0, 0, 0, 0, 0, 0, 0, 0}, anyFlag = 0, mism = 0, entered = 0, mismFlagged = 0;
#pragma omp parallel
        {
            Cnt c; uint64_t fl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, af = 0, mm = 0, en = 0, mf = 0;
#pragma omp for schedule(dynamic, 256)
            for (size_t i = 0; i < N; i++) {
                const uint64_t p0 = c.pairs;
                href[i] = ref_trace(s, ro[i], rd[i], FLOAT_MAX, c);
                if (c.pairs > p0) en++;
                hw[i] = wide_trace(s, WB, ro[i], rd[i], FLOAT_MAX, policy, stackCap, c);
                const uint32_t f = hw[i].flags;
                for (int b = 0; b < 7; b++) if (f & (1u << b)) fl[b]++;
                const bool same = memcmp(&href[i], &hw[i].h, sizeof(Hit)) == 0;
                if (f) { af++; if (!same) mf++; }
                else if (!same) { mm++; if (mm < 4) fprintf(stderr, "MISMATCH ray %zu: ref T %.9g tri %u | wide T %.9g tri %u flags %u\n", i, href[i].T, href[i].tri, hw[i].h.T, hw[i].h.tri, f); }
            }
#pragma omp critical
            { tot.pairs += c.pairs; tot.tris += c.tris; tot.wnodes += c.wnodes; tot.wleaves += c.wleaves; tot.wleafPass += c.wleafPass; tot.wtris += c.wtris; tot.popSkips += c.popSkips; for (int b = 0; b < 64; b++) tot.spHist[b] += c.spHist[b];
              for (int b = 0; b < 7; b++) flagged[b] += fl[b]; anyFlag += af; mism += mm; entered += en; mismFlagged += mf; }
        }
        totalMismatch += mism;
        const double e = (double)std::max<uint64_t>(entered, 1);
        printf("%s bounce %d: %zu rays, %llu enter the BVH | BVH2: %.2f pair visits + %.2f triangle tests per entering ray | wide (policy %d): %.2f node visits + %.2f leaf records (%.2f pass their box) + %.2f triangle tests, %.2f popped entries skipped"
               " | round trips %.2f -> %.2f (x%.3f), bytes %.0f -> %.0f\n", view.c_str(), bounce, N, (unsigned long long)entered, tot.pairs / e, tot.tris / e, policy, tot.wnodes / e, tot.wleaves / e, tot.wleafPass / e, tot.wtris / e, tot.popSkips / e,
               (tot.pairs + tot.tris) / e, (tot.wnodes + tot.wleaves + (tot.wtris - tot.wleafPass)) / e, (double)(tot.wnodes + tot.wleaves + (tot.wtris - tot.wleafPass)) / (double)(tot.pairs + tot.tris),
               (tot.pairs * 64.0 + tot.tris * 48.0) / e, (tot.wnodes * 64.0 + tot.wleaves * 80.0 + (tot.wtris - tot.wleafPass) * 48.0) / e);
        printf("    flagged %llu (%.4f %%): inv %llu, overflow %llu, second candidate %llu, leaf t1 %llu, near miss %llu, marked best hit %llu; assumption violations %llu; flagged rays that WOULD have differed %llu; UNFLAGGED MISMATCHES %llu\n", (unsigned long long)anyFlag, 100.0 * anyFlag / (double)N,
               (unsigned long long)flagged[0], (unsigned long long)flagged[1], (unsigned long long)flagged[2], (unsigned long long)flagged[3], (unsigned long long)flagged[5], (unsigned long long)flagged[6], (unsigned long long)flagged[4], (unsigned long long)mismFlagged, (unsigned long long)mism);
        { printf("    wide stack depth needed (rays): "); uint64_t cum = 0, all = 0; for (int b = 0; b < 64; b++) all += tot.spHist[b]; for (int b = 0; b < 64; b++) { cum += tot.spHist[b]; if (tot.spHist[b] && (b % 2 == 0 || cum == all)) printf("<=%d: %.4f%%  ", b, 100.0 * cum / all); } printf("\n"); }
        // next bounce: cosine-distributed directions about the geometric normal (towards the incoming side)
        std::vector<V3> no, nd;
        for (size_t i = 0; i < N; i++) {
            if (href[i].tri == ~0u) continue;
            const float* v = &s.tv[12 * (size_t)href[i].tri];
            V3 p0 = {v[0], v[1], v[2]}, p1 = {v[4], v[5], v[6]}, p2 = {v[8], v[9], v[10]};
            V3 n = normalize(cross(p1 - p0, p2 - p0)); if (dot(n, rd[i]) > 0.0f) n = neg(n);
            V3 p = ro[i] + rd[i] * href[i].T + n * 0.001f;
            uint32_t st = (uint32_t)i * 2654435761u + (uint32_t)bounce;
            float c0 = rnd(st) * 2.0f - 1.0f, ph = rnd(st) * 6.2831853f, sn = sqrtf(fmaxf(0.0f, 1.0f - c0 * c0));
            V3 d = normalize(n + V3{sn * cosf(ph), sn * sinf(ph), c0});
            if (!(d.x == d.x)) continue;
            no.push_back(p); nd.push_back(d);
        }
        ro.swap(no); rd.swap(nd);
    }
    (void)rng_state;
    return totalMismatch ? 2 : 0;
}
