// packet_sim.cpp — CPU model of a wave-uniform PACKET walk of the BVH2 (one shared stack, the node pair fetched once per wave, every lane tests both boxes
// with its own ray and its own T) beside the reference's per-ray walk (BVHIntersect.glsl:27-105), for the rays one traversal wave of k_trace2 holds under
// k_gen_primary's pixel-major order: PX neighbouring pixels of a tile row x SMP samples (round 5: 4 x 16).  Developer tool, round 6: decides whether a
// packet kernel can beat the while-while kernel BEFORE one is written.  Scene files come from tools/dump_scene_for_sim.py.
//   g++ -O2 -std=c++17 -ffp-contract=off -fopenmp tools/packet_sim.cpp -o /tmp/packet_sim
//   /tmp/packet_sim scene.bin <headline|interior|atrium> [PX SMP] [width height]
// Reports, per 64 rays: the per-ray walk's pair visits and what the while-while kernel makes of them (wave steps at the measured 39 live lanes per step),
// the packet walk's node steps (union of the nodes any live ray wants), its live lanes per step, its leaf triangle rounds (one wave-wide test per triangle of a
// leaf any live ray enters), and how many rays would be flagged for the exact re-trace (their hit could depend on the visiting order: a second candidate within
// 2^-16 of T — the wide-node walk's rule).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>

struct Bvh2Node { float mn[3]; uint32_t startOrChild; float mx[3]; uint32_t triCount; };
struct V3 { float x, y, z; };
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static inline V3 normalize(V3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
#define FLOAT_MAX 3.4028235e+38f

static inline bool ray_tri(V3 ro, V3 rd, V3 p0, V3 p1, V3 p2, float* by, float* bz, float* t)
{
    V3 p1p0 = p1 - p0, p2p0 = p2 - p0, rop0 = ro - p0;
    V3 normal = cross(p1p0, p2p0), q = cross(rop0, rd);
    float invDet = 1.0f / dot(rd, normal);
    *t = dot(neg(normal), rop0) * invDet; *by = dot(neg(q), p2p0) * invDet; *bz = dot(q, p1p0) * invDet;
    float bx = 1.0f - *by - *bz;
    return bx >= 0.0f && *by >= 0.0f && *bz >= 0.0f && *t >= 0.0f;
}
static inline bool ray_box(V3 o, V3 inv, const float* mn, const float* mx, float* t1)
{
    float t0x = (mn[0] - o.x) * inv.x, t0y = (mn[1] - o.y) * inv.y, t0z = (mn[2] - o.z) * inv.z;
    float t1x = (mx[0] - o.x) * inv.x, t1y = (mx[1] - o.y) * inv.y, t1z = (mx[2] - o.z) * inv.z;
    float sx = fminf(t0x, t1x), sy = fminf(t0y, t1y), sz = fminf(t0z, t1z);
    float bx = fmaxf(t0x, t1x), by = fmaxf(t0y, t1y), bz = fmaxf(t0z, t1z);
    *t1 = fmaxf(sx, fmaxf(sy, fmaxf(sz, 0.0f)));
    float t2 = fminf(bx, fminf(by, bz));
    return *t1 <= t2;
}

struct Scene { std::vector<Bvh2Node> nodes; std::vector<float> tv; int nTris; };
struct Hit { float T, by, bz; uint32_t tri; };

static Hit ref_trace(const Scene& s, V3 ro, V3 rd, uint64_t& pairs, uint64_t& tris, bool& entered)
{
    Hit h = {FLOAT_MAX, 0, 0, ~0u};
    V3 inv = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    float tl, tr; entered = false;
    if (!(ray_box(ro, inv, s.nodes[1].mn, s.nodes[1].mx, &tl) && tl < h.T)) return h;
    entered = true;
    uint32_t stack[128]; int sp = 0; uint32_t top = 2;
    while (true) {
        pairs++;
        const Bvh2Node& L = s.nodes[top]; const Bvh2Node& R = s.nodes[top + 1];
        bool hitL = ray_box(ro, inv, L.mn, L.mx, &tl) && tl <= h.T;
        bool hitR = ray_box(ro, inv, R.mn, R.mx, &tr) && tr <= h.T;
        bool iL = hitL && L.triCount > 0, iR = hitR && R.triCount > 0;
        if (iL || iR) {
            uint32_t first = iL ? L.startOrChild : R.startOrChild;
            uint32_t end = !iR ? (L.startOrChild + L.triCount) : (R.startOrChild + R.triCount);
            for (uint32_t i = first; i < end; i++) {
                tris++;
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

struct PStat { uint64_t packets = 0, rays = 0, nodeSteps = 0, liveLanes = 0, leafVisits = 0, triRounds = 0, triLanes = 0, flagged = 0, mismatchUnflagged = 0, maxSp = 0, refPairs = 0, refTris = 0, pops = 0, popSkips = 0; };

// Packet walk.  Stack entries carry the mask of lanes that hit the pushed child's box (their own test at push time); a popped entry is visited by the lanes of that mask whose
// T still admits it... the entry distance is per lane, so the model keeps per-lane entry distances only implicitly: a popped node pair is re-tested by every lane of the mask (the test
// at the children is what culls).  A leaf child is entered by the lanes that hit it.
static void packet_trace(const Scene& s, const V3* ro, const V3* rd, int n, Hit* out, bool* flag, PStat& st)
{
    V3 inv[64]; float second[64];
    uint64_t live = 0;
    for (int i = 0; i < n; i++) {
        inv[i] = {1.0f / rd[i].x, 1.0f / rd[i].y, 1.0f / rd[i].z}; out[i] = {FLOAT_MAX, 0, 0, ~0u}; second[i] = FLOAT_MAX; flag[i] = false;
        float t; if (ray_box(ro[i], inv[i], s.nodes[1].mn, s.nodes[1].mx, &t) && t < FLOAT_MAX) live |= 1ull << i;
    }
    if (!live) return;
    struct E { uint32_t node; uint64_t mask; float t1[64]; } stack[64]; int sp = 0; const bool popCull = getenv("POP_CULL") != nullptr;
    uint32_t top = 2; uint64_t mask = live;
    while (true) {
        st.nodeSteps++; st.liveLanes += __builtin_popcountll(mask);
        const Bvh2Node& L = s.nodes[top]; const Bvh2Node& R = s.nodes[top + 1];
        uint64_t mL = 0, mR = 0; int votesL = 0, votesR = 0; float t1L[64], t1R[64];
        for (int i = 0; i < n; i++) if (mask >> i & 1) {
            float& tl = t1L[i]; float& tr = t1R[i];
            bool hl = ray_box(ro[i], inv[i], L.mn, L.mx, &tl) && tl <= out[i].T;
            bool hr = ray_box(ro[i], inv[i], R.mn, R.mx, &tr) && tr <= out[i].T;
            if (hl) mL |= 1ull << i; if (hr) mR |= 1ull << i;
            if (hl && hr) { if (tl < tr) votesL++; else votesR++; } else if (hl) votesL++; else if (hr) votesR++;
        }
        // leaves first (as the reference: the leaf children of this pair are tested before the walk moves on), left leaf then right leaf
        for (int side = 0; side < 2; side++) {
            const Bvh2Node& C = side ? R : L; const uint64_t m = side ? mR : mL;
            if (C.triCount == 0 || !m) continue;
            st.leafVisits++;
            for (uint32_t k = 0; k < C.triCount; k++) {
                st.triRounds++;
                // (lanes whose T shrank below their entry distance meanwhile still test: cheap and harmless)
                const uint32_t id = C.startOrChild + k; const float* v = &s.tv[12 * (size_t)id];
                for (int i = 0; i < n; i++) if (m >> i & 1) {
                    st.triLanes++;
                    float by, bz, t;
                    if (ray_tri(ro[i], rd[i], {v[0], v[1], v[2]}, {v[4], v[5], v[6]}, {v[8], v[9], v[10]}, &by, &bz, &t)) {
                        if (t < out[i].T) { if (out[i].tri != ~0u) second[i] = fminf(second[i], out[i].T); out[i] = {t, by, bz, id}; }
                        else if (id != out[i].tri) second[i] = fminf(second[i], t);
                    }
                }
            }
        }
        const uint64_t tL = L.triCount == 0 ? mL : 0, tR = R.triCount == 0 ? mR : 0;
        if (tL && tR) {
            const bool lc = votesL >= votesR;
            if (sp >= 63) { fprintf(stderr, "stack overflow\n"); exit(3); }
            stack[sp].node = lc ? R.startOrChild : L.startOrChild; stack[sp].mask = lc ? tR : tL; memcpy(stack[sp].t1, lc ? t1R : t1L, sizeof(t1L)); sp++;
            if ((uint64_t)sp > st.maxSp) st.maxSp = sp;
            top = lc ? L.startOrChild : R.startOrChild; mask = lc ? tL : tR;
        } else if (tL || tR) { top = tL ? L.startOrChild : R.startOrChild; mask = tL ? tL : tR; }
        else {
            bool got = false;
            while (sp > 0) {
                --sp; st.pops++; top = stack[sp].node; mask = stack[sp].mask;
                if (popCull) { for (int i = 0; i < n; i++) if ((mask >> i & 1) && !(stack[sp].t1[i] <= out[i].T)) mask &= ~(1ull << i); }
                if (mask) { got = true; break; }
                st.popSkips++;
            }
            if (!got) break;
        }
    }
    for (int i = 0; i < n; i++) if (out[i].tri != ~0u && second[i] <= out[i].T * (1.0f + 1.0f / 65536.0f)) flag[i] = true;
}

static inline uint32_t pcg(uint32_t& st) { st = st * 747796405u + 2891336453u; uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u; return (w >> 22u) ^ w; }
static inline float rnd(uint32_t& st) { return (float)pcg(st) * 2.3283064365386962890625e-10f; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: packet_sim scene.bin view [PX SMP] [w h]\n"); return 1; }
    const std::string view = argv[2];
    const int PX = argc > 3 ? atoi(argv[3]) : 4, SMP = argc > 4 ? atoi(argv[4]) : 16;
    const int Wd = argc > 5 ? atoi(argv[5]) : 1920, Ht = argc > 6 ? atoi(argv[6]) : 1080;
    const int PXY = argc > 7 ? atoi(argv[7]) : 1;      // pixels of a packet: PX wide x PXY high
    Scene s;
    { FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; } int32_t hd[2]; if (fread(hd, 4, 2, f) != 2) return 1; s.nodes.resize(hd[0]); s.nTris = hd[1]; s.tv.resize(12 * (size_t)hd[1]);
      if (fread(s.nodes.data(), 32, hd[0], f) != (size_t)hd[0] || fread(s.tv.data(), 48, hd[1], f) != (size_t)hd[1]) return 1; fclose(f); }
    V3 eye = {0, 0, 25}, fwd = {0, 0, -1}; float fovy = 102.0f;
    if (view == "interior") eye = {0, 0, 0};
    else if (view == "atrium") { eye = {-15.5f, 2.2f, 0.6f}; fwd = normalize(V3{1.0f, 0.12f, -0.05f}); fovy = 70.0f; }
    V3 up = {0, 1, 0}, right = normalize(cross(fwd, up)); up = cross(right, fwd);
    const float th = tanf(0.5f * fovy * 3.14159265f / 180.0f), aspect = (float)Wd / Ht;
    // every 4th packet row / column block of the frame (a uniform subsample keeps the run short)
    const int stepY = 8 * PXY, stepX = 8 * PX;
    PStat tot;
#pragma omp parallel
    {
        PStat st;
#pragma omp for schedule(dynamic, 4) collapse(2)
        for (int y0 = 0; y0 < Ht - PXY + 1; y0 += stepY) for (int x0 = 0; x0 < Wd - PX + 1; x0 += stepX) {
            V3 ro[64], rd[64]; Hit hp[64], hr[64]; bool fl[64]; int n = 0;
            for (int py = 0; py < PXY; py++) for (int px = 0; px < PX; px++) for (int sm = 0; sm < SMP && n < 64; sm++) {
                const int x = x0 + px, y = y0 + py;
                uint32_t rs = ((uint32_t)(y * Wd + x) * 9781u + 7u) ^ ((uint32_t)sm * 0x9E3779B9u);
                float u = ((x + rnd(rs)) / Wd * 2.0f - 1.0f) * th * aspect, v = ((y + rnd(rs)) / Ht * 2.0f - 1.0f) * th;
                ro[n] = eye; rd[n] = normalize(fwd + right * u + up * v); n++;
            }
            uint64_t entered = 0;
            for (int i = 0; i < n; i++) { bool en; hr[i] = ref_trace(s, ro[i], rd[i], st.refPairs, st.refTris, en); entered += en; }
            if (!entered) continue;
            st.packets++; st.rays += entered;
            packet_trace(s, ro, rd, n, hp, fl, st);
            for (int i = 0; i < n; i++) { if (fl[i]) st.flagged++; else if (memcmp(&hp[i], &hr[i], sizeof(Hit)) != 0) st.mismatchUnflagged++; }
        }
#pragma omp critical
        { tot.packets += st.packets; tot.rays += st.rays; tot.nodeSteps += st.nodeSteps; tot.liveLanes += st.liveLanes; tot.leafVisits += st.leafVisits; tot.triRounds += st.triRounds; tot.triLanes += st.triLanes;
          tot.flagged += st.flagged; tot.mismatchUnflagged += st.mismatchUnflagged; tot.maxSp = std::max(tot.maxSp, st.maxSp); tot.refPairs += st.refPairs; tot.refTris += st.refTris; tot.pops += st.pops; tot.popSkips += st.popSkips; }
    }
    const double P = (double)std::max<uint64_t>(tot.packets, 1), R = (double)std::max<uint64_t>(tot.rays, 1);
    printf("%s, packets of %d x %d pixels x %d samples: %llu packets with %.1f entering rays each\n", view.c_str(), PX, PXY, SMP, (unsigned long long)tot.packets, R / P);
    printf("  per-ray walk : %.1f pair visits + %.2f triangle tests per ray  -> per 64 rays: %.0f lane-steps = %.0f wave steps at 39 live lanes, %.0f lane triangle tests\n", tot.refPairs / R, tot.refTris / R, 64.0 * tot.refPairs / R, 64.0 * tot.refPairs / R / 39.0, 64.0 * tot.refTris / R);
    printf("  packet walk  : %.1f node steps per packet (%.1f live lanes per step: %.2f of the entering rays), %.1f leaf visits, %.1f wave-wide triangle rounds (%.1f lanes each), stack depth <= %llu\n", tot.nodeSteps / P, (double)tot.liveLanes / tot.nodeSteps, (double)tot.liveLanes / tot.nodeSteps / (R / P),
           tot.leafVisits / P, tot.triRounds / P, (double)tot.triLanes / std::max<uint64_t>(tot.triRounds, 1), (unsigned long long)tot.maxSp);
    printf("  normalised to 64 entering rays: packet node steps %.0f vs while-while wave steps %.0f (x%.2f); triangle rounds %.0f\n", tot.nodeSteps / R * 64.0, 64.0 * tot.refPairs / R / 39.0, (tot.nodeSteps / R * 64.0) / (64.0 * tot.refPairs / R / 39.0), tot.triRounds / R * 64.0);
    printf("  pops %.1f per packet, %.1f of them skipped whole (POP_CULL)\n", tot.pops / P, tot.popSkips / P);
    printf("  flagged for the exact re-trace %.4f %% of the rays; unflagged rays whose hit differs from the per-ray walk: %llu\n", 100.0 * tot.flagged / R, (unsigned long long)tot.mismatchUnflagged);
    return 0;
}
