// sim_layout.cpp — offline model of what a node-pair permutation (idkengine_amd/csrc/node_layout.hpp) does to the L2 of one XCD.
// Not part of the product: a developer tool that ranks layouts before GPU time is spent on them (the GPU's PMC counters decide).
//
//   g++ -O2 -std=c++17 -o /tmp/sim_layout tools/sim_layout.cpp && /tmp/sim_layout scene.bin [raysInFlight] [totalRays] [kind]
//
// scene.bin (tools/dump_scene_for_sim.py): int32 nodeCount, int32 triCount, nodes[nodeCount] (32 B), triVerts[triCount] (3 x float4).
// Model: `raysInFlight` rays advance round-robin, one node step (one 64-B pair fetch) or one leaf (its 48-B triangle records) per turn —
// the interleaving k_trace2's persistent waves produce on one XCD (32 CUs x 24 waves x 64 lanes = 49 k rays); every fetch goes through a
// 4 MiB, 16-way, 128-B-line LRU cache (the XCD's L2; L1 is ignored, which makes every layout look worse by the same hits).  Rays: kind 0 =
// diffuse bounce rays (origin on a random triangle, cosine-free random direction), kind 1 = primary rays of a camera inside the scene,
// kind 2 = bounce rays handed out in the order of their origin triangle (BLAS leaf order: a spatially sorted queue).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include "../idkengine_amd/csrc/node_layout.hpp"

using nodelayout::Node;
struct TV { float a[4], b[4], c[4]; };

struct Cache {
    static const int WAYS = 16; int sets; std::vector<uint64_t> tag; std::vector<uint32_t> age; uint32_t clock = 1; uint64_t hits = 0, misses = 0; int lineShift;
    Cache(size_t bytes, int lineBytes) { lineShift = 0; while ((1 << lineShift) < lineBytes) lineShift++; sets = (int)(bytes / lineBytes / WAYS); tag.assign((size_t)sets * WAYS, ~0ull); age.assign((size_t)sets * WAYS, 0); }
    bool access(uint64_t addr) {
        const uint64_t line = addr >> lineShift; const size_t s = (size_t)((line * 0x9E3779B97F4A7C15ull) >> 40) % sets * WAYS;
        int victim = 0; uint32_t oldest = ~0u;
        for (int w = 0; w < WAYS; w++) { if (tag[s + w] == line) { age[s + w] = clock++; hits++; return true; } if (age[s + w] < oldest) { oldest = age[s + w]; victim = w; } }
        tag[s + victim] = line; age[s + victim] = clock++; misses++; return false;
    }
};

struct Ray { float o[3], d[3], inv[3], T; uint32_t stack[64]; int sp; uint32_t top; uint32_t leafFirst, leafEnd; bool leafPending, active; };

static bool box(const Ray& r, const Node& n, float T, float* tmin)
{
    float t0 = 0.0f, t1 = T;
    for (int k = 0; k < 3; k++) { float a = (n.mn[k] - r.o[k]) * r.inv[k], b = (n.mx[k] - r.o[k]) * r.inv[k]; float lo = fminf(a, b), hi = fmaxf(a, b); t0 = fmaxf(t0, lo); t1 = fminf(t1, hi); }
    *tmin = t0; return t0 <= t1;
}
static bool tri(const Ray& r, const TV& t, float* out)
{
    float e1[3], e2[3], p[3], s[3], q[3];
    for (int k = 0; k < 3; k++) { e1[k] = t.b[k] - t.a[k]; e2[k] = t.c[k] - t.a[k]; }
    p[0] = r.d[1] * e2[2] - r.d[2] * e2[1]; p[1] = r.d[2] * e2[0] - r.d[0] * e2[2]; p[2] = r.d[0] * e2[1] - r.d[1] * e2[0];
    const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2]; if (fabsf(det) < 1e-12f) return false;
    const float id = 1.0f / det; for (int k = 0; k < 3; k++) s[k] = r.o[k] - t.a[k];
    const float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * id; if (u < 0.0f || u > 1.0f) return false;
    q[0] = s[1] * e1[2] - s[2] * e1[1]; q[1] = s[2] * e1[0] - s[0] * e1[2]; q[2] = s[0] * e1[1] - s[1] * e1[0];
    const float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * id; if (v < 0.0f || u + v > 1.0f) return false;
    const float tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * id; if (tt <= 1e-4f) return false;
    *out = tt; return true;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: sim_layout scene.bin [raysInFlight=49152] [totalRays=400000] [kind=0]\n"); return 2; }
    const int inFlight = argc > 2 ? atoi(argv[2]) : 49152; const long total = argc > 3 ? atol(argv[3]) : 400000; const int kind = argc > 4 ? atoi(argv[4]) : 0;
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
    int32_t nc = 0, tc = 0; if (fread(&nc, 4, 1, f) != 1 || fread(&tc, 4, 1, f) != 1) return 1;
    std::vector<Node> nodes((size_t)nc); std::vector<TV> tv((size_t)tc);
    if (fread(nodes.data(), 32, nc, f) != (size_t)nc || fread(tv.data(), 48, tc, f) != (size_t)tc) return 1;
    fclose(f);
    printf("scene: %d nodes, %d triangles; %d rays in flight, %ld rays, kind %d\n", nc, tc, inFlight, total, kind);
    const Node& root = nodes[1];
    struct Cfg { const char* name; int mode, depth; };
    const Cfg cfgs[] = {{"reference order", 0, 0}, {"couples + depth-first", 1, 0}};
    for (const Cfg& cfg : cfgs) {
        std::vector<uint32_t> slot;
        nodelayout::compute(nodes.data(), nc, 0, cfg.mode, cfg.depth, slot);
        // sanity: permutation
        { std::vector<char> seen(slot.size(), 0); for (uint32_t s : slot) { if (s >= slot.size() || seen[s]) { fprintf(stderr, "not a permutation\n"); return 1; } seen[s] = 1; } }
        // static figure: weight of parent->child edges inside one 128-B line
        long sameLineEdges = 0, edges = 0;
        for (int k = 1; k < nc / 2; k++) for (int i = 0; i < 2; i++) { const Node& nd = nodes[2 * k + i]; if (nodelayout::internal(nd)) { edges++; if ((slot[k] >> 1) == (slot[nd.startOrChild / 2] >> 1)) sameLineEdges++; } }
        Cache l2((size_t)4 << 20, 128);
        const uint64_t triBase = (uint64_t)1 << 32;
        std::mt19937 rng(12345);
        std::uniform_real_distribution<float> U(0.0f, 1.0f);
        std::vector<Ray> rays((size_t)inFlight);
        long launched = 0, retired = 0; uint64_t steps = 0, sameLineAsPrev = 0, nodeHits = 0, nodeAcc = 0;
        std::vector<uint32_t> prevLine((size_t)inFlight, ~0u);
        auto spawn = [&](Ray& r) {
            float d[3]; float l;
            do { d[0] = 2 * U(rng) - 1; d[1] = 2 * U(rng) - 1; d[2] = 2 * U(rng) - 1; l = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]; } while (l > 1.0f || l < 1e-4f);
            l = 1.0f / sqrtf(l);
            if (kind == 0 || kind == 2) { const TV& t = tv[kind == 2 ? (size_t)((double)launched / (double)total * tc) % tc : rng() % tc]; for (int k = 0; k < 3; k++) r.o[k] = (t.a[k] + t.b[k] + t.c[k]) / 3.0f + d[k] * l * 1e-3f; }
            else { for (int k = 0; k < 3; k++) r.o[k] = 0.5f * (root.mn[k] + root.mx[k]); long i = launched; float x = (float)(i % 1920) / 1920.0f - 0.5f, y = (float)((i / 1920) % 1080) / 1080.0f - 0.5f; d[0] = x * 2.4f; d[1] = y * 1.35f; d[2] = -1.0f; l = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + 1.0f); }
            for (int k = 0; k < 3; k++) { r.d[k] = d[k] * l; r.inv[k] = 1.0f / r.d[k]; }
            r.T = 3.4e38f; r.sp = 0; r.top = 2; r.leafPending = false; r.active = true; launched++;
        };
        for (int i = 0; i < inFlight && launched < total; i++) spawn(rays[i]);
        while (retired < launched) {
            for (int i = 0; i < inFlight; i++) {
                Ray& r = rays[i]; if (!r.active) continue;
                if (r.leafPending) {
                    for (uint32_t t = r.leafFirst; t < r.leafEnd; t++) { l2.access(triBase + (uint64_t)t * 48); l2.access(triBase + (uint64_t)t * 48 + 47); float tt; if (tri(r, tv[t], &tt) && tt < r.T) r.T = tt; }
                    r.leafPending = false;
                } else if (r.top != 0) {
                    const uint32_t pairIdx = r.top / 2, s = slot[pairIdx];
                    const bool h = l2.access((uint64_t)s * 64); nodeAcc++; if (h) nodeHits++;
                    steps++; if ((s >> 1) == prevLine[i]) sameLineAsPrev++; prevLine[i] = s >> 1;
                    const Node& L = nodes[r.top]; const Node& R = nodes[r.top + 1];
                    float tl, tr; const bool hl = box(r, L, r.T, &tl) && tl <= r.T, hr = box(r, R, r.T, &tr) && tr <= r.T;
                    const bool il = hl && L.count > 0, ir = hr && R.count > 0;
                    if (il || ir) { r.leafFirst = il ? L.startOrChild : R.startOrChild; r.leafEnd = !ir ? L.startOrChild + L.count : R.startOrChild + R.count; r.leafPending = true; }
                    const bool tL = hl && L.count == 0, tR = hr && R.count == 0;
                    if (tL || tR) { if (tL && tR) { const bool lc = tl < tr; r.top = lc ? L.startOrChild : R.startOrChild; if (r.sp < 64) r.stack[r.sp++] = lc ? R.startOrChild : L.startOrChild; } else r.top = tL ? L.startOrChild : R.startOrChild; }
                    else r.top = r.sp ? r.stack[--r.sp] : 0;
                }
                if (!r.leafPending && r.top == 0) { retired++; r.active = false; prevLine[i] = ~0u; if (launched < total) spawn(r); }
            }
        }
        printf("%-26s static same-line edges %5.1f %% | node steps/ray %6.1f | next fetch in the previous line %5.1f %% | L2 model: node-fetch hit %5.1f %%, all %5.1f %%, misses/ray %6.2f\n",
               cfg.name, 100.0 * sameLineEdges / (double)edges, (double)steps / retired, 100.0 * sameLineAsPrev / (double)steps, 100.0 * nodeHits / (double)nodeAcc, 100.0 * l2.hits / (double)(l2.hits + l2.misses), (double)l2.misses / retired);
    }
    return 0;
}
