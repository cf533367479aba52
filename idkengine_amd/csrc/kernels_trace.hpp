// kernels_trace.hpp — traversal kernels: thread-per-ray general kernels (k_trace_primary / k_trace_queue), coherent primary-ray generation with pre-cull (k_gen_primary) and the persistent while-while kernel k_trace2 (BVHIntersect.glsl:27-105, 183-291).
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

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
            gen_primary(f, 0u, pix, f.accumulated, origin, pd, seed);
            rays.o_ior[pix] = make_float4(origin.x, origin.y, origin.z, 1.0f);
            rays.thr_px[pix] = make_float4(1.0f, 1.0f, 1.0f, pd.x);
            rays.rad_py[pix] = make_float4(0.0f, 0.0f, 0.0f, pd.y);
            f3 rd = DecodeUnitVec(pd.x, pd.y);
            HitRec hit; float cost;
            TraceRay<COUNT, COST>(s, f, origin, rd, hit, cost, stk, WAVE, nPairs, nTris);
            store_hit(hits, pix, hit.T, hit.bx, hit.by, hit.tri, hit.xform);
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
            store_hit(hits, slot, hit.T, hit.bx, hit.by, hit.tri, hit.xform);
        }
    }
    if (COUNT) flush_counters(counters, nPairs, nTris);
}


// k_classify_tiles: once per batch, one thread per 8x8 tile (sample-independent).  A tile whose whole beam of possible primary rays
// — every jitter offset, every point of the lens — provably misses the root box, and provably looks at one face of a constant-per-face
// sky, needs no per-pixel ray generation at all: its pixels are the FirstHit miss branch with a known colour (FirstHit:225-233).
// The test is CONSERVATIVE (box strictly outside one side plane of the tile's pyramid, by a margin that covers the lens radius, the
// direction tilt LenseRadius/FocalLength, one extra pixel of jitter and rounding); tiles that fail it take the exact per-pixel path,
// so results are bit-identical either way.  class 0 = per-pixel path, 1..6 = miss + sky face (class-1), 7 = miss + no sky (black),
// 8 = miss + a textured sky (an HDR cube map, the engine's default: SkyBoxManager.cs:44,74): k_final_draw generates the pixel's ray and samples the sky itself
// (the arithmetic of k_gen_primary's miss branch), so nothing is stored per sample for those pixels either.
template <bool VER>
__global__ __launch_bounds__(256) void k_classify_tiles(DScene s0, Frame f, uint8_t* tileClass, uint32_t tilesX, uint32_t tilesY)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tilesX * tilesY) return;
    const uint32_t smp = blockIdx.y;                                 // > 0 only with per-sample cameras (frame ring) or scene versions: one classification per sample
    const DScene s = VER ? scene_of_sample(s0, smp) : s0;
    const float* cam = f.cams ? f.cams + 36u * smp : f.invProj;
    const float* invProj = cam; const float* invView = cam + 16; const float* vp = cam + 32;
    const uint32_t tx = t % tilesX, ty = t / tilesX;
    uint8_t cls = 0;
    const float r = f.g.LenseRadius, F = f.g.FocalLength;
    if (s.instanceCount >= 1 && s.instanceCount <= 256 && !f.outputAovs && r >= 0.0f && F > 1e-3f && r / F <= 0.05f) {
        const float W = (float)f.W, H = (float)f.H;
        const int gy0 = global_row(f, (int)(ty * 8)), gy1 = global_row(f, (int)(ty * 8 + 7));   // global rows of the tile's first / last local row (monotone in the local row)
        const float nx0 = ((float)(tx * 8) - 1.0f) / W * 2.0f - 1.0f, nx1 = ((float)(tx * 8) + 9.0f) / W * 2.0f - 1.0f;   // one pixel of slack on every side
        const float ny0 = ((float)gy0 - 1.0f) / H * 2.0f - 1.0f, ny1 = ((float)gy1 + 2.0f) / H * 2.0f - 1.0f;
        const f3 u[4] = {GetWorldSpaceDirection(invProj, invView, nx0, ny0), GetWorldSpaceDirection(invProj, invView, nx1, ny0),
                         GetWorldSpaceDirection(invProj, invView, nx1, ny1), GetWorldSpaceDirection(invProj, invView, nx0, ny1)};
        const f3 mid = (u[0] + u[1]) + (u[2] + u[3]);
        f3 pn[4]; bool planeOk[4];                                  // outward unit normals of the four side planes of the tile's pyramid
        for (int i = 0; i < 4; i++) {
            f3 n = cross(u[i], u[(i + 1) & 3]);
            const float len = gsqrt(dot(n, n));
            planeOk[i] = len > 1e-12f;                              // a degenerate side (cannot happen for a real tile) gives no decision
            n = n * (1.0f / (planeOk[i] ? len : 1.0f));
            if (dot(n, mid) > 0.0f) n = n * -1.0f;
            pn[i] = n;
        }
        const f3 C = mk3(vp[0], vp[1], vp[2]);
        // The boxes the traversal itself tests first (so that skipped rays would not have visited — or counted — anything):
        //   no TLAS: every instance's BLAS root box, an oriented box in world space (Model rows of its GpuMeshTransform; BVHIntersect.glsl:32-39)
        //   USE_TLAS: the two children of the TLAS root, world-space AABBs (:242-249); a leaf root is entered unconditionally -> no shortcut
        // Each must lie beyond one side plane by more than the margin.
        int nBoxes = s.instanceCount;
        uint32_t tlasChild = 0;
        const bool viaTlas = f.useTlas || f.instTlas;        // (instTlas: s.tlas is the library's own, padded tree over the instances — kernels_trace_inst.hpp: beyond its boxes no instance can be hit)
        if (viaTlas) {
            nBoxes = 0;
            if (s.tlasCount > 0) { const uint32_t packed = __float_as_uint(s.tlas[0].w); if ((packed >> 31) == 0u) { nBoxes = 2; tlasChild = packed & 0x7fffffffu; } }
        }
        bool outside = nBoxes > 0;
        for (int ii = 0; ii < nBoxes && outside; ii++) {
            float4 bmin, bmax, m0 = make_float4(1.0f, 0.0f, 0.0f, 0.0f), m1 = make_float4(0.0f, 1.0f, 0.0f, 0.0f), m2 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
            if (viaTlas) { bmin = s.tlas[2 * (size_t)(tlasChild + ii)]; bmax = s.tlas[2 * (size_t)(tlasChild + ii) + 1]; }
            else {
                const GpuBlasInstance inst = s.instances[ii];
                const float4* root = s.nodes + 2 * (size_t)s.descs[inst.BlasId].NodeOffset + 2;
                bmin = root[0]; bmax = root[1];
                const float4* x = s.xforms + 9 * (size_t)inst.MeshTransformId;
                m0 = x[0]; m1 = x[1]; m2 = x[2];
            }
            f3 rel[8]; float D = 0.0f;
            for (int c = 0; c < 8; c++) {
                const float cx = (c & 1) ? bmax.x : bmin.x, cy = (c & 2) ? bmax.y : bmin.y, cz = (c & 4) ? bmax.z : bmin.z;
                rel[c] = mk3(m0.x * cx + m0.y * cy + m0.z * cz + m0.w, m1.x * cx + m1.y * cy + m1.z * cz + m1.w, m2.x * cx + m2.y * cy + m2.z * cz + m2.w) - C;
                D = gmax(D, gsqrt(dot(rel[c], rel[c])));
            }
            const float margin = r + (D + r) * (r / F) * 1.5f + 1e-4f * (D + 1.0f);
            bool boxOutside = false;
            for (int i = 0; i < 4; i++) {
                if (!planeOk[i]) continue;
                float dmin = PT_FLOAT_MAX;
                for (int c = 0; c < 8; c++) dmin = gmin(dmin, dot(pn[i], rel[c]));
                if (dmin > margin) boxOutside = true;
            }
            outside = boxOutside;
        }
        if (outside) {
            if (s.skySize <= 0) cls = 7;
            else if (s.skySize > 1) cls = 8;
            else {
                // one sky face for every direction of the beam: a strictly dominant axis, same sign, at all four corners, by more than
                // twice the possible tilt (lens + oct-encoding round trip)
                const float eta = 2.0f * (r / F) + 1e-4f;
                int face = -1; bool same = true;
                for (int i = 0; i < 4; i++) {
                    const float ax = gabs(u[i].x), ay = gabs(u[i].y), az = gabs(u[i].z);
                    int fc = -1;
                    if (ax >= gmax(ay, az) + eta) fc = u[i].x > 0.0f ? 0 : 1;
                    else if (ay >= gmax(ax, az) + eta) fc = u[i].y > 0.0f ? 2 : 3;
                    else if (az >= gmax(ax, ay) + eta) fc = u[i].z > 0.0f ? 4 : 5;
                    if (fc < 0 || (face >= 0 && fc != face)) same = false;
                    face = fc;
                }
                if (same && face >= 0) cls = (uint8_t)(1 + face);
            }
        }
    }
    tileClass[(size_t)smp * (tilesX * tilesY) + t] = cls;
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path (single BLAS instance, no TLAS): coherent ray generation + persistent "while-while" traversal.
//
// k_gen_primary: one thread per pixel, 8x8 pixel tiles per wave.  Generates the primary ray (FirstHit:44-77), stores it,
// and pre-culls rays whose root-box test (BVHIntersect.glsl:32-39 with T = FLOAT_MAX) fails: those get their miss
// record written here and never reach the traversal kernel.  Survivors are appended (wave ballot + one atomic per
// wave) to an unordered active list; results are stored per pixel, so the list order is free.
template <bool VER>
__global__ __launch_bounds__(1024) void k_gen_primary(DScene s0, Frame f, RayBufs rays, TraceBufs tr, int cull, uint32_t* activeList, uint32_t* activeCount, uint32_t* seedOut, uint8_t* contFlag,
                                                     const uint8_t* tileClass /* null: no tile pre-classification */,
                                                     int lean /* the traversal reads only the trace-ready record: k_shade_first regenerates the state of a surviving ray instead of reading it back */)
{
    __shared__ uint32_t waveKeep[16]; __shared__ uint32_t blockBase; __shared__ unsigned long long keepMask[16];
    // grid = (samples, tile groups): the samples of one tile group are dispatched back to back, so the active list keeps
    // rays of the same screen region (all samples) together -> coherent waves in the traversal kernel.
    // f.genPixelMajor (batches of >= 8 samples): grid = (groups of G <= 16 samples, tiles), G waves per workgroup — a workgroup is ONE tile under G samples (the batch split into equal
    // groups: 32 samples = 2 x 16, the driver's 20 = 2 x 10), and it appends its survivors pixel by pixel (all samples of a pixel side by side): a wave of the traversal kernel then
    // holds 64 / G pixels x G samples, rays that differ by their sub-pixel jitter only
    const bool pm = f.genPixelMajor != 0;
    const uint32_t G = blockDim.x >> 6;                                 // pixel-major: samples (= waves) of this workgroup
    const uint32_t smp = pm ? blockIdx.x * G + (threadIdx.x >> 6) : blockIdx.x;   // sample of the batch
    const DScene s = VER ? scene_of_sample(s0, min(smp, (uint32_t)f.batch - 1u)) : s0;   // (wave-uniform)
    const uint32_t wave = pm ? blockIdx.y : (blockIdx.y * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t tilesX = ((uint32_t)f.W + 7) / 8;
    const uint32_t tx = wave % tilesX, ty = wave / tilesX;
    // (the 4 pixels a traversal wave gets are 4 neighbours of a tile row; walking the tile in Morton order instead — 2x2 blocks — measured the same in the traversal and cost the
    // streaming kernels their coalescing: atrium -4 %, profiles/r05_pixel_major.md)
    const uint32_t x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    const bool valid = x < (uint32_t)f.W && y < (uint32_t)f.rows && smp < (uint32_t)f.batch;
    const uint32_t pix = y * (uint32_t)f.W + x;
    const uint32_t rid = smp * f.Npad + pix;                           // ray id inside the batch
    bool keep = false;
    const uint32_t nTilesAll = tilesX * (((uint32_t)f.rows + 7) / 8);
    const uint32_t cls = (tileClass && wave < nTilesAll) ? tileClass[(f.tilePerSample ? (size_t)min(smp, (uint32_t)f.batch - 1u) * nTilesAll : 0) + wave] : 0u;   // wave-uniform
    if (valid && cls != 0u) {
        // the whole tile is a proven miss with a known sky colour (k_classify_tiles): FirstHit's miss branch without generating the ray
        // k_final_draw takes the colour from the tile class, idkptDownloadRays regenerates the ray state: one flag byte is all that is stored
        contFlag[rid] = 4;                                               // (bit 0 = "continues" must stay clear) origin / direction planes regenerated on demand (k_regen_culled)
    } else if (valid) {
        f3 origin; f2 pd; uint32_t seed;
        gen_primary(f, smp, pix, sample_index(f, smp), origin, pd, seed);
        f3 rd = DecodeUnitVec(pd.x, pd.y);
        f3 lo = origin, ld = rd, invDir = splat3(0.0f);   // several instances / TLAS: the traversal kernel transforms the world ray per instance
        float rootT = __builtin_inff();                   // single instance: tMin of the root-box test (+inf = miss), consumed by k_trace2
        keep = !cull;
        if (f.useTlas || f.instTlas) {
            // first TLAS step (BVHIntersect.glsl:242-249) with T = FLOAT_MAX: a ray that misses both children of the root is a miss
            // (instTlas — the instance loop walked through the library's own tree, kernels_trace_inst.hpp: its padded boxes hold every instance, so the same two tests stand for the loop's n root tests)
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
            if (cull || s.instanceCount == 1) {
                const float4* root = s.nodes + 2 * (size_t)s.descs[inst.BlasId].NodeOffset + 2;
                float t1;
                const bool boxHit = RayBoxIntersect(l0, iv, root[0], root[1], &t1);
                if (s.instanceCount == 1 && boxHit) rootT = t1;
                if (cull && boxHit && t1 < PT_FLOAT_MAX) keep = true;
            }
        }
        f3 radiance = splat3(0.0f);
        if (keep) {
            tr.rec[4 * (size_t)rid] = make_float4(lo.x, lo.y, lo.z, rootT); tr.rec[4 * (size_t)rid + 1] = make_float4(ld.x, ld.y, ld.z, 0.0f); tr.rec[4 * (size_t)rid + 2] = make_float4(invDir.x, invDir.y, invDir.z, 0.0f);
            if (!lean) seedOut[rid] = seed;                             // RNG state after ray generation, consumed by k_shade_first
        } else {
            // miss branch of FirstHit TraceRay (FirstHit/compute.glsl:225-233), evaluated right here
            f3 albedo = SampleSky(s, rd);
            radiance = radiance + albedo * splat3(1.0f);
            if (f.outputAovs) { f3 fn = CubemapFaceNormal(rd); rays.aovA[rid] = make_float4(albedo.x, albedo.y, albedo.z, 0.0f); rays.aovN[rid] = make_float4(fn.x, fn.y, fn.z, 0.0f); }
        }
        // A culled pixel's ray is finished: FinalDraw only needs its radiance.  Origin/throughput planes (32 of the 48 B) are not
        // written; the flag lets idkptDownloadRays regenerate them on demand (k_regen_culled).
        // A surviving ray's planes are a function of (pixel, sample): in lean mode (52 of its 105 B) k_shade_first recomputes them.
        if (keep && !lean) { rays.o_ior[rid] = make_float4(origin.x, origin.y, origin.z, 1.0f); rays.thr_px[rid] = make_float4(1.0f, 1.0f, 1.0f, pd.x); }
        contFlag[rid] = keep ? 0 : 2;       // also resets the continue flag of this ray id (k_shade_first sets 1); pad ids stay 0 from allocation
        if (!(keep && lean)) rays.rad_py[rid] = make_float4(radiance.x, radiance.y, radiance.z, pd.y);
    }
    // append the survivors: one atomic per 16-wave workgroup (a single counter word saturates at ~88 atomics/us; one atomic per 32 / 64 / 128 waves was
    // measured in round 3 and changes nothing: this kernel is not bound by its counter, profiles/r03_trace_experiments.md)
    const unsigned long long m = __ballot(keep);
    const uint32_t wv = threadIdx.x >> 6;
    if (pm) {
        if (cls != 0u) return;                               // (a tile classified as sky — pixel-major batches share one classification: workgroup-uniform, nothing to append)
        if (lane == 0) keepMask[wv] = m;
        __syncthreads();
        uint32_t cnt = 0, before = 0;                       // samples of this workgroup that keep pixel `lane`; those of them in waves before this one
        for (uint32_t w = 0; w < G; w++) { const uint32_t b = (uint32_t)(keepMask[w] >> lane) & 1u; cnt += b; before += w < wv ? b : 0u; }
        uint32_t incl = cnt;                                // inclusive scan over the 64 pixels
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off); incl += lane >= (uint32_t)off ? v : 0u; }
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
        if (threadIdx.x == 0) blockBase = tot ? atomicAdd(activeCount, tot) : 0u;
        __syncthreads();
        if (keep) activeList[blockBase + (incl - cnt) + before] = rid;
        return;
    }
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
// The node step is branch-free: the entry a pop would return is read from the LDS stack together with the node pair, the far child is
// stored unconditionally into the row above the stack top (it only joins the stack when sp moves), and every decision of
// BVHIntersect.glsl:81-101 is a select — no nested exec-mask regions (15 branches and 20 instructions fewer per step than the
// if/else form; +2.5 to +6 % on every view).  The stack pointer is the LDS address itself (pop = [sp], push = [sp + one row]: no index arithmetic).
#define GRAB_SLICES 8u          // work-list counters of k_trace2 (power of two)
#define GRAB_STRIDE 256u        // words between them (1 KB: separate cache lines and memory channels).  Word j of a line: the launch of bounce j; word 64 + j: the exact re-trace
                                // launch behind a wide-node launch (kernels_wide.hpp); words 128 + j of line 0: that launch's count of flagged rays
// VER: scene versions (DScene::ver): every ray traverses the geometry its sample was queued with; a lane keeps where that version's node pairs / triangle
// records (MULTI: also its TLAS and transforms) start, in 16-byte units, and adds it to every fetch.  Hit records stay version-independent.
// ANY: TraceRayAny (BVHIntersect.glsl:107-181, 299-411) for idkptTraceRays' any-hit queries: inside a BLAS the left child is visited first whatever the distances, the first
// triangle of a leaf with t < T ends the ray (its remaining triangles, its stack, its remaining instances / TLAS nodes are dropped); T is the query's maxDist until then.
// FAST (MODE 0, closest hit, one scene version, no counters — the launches the bench line prices): the node step on DScene::pairNodes, the same pairs with their fields regrouped so
// that BOTH boxes' slabs are six 2-wide subtractions and six 2-wide multiplications (the reference's layout allows eight + eight scalar ones); leaf flags derived once; no stack-full
// test (the stack is sized from the tree's validated need: k_trace2's own comment calls the test unreachable) and no stack-empty test (the row a pop of the empty stack reads holds 0,
// and the lane is finished then): 62 -> 53 vector instructions per step, every number the same operation on the same operands — bit-identical (tests/test_gpu_parity.py & co.).
template <bool PRIMARY, bool COUNT, int REFILL_MIN = 32, int OCC = 1, bool PROF = false, int LEAF_MIN = 24, int MODE = 0, int DBG = 0, bool VER = false, bool ANY = false, bool FAST = false>
__global__ __launch_bounds__(WAVE, OCC) void k_trace2(DScene s, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* list, const uint32_t* countPtr, uint32_t* workCounter, uint64_t* counters)
{
    static_assert(MODE >= 0 && MODE <= 2 && (DBG == 0 || DBG == 16), "k_trace2: MODE 0-2, DBG 0 or 16 (pooled leaf phase)");
    static_assert(!FAST || (MODE == 0 && !COUNT && !VER && !ANY && !PROF), "k_trace2: FAST is the plain one-BLAS closest-hit walk");
    constexpr bool MULTI = MODE != 0, TLAS = MODE == 2;
    // DBG 16 ("pooled leaves", MODE 0): the leaf phase tests the wave's (ray, triangle) PAIRS with all 64 lanes in one round trip instead of every parked lane
    // walking its own 1-8 triangles one dependent fetch after the other with 18-22 lanes active (instrumented: 23-48 pairs per leaf phase in 2.0-3.1 loop trips).
    // Pairs are numbered by a ballot prefix sum over the parked lanes' triangle counts; the owners write (lane, k) for their pairs into the 64 words of the stack's
    // dummy row, lane j picks up pair j, fetches its owner's ray through ds_bpermute and tests the triangle; the owners then collect their pairs' results IN ORDER with
    // the reference's `t < T` (BVHIntersect.glsl:57-79) — the same tests on the same operands, the same sequence of T updates: bit-identical hits.
    constexpr bool POOL = DBG == 16;     // (the host selects it for MODE 0 only: in the instance-loop / TLAS kernels it measured neutral to slightly negative, profiles/r04_leaf_pool.md)
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    // LDS rows of this wave, one word per lane: row 0 = dummy (what a pop of the empty stack reads), rows 1 .. cap = stack entries 0 .. cap-1,
    // row cap + 1 = spare (takes the store of a full stack), then the TLAS rows.  The stack pointer IS an LDS address (stkBase + sp rows).
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // (FAST: one row more in front — row 0 stays the pooled leaf phase's table, row 1 is the dummy and holds 0, so that a pop of the empty stack yields "no node" by itself)
    lds_u32* const stkBase = (lds_u32*)lds + lane + (FAST ? WAVE : 0);
    if (FAST) stkBase[0] = 0u;
    const int cap = f.stackCap;
    lds_u32* const stkFull = stkBase + cap * WAVE;
    uint32_t* tstk = lds + lane + (cap + 2) * WAVE;     // TLAS only
    const uint32_t N = *countPtr;
    {   // How many waves this launch wants is a function of its ray count, which only the device knows for sure (the host sizes the grid from the previous batch's
        // count when it has one, small_launch_grid in idkpt.hip): waves beyond that retire here, before they touch the work list.  Any number >= 1 is correct.
        uint32_t want = gridDim.x;
        if (f.gridRaysX4 > 0u) want = max((uint32_t)(((unsigned long long)N * 4ull / f.gridRaysX4 + 63ull) / 64ull), min(want, 1024u));
        if (f.gridMid > 0u && N < f.gridMidRays) want = min(want, f.gridMid);
        if (blockIdx.x >= max(want, 1u)) return;
    }
    // wave-uniform scene constants
    const GpuBlasInstance inst = s.instances[0];
    const int nodeOffset = s.descs[inst.BlasId].NodeOffset;
    const uint32_t triOffset = (uint32_t)s.descs[inst.BlasId].TriangleOffset;
    const float4* nodes = s.nodes + 2 * (size_t)nodeOffset;
    const float4* const pairBase = FAST ? s.pairNodes + 2 * (size_t)nodeOffset : nullptr;

    bool active = false, leafPending = false, workLeft = true;
    // work-list state of this wave (wave-uniform): the slice it grabs from, how many slices it has seen handed out, and the positions
    // [chunkNext, chunkEnd) of slice chunkSlice it has reserved but not yet given to lanes (only with a reservation size grabChunk > 0)
    uint32_t slice = blockIdx.x & (GRAB_SLICES - 1u), slicesDone = 0, chunkNext = 0, chunkEnd = 0, chunkSlice = 0;
    const uint32_t unitLog2 = (uint32_t)f.grabUnitLog2;           // a slice owns runs of 2^unitLog2 consecutive entries
    const uint32_t nBlocks = (N + (1u << unitLog2) - 1u) >> unitLog2;
    const uint32_t grabChunk = f.grabFixed > 0 ? (uint32_t)f.grabFixed : 0u;
    if (N == 0u) workLeft = false;
    uint32_t top = 0, slot = 0, leafFirst = 0, leafEnd = 0;
    uint32_t instIdx = 0, rayId = 0, nodeOff = 0, triOff = 0, xformId = 0;   // MULTI only: per-lane instance cursor (TLAS: next TLAS node) and BLAS offsets
    uint32_t vNode = 0, vTri = 0, vTlas = 0, vXform = 0;         // VER only: where this ray's scene version starts in nodes / triVerts (MULTI: also tlas / xforms), in float4 units
    int tsp = 0; bool moreInst = false;                                       // TLAS stack pointer; "there are instances / TLAS nodes left for this ray"
    lds_u32* sp = stkBase;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), invDir = splat3(0.0f);
    float hitT = 0.0f, hbx = 0.0f, hby = 0.0f; uint32_t hitTri = ~0u, hitXform = 0;
    uint32_t nPairs = 0, nTris = 0;
    bool ovf = false;                      // a push found the stack full (reported once, when the wave ends)
    // PROF: per-wave cycle buckets [refill, node, leaf, other], step counts and active-lane sums (developer instrumentation)
    unsigned long long pc[4] = {0, 0, 0, 0}, pn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tPrev = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
#define PROF_MARK(b) do { if (PROF) { unsigned long long _t = __builtin_amdgcn_s_memtime(); pc[b] += _t - tPrev; tPrev = _t; } } while (0)

    while (true) {
        PROF_MARK(3);
        // ---- refill idle lanes
        unsigned long long idle = __ballot(!active);
        if (workLeft && ((uint32_t)__popcll(idle) >= REFILL_MIN || idle == ~0ull)) {
            const uint32_t n = (uint32_t)__popcll(idle);
            if (PROF) { pn[1] += n; }
            // Handing out the work list.  One counter word takes about 88 atomics per microsecond on this chip (any single address does), and
            // coherent short rays ask for more: an atrium's primary rays (33 node steps, a wave refills ~50 lanes at a time) ran at exactly
            // 87 refills per microsecond whatever the traversal did.  So the list is dealt over GRAB_SLICES counters on different cache lines:
            // slice k owns the runs k, k + K, k + 2K, ... of 2^grabUnitLog2 consecutive entries and hands out positions of that sub-list.  All slices advance at the same
            // rate, so the entries in flight are the same contiguous window of the list as with one counter (reserving big chunks instead
            // costs incoherent scenes 2-13 %: neighbouring entries share nodes, and tails get longer); a wave starts at slice blockIdx % K
            // and moves on to the next slice when its own is handed out.  Which lane traces which entry is free.
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t avail = chunkEnd - chunkNext;          // wave-uniform: what is left of this wave's last reservation
            uint32_t q, sl; bool valid = true;                    // per lane: position inside a slice, the slice
            if (avail >= n) { q = chunkNext + rank; sl = chunkSlice; chunkNext += n; }
            else {
                const uint32_t need = n - avail, want = grabChunk > need ? grabChunk : need;   // (grabChunk 0: exactly what this refill needs)
                uint32_t fresh = 0, len = 0; bool got = false;
                while (slicesDone < GRAB_SLICES) {
                    len = ((nBlocks + GRAB_SLICES - 1u - slice) / GRAB_SLICES) << unitLog2;   // entries of this slice (the list's last run may be partial)
                    fresh = wave_grab(workCounter + GRAB_STRIDE * slice, want);
                    if (PROF) pn[0]++;
                    if (fresh < len) { got = true; break; }
                    slice = (slice + 1u) & (GRAB_SLICES - 1u); slicesDone++;                  // handed out: try the next one
                }
                q = rank < avail ? chunkNext + rank : fresh + (rank - avail); sl = rank < avail ? chunkSlice : slice;
                valid = rank < avail || (got && q < len);
                const uint32_t end = got ? (fresh + want < len ? fresh + want : len) : 0u;
                chunkNext = got ? (fresh + need < end ? fresh + need : end) : 0u; chunkEnd = end; chunkSlice = slice;
                if (got && fresh + want >= len) { slice = (slice + 1u) & (GRAB_SLICES - 1u); slicesDone++; }   // this reservation reached the slice's end
            }
            const uint32_t item = valid ? ((((q >> unitLog2) * GRAB_SLICES + sl) << unitLog2) | (q & ((1u << unitLog2) - 1u))) : N;
            if (slicesDone >= GRAB_SLICES && chunkNext >= chunkEnd) workLeft = false;
            if (!active && item < N) {
                // (bounce launches may be handed out in trace order, kernels_queue.hpp: position -> slot and ray id; the slot a hit is stored at does not change)
                const bool ordered = !PRIMARY && tr.order != nullptr;
                const uint32_t idx = ordered ? tr.orderIdx[item] : list[item];
                slot = PRIMARY ? idx : (ordered ? tr.order[item] : item);
                hitT = PT_FLOAT_MAX; hitTri = ~0u; hitXform = 0; hbx = 0.0f; hby = 0.0f;
                if (f.queryMode) { hitT = tr.rec[4 * (size_t)idx + 1].w; hitXform = __float_as_uint(tr.rec[4 * (size_t)idx + 2].w); }   // idkptTraceRays (kernels_query.hpp k_query_prepare): T = maxDist or the nearest light, and that light
                if (VER) { const uint32_t* vt = s.ver + SCENE_VER_WORDS * (size_t)(idx / f.Npad); vNode = vt[0]; vTri = vt[1]; if (MULTI) { vTlas = vt[3]; vXform = vt[4]; } }
                if (f.g.DoTraceLights) { // BVHIntersect.glsl:189-203 (world-space ray)
                    float4 o = rays.o_ior[idx];
                    f3 wd = DecodeUnitVec(rays.thr_px[idx].w, rays.rad_py[idx].w), wo = mk3(o.x, o.y, o.z);
                    for (int i = 0; i < s.lightCount; i++) {
                        const GpuLight& l = s.lights[i];
                        float tMin, tMax;
                        if (RaySphereIntersect(wo, wd, mk3(l.Position[0], l.Position[1], l.Position[2]), l.Radius, &tMin, &tMax) && tMin < hitT) { hitT = tMin < 0.0f ? tMax : tMin; hitXform = (uint32_t)i; hitTri = ~0u; }
                    }
                }
                if (MULTI) { rayId = idx; instIdx = 0; tsp = 0; moreInst = TLAS ? s.tlasCount > 0 : true; active = true; leafPending = false; sp = stkBase; top = 0u; }
                else {
                    // local-space ray and 1/dir were prepared by the (coherent, full-lane) kernel that produced this ray
                    float rootT;
                    { float4 a = tr.rec[4 * (size_t)idx], b = tr.rec[4 * (size_t)idx + 1], c = tr.rec[4 * (size_t)idx + 2]; ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); invDir = mk3(c.x, c.y, c.z); rootT = a.w; }
                    const bool enter = rootT < hitT;   // root test (:32-39): the box arithmetic ran in the kernel that produced the ray (record[0].w = tMin, +inf = miss)
                    active = true; leafPending = false; sp = stkBase; top = enter ? 2u : 0u;
                }
            }
        }
        PROF_MARK(0);
        if (__ballot(active) == 0ull) { if (!workLeft) break; continue; }

        if (TLAS) {
            // lanes whose current BLAS is exhausted continue their TLAS walk until it reaches the next leaf (= instance) or ends
            bool adv = active && !leafPending && top == 0u && moreInst;
            // (like parked leaves: the few lanes whose BLAS is exhausted wait until f.advMin of them can take their TLAS steps together — or nobody else has anything to do)
            if (MULTI && (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(adv)) < (uint32_t)f.advMin && __builtin_amdgcn_ballot_w64(active && (leafPending || top != 0u)) != 0ull) adv = false;
            while (__any(adv)) {
                if (adv) {
                    const float4* tl4 = VER ? s.tlas + vTlas : s.tlas;
                    const float4 pmin = tl4[2 * (size_t)instIdx];
                    const uint32_t packed = __float_as_uint(pmin.w), id = packed & 0x7fffffffu;
                    if ((packed >> 31) == 1u) {                                             // leaf: BVHIntersect.glsl:223-240
                        const GpuBlasInstance in2 = s.instances[id];
                        const M34 inv = load_inv_model_at(VER ? s.xforms + vXform : s.xforms, in2.MeshTransformId);
                        float4 a = tr.rec[4 * (size_t)rayId], b = tr.rec[4 * (size_t)rayId + 1];
                        ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                        invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        nodeOff = (uint32_t)s.descs[in2.BlasId].NodeOffset; triOff = (uint32_t)s.descs[in2.BlasId].TriangleOffset; xformId = in2.MeshTransformId;
                        sp = stkBase; top = 2u;                                             // no root test under USE_TLAS (:32)
                        if (tsp == 0 || tsp > f.tlasCap) moreInst = false; else instIdx = tstk[--tsp * WAVE];  // the pop the reference does after the BLAS; order-independent
                    } else {
                        const uint32_t l = id, r = id + 1;
                        const float4* w4 = tr.rec + 4 * (size_t)rayId;
                        float4 a = w4[0], c = w4[2];                                                                          // world-space origin and 1/dir
                        const f3 wo = mk3(a.x, a.y, a.z), winv = mk3(c.x, c.y, c.z);
                        float4 lmin = tl4[2 * (size_t)l], lmax = tl4[2 * (size_t)l + 1], rmin = tl4[2 * (size_t)r], rmax = tl4[2 * (size_t)r + 1];
                        float tMinLeft, tMinRight;
                        const bool tl = RayBoxIntersect(wo, winv, lmin, lmax, &tMinLeft) && tMinLeft < hitT;
                        const bool tr2 = RayBoxIntersect(wo, winv, rmin, rmax, &tMinRight) && tMinRight < hitT;
                        if (tl || tr2) {
                            if (tl && tr2) { const bool lc = tMinLeft < tMinRight; instIdx = lc ? l : r; if (tsp < f.tlasCap) tstk[tsp * WAVE] = lc ? r : l; else *s.overflow = 1u; tsp++; }
                            else instIdx = tl ? l : r;
                        } else { if (tsp == 0 || tsp > f.tlasCap) moreInst = false; else instIdx = tstk[--tsp * WAVE]; }
                    }
                }
                adv = active && !leafPending && top == 0u && moreInst;
            }
        } else if (MULTI) {
            // lanes whose current BLAS is exhausted move on to the next instance (loop: the root test may fail right away)
            bool adv = active && !leafPending && top == 0u && instIdx < (uint32_t)s.instanceCount;
            if (MULTI && (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(adv)) < (uint32_t)f.advMin && __builtin_amdgcn_ballot_w64(active && (leafPending || top != 0u)) != 0ull) adv = false;
            while (__any(adv)) {
                if (adv) {
                    // (three dependent fetches — instance, transform + descriptor, root node — that are cheap: the tables are tiny and hot, and the root node shares its 128-byte line
                    // with the pair the walk fetches next.  One gathered record per instance, DScene::instRec, measured 4-14 % SLOWER here in round 5: profiles/r05_instance_tlas.md)
                    const GpuBlasInstance in2 = s.instances[instIdx];
                    const M34 inv = load_inv_model_at(VER ? s.xforms + vXform : s.xforms, in2.MeshTransformId);
                    float4 a = tr.rec[4 * (size_t)rayId], b = tr.rec[4 * (size_t)rayId + 1];                         // world-space origin / direction
                    ro = xform34(inv, mk3(a.x, a.y, a.z), 1.0f); rd = xform34(inv, mk3(b.x, b.y, b.z), 0.0f);
                    invDir = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                    nodeOff = (uint32_t)s.descs[in2.BlasId].NodeOffset; triOff = (uint32_t)s.descs[in2.BlasId].TriangleOffset; xformId = in2.MeshTransformId;
                    const float4* root = (VER ? s.nodes + vNode : s.nodes) + 2 * (size_t)nodeOff + 2;
                    float t1;
                    const bool enter = RayBoxIntersect(ro, invDir, root[0], root[1], &t1) && t1 < hitT;
                    sp = stkBase; top = enter ? 2u : 0u;
                    instIdx++;
                }
                adv = active && !leafPending && top == 0u && instIdx < (uint32_t)s.instanceCount;
            }
        }

        // ---- node phase (branch-free step: see the kernel's header comment)
        while (true) {
            const bool canStep = active && !leafPending && top != 0u;
            const unsigned long long stepMask = __builtin_amdgcn_ballot_w64(canStep);
            if (stepMask == 0ull) break;
            if (LEAF_MIN <= 64 && __builtin_popcountll(__builtin_amdgcn_ballot_w64(active && leafPending)) >= (LEAF_MIN == 24 ? f.leafMin : LEAF_MIN)) break;   // (24 = "the host's choice")
            if (PROF) { pn[2]++; pn[3] += (unsigned long long)__builtin_popcountll(stepMask); }
            if (FAST) {
                if (canStep) {
                    const float4* p = pairBase + 2 * (size_t)top;
                    const uint32_t popped = sp[0];                      // (0 for an empty stack)
                    const float4 A = p[0], B = p[1], Z = p[2], D = p[3];
                    const uint32_t lStart = __float_as_uint(D.x), lCount = __float_as_uint(D.y), rStart = __float_as_uint(D.z), rCount = __float_as_uint(D.w);
                    // RayBoxIntersect (pt_device.hpp) on both boxes: the same subtractions, multiplications, minima and maxima on the same operands
                    const v2f oxy = {ro.x, ro.y}, ixy = {invDir.x, invDir.y}, ozz = {ro.z, ro.z}, izz = {invDir.z, invDir.z};
                    const v2f aL = (v2f{A.x, A.y} - oxy) * ixy, aR = (v2f{A.z, A.w} - oxy) * ixy, bL = (v2f{B.x, B.y} - oxy) * ixy, bR = (v2f{B.z, B.w} - oxy) * ixy;
                    const v2f az = (v2f{Z.x, Z.y} - ozz) * izz, bz = (v2f{Z.z, Z.w} - ozz) * izz;
                    const float tMinLeft = gmax(gmin(aL.x, bL.x), gmax(gmin(aL.y, bL.y), gmax(gmin(az.x, bz.x), 0.0f))), tMaxLeft = gmin(gmax(aL.x, bL.x), gmin(gmax(aL.y, bL.y), gmax(az.x, bz.x)));
                    const float tMinRight = gmax(gmin(aR.x, bR.x), gmax(gmin(aR.y, bR.y), gmax(gmin(az.y, bz.y), 0.0f))), tMaxRight = gmin(gmax(aR.x, bR.x), gmin(gmax(aR.y, bR.y), gmax(az.y, bz.y)));
                    const bool hitLeft = tMinLeft <= tMaxLeft && tMinLeft <= hitT, hitRight = tMinRight <= tMaxRight && tMinRight <= hitT;
                    const bool leafL = lCount != 0u, leafR = rCount != 0u;
                    // ("inner" = not "leaf" through the lane mask — one compare per child; written as `lCount == 0` the compiler issues a second one)
                    const bool innerL = __builtin_amdgcn_inverse_ballot_w64(~__builtin_amdgcn_ballot_w64(leafL)), innerR = __builtin_amdgcn_inverse_ballot_w64(~__builtin_amdgcn_ballot_w64(leafR));
                    const bool intersectLeft = hitLeft && leafL, intersectRight = hitRight && leafR;
                    leafFirst = intersectLeft ? lStart : rStart; leafEnd = !intersectRight ? lStart + lCount : rStart + rCount; leafPending = intersectLeft || intersectRight;
                    const bool traverseLeft = hitLeft && innerL, traverseRight = hitRight && innerR;
                    const bool both = traverseLeft && traverseRight, none = !(traverseLeft || traverseRight);
                    const bool leftCloser = tMinLeft < tMinRight;
                    const uint32_t nearChild = both ? (leftCloser ? lStart : rStart) : (traverseLeft ? lStart : rStart);
                    sp[WAVE] = leftCloser ? rStart : lStart;
                    top = none ? popped : nearChild;
                    sp += both ? (int)WAVE : (none ? -(int)WAVE : 0);      // (a pop of the empty stack leaves sp one row low: the lane is finished, its next ray resets it)
                }
            } else
            if (canStep) {
                if (COUNT) nPairs++;
                const float4* p = (MULTI ? s.nodes + 2 * ((size_t)nodeOff + top) : nodes + 2 * (size_t)top) + (VER ? vNode : 0u);
                const uint32_t popped = sp[0];                          // what a pop would return (in flight with the node pair; row 0 for an empty stack)
                float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
                const uint32_t lStart = __float_as_uint(lmin.w), lCount = __float_as_uint(lmax.w), rStart = __float_as_uint(rmin.w), rCount = __float_as_uint(rmax.w);
                float tMinLeft, tMinRight;
                const bool hitLeft = RayBoxIntersect(ro, invDir, lmin, lmax, &tMinLeft) && tMinLeft <= hitT;
                const bool hitRight = RayBoxIntersect(ro, invDir, rmin, rmax, &tMinRight) && tMinRight <= hitT;
                const bool intersectLeft = hitLeft && lCount > 0, intersectRight = hitRight && rCount > 0;
                // (a lane that steps has no parked leaf, so its leaf registers are free: written unconditionally, BLAS-local; the leaf phase adds the offset)
                leafFirst = intersectLeft ? lStart : rStart; leafEnd = !intersectRight ? lStart + lCount : rStart + rCount; leafPending = intersectLeft || intersectRight;
                if (COUNT) nTris += leafPending ? leafEnd - leafFirst : 0u;
                const bool traverseLeft = hitLeft && lCount == 0, traverseRight = hitRight && rCount == 0;
                const bool both = traverseLeft && traverseRight, none = !(traverseLeft || traverseRight);
                const bool leftCloser = ANY ? true : tMinLeft < tMinRight;   // (ANY: left first, BVHIntersect.glsl:165-168)
                const uint32_t nearChild = both ? (leftCloser ? lStart : rStart) : (traverseLeft ? lStart : rStart);
                sp[WAVE] = leftCloser ? rStart : lStart;                // the far child, above the top: part of the stack only if sp moves
                const bool full = sp == stkFull, nonEmpty = sp != stkBase;
                ovf = ovf || (both && full);                            // (the push is dropped and flagged: the upload-time validation makes this unreachable)
                top = none ? (nonEmpty ? popped : 0u) : nearChild;
                sp += (both && !full) ? (int)WAVE : ((none && nonEmpty) ? -(int)WAVE : 0);
            }
        }
        PROF_MARK(1);
        if (PROF) {   // leaf phases, lanes in them, and — for the "pooled (ray, triangle) pairs" question — the triangle tests of the wave and the loop trips (= the longest lane's count)
            unsigned long long lm = __ballot(leafPending);
            if (lm) {
                pn[4]++; pn[5] += (unsigned long long)__popcll(lm);
                uint32_t cntL = leafPending ? leafEnd - leafFirst : 0u, sum = cntL, mx = cntL;
                for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); }
                pn[6] += sum; pn[7] += mx;
            }
        }
        // ---- leaf phase
        if (POOL) {
            const unsigned long long pend = __builtin_amdgcn_ballot_w64(leafPending);
            const uint32_t cnt = leafPending ? leafEnd - leafFirst : 0u;
            if (pend != 0ull && __builtin_amdgcn_ballot_w64(cnt > 31u) == 0ull) {
                // exclusive prefix sum of the counts over the lanes (counts < 32: five ballots), and the wave's total
                uint32_t off = 0, total = 0;
#pragma unroll
                for (int b = 0; b < 5; b++) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(((cnt >> b) & 1u) != 0u);
                    off += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
                    total += (uint32_t)__builtin_popcountll(m) << b;
                }
                uint32_t maxCnt = cnt;
                for (int o = 32; o > 0; o >>= 1) maxCnt = max(maxCnt, (uint32_t)__shfl_xor((int)maxCnt, o));
                lds_u32* const tbl = (lds_u32*)lds;                                   // the dummy row (row 0) of the stack: 64 words nobody's result depends on
                // pooling saves the round trips after the first (maxCnt - 1 of them) and costs its bookkeeping: only where a lane has several triangles and the wave enough pairs
                if (maxCnt >= 2u && total >= (uint32_t)f.poolMin)
                for (uint32_t base = 0; base < total; base += 64u) {
                    for (uint32_t k = 0; k < maxCnt; k++) { const uint32_t j = off + k - base; if (k < cnt && j < 64u) tbl[j] = lane | (k << 8); }
                    __builtin_amdgcn_wave_barrier();
                    const bool have = base + lane < total;
                    const uint32_t e = have ? tbl[lane] : lane;                        // (idle slots test nothing: their result is +inf)
                    const int src = (int)((e & 63u) << 2);
                    const uint32_t kk = e >> 8;
                    const f3 oro = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ro.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ro.y))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ro.z))));
                    const f3 ord = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rd.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rd.y))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rd.z))));
                    const float oT = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(hitT)));
                    const uint32_t oFirst = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(leafFirst + (MULTI ? triOff : triOffset)));   // the owner's first triangle, scene-wide index
                    const uint32_t oVer = VER ? (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)vTri) : 0u;                               // ... and where its scene version's records start
                    float pt = __builtin_inff(), pby = 0.0f, pbz = 0.0f;
                    if (have) {
                        const float4* tv = s.triVerts + 3 * (size_t)(oFirst + kk) + oVer;
                        const float4 a = tv[0], b = tv[1], c = tv[2];
                        float by, bz, t;
                        if (RayTriangleIntersect(oro, ord, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < oT) { pt = t; pby = by; pbz = bz; }
                    }
                    // the owners collect their pairs of this pass in triangle order: exactly the reference's loop, the tests already done
                    for (uint32_t k = 0; k < maxCnt; k++) {
                        const uint32_t j = off + k - base;
                        const int from = (int)((j & 63u) << 2);
                        const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(pt)));
                        const float by = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(pby))), bz = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(pbz)));
                        if (k < cnt && j < 64u && t < hitT) { hitTri = leafFirst + k + (MULTI ? triOff : triOffset); hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = MULTI ? xformId : inst.MeshTransformId; }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (base + 64u >= total) leafPending = false;
                }
            }
        }
        if (leafPending) {
            const uint32_t tOff = MULTI ? triOff : triOffset;
            // (one triangle per round trip.  Requesting the two 48-B records of a two-triangle leaf together was measured in round 3: 88 instead of 75 VGPRs,
            // 5 instead of 6 waves per SIMD, every view 2-11 % slower; with the occupancy forced back, spills cost more — profiles/r03_trace_experiments.md)
            for (uint32_t i = leafFirst + tOff, e = leafEnd + tOff; i < e; i++) {
                const float4* tv = s.triVerts + 3 * (size_t)i + (VER ? vTri : 0u);
                float4 a = tv[0], b = tv[1], c = tv[2];
                float by, bz, t;
                if (RayTriangleIntersect(ro, rd, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &by, &bz, &t) && t < hitT) {
                    hitTri = i; hbx = 1.0f - by - bz; hby = by; hitT = t; hitXform = MULTI ? xformId : inst.MeshTransformId;
                    if (ANY) {   // the first intersection found wins: nothing of this ray is left to do
                        top = 0u; sp = stkBase;
                        if (MULTI) { instIdx = (uint32_t)s.instanceCount; moreInst = false; }
                        break;
                    }
                }
            }
            leafPending = false;
        }
        PROF_MARK(2);
        // ---- retire finished rays (MULTI: only after the last instance)
        if (active && top == 0u && (!MULTI || (TLAS ? !moreInst : instIdx >= (uint32_t)s.instanceCount))) {
            store_hit(hits, slot, hitT, hbx, hby, hitTri, hitXform);
            active = false;
        }
    }
    if (ovf) *s.overflow = 1u;
    if (COUNT) flush_counters(counters, nPairs, nTris);
    if (PROF && lane == 0) { for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)&counters[4 + i], pc[i]); for (int i = 0; i < 8; i++) atomicAdd((unsigned long long*)&counters[8 + i], pn[i]); }
#undef PROF_MARK
}
