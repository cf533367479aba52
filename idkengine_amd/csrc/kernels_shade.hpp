// kernels_shade.hpp — FirstHit / NHit shading kernels (FirstHit/compute.glsl:114-233, NHit/compute.glsl:98-214).
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

// FirstHit / NHit part 2: shade + BSDF sample + continue decision.  One thread per queue slot; the continue bits of a
// wave are published as one 64-bit ballot + popcount for the ordered compaction that follows.
// local-space ray + 1/dir of a continuing ray, for the next traversal launch (single-instance fast path): exactly what
// NHit does first (decode the packed direction, NHit:93; RayTransform, BVHIntersect.glsl:281-282; 1/dir, IntersectionRoutines.glsl:29)
DEV void write_trace_ready(const DScene& s, const Frame& f, const TraceBufs& tr, uint32_t rid, const RayState& r)
{
    f3 rd = DecodeUnitVec(r.pdx, r.pdy);
    if (f.useTlas || s.instanceCount > 1) {   // the traversal kernel walks the TLAS / instance list and transforms the world ray itself
        tr.rec[4 * (size_t)rid] = make_float4(r.origin.x, r.origin.y, r.origin.z, 0.0f); tr.rec[4 * (size_t)rid + 1] = make_float4(rd.x, rd.y, rd.z, 0.0f);
        if (f.useTlas || f.instTlas) tr.rec[4 * (size_t)rid + 2] = make_float4(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z, 0.0f);   // world 1/dir for the TLAS slab tests (:209)
        return;
    }
    GpuBlasInstance inst = s.instances[0];
    M34 inv = load_inv_model(s, inst.MeshTransformId);
    f3 lo = xform34(inv, r.origin, 1.0f), ld = xform34(inv, rd, 0.0f);
    const f3 iv = mk3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
    // lo.w = tMin of the root-box test IntersectBlas does first (BVHIntersect.glsl:32-39), +inf when the ray misses the box: the traversal
    // kernel only has to compare it with its T, the box arithmetic runs here with all lanes busy
    const float4* root = s.nodes + 2 * (size_t)s.descs[inst.BlasId].NodeOffset + 2;
    float t1;
    const float rootT = RayBoxIntersect(lo, iv, root[0], root[1], &t1) ? t1 : __builtin_inff();
    tr.rec[4 * (size_t)rid] = make_float4(lo.x, lo.y, lo.z, rootT); tr.rec[4 * (size_t)rid + 1] = make_float4(ld.x, ld.y, ld.z, 0.0f);
    tr.rec[4 * (size_t)rid + 2] = make_float4(iv.x, iv.y, iv.z, 0.0f);
}

// Fast-path FirstHit shading: only the rays that entered the traversal (active list, any order).  The continue decision
// goes to a per-ray byte (pre-zeroed), which the ordered compaction turns back into pixel order.
template <bool VER /* scene versions: every sample shades with the geometry it was queued with (pt_kernels.hpp DScene::ver) */>
__global__ __launch_bounds__(256) void k_shade_first(DScene s0, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* activeList, const uint32_t* activeCount,
                                                     uint8_t* contFlag, uint32_t* seedsAndKeys, int lean /* k_gen_primary stored nothing but the trace-ready record of this ray */,
                                                     uint32_t* pmList /* or null */, uint32_t* pmCount)
{
    __shared__ uint32_t waveCont[4]; __shared__ uint32_t blockBase;
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = item < *activeCount;
    if (!pmList && !live) return;
    bool cont = false; uint32_t rid = 0;
    if (live) {
        rid = activeList[item];
        const uint32_t smp = rid / f.Npad, pix = rid - smp * f.Npad;
        const DScene s = VER ? scene_of_sample(s0, smp) : s0;
        const uint32_t acc = sample_index(f, smp);
        const HitRec hit = load_hit(hits, rid);
        RayState r; uint32_t rng, key = 0;
        if (lean) {   // the state FirstHit:44-77 starts a primary ray with, recomputed (same arithmetic, same bits) instead of 52 B written and read back
            f2 pd; gen_primary(f, smp, pix, acc, r.origin, pd, rng);
            r.prevIor = 1.0f; r.throughput = splat3(1.0f); r.pdx = pd.x; r.radiance = splat3(0.0f); r.pdy = pd.y;
        } else {
            float4 a = rays.o_ior[rid], b = rays.thr_px[rid], c = rays.rad_py[rid];
            r.origin = mk3(a.x, a.y, a.z); r.prevIor = a.w; r.throughput = mk3(b.x, b.y, b.z); r.pdx = b.w; r.radiance = mk3(c.x, c.y, c.z); r.pdy = c.w;
            rng = seedsAndKeys[rid];
        }
        AovState aov; aov.albedo = splat3(0.0f); aov.normal = splat3(0.0f); aov.newWeight = 1.0f;
        int lx = (int)(pix % (uint32_t)f.W), ly = (int)(pix / (uint32_t)f.W);
        uint32_t gidSeed = first_hit_gid_seed(f.W, f.H, lx, global_row(f, ly));
        f3 rd = DecodeUnitVec(r.pdx, r.pdy);
        cont = ShadeHit<true>(s, f, acc, hit, hit.T != PT_FLOAT_MAX, rd, r, aov, rng, gidSeed, key);
        rays.o_ior[rid] = make_float4(r.origin.x, r.origin.y, r.origin.z, r.prevIor);
        rays.thr_px[rid] = make_float4(r.throughput.x, r.throughput.y, r.throughput.z, r.pdx);
        rays.rad_py[rid] = make_float4(r.radiance.x, r.radiance.y, r.radiance.z, r.pdy);
        if (f.outputAovs) { rays.aovA[rid] = make_float4(aov.albedo.x, aov.albedo.y, aov.albedo.z, aov.newWeight); rays.aovN[rid] = make_float4(aov.normal.x, aov.normal.y, aov.normal.z, 0.0f); }
        seedsAndKeys[rid] = (key & ((1u << IDKPT_SORT_KEY_BITS) - 1u)) | (smp << IDKPT_SORT_KEY_BITS);
        if (cont) { contFlag[rid] = 1; write_trace_ready(s, f, tr, rid, r); }
    }
    if (pmList) {
        // the first bounce's work list in THIS kernel's order — the pixel-major order of the primary list (k_gen_primary): the continuing rays of a pixel's samples side by side,
        // rays that start within a pixel's footprint of each other.  The alive queue (ordered compaction, slot = RNG seed) is built as always; this list only says in which
        // order the traversal kernel takes the rays, and their hits are stored per ray id (Frame::hitsByRid).  One atomic per workgroup.
        const unsigned long long m = __ballot(cont);
        const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) waveCont[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t tot = 0; for (int i = 0; i < 4; i++) { const uint32_t c = waveCont[i]; waveCont[i] = tot; tot += c; } blockBase = tot ? atomicAdd(pmCount, tot) : 0u; }
        __syncthreads();
        if (cont) pmList[blockBase + waveCont[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = rid;
    }
}

template <bool FIRST, bool VER = false>
__global__ __launch_bounds__(256) void k_shade(DScene s0, Frame f, RayBufs rays, TraceBufs tr, HitBufs hits, const uint32_t* queue, const uint32_t* countPtr, uint32_t countImm,
                                               const uint32_t* qbase, const uint32_t* gbase /* null: single context */, unsigned long long* contMask, uint32_t* waveCounts, uint32_t* keysTmp)
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
        const DScene s = VER ? scene_of_sample(s0, smp) : s0;
        const uint32_t acc = sample_index(f, smp);
        float4 a = rays.o_ior[idx], b = rays.thr_px[idx], c = rays.rad_py[idx];
        const HitRec hit = load_hit(hits, (!FIRST && f.hitsByRid) ? idx : slot);   // (hitsByRid: the bounce was traced by k_trace_fused, which stores hits per ray id)
        if (FIRST && f.g.DoDebugBVHTraversal) {
            rays.o_ior[idx] = make_float4(a.x, a.y, a.z, hits.cost[slot]); // FirstHit:108-112
        } else {
            RayState r; r.origin = mk3(a.x, a.y, a.z); r.prevIor = a.w; r.throughput = mk3(b.x, b.y, b.z); r.pdx = b.w; r.radiance = mk3(c.x, c.y, c.z); r.pdy = c.w;
            AovState aov; aov.albedo = splat3(0.0f); aov.normal = splat3(0.0f); aov.newWeight = 1.0f;
            if (!FIRST && f.outputAovs) { float4 aa = rays.aovA[idx], an = rays.aovN[idx]; aov.albedo = mk3(aa.x, aa.y, aa.z); aov.newWeight = aa.w; aov.normal = mk3(an.x, an.y, an.z); }
            uint32_t rng, gidSeed;
            if (FIRST) {
                f3 o2; f2 pd2; gen_primary(f, smp, pix, acc, o2, pd2, rng); // re-derives the RNG state after ray generation (cheaper than 4 B/pixel of HBM)
                int lx = (int)(pix % (uint32_t)f.W), ly = (int)(pix / (uint32_t)f.W);
                gidSeed = first_hit_gid_seed(f.W, f.H, lx, global_row(f, ly));
            } else {
                uint32_t gslot = (gbase ? gbase[f.gbBands ? smp * (uint32_t)f.gbStride + ((pix / (uint32_t)f.W) >> f.rowBandLog2) : smp] : 0u) + (slot - qbase[smp]);  // slot inside this sample's own queue (+ the alive rays of the strips above, idkptSetBounceExchange)
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
            if (cont && tr.rec) write_trace_ready(s, f, tr, idx, r);
        }
        // NHit:81 masks the key to 21 bits; the sample index goes above it so that the batch-wide sort stays grouped by sample
        keysTmp[slot] = (key & ((1u << IDKPT_SORT_KEY_BITS) - 1u)) | (smp << IDKPT_SORT_KEY_BITS);
    }
    unsigned long long m = __ballot(cont);
    if ((threadIdx.x & 63) == 0) contMask[slot >> 6] = m;
    (void)waveCounts;
}

// ---- the LAST bounce of a sample, when nothing of it is visible but radiance ---------------------------------------------------------------------------
// NHit of the last bounce computes a continuation nobody traces: a new direction, throughput, a Russian-roulette decision, the next queue.  The image needs one
// thing of it: the radiance a ray picks up at this hit — the sky on a miss, the surface's (or light's) emission on a hit.  Without AOVs that is all k_shade_last
// computes.  In a scene without emission (checked at upload: every EmissiveFactor and EmissiveBias zero, every texel finite; no light hits enabled) a hit adds
// sEmissive * throughput = 0, so only the misses are touched: the sky is added (the arithmetic of ShadeHit's miss branch) and the replaced radiance remembered.
// Otherwise (ALL_HITS) every hit runs ShadeHit on a copy of its state up to the point where its radiance is final (alpha test, absorption, emission).  Everything
// else of the bounce — ray state, alive queue, counts: what idkptDownloadRays / idkptDownloadAliveQueue / the next scene update may look at — is produced on demand
// by the ordinary kernels (finish_deferred in idkpt.hip: k_restore_last, then k_shade<false>, the scan and the scatter), bit for bit what the eager path
// leaves.  A hit whose throughput is not finite (0 * inf is not 0) takes ShadeHit on a copy of its state right here.
template <bool ALL_HITS /* the scene emits, or lights are hit: every hit may add radiance */, bool VER = false>
__global__ __launch_bounds__(256) void k_shade_last(DScene s0, Frame f, RayBufs rays, HitBufs hits, const uint32_t* queue, const uint32_t* countPtr, const uint32_t* qbase,
                                                    float4* radSave, uint32_t* deferCount)
{
    const uint32_t N = *countPtr;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot == 0u) *deferCount = N;                                // (the batch's last kernel resets the queue lengths: the on-demand pass reads this copy)
    if (slot >= N) return;
    const uint32_t idx = queue[slot];
    const HitRec hit = load_hit(hits, f.hitsByRid ? idx : slot);
    const float4 b = rays.thr_px[idx];
    const bool miss = hit.T == PT_FLOAT_MAX;
    const bool odd = ALL_HITS || !(__builtin_isfinite(b.x) && __builtin_isfinite(b.y) && __builtin_isfinite(b.z));
    if (!miss && !odd) return;
    const float4 c = rays.rad_py[idx];
    radSave[slot] = c;
    if (miss) {
        const f3 albedo = SampleSky(s0, DecodeUnitVec(b.w, c.w));   // (the sky is not versioned)
        const f3 rad = mk3(c.x, c.y, c.z) + albedo * mk3(b.x, b.y, b.z);
        rays.rad_py[idx] = make_float4(rad.x, rad.y, rad.z, c.w);
    } else {
        const uint32_t smp = idx / f.Npad;
        const DScene s = VER ? scene_of_sample(s0, smp) : s0;
        const uint32_t acc = sample_index(f, smp);
        const float4 a = rays.o_ior[idx];
        RayState r; r.origin = mk3(a.x, a.y, a.z); r.prevIor = a.w; r.throughput = mk3(b.x, b.y, b.z); r.pdx = b.w; r.radiance = mk3(c.x, c.y, c.z); r.pdy = c.w;
        AovState aov; aov.albedo = splat3(0.0f); aov.normal = splat3(0.0f); aov.newWeight = 1.0f;
        const uint32_t gslot = slot - qbase[smp];
        uint32_t rng = gslot * 4096u + acc, key = 0;
        (void)ShadeHit<false, true>(s, f, acc, hit, true, DecodeUnitVec(b.w, c.w), r, aov, rng, gslot, key);   // the radiance of this hit, on a copy of the state
        rays.rad_py[idx] = make_float4(r.radiance.x, r.radiance.y, r.radiance.z, c.w);
    }
}
// on demand, before the ordinary kernels run for the deferred bounce: the radiance k_shade_last replaced
template <bool ALL_HITS>
__global__ __launch_bounds__(256) void k_restore_last(RayBufs rays, HitBufs hits, const uint32_t* queue, const uint32_t* countPtr, const float4* radSave, int hitsByRid)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= *countPtr) return;
    const uint32_t idx = queue[slot];
    const HitRec hit = load_hit(hits, hitsByRid ? idx : slot);
    const float4 b = rays.thr_px[idx];
    if (ALL_HITS || hit.T == PT_FLOAT_MAX || !(__builtin_isfinite(b.x) && __builtin_isfinite(b.y) && __builtin_isfinite(b.z))) rays.rad_py[idx] = radSave[slot];
}

