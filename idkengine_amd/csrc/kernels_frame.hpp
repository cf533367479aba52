// kernels_frame.hpp — FinalDraw, ray-state completion for idkptDownloadRays, miss pre-fill, derived triangle layout.
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

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
__global__ __launch_bounds__(256) void k_final_draw(DScene s, Frame f, RayBufs rays, float4* imgResult, float4* imgAlbedo, float4* imgNormal, uint32_t N, const uint8_t* tileClass,
                                                    uint32_t* zeroWork, uint32_t nWork, uint32_t* zeroCounts, uint32_t nCounts)
{
    // last kernel of a batch: nothing reads this batch's work-list and queue-length counters any more, so they are reset here for the next batch
    // (two memset launches per batch less: they were 5 % of a frame that is rendered alone)
    if (blockIdx.x == 0) { for (uint32_t k = threadIdx.x; k < nWork; k += blockDim.x) zeroWork[k] = 0u; for (uint32_t k = threadIdx.x; k < nCounts; k += blockDim.x) zeroCounts[k] = 0u; }
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // the images of ring slot q start at q*N; samples are accumulated in submission order, exactly like consecutive FinalDraw dispatches,
    // each into the image of its own frame (consecutive samples of one frame share a slot)
    uint32_t slot = f.slotOf[0];
    float4 o = imgResult[(size_t)slot * N + i];
    f3 r = mk3(o.x, o.y, o.z), ra = splat3(0.0f), rn = splat3(0.0f);
    if (f.outputAovs) { float4 oa = imgAlbedo[(size_t)slot * N + i], on = imgNormal[(size_t)slot * N + i]; ra = mk3(oa.x, oa.y, oa.z); rn = mk3(on.x, on.y, on.z); }
    const uint32_t px = i % (uint32_t)f.W, py = i / (uint32_t)f.W, tilesX = ((uint32_t)f.W + 7) / 8;
    const uint32_t tile = (py >> 3) * tilesX + (px >> 3), nTilesAll = tilesX * (((uint32_t)f.rows + 7) / 8);
    const uint32_t sharedCls = (tileClass && !f.tilePerSample) ? tileClass[tile] : 0u;   // one camera and one scene version for the whole batch: one lookup
    for (int k = 0; k < f.batch; k++) {
        if (f.slotOf[k] != slot) {                       // next frame of the ring: store, switch, load
            imgResult[(size_t)slot * N + i] = make_float4(r.x, r.y, r.z, 1.0f);
            if (f.outputAovs) { imgAlbedo[(size_t)slot * N + i] = make_float4(ra.x, ra.y, ra.z, 1.0f); imgNormal[(size_t)slot * N + i] = make_float4(rn.x, rn.y, rn.z, 1.0f); }
            slot = f.slotOf[k];
            o = imgResult[(size_t)slot * N + i]; r = mk3(o.x, o.y, o.z);
            if (f.outputAovs) { float4 oa = imgAlbedo[(size_t)slot * N + i], on = imgNormal[(size_t)slot * N + i]; ra = mk3(oa.x, oa.y, oa.z); rn = mk3(on.x, on.y, on.z); }
        }
        const size_t rid = (size_t)k * f.Npad + i;
        float w = 1.0f / ((float)f.accum[k] + 1.0f);
        // pixels of a pre-classified tile (k_classify_tiles): the radiance is the known sky colour of the class; nothing was stored per sample
        const uint32_t cls = !tileClass ? 0u : (f.tilePerSample ? tileClass[(size_t)k * nTilesAll + tile] : sharedCls);   // one class per camera / scene version
        f3 nr = splat3(0.0f);
        if (cls == 0u) { float4 c = rays.rad_py[rid]; nr = mk3(c.x, c.y, c.z); }
        else if (cls <= 6u) { const float4 p = s.sky[cls - 1u]; nr = splat3(0.0f) + mk3(p.x, p.y, p.z) * splat3(1.0f); }
        else if (cls == 8u) {                            // textured sky: the miss branch of FirstHit (FirstHit:225-233) for this sample's ray, generated here (k_regen_culled does the same on demand)
            f3 origin; f2 pd; uint32_t seed;
            gen_primary(f, (uint32_t)k, i, sample_index(f, (uint32_t)k), origin, pd, seed);
            const f3 albedo = SampleSky(s, DecodeUnitVec(pd.x, pd.y));
            nr = splat3(0.0f) + albedo * splat3(1.0f);
        }
        if (f.g.DoDebugBVHTraversal) nr = TurboColormap(rays.o_ior[rid].w / 150.0f);
        r = gmix(r, nr, w);
        if (f.outputAovs) { float4 a = rays.aovA[rid], n = rays.aovN[rid]; ra = gmix(ra, mk3(a.x, a.y, a.z), w); rn = gmix(rn, mk3(n.x, n.y, n.z), w); }
    }
    imgResult[(size_t)slot * N + i] = make_float4(r.x, r.y, r.z, 1.0f);
    if (f.outputAovs) { imgAlbedo[(size_t)slot * N + i] = make_float4(ra.x, ra.y, ra.z, 1.0f); imgNormal[(size_t)slot * N + i] = make_float4(rn.x, rn.y, rn.z, 1.0f); }
}

// idkptDownloadRays support: the ray-state planes k_gen_primary skipped for culled pixels (flag 2) of one sample of the batch
__global__ __launch_bounds__(256) void k_regen_culled(DScene s, Frame f, RayBufs rays, const uint8_t* contFlag, uint32_t smp, uint32_t N)
{
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= N) return;
    const size_t rid = (size_t)smp * f.Npad + pix;
    const uint8_t flag = contFlag[rid];
    if (flag != 2 && flag != 4) return;                                 // 2: culled per pixel, 4: culled per tile (no ray was generated at all); bit 0 = continues
    f3 origin; f2 pd; uint32_t seed;
    gen_primary(f, smp, pix, sample_index(f, smp), origin, pd, seed);
    rays.o_ior[rid] = make_float4(origin.x, origin.y, origin.z, 1.0f);
    rays.thr_px[rid] = make_float4(1.0f, 1.0f, 1.0f, pd.x);
    if (flag == 4) {                                                    // the miss branch of FirstHit (FirstHit:225-233) for a pixel nothing was stored for
        const f3 albedo = SampleSky(s, DecodeUnitVec(pd.x, pd.y));
        const f3 radiance = splat3(0.0f) + albedo * splat3(1.0f);
        rays.rad_py[rid] = make_float4(radiance.x, radiance.y, radiance.z, pd.y);
    }
}

// test support (idkptEnablePrimaryHitCapture): miss records for the pixels the pre-cull removes before the traversal
__global__ void k_fill_miss(HitBufs hits, size_t first, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store_hit(hits, first + i, PT_FLOAT_MAX, 0.0f, 0.0f, ~0u, 0u);
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

// test support (idkptEnablePrimaryHitCapture): the first half of `n` consecutive 32-B hit records -> a dense float4 array
__global__ void k_capture_primary(HitBufs hits, size_t first, uint32_t n, float4* out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hits.hit[2 * (first + i)];
}
