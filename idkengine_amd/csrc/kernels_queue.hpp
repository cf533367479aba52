// kernels_queue.hpp — ordered compaction (scan + scatter) and the stable radix sort of the alive queue.
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

// idkptSetBandExchangeDevice: counts[k][b] = starts[k][b + 1] - starts[k][b] before the host's enqueued exchange, tab[k][b] = bases[k][b] - starts[k][b] after it (k_shade adds
// the position inside the sample's segment): the two ends of the device-side path, no host synchronisation in between
__global__ __launch_bounds__(256) void k_band_counts(const uint32_t* starts, int B, int LB, uint32_t* counts)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)B * (uint32_t)LB) return;
    const uint32_t k = t / (uint32_t)LB, b = t % (uint32_t)LB;
    counts[t] = starts[k * (uint32_t)(LB + 1) + b + 1] - starts[k * (uint32_t)(LB + 1) + b];
}
__global__ __launch_bounds__(256) void k_band_tab(const uint32_t* starts, const uint32_t* bases, int B, int LB, uint32_t* tab)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)B * (uint32_t)LB) return;
    const uint32_t k = t / (uint32_t)LB, b = t % (uint32_t)LB;
    tab[t] = bases[t] - starts[k * (uint32_t)(LB + 1) + b];
}

// idkptSetBandExchange: where, inside sample k's segment [bases[k], bases[k + 1]) of the alive queue (ray ids = k * Npad + local pixel, ascending: ordered compaction),
// the rays of local band b start — starts[k * (LB + 1) + b], relative to bases[k]; b = LB: the segment's length.  One binary search per (sample, band).
__global__ __launch_bounds__(256) void k_band_starts(const uint32_t* queue, const uint32_t* bases, int B, int LB, uint32_t bandPixels, uint32_t Npad, uint32_t* starts)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)B * (uint32_t)(LB + 1)) return;
    const uint32_t k = t / (uint32_t)(LB + 1), b = t % (uint32_t)(LB + 1);
    const uint32_t lo0 = bases[k], hi0 = bases[k + 1];
    const unsigned long long key = (unsigned long long)k * Npad + (unsigned long long)b * bandPixels;   // first ray id of that band
    uint32_t lo = lo0, hi = hi0;
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((unsigned long long)queue[mid] < key) lo = mid + 1; else hi = mid; }
    starts[t] = lo - lo0;
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
                                                      const unsigned long long* contMask, const uint32_t* curBase /* null: FIRST (slots are ray ids) */, uint32_t Npad, int batch, uint32_t* nextBase,
                                                      uint32_t* hostNextCount, uint32_t* hostNextBase /* mirrors in host-mapped memory (read by the host after a synchronisation: no copy kernels) */,
                                                      const uint32_t* activeCount, uint32_t* hostActiveCount /* FIRST only: length of the primary active list */)
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
    if (t == 1023) { *nextCount = part[1023]; *hostNextCount = part[1023]; if (tracedRays) atomicAdd(tracedRays, (unsigned long long)part[1023]); }
    if (t == 0 && hostActiveCount) *hostActiveCount = *activeCount;
    __threadfence_block();
    __syncthreads();
    // first slot of every sample in the NEXT queue = number of survivors in front of the sample's first current slot
    if (t <= (uint32_t)batch) {
        uint32_t g = (t == (uint32_t)batch) ? N : (curBase ? curBase[t] : t * Npad);
        g = min(g, N);
        uint32_t w = g >> 6, l = g & 63;
        uint32_t v = part[1023];
        if (w < nW) v = blockSums[w / SCAN_WAVES_PER_BLOCK] + waveLocal[w] + (uint32_t)__popcll(contMask[w] & ((1ull << l) - 1ull));
        nextBase[t] = v; hostNextBase[t] = v;
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
        uint32_t key = valid ? keys[i] : 0, val = valid ? (vals ? vals[i] : i) : 0;   // (no value array: the item's own index)
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
