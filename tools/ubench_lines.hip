// Developer micro-benchmark (gfx950): what does a divergent global_load_dwordx4 cost as a function of the number of DISTINCT
// cache lines it touches?  Independent loads (no dependent chain), L2/MALL-resident set, many waves -> L1/TA throughput.
//   mode 0: every lane its own random 64-B block, loads 4 x dwordx4 of it (today's node-pair fetch: 4 instr, 64 lines each)
//   mode 1: lane pairs share a block: even lane loads bytes 0..31, odd lane 32..63 (2 instr, 32 lines each) -> 32 blocks/wave-step
//   mode 2: quads share a block: each lane 16 B (1 instr, 16 lines)                                       -> 16 blocks/wave-step
//   mode 3: every lane its own block, ONE dwordx4 (request-rate probe)
//   mode 4: every lane its own block, ONE dword
//   mode 5: every lane its own random 128-B LINE (two adjacent blocks), 8 x dwordx4: does the second half of a line come for free?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lines.hip -o tools/ubench_lines.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }

template <int MODE>
__global__ __launch_bounds__(64) void k(const float4* __restrict__ buf, uint32_t mask, int iters, uint32_t* out)
{
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    const uint32_t owner = MODE == 1 ? (gid >> 1) : MODE == 2 ? (gid >> 2) : gid;
    #pragma unroll 4
    for (int i = 0; i < iters; i++) {
        const uint32_t idx = hash32(owner * 2654435761u + (uint32_t)i * 40503u) & mask;   // independent of loaded data
        const float4* p = buf + (size_t)idx * 4;
        if (MODE == 0) { float4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += __float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z) ^ __float_as_uint(d.w); }
        if (MODE == 1) { const float4* q = p + 2 * (lane & 1); float4 a = q[0], b = q[1]; acc += __float_as_uint(a.x) ^ __float_as_uint(b.y); }
        if (MODE == 2) { float4 a = p[lane & 3]; acc += __float_as_uint(a.x); }
        if (MODE == 3) { float4 a = p[0]; acc += __float_as_uint(a.x) ^ __float_as_uint(a.w); }
        if (MODE == 5) { const float4* q = buf + (size_t)(idx & ~1u) * 4; float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7];
                         acc += __float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z) ^ __float_as_uint(d.w) ^ __float_as_uint(e.x) ^ __float_as_uint(f.y) ^ __float_as_uint(g.z) ^ __float_as_uint(h.w); }
        if (MODE == 4) { acc += __float_as_uint(((const float*)p)[0]); }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv)
{
    int logBlocks = argc > 1 ? atoi(argv[1]) : 20;   // 2^20 x 64 B = 64 MB
    const int onlyMode = argc > 2 ? atoi(argv[2]) : -1, onlyWaves = argc > 3 ? atoi(argv[3]) : -1;   // bench.py: "<log2 blocks> 0 32" = the node-pair pattern at full occupancy
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    size_t nBlocks = (size_t)1 << logBlocks;
    float4* buf; CHECK(hipMalloc(&buf, nBlocks * 64));
    CHECK(hipMemset(buf, 1, nBlocks * 64));
    uint32_t* out; CHECK(hipMalloc(&out, 4));
    printf("CUs %d, set %zu MB\n", cus, nBlocks * 64 >> 20);
    for (int wavesPerCU : {16, 32}) for (int mode = 0; mode < 6; mode++) {
        if ((onlyMode >= 0 && mode != onlyMode) || (onlyWaves >= 0 && wavesPerCU != onlyWaves)) continue;
        int iters = 4000;
        dim3 grid(cus * wavesPerCU), block(64);
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out); break;
                case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out); break;
                case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double instrPerIter = mode == 5 ? 8 : mode == 0 ? 4 : mode == 1 ? 2 : 1;
        const double blocksPerIter = mode == 1 ? 32 : mode == 2 ? 16 : 64;
        const double waveIters = (double)grid.x * iters;
        const double clk = ms * 1e-3 * 2.4e9;
        printf("waves/CU %2d mode %d: %8.3f ms | %6.1f clk per wave-instr per CU | %6.3f blocks/clk/CU | %6.3f lane-requests/clk/CU\n", wavesPerCU, mode, ms,
               clk / (waveIters / cus * instrPerIter), waveIters * blocksPerIter / cus / clk, waveIters * instrPerIter * 64 / cus / clk);
        // machine-readable (bench.py): distinct 64-B blocks fetched per second, as GB/s of 64-B blocks
        printf("RESULT set_mb=%zu waves_per_cu=%d mode=%d ms=%.4f blocks_per_s=%.6e gbs=%.2f\n", nBlocks * 64 >> 20, wavesPerCU, mode, ms, waveIters * blocksPerIter / (ms * 1e-3), waveIters * blocksPerIter * 64.0 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
