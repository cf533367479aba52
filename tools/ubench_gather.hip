// Developer micro-benchmark (gfx950): how fast can a CU gather random 64-byte node pairs?
//   A: each lane fetches its own 64-B block as 4 x global_load_dwordx4 (what a thread-per-ray traversal does)
//   B: quad-cooperative: 4 consecutive lanes fetch one 64-B block (16 B each) -> 16 blocks per wave-instruction
//   C: as A but 2 x dwordx4 (32 B, one node)
// Dependent chain per lane (next index derives from loaded data), many waves for latency hiding.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o gpurun_out/ubench_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }

template <int MODE>
__global__ __launch_bounds__(64) void k_gather(const float4* __restrict__ buf, uint32_t nBlocksMask, int iters, uint32_t* out)
{
    uint32_t lane = threadIdx.x;
    uint32_t gid = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    if (MODE >= 3) { // as A, but only every 2nd (MODE 3) / 4th (MODE 4) lane active, or a random half (MODE 5): does the memory pipeline charge for masked lanes?
        bool on = MODE == 3 ? (lane & 1) == 0 : MODE == 4 ? (lane & 3) == 0 : (hash32(lane * 7919u) & 1) == 0;
        uint32_t idx = hash32(gid) & nBlocksMask;
        if (on) for (int i = 0; i < iters; i++) {
            const float4* p = buf + (size_t)idx * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            uint32_t v = __float_as_uint(a.x) ^ __float_as_uint(b.w) ^ __float_as_uint(c.y) ^ __float_as_uint(d.z);
            acc += v;
            idx = hash32(idx ^ v) & nBlocksMask;
        }
    } else
    if (MODE == 0 || MODE == 2) {
        uint32_t idx = hash32(gid) & nBlocksMask;
        for (int i = 0; i < iters; i++) {
            const float4* p = buf + (size_t)idx * 4;
            float4 a = p[0], b = p[1];
            uint32_t v = __float_as_uint(a.x) ^ __float_as_uint(b.w);
            if (MODE == 0) { float4 c = p[2], d = p[3]; v ^= __float_as_uint(c.y) ^ __float_as_uint(d.z); }
            acc += v;
            idx = hash32(idx ^ v) & nBlocksMask;
        }
    } else {
        uint32_t grp = gid >> 2, sub = lane & 3;
        uint32_t idx = hash32(grp) & nBlocksMask;
        for (int i = 0; i < iters; i++) {
            float4 a = buf[(size_t)idx * 4 + sub];
            uint32_t v = __float_as_uint(a.x) ^ __float_as_uint(a.w);
            v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
            acc += v;
            idx = hash32(idx ^ v) & nBlocksMask;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv)
{
    int onlyLog = argc > 1 ? atoi(argv[1]) : 20; int onlyMode = argc > 2 ? atoi(argv[2]) : -1;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* out; CHECK(hipMalloc(&out, 4));
    for (int logBlocks : {onlyLog}) { // 4 MB, 64 MB, 1 GB of 64-B blocks
        size_t nBlocks = (size_t)1 << logBlocks;
        float4* buf; CHECK(hipMalloc(&buf, nBlocks * 64));
        std::vector<uint32_t> h(nBlocks * 16);
        uint32_t s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s; }
        CHECK(hipMemcpy(buf, h.data(), nBlocks * 64, hipMemcpyHostToDevice));
        for (int wavesPerCU : {16, 32}) {
            for (int mode = 0; mode < 6; mode++) {
                if (onlyMode >= 0 && mode != onlyMode) continue;
                int iters = 2000;
                dim3 grid(cus * wavesPerCU), block(64);
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k_gather<0>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out);
                    if (mode == 1) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out);
                    if (mode == 2) hipLaunchKernelGGL(k_gather<2>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out);
                    if (mode == 3) hipLaunchKernelGGL(k_gather<3>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out);
                    if (mode == 4) hipLaunchKernelGGL(k_gather<4>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out);
                    if (mode == 5) hipLaunchKernelGGL(k_gather<5>, grid, block, 0, 0, buf, (uint32_t)(nBlocks - 1), iters, out);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                }
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                double fetches = (double)grid.x * (mode == 1 ? 16 : mode == 3 ? 32 : mode == 4 ? 16 : mode == 5 ? 32 : 64) * iters; // 64-B (or 32-B for mode 2) block fetches
                double perSec = fetches / (ms * 1e-3);
                printf("set %4zu MB waves/CU %2d mode %c: %8.3f ms  %7.2f Gfetch/s  %6.3f fetch/clk/CU@2.4GHz  %7.1f GB/s\n", nBlocks * 64 >> 20, wavesPerCU, "ABCDEF"[mode], ms, perSec / 1e9,
                       perSec / cus / 2.4e9, perSec * (mode == 2 ? 32 : 64) / 1e9);
            }
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
