// Developer micro-benchmark: does a gfx950 SIMD-32 skip the second pass of a wave64 VALU instruction when the upper (or lower) 32 lanes
// are masked off?  Runs a dependent-chain-free FMA loop with (a) all 64 lanes, (b) lanes 0-31 only, (c) lanes 32-63 only, (d) every
// other lane.  Equal times for (a)-(d) = no skipping (cost per wave instruction is independent of the exec mask).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_halfwave.bin tools/ubench_halfwave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void k(float* out, int iters, int mode)
{
    const int lane = threadIdx.x;
    bool on = mode == 0 || (mode == 1 && lane < 32) || (mode == 2 && lane >= 32) || (mode == 3 && (lane & 1) == 0) || (mode == 4 && lane < 16);
    float a0 = lane * 0.001f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    if (on) {
        for (int i = 0; i < iters; i++) {
            a0 = a0 * 1.0001f + 0.5f; a1 = a1 * 1.0001f + 0.5f; a2 = a2 * 1.0001f + 0.5f; a3 = a3 * 1.0001f + 0.5f;
            a4 = a4 * 1.0001f + 0.5f; a5 = a5 * 1.0001f + 0.5f; a6 = a6 * 1.0001f + 0.5f; a7 = a7 * 1.0001f + 0.5f;
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 32 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"all 64 lanes", "lanes 0-31", "lanes 32-63", "even lanes", "lanes 0-15"};
    for (int wavesPerCU : {4, 16, 32}) {
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(k, dim3(256 * wavesPerCU), dim3(64), 0, 0, d, 1000, mode);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256 * wavesPerCU), dim3(64), 0, 0, d, 200000, mode);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions issued per CU: wavesPerCU * iters * 8 (v_fma or v_mul+v_add under -ffp-contract default = fma)
            printf("waves/CU %2d  %-14s %8.3f ms   %.2f clk/wave-instr/SIMD @2.4GHz\n", wavesPerCU, names[mode], ms, ms * 1e-3 * 2.4e9 / (wavesPerCU / 4.0 * 200000.0 * 8.0));
        }
    }
    return 0;
}
