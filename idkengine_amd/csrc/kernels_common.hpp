// kernels_common.hpp — wave-level helpers shared by the kernels of libidkpt.so.
// Part of the single translation unit idkpt.hip (included there, in this order); see DESIGN.md §4 for the kernel table.
#pragma once

DEV uint32_t wave_grab(uint32_t* counter, uint32_t amount)
{
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, amount);
    return __builtin_amdgcn_readfirstlane(base);
}
DEV void flush_counters(uint64_t* counters, uint32_t nPairs, uint32_t nTris)
{
    // wave reduction then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { nPairs += __shfl_down(nPairs, off); nTris += __shfl_down(nTris, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd((unsigned long long*)&counters[0], (unsigned long long)nPairs); atomicAdd((unsigned long long*)&counters[1], (unsigned long long)nTris); }
}
