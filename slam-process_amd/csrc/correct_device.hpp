// Device-side building blocks of the correction shared by its kernels (k_correct.hip) and the fused decode + correct
// kernel (k_fused.hip): clock phase by multiply-high division, wave-level bitonic sort of dword keys.
#pragma once

#include "pipeline.hpp"

namespace {

// floor(n / cycle) and n mod cycle for n = clk + bias >= 0 (bias is a multiple of cycle, so quotient
// differences and remainders are those of clk itself).  Division by the launch-constant cycle is a
// multiply-high with a host-computed magic number (exact for every 32-bit n).
__device__ __forceinline__ void phase_of(int32_t clk, const correct_params &p, uint32_t *q, uint32_t *r)
{
    const uint32_t n = (uint32_t)clk + p.bias;
    const uint32_t hi = __umulhi(p.div_m, n);
    const uint32_t qq = (hi + ((n - hi) >> p.div_s1)) >> p.div_s2;
    *q = qq;
    *r = n - qq * (uint32_t)p.cycle;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware runs a wave's LDS
// operations in order, so only the compiler has to be told not to move them across this point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Bitonic sort of 64 * E dword keys held E per lane (element i = lane + 64 e sits in kv[e] of lane i % 64): stages
// whose partner is less than 64 elements away exchange across lanes, the others compare registers of one lane.
template <int E>
__device__ __forceinline__ void wave_sort_regs(uint32_t (&kv)[4], uint32_t lane)
{
#pragma unroll
    for (uint32_t k = 2; k <= 64u * E; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const uint32_t je = j >> 6;
#pragma unroll
                for (uint32_t e = 0; e < (uint32_t)E; e++) {
                    if (e & je) continue;
                    const uint32_t a = kv[e], b = kv[e | je], lo = min(a, b), hi = max(a, b);
                    const bool asc = ((64u * e) & k) == 0;
                    kv[e] = asc ? lo : hi;
                    kv[e | je] = asc ? hi : lo;
                }
            } else {
#pragma unroll
                for (uint32_t e = 0; e < (uint32_t)E; e++) {
                    const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ j) << 2), (int)kv[e]);
                    const bool take_min = (((lane + 64u * e) & k) == 0) == ((lane & j) == 0);
                    kv[e] = take_min ? min(kv[e], other) : max(kv[e], other);
                }
            }
        }
    }
}

}  // namespace
