// Shared host/device helpers for libslamhip (gfx950 only: 64-lane waves are assumed throughout).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/slamhip.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct slam_timer_slot {
    char name[32];
    uint32_t launches;
    double ms;
};

struct slam_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // device workspace: one allocation, bump-carved per call
    uint8_t *ws = nullptr;
    size_t ws_size = 0;
    size_t ws_used = 0;
    bool ws_pinned = false;     // slam_pin_workspace: calls that would grow (reallocate) the workspace fail instead
    // pinned host staging for small read-backs
    uint32_t *pinned = nullptr;
    // Gaussian blur kernel cached on the device
    double *kern_dev = nullptr;
    double kern_sigma = 0.0;
    int kern_size = 0;
    // batched calls: decode + correction as one fused kernel (slam_set_fused); test hook: every frame through the
    // two-pass framing kernels (environment variable SLAMHIP_FORCE_TWO_PASS=1 when the ctx is created)
    int fused = 0;
    int force_two_pass = 0;
    // optional per-kernel timing
    bool timing = false;
    std::string timing_only;    // "," + names + ",": only these kernels are timed (slam_timing_select); empty: all
    std::vector<slam_timer_slot> timers;
    struct pending_ev { int slot; hipEvent_t a, b; };
    std::vector<pending_ev> pending;
    std::vector<hipEvent_t> ev_pool;
};

int slam_fail(slam_ctx *ctx, int code, const char *fmt, ...);

#define SLAM_HIP(ctx, call)                                                                    \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return slam_fail((ctx), SLAM_E_HIP, "%s failed: %s (%s:%d)", #call,                \
                             hipGetErrorString(e__), __FILE__, __LINE__);                      \
    } while (0)

#define SLAM_TRY(expr)                                                                         \
    do {                                                                                       \
        int r__ = (expr);                                                                      \
        if (r__ != SLAM_OK) return r__;                                                        \
    } while (0)

// workspace carving (256-byte aligned pieces)
int slam_ws_reserve(slam_ctx *ctx, size_t bytes);          // grow to at least `bytes` (may reallocate)
void slam_ws_reset(slam_ctx *ctx);
void *slam_ws_take(slam_ctx *ctx, size_t bytes);           // nullptr when the reservation is too small

struct ws_plan {                                           // first pass: sizes; second pass: pointers
    size_t total = 0;
    size_t add(size_t bytes) {
        size_t off = total;
        total += (bytes + 255) & ~size_t(255);
        return off;
    }
};

// kernel launch with optional event timing
int slam_timer_begin(slam_ctx *ctx, const char *name);
void slam_timer_end(slam_ctx *ctx, int token);

#define SLAM_LAUNCH(ctx, name, kern, grid, block, ...)                                          \
    do {                                                                                       \
        int tk__ = (ctx)->timing ? slam_timer_begin((ctx), name) : -1;                          \
        hipLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, __VA_ARGS__);                   \
        if (tk__ >= 0) slam_timer_end((ctx), tk__);                                            \
        hipError_t le__ = hipGetLastError();                                                   \
        if (le__ != hipSuccess)                                                                \
            return slam_fail((ctx), SLAM_E_HIP, "launch of %s failed: %s", name,               \
                             hipGetErrorString(le__));                                         \
    } while (0)

// Zero `bytes` (a multiple of 4) at a 4-byte aligned device address with a kernel of the library's own on the ctx's
// stream: an ordinary kernel node when the caller captures the stream into a graph.
int slam_zero_async(slam_ctx *ctx, void *dev, size_t bytes);

static inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__

#define WAVE 64

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    uint32_t lo = __shfl((int)(uint32_t)v, src, 64);
    uint32_t hi = __shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up((int)v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor((int)v, d, 64);
    return v;
}

// Block-wide exclusive prefix sum for blocks of up to 1024 threads (16 waves).
// `sh` needs 17 words.  Returns the exclusive prefix; *total receives the block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sh, uint32_t *total)
{
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint32_t x = lane < nw ? sh[lane] : 0;
        uint32_t xi = wave_incl_scan(x);
        if (lane < nw) sh[lane] = xi - x;
        if (lane == nw - 1) sh[16] = xi;
    }
    __syncthreads();
    uint32_t r = inc - v + sh[w];
    *total = sh[16];
    __syncthreads();
    return r;
}

// Python's % for a positive modulus.
__device__ __forceinline__ int32_t pymod(int32_t a, int32_t m)
{
    int32_t r = a % m;
    return r < 0 ? r + m : r;
}

#endif  // __HIPCC__
