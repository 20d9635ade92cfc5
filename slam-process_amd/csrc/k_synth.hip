// Deterministic synthetic capture generator (device twin of slamhip/synth.py: byte-identical output).
//
// A frame of shape B x N is the hex text of B UE sweeps ("rows") of N eleven-byte records in the
// format of the reference's debugDoc captures: "« " then "XX " per byte, no newlines.  Each row starts
// with 7 junk bytes (33 33 33 33 33 2F E2: padding plus a truncated record, as real sweep starts show)
// and one record in 256 has a tag bit flipped, so the resynchronising scan of
// serial_hex_to_excel_v3.py:41-103 is exercised.  Record fields come from splitmix64 of
// (seed, frame, record index); see synth.py for the field rules.

#include "pipeline.hpp"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ void put_hex(uint8_t *p, uint32_t b)
{
    const uint32_t h = b >> 4, l = b & 15u;
    p[0] = (uint8_t)(h < 10 ? '0' + h : 'A' + h - 10);
    p[1] = (uint8_t)(l < 10 ? '0' + l : 'A' + l - 10);
    p[2] = ' ';
}

constexpr uint32_t JUNK = 7;

__global__ __launch_bounds__(256) void k_synth(uint8_t *text, uint64_t frame_stride, uint32_t first_frame,
                                                uint32_t B, uint32_t N, uint64_t seed)
{
    const uint32_t fi = blockIdx.y;
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= (uint64_t)B * N) return;
    const uint32_t b = (uint32_t)(r / N), n = (uint32_t)(r % N);
    const uint64_t frame = first_frame + fi;
    uint8_t *ft = text + frame_stride * fi;
    const uint64_t key = seed ^ (frame << 40);
    const uint64_t h = splitmix64(key ^ r);
    const bool handover = (n % 64u) == 63u;
    const uint32_t ue = (uint32_t)(((uint64_t)n * 64u) / N);
    uint32_t bs, rss;
    if (handover) {
        bs = (b + n / 64u) % 64u;
        rss = 15000u + (uint32_t)((splitmix64(key ^ (r - 1)) >> 16) % 78000u);
    } else {
        bs = n < 63u ? 63u : (b + (((n + 1u) / 64u) * 64u - 1u) / 64u) % 64u;
        rss = 15000u + (uint32_t)((h >> 16) % 78000u);
    }
    const uint32_t base = 1563u + n * 61000u;
    const bool off = ((h >> 40) % 100u) < 40u && !handover && (n % 64u) != 62u;
    const uint32_t clk = off ? base + 2000u + (uint32_t)((h >> 8) % 57000u) : base + (uint32_t)(h % 401u) - 200u;
    uint8_t rec[11];
    rec[0] = handover ? 0xCC : 0x33;
    rec[1] = (uint8_t)ue;
    rec[2] = (uint8_t)(0xC0u | bs);
    for (int j = 0; j < 5; j++) rec[3 + j] = (uint8_t)(0x40u | ((clk >> (6 * j)) & 63u));
    for (int j = 0; j < 3; j++) rec[8 + j] = (uint8_t)(0x80u | ((rss >> (6 * j)) & 63u));
    if (((h >> 52) % 256u) == 0u) rec[(h >> 60) % 11u] ^= 0x40u;
    const uint64_t rowb = JUNK + 11ull * N;
    uint64_t pos = (uint64_t)b * rowb;              // data-byte index of the row's junk
    if (r == 0) { ft[0] = 0xC2; ft[1] = 0xAB; ft[2] = 0x20; }
    if (n == 0) {
        const uint8_t junk[JUNK] = {0x33, 0x33, 0x33, 0x33, 0x33, 0x2F, 0xE2};
        for (uint32_t j = 0; j < JUNK; j++) put_hex(ft + 3 + 3 * (pos + j), junk[j]);
    }
    pos += JUNK + 11ull * n;
    for (int j = 0; j < 11; j++) put_hex(ft + 3 + 3 * (pos + j), rec[j]);
}

// dst <- src with 16-byte accesses, grid-stride (bandwidth probe)
__global__ __launch_bounds__(256) void k_copy16(uint4 *dst, const uint4 *src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// read-only stream: xor of everything, one word written per workgroup (bandwidth probe)
__global__ __launch_bounds__(256) void k_read16(uint32_t *dst, const uint4 *src, size_t n16)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    acc = wave_sum(acc);
    if (lane_id() == 0) dst[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

// read-only streams shaped like the framing kernel's tile loads: every wave reads its own contiguous
// 9216-byte chunk, 12 loads per lane.  mode 1: 16 bytes per lane per load (768 B of 1024 used: 12-byte lane
// stride, overlapping);  mode 2: 12 bytes per lane (dwordx3);  mode 3: 16 bytes per lane, no overlap (12288 B).
struct __attribute__((packed, aligned(4))) p_u4 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) p_u3 { uint32_t v[3]; };
template <int MODE>
__global__ __launch_bounds__(256) void k_read_tiles(uint32_t *dst, const uint8_t *src, size_t n_tiles)
{
    const size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const uint32_t lane = lane_id();
    uint32_t acc = 0;
    if (MODE == 3) {
        const uint8_t *b = src + t * 12288 + 16u * lane;
#pragma unroll
        for (int it = 0; it < 12; it++) { uint4 q = *reinterpret_cast<const uint4 *>(b + 1024u * it); acc ^= q.x ^ q.y ^ q.z ^ q.w; }
    } else if (MODE == 2) {
        const uint8_t *b = src + 3 + t * 9216 + 12u * lane;
#pragma unroll
        for (int it = 0; it < 12; it++) { p_u3 q = *reinterpret_cast<const p_u3 *>(b + 768u * it); acc ^= q.v[0] ^ q.v[1] ^ q.v[2]; }
    } else {
        const uint8_t *b = src + t * 9216 + 12u * lane;
#pragma unroll
        for (int it = 0; it < 12; it++) { p_u4 q = *reinterpret_cast<const p_u4 *>(b + 768u * it); acc ^= q.v[0] ^ q.v[1] ^ q.v[2] ^ q.v[3]; }
    }
    acc = wave_sum(acc);
    if (lane == 0 && acc == 0x12345u) dst[0] = acc;
}

// copy-like probes: read n16 16-byte words, write every RATIO-th of them; NT: 0 plain, 1 non-temporal stores,
// 2 non-temporal loads and stores
template <int RATIO, int NT>
__global__ __launch_bounds__(256) void k_mix16(uint4 *dst, const uint4 *src, size_t n16)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u *s = reinterpret_cast<const v4u *>(src);
    v4u *d = reinterpret_cast<v4u *>(dst);
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * RATIO; i + RATIO <= n16; i += (size_t)gridDim.x * 256 * RATIO) {
        v4u acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < RATIO; r++) {
            v4u v = NT == 2 ? __builtin_nontemporal_load(s + i + r) : s[i + r];
            acc ^= v;
        }
        if (NT) __builtin_nontemporal_store(acc, d + i / RATIO);
        else d[i / RATIO] = acc;
    }
}

}  // namespace

extern "C" int slam_read_probe_dev(slam_ctx *ctx, void *scratch_dev, const void *src_dev, size_t n, int mode)
{
    if (!ctx || !scratch_dev || !src_dev) return SLAM_E_INVALID;
    if (mode >= 10) {         // scratch_dev is a destination of n bytes (modes 10-12) or n/3 bytes (20-22)
        dim3 g(4096), b(256);
        uint4 *d = (uint4 *)scratch_dev;
        const uint4 *sp = (const uint4 *)src_dev;
        const size_t n16 = n / 16;
        switch (mode) {
        case 10: SLAM_LAUNCH(ctx, "mix1_0", (k_mix16<1, 0>), g, b, d, sp, n16); break;
        case 11: SLAM_LAUNCH(ctx, "mix1_1", (k_mix16<1, 1>), g, b, d, sp, n16); break;
        case 12: SLAM_LAUNCH(ctx, "mix1_2", (k_mix16<1, 2>), g, b, d, sp, n16); break;
        case 20: SLAM_LAUNCH(ctx, "mix3_0", (k_mix16<3, 0>), g, b, d, sp, n16); break;
        case 21: SLAM_LAUNCH(ctx, "mix3_1", (k_mix16<3, 1>), g, b, d, sp, n16); break;
        case 22: SLAM_LAUNCH(ctx, "mix3_2", (k_mix16<3, 2>), g, b, d, sp, n16); break;
        default: return slam_fail(ctx, SLAM_E_INVALID, "unknown probe mode");
        }
        return SLAM_OK;
    }
    const size_t per = mode == 3 ? 12288 : 9216;
    const size_t nt = (n - 64) / per;
    dim3 grid((uint32_t)((nt + 3) / 4)), block(256);
    if (mode == 3) { SLAM_LAUNCH(ctx, "read_tiles3", (k_read_tiles<3>), grid, block, (uint32_t *)scratch_dev, (const uint8_t *)src_dev, nt); }
    else if (mode == 2) { SLAM_LAUNCH(ctx, "read_tiles2", (k_read_tiles<2>), grid, block, (uint32_t *)scratch_dev, (const uint8_t *)src_dev, nt); }
    else { SLAM_LAUNCH(ctx, "read_tiles1", (k_read_tiles<1>), grid, block, (uint32_t *)scratch_dev, (const uint8_t *)src_dev, nt); }
    return SLAM_OK;
}

extern "C" size_t slam_synth_text_len(uint32_t B, uint32_t N)
{
    return 3 + 3 * ((size_t)B * (JUNK + 11ull * N));
}

extern "C" int slam_synth_frames_dev(slam_ctx *ctx, uint8_t *text_dev, uint64_t frame_stride, uint32_t first_frame,
                                     uint32_t n_frames, uint32_t B, uint32_t N, uint64_t seed)
{
    if (!ctx) return SLAM_E_INVALID;
    if (!text_dev || B == 0 || N == 0 || N > 16384 || frame_stride < slam_synth_text_len(B, N))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_synth_frames_dev: bad shape or stride (N <= 16384)");
    if (n_frames == 0) return SLAM_OK;
    const uint64_t recs = (uint64_t)B * N;
    SLAM_LAUNCH(ctx, "synth", k_synth, dim3((uint32_t)((recs + 255) / 256), n_frames), dim3(256), text_dev,
                frame_stride, first_frame, B, N, seed);
    return SLAM_OK;
}

extern "C" int slam_copy_probe_dev(slam_ctx *ctx, void *dst_dev, const void *src_dev, size_t n)
{
    if (!ctx) return SLAM_E_INVALID;
    if (!dst_dev || !src_dev || (n & 15)) return slam_fail(ctx, SLAM_E_INVALID, "copy probe needs 16-byte multiples");
    if (dst_dev == src_dev) {       // same pointer: read-only probe (writes 32 KiB of per-workgroup words at dst)
        SLAM_LAUNCH(ctx, "read_probe", k_read16, dim3(2048), dim3(256), (uint32_t *)dst_dev, (const uint4 *)src_dev + 2048, n / 16 - 2048);
        return SLAM_OK;
    }
    SLAM_LAUNCH(ctx, "copy_probe", k_copy16, dim3(2048), dim3(256), (uint4 *)dst_dev, (const uint4 *)src_dev, n / 16);
    return SLAM_OK;
}
