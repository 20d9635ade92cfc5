// Device-side building blocks of the record framing, shared by the framing kernels (k_frame.hip) and the fused
// decode + correct kernel (k_fused.hip): text decode, per-run position masks, orbit closure, entry resolution.
// See k_frame.hip for the method.
#pragma once

#include "pipeline.hpp"

namespace {
struct __attribute__((packed, aligned(4))) u32x4_a4 { uint32_t v[4]; };
struct __attribute__((packed, aligned(4))) u32x2_a4 { uint32_t v[2]; };

struct src_view {
    const uint8_t *p;     // first token character (SRC_TEXT3) or first data byte (SRC_RAW / SRC_DEC)
    uint32_t n_src;       // source bytes from p
    uint32_t nb;          // data bytes
    bool bad;             // text length cannot be a whole number of "XX " tokens
};


template <int MODE>
__device__ __forceinline__ src_view resolve_src(const frame_src &s, uint32_t f)
{
    const frame_desc d = s.fdesc[f];
    src_view v;
    v.bad = false;
    v.nb = d.nb;
    if (MODE == SRC_DEC) {                 // bytes decoded by pass 1 of this launch sequence
        v.p = s.dec + (size_t)f * s.dec_stride;
        v.n_src = d.nb;
    } else {
        v.p = s.base + d.off;
        v.n_src = d.n_src;
    }
    return v;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware runs a wave's LDS
// operations in order, so only the compiler has to be told not to move them across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t nib4(uint32_t w)
{
    const uint32_t a = (w >> 6) & 0x01010101u;          // 1 for letters
    return (w & 0x0F0F0F0Fu) + a + (a << 3);
}

// Bit 7 of every byte of the result is set when that byte of w is not an ASCII hex digit, provided no byte
// of w has its own bit 7 set (a byte >= 0x80 can carry into its neighbour's test; callers also accumulate w
// itself, whose bit 7 then marks the word as bad anyway).  The other bits of the result are noise.
__device__ __forceinline__ uint32_t not_hex4(uint32_t w)
{
    const uint32_t a = (w ^ 0x30303030u) + 0x76767676u;              // bit 7 set: not '0'..'9'
    const uint32_t u = (w | 0x20202020u) ^ 0x60606060u;              // letters a-f / A-F -> 1..6
    const uint32_t b = u + 0x79797979u, c = u + 0x7F7F7F7Fu;         // bit 7 of b: u > 6;  bit 7 clear in c: u == 0
    return a & (b | ~c);
}

__device__ __forceinline__ int hexval(uint32_t c)
{
    if (c - '0' < 10u) return (int)(c - '0');
    uint32_t l = c | 0x20u;
    if (l - 'a' < 6u) return (int)(l - 'a' + 10);
    return -1;
}

// 12 text bytes "HL HL HL HL " (t0 = bytes 0..3, t1 = 4..7, t2 = 8..11) -> 4 data bytes.
// v_perm gathers the four high-nibble characters, the four low-nibble characters and the four
// separators into one dword each; nibble values and the format check are then 4-wide SWAR.
// *bad collects two things: bit 7 of any byte set = a character is not a hex digit (or not ASCII);
// any bit of the low 7 of any byte set through the separator term = a separator is not a space.  The
// separator term is shifted out of the way of the bit-7 noise of not_hex4, so callers test (*bad & BAD_MASK).
constexpr uint32_t BAD_MASK = 0x80808080u;
template <bool VALIDATE>
__device__ __forceinline__ uint32_t decode12(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t *bad)
{
    const uint32_t ha = __builtin_amdgcn_perm(t1, t0, 0x0C060300u);
    const uint32_t h = __builtin_amdgcn_perm(t2, ha, 0x05020100u);
    const uint32_t la = __builtin_amdgcn_perm(t1, t0, 0x0C070401u);
    const uint32_t l = __builtin_amdgcn_perm(t2, la, 0x06020100u);
    if (VALIDATE) {
        const uint32_t sa = __builtin_amdgcn_perm(t1, t0, 0x0C0C0502u);
        const uint32_t sp = __builtin_amdgcn_perm(t2, sa, 0x07040100u);
        // separators: x = sp ^ "    " is zero when right; x + 0x7F.. sets bit 7 of every non-zero byte below 0x81,
        // and a byte >= 0x80 shows in x itself
        const uint32_t x = sp ^ 0x20202020u;
        *bad |= not_hex4(h) | not_hex4(l) | h | l | x | (x + 0x7F7F7F7Fu);
    }
    return (nib4(h) << 4) | nib4(l);
}

// One dword (4 data bytes starting at data index gb) of a frame; zero beyond the end.
template <int MODE, bool VALIDATE>
__device__ __forceinline__ uint32_t load_data_dword(const src_view &sv, uint32_t gb, uint32_t *bad)
{
    if (gb >= sv.nb) return 0;
    if (MODE == SRC_TEXT3) {
        if (gb + 4 < sv.nb) {
            // a 4-byte aligned 16-byte load + funnel shifts by the frame's constant misalignment
            const uint8_t *a = sv.p + 3ull * gb;
            uint32_t s = (uint32_t)((uintptr_t)a & 3u);
            const u32x4_a4 q = *reinterpret_cast<const u32x4_a4 *>(a - s);
            uint32_t t0 = __builtin_amdgcn_alignbyte(q.v[1], q.v[0], s);
            uint32_t t1 = __builtin_amdgcn_alignbyte(q.v[2], q.v[1], s);
            uint32_t t2 = __builtin_amdgcn_alignbyte(q.v[3], q.v[2], s);
            return decode12<VALIDATE>(t0, t1, t2, bad);
        }
        uint32_t val = 0;
        for (uint32_t k = 0; k < 4; k++) {
            uint32_t b = gb + k;
            if (b >= sv.nb) break;
            uint32_t o = 3u * b;
            int h = hexval(sv.p[o]);
            int l = (o + 1 < sv.n_src) ? hexval(sv.p[o + 1]) : -1;
            if (h < 0 || l < 0) { *bad |= BAD_MASK; h = 0; l = 0; }
            if (o + 2 < sv.n_src && sv.p[o + 2] != 0x20) *bad |= BAD_MASK;
            val |= (uint32_t)((h << 4) | l) << (8 * k);
        }
        return val;
    } else {
        const uint8_t *a = sv.p + gb;
        if (gb + 8 <= sv.nb) {
            uint32_t s = (uint32_t)((uintptr_t)a & 3u);
            const u32x2_a4 q = *reinterpret_cast<const u32x2_a4 *>(a - s);
            return __builtin_amdgcn_alignbyte(q.v[1], q.v[0], s);
        }
        uint32_t val = 0;
        for (uint32_t k = 0; k < 4 && gb + k < sv.nb; k++) val |= (uint32_t)a[k] << (8 * k);
        return val;
    }
}

// Fill the wave's LDS tile (FRAME_TILE_DW dwords from data index tile_start).  dec_out (optional)
// receives the tile's own 768 decoded dwords, so a later pass can read bytes instead of text.
template <int MODE, bool VALIDATE>
__device__ __forceinline__ void fill_tile(uint32_t *tile, const src_view &sv, uint32_t tile_start,
                                          uint32_t *bad, uint32_t *dec_out)
{
    const uint32_t lane = lane_id();
    const bool whole = (uint64_t)tile_start + 4ull * FRAME_TILE_DW + 8 < sv.nb;      // no end-of-data case inside
    if (MODE == SRC_TEXT3 && whole) {
        // 12 text bytes per lane per step, contiguous across the wave.  The "« " prefix shifts every token
        // by 3 bytes, so each lane loads the 16 aligned bytes that cover its 12 and funnel-shifts them by the
        // frame's constant misalignment.  All loads are issued before anything is stored: the stores may
        // alias the text as far as the compiler knows, and would otherwise serialise the loads.
        const uint8_t *a0 = sv.p + 3ull * tile_start;
        const uint32_t s = (uint32_t)((uintptr_t)a0 & 3u);
        const uint8_t *base = a0 - s + 12u * lane;
        u32x4_a4 q[12];
#pragma unroll
        for (int it = 0; it < 12; it++) q[it] = *reinterpret_cast<const u32x4_a4 *>(base + 768u * it);
        u32x4_a4 qh;
        qh.v[0] = qh.v[1] = qh.v[2] = qh.v[3] = 0;
        if (lane < 4) qh = *reinterpret_cast<const u32x4_a4 *>(base + 768u * 12);
#pragma unroll
        for (int it = 0; it < 12; it++) {
            uint32_t t0 = __builtin_amdgcn_alignbyte(q[it].v[1], q[it].v[0], s);
            uint32_t t1 = __builtin_amdgcn_alignbyte(q[it].v[2], q[it].v[1], s);
            uint32_t t2 = __builtin_amdgcn_alignbyte(q[it].v[3], q[it].v[2], s);
            uint32_t v = decode12<VALIDATE>(t0, t1, t2, bad);
            tile[lane + 64 * it] = v;
            if (dec_out) dec_out[lane + 64 * it] = v;
        }
        if (lane < 4) {
            uint32_t t0 = __builtin_amdgcn_alignbyte(qh.v[1], qh.v[0], s);
            uint32_t t1 = __builtin_amdgcn_alignbyte(qh.v[2], qh.v[1], s);
            uint32_t t2 = __builtin_amdgcn_alignbyte(qh.v[3], qh.v[2], s);
            uint32_t dummy = 0;
            tile[768 + lane] = decode12<false>(t0, t1, t2, &dummy);           // halo: validated by its own tile
        }
        return;
    }
    if (MODE != SRC_TEXT3 && whole && (((uintptr_t)(sv.p + tile_start)) & 15u) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sv.p + tile_start);
        uint4 *dst = reinterpret_cast<uint4 *>(tile);
#pragma unroll
        for (int it = 0; it < 3; it++) dst[lane + 64 * it] = src[lane + 64 * it];
        if (lane == 0) dst[192] = src[192];
        return;
    }
    for (uint32_t d = lane; d < (uint32_t)FRAME_TILE_DW; d += 64) {
        uint32_t v = load_data_dword<MODE, VALIDATE>(sv, tile_start + 4u * d, bad);
        tile[d] = v;
        if (dec_out && d < (uint32_t)(FRAME_TILE / 4)) dec_out[d] = v;
    }
}

// ---- per-run position masks -----------------------------------------------------------------------
struct run_masks {
    uint64_t ok;      // record starts among the run's 48 positions
    uint64_t cc;      // flag byte is 0xCC (FLAG = 1)
    uint64_t rej;     // flag bytes that are neither record starts nor tail positions (each visit = 1 discard)
    uint64_t tail;    // flag bytes with fewer than 11 bytes left: a visit counts 1 discard and ends the scan
    uint64_t walk;    // positions the scan leaves by +1
    uint64_t endbit;  // where a visited tail position sends the orbit (first position past the data)
};

// One bit per byte of a dword pair -> 8 mask bits.  v_dot4_u32_u8 sums byte products, so with the wanted bit
// isolated (byte value 2^k or 0) and weights 1,2,4,8 / 16,32,64,128 two dot products leave the eight bits of
// dwords (a, b) at bit positions k .. k+7 of the result.
__device__ __forceinline__ uint32_t pair_bits(uint32_t a, uint32_t b, uint32_t bitmask)
{
    return __builtin_amdgcn_udot4(b & bitmask, 0x80402010u, __builtin_amdgcn_udot4(a & bitmask, 0x08040201u, 0u, false), false);
}

// bit 7 of every byte that is NOT a flag byte (0x33 / 0xCC): x ^ 0x33 is then 0x00 or 0xFF, i.e. all eight bits
// equal, i.e. the low seven bits of v ^ (v >> 1) are zero
__device__ __forceinline__ uint32_t not_flag4(uint32_t x)
{
    const uint32_t u = (x ^ (x >> 1) ^ 0x2AAAAAAAu) & 0x7F7F7F7Fu;     // 0x2AAAAAAA = 0x33333333 ^ (0x33333333 >> 1)
    return u + 0x7F7F7F7Fu;
}

// `tile` is the wave's LDS tile; `rem` = data bytes from the run's first position to the end (<= 0: none).
__device__ __forceinline__ run_masks make_masks(const uint32_t *tile, int64_t rem)
{
    const uint4 *t4 = reinterpret_cast<const uint4 *>(tile) + lane_id() * 3;
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 q = t4[i];
        w[4 * i + 0] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
    }
    uint32_t p7[8], p6[8], pn[6];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        p7[j] = pair_bits(w[2 * j], w[2 * j + 1], 0x80808080u);        // 8 bits at 7..14
        p6[j] = pair_bits(w[2 * j], w[2 * j + 1], 0x40404040u);        // 8 bits at 6..13
    }
#pragma unroll
    for (int j = 0; j < 6; j++)                                        // flags matter at the run's own 48 positions
        pn[j] = pair_bits(not_flag4(w[2 * j]), not_flag4(w[2 * j + 1]), 0x80808080u);
    const uint32_t b7l = (p7[0] >> 7) | (p7[1] << 1) | (p7[2] << 9) | (p7[3] << 17);
    const uint32_t b7h = (p7[4] >> 7) | (p7[5] << 1) | (p7[6] << 9) | (p7[7] << 17);
    const uint32_t b6l = (p6[0] >> 6) | (p6[1] << 2) | (p6[2] << 10) | (p6[3] << 18);
    const uint32_t b6h = (p6[4] >> 6) | (p6[5] << 2) | (p6[6] << 10) | (p6[7] << 18);
    const uint32_t nfl = (pn[0] >> 7) | (pn[1] << 1) | (pn[2] << 9) | (pn[3] << 17);
    const uint32_t nfh = (pn[4] >> 7) | (pn[5] << 1);
    const uint64_t b7 = ((uint64_t)b7h << 32) | b7l, b6 = ((uint64_t)b6h << 32) | b6l;
    const uint64_t fl = ~(((uint64_t)nfh << 32) | nfl) & ((1ull << FRAME_RUN) - 1);
    const uint64_t own = (1ull << FRAME_RUN) - 1;
    uint64_t m0 = ~b7 & ~b6, m1 = ~b7 & b6, m2 = b7 & ~b6, m3 = b7 & b6;
    uint64_t c2 = m1 & (m1 >> 1);
    uint64_t c5 = c2 & (c2 >> 2) & (m1 >> 4);          // five CLK tags in a row
    uint64_t r3 = m2 & (m2 >> 1) & (m2 >> 2);          // three RSS tags in a row
    uint64_t cand = fl & own;
    run_masks m;
    // bytes past the end are zero (tag 00), so a start whose record would run past the end fails the tags
    m.ok = cand & (m0 >> 1) & (m3 >> 2) & (c5 >> 3) & (r3 >> 8);
    m.cc = fl & b7;                                     // 0xCC has its top bit set, 0x33 has not
    m.tail = 0;
    m.endbit = 0;
    if (rem < FRAME_RUN + 10 && rem > 0) {
        uint64_t below = (1ull << rem) - 1;            // rem <= 57
        uint64_t from = rem > 10 ? ~((1ull << (rem - 10)) - 1) : ~0ull;
        m.tail = cand & below & from;
        if (m.tail) m.endbit = 1ull << rem;
    }
    m.rej = cand & ~m.ok & ~m.tail;
    m.walk = own & ~m.ok & ~m.tail;
    return m;
}

// All positions (bits 0..58) an orbit entering at the seed positions visits.
// A run of 48 positions holds at most 5 records on one orbit (starts >= 11 apart), so five
// walk-then-jump rounds plus a last walk reach everything.  Tail positions are neither walkable nor
// record starts; if one was visited the orbit continues from the first position past the data.
__device__ __forceinline__ uint64_t closure(uint64_t seed, const run_masks &m)
{
    uint64_t v = seed;
#pragma unroll
    for (int it = 0; it < 5; it++) {
        v |= ((m.walk & v) + m.walk) ^ m.walk;                 // walk +1 through non-record positions
        v |= (v & m.ok) << 11;                                 // jump over each visited record
    }
    v |= ((m.walk & v) + m.walk) ^ m.walk;
    if (v & m.tail) {
        v |= m.endbit;
        v |= ((m.walk & v) + m.walk) ^ m.walk;
    }
    return v;
}

__device__ __forceinline__ uint32_t count_valid(uint64_t v, const run_masks &m) { return __popcll(v & m.ok); }
__device__ __forceinline__ uint32_t count_disc(uint64_t v, const run_masks &m)
{
    return __popcll(v & m.rej) + ((v & m.tail) ? 1u : 0u);
}

constexpr uint32_t UNK = 0xFFu;

// Entry offset of every lane's run given the tile's entry offset for lane 0 (all lanes return).
// x = exit offset when the run is synchronising, else UNK.  Returns this lane's entry offset;
// *exit_out receives this lane's exit offset.
__device__ __forceinline__ uint32_t resolve_entries(const run_masks &m, uint32_t x, uint32_t tile_entry,
                                                    uint32_t *exit_out)
{
    uint32_t r = x;
    uint32_t entry = UNK;
    for (int guard = 0; guard < 66; guard++) {
        uint32_t prev = __shfl_up((int)r, 1, 64);
        if (lane_id() == 0) prev = tile_entry;
        entry = prev;
        if (r == UNK && prev != UNK) {
            uint64_t v = closure(1ull << prev, m);
            r = (uint32_t)__builtin_ctzll(v >> FRAME_RUN);
        }
        if (!__any(r == UNK || entry == UNK)) break;
    }
    *exit_out = r;
    return entry;
}

// ---- single pass: text -> records in the tile's own row slots -----------------------------------------
constexpr int STAGE_DW = 2 * SLOT_ROWS + 3 * (SLOT_ROWS / 4) + 1;      // clk, rss dwords; flag, ue, bs bytes; pad to 16 B
static_assert(STAGE_DW % 4 == 0, "staging areas stay 16-byte aligned");

// The one-pass tile t holds data bytes [t*SLOT_TILE - 48, t*SLOT_TILE + SLOT_TILE + 16): lane 0's run is the
// look-behind run (the bytes before the data are taken as zero for t == 0).
template <int MODE>
__device__ __forceinline__ void fill_slot_tile(uint32_t *tile, const src_view &sv, uint32_t t, uint32_t *bad)
{
    if (t > 0) {
        fill_tile<MODE, true>(tile, sv, t * SLOT_TILE - FRAME_RUN, bad, nullptr);
        return;
    }
    for (uint32_t d = lane_id(); d < (uint32_t)FRAME_TILE_DW; d += 64)
        tile[d] = d < FRAME_RUN / 4 ? 0u : load_data_dword<MODE, true>(sv, 4u * (d - FRAME_RUN / 4), bad);
}

}  // namespace
