// Record framing: hex text (or raw bytes) -> 11-byte records FLAG UE BS CLK0..4 RSS0..2.
//
// Replaces the data path of _read_bytes + _parse_groups (serial_hex_to_excel_v3.py:18-36, :41-103).
// The reference scan is sequential: at position i a non-flag byte advances by 1 (not counted), a
// flag byte whose ten followers carry the right tag bits yields a record and advances by 11, any
// other flag byte counts as discarded and advances by 1; a flag byte with fewer than 11 bytes
// left counts once and ends the scan.
//
// Parallel form.  The scan visits the orbit of position 0 under next(p) = p+11 (record) or p+1.
// A wave owns a tile of 64 runs x 48 bytes; each lane turns its run into 64-bit position masks
// (flag bytes, record starts, walkable positions) and finds the positions an orbit visits with a
// few carry-propagating 64-bit adds ("flood" through walkable positions, then jump 11 from every
// visited record start, until nothing changes).  A run is "synchronising" when all 11 possible
// entry offsets leave it at the same offset -- true for real captures after at most one record --
// so lanes learn their entry offset from their left neighbour and only the first run of a tile
// depends on the tile's entry offset.  Pass 1 stores, per tile and entry offset, (exit offset,
// records, discards); a per-frame scan picks each tile's entry and output offset; pass 2 emits.
// Non-synchronising data (adversarial byte patterns) takes a slower chained path, same results.
//
// Single-pass form (batched pipeline).  Inside a record no other position can start a record (only byte 1
// has tag 00 and only byte 2 has tag 11, and a flag byte is followed by tags 00, 11), so two scan heads less
// than 11 apart meet within 21 bytes: one 48-byte run before a tile pins the scan position at the tile's
// start whatever happened earlier.  k_frame_onepass gives every wave such a look-behind run (lane 0) plus 63
// runs of its own, needs nothing from other tiles, and writes each tile's records into the tile's own
// block of row slots, so no output offsets have to be scanned either.  A frame with a look-behind run that does
// not pin the position (only possible next to the end of the data) is redone by the two-pass kernels.

#include <stdlib.h>

#include <algorithm>

#include "pipeline.hpp"

#include "frame_device.hpp"

namespace {

// Per-frame source description, prepared once per batch so the tile kernels start with one 16-byte load
// instead of a chain of dependent reads (offset table -> text prefix).
template <int MODE>
__global__ __launch_bounds__(256) void k_frame_prepare(frame_src s, uint32_t F, frame_desc *fd, uint32_t *stats,
                                                        int slotted, int force_two_pass, uint32_t max_len)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const uint8_t *p = s.base + s.off[f];
    // Grids, tile tables and row arrays are sized from the caller's bound: a frame whose length word exceeds it is
    // cut there rather than trusted.
    const uint32_t n = min(s.len[f], max_len);
    frame_desc d;
    bool bad = false;
    if (MODE == SRC_TEXT3) {
        uint32_t pre = 0;
        if (n >= 3 && p[0] == 0xC2 && p[1] == 0xAB && p[2] == 0x20) pre = 3;   // leading "« "
        const uint32_t L = n - pre;
        d.off = s.off[f] + pre;
        d.n_src = L;
        d.nb = (L + 1) / 3;
        bad = (L % 3) == 1;              // a dangling character: not a whole number of "XX " tokens
    } else {
        d.off = s.off[f];
        d.n_src = n;
        d.nb = n;
    }
    fd[f] = d;
    stats[f * SLAM_STAT_WORDS + SLAM_STAT_NBYTES] = d.nb;
    if (bad) stats[f * SLAM_STAT_WORDS + SLAM_STAT_NOT_STRICT] = 1;
    if (slotted) {
        stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS] = (d.nb + SLOT_TILE - 1) / SLOT_TILE * SLOT_ROWS;
        stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] = force_two_pass ? SLAM_LAYOUT_TWO_PASS : 0u;
    }
}


// ---- pass 1: per tile, per entry offset: exit offset, records, discards -------------------------
// Each wave owns its LDS tile and walks FRAME_TPW tiles; waves never wait for each other.
template <int MODE>
__global__ __launch_bounds__(256) void k_frame_summarize(frame_src src, uint32_t nt_max, uint32_t *summ,
                                                          uint32_t *stats, uint8_t *dec, uint64_t dec_stride,
                                                          int gated)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][FRAME_TILE_DW];
    const uint32_t f = blockIdx.y, wv = threadIdx.x >> 6, lane = lane_id();
    if (gated && !(stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS)) return;
    const src_view sv = resolve_src<MODE>(src, f);
    const uint32_t nt = (sv.nb + FRAME_TILE - 1) / FRAME_TILE;
    uint32_t *tile = lds[wv];
    uint32_t bad = 0;
    // grid-stride over groups of 4 * FRAME_TPW tiles: the gated launches use a small grid
    for (uint32_t bx = blockIdx.x; bx * (4 * FRAME_TPW) < nt; bx += gridDim.x)
    for (uint32_t j = 0; j < FRAME_TPW; j++) {
    const uint32_t t = (bx * FRAME_TPW + j) * 4 + wv;
    if (t >= nt) break;
    const uint32_t tile_start = t * FRAME_TILE;
    {
        uint32_t *dec_out = dec ? reinterpret_cast<uint32_t *>(dec + (size_t)f * dec_stride + tile_start) : nullptr;
        fill_tile<MODE, true>(tile, sv, tile_start, &bad, dec_out);
    }
    wave_lds_sync();

    const run_masks m = make_masks(tile, (int64_t)sv.nb - (int64_t)tile_start - (int64_t)lane * FRAME_RUN);
    const uint64_t vall = closure(0x7FFull, m);                  // all 11 entry offsets at once
    const uint64_t exits = vall >> FRAME_RUN;
    const bool sync = __popcll(exits) == 1;
    const uint32_t x = sync ? (uint32_t)__builtin_ctzll(exits) : UNK;
    const uint64_t syncmask = __ballot(sync);
    const uint32_t s = syncmask ? (uint32_t)__builtin_ctzll(syncmask) : 64u;   // first synchronising run

    // lanes 0..10 chain entry offset e = lane through runs 0..s (run s included)
    uint32_t e = lane < 11 ? lane : 0, accv = 0, accd = 0;
    const uint32_t last = s < 64 ? s : 63;
    for (uint32_t q = 0; q <= last; q++) {
        run_masks mj;
        mj.ok = shfl64(m.ok, q); mj.rej = shfl64(m.rej, q); mj.tail = shfl64(m.tail, q);
        mj.walk = shfl64(m.walk, q); mj.endbit = shfl64(m.endbit, q); mj.cc = 0;
        uint64_t v = closure(1ull << e, mj);
        accv += count_valid(v, mj);
        accd += count_disc(v, mj);
        e = (uint32_t)__builtin_ctzll(v >> FRAME_RUN);
    }
    // runs after s do not depend on the tile's entry
    uint32_t restv = 0, restd = 0, tile_exit = e;
    if (s < 63) {
        uint32_t r = lane <= s ? ((lane == s) ? x : 0u) : x;      // lanes before s are irrelevant here
        uint32_t ex;
        // treat run s as lane "0" of the remaining chain: entry of lane s+1 is x_s
        uint32_t entry = UNK;
        for (int guard = 0; guard < 66; guard++) {
            uint32_t prev = __shfl_up((int)r, 1, 64);
            entry = prev;
            if (lane > s && r == UNK && prev != UNK) {
                uint64_t v = closure(1ull << prev, m);
                r = (uint32_t)__builtin_ctzll(v >> FRAME_RUN);
            }
            if (!__any(lane > s && (r == UNK || entry == UNK))) break;
        }
        ex = r;
        uint32_t cv = 0, cd = 0;
        if (lane > s) {
            uint64_t v = closure(1ull << entry, m);
            cv = count_valid(v, m);
            cd = count_disc(v, m);
        }
        restv = wave_sum(cv);
        restd = wave_sum(cd);
        tile_exit = (uint32_t)__shfl((int)ex, 63, 64);
    }
    if (lane < 11) {
        uint32_t exit_e = (s < 63) ? tile_exit : e;
        summ[((size_t)f * nt_max + t) * FRAME_SUMM_WORDS + lane] =
            (exit_e << 28) | ((accv + restv) << 16) | (accd + restd);
    }
    wave_lds_sync();
    }
    if (bad & BAD_MASK) stats[f * SLAM_STAT_WORDS + SLAM_STAT_NOT_STRICT] = 1;
}

// ---- per-frame scan over tiles --------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_frame_scan(const uint32_t *summ, uint32_t nt_max,
                                                      uint8_t *tile_entry, uint32_t *tile_off, uint32_t *stats,
                                                      int gated, uint32_t *slot_cnt, uint32_t slot_stride,
                                                      uint32_t *tile_off_d)
{
    __shared__ uint8_t ent[1025];
    __shared__ uint32_t sh[17];
    __shared__ uint32_t pending;
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    if (gated && !(stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS)) return;
    const uint32_t nb = stats[f * SLAM_STAT_WORDS + SLAM_STAT_NBYTES];
    const uint32_t nt = (nb + FRAME_TILE - 1) / FRAME_TILE;
    uint32_t carry_e = 0, carry_v = 0, carry_d = 0;
    for (uint32_t seg = 0; seg < nt; seg += 1024) {
        const uint32_t t = seg + tid;
        const bool have = t < nt;
        const uint32_t *mp = summ + ((size_t)f * nt_max + t) * FRAME_SUMM_WORDS;
        uint32_t x0 = UNK;
        if (have) {
            x0 = mp[0] >> 28;
            for (int e = 1; e < 11; e++)
                if ((mp[e] >> 28) != x0) x0 = UNK;
        }
        ent[tid + 1] = (uint8_t)x0;
        if (tid == 0) ent[0] = (uint8_t)carry_e;
        __syncthreads();
        for (;;) {                                    // chains through non-synchronising tiles (rare)
            if (tid == 0) pending = 0;
            uint32_t ep = ent[tid], mine = ent[tid + 1];
            __syncthreads();
            if (have && mine == UNK) {
                if (ep != UNK) ent[tid + 1] = (uint8_t)(mp[ep] >> 28);
                else pending = 1;
            }
            __syncthreads();
            if (!pending) break;
            __syncthreads();
        }
        uint32_t e = ent[tid], v = 0, d = 0;
        if (have) {
            uint32_t word = mp[e];
            v = (word >> 16) & 0xFFFu;
            d = word & 0xFFFFu;
            tile_entry[(size_t)f * nt_max + t] = (uint8_t)e;
        }
        uint32_t totv, totd;
        uint32_t ex = block_excl_scan(v, sh, &totv);
        uint32_t exd = block_excl_scan(d, sh, &totd);
        if (have) {
            tile_off[(size_t)f * nt_max + t] = carry_v + ex;
            tile_off_d[(size_t)f * nt_max + t] = carry_d + exd;
        }
        uint32_t cnt = nt - seg < 1024 ? nt - seg : 1024;
        carry_e = ent[cnt];
        carry_v += totv;
        carry_d += totd;
        __syncthreads();
    }
    if (tid == 0) {
        stats[f * SLAM_STAT_WORDS + SLAM_STAT_VALID] = carry_v;
        stats[f * SLAM_STAT_WORDS + SLAM_STAT_DISCARDED] = carry_d;
        stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS] =
            slot_cnt ? (carry_v + SLOT_ROWS - 1) / SLOT_ROWS * SLOT_ROWS : carry_v;
    }
    if (slot_cnt) {       // dense rows described as slots: full tiles, then a partial one
        for (uint32_t t = tid; t < slot_stride; t += 1024) {
            const uint64_t lo = (uint64_t)t * SLOT_ROWS;
            slot_cnt[(size_t)f * slot_stride + t] = lo >= carry_v ? 0u : (uint32_t)min((uint64_t)SLOT_ROWS, carry_v - lo);
        }
    }
}

// ---- pass 2: emit records ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_frame_emit(frame_src src, uint32_t nt_max, const uint8_t *tile_entry,
                                                     const uint32_t *tile_off, rows_out out, const uint32_t *gate)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][FRAME_TILE_DW];
    const uint32_t f = blockIdx.y, wv = threadIdx.x >> 6, lane = lane_id();
    if (gate && !(gate[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS)) return;
    const src_view sv = resolve_src<MODE>(src, f);
    const uint32_t nt = (sv.nb + FRAME_TILE - 1) / FRAME_TILE;
    uint32_t bad = 0;
    uint32_t *tile = lds[wv];
    for (uint32_t bx = blockIdx.x; bx * (4 * FRAME_TPW) < nt; bx += gridDim.x)
    for (uint32_t j = 0; j < FRAME_TPW; j++) {
    const uint32_t t = (bx * FRAME_TPW + j) * 4 + wv;
    if (t >= nt) break;
    const uint32_t tile_start = t * FRAME_TILE;
    fill_tile<MODE, false>(tile, sv, tile_start, &bad, nullptr);
    wave_lds_sync();

    const run_masks m = make_masks(tile, (int64_t)sv.nb - (int64_t)tile_start - (int64_t)lane * FRAME_RUN);
    const uint64_t exits = closure(0x7FFull, m) >> FRAME_RUN;
    const uint32_t x = __popcll(exits) == 1 ? (uint32_t)__builtin_ctzll(exits) : UNK;
    uint32_t ex;
    const uint32_t entry = resolve_entries(m, x, tile_entry[(size_t)f * nt_max + t], &ex);
    const uint64_t v = closure(1ull << entry, m);
    uint64_t starts = v & m.ok;
    const uint32_t cnt = __popcll(starts);
    const uint32_t inc = wave_incl_scan(cnt);
    size_t o = (size_t)f * out.row_cap + tile_off[(size_t)f * nt_max + t] + (inc - cnt);
    const uint32_t lane_byte = lane * FRAME_RUN;
    while (starts) {
        uint32_t p = (uint32_t)__builtin_ctzll(starts);
        starts &= starts - 1;
        uint32_t off = lane_byte + p, sh = off & 3u, wi = off >> 2;
        uint32_t w0 = tile[wi], w1 = tile[wi + 1], w2 = tile[wi + 2], w3 = tile[wi + 3];
        uint32_t d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
        uint32_t d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
        uint32_t d2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
        out.flag[o] = (uint8_t)((m.cc >> p) & 1u);
        out.ue[o] = (uint8_t)((d0 >> 8) & 63u);
        out.bs[o] = (uint8_t)((d0 >> 16) & 63u);
        out.clk[o] = ((d0 >> 24) & 63u) | ((d1 & 63u) << 6) | (((d1 >> 8) & 63u) << 12) |
                     (((d1 >> 16) & 63u) << 18) | (((d1 >> 24) & 63u) << 24);
        out.rss[o] = (d2 & 63u) | (((d2 >> 8) & 63u) << 6) | (((d2 >> 16) & 63u) << 12);
        o++;
    }
    wave_lds_sync();
    }
}

// ---- diagnostics: positions of the flag bytes the scan discards -------------------------------------------
// Same walk as pass 2, emitting instead of records the data-byte index of every discarded flag byte, in scan
// order: the positions `_parse_groups` logs one "丢弃组" line for (serial_hex_to_excel_v3.py:53-58, :62-87).
// The reason and the hex dump of a line follow from the bytes at that index.
template <int MODE>
__global__ __launch_bounds__(256) void k_frame_rejects(frame_src src, uint32_t nt_max, const uint8_t *tile_entry,
                                                        const uint32_t *tile_off_d, uint32_t *pos_out, uint32_t pos_cap)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][FRAME_TILE_DW];
    const uint32_t f = blockIdx.y, wv = threadIdx.x >> 6, lane = lane_id();
    const src_view sv = resolve_src<MODE>(src, f);
    const uint32_t nt = (sv.nb + FRAME_TILE - 1) / FRAME_TILE;
    uint32_t bad = 0;
    uint32_t *tile = lds[wv];
    for (uint32_t j = 0; j < FRAME_TPW; j++) {
        const uint32_t t = (blockIdx.x * FRAME_TPW + j) * 4 + wv;
        if (t >= nt) break;
        const uint32_t tile_start = t * FRAME_TILE;
        fill_tile<MODE, false>(tile, sv, tile_start, &bad, nullptr);
        wave_lds_sync();
        const run_masks m = make_masks(tile, (int64_t)sv.nb - (int64_t)tile_start - (int64_t)lane * FRAME_RUN);
        const uint64_t exits = closure(0x7FFull, m) >> FRAME_RUN;
        const uint32_t x = __popcll(exits) == 1 ? (uint32_t)__builtin_ctzll(exits) : UNK;
        uint32_t ex;
        const uint32_t entry = resolve_entries(m, x, tile_entry[(size_t)f * nt_max + t], &ex);
        const uint64_t v = closure(1ull << entry, m);
        uint64_t rej = v & m.rej;
        const uint64_t tl = v & m.tail;
        if (tl) rej |= tl & (~tl + 1);                    // the scan ends at the first tail position it visits
        const uint32_t cnt = __popcll(rej);
        const uint32_t inc = wave_incl_scan(cnt);
        size_t o = (size_t)f * pos_cap + tile_off_d[(size_t)f * nt_max + t] + (inc - cnt);
        while (rej) {
            const uint32_t p = (uint32_t)__builtin_ctzll(rej);
            rej &= rej - 1;
            pos_out[o++] = tile_start + lane * FRAME_RUN + p;
        }
        wave_lds_sync();
    }
}


#ifndef SLAM_ONEPASS_WAVES
#define SLAM_ONEPASS_WAVES 4
#endif
// Measurement builds only (tools/exp_framing_bound.py): bit 0 leaves the row columns unwritten, bit 1 the row bits,
// to time what the kernel costs without its stores.  The results of such a build are wrong; the default is 0.
#ifndef SLAM_EXP_SKIP
#define SLAM_EXP_SKIP 0
#endif
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SLAM_ONEPASS_WAVES)))      // 4 waves per SIMD (<= 128 VGPRs); 5 spills: 10.0 instead of 7.6 ms
void k_frame_onepass(frame_src src, rows_out out, uint32_t *stats, uint32_t *wg_valid, uint32_t *wg_disc, uint32_t wg_stride)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][FRAME_TILE_DW];
    __shared__ __attribute__((aligned(16))) uint32_t stage[4][STAGE_DW];
    __shared__ uint32_t prevrec[4][4];          // per wave: the record before the tile's first (clk, rss, ue | flag << 8, state)
    const uint32_t f = blockIdx.y, wv = threadIdx.x >> 6, lane = lane_id();
    __shared__ uint32_t wave_cnt[4][2];         // per wave: records, discards (a wave that leaves early leaves zeros)
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    if (lane == 0) { wave_cnt[wv][0] = 0; wave_cnt[wv][1] = 0; }
    if (st[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS) return;
    const src_view sv = resolve_src<MODE>(src, f);
    const uint32_t nt = (sv.nb + SLOT_TILE - 1) / SLOT_TILE;
    uint32_t *tile = lds[wv];
    uint32_t *stg = stage[wv];
    uint8_t *stg8 = reinterpret_cast<uint8_t *>(stg + 2 * SLOT_ROWS);
    uint32_t bad = 0, accv = 0, accd = 0, unsync = 0;
    for (uint32_t j = 0; j < FRAME_TPW; j++) {
        const uint32_t t = (blockIdx.x * FRAME_TPW + j) * 4 + wv;
        if (t >= nt) break;
        fill_slot_tile<MODE>(tile, sv, t, &bad);
        wave_lds_sync();

        run_masks m = make_masks(tile, (int64_t)sv.nb - ((int64_t)t * SLOT_TILE - FRAME_RUN) - (int64_t)lane * FRAME_RUN);
        uint32_t x;
        const uint64_t vall = closure(0x7FFull, m);
        {
            const uint64_t exits = vall >> FRAME_RUN;
            x = __popcll(exits) == 1 ? (uint32_t)__builtin_ctzll(exits) : UNK;
        }
        if (lane == 0) {
            if (t == 0) x = 0;                       // the scan starts at data byte 0
            else if (x == UNK) unsync = 1;
        }
        uint32_t ex;
        const uint32_t entry = resolve_entries(m, x, 0u, &ex);
        const uint64_t v = closure(1ull << entry, m);
        uint64_t starts = lane ? (v & m.ok) : 0ull;  // the look-behind run's records belong to the tile before
        const uint32_t cnt = __popcll(starts);
        const uint32_t inc = wave_incl_scan(cnt);
        accd += lane ? count_disc(v, m) : 0u;
        uint32_t li = inc - cnt;
        const uint32_t lane_byte = lane * FRAME_RUN;
        if (lane == 0) {
            // The record before the tile's first one, for the group / baseline tests: the last record start of
            // the look-behind run.  Orbits from all 11 entry offsets have merged by position 21, so a start at or
            // after it is on the scan's own orbit; an earlier one (or none) is not trusted and the tile's bits
            // are recomputed from the row arrays later.  Tile 0 has nothing before it.
            const uint64_t hs = vall & m.ok;
            const uint32_t last = hs ? 63u - (uint32_t)__builtin_clzll(hs) : 0u;
            uint32_t state = t == 0 ? 2u : ((hs && last >= 21u) ? 1u : 0u);
            if (state == 1u) starts = 1ull << last;
            prevrec[wv][3] = state;
        }
        while (starts) {
            uint32_t p = (uint32_t)__builtin_ctzll(starts);
            starts &= starts - 1;
            uint32_t off = lane_byte + p, sh = off & 3u, wi = off >> 2;
            uint32_t w0 = tile[wi], w1 = tile[wi + 1], w2 = tile[wi + 2], w3 = tile[wi + 3];
            uint32_t d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
            uint32_t d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
            uint32_t d2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
            const uint32_t clk = ((d0 >> 24) & 63u) | ((d1 & 63u) << 6) | (((d1 >> 8) & 63u) << 12) |
                                 (((d1 >> 16) & 63u) << 18) | (((d1 >> 24) & 63u) << 24);
            const uint32_t rss = (d2 & 63u) | (((d2 >> 8) & 63u) << 6) | (((d2 >> 16) & 63u) << 12);
            const uint32_t flg = (uint32_t)((m.cc >> p) & 1u), uev = (d0 >> 8) & 63u;
            if (lane == 0) {
                prevrec[wv][0] = clk;
                prevrec[wv][1] = rss;
                prevrec[wv][2] = uev | (flg << 8);
            } else {
                stg[li] = clk;
                stg[SLOT_ROWS + li] = rss;
                stg8[li] = (uint8_t)flg;
                stg8[SLOT_ROWS + li] = (uint8_t)uev;
                stg8[2 * SLOT_ROWS + li] = (uint8_t)((d0 >> 16) & 63u);
                li++;
            }
        }
        wave_lds_sync();
        // whole slot block of the tile, full-width stores (slots past the tile's count hold leftovers)
        if (!(SLAM_EXP_SKIP & 1) || t == 0xFFFFFFFFu) {
            constexpr uint32_t Q = SLOT_ROWS / 4;                    // 69 16-byte pieces per dword column
            const size_t rb = (size_t)f * out.row_cap + (size_t)t * SLOT_ROWS;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(stg);
            uint4 *gc = reinterpret_cast<uint4 *>(out.clk + rb), *gr = reinterpret_cast<uint4 *>(out.rss + rb);
            uint32_t *gf = reinterpret_cast<uint32_t *>(out.flag + rb), *gu = reinterpret_cast<uint32_t *>(out.ue + rb),
                     *gb = reinterpret_cast<uint32_t *>(out.bs + rb);
            const uint32_t *s1 = stg + 2 * SLOT_ROWS;
            gc[lane] = s4[lane];
            gr[lane] = s4[Q + lane];
            gf[lane] = s1[lane];
            gu[lane] = s1[Q + lane];
            gb[lane] = s1[2 * Q + lane];
            if (lane < Q - 64) {
                gc[64 + lane] = s4[64 + lane];
                gr[64 + lane] = s4[Q + 64 + lane];
                gf[64 + lane] = s1[64 + lane];
                gu[64 + lane] = s1[Q + 64 + lane];
                gb[64 + lane] = s1[2 * Q + 64 + lane];
            }
            if (lane == 63) out.slot_cnt[(size_t)f * out.slot_stride + t] = inc;
        }
        if (out.bits.rmask) {
            // Group starts (UE drops, _detect_groups bs_beam_correction.py:43-47) and baselines (FLAG 0 -> 1 with
            // equal RSS inside a group, _identify_baselines :50-61) of the tile's rows, one entry per lane and
            // pass, from the staged columns; ballots give the bits of 16 four-entry groups at a time.
            const uint32_t C = (uint32_t)__shfl((int)inc, 63, 64);
            const uint32_t pstate = prevrec[wv][3];
            const size_t tix = (size_t)f * out.bits.stride + t;
            uint16_t *rm = out.bits.rmask + (((size_t)f * out.row_cap + (size_t)t * SLOT_ROWS) >> 2);
            unsigned long long *pay = out.bits.pay + tix * BL_PAY_CAP;
            constexpr uint32_t NG = SLOT_ROWS / 4;
            const uint32_t *s1 = stg + 2 * SLOT_ROWS;             // flag dwords [0, NG), ue [NG, 2NG), bs [2NG, 3NG)
            uint32_t nbound = 0, nbase = 0;
#pragma unroll
            for (int ps = 0; ps < (int)((NG + 63) / 64); ps++) {
                // four entries per lane, byte-parallel: bit 7 of byte k speaks for entry 4g + k
                const uint32_t g = lane + 64u * ps;
                uint32_t bnd7 = 0, bas7 = 0, liv7 = 0;
                if (g < NG && C > 4u * g) {
                    const uint32_t nl = min(4u, C - 4u * g);
                    liv7 = nl == 4u ? 0x80808080u : (0x80808080u & ((1u << (8u * nl)) - 1u));
                    const uint32_t flw = s1[g], uew = s1[NG + g];
                    uint32_t pu, pf;
                    if (g > 0) { pu = s1[NG + g - 1] >> 24; pf = s1[g - 1] >> 24; }
                    else if (pstate == 1u) { pu = prevrec[wv][2] & 0xFFu; pf = prevrec[wv][2] >> 8; }
                    else { pu = 0; pf = 0; }
                    const uint32_t pvu = (uew << 8) | pu, pvf = (flw << 8) | pf;
                    bnd7 = ~((uew | 0x80808080u) - pvu) & liv7;                    // previous UE above this one
                    uint32_t cand = (flw << 7) & ~(pvf << 7) & ~bnd7 & liv7;       // FLAG 1 after FLAG 0, not a group start
                    if (g == 0) {
                        if (pstate == 2u) { bnd7 |= 0x80u; cand &= ~0x80u; }       // first row of the frame
                        else if (pstate == 0u) { bnd7 &= ~0x80u; cand &= ~0x80u; } // row before unknown: redone later
                    }
                    if (cand) {
                        const uint4 r4 = reinterpret_cast<const uint4 *>(stg + SLOT_ROWS)[g];
                        const uint32_t prs = g > 0 ? stg[SLOT_ROWS + 4u * g - 1] : prevrec[wv][1];
                        const uint32_t eq = (r4.x == prs ? 0x80u : 0u) | (r4.y == r4.x ? 0x8000u : 0u) |
                                            (r4.z == r4.y ? 0x800000u : 0u) | (r4.w == r4.z ? 0x80000000u : 0u);
                        bas7 = cand & eq;
                    }
                }
                const uint32_t bound = __builtin_amdgcn_udot4(bnd7, 0x08040201u, 0u, false) >> 7;
                const uint32_t base = __builtin_amdgcn_udot4(bas7, 0x08040201u, 0u, false) >> 7;
                const uint32_t live = __builtin_amdgcn_udot4(liv7, 0x08040201u, 0u, false) >> 7;
                // FLAG == 1 bits ride in the top nibble, so the correction need not read the flag column again
                const uint32_t flag1 = g < NG ? (__builtin_amdgcn_udot4(s1[g] & 0x01010101u, 0x08040201u, 0u, false) & live) : 0u;
                if (g < NG && (!(SLAM_EXP_SKIP & 2) || t == 0xFFFFFFFFu)) rm[g] = (uint16_t)(bound | (base << 4) | (live << 8) | (flag1 << 12));
                const uint32_t c = __popc(base);                                   // 0, 1 or 2 baselines in four rows
                const uint64_t m1 = __ballot(c >= 1u), m2 = __ballot(c >= 2u);
                if (c) {
                    uint32_t idx = nbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)) +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
                    const uint32_t bsw = s1[2 * NG + g];
                    uint32_t bb = base;
                    while (bb) {
                        const uint32_t k = (uint32_t)__builtin_ctz(bb);
                        bb &= bb - 1;
                        const uint32_t e = 4u * g + k;
                        const uint32_t pck = e > 0 ? stg[e - 1] : prevrec[wv][0];
                        pay[idx++] = (unsigned long long)pck | ((unsigned long long)((bsw >> (8u * k)) & 0xFFu) << 32) |
                                     ((unsigned long long)e << 40);
                    }
                }
                nbound += __popc(bound);
                nbase += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            }
            nbound = wave_sum(nbound);
            if (lane == 0) {
                out.bits.nbound[tix] = nbound;
                out.bits.nbase[tix] = nbase | ((pstate == 0u && C > 0u) ? TILE_NEEDS_BITS : 0u);
            }
        }
        accv += cnt;
        wave_lds_sync();
    }
    // The workgroup's two counts go to words of its own and k_frame_finish adds them up: with thousands of workgroups
    // per frame, atomics on the frame's counters made them a hot line (1024 x 8192 frames: 7.93 -> 7.59 ms without them).
    accv = wave_sum(accv);
    accd = wave_sum(accd);
    if (lane == 0) { wave_cnt[wv][0] = accv; wave_cnt[wv][1] = accd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        wg_valid[(size_t)f * wg_stride + blockIdx.x] = wave_cnt[0][0] + wave_cnt[1][0] + wave_cnt[2][0] + wave_cnt[3][0];
        wg_disc[(size_t)f * wg_stride + blockIdx.x] = wave_cnt[0][1] + wave_cnt[1][1] + wave_cnt[2][1] + wave_cnt[3][1];
    }
    if (bad & BAD_MASK) st[SLAM_STAT_NOT_STRICT] = 1;
    if (unsync) st[SLAM_STAT_LAYOUT] = SLAM_LAYOUT_TWO_PASS;
}

// After the single pass, one wave per frame: the frame's VALID and DISCARDED are the sums of its workgroups' counts
// (frames the two-pass kernels redid got theirs from k_frame_scan).  A frame whose text is not in the strict capture
// format was decoded from characters that are not what they were taken for: it ends up with no rows at all (no
// entries to walk, both counters zero), so every later stage sees an empty frame and only NOT_STRICT tells the
// caller to send the text through the general tokenizer.
__global__ __launch_bounds__(256) void k_frame_finish(uint32_t F, uint32_t *stats, const uint32_t *wg_valid,
                                                       const uint32_t *wg_disc, uint32_t wg_stride, uint32_t n_wg, int void_loose)
{
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (f >= F) return;
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    if (!(st[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS)) {
        uint32_t v = 0, d = 0;
        for (uint32_t w = lane; w < n_wg; w += 64) {
            v += wg_valid[(size_t)f * wg_stride + w];
            d += wg_disc[(size_t)f * wg_stride + w];
        }
        v = wave_sum(v);
        d = wave_sum(d);
        if (lane == 0) { st[SLAM_STAT_VALID] = v; st[SLAM_STAT_DISCARDED] = d; }
    }
    if (void_loose && lane == 0 && st[SLAM_STAT_NOT_STRICT]) {
        st[SLAM_STAT_VALID] = 0;
        st[SLAM_STAT_DISCARDED] = 0;
        st[SLAM_STAT_ROW_SLOTS] = 0;
    }
}

}  // namespace

uint32_t frame_tiles_for(int mode, uint32_t max_src_len)
{
    uint64_t nb = mode == SRC_TEXT3 ? ((uint64_t)max_src_len + 1) / 3 : max_src_len;
    uint32_t nt = (uint32_t)((nb + FRAME_TILE - 1) / FRAME_TILE);
    return nt ? nt : 1;
}

int frame_rejects(slam_ctx *ctx, int mode, const frame_src &src_in, uint32_t F, const frame_ws &ws, uint32_t nt_max,
                  uint32_t *pos_out, uint32_t pos_cap)
{
    if (F == 0) return SLAM_OK;
    dim3 grid((nt_max + 4 * FRAME_TPW - 1) / (4 * FRAME_TPW), F), block(256);
    frame_src src = src_in;
    src.dec = ws.dec;
    src.dec_stride = ws.dec_stride;
    src.fdesc = ws.fdesc;
    if (mode == SRC_TEXT3) {
        SLAM_LAUNCH(ctx, "frame_rejects", (k_frame_rejects<SRC_DEC>), grid, block, src, nt_max, ws.tile_entry,
                    ws.tile_off_d, pos_out, pos_cap);
    } else {
        SLAM_LAUNCH(ctx, "frame_rejects", (k_frame_rejects<SRC_RAW>), grid, block, src, nt_max, ws.tile_entry,
                    ws.tile_off_d, pos_out, pos_cap);
    }
    return SLAM_OK;
}

uint32_t slot_tiles_for(int mode, uint32_t max_src_len)
{
    uint64_t nb = mode == SRC_TEXT3 ? ((uint64_t)max_src_len + 1) / 3 : max_src_len;
    uint32_t nt = (uint32_t)((nb + SLOT_TILE - 1) / SLOT_TILE);
    return nt ? nt : 1;
}

extern "C" uint32_t slam_row_tiles(uint32_t max_text_len, uint32_t src_is_bytes)
{
    return slot_tiles_for(src_is_bytes ? SRC_RAW : SRC_TEXT3, max_text_len);
}

extern "C" uint32_t slam_kept_blocks(uint32_t row_cap, uint32_t max_text_len, uint32_t src_is_bytes)
{
    return std::max((row_cap + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, slam_row_tiles(max_text_len, src_is_bytes));
}

void frame_ws::plan(ws_plan &p, uint32_t F, uint32_t nt_max, size_t off[6])
{
    off[4] = p.add((size_t)F * sizeof(frame_desc));
    off[5] = p.add((size_t)F * nt_max * sizeof(uint32_t));
    off[0] = p.add((size_t)F * nt_max * FRAME_SUMM_WORDS * sizeof(uint32_t));
    off[1] = p.add((size_t)F * nt_max);
    off[2] = p.add((size_t)F * nt_max * sizeof(uint32_t));
    off[3] = p.add((size_t)F * dec_stride_for(nt_max));
}

void frame_ws::bind(uint8_t *b, const size_t off[6], uint32_t nt_max)
{
    fdesc = (frame_desc *)(b + off[4]);
    tile_off_d = (uint32_t *)(b + off[5]);
    summ = (uint32_t *)(b + off[0]);
    tile_entry = b + off[1];
    tile_off = (uint32_t *)(b + off[2]);
    dec = b + off[3];
    dec_stride = dec_stride_for(nt_max);
}

int frame_decode(slam_ctx *ctx, int mode, const frame_src &src_in, uint32_t F, uint32_t max_src_len,
                 const frame_ws &ws, uint32_t nt_max, const rows_out &out, uint32_t *stats, const fused_args *fused,
                 int force_two_pass)
{
    if (F == 0) return SLAM_OK;
    dim3 grid((nt_max + 4 * FRAME_TPW - 1) / (4 * FRAME_TPW), F), block(256);
    frame_src src = src_in;
    src.dec = ws.dec;
    src.dec_stride = ws.dec_stride;
    src.fdesc = ws.fdesc;
    const int slotted = out.slot_cnt != nullptr;
    const int force = slotted && force_two_pass;                    // test hook of the ctx: every frame takes the two-pass kernels
    if (mode == SRC_TEXT3) {
        SLAM_LAUNCH(ctx, "frame_prepare", (k_frame_prepare<SRC_TEXT3>), dim3((F + 255) / 256), dim3(256), src, F, ws.fdesc,
                    stats, slotted, force, max_src_len);
    } else {
        SLAM_LAUNCH(ctx, "frame_prepare", (k_frame_prepare<SRC_RAW>), dim3((F + 255) / 256), dim3(256), src, F, ws.fdesc,
                    stats, slotted, force, max_src_len);
    }
    if (slotted) {
        const uint32_t st_max = slot_tiles_for(mode, max_src_len);
        if (out.slot_stride < st_max || (uint64_t)out.row_cap < (uint64_t)st_max * SLOT_ROWS)
            return slam_fail(ctx, SLAM_E_CAPACITY, "row arrays too small for the tile-slotted layout");
        dim3 g1((st_max + 4 * FRAME_TPW - 1) / (4 * FRAME_TPW), F);
        // per-workgroup counts: in the words the two-pass kernels use for a frame's per-tile offsets (a frame is served
        // by one or the other; a workgroup spans 16 tiles, so a frame has fewer workgroups than tiles)
        if (g1.x > nt_max) return slam_fail(ctx, SLAM_E_CAPACITY, "framing scratch: fewer tile words than workgroups");
        if (fused) {
            SLAM_TRY(fused_launch(ctx, mode, src, F, st_max, out, *fused, stats));
        } else if (mode == SRC_TEXT3) {
            SLAM_LAUNCH(ctx, "frame_onepass", (k_frame_onepass<SRC_TEXT3>), g1, block, src, out, stats, ws.tile_off, ws.tile_off_d, nt_max);
        } else {
            SLAM_LAUNCH(ctx, "frame_onepass", (k_frame_onepass<SRC_RAW>), g1, block, src, out, stats, ws.tile_off, ws.tile_off_d, nt_max);
        }
    }
    // two passes: every frame (dense layout) or only the frames the single pass flagged (rare: a small grid whose
    // workgroups stride over the tiles keeps the idle launches cheap)
    if (slotted) grid.x = std::min<uint32_t>(grid.x, 8u);
    if (mode == SRC_TEXT3) {
        // pass 1 reads the text once, validates it and leaves the decoded bytes; pass 2 reads bytes
        SLAM_LAUNCH(ctx, "frame_summarize", (k_frame_summarize<SRC_TEXT3>), grid, block, src, nt_max, ws.summ, stats,
                    ws.dec, ws.dec_stride, slotted);
    } else {
        SLAM_LAUNCH(ctx, "frame_summarize", (k_frame_summarize<SRC_RAW>), grid, block, src, nt_max, ws.summ, stats,
                    (uint8_t *)nullptr, (uint64_t)0, slotted);
    }
    SLAM_LAUNCH(ctx, "frame_scan", k_frame_scan, dim3(F), dim3(1024), ws.summ, nt_max, ws.tile_entry,
                ws.tile_off, stats, slotted, out.slot_cnt, out.slot_stride, ws.tile_off_d);
    const uint32_t *gate = slotted ? stats : nullptr;
    if (mode == SRC_TEXT3) {
        SLAM_LAUNCH(ctx, "frame_emit", (k_frame_emit<SRC_DEC>), grid, block, src, nt_max, ws.tile_entry,
                    ws.tile_off, out, gate);
    } else {
        SLAM_LAUNCH(ctx, "frame_emit", (k_frame_emit<SRC_RAW>), grid, block, src, nt_max, ws.tile_entry,
                    ws.tile_off, out, gate);
    }
    if (slotted && fused) {
        SLAM_TRY(fused_finish_launch(ctx, mode, F, *fused, stats));
    } else if (slotted) {      // counters of the single-pass frames; loose text voided (the dense callers read NOT_STRICT and decode again)
        const uint32_t st_max = slot_tiles_for(mode, max_src_len);
        SLAM_LAUNCH(ctx, "frame_finish", k_frame_finish, dim3((F + 3) / 4), dim3(256), F, stats, ws.tile_off, ws.tile_off_d, nt_max,
                    (st_max + 4 * FRAME_TPW - 1) / (4 * FRAME_TPW), mode == SRC_TEXT3 ? 1 : 0);
    }
    return SLAM_OK;
}
