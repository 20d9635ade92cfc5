// Fused decode + correction: hex text -> records AND corrected rows in one pass over the text.
//
// Replaces, like k_frame.hip and k_correct.hip together, the data path of _read_bytes + _parse_groups
// (serial_hex_to_excel_v3.py:18-36, :41-103) followed by _detect_groups (:43-47), _identify_baselines (:50-61) and
// _filter_group_corrected (:140-178) of bs_beam_correction.py as process_excel_filtered drives them (:186-198).
//
// Why.  With two kernels the decoded rows go to memory and the correction reads them again (mask, UE, RSS, CLK:
// 9.5 bytes a row), because a row's correction needs every baseline of its sweep, later ones included.  Here a
// workgroup of eight waves decodes 16 consecutive framing tiles exactly as k_frame_onepass does (phase A: every wave
// two tiles; records, group-start and baseline bits per tile), keeps the tiles' rows in LDS, and corrects them there
// (phase B), so the rows are written once and never read back.
//
// A sweep (group) does not end where a workgroup ends, so workgroups exchange baselines: after phase A each one
// publishes the baselines of its two border sweeps -- those before its first group start and those after its last --
// with write-through stores and then a header word (how many of each, how many group starts), and reads what its
// neighbours published (backwards until a workgroup with a group start, forwards likewise).  A workgroup only ever
// waits for workgroups that publish without waiting for anybody, so whatever order the hardware dispatches them in,
// some workgroup can always finish; waits are bounded by a clock all the same, and a wait that runs out gives the
// frame up like every other case the kernel does not serve: the frame is flagged and the separate correction
// kernels redo it from the row arrays (which are complete).
//
// Phase B in steps, three block barriers in all: (1) wave 0 gathers the tiles' lists into the workgroup's and publishes
// while waves 1 and 2 poll the neighbours behind and ahead; (2) per group a workgroup touches (the one running into
// it, those starting in it) one wave sorts the baselines in row order (neighbours' before / own / neighbours' after)
// into a table of distinct phases (clk_b mod CYCLE, earliest baseline of each phase) in LDS; (3) every wave takes
// its two tiles: every FLAG = 0 row does a lower-bound search in its group's table and takes the nearer of the two
// circular neighbours (ties: the earlier baseline), as k_rows_finish does -- five 64-row chunks of a tile side by
// side, so the dependent LDS reads of one search overlap with the others' -- packs the kept rows in place and writes
// them to the front of the tile's own entries of the kept arrays (SLAM_LAYOUT_KEPT_TILES).
//
// Served: 0 <= 2*TOL < CYCLE < 2^24, MOD_BASE a power of two (host-side switch, fused_supported); per frame: every
// group of at most 256 baselines, at most 7 group starts per workgroup, at most 16 baselines and 8 group starts per
// tile, a border sweep within 12 workgroups either side.  Everything else: flagged (SLAM_FUSED_GAVE_UP_* says why), redone.

#include "pipeline.hpp"

#include "correct_device.hpp"
#include "frame_device.hpp"

// tuning builds (SLAM_EXTRA_FLAGS=-DFU_EXP=n): leave out parts of phase B to see what each costs; results are wrong then
#ifndef FU_EXP
#define FU_EXP 0
#endif

namespace {

constexpr int FU_WAVES = FUSED_TILES / 2;            // waves per workgroup (two tiles each)
constexpr int FU_TPW = FUSED_TILES / FU_WAVES;       // tiles per wave
constexpr int FU_BLT = 16;                           // baselines listed per tile
constexpr int FU_GST = 8;                            // group starts listed per tile
constexpr int FU_MAXG = FU_WAVES;                    // groups a workgroup corrects: the one running into it + those starting in it (one wave sorts one)
constexpr int FU_NBR = FUSED_TILES >= 16 ? 256 : 192; // baselines taken from the neighbours, per side
constexpr int FU_MAXM = 256;                         // baselines per group
constexpr int FU_POOL = 1024;                        // table entries of all the workgroup's groups together (each padded alike)
constexpr int FU_HOPS = 12;                          // workgroups walked per side
constexpr uint64_t FU_TIMEOUT = 1000000ull;          // ticks of the 100 MHz wall clock a wait may last (10 ms)
constexpr int FU_WBL = FUSED_TILES * FU_BLT;         // 256 own baselines at most
constexpr int FU_CHUNKS = (SLOT_ROWS + 63) / 64;     // 64-row chunks of a tile
constexpr uint32_t FU_PUB_LIST = 127;                // baselines per published list
static_assert(FU_TPW * FU_WAVES == FUSED_TILES && FU_MAXG <= FU_WAVES, "one wave sorts one group");
static_assert(2 * FU_PUB_LIST <= FUSED_PUB_WORDS, "published lists fit their words");
static_assert(FUSED_TILES <= 64 && FU_HOPS < 64, "one lane per tile / per hop");

// Header word a workgroup publishes (also the "published" signal: the words are zeroed before every launch):
// bit 31 set | gave up << 30 | baselines after the last group start << 16 | before the first << 8 | group starts
constexpr uint32_t HDR_VALID = 0x80000000u, HDR_FAIL = 0x40000000u;
__device__ __forceinline__ uint32_t hdr_ng(uint32_t h) { return h & 0xFFu; }
__device__ __forceinline__ uint32_t hdr_head(uint32_t h) { return (h >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t hdr_tail(uint32_t h) { return (h >> 16) & 0xFFu; }

__device__ __forceinline__ unsigned long long ld_pub(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // bypasses this CU's L1
}
__device__ __forceinline__ void st_pub(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // write-through
}
// the header goes out behind everything the wave stored before it
__device__ __forceinline__ void st_hdr(uint32_t *p, uint32_t v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Walk over the neighbours' headers, one lane per hop (lane l looks at the workgroup l + 1 places away in direction
// `dir`, `reach` of them exist): polls until the headers from the nearest one up to the first that closes the sweep
// (a workgroup with a group start, one that gave up, or the last that exists) have all arrived, or the clock runs
// out.  Returns this lane's header (0: not needed or never arrived); *hops = how many workgroups take part (0: no
// neighbour needed), *ok = false when a header did not arrive in time.
__device__ __forceinline__ uint32_t poll_headers(const uint32_t *flag0, int dir, uint32_t reach, uint32_t lane,
                                                 uint32_t *hops, bool *ok)
{
    const uint32_t n = min(reach, (uint32_t)FU_HOPS);
    *hops = 0;
    *ok = true;
    if (n == 0) return 0;
    const uint32_t *mine = flag0 + (ptrdiff_t)dir * (ptrdiff_t)(lane + 1);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t h = 0;
    for (;;) {
        if (lane < n && !(h & HDR_VALID)) h = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t valid = __ballot((h & HDR_VALID) != 0);
        const uint64_t closing = __ballot((h & HDR_VALID) && (hdr_ng(h) != 0 || (h & HDR_FAIL)));
        const uint32_t run = (uint32_t)__builtin_ctzll(~valid);                     // headers in a row from the nearest
        const uint32_t first = closing ? (uint32_t)__builtin_ctzll(closing) : 64u;
        if (first < run) { *hops = first + 1; break; }
        if (run >= n) { *hops = n; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > FU_TIMEOUT) { *ok = false; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return lane < *hops ? h : 0u;
}

template <int MODE>
__global__ __launch_bounds__(64 * FU_WAVES) __attribute__((amdgpu_waves_per_eu(4)))
void k_frame_fused(frame_src src, rows_out out, kept_out_t<uint8_t, uint32_t> kout, uint32_t *stats, fused_ws fw,
                   correct_params prm)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[FU_WAVES][FRAME_TILE_DW];      // phase A: text tiles; phase B: lists, tables
    __shared__ __attribute__((aligned(16))) uint32_t stage[FUSED_TILES][STAGE_DW];      // the tiles' rows, column by column
    __shared__ uint32_t prevrec[FUSED_TILES][4];         // per tile: the record before its first (clk, rss, ue | flag << 8, state)
    __shared__ uint16_t rmaskL[FUSED_TILES][72];         // per 4 entries: group-start | baseline << 4 | row << 8 | FLAG 1 << 12
    __shared__ unsigned long long t_bl[FUSED_TILES][FU_BLT];   // per tile: baselines, clk of the row before | bs << 32 | entry << 40
    __shared__ uint16_t t_gs[FUSED_TILES][FU_GST];       // per tile: entries that start a group
    __shared__ uint32_t t_cnt[FUSED_TILES], t_nbound[FUSED_TILES], t_nbase[FUSED_TILES], t_state[FUSED_TILES];
    __shared__ uint32_t t_gbase[FUSED_TILES], t_bbase[FUSED_TILES];
    __shared__ uint32_t g_start[FU_MAXG + 1], g_bllo[FU_MAXG + 2];
    __shared__ uint32_t tab_nd[FU_MAXG];
    __shared__ uint32_t wave_cnt[FU_WAVES][2];
    __shared__ uint32_t s_fail, s_ng, s_npred, s_nsucc;

    const uint32_t f = blockIdx.y, c = blockIdx.x, wv = threadIdx.x >> 6, lane = lane_id(), tid = threadIdx.x;
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    const src_view sv = resolve_src<MODE>(src, f);
    const uint32_t nt = (sv.nb + SLOT_TILE - 1) / SLOT_TILE;
    const uint32_t nwg = (nt + FUSED_TILES - 1) / FUSED_TILES;
    if (c >= nwg) return;                                             // nobody walks past the frame's last workgroup
    if (st[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS) {
        // the frame goes to the two-pass kernels (asked for by the caller, or flagged by a workgroup of this launch that
        // is already done): nothing to do here, but a neighbour that started earlier may be waiting for this workgroup
        if (tid == 0) st_hdr(fw.flag + (size_t)f * fw.wg_stride + c, HDR_VALID | HDR_FAIL);
        return;
    }
    if (tid == 0) { s_fail = 0; s_npred = 0; s_nsucc = 0; }
    __syncthreads();
    uint32_t *tile = lds[wv];
    uint32_t bad = 0, accv = 0, accd = 0, unsync = 0, giveup = 0;

    // ---- phase A: text -> records of two tiles per wave, as k_frame_onepass does ------------------------------
    for (uint32_t j = 0; j < (uint32_t)FU_TPW; j++) {
        const uint32_t ti = j * FU_WAVES + wv, t = c * FUSED_TILES + ti;
        uint32_t *stg = stage[ti];
        uint8_t *stg8 = reinterpret_cast<uint8_t *>(stg + 2 * SLOT_ROWS);
        if (t >= nt) {
            if (lane == 0) { t_cnt[ti] = 0; t_nbound[ti] = 0; t_nbase[ti] = 0; t_state[ti] = 1; }
            continue;
        }
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
            if (t == 0) x = 0;
            else if (x == UNK) unsync = 1;
        }
        uint32_t ex;
        const uint32_t entry = resolve_entries(m, x, 0u, &ex);
        const uint64_t v = closure(1ull << entry, m);
        uint64_t starts = lane ? (v & m.ok) : 0ull;
        const uint32_t cnt = __popcll(starts);
        const uint32_t inc = wave_incl_scan(cnt);
        accd += lane ? count_disc(v, m) : 0u;
        uint32_t li = inc - cnt;
        const uint32_t lane_byte = lane * FRAME_RUN;
        if (lane == 0) {
            const uint64_t hs = vall & m.ok;
            const uint32_t last = hs ? 63u - (uint32_t)__builtin_clzll(hs) : 0u;
            uint32_t state = t == 0 ? 2u : ((hs && last >= 21u) ? 1u : 0u);
            if (state == 1u) starts = 1ull << last;
            prevrec[ti][3] = state;
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
                prevrec[ti][0] = clk;
                prevrec[ti][1] = rss;
                prevrec[ti][2] = uev | (flg << 8);
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
        {   // the tile's whole slot block, full-width stores
            constexpr uint32_t Q = SLOT_ROWS / 4;
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
        {
            // group starts (UE drops, _detect_groups :43-47) and baselines (FLAG 0 -> 1 with equal RSS inside a group,
            // _identify_baselines :50-61) of the tile's rows, four entries per lane, byte-parallel
            const uint32_t C = (uint32_t)__shfl((int)inc, 63, 64);
            const uint32_t pstate = prevrec[ti][3];
            constexpr uint32_t NG = SLOT_ROWS / 4;
            const uint32_t *s1 = stg + 2 * SLOT_ROWS;             // flag dwords [0, NG), ue [NG, 2NG), bs [2NG, 3NG)
            uint32_t nbound = 0, nbase = 0;
#pragma unroll
            for (int ps = 0; ps < (int)((NG + 63) / 64); ps++) {
                const uint32_t g = lane + 64u * ps;
                uint32_t bnd7 = 0, bas7 = 0, liv7 = 0;
                if (g < NG && C > 4u * g) {
                    const uint32_t nl = min(4u, C - 4u * g);
                    liv7 = nl == 4u ? 0x80808080u : (0x80808080u & ((1u << (8u * nl)) - 1u));
                    const uint32_t flw = s1[g], uew = s1[NG + g];
                    uint32_t pu, pf;
                    if (g > 0) { pu = s1[NG + g - 1] >> 24; pf = s1[g - 1] >> 24; }
                    else if (pstate == 1u) { pu = prevrec[ti][2] & 0xFFu; pf = prevrec[ti][2] >> 8; }
                    else { pu = 0; pf = 0; }
                    const uint32_t pvu = (uew << 8) | pu, pvf = (flw << 8) | pf;
                    bnd7 = ~((uew | 0x80808080u) - pvu) & liv7;
                    uint32_t cand = (flw << 7) & ~(pvf << 7) & ~bnd7 & liv7;
                    if (g == 0) {
                        if (pstate == 2u) { bnd7 |= 0x80u; cand &= ~0x80u; }
                        else if (pstate == 0u) { bnd7 &= ~0x80u; cand &= ~0x80u; }      // settled in phase B from the tile before
                    }
                    if (cand) {
                        const uint4 r4 = reinterpret_cast<const uint4 *>(stg + SLOT_ROWS)[g];
                        const uint32_t prs = g > 0 ? stg[SLOT_ROWS + 4u * g - 1] : prevrec[ti][1];
                        const uint32_t eq = (r4.x == prs ? 0x80u : 0u) | (r4.y == r4.x ? 0x8000u : 0u) |
                                            (r4.z == r4.y ? 0x800000u : 0u) | (r4.w == r4.z ? 0x80000000u : 0u);
                        bas7 = cand & eq;
                    }
                }
                const uint32_t bound = __builtin_amdgcn_udot4(bnd7, 0x08040201u, 0u, false) >> 7;
                const uint32_t base = __builtin_amdgcn_udot4(bas7, 0x08040201u, 0u, false) >> 7;
                const uint32_t live = __builtin_amdgcn_udot4(liv7, 0x08040201u, 0u, false) >> 7;
                const uint32_t flag1 = g < NG ? (__builtin_amdgcn_udot4(s1[g] & 0x01010101u, 0x08040201u, 0u, false) & live) : 0u;
                if (g < NG) rmaskL[ti][g] = (uint16_t)(bound | (base << 4) | (live << 8) | (flag1 << 12));
                const uint32_t cb = __popc(base);                                  // 0, 1 or 2 baselines in four rows
                const uint64_t m1 = __ballot(cb >= 1u), m2 = __ballot(cb >= 2u);
                if (cb) {
                    uint32_t idx = nbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)) +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
                    const uint32_t bsw = s1[2 * NG + g];
                    uint32_t bb = base;
                    while (bb) {
                        const uint32_t k = (uint32_t)__builtin_ctz(bb);
                        bb &= bb - 1;
                        const uint32_t e = 4u * g + k;
                        const uint32_t pck = e > 0 ? stg[e - 1] : prevrec[ti][0];
                        if (idx < (uint32_t)FU_BLT)
                            t_bl[ti][idx] = (unsigned long long)pck | ((unsigned long long)((bsw >> (8u * k)) & 0xFFu) << 32) |
                                            ((unsigned long long)e << 40);
                        idx++;
                    }
                }
                if (__any(bound != 0u)) {                                          // rare: a sweep starts in these 256 entries
                    const uint32_t nbq = __popc(bound);
                    const uint32_t isc = wave_incl_scan(nbq);
                    uint32_t idx = nbound + isc - nbq, bb = bound;
                    while (bb) {
                        const uint32_t k = (uint32_t)__builtin_ctz(bb);
                        bb &= bb - 1;
                        if (idx < (uint32_t)FU_GST) t_gs[ti][idx] = (uint16_t)(4u * g + k);
                        idx++;
                    }
                    nbound += (uint32_t)__shfl((int)isc, 63, 64);
                }
                nbase += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            }
            if (nbase > (uint32_t)FU_BLT || nbound > (uint32_t)FU_GST) giveup = SLAM_FUSED_GAVE_UP_TILE_LISTS;
            if (lane == 0) { t_cnt[ti] = C; t_nbound[ti] = nbound; t_nbase[ti] = nbase; t_state[ti] = C ? pstate : 1u; }
        }
        accv += cnt;
        wave_lds_sync();
    }
    accv = wave_sum(accv);
    accd = wave_sum(accd);
    if (lane == 0) { wave_cnt[wv][0] = accv; wave_cnt[wv][1] = accd; }
    if (bad & BAD_MASK) st[SLAM_STAT_NOT_STRICT] = 1;
    if (unsync) { st[SLAM_STAT_LAYOUT] = SLAM_LAYOUT_TWO_PASS; giveup |= SLAM_FUSED_GAVE_UP_UNSYNC; }
    giveup = (uint32_t)__builtin_amdgcn_readfirstlane((int)giveup);         // wave-uniform bits, and lane 0's own
    if (giveup && lane == 0) atomicOr(&s_fail, giveup);
    __syncthreads();

    // ---- phase B ------------------------------------------------------------------------------------------------
    // the text tiles are done with: their LDS holds the lists and tables from here on
    uint32_t *pool = &lds[0][0];
    unsigned long long *wbl = reinterpret_cast<unsigned long long *>(pool);                 // own baselines, row order
    unsigned long long *pred = reinterpret_cast<unsigned long long *>(pool + 2 * FU_WBL);   // neighbours' before, filled from the back
    unsigned long long *succ = pred + FU_NBR;                                               // neighbours' after
    uint2 *tab = reinterpret_cast<uint2 *>(pool + 2 * FU_WBL + 4 * FU_NBR);                 // tables: (phase, index in group << 8 | beam term)
    static_assert(2 * FU_WBL + 4 * FU_NBR + 2 * FU_POOL <= FU_WAVES * FRAME_TILE_DW, "phase B fits the text tiles' LDS");
    const size_t pix = (size_t)f * fw.wg_stride + c;
    unsigned long long *mypub = fw.pub + pix * FUSED_PUB_WORDS;

    // ---- step 1: wave 0 puts the workgroup's lists together and publishes; waves 1 and 2 fetch the neighbours' ----
    if (wv == 0 && !(FU_EXP & 16)) {
        uint32_t fail = s_fail;
        const bool tl = lane < (uint32_t)FUSED_TILES;
        const uint32_t Cl = tl ? t_cnt[lane] : 0u, stl = tl ? t_state[lane] : 1u;
        const uint64_t rowsm = __ballot(Cl != 0u);
        uint64_t fix = __ballot(Cl != 0u && stl == 0u);
        if (fix) {
            // a tile whose look-behind run held no usable record: its first row's bits from the last row of the tile
            // before (inside the workgroup; the workgroup's own first rows cannot be settled here)
            const uint32_t first = (uint32_t)__builtin_ctzll(rowsm);
            if ((fix >> first) & 1u) { fail |= SLAM_FUSED_GAVE_UP_FIRST_ROW; fix &= ~(1ull << first); }
            while (fix) {                                                   // wave-uniform walk, lane 0 works
                const uint32_t ti = (uint32_t)__builtin_ctzll(fix);
                fix &= fix - 1;
                uint32_t bad_fix = 0;
                if (lane == 0) {
                    uint32_t tp = ti;
                    do { tp--; } while (!t_cnt[tp]);
                    const uint32_t lp = t_cnt[tp] - 1;
                    const uint32_t *sp = stage[tp];
                    const uint8_t *sp8 = reinterpret_cast<const uint8_t *>(sp + 2 * SLOT_ROWS);
                    const uint32_t *sq = stage[ti];
                    const uint8_t *sq8 = reinterpret_cast<const uint8_t *>(sq + 2 * SLOT_ROWS);
                    const bool b = sp8[SLOT_ROWS + lp] > sq8[SLOT_ROWS];
                    const bool bl = !b && sq8[0] == 1 && sp8[lp] == 0 && sq[SLOT_ROWS] == sp[SLOT_ROWS + lp];
                    if (b) {
                        const uint32_t n = t_nbound[ti];
                        if (n >= (uint32_t)FU_GST) bad_fix = 1;
                        else {
                            for (uint32_t k = n; k > 0; k--) t_gs[ti][k] = t_gs[ti][k - 1];
                            t_gs[ti][0] = 0;
                            t_nbound[ti] = n + 1;
                            rmaskL[ti][0] |= 1u;
                        }
                    }
                    if (bl) {
                        const uint32_t n = t_nbase[ti];
                        if (n >= (uint32_t)FU_BLT) bad_fix = 1;
                        else {
                            for (uint32_t k = n; k > 0; k--) t_bl[ti][k] = t_bl[ti][k - 1];
                            t_bl[ti][0] = (unsigned long long)sp[lp] | ((unsigned long long)sq8[2 * SLOT_ROWS] << 32);
                            t_nbase[ti] = n + 1;
                            rmaskL[ti][0] |= 16u;
                        }
                    }
                }
                if (__builtin_amdgcn_readfirstlane((int)bad_fix)) fail |= SLAM_FUSED_GAVE_UP_TILE_LISTS;
                wave_lds_fence();
            }
        }
        // per-tile counts -> bases, totals
        const uint32_t nbnd = tl ? t_nbound[lane] : 0u, nbs = tl ? t_nbase[lane] : 0u;
        const uint32_t packed = nbnd | (nbs << 16), incp = wave_incl_scan(packed), exq = incp - packed;
        if (tl) { t_gbase[lane] = exq & 0xFFFFu; t_bbase[lane] = exq >> 16; }
        const uint32_t tot = (uint32_t)__shfl((int)incp, FUSED_TILES - 1, 64);
        const uint32_t NG = tot & 0xFFFFu, NB = tot >> 16;
        if (NG > (uint32_t)FU_MAXG - 1u || NB > (uint32_t)FU_WBL) fail |= SLAM_FUSED_GAVE_UP_WG_LISTS;
        wave_lds_fence();
        uint32_t nb_head = 0, nb_tail = 0;
        if (!fail) {
            // own group starts and baselines, in row order, entries counted from the workgroup's first
            if (tl)
                for (uint32_t k = 0; k < nbnd; k++) g_start[1 + (exq & 0xFFFFu) + k] = lane * SLOT_ROWS + t_gs[lane][k];
#pragma unroll
            for (int r = 0; r < FUSED_TILES * FU_BLT / 64; r++) {
                const uint32_t idx = lane + 64u * r, ti = idx / FU_BLT, k = idx % FU_BLT;
                if (k < t_nbase[ti]) {
                    const unsigned long long p = t_bl[ti][k];
                    wbl[t_bbase[ti] + k] = (p & 0xFFFFFFFFFFull) | ((unsigned long long)(ti * SLOT_ROWS + (uint32_t)(p >> 40)) << 40);
                }
            }
            if (lane == 0) { g_start[0] = 0; g_bllo[0] = 0; g_bllo[NG + 1] = NB; }
            wave_lds_fence();
            // baselines before each own group start
            for (uint32_t s = 1; s <= NG; s++) {
                const uint32_t gs = g_start[s];
                uint32_t below = 0;
                for (uint32_t i = lane; i < NB; i += 64) below += ((uint32_t)(wbl[i] >> 40) < gs) ? 1u : 0u;
                below = wave_sum(below);
                if (lane == 0) g_bllo[s] = below;
                if (s == 1) nb_head = below;
                if (s == NG) nb_tail = NB - below;
            }
            if (NG == 0) nb_head = NB;
            if (nb_head > FU_PUB_LIST || nb_tail > FU_PUB_LIST) fail |= SLAM_FUSED_GAVE_UP_WG_LISTS;
        }
        // publish: lists, then the header behind this wave's own drain
        if (!fail) {
            for (uint32_t i = lane; i < nb_head; i += 64) st_pub(mypub + i, wbl[i] & 0xFFFFFFFFFFull);
            for (uint32_t i = lane; i < nb_tail; i += 64) st_pub(mypub + nb_head + i, wbl[NB - nb_tail + i] & 0xFFFFFFFFFFull);
        }
        if (lane == 0) {
            st_hdr(fw.flag + pix, fail ? (HDR_VALID | HDR_FAIL) : (HDR_VALID | NG | (nb_head << 8) | (nb_tail << 16)));
            s_ng = NG;
            if (fail) atomicOr(&s_fail, fail);
            uint32_t *wc = fw.wg_cnt + pix * FUSED_CNT_WORDS;
            uint32_t v = 0, d = 0;
            for (int w = 0; w < FU_WAVES; w++) { v += wave_cnt[w][0]; d += wave_cnt[w][1]; }
            wc[0] = v; wc[1] = d; wc[2] = fail ? 0u : NG; wc[3] = fail ? 0u : NB;
        }
    }
    if ((FU_EXP & 16) && tid == 0) s_ng = 0;
    if ((wv == 1 || wv == 2) && !(FU_EXP & 8)) {
        // does the workgroup's first row start a group?  (then no row here belongs to an earlier sweep)
        const uint32_t Cl = lane < (uint32_t)FUSED_TILES ? t_cnt[lane] : 0u;
        const uint64_t rowsm = __ballot(Cl != 0u);
        const bool has_rows = rowsm != 0;
        const bool first_start = !has_rows || (rmaskL[has_rows ? __builtin_ctzll(rowsm) : 0][0] & 1u);
        const bool back = wv == 1;
        if (back ? !first_start : has_rows) {
            uint32_t hops;
            bool ok;
            const uint32_t reach = back ? c : nwg - 1u - c;
            const uint32_t h = poll_headers(fw.flag + pix, back ? -1 : 1, reach, lane, &hops, &ok);
            uint32_t fail = ok ? 0u : SLAM_FUSED_GAVE_UP_TIMEOUT;
            if (__any((h & HDR_FAIL) != 0u)) fail |= SLAM_FUSED_GAVE_UP_NEIGHBOUR;
            // the sweep is closed by a workgroup with a group start, or by the frame's end: not by running out of hops
            const bool closed = __any(lane + 1 == hops && hdr_ng(h) != 0u);
            if (!fail && !closed && hops < reach) fail |= SLAM_FUSED_GAVE_UP_NEIGHBOUR;
            // what each workgroup on the way contributes: behind, the baselines after its last group start (all of them
            // if it has none); ahead, those before its first
            const uint32_t ng = hdr_ng(h), nh = hdr_head(h), ntl = hdr_tail(h);
            const uint32_t n = lane < hops ? (back ? (ng ? ntl : nh) : nh) : 0u;
            const uint32_t so = back && ng ? nh : 0u;
            const uint32_t inc = wave_incl_scan(n), total = (uint32_t)__shfl((int)inc, 63, 64);
            if (total > (uint32_t)FU_NBR) fail |= SLAM_FUSED_GAVE_UP_NEIGHBOUR;
            if (!fail) {
                for (uint32_t d = 0; d < hops; d++) {
                    const uint32_t nd_ = (uint32_t)__shfl((int)n, (int)d, 64), sod = (uint32_t)__shfl((int)so, (int)d, 64);
                    const uint32_t before = (uint32_t)__shfl((int)(inc - n), (int)d, 64);
                    const unsigned long long *src_ = fw.pub + (back ? pix - (d + 1) : pix + (d + 1)) * FUSED_PUB_WORDS + sod;
                    // behind: the nearest workgroup's rows come last; ahead: first
                    unsigned long long *dst = back ? pred + (FU_NBR - before - nd_) : succ + before;
                    for (uint32_t i = lane; i < nd_; i += 64) dst[i] = ld_pub(src_ + i);
                }
            }
            if (lane == 0) {
                if (back) s_npred = fail ? 0u : total; else s_nsucc = fail ? 0u : total;
                if (fail) atomicOr(&s_fail, fail);
            }
        }
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) atomicOr(fw.fail + f, s_fail);
        return;
    }
    const uint32_t NG = s_ng;

    // ---- step 2: per group, baselines in row order -> table of distinct phases (one wave per group) -------------
    // every table is padded to the same power of two above the largest group, so the search runs the same number of
    // steps whatever group a row belongs to
    uint32_t Pmax = 8;
    {
        const uint32_t S = NG + 1;                             // slot 0: the sweep running into the workgroup
        const uint32_t npred = s_npred, nsucc = s_nsucc;
        uint32_t my_m = 0, mmax = 0;
        for (uint32_t s = 0; s < S; s++) {
            uint32_t m = g_bllo[s + 1] - g_bllo[s];
            if (s == 0) m += npred;
            if (s == NG) m += nsucc;
            if (s == wv) my_m = m;
            mmax = max(mmax, m);
        }
        while (Pmax <= mmax && Pmax <= (uint32_t)FU_POOL) Pmax <<= 1;       // bounded whatever the list words hold
        if (mmax > (uint32_t)FU_MAXM || S * Pmax > (uint32_t)FU_POOL) {
            if (tid == 0) atomicOr(fw.fail + f, SLAM_FUSED_GAVE_UP_GROUP_SIZE);
            return;
        }
        if (wv < S && !(FU_EXP & 4)) {
            const uint32_t s = wv, m = my_m, my_off = s * Pmax;
            uint32_t nd = 0;
            if (m) {
                const uint32_t lo = g_bllo[s], own = g_bllo[s + 1] - lo, np = s == 0 ? npred : 0u;
                // element i of the group's list: neighbours' before, own, neighbours' after
                auto elem = [&](uint32_t i) -> unsigned long long {
                    if (i < np) return pred[FU_NBR - np + i];
                    if (i < np + own) return wbl[lo + i - np];
                    return succ[i - np - own];
                };
                uint32_t kv[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t i = lane + 64u * e;
                    if (i < m) {
                        uint32_t q, r;
                        phase_of((int32_t)(uint32_t)elem(i), prm, &q, &r);
                        kv[e] = (r << 8) | i;
                    }
                }
                const int E = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
                if (E == 1) wave_sort_regs<1>(kv, lane);
                else if (E == 2) wave_sort_regs<2>(kv, lane);
                else wave_sort_regs<4>(kv, lane);
                uint32_t run = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (e >= E) break;
                    const uint32_t i = lane + 64u * e;
                    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)kv[e > 0 ? e - 1 : 0], 63);
                    uint32_t left = (uint32_t)__shfl_up((int)kv[e], 1, 64);
                    if (lane == 0) left = before;
                    const bool keep = i < m && (i == 0 || (left >> 8) != (kv[e] >> 8));
                    const uint64_t km = __ballot(keep);
                    if (keep) {
                        const uint32_t o = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                        const uint32_t idx = kv[e] & 255u;
                        const unsigned long long p = elem(idx);
                        uint32_t q, r;
                        phase_of((int32_t)(uint32_t)p, prm, &q, &r);
                        tab[my_off + o] = make_uint2(r, (idx << 8) | (((uint32_t)(p >> 32) - q) & (uint32_t)(prm.mod_base - 1)));
                    }
                    run += (uint32_t)__popcll(km);
                }
                nd = run;
            }
            for (uint32_t i = nd + lane; i < Pmax; i += 64) tab[my_off + i] = make_uint2(0xFFFFFFFFu, 0u);
            if (lane == 0) tab_nd[s] = nd;
        }
    }
    __syncthreads();

    // ---- step 3: every wave, its two tiles' rows against their groups' tables; kept rows packed in place and written
    // to the front of the tile's own entries of the kept arrays ------------------------------------------------------
    if (!(FU_EXP & 2)) {
        const uint32_t cyc = (uint32_t)prm.cycle, tol = (uint32_t)prm.tol, modm = (uint32_t)(prm.mod_base - 1);
        for (uint32_t j = 0; j < (uint32_t)FU_TPW; j++) {
            const uint32_t ti = wv * FU_TPW + j, t = c * FUSED_TILES + ti;
            if (t >= nt) break;
            const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_cnt[ti]);
            uint32_t *stg = stage[ti];
            uint8_t *stg8 = reinterpret_cast<uint8_t *>(stg + 2 * SLOT_ROWS);
            const uint32_t gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_gbase[ti]);
            const uint32_t nbnd = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_nbound[ti]);
            uint32_t clk[FU_CHUNKS], rss[FU_CHUNKS], uev[FU_CHUNKS], tb[FU_CHUNKS], ndv[FU_CHUNKS], r1[FU_CHUNKS], q1[FU_CHUNKS], lo[FU_CHUNKS];
            bool cand[FU_CHUNKS];
#pragma unroll
            for (int k = 0; k < FU_CHUNKS; k++) {
                const uint32_t e = lane + 64u * k, ec = min(e, (uint32_t)SLOT_ROWS - 1u);
                clk[k] = stg[ec];
                rss[k] = stg[SLOT_ROWS + ec];
                uev[k] = stg8[SLOT_ROWS + ec];
                cand[k] = e < C && !((rmaskL[ti][ec >> 2] >> (12u + (ec & 3u))) & 1u);
                uint32_t s = gbase;
                for (uint32_t q = 0; q < nbnd; q++) s += (t_gs[ti][q] <= e) ? 1u : 0u;
                tb[k] = s * Pmax;
                ndv[k] = tab_nd[s];
                phase_of((int32_t)clk[k], prm, &q1[k], &r1[k]);
                lo[k] = 0;
            }
            for (uint32_t h = Pmax >> 1; h; h >>= 1) {
#pragma unroll
                for (int k = 0; k < FU_CHUNKS; k++) lo[k] += tab[tb[k] + lo[k] + h - 1].x < r1[k] ? h : 0u;
            }
            uint64_t km[FU_CHUNKS];
            uint32_t res[FU_CHUNKS];
#pragma unroll
            for (int k = 0; k < FU_CHUNKS; k++) {
                const uint32_t nd = ndv[k], ndm = max(nd, 1u);
                const bool wrap_s = lo[k] >= nd, wrap_p = lo[k] == 0;
                const uint32_t is = wrap_s ? 0u : lo[k], ip = wrap_p ? ndm - 1u : lo[k] - 1u;
                const uint2 es = tab[tb[k] + is], ep = tab[tb[k] + ip];
                const uint32_t rs_ = es.x + (wrap_s ? cyc : 0u) - r1[k], rp_ = r1[k] + (wrap_p ? cyc : 0u) - ep.x;
                const bool pd = rp_ < rs_ || (rp_ == rs_ && (ep.y >> 8) < (es.y >> 8));       // ties: the earlier baseline
                const uint32_t adj = pd ? (wrap_p ? 0xFFFFFFFFu : 0u) : (wrap_s ? 1u : 0u);
                res[k] = (((pd ? ep.y : es.y) & 0xFFu) + q1[k] + adj) & modm;
                km[k] = __ballot(cand[k] && nd != 0u && (pd ? rp_ : rs_) <= tol);
            }
            wave_lds_fence();                  // every row of the tile is in registers: pack in place
            uint32_t kept = 0;
#pragma unroll
            for (int k = 0; k < FU_CHUNKS; k++) {
                if ((km[k] >> lane) & 1u) {
                    const uint32_t o = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(km[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km[k], 0u));
                    stg[o] = clk[k];
                    stg[SLOT_ROWS + o] = rss[k];
                    stg8[SLOT_ROWS + o] = (uint8_t)uev[k];
                    stg8[2 * SLOT_ROWS + o] = (uint8_t)res[k];
                }
                kept += (uint32_t)__popcll(km[k]);
            }
            wave_lds_fence();
            if (!(FU_EXP & 1)) {
                constexpr uint32_t Q = SLOT_ROWS / 4;
                const size_t ob = (size_t)f * out.row_cap + (size_t)t * SLOT_ROWS;
                const uint32_t nq = (kept + 3u) >> 2;
                const uint4 *s4 = reinterpret_cast<const uint4 *>(stg);
                const uint32_t *s1 = stg + 2 * SLOT_ROWS;             // flag dwords [0, Q), ue [Q, 2Q), corrected bs [2Q, 3Q)
                uint4 *gc = reinterpret_cast<uint4 *>(kout.clk + ob), *gr = reinterpret_cast<uint4 *>(kout.rss + ob);
                uint32_t *gu = reinterpret_cast<uint32_t *>(kout.ue + ob), *gb = reinterpret_cast<uint32_t *>(kout.bs + ob);
                if (lane < nq) {
                    gc[lane] = s4[lane];
                    gr[lane] = s4[Q + lane];
                    gu[lane] = s1[Q + lane];
                    gb[lane] = s1[2 * Q + lane];
                }
                if (64u + lane < nq) {
                    gc[64 + lane] = s4[64 + lane];
                    gr[64 + lane] = s4[Q + 64 + lane];
                    gu[64 + lane] = s1[Q + 64 + lane];
                    gb[64 + lane] = s1[2 * Q + 64 + lane];
                }
            }
            if (lane == 0) kout.slot_cnt[(size_t)f * kout.slot_stride + t] = kept;
        }
    }
}

// After the fused kernel (and the two-pass kernels for the frames it flagged), one wave per frame: which frames the
// separate correction kernels have to serve (`redo`), and for the others the frame's counters as the sums of its
// workgroups' counts and the layout bit of its kept rows.  Loose text: voided as k_frame_finish does.
__global__ __launch_bounds__(256) void k_fused_finish(uint32_t F, uint32_t *stats, fused_ws fw, int void_loose)
{
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (f >= F) return;
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    const uint32_t nt = (st[SLAM_STAT_NBYTES] + SLOT_TILE - 1) / SLOT_TILE, nwg = (nt + FUSED_TILES - 1) / FUSED_TILES;
    const bool two = (st[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS) != 0;
    const bool loose = void_loose && st[SLAM_STAT_NOT_STRICT];
    const bool redo = two || loose || fw.fail[f] != 0;
    if (!two) {
        uint32_t v = 0, d = 0, g = 0, b = 0;
        for (uint32_t w = lane; w < nwg; w += 64) {
            const uint32_t *wc = fw.wg_cnt + ((size_t)f * fw.wg_stride + w) * FUSED_CNT_WORDS;
            v += wc[0]; d += wc[1]; g += wc[2]; b += wc[3];
        }
        v = wave_sum(v); d = wave_sum(d); g = wave_sum(g); b = wave_sum(b);
        if (lane == 0) {
            st[SLAM_STAT_VALID] = v;
            st[SLAM_STAT_DISCARDED] = d;
            if (!redo) {
                st[SLAM_STAT_GROUPS] = g;
                st[SLAM_STAT_BASELINES] = b;
                st[SLAM_STAT_LAYOUT] = SLAM_LAYOUT_KEPT_TILES;
            }
        }
    }
    if (lane == 0) {
        fw.redo[f] = redo ? 1u : 0u;
        if (fw.fail[f]) st[SLAM_STAT_LAYOUT] |= fw.fail[f] << SLAM_LAYOUT_FUSED_WHY_SHIFT;      // diagnostic: why the frame was handed on
        if (loose) {
            st[SLAM_STAT_VALID] = 0;
            st[SLAM_STAT_DISCARDED] = 0;
            st[SLAM_STAT_ROW_SLOTS] = 0;
        }
    }
}

}  // namespace

void fused_ws::plan(ws_plan &p, uint32_t F, uint32_t st_max, size_t off[5])
{
    const uint32_t wgs = wgs_for(st_max);
    off[0] = p.add(((size_t)F * wgs + F) * sizeof(uint32_t));              // flags, then the frames' give-up words: one clear
    off[1] = p.add((size_t)F * wgs * FUSED_PUB_WORDS * sizeof(unsigned long long));
    off[2] = p.add((size_t)F * wgs * FUSED_CNT_WORDS * sizeof(uint32_t));
    off[3] = p.add((size_t)F * sizeof(uint32_t));
    off[4] = 0;
}

void fused_ws::bind(uint8_t *base, const size_t off[5], uint32_t st_max)
{
    wg_stride = wgs_for(st_max);
    flag = (uint32_t *)(base + off[0]);
    pub = (unsigned long long *)(base + off[1]);
    wg_cnt = (uint32_t *)(base + off[2]);
    redo = (uint32_t *)(base + off[3]);
}

bool fused_supported(const correct_params &prm)
{
    return prm.cycle > 0 && prm.cycle < (1 << 24) && prm.tol >= 0 && 2 * (int64_t)prm.tol < (int64_t)prm.cycle &&
           prm.mod_base > 0 && prm.mod_base <= 256 && (prm.mod_base & (prm.mod_base - 1)) == 0;
}

int fused_launch(slam_ctx *ctx, int mode, const frame_src &src, uint32_t F, uint32_t st_max, const rows_out &out,
                 const fused_args &fz, uint32_t *stats)
{
    fused_ws fw = fz.fw;
    fw.fail = fw.flag + (size_t)F * fw.wg_stride;
    SLAM_TRY(slam_zero_async(ctx, fw.flag, ((size_t)F * fw.wg_stride + F) * sizeof(uint32_t)));
    const dim3 grid(fused_ws::wgs_for(st_max), F), block(64 * FU_WAVES);
    if (mode == SRC_TEXT3) {
        SLAM_LAUNCH(ctx, "frame_fused", (k_frame_fused<SRC_TEXT3>), grid, block, src, out, fz.kout, stats, fw, fz.prm);
    } else {
        SLAM_LAUNCH(ctx, "frame_fused", (k_frame_fused<SRC_RAW>), grid, block, src, out, fz.kout, stats, fw, fz.prm);
    }
    return SLAM_OK;
}

int fused_finish_launch(slam_ctx *ctx, int mode, uint32_t F, const fused_args &fz, uint32_t *stats)
{
    fused_ws fw = fz.fw;
    fw.fail = fw.flag + (size_t)F * fw.wg_stride;
    SLAM_LAUNCH(ctx, "fused_finish", k_fused_finish, dim3((F + 3) / 4), dim3(256), F, stats, fw, mode == SRC_TEXT3 ? 1 : 0);
    return SLAM_OK;
}
