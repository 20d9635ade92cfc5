// BS-beam index correction: groups -> baselines -> nearest-baseline search -> kept rows.
//
// Replaces the data path of bs_beam_correction.py: _detect_groups (:43-47), _identify_baselines
// (:50-61), _filter_group_corrected (:140-178) / _correct_group (:79-116), driven over a whole
// table as process_excel_filtered does (:186-198).
//
// The reference tries every baseline of the row's group: d = clk - clk_b, k = round(d / CYCLE),
// resid = |d - k*CYCLE|, keeps those with resid <= TOL and takes the smallest resid, the earliest
// baseline on ties.  resid is the circular distance between (clk mod CYCLE) and (clk_b mod CYCLE),
// so the winner is one of the two circular neighbours of clk mod CYCLE among the group's distinct
// baseline phases.  Each group's phases are sorted once (ties keep the earliest baseline), then
// every row does one binary search instead of a pass over all baselines.  All integer arithmetic.
//
// Two drivers share the search.  The dense one (slam_correct: int32 columns, rows 0..n-1) works per block of 1024
// rows: k_rows_summarize -> k_tile_scan -> k_rows_groups -> k_bl_sort[_small] -> k_rows_correct -> k_tile_scan ->
// k_rows_emit_kept (ordered compaction).  The batched one works on the tile-slotted records the framing pass
// leaves, with the per-tile bits, counts and baseline payloads it already derived: k_tile_bits (only for what
// the framing pass could not do) -> k_tile_scan -> k_tile_groups -> k_bl_sort[_small] -> k_rows_finish, which
// writes the kept rows into per-block slots, so neither a corrected-BS column nor a second scan is needed.
// k_rows_finish is launched twice in that path: its lean build (register-resident tables only) on frames whose
// groups all have at most 64 baselines, then its full build on what the lean one left; the sort marks the frames.

#include <type_traits>

#include "pipeline.hpp"
#include "correct_device.hpp"

namespace {

constexpr int SORT_CAP = 2048;       // baselines per group sorted in LDS; larger groups use the rank path

template <typename T>
struct vec4_of;
template <>
struct vec4_of<uint8_t> { using type = uchar4; };
template <>
struct vec4_of<uint32_t> { using type = uint4; };
template <>
struct vec4_of<int32_t> { using type = int4; };

template <typename T>
__device__ __forceinline__ void load4(const T *p, T v[4])
{
    using V = typename vec4_of<T>::type;
    V q = *reinterpret_cast<const V *>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}

// The rows of one frame as the row kernels walk them: n entries, of which some may be unused slots
// (slotted layout: the records of framing tile t sit at the front of entries [t*SLOT_ROWS, (t+1)*SLOT_ROWS)).
struct frame_span {
    uint32_t n;                 // entries to walk
    const uint32_t *cnt;        // records per framing tile, or nullptr when every entry is a row
};
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;

template <typename T8, typename T32>
__device__ __forceinline__ frame_span span_of(const rows_in_t<T8, T32> &in, uint32_t f, const uint32_t *stats)
{
    frame_span s;
    s.n = stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS];
    s.cnt = in.slot_cnt ? in.slot_cnt + (size_t)f * in.slot_stride : nullptr;
    return s;
}

// Which of the entries [i0, i0+4) are rows (bit k; always a prefix of the four since i0 % 4 == 0 and
// SLOT_ROWS % 4 == 0), and the index of the last row before i0 (NO_ROW if there is none).
__device__ __forceinline__ uint32_t live4(const frame_span &s, uint32_t i0, uint32_t *prev)
{
    *prev = NO_ROW;
    if (i0 >= s.n) return 0;
    if (!s.cnt) {
        if (i0) *prev = i0 - 1;
        const uint32_t left = s.n - i0;
        return left >= 4 ? 0xFu : (1u << left) - 1u;
    }
    const uint32_t t = i0 / SLOT_ROWS, k0 = i0 - t * SLOT_ROWS, c = s.cnt[t];
    if (k0 > 0) {
        *prev = i0 - 1;                             // a row whenever any of the four is
    } else {
        for (uint32_t tt = t; tt > 0;) {
            --tt;
            const uint32_t cc = s.cnt[tt];
            if (cc) { *prev = tt * SLOT_ROWS + cc - 1; break; }
        }
    }
    if (c <= k0) return 0;
    return c - k0 >= 4 ? 0xFu : (1u << (c - k0)) - 1u;
}

// boundary / baseline bits of the 4 entries [i0, i0+4) of a frame; *live as live4, *prev the row before i0
template <typename T8, typename T32>
__device__ __forceinline__ void row_bits(const rows_in_t<T8, T32> &in, size_t fb, const frame_span &sp, uint32_t i0,
                                         int mode, uint32_t *bound, uint32_t *base, uint32_t *live, uint32_t *prev)
{
    // mode bit 0: one group (only the first row starts a group); bit 1: baselines are given, detect none
    *bound = 0;
    *base = 0;
    *live = live4(sp, i0, prev);
    if (!*live) return;
    const bool one_group = mode & 1;
    if (mode & 2) {
        if (*prev == NO_ROW) *bound = 1;
        return;
    }
    T8 ue[4], fl[4];
    T32 rs[4];
    load4(in.ue + fb + i0, ue);
    load4(in.flag + fb + i0, fl);
    load4(in.rss + fb + i0, rs);
    T8 pue = 0, pfl = 0;
    T32 prs = 0;
    bool first = *prev == NO_ROW;
    if (!first) {
        pue = in.ue[fb + *prev];
        pfl = in.flag[fb + *prev];
        prs = in.rss[fb + *prev];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((*live >> k) & 1u) {
            bool b = first || (!one_group && pue > ue[k]);
            bool bl = !b && fl[k] == 1 && pfl == 0 && rs[k] == prs;
            *bound |= (uint32_t)b << k;
            *base |= (uint32_t)bl << k;
            first = false;
        }
        pue = ue[k]; pfl = fl[k]; prs = rs[k];
    }
}

template <typename T8, typename T32>
__global__ __launch_bounds__(256) void k_rows_summarize(rows_in_t<T8, T32> in, uint32_t rt_max, int one_group,
                                                         uint32_t *rt_nbound, uint32_t *rt_nbase, uint16_t *rmask,
                                                         const uint32_t *stats)
{
    __shared__ uint32_t sh[8];
    const uint32_t f = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const frame_span sp = span_of(in, f, stats);
    uint32_t bound, base, live, prev;
    const uint32_t i0 = tile * ROWS_PER_BLOCK + tid * 4;
    row_bits(in, (size_t)f * in.row_cap, sp, i0, one_group, &bound, &base, &live, &prev);
    // the later row kernels take these bits instead of reading the columns again
    uint32_t flag1 = 0;
    if (live) {
        T8 fl[4];
        load4(in.flag + (size_t)f * in.row_cap + i0, fl);
#pragma unroll
        for (int k = 0; k < 4; k++) flag1 |= (uint32_t)(fl[k] == 1) << k;
        flag1 &= live;
    }
    if (i0 < sp.n) rmask[((size_t)f * in.row_cap + i0) >> 2] = (uint16_t)(bound | (base << 4) | (live << 8) | (flag1 << 12));
    uint32_t cb = wave_sum(__popc(bound)), cl = wave_sum(__popc(base));
    if (lane_id() == 0) { sh[tid >> 6] = cb; sh[4 + (tid >> 6)] = cl; }
    __syncthreads();
    if (tid == 0) {
        rt_nbound[(size_t)f * rt_max + tile] = sh[0] + sh[1] + sh[2] + sh[3];
        rt_nbase[(size_t)f * rt_max + tile] = sh[4] + sh[5] + sh[6] + sh[7];
    }
}

// Per frame: counts per row tile -> exclusive bases in place; totals to stats.
// a/b are scanned together (b may be null).  In group mode also writes the group-table sentinels.
__global__ __launch_bounds__(1024) void k_tile_scan(uint32_t *a, uint32_t *b, uint32_t rt_max, uint32_t *stats,
                                                     int stat_a, int stat_b, int keep_stat_b,
                                                     uint32_t *grp_row, uint32_t *grp_bl, uint32_t grp_stride,
                                                     uint32_t unit, const uint32_t *gate)
{
    __shared__ uint32_t sh[17];
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    if (gate && !gate[f]) return;
    const uint32_t n = stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS];
    const uint32_t nt = (n + unit - 1) / unit;          // counts are per `unit` entries
    uint32_t ca = 0, cb = 0;
    for (uint32_t seg = 0; seg < nt; seg += 1024) {
        uint32_t t = seg + tid;
        uint32_t va = t < nt ? a[(size_t)f * rt_max + t] : 0;
        uint32_t vb = (b && t < nt) ? b[(size_t)f * rt_max + t] : 0;
        uint32_t ta, tb;
        uint32_t ea = block_excl_scan(va, sh, &ta);
        uint32_t eb = block_excl_scan(vb, sh, &tb);
        if (t < nt) {
            a[(size_t)f * rt_max + t] = ca + ea;
            if (b) b[(size_t)f * rt_max + t] = cb + eb;
        }
        ca += ta;
        cb += tb;
    }
    if (tid == 0) {
        stats[f * SLAM_STAT_WORDS + stat_a] = ca;
        uint32_t nbl = cb;
        if (b) {
            if (keep_stat_b) nbl = stats[f * SLAM_STAT_WORDS + stat_b];
            else stats[f * SLAM_STAT_WORDS + stat_b] = cb;
        }
        if (grp_row) {
            // the kernels that follow slot the kept rows per ROWS_PER_BLOCK entries, whatever an earlier call left
            stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] &= ~SLAM_LAYOUT_KEPT_TILES;
            stats[f * SLAM_STAT_WORDS + SLAM_STAT_KEPT] = 0;       // accumulated by the fused correction kernel
            grp_row[(size_t)f * grp_stride + ca] = n;
            grp_bl[(size_t)f * grp_stride + ca] = nbl;
        }
    }
}

// Group table (first row, first baseline of every group), baseline list, optional gid per row.
template <typename T8, typename T32>
__global__ __launch_bounds__(256) void k_rows_groups(rows_in_t<T8, T32> in, uint32_t rt_max,
                                                      const uint32_t *rt_gbase, const uint32_t *rt_bbase,
                                                      correct_ws ws, int32_t *gid_out, const uint32_t *stats)
{
    __shared__ uint32_t sh[17];
    const uint32_t f = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const frame_span sp = span_of(in, f, stats);
    const uint32_t i0 = tile * ROWS_PER_BLOCK + tid * 4;
    if (tile * ROWS_PER_BLOCK >= sp.n) return;
    const size_t fb = (size_t)f * in.row_cap;
    uint32_t bound = 0, base = 0, live = 0, prev = NO_ROW;
    if (i0 < sp.n) {
        const uint32_t mk = ws.rmask[(fb + i0) >> 2];
        bound = mk & 15u;
        base = (mk >> 4) & 15u;
        live = (mk >> 8) & 15u;
        if (base & 1u) live4(sp, i0, &prev);          // a baseline's clock is that of the row before it
    }
    uint32_t tot;
    uint32_t packed = __popc(bound) | (__popc(base) << 16);       // both counts fit 16 bits per block
    uint32_t ex = block_excl_scan(packed, sh, &tot);
    uint32_t g = rt_gbase[(size_t)f * rt_max + tile] + (ex & 0xFFFFu);      // groups started before this thread's rows
    uint32_t j = rt_bbase[(size_t)f * rt_max + tile] + (ex >> 16);
    const size_t gs = (size_t)f * (in.row_cap + 1);
    const size_t bsb = (size_t)f * ws.bl_cap;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t i = i0 + k;
        if (!((live >> k) & 1u)) break;
        if ((bound >> k) & 1u) {
            ws.grp_row[gs + g] = i;
            ws.grp_bl[gs + g] = j;
            g++;
        }
        if ((base >> k) & 1u) {
            ws.bl_clk[bsb + j] = (int32_t)in.clk[fb + (k ? i - 1 : prev)];
            ws.bl_bs[bsb + j] = (int32_t)in.bs[fb + i];
            ws.bl_gid[bsb + j] = g - 1;
            j++;
        }
        if (gid_out) gid_out[fb + i] = (int32_t)(g - 1);
    }
}

// ---- per framing tile: bits from the row arrays, group table and baseline list from the bits ---------------

// The same bits the single-pass framing kernel leaves (tile_bits), computed from the row arrays: for every tile
// of a frame the framing fallback decoded (all == 0: only those, and tiles the framing kernel flagged), or for
// every tile (all == 1).  One wave per tile, one entry per lane and pass.  A wave looks after `span` tiles (64 where
// few need work; fewer where every tile of the frames served does, so that the tiles spread over more waves).
template <typename T8, typename T32>
__global__ __launch_bounds__(256) void k_tile_bits(rows_in_t<T8, T32> in, int all, const uint32_t *stats, const uint32_t *gate,
                                                   uint32_t span)
{
    const uint32_t f = blockIdx.y, lane = lane_id();
    if (gate && !gate[f]) return;
    const frame_span sp = span_of(in, f, stats);
    const uint32_t nt = sp.n / SLOT_ROWS;                           // tiles the frame spans
    // every lane looks at one tile; the wave then works through the tiles that need it, one at a time
    const uint32_t tl = (blockIdx.x * 4 + (threadIdx.x >> 6)) * span + lane;
    const bool whole = all || (stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS);
    const bool mine = lane < span && tl < nt && (whole || (in.bits.nbase[(size_t)f * in.bits.stride + tl] & TILE_NEEDS_BITS));
    uint64_t todo = __ballot(mine);
    const size_t fb = (size_t)f * in.row_cap;
    while (todo) {
        const uint32_t t = tl - lane + (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const size_t tix = (size_t)f * in.bits.stride + t;
        const uint32_t C = sp.cnt[t];
        uint16_t *rm = in.bits.rmask + ((fb + (size_t)t * SLOT_ROWS) >> 2);
        unsigned long long *pay = in.bits.pay + tix * BL_PAY_CAP;
        uint32_t nbound = 0, nbase = 0;
        for (uint32_t ps = 0; ps < (SLOT_ROWS + 63) / 64; ps++) {
            const uint32_t e = lane + 64u * ps, i = t * SLOT_ROWS + e;
            const bool live = e < C;
            bool b = false, bl = false, f1 = false;
            uint32_t pck = 0, bsv = 0;
            if (live) {
                uint32_t prev = i - 1;
                if (e == 0) {                              // last row of the nearest earlier tile that has one
                    prev = NO_ROW;
                    for (uint32_t tt = t; tt > 0;) {
                        --tt;
                        const uint32_t cc = sp.cnt[tt];
                        if (cc) { prev = tt * SLOT_ROWS + cc - 1; break; }
                    }
                }
                const uint32_t uev = (uint32_t)in.ue[fb + i], flg = (uint32_t)in.flag[fb + i];
                const uint32_t rsv = (uint32_t)in.rss[fb + i];
                bsv = (uint32_t)in.bs[fb + i];
                f1 = flg == 1u;
                if (prev == NO_ROW) {
                    b = true;
                } else {
                    const uint32_t pue = (uint32_t)in.ue[fb + prev], pfl = (uint32_t)in.flag[fb + prev];
                    const uint32_t prs = (uint32_t)in.rss[fb + prev];
                    pck = (uint32_t)in.clk[fb + prev];
                    b = pue > uev;
                    bl = !b && flg == 1u && pfl == 0u && rsv == prs;
                }
            }
            const uint64_t bm = __ballot(b), lm = __ballot(bl), vm = __ballot(live);
            const uint64_t fm = __ballot(f1);
            if (bl) {
                const uint32_t idx = nbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                pay[idx] = (unsigned long long)pck | ((unsigned long long)bsv << 32) | ((unsigned long long)e << 40);
            }
            nbound += (uint32_t)__popcll(bm);
            nbase += (uint32_t)__popcll(lm);
            const uint32_t grp = 16u * ps + lane;
            if (lane < 16 && grp < SLOT_ROWS / 4)
                rm[grp] = (uint16_t)(((bm >> (4 * lane)) & 15u) | (((lm >> (4 * lane)) & 15u) << 4) |
                                     (((vm >> (4 * lane)) & 15u) << 8) | (((fm >> (4 * lane)) & 15u) << 12));
        }
        if (lane == 0) {
            in.bits.nbound[tix] = nbound;
            in.bits.nbase[tix] = nbase;
        }
    }
}

// From the per-tile bits and the scanned per-tile bases: the group table (first row and first baseline of every
// group) and the number of groups started before every ROWS_PER_BLOCK-entry block (what k_rows_finish indexes by).
// Every lane looks at one tile; tiles without a group start need at most the block entry, the few with one are
// worked through by the whole wave.
template <typename T8, typename T32>
__global__ __launch_bounds__(256) void k_tile_groups(rows_in_t<T8, T32> in, correct_ws ws, uint32_t rt_max,
                                                      uint32_t *rt_gbase, const uint32_t *stats)
{
    constexpr uint32_t NG = SLOT_ROWS / 4;                  // 69 four-entry groups per tile
    const uint32_t f = blockIdx.y, lane = lane_id();
    if (ws.gate && !ws.gate[f]) return;
    const uint32_t n = stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS];
    const uint32_t nt = n / SLOT_ROWS;
    const uint32_t tl = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    const size_t fb = (size_t)f * in.row_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) ws.rt_nbase[f] = 0;              // "frame has a large group", set by the sort
    const uint32_t *nb = in.bits.nbound + (size_t)f * in.bits.stride;       // exclusive bases after the scan
    const uint32_t *nl = in.bits.nbase + (size_t)f * in.bits.stride;
    bool work = false;
    if (tl < nt) {
        const uint32_t gb = nb[tl];
        const uint32_t gnext = tl + 1 < nt ? nb[tl + 1] : stats[f * SLAM_STAT_WORDS + SLAM_STAT_GROUPS];
        work = gnext > gb;
        if (!work) {                                        // no group starts here: a block that begins in the tile
            const uint32_t lo = tl * SLOT_ROWS;             // has as many groups before it as the tile
            const uint32_t bs = (lo + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK * ROWS_PER_BLOCK;
            if (bs < lo + SLOT_ROWS) rt_gbase[(size_t)f * rt_max + bs / ROWS_PER_BLOCK] = gb;
        }
    }
    uint64_t todo = __ballot(work);
    const size_t gs = (size_t)f * (in.row_cap + 1);
    while (todo) {
        const uint32_t t = tl - lane + (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t gb = nb[t], bb = nl[t];
        const uint16_t *rm = in.bits.rmask + ((fb + (size_t)t * SLOT_ROWS) >> 2);
        uint32_t cg = 0, cbl = 0;                           // groups / baselines of the tile before this pass
        for (uint32_t ps = 0; ps < (NG + 63) / 64; ps++) {
            const uint32_t g = lane + 64u * ps;
            uint32_t bound = 0, base = 0;
            if (g < NG) {
                const uint32_t mk = rm[g];
                bound = mk & 15u;
                base = (mk >> 4) & 15u;
            }
            const uint32_t packed = __popc(bound) | (__popc(base) << 16);
            const uint32_t inc = wave_incl_scan(packed);
            uint32_t eg = cg + ((inc - packed) & 0xFFFFu), eb = cbl + ((inc - packed) >> 16);
            if (g < NG) {
                const uint32_t i0 = t * SLOT_ROWS + 4u * g;
                if ((i0 % ROWS_PER_BLOCK) == 0) rt_gbase[(size_t)f * rt_max + i0 / ROWS_PER_BLOCK] = gb + eg;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if ((bound >> k) & 1u) {
                        ws.grp_row[gs + gb + eg] = i0 + k;
                        ws.grp_bl[gs + gb + eg] = bb + eb;  // a group's first row is never a baseline
                        eg++;
                    }
                    if ((base >> k) & 1u) eb++;
                }
            }
            const uint32_t tot = (uint32_t)__shfl((int)inc, 63, 64);
            cg += tot & 0xFFFFu;
            cbl += tot >> 16;
        }
    }
}

// Baseline j (row order) of frame f out of the per-tile lists; T0..T1 are the tiles of the rows of its group.
__device__ __forceinline__ unsigned long long baseline_from_tiles(const tile_bits &bits, uint32_t f, uint32_t j,
                                                                  uint32_t T0, uint32_t T1)
{
    const uint32_t *bbase = bits.nbase + (size_t)f * bits.stride;        // baselines before each tile
    uint32_t lo = T0, hi = T1;                                            // last tile whose base is <= j
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (bbase[mid] <= j) lo = mid; else hi = mid - 1;
    }
    return bits.pay[((size_t)f * bits.stride + lo) * BL_PAY_CAP + (j - bbase[lo])];
}

constexpr uint32_t WAVE_SORT_MAX = 64;      // groups of up to this many baselines are sorted by one wave in registers
constexpr uint32_t MID_SORT_MAX = 256;      // ... and up to this many by one wave in its own piece of LDS
constexpr uint32_t MID_TAB_CAP = 512;       // phase tables below this length fit the LDS of k_rows_finish's mid build

// Groups with at most 64 baselines, one wave per group: bitonic sort of (phase, index) across the lanes with lane
// exchanges, no LDS and no barriers.  Groups of 65 .. 256 baselines: still one wave per group, the network runs
// over the wave's own piece of LDS (two or four elements per lane and stage), again without block barriers.
// Same outputs as k_bl_sort, which skips these groups when told so.
__global__ __launch_bounds__(256) void k_bl_sort_small(correct_ws ws, uint32_t row_cap, correct_params prm,
                                                        const uint32_t *stats, tile_bits bits)
{
    __shared__ unsigned long long mid_keys[4][MID_SORT_MAX];
    __shared__ uint32_t mid_q[4][MID_SORT_MAX], mid_bs[4][MID_SORT_MAX];
    __shared__ uint32_t wg_flags;                        // the frame's flag bits this workgroup found, one atomic at the end
    const uint32_t f = blockIdx.y, lane = lane_id();
    if (ws.gate && !ws.gate[f]) return;
    const uint32_t G = stats[f * SLAM_STAT_WORDS + SLAM_STAT_GROUPS];
    const bool wide_cycle = (uint32_t)prm.cycle >= (1u << 24);          // phase and index do not fit one dword
    uint32_t my_flags = 0;
    if (threadIdx.x == 0) wg_flags = 0;
    __syncthreads();
    const size_t gs = (size_t)f * (row_cap + 1);
    const size_t bsb = (size_t)f * ws.bl_cap;
    for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < G; g += gridDim.x * 4) {
        const uint32_t b0 = ws.grp_bl[gs + g], m = ws.grp_bl[gs + g + 1] - b0;
        if (m == 0) {
            if (lane == 0) ws.grp_nd[gs + g] = 0;
            continue;
        }
        // in the batched path rt_nbase is free: its entry f tells k_rows_finish that frame f has a group whose phase
        // table may not fit a wave's registers (bit 0) and k_bl_sort that one is left for it to sort (bit 2)
        // bit 3: a table the mid build of k_rows_finish cannot hold.  Gathered per workgroup: a frame of a thousand such
        // groups would otherwise send a thousand atomics to one word
        if (m > WAVE_SORT_MAX && bits.pay) my_flags |= (m > MID_SORT_MAX ? 5u : 1u) | (m >= MID_TAB_CAP ? 8u : 0u);
        if (m > MID_SORT_MAX) continue;
        if (m > WAVE_SORT_MAX) {
            unsigned long long *kk = mid_keys[threadIdx.x >> 6];
            uint32_t *sq = mid_q[threadIdx.x >> 6], *sb = mid_bs[threadIdx.x >> 6];
            const uint32_t P = m <= 128 ? 128u : 256u;
            uint32_t T0 = 0, T1 = 0;
            if (bits.pay) { T0 = ws.grp_row[gs + g] / SLOT_ROWS; T1 = (ws.grp_row[gs + g + 1] - 1) / SLOT_ROWS; }
            // which tile's list holds baseline b0 + lane + 64 e: with the tiles' bases in one register per lane the
            // owner is found by lane broadcasts, as below, instead of a chain of dependent loads per baseline
            const bool by_lanes = bits.pay && T1 - T0 < 64;
            uint32_t own[4] = {0, 0, 0, 0}, obase[4] = {0, 0, 0, 0};
            uint32_t kv[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};      // dword keys (phase << 8 | index)
            if (by_lanes) {
                const uint32_t ntile = T1 - T0 + 1;
                const uint32_t mine = lane < ntile ? bits.nbase[(size_t)f * bits.stride + T0 + lane] : 0xFFFFFFFFu;
                const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)mine, 0);
#pragma unroll
                for (int e = 0; e < 4; e++) obase[e] = first;
                for (uint32_t t = 1; t < ntile; t++) {
                    const uint32_t bt = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)t);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (bt <= b0 + lane + 64u * e) { own[e] = t; obase[e] = bt; }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t i = lane + 64u * e;
                if (i >= P) break;
                unsigned long long key = ~0ull;
                if (i < m) {
                    int32_t clk;
                    uint32_t bsv, q, r;
                    if (bits.pay) {
                        const unsigned long long p = by_lanes
                            ? bits.pay[((size_t)f * bits.stride + T0 + own[e]) * BL_PAY_CAP + (b0 + i - obase[e])]
                            : baseline_from_tiles(bits, f, b0 + i, T0, T1);
                        clk = (int32_t)(uint32_t)p;
                        bsv = (uint32_t)((p >> 32) & 0xFFu);
                    } else {
                        clk = ws.bl_clk[bsb + b0 + i];
                        bsv = (uint32_t)ws.bl_bs[bsb + b0 + i];
                    }
                    phase_of(clk, prm, &q, &r);
                    key = ((unsigned long long)r << 32) | i;
                    sq[i] = q;
                    sb[i] = bsv;
                    kv[e] = (r << 8) | i;
                }
                if (wide_cycle) kk[i] = key;
            }
            wave_lds_fence();
            if (!wide_cycle) {
                // phase and index fit one dword: the whole network runs in registers and lane exchanges
                if (P == 128) wave_sort_regs<2>(kv, lane); else wave_sort_regs<4>(kv, lane);
                uint32_t run = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t i = lane + 64u * e;
                    if (64u * e >= P) break;
                    // the element before: the lane to the left, or lane 63 of the register before for lane 0
                    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)kv[e > 0 ? e - 1 : 0], 63);
                    uint32_t left = (uint32_t)__shfl_up((int)kv[e], 1, 64);
                    if (lane == 0) left = before;
                    const bool keep = i < m && (i == 0 || (left >> 8) != (kv[e] >> 8));
                    const uint64_t km = __ballot(keep);
                    if (keep) {
                        const uint32_t o = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                        const uint32_t idx = kv[e] & 255u;
                        ws.srt_rb[bsb + b0 + o] = kv[e] >> 8;
                        ws.srt_pay[bsb + b0 + o] = make_uint4(idx, sq[idx], sb[idx], 0);
                    }
                    run += (uint32_t)__popcll(km);
                }
                if (lane == 0) ws.grp_nd[gs + g] = run;
                wave_lds_fence();
                continue;
            }
            for (uint32_t k = 2; k <= P; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t p = lane; p < P / 2; p += 64) {         // pair p: element i and its partner i | j
                        const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                        const unsigned long long a = kk[i], b = kk[l];
                        if ((a > b) == ((i & k) == 0)) { kk[i] = b; kk[l] = a; }
                    }
                    wave_lds_fence();
                }
            }
            uint32_t run = 0;                                  // distinct phases written so far
            for (uint32_t seg = 0; seg < P; seg += 64) {
                const uint32_t i = seg + lane;
                bool keep = false;
                unsigned long long key = 0;
                if (i < m) {
                    key = kk[i];
                    keep = (i == 0) || ((uint32_t)(kk[i - 1] >> 32) != (uint32_t)(key >> 32));
                }
                const uint64_t km = __ballot(keep);
                if (keep) {
                    const uint32_t o = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                    const uint32_t idx = (uint32_t)key;
                    ws.srt_rb[bsb + b0 + o] = (uint32_t)(key >> 32);
                    ws.srt_pay[bsb + b0 + o] = make_uint4(idx, sq[idx], sb[idx], 0);
                }
                run += (uint32_t)__popcll(km);
            }
            if (lane == 0) ws.grp_nd[gs + g] = run;
            wave_lds_fence();                                  // the next group reuses the wave's piece of LDS
            continue;
        }
        uint32_t q = 0, r = 0, bsv = 0;
        unsigned long long key = ~0ull;
        unsigned long long pay = 0;
        if (bits.pay) {
            // baseline b0 + lane out of the per-tile lists: the tiles of the group's rows each hold a few; with their
            // bases in one register per lane the owning tile is a count over lane broadcasts, no dependent loads
            const uint32_t T0 = ws.grp_row[gs + g] / SLOT_ROWS, T1 = (ws.grp_row[gs + g + 1] - 1) / SLOT_ROWS;
            const uint32_t ntile = T1 - T0 + 1;
            if (ntile <= 64) {
                const uint32_t *bbase = bits.nbase + (size_t)f * bits.stride;
                const uint32_t mine = lane < ntile ? bbase[T0 + lane] : 0xFFFFFFFFu;
                const uint32_t j = b0 + lane;
                uint32_t owner = 0, obase = (uint32_t)__builtin_amdgcn_readlane((int)mine, 0);
                for (uint32_t t = 1; t < ntile; t++) {            // last tile whose base is <= j
                    const uint32_t bt = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)t);
                    if (bt <= j) { owner = t; obase = bt; }
                }
                if (lane < m) pay = bits.pay[((size_t)f * bits.stride + T0 + owner) * BL_PAY_CAP + (j - obase)];
            } else if (lane < m) {
                pay = baseline_from_tiles(bits, f, b0 + lane, T0, T1);
            }
        }
        if (lane < m) {
            int32_t clk;
            if (bits.pay) {
                clk = (int32_t)(uint32_t)pay;
                bsv = (uint32_t)((pay >> 32) & 0xFFu);
            } else {
                clk = ws.bl_clk[bsb + b0 + lane];
                bsv = (uint32_t)ws.bl_bs[bsb + b0 + lane];
            }
            phase_of(clk, prm, &q, &r);
            key = ((unsigned long long)r << 32) | lane;
        }
        uint32_t ph, idx;
        if ((uint32_t)prm.cycle < (1u << 26)) {
            // phase and lane fit one dword: the network exchanges single registers
            uint32_t k32 = lane < m ? (r << 6) | lane : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ j) << 2), (int)k32);
                    const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
                    k32 = take_min ? min(k32, other) : max(k32, other);
                }
            }
            ph = k32 >> 6;
            idx = k32 & 63u;
        } else {
#pragma unroll
            for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    const unsigned long long other = shfl64(key, (int)(lane ^ j));
                    const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
                    key = take_min ? (key < other ? key : other) : (key > other ? key : other);
                }
            }
            ph = (uint32_t)(key >> 32);
            idx = (uint32_t)key;
        }
        // lane i now holds the i-th smallest (phase, index); the first of every phase is kept
        const uint32_t ph_left = (uint32_t)__shfl_up((int)ph, 1, 64);
        const bool keep = lane < m && (lane == 0 || ph_left != ph);
        const uint64_t km = __ballot(keep);
        const uint32_t ia = (idx & 63u) << 2;
        const uint32_t qi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)ia, (int)q), bi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)ia, (int)bsv);
        if (keep) {
            const uint32_t o = __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            ws.srt_rb[bsb + b0 + o] = ph;
            ws.srt_pay[bsb + b0 + o] = make_uint4(idx, qi, bi, 0);
        }
        if (lane == 0) ws.grp_nd[gs + g] = (uint32_t)__popcll(km);
    }
    if (my_flags && lane == 0) atomicOr(&wg_flags, my_flags);
    __syncthreads();
    if (threadIdx.x == 0 && wg_flags) atomicOr(ws.rt_nbase + f, wg_flags);
}

// Per group: sort baseline phases (clk_b mod cycle), keep the earliest baseline of each phase.
// Output per distinct phase, ascending: srt_rb = phase, srt_pay = (baseline index in its group,
// clk_b div cycle, bs_b, 0).
__global__ __launch_bounds__(256) void k_bl_sort(correct_ws ws, uint32_t row_cap, correct_params prm,
                                                  const uint32_t *stats, tile_bits bits, uint32_t skip_upto)
{
    __shared__ unsigned long long keys[SORT_CAP];
    __shared__ uint32_t sh[17];
    const uint32_t f = blockIdx.y, tid = threadIdx.x;
    if (ws.gate && !ws.gate[f]) return;
    if (bits.pay && skip_upto >= MID_SORT_MAX && !(ws.rt_nbase[f] & 4u)) return;    // k_bl_sort_small saw no group for this kernel
    const uint32_t G = stats[f * SLAM_STAT_WORDS + SLAM_STAT_GROUPS];
    const size_t gs = (size_t)f * (row_cap + 1);
    const size_t bsb = (size_t)f * ws.bl_cap;
    for (uint32_t g = blockIdx.x; g < G; g += gridDim.x) {
        const uint32_t b0 = ws.grp_bl[gs + g], b1 = ws.grp_bl[gs + g + 1];
        const uint32_t m = b1 - b0;
        if (m <= skip_upto) continue;                       // empty, or sorted by k_bl_sort_small
        if (m == 0) {
            if (tid == 0) ws.grp_nd[gs + g] = 0;
            continue;
        }
        if (bits.pay) {
            // The baselines still sit in per-tile lists (row order inside a tile): write this group's stretch of
            // the row-ordered list.  Its tiles are those of its rows.
            const uint32_t T0 = ws.grp_row[gs + g] / SLOT_ROWS, T1 = (ws.grp_row[gs + g + 1] - 1) / SLOT_ROWS;
            for (uint32_t i = tid; i < m; i += 256) {
                const unsigned long long p = baseline_from_tiles(bits, f, b0 + i, T0, T1);
                ws.bl_clk[bsb + b0 + i] = (int32_t)(uint32_t)p;
                ws.bl_bs[bsb + b0 + i] = (int32_t)((p >> 32) & 0xFFu);
            }
            __syncthreads();
        }
        if (m <= SORT_CAP) {
            uint32_t P = 1;
            while (P < m) P <<= 1;
            for (uint32_t i = tid; i < P; i += 256) {
                unsigned long long key = ~0ull;
                if (i < m) {
                    uint32_t q, r;
                    phase_of(ws.bl_clk[bsb + b0 + i], prm, &q, &r);
                    key = ((unsigned long long)r << 32) | i;
                }
                keys[i] = key;
            }
            __syncthreads();
            for (uint32_t k = 2; k <= P; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < P; i += 256) {
                        uint32_t l = i ^ j;
                        if (l > i) {
                            unsigned long long a = keys[i], b = keys[l];
                            bool up = (i & k) == 0;
                            if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                        }
                    }
                    __syncthreads();
                }
            }
            uint32_t run = 0;                                  // distinct phases written so far
            for (uint32_t seg = 0; seg < m; seg += 256) {
                uint32_t i = seg + tid;
                bool keep = false;
                unsigned long long key = 0;
                if (i < m) {
                    key = keys[i];
                    keep = (i == 0) || ((uint32_t)(keys[i - 1] >> 32) != (uint32_t)(key >> 32));
                }
                uint32_t tot;
                uint32_t ex = block_excl_scan(keep ? 1u : 0u, sh, &tot);
                if (keep) {
                    uint32_t o = run + ex, idx = (uint32_t)key, q, r;
                    phase_of(ws.bl_clk[bsb + b0 + idx], prm, &q, &r);
                    ws.srt_rb[bsb + b0 + o] = r;
                    ws.srt_pay[bsb + b0 + o] = make_uint4(idx, q, (uint32_t)ws.bl_bs[bsb + b0 + idx], 0);
                }
                run += tot;
            }
            if (tid == 0) ws.grp_nd[gs + g] = run;
            __syncthreads();
        } else {
            // rank path: a phase is kept when no earlier baseline has it; its position is the number
            // of kept phases below it.  O(m^2) per group, for groups too large for the LDS sort.
            uint32_t *tmp_rb = ws.tmp_a + bsb + b0, *tmp_keep = ws.tmp_b + bsb + b0;
            for (uint32_t i = tid; i < m; i += 256) {
                uint32_t q, r;
                phase_of(ws.bl_clk[bsb + b0 + i], prm, &q, &r);
                tmp_rb[i] = r;
            }
            __syncthreads();
            for (uint32_t i = tid; i < m; i += 256) {
                uint32_t rb = tmp_rb[i], keep = 1;
                for (uint32_t q = 0; q < i; q++)
                    if (tmp_rb[q] == rb) { keep = 0; break; }
                tmp_keep[i] = keep;
            }
            __syncthreads();
            uint32_t kept_total = 0;
            for (uint32_t i = tid; i < m; i += 256) {
                if (!tmp_keep[i]) continue;
                uint32_t rb = tmp_rb[i], pos = 0, q, r;
                for (uint32_t z = 0; z < m; z++) pos += (tmp_keep[z] && tmp_rb[z] < rb) ? 1u : 0u;
                phase_of(ws.bl_clk[bsb + b0 + i], prm, &q, &r);
                ws.srt_rb[bsb + b0 + pos] = rb;
                ws.srt_pay[bsb + b0 + pos] = make_uint4(i, q, (uint32_t)ws.bl_bs[bsb + b0 + i], 0);
                kept_total++;
            }
            uint32_t tot;
            block_excl_scan(kept_total, sh, &tot);
            if (tid == 0) ws.grp_nd[gs + g] = tot;
        }
    }
}

template <typename TC>
struct corr_none;
template <>
struct corr_none<uint8_t> { static constexpr uint8_t v = 0xFF; };
template <>
struct corr_none<int32_t> { static constexpr int32_t v = -1; };

constexpr int TAB_CAP = 1024;         // phases of one group staged in LDS by k_rows_correct

constexpr uint32_t TAB_MAX_GROUPS = 16; // tiles spanning more groups than this search the tables in global memory

// Corrected BS for a row (clk div cycle = q1, clk mod cycle = r) against the chosen baseline (clk_b div cycle = qb,
// clk_b mod cycle = vw, beam bs_b): d = clk - clk_b = Q*cycle + x with |x| < cycle, k = round-half-even(d / cycle),
// result (bs_b + k) mod MOD_BASE (bs_beam_correction.py:103-112).
__device__ __forceinline__ int32_t round_to_beam(uint32_t q1, uint32_t r, uint32_t qb, uint32_t vw, uint32_t bs_b,
                                                 const correct_params &prm)
{
    const uint32_t cyc = (uint32_t)prm.cycle;
    const int32_t Q = (int32_t)(q1 - qb);
    const int32_t x = (int32_t)r - (int32_t)vw;
    if (prm.small_tol) {
        // 0 <= 2*tol < cycle and the circular distance is within tol: |x| <= tol, or the pair wraps around the
        // cycle (|x| >= cycle - tol > cycle / 2) and k moves one step; an exact half never occurs
        const uint32_t ax = (uint32_t)(x < 0 ? -x : x);
        const int32_t k = Q + (ax > (uint32_t)prm.tol ? (x > 0 ? 1 : -1) : 0);
        return ((int32_t)bs_b + k) & (prm.mod_base - 1);
    }
    const uint32_t ax2 = 2u * (uint32_t)(x < 0 ? -x : x);
    int32_t k;
    if (ax2 < cyc) k = Q;
    else if (ax2 > cyc) k = x > 0 ? Q + 1 : Q - 1;
    else { const int32_t base = x > 0 ? Q : Q - 1; k = (base & 1) ? base + 1 : base; }
    const int32_t v = (int32_t)bs_b + k;
    return (prm.mod_base & (prm.mod_base - 1)) == 0 ? (v & (prm.mod_base - 1)) : pymod(v, prm.mod_base);
}

// Given lo = number of table phases below r: pick between the circular neighbours, apply the tolerance
// and round.  Returns true and the corrected BS when a baseline qualifies.
template <typename RB, typename PAY>
__device__ __forceinline__ bool decide_phase(const RB *rb, const PAY *pay, uint32_t nd, uint32_t lo, uint32_t q1,
                                             uint32_t r, const correct_params &prm, int32_t *out_bs)
{
    const uint32_t is = lo < nd ? lo : 0, ip = lo > 0 ? lo - 1 : nd - 1;      // circular successor / predecessor
    const uint32_t vs = rb[is], vp = rb[ip];
    const uint32_t cyc = (uint32_t)prm.cycle;
    const uint32_t xs = vs > r ? vs - r : r - vs, xp = vp > r ? vp - r : r - vp;
    const uint32_t rs_ = min(xs, cyc - xs), rp_ = min(xp, cyc - xp);
    uint4 ws_ = pay[is];
    uint32_t resid = rs_, vw = vs;
    if (rp_ <= rs_) {
        const uint4 wp = pay[ip];
        if (rp_ < rs_ || wp.x < ws_.x) { ws_ = wp; resid = rp_; vw = vp; }      // ties: the earlier baseline
    }
    if ((int64_t)resid > (int64_t)prm.tol) return false;
    *out_bs = round_to_beam(q1, r, ws_.y, vw, ws_.z, prm);
    return true;
}

// Nearest phase of (q1, r) in a group's table rb[0..nd) / pay[0..nd): returns true and the corrected BS.
template <typename RB, typename PAY>
__device__ __forceinline__ bool best_phase(const RB *rb, const PAY *pay, uint32_t nd, uint32_t q1, uint32_t r,
                                           const correct_params &prm, int32_t *out_bs)
{
    uint32_t lo = 0, hi = nd;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (rb[mid] < r) lo = mid + 1; else hi = mid;
    }
    return decide_phase(rb, pay, nd, lo, q1, r, prm, out_bs);
}

// Per row: nearest baseline phase of its group -> corrected BS or "none"; kept count per tile.
template <typename T8, typename T32, typename TC>
__global__ __launch_bounds__(256) void k_rows_correct(rows_in_t<T8, T32> in, uint32_t rt_max,
                                                       const uint32_t *rt_gbase, correct_ws ws, TC *corr,
                                                       correct_params prm, uint32_t *rt_nkept,
                                                       const uint32_t *stats)
{
    __shared__ uint32_t sh[17];
    __shared__ uint32_t t_rb[TAB_CAP];
    __shared__ uint4 t_pay[TAB_CAP];
    __shared__ uint32_t shk[4];
    const uint32_t f = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const frame_span sp = span_of(in, f, stats);
    const uint32_t n = sp.n;
    const uint32_t i0 = tile * ROWS_PER_BLOCK + tid * 4;
    if (tile * ROWS_PER_BLOCK >= n) return;
    const size_t fb = (size_t)f * in.row_cap;
    uint32_t bound = 0, live = 0;
    if (i0 < n) {
        const uint32_t mk = ws.rmask[(fb + i0) >> 2];
        bound = mk & 15u;
        live = (mk >> 8) & 15u;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(__popc(bound), sh, &tot);
    const uint32_t gbase = rt_gbase[(size_t)f * rt_max + tile];
    const size_t gs = (size_t)f * (in.row_cap + 1);
    const size_t bsb = (size_t)f * ws.bl_cap;
    // groups this tile touches: the one running into it (gbase - 1) and those starting in it
    const uint32_t glo = gbase ? gbase - 1 : 0, ghi = gbase + tot - 1;
    T8 fl[4] = {1, 1, 1, 1};
    T32 ck[4] = {0, 0, 0, 0};
    uint32_t gi[4], q1[4], r1[4];
    TC res[4];
    if (live) {
        load4(in.flag + fb + i0, fl);
        load4(in.clk + fb + i0, ck);
    }
    {
        uint32_t g = gbase + ex;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((bound >> k) & 1u) g++;
            gi[k] = g - 1;
            res[k] = corr_none<TC>::v;
            if (!((live >> k) & 1u)) fl[k] = 1;               // entries that are not rows take no part
            phase_of((int32_t)ck[k], prm, &q1[k], &r1[k]);
        }
    }
    uint32_t kept = 0;
    const bool staged = (ghi - glo) < TAB_MAX_GROUPS;
    if (staged) {
        for (uint32_t gg = glo; gg <= ghi; gg++) {
            const uint32_t nd = ws.grp_nd[gs + gg];
            if (nd == 0) continue;                              // block-uniform
            const uint32_t *grb = ws.srt_rb + bsb + ws.grp_bl[gs + gg];
            const uint4 *gpay = ws.srt_pay + bsb + ws.grp_bl[gs + gg];
            const bool in_lds = nd <= TAB_CAP;
            if (in_lds) {
                for (uint32_t i = tid; i < nd; i += 256) { t_rb[i] = grb[i]; t_pay[i] = gpay[i]; }
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (fl[k] == 1 || gi[k] != gg) continue;
                int32_t bsn;
                const bool hit = in_lds ? best_phase(t_rb, t_pay, nd, q1[k], r1[k], prm, &bsn)
                                        : best_phase(grb, gpay, nd, q1[k], r1[k], prm, &bsn);
                if (hit) { res[k] = (TC)bsn; kept++; }
            }
            if (in_lds) __syncthreads();
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (fl[k] == 1) continue;
            const uint32_t nd = ws.grp_nd[gs + gi[k]];
            if (nd == 0) continue;
            int32_t bsn;
            if (best_phase(ws.srt_rb + bsb + ws.grp_bl[gs + gi[k]], ws.srt_pay + bsb + ws.grp_bl[gs + gi[k]], nd,
                           q1[k], r1[k], prm, &bsn)) {
                res[k] = (TC)bsn;
                kept++;
            }
        }
    }
    if (i0 < n) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i0 + k < n) corr[fb + i0 + k] = res[k];
    }
    const uint32_t c = wave_sum(kept);
    if (lane_id() == 0) shk[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) rt_nkept[(size_t)f * rt_max + tile] = shk[0] + shk[1] + shk[2] + shk[3];
}

template <typename T8, typename T32, typename TC>
__global__ __launch_bounds__(256) void k_rows_emit_kept(rows_in_t<T8, T32> in, uint32_t rt_max, const TC *corr,
                                                         const uint32_t *rt_kbase, kept_out_t<T8, T32> out,
                                                         const uint32_t *stats)
{
    __shared__ uint32_t sh[17];
    const uint32_t f = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS];
    const uint32_t i0 = tile * ROWS_PER_BLOCK + tid * 4;
    if (tile * ROWS_PER_BLOCK >= n) return;
    const size_t fb = (size_t)f * in.row_cap;
    TC c[4] = {corr_none<TC>::v, corr_none<TC>::v, corr_none<TC>::v, corr_none<TC>::v};
    if (i0 < n) load4(corr + fb + i0, c);
    uint32_t keep = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (i0 + k < n && c[k] != corr_none<TC>::v) keep |= 1u << k;
    uint32_t tot;
    uint32_t ex = block_excl_scan(__popc(keep), sh, &tot);
    if (!keep) return;
    size_t o = fb + rt_kbase[(size_t)f * rt_max + tile] + ex;
    T8 ue[4];
    T32 rs[4], ck[4];
    load4(in.ue + fb + i0, ue);
    load4(in.rss + fb + i0, rs);
    load4(in.clk + fb + i0, ck);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((keep >> k) & 1u) {
            out.ue[o] = ue[k];
            out.bs[o] = (T8)c[k];
            out.rss[o] = rs[k];
            out.clk[o] = ck[k];
            o++;
        }
    }
}

// Correction fused with the emission of kept rows (slotted layout of kept_out_t):
// group ids come from the bits k_rows_summarize left per 4 entries, the four rows of a thread search their
// group's LDS-staged phase table in lock step (fixed trip count over the table padded to a power of two), and a
// kept row goes straight to the front of its block's entries.
#ifndef SLAM_FIN_TPB
#define SLAM_FIN_TPB 3
#endif
// consecutive blocks of entries one workgroup of k_rows_finish walks (lean, mid build): 2 / 3 / 4 measured 2.95 / 2.93 / 3.17 ms
// per 800 frames after the flag column went; asking the compiler for seven waves per SIMD instead of six: 3.25 ms
constexpr int FIN_TPB = SLAM_FIN_TPB;
#ifndef SLAM_FIN_TPB_FULL
#define SLAM_FIN_TPB_FULL 3
#endif
constexpr int FIN_TPB_FULL = SLAM_FIN_TPB_FULL;       // ... and the full build

// v of the lane whose byte address (4 * lane, below 256) is given: a bare ds_bpermute, without the lane
// arithmetic __shfl adds for sub-wave widths.
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t lane_byte_addr)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)lane_byte_addr, (int)v);
}

template <typename V, typename T>
__device__ __forceinline__ void unpack4(const V &q, T v[4])
{
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}

//
// Three builds of one body.  The kernel is bound by occupancy as much as by anything (halving its waves per SIMD costs
// 68 %), so the common cases each get a build that holds only their own paths:
//   LEAN  every phase table of the frame fits a wave's registers (no group above 64 baselines: bit 0 of the frame's
//         flags clear), tolerance below half a cycle: 80 VGPRs, 10 KiB of LDS, six waves per SIMD;
//   MID   frames with larger groups whose tables all stay below MID_TAB_CAP phases (bit 0 set, bit 3 clear), same
//         tolerance rule: every table, small ones included, is searched in LDS;
//   FULL  every path (tables in registers, LDS or global memory, any tolerance, any modulus).
// LEAN and MID leave a workgroup that spans TAB_MAX_GROUPS groups or more to FULL: they set its flag in `defer` and bit 1
// of the frame's flags.  Given `defer`, FULL serves only those workgroups and the frames with bit 3 set; where no bit
// asks for it -- the usual case -- its workgroups leave after one load.
enum { FIN_LEAN = 0, FIN_MID = 1, FIN_FULL = 2 };
template <typename T8, typename T32, int MODE, int TPB>
__global__ __launch_bounds__(256) void k_rows_finish(rows_in_t<T8, T32> in, uint32_t rt_max, const uint32_t *rt_gbase,
                                                      correct_ws ws, correct_params prm, kept_out_t<T8, T32> out,
                                                      uint32_t *stats, uint32_t *defer, uint32_t *frame_flags)
{
    constexpr bool LEAN = MODE == FIN_LEAN, MID = MODE == FIN_MID, FULL = MODE == FIN_FULL;
    constexpr uint32_t CAP = MID ? MID_TAB_CAP : (uint32_t)TAB_CAP;
    using V8 = typename vec4_of<T8>::type;
    using V32 = typename vec4_of<T32>::type;
    __shared__ uint32_t sh[17];
    __shared__ uint32_t t_rb[LEAN ? 1 : CAP];
    __shared__ uint4 t_pay[LEAN ? 1 : CAP];
    __shared__ uint32_t g_nd[TAB_MAX_GROUPS], g_bl[TAB_MAX_GROUPS];
    __shared__ uint32_t wave_tot[2][4];                   // kept rows per wave, double-buffered over the blocks walked
    __shared__ __attribute__((aligned(16))) T32 k_rs[ROWS_PER_BLOCK], k_ck[ROWS_PER_BLOCK];   // kept rows of a block, packed
    __shared__ __attribute__((aligned(16))) T8 k_ue[ROWS_PER_BLOCK], k_bs[ROWS_PER_BLOCK];
    const uint32_t f = blockIdx.y, bx = blockIdx.x, tile0 = bx * TPB, tid = threadIdx.x;
    if (ws.gate && !ws.gate[f]) return;
    bool only_deferred = false;                          // full build: serve only the blocks of workgroups that deferred
    if (defer) {
        // bit 0: a group above 64 baselines; bit 1: a workgroup deferred; bit 3: a table of MID_TAB_CAP phases or more
        const uint32_t ff = frame_flags[f];
        if (LEAN && (ff & 1u)) return;
        if (MID && (!(ff & 1u) || (ff & 8u))) return;
        if (FULL && !(ff & 10u)) return;
        if (FULL && !(ff & 8u)) {
            only_deferred = true;
            bool any = false;
#pragma unroll
            for (int j = 0; j < TPB; j++)
                if (tile0 + j < rt_max) any = any || defer[(size_t)f * rt_max + (tile0 + j) / FIN_TPB] != 0;
            if (!any) return;
        }
    }
    const uint32_t n = stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS];
    if (tile0 * ROWS_PER_BLOCK >= n) return;
    const size_t fb = (size_t)f * in.row_cap;
    // every load of the workgroup's TPB blocks of entries is issued up front
    uint32_t mk[TPB], gb[TPB + 1];
    V8 uev[TPB];
    V32 ckv[TPB], rsv[TPB];
    const uint32_t G = stats[f * SLAM_STAT_WORDS + SLAM_STAT_GROUPS];
#pragma unroll
    for (int j = 0; j < TPB; j++) {
        const uint32_t i0 = (tile0 + j) * ROWS_PER_BLOCK + tid * 4;
        mk[j] = 0;
        if (i0 < n) {
            mk[j] = ws.rmask[(fb + i0) >> 2];          // the FLAG == 1 bits come with the mask: the flag column is not read
            ckv[j] = *reinterpret_cast<const V32 *>(in.clk + fb + i0);
            uev[j] = *reinterpret_cast<const V8 *>(in.ue + fb + i0);
            rsv[j] = *reinterpret_cast<const V32 *>(in.rss + fb + i0);
        }
    }
#pragma unroll
    for (int j = 0; j <= TPB; j++)        // groups started before each block (exclusive bases; G past the end)
        gb[j] = (tile0 + j) * ROWS_PER_BLOCK < n ? rt_gbase[(size_t)f * rt_max + tile0 + j] : G;
    // groups the workgroup touches: the one running into it and those starting in it
    const size_t gs = (size_t)f * (in.row_cap + 1);
    const size_t bsb = (size_t)f * ws.bl_cap;
    const uint32_t glo_all = gb[0] ? gb[0] - 1 : 0, ghi_all = gb[TPB] - 1;
    const bool staged = (ghi_all - glo_all) < TAB_MAX_GROUPS;
    if (staged && tid <= ghi_all - glo_all) {
        g_nd[tid] = ws.grp_nd[gs + glo_all + tid];
        g_bl[tid] = ws.grp_bl[gs + glo_all + tid];
    }
    __syncthreads();
    if (!FULL) {
        if (tid == 0) {
            defer[(size_t)f * rt_max + bx] = staged ? 0u : 1u;
            if (!staged) atomicOr(frame_flags + f, 2u);
        }
        if (!staged) return;
    }
    uint32_t in_lds_group = 0xFFFFFFFFu, P = 2;          // group whose table sits in t_rb / t_pay, its padded length
    uint32_t in_reg_group = 0xFFFFFFFFu, t_r = 0xFFFFFFFFu, t_x = 0, t_y = 0, t_z = 0;   // small table: entry `lane`
    uint32_t pv[7] = {0, 0, 0, 0, 0, 0, 0};              // its entries 7, 15, ..., 55 (wave-uniform)
#pragma unroll
    for (int j = 0; j < TPB; j++) {
        const uint32_t tile = tile0 + j;
        if (tile * ROWS_PER_BLOCK >= n) break;
        if (FULL && only_deferred && !defer[(size_t)f * rt_max + tile / FIN_TPB]) continue;     // another build did this block
        const uint32_t bound = mk[j] & 15u, live = (mk[j] >> 8) & 15u;
        // fl[k] == 1: the entry takes no part (a FLAG = 1 row, or not a row at all)
        const uint32_t skip = ((mk[j] >> 12) | ~live) & 15u;
        uint32_t fl[4] = {skip & 1u, (skip >> 1) & 1u, (skip >> 2) & 1u, (skip >> 3) & 1u};
        T8 ue[4] = {0, 0, 0, 0};
        T32 ck[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0};
        if (live) { unpack4(ckv[j], ck); unpack4(uev[j], ue); unpack4(rsv[j], rs); }
        // rows before this thread's that start a group: none in most blocks (the bases say so without a scan)
        uint32_t ex = 0;
        if (gb[j + 1] != gb[j]) {
            uint32_t tot;
            ex = block_excl_scan(__popc(bound), sh, &tot);
        }
        const uint32_t glo = gb[j] ? gb[j] - 1 : 0, ghi = gb[j + 1] - 1;
        uint32_t gi[4], q1[4], r1[4];
        int32_t res[4];
        {
            uint32_t g = gb[j] + ex;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((bound >> k) & 1u) g++;
                gi[k] = g - 1;
                res[k] = -1;
                phase_of((int32_t)ck[k], prm, &q1[k], &r1[k]);
            }
        }
        uint32_t keep = 0;
        if (!FULL || staged) {
            for (uint32_t gg = glo; gg <= ghi; gg++) {
                const uint32_t nd = g_nd[gg - glo_all];
                if (nd == 0) continue;                              // block-uniform
                const uint32_t *grb = ws.srt_rb + bsb + g_bl[gg - glo_all];
                const uint4 *gpay = ws.srt_pay + bsb + g_bl[gg - glo_all];
                if (!MID && nd <= 64) {
                    // the whole table in registers, one entry per lane; lookups are lane permutes (no LDS banks).
                    // A wave none of whose rows belongs to this group skips it (no barrier on this path): in a
                    // block that spans two groups only the wave holding the boundary searches twice.
                    if (!__any((fl[0] != 1 && gi[0] == gg) || (fl[1] != 1 && gi[1] == gg) || (fl[2] != 1 && gi[2] == gg) ||
                               (fl[3] != 1 && gi[3] == gg)))
                        continue;
                    if (in_reg_group != gg) {
                        const uint32_t lane = lane_id();
                        t_r = lane < nd ? grb[lane] : 0xFFFFFFFFu;
                        if (lane < nd) { const uint4 q = gpay[lane]; t_x = q.x; t_y = q.y; t_z = q.z; }
#pragma unroll
                        for (int i = 0; i < 7; i++) pv[i] = (uint32_t)__builtin_amdgcn_readlane((int)t_r, 8 * i + 7);
                        in_reg_group = gg;
                        in_lds_group = 0xFFFFFFFFu;       // pv no longer belongs to the table staged in LDS
                    }
                    // lower bound of every row's phase: the first three levels compare against seven entries held
                    // in scalar registers, the last four fetch one entry per row and step
                    // positions are kept as lane byte addresses (4 * index), the unit ds_bpermute takes
                    uint32_t lo[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r = r1[k];
                        const bool a = pv[3] < r;                                   // entry 31
                        const uint32_t m = a ? pv[5] : pv[1];                        // entry 47 : 15
                        const bool b = m < r;
                        const uint32_t q = a ? (b ? pv[6] : pv[4]) : (b ? pv[2] : pv[0]);   // entries 55, 39, 23, 7
                        lo[k] = (a ? 128u : 0u) + (b ? 64u : 0u) + (q < r ? 32u : 0u);
                    }
#pragma unroll
                    for (uint32_t h = 4; h; h >>= 1) {
#pragma unroll
                        for (int k = 0; k < 4; k++) lo[k] += lane_read(t_r, lo[k] + 4u * (h - 1)) < r1[k] ? 4u * h : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) lo[k] += lane_read(t_r, lo[k]) < r1[k] ? 4u : 0u;
                    if (LEAN || prm.small_tol) {
                        // 0 <= 2*tol < cycle: a neighbour qualifies only through its short way round, so the forward
                        // distance to the successor and the backward distance to the predecessor (either may wrap past
                        // the cycle's end, which moves k by one) are the residuals; no absolute values, no halves.
                        const uint32_t nd4 = 4u * nd, cyc = (uint32_t)prm.cycle;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool wrap_s = lo[k] >= nd4, wrap_p = lo[k] == 0;
                            const uint32_t is = wrap_s ? 0u : lo[k], ip = wrap_p ? nd4 - 4u : lo[k] - 4u;
                            const uint32_t vs = lane_read(t_r, is), vp = lane_read(t_r, ip);
                            const uint32_t rs_ = vs + (wrap_s ? cyc : 0u) - r1[k], rp_ = r1[k] + (wrap_p ? cyc : 0u) - vp;
                            bool pred = rp_ < rs_;
                            if (__any(rp_ == rs_)) {                                 // ties: the earlier baseline (rare)
                                const uint32_t xs_ = lane_read(t_x, is), xp_ = lane_read(t_x, ip);
                                pred = pred || (rp_ == rs_ && xp_ < xs_);
                            }
                            const uint32_t w = pred ? ip : is, resid = pred ? rp_ : rs_;
                            const uint32_t adj = pred ? (wrap_p ? 0xFFFFFFFFu : 0u) : (wrap_s ? 1u : 0u);
                            const uint32_t wq = lane_read(t_y, w), wb = lane_read(t_z, w);
                            if (fl[k] == 1 || gi[k] != gg || resid > (uint32_t)prm.tol) continue;
                            res[k] = (int32_t)((wb + (q1[k] - wq) + adj) & (uint32_t)(prm.mod_base - 1));
                            keep |= 1u << k;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            // all lanes take part in the permutes; rows of other groups drop the result
                            const uint32_t li = lo[k] >> 2;
                            const uint32_t is = li < nd ? li : 0, ip = li > 0 ? li - 1 : nd - 1;
                            const uint32_t vs = lane_read(t_r, 4u * is), vp = lane_read(t_r, 4u * ip);
                            const uint32_t cyc = (uint32_t)prm.cycle;
                            const uint32_t ds_ = vs > r1[k] ? vs - r1[k] : r1[k] - vs, dp_ = vp > r1[k] ? vp - r1[k] : r1[k] - vp;
                            const uint32_t rs_ = min(ds_, cyc - ds_), rp_ = min(dp_, cyc - dp_);
                            bool pred = rp_ < rs_;
                            if (__any(rp_ == rs_)) {                                     // ties: the earlier baseline (rare)
                                const uint32_t xs_ = lane_read(t_x, 4u * is), xp_ = lane_read(t_x, 4u * ip);
                                pred = pred || (rp_ == rs_ && xp_ < xs_);
                            }
                            const uint32_t w = pred ? ip : is, resid = pred ? rp_ : rs_, vw = pred ? vp : vs;
                            const uint32_t wq = lane_read(t_y, 4u * w), wb = lane_read(t_z, 4u * w);
                            if (fl[k] == 1 || gi[k] != gg || (int64_t)resid > (int64_t)prm.tol) continue;
                            res[k] = round_to_beam(q1[k], r1[k], wq, vw, wb, prm);
                            keep |= 1u << k;
                        }
                    }
                } else if (LEAN) {
                    // not reached: such a workgroup has deferred
                } else if (MID || nd < CAP) {                       // (the mid build only sees frames whose tables all fit)
                    if (in_lds_group != gg) {                       // consecutive blocks mostly stay in one group
                        __syncthreads();
                        P = 8;                                      // power of two > nd, at least 8: the padded table length
                        while (P <= nd) P <<= 1;
                        for (uint32_t i = tid; i < P; i += 256) {
                            t_rb[i] = i < nd ? grb[i] : 0xFFFFFFFFu;
                            if (i < nd) t_pay[i] = gpay[i];
                        }
                        __syncthreads();
                        in_lds_group = gg;
                        // the seven entries the first three levels of every search compare against, kept in scalar
                        // registers; the register table's copies are gone with them
                        const uint32_t e8 = P >> 3;
#pragma unroll
                        for (int i = 0; i < 7; i++) pv[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_rb[e8 * (i + 1) - 1]);
                        in_reg_group = 0xFFFFFFFFu;
                    }
                    uint32_t lo[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r = r1[k];
                        const bool a = pv[3] < r;
                        const uint32_t m = a ? pv[5] : pv[1];
                        const bool b = m < r;
                        const uint32_t q = a ? (b ? pv[6] : pv[4]) : (b ? pv[2] : pv[0]);
                        lo[k] = (a ? P >> 1 : 0u) + (b ? P >> 2 : 0u) + (q < r ? P >> 3 : 0u);
                    }
                    for (uint32_t h = P >> 4; h; h >>= 1) {
#pragma unroll
                        for (int k = 0; k < 4; k++) lo[k] += t_rb[lo[k] + h - 1] < r1[k] ? h : 0u;
                    }
                    if (MID || prm.small_tol) {
                        // as in the register path: with 0 <= 2*tol < cycle the forward distance to the successor and
                        // the backward distance to the predecessor are the residuals; only the winner's payload is read
                        const uint32_t cyc = (uint32_t)prm.cycle;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (fl[k] == 1 || gi[k] != gg) continue;
                            const bool wrap_s = lo[k] >= nd, wrap_p = lo[k] == 0;
                            const uint32_t is = wrap_s ? 0u : lo[k], ip = wrap_p ? nd - 1u : lo[k] - 1u;
                            const uint32_t rs_ = t_rb[is] + (wrap_s ? cyc : 0u) - r1[k], rp_ = r1[k] + (wrap_p ? cyc : 0u) - t_rb[ip];
                            const bool pred = rp_ < rs_ || (rp_ == rs_ && t_pay[ip].x < t_pay[is].x);   // ties: the earlier baseline
                            if ((pred ? rp_ : rs_) > (uint32_t)prm.tol) continue;
                            const uint4 w = t_pay[pred ? ip : is];
                            const uint32_t adj = pred ? (wrap_p ? 0xFFFFFFFFu : 0u) : (wrap_s ? 1u : 0u);
                            res[k] = (int32_t)((w.z + (q1[k] - w.y) + adj) & (uint32_t)(prm.mod_base - 1));
                            keep |= 1u << k;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (fl[k] == 1 || gi[k] != gg) continue;
                            int32_t bsn;
                            if (decide_phase(t_rb, t_pay, nd, lo[k], q1[k], r1[k], prm, &bsn)) { res[k] = bsn; keep |= 1u << k; }
                        }
                    }
                } else if (FULL) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (fl[k] == 1 || gi[k] != gg) continue;
                        int32_t bsn;
                        if (best_phase(grb, gpay, nd, q1[k], r1[k], prm, &bsn)) { res[k] = bsn; keep |= 1u << k; }
                    }
                }
            }
        } else if (FULL) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (fl[k] == 1) continue;
                const uint32_t nd = ws.grp_nd[gs + gi[k]];
                if (nd == 0) continue;
                int32_t bsn;
                if (best_phase(ws.srt_rb + bsb + ws.grp_bl[gs + gi[k]], ws.srt_pay + bsb + ws.grp_bl[gs + gi[k]], nd,
                               q1[k], r1[k], prm, &bsn)) {
                    res[k] = bsn;
                    keep |= 1u << k;
                }
            }
        }
        // kept rows before this thread's: ballots inside the wave, one exchange of wave totals across the block
        uint32_t kx = 0, wtot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t bk = __ballot((keep >> k) & 1u);
            kx += __builtin_amdgcn_mbcnt_hi((uint32_t)(bk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bk, 0u));
            wtot += (uint32_t)__popcll(bk);
        }
        uint32_t *wt = wave_tot[j & 1];
        if (lane_id() == 0) wt[tid >> 6] = wtot;
        __syncthreads();
        const uint32_t w0 = wt[0], w1 = wt[1], w2 = wt[2], w3 = wt[3], wv = tid >> 6;
        const uint32_t nk = w0 + w1 + w2 + w3;
        kx += (wv > 0 ? w0 : 0u) + (wv > 1 ? w1 : 0u) + (wv > 2 ? w2 : 0u);
        // the block's kept rows are packed in LDS, then leave as full-width stores (the entries between the count
        // and the next multiple of four receive leftovers; they are unused slots)
        {
            uint32_t o = kx;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((keep >> k) & 1u) {
                    k_ue[o] = ue[k];
                    k_bs[o] = (T8)res[k];
                    k_rs[o] = rs[k];
                    k_ck[o] = ck[k];
                    o++;
                }
            }
        }
        __syncthreads();
        if (tid * 4 < nk) {
            const size_t ob = fb + (size_t)tile * ROWS_PER_BLOCK + tid * 4;
            *reinterpret_cast<V8 *>(out.ue + ob) = reinterpret_cast<const V8 *>(k_ue)[tid];
            *reinterpret_cast<V8 *>(out.bs + ob) = reinterpret_cast<const V8 *>(k_bs)[tid];
            *reinterpret_cast<V32 *>(out.rss + ob) = reinterpret_cast<const V32 *>(k_rs)[tid];
            *reinterpret_cast<V32 *>(out.clk + ob) = reinterpret_cast<const V32 *>(k_ck)[tid];
        }
        if (tid == 0) out.slot_cnt[(size_t)f * out.slot_stride + tile] = nk;
    }
}

// The frame's KEPT is the sum of its blocks' counts, added up by one wave per frame after the correction kernels: an
// atomic per workgroup on the frame's counter made it a hot word on long frames (2,700 workgroups a frame at 1024 x 8192:
// 3.18 -> 3.05 ms without them).  Set, not added: running the stage again gives the same counter.
__global__ __launch_bounds__(256) void k_kept_total(uint32_t F, const uint32_t *kept_cnt, uint32_t kept_stride, uint32_t *stats)
{
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (f >= F) return;
    const uint32_t n = stats[f * SLAM_STAT_WORDS + SLAM_STAT_ROW_SLOTS];
    // blocks of ROWS_PER_BLOCK entries, or of a fused workgroup's entries where the fused kernel served the frame
    const uint32_t unit = (stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] & SLAM_LAYOUT_KEPT_TILES) ? (uint32_t)SLOT_ROWS : (uint32_t)ROWS_PER_BLOCK;
    const uint32_t nblk = (n + unit - 1) / unit;
    uint32_t k = 0;
    for (uint32_t b = lane; b < nblk; b += 64) k += kept_cnt[(size_t)f * kept_stride + b];
    k = wave_sum(k);
    if (lane == 0) stats[f * SLAM_STAT_WORDS + SLAM_STAT_KEPT] = k;
}

}  // namespace

void correct_ws::plan(ws_plan &p, uint32_t F, uint32_t row_cap, uint32_t bl_cap, size_t off[16])
{
    const uint32_t rt_max = (row_cap + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    off[0] = p.add((size_t)F * rt_max * 4);
    off[1] = p.add((size_t)F * rt_max * 4);
    off[2] = p.add((size_t)F * rt_max * 4);
    off[3] = p.add((size_t)F * (row_cap + 1) * 4);
    off[4] = p.add((size_t)F * (row_cap + 1) * 4);
    off[5] = p.add((size_t)F * (row_cap + 1) * 4);
    for (int i = 6; i < 10; i++) off[i] = p.add((size_t)F * bl_cap * 4);
    off[10] = p.add((size_t)F * bl_cap * 16);
    off[11] = p.add((size_t)F * row_cap * 4);
    off[12] = p.add((size_t)F * bl_cap * 4);
    off[13] = p.add((size_t)F * bl_cap * 4);
    off[14] = p.add((size_t)F * row_cap / 2 + 16);
    off[15] = p.add((size_t)F * rt_max * 4);
}

void correct_ws::bind(uint8_t *base, const size_t off[16], uint32_t bl_cap_)
{
    rt_nbound = (uint32_t *)(base + off[0]);
    rt_nbase = (uint32_t *)(base + off[1]);
    rt_nkept = (uint32_t *)(base + off[2]);
    grp_row = (uint32_t *)(base + off[3]);
    grp_bl = (uint32_t *)(base + off[4]);
    grp_nd = (uint32_t *)(base + off[5]);
    bl_clk = (int32_t *)(base + off[6]);
    bl_bs = (int32_t *)(base + off[7]);
    bl_gid = (uint32_t *)(base + off[8]);
    srt_rb = (uint32_t *)(base + off[9]);
    srt_pay = (uint4 *)(base + off[10]);
    corr = (int32_t *)(base + off[11]);
    tmp_a = (uint32_t *)(base + off[12]);
    tmp_b = (uint32_t *)(base + off[13]);
    rmask = (uint16_t *)(base + off[14]);
    defer = (uint32_t *)(base + off[15]);
    bl_cap = bl_cap_;
}

// magic number for unsigned division by d (1 <= d <= 2^30): q = (hi + ((n - hi) >> s1)) >> s2, hi = mulhi(m, n)
void correct_params::set_cycle(int32_t c)
{
    cycle = c;
    const uint32_t d = (uint32_t)c;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    div_m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d) + 1;
    div_s1 = l < 1 ? l : 1;
    div_s2 = l > 0 ? l - 1 : 0;
    bias = d * (uint32_t)((((uint64_t)1 << 30) + d - 1) / d);
}

template <typename T8, typename T32>
int correct_rows(slam_ctx *ctx, const rows_in_t<T8, T32> &in, uint32_t F, const correct_ws &ws,
                 const correct_params &prm, const kept_out_t<T8, T32> &out, int32_t *gid_out, uint32_t *stats)
{
    using TC = typename std::conditional<sizeof(T8) == 1, uint8_t, int32_t>::type;
    if (F == 0) return SLAM_OK;
    if (prm.cycle <= 0 || prm.cycle > (1 << 30) || prm.mod_base <= 0 || prm.div_m == 0)
        return slam_fail(ctx, SLAM_E_INVALID, "cycle must be in 1..2^30 (set through set_cycle) and mod_base positive");
    if (sizeof(T8) == 1 && prm.mod_base > 255)
        return slam_fail(ctx, SLAM_E_RANGE, "mod_base > 255 does not fit the 8-bit beam columns");
    if (in.row_cap % 16)
        return slam_fail(ctx, SLAM_E_INVALID, "row_cap must be a multiple of 16");
    const uint32_t rt_max = (in.row_cap + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    dim3 grid(rt_max, F), block(256);
    TC *corr = reinterpret_cast<TC *>(ws.corr);
    const int og = (prm.one_group || prm.explicit_baselines ? 1 : 0) | (prm.explicit_baselines ? 2 : 0);
    const int keep_bl = prm.explicit_baselines ? 1 : 0;
    correct_params prm_k = prm;
    prm_k.small_tol = prm.tol >= 0 && 2 * (int64_t)prm.tol < (int64_t)prm.cycle && (prm.mod_base & (prm.mod_base - 1)) == 0;
    if (in.slot_cnt && in.bits.rmask && out.slot_cnt && !og) {
        // bits per framing tile (left by the framing pass, completed here) -> scan over tiles -> tables -> fused
        // correction; the per-block summarize / groups kernels are not needed
        const uint32_t st_max = in.bits.stride;
        if (out.slot_stride < rt_max) return slam_fail(ctx, SLAM_E_CAPACITY, "kept slot counts: stride too small");
        correct_ws w2 = ws;
        w2.rmask = in.bits.rmask;
        // all tiles of the frames served need their bits (nothing left them): four tiles a wave instead of 64
        const uint32_t span = in.bits_fresh ? 64u : 4u;
        SLAM_LAUNCH(ctx, "tile_bits", (k_tile_bits<T8, T32>), dim3((st_max + 4 * span - 1) / (4 * span), F), block, in,
                    in.bits_fresh ? 0 : 1, stats, ws.gate, span);
        SLAM_LAUNCH(ctx, "tile_scan", k_tile_scan, dim3(F), dim3(1024), in.bits.nbound, in.bits.nbase, st_max, stats,
                    (int)SLAM_STAT_GROUPS, (int)SLAM_STAT_BASELINES, 0, ws.grp_row, ws.grp_bl, in.row_cap + 1,
                    (uint32_t)SLOT_ROWS, ws.gate);
        SLAM_LAUNCH(ctx, "tile_groups", (k_tile_groups<T8, T32>), dim3((st_max + 255) / 256, F), block, in, w2, rt_max,
                    ws.rt_nbound, stats);
        // 64 workgroups a frame (four groups per wave at 1024 groups): 128, 256 and 512 measured 10 - 15 % slower
        SLAM_LAUNCH(ctx, "bl_sort_small", k_bl_sort_small, dim3(F <= 8 ? 256 : 64, F), dim3(256), w2, in.row_cap, prm,
                    stats, in.bits);
        SLAM_LAUNCH(ctx, "bl_sort", k_bl_sort, dim3(F <= 8 ? 1024 : (F <= 64 ? 256 : 32), F), dim3(256), w2, in.row_cap, prm, stats,
                    in.bits, MID_SORT_MAX);
        // the lean build first; what it defers (large groups, a tolerance of half a cycle or more) goes to the full one
        // (an array of its own: rt_nkept holds the kept-row counts when the caller leaves kept_cnt_dev to the workspace)
        uint32_t *defer = prm_k.small_tol ? ws.defer : nullptr;
        const uint32_t wgs = (rt_max + FIN_TPB - 1) / FIN_TPB, wgs_full = (rt_max + FIN_TPB_FULL - 1) / FIN_TPB_FULL;
        if (defer) {
            SLAM_LAUNCH(ctx, "rows_finish", (k_rows_finish<T8, T32, FIN_LEAN, FIN_TPB>), dim3(wgs, F), block, in, rt_max,
                        ws.rt_nbound, w2, prm_k, out, stats, defer, ws.rt_nbase);
            SLAM_LAUNCH(ctx, "rows_finish_mid", (k_rows_finish<T8, T32, FIN_MID, FIN_TPB>), dim3(wgs, F), block, in, rt_max,
                        ws.rt_nbound, w2, prm_k, out, stats, defer, ws.rt_nbase);
        }
        SLAM_LAUNCH(ctx, "rows_finish_rest", (k_rows_finish<T8, T32, FIN_FULL, FIN_TPB_FULL>), dim3(wgs_full, F), block,
                    in, rt_max, ws.rt_nbound, w2, prm_k, out, stats, defer, ws.rt_nbase);
        SLAM_LAUNCH(ctx, "kept_total", k_kept_total, dim3((F + 3) / 4), block, F, out.slot_cnt, out.slot_stride, stats);
        return SLAM_OK;
    }
    SLAM_LAUNCH(ctx, "rows_summarize", (k_rows_summarize<T8, T32>), grid, block, in, rt_max, og, ws.rt_nbound,
                ws.rt_nbase, ws.rmask, stats);
    SLAM_LAUNCH(ctx, "tile_scan", k_tile_scan, dim3(F), dim3(1024), ws.rt_nbound, ws.rt_nbase, rt_max, stats,
                (int)SLAM_STAT_GROUPS, (int)SLAM_STAT_BASELINES, keep_bl, ws.grp_row, ws.grp_bl, in.row_cap + 1,
                (uint32_t)ROWS_PER_BLOCK, (const uint32_t *)nullptr);
    SLAM_LAUNCH(ctx, "rows_groups", (k_rows_groups<T8, T32>), grid, block, in, rt_max, ws.rt_nbound,
                ws.rt_nbase, ws, gid_out, stats);
    SLAM_LAUNCH(ctx, "bl_sort_small", k_bl_sort_small, dim3(F <= 8 ? 256 : 64, F), dim3(256), ws, in.row_cap, prm, stats,
                tile_bits());
    SLAM_LAUNCH(ctx, "bl_sort", k_bl_sort, dim3(F <= 8 ? 1024 : (F <= 64 ? 256 : 32), F), dim3(256), ws, in.row_cap, prm, stats,
                tile_bits(), MID_SORT_MAX);
    if (out.slot_cnt) {       // fused correction + kept rows into per-block slots
        if (og) return slam_fail(ctx, SLAM_E_INVALID, "slotted kept rows: whole-table mode only");
        if (out.slot_stride < rt_max) return slam_fail(ctx, SLAM_E_CAPACITY, "kept slot counts: stride too small");
        SLAM_LAUNCH(ctx, "rows_finish_rest", (k_rows_finish<T8, T32, FIN_FULL, FIN_TPB_FULL>), dim3((rt_max + FIN_TPB_FULL - 1) / FIN_TPB_FULL, F), block,
                    in, rt_max, ws.rt_nbound, ws, prm_k, out, stats, (uint32_t *)nullptr, (uint32_t *)nullptr);
        SLAM_LAUNCH(ctx, "kept_total", k_kept_total, dim3((F + 3) / 4), block, F, out.slot_cnt, out.slot_stride, stats);
        return SLAM_OK;
    }
    SLAM_LAUNCH(ctx, "rows_correct", (k_rows_correct<T8, T32, TC>), grid, block, in, rt_max, ws.rt_nbound, ws,
                corr, prm_k, ws.rt_nkept, stats);
    SLAM_LAUNCH(ctx, "tile_scan", k_tile_scan, dim3(F), dim3(1024), ws.rt_nkept, (uint32_t *)nullptr, rt_max, stats,
                (int)SLAM_STAT_KEPT, 0, 0, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (uint32_t)ROWS_PER_BLOCK,
                (const uint32_t *)nullptr);
    if (out.ue) {
        SLAM_LAUNCH(ctx, "rows_emit_kept", (k_rows_emit_kept<T8, T32, TC>), grid, block, in, rt_max, corr,
                    ws.rt_nkept, out, stats);
    }
    return SLAM_OK;
}

template int correct_rows<uint8_t, uint32_t>(slam_ctx *, const rows_in_t<uint8_t, uint32_t> &, uint32_t,
                                             const correct_ws &, const correct_params &,
                                             const kept_out_t<uint8_t, uint32_t> &, int32_t *, uint32_t *);
template int correct_rows<int32_t, int32_t>(slam_ctx *, const rows_in_t<int32_t, int32_t> &, uint32_t,
                                            const correct_ws &, const correct_params &,
                                            const kept_out_t<int32_t, int32_t> &, int32_t *, uint32_t *);
