// Angle mapping + mean pivot, NaN-aware Gaussian blur, colour scale and mesh raster.
//
// Replaces the data path of excel_heatmap_v3.py: map_angles (:264-280), compute_rssi_matrix
// (:286-314), gaussian_blur_nan_aware (:336-368) and generate_heatmap's scale / colormap /
// pcolormesh steps (:403-456, _edges :430-440).
//
// The pivot is an exact integer sum and count per (AoD axis, AoA axis) cell, accumulated in LDS
// per workgroup and merged with integer atomics, so the result does not depend on arrival order;
// the mean is one f64 divide per cell.  Float steps are f64 in the reference's operation order
// (the library is built with -ffp-contract=off so no multiply-add is fused).

#include "pipeline.hpp"

namespace {

// Entries per workgroup of the pivot, as a power of two.  A cell's accumulator packs the count (CHUNK_LOG2 + 1
// bits at the top) over the sum, so the larger the chunk the smaller the values it can take: 2^16 entries leave
// room for any non-negative int32, 2^18 for values below 2^27 (the capture format's RSS has 18 bits).  Values
// outside the room bypass the accumulator.  Fewer, larger chunks mean fewer flushes of 4096 cells each.
#ifndef SLAM_PIVOT_LOG2
#define SLAM_PIVOT_LOG2 18
#endif
#ifndef SLAM_PIVOT_NH
#define SLAM_PIVOT_NH 4
#endif
template <typename T32> struct pivot_chunk { static constexpr int log2 = 16; };
template <> struct pivot_chunk<uint32_t> { static constexpr int log2 = SLAM_PIVOT_LOG2; };
constexpr int MAX_AXIS = SMALL_AXIS;                 // the LDS-resident path: at most 64 x 64 cells
constexpr int MAX_CELLS = MAX_AXIS * MAX_AXIS;
constexpr int N_BEAMS = 64;                          // beam ids are 6 bit (validate_beam_id, excel_heatmap_v3.py:68-72)

template <typename T>
struct v4;
template <>
struct v4<uint8_t> { using type = uchar4; };
template <>
struct v4<uint32_t> { using type = uint4; };
template <>
struct v4<int32_t> { using type = int4; };

template <typename T8, typename T32>
__global__ __launch_bounds__(1024) void k_pivot(const T8 *ue, const T8 *bs, const T32 *rss, uint32_t row_cap,
                                                const uint32_t *stats, int stat_word, const int16_t *beam_to_axis,
                                                int n_axis, uint32_t cell_stride, unsigned long long *gsum, uint32_t *gcnt,
                                                const uint32_t *slot_cnt, uint32_t slot_stride, int beam_space)
{
    // beam_space: the global accumulators are the 64 x 64 beam pairs themselves (cell = bs * 64 + ue); the caller maps
    // them onto a wide axis grid once per frame (k_wide_scatter) instead of every workgroup scattering over it
    // One packed accumulator per (BS beam, UE beam) pair: count at the top, sum below (see pivot_chunk).  Rows are
    // accumulated in beam space, whatever the size of the axis table; the beam -> axis mapping is applied once per
    // pair when the workgroup flushes (pairs with an unmapped beam are dropped there, map_angles :275).  Rows are
    // padded to 65 entries: consecutive BS beams at one UE beam would otherwise put a whole wave on one LDS bank.
    constexpr int CHUNK_LOG2 = pivot_chunk<T32>::log2, PIVOT_CHUNK = 1 << CHUNK_LOG2, SHIFT = 63 - CHUNK_LOG2;
    constexpr int64_t ROW_LIMIT = 1ll << (SHIFT - CHUNK_LOG2);
    constexpr int NH = SLAM_PIVOT_NH;      // groups of four entries per thread and step
    static_assert(PIVOT_CHUNK % (4096 * NH) == 0, "a step covers 4096 * NH entries");
    constexpr int LSTRIDE = N_BEAMS + 1;
    __shared__ unsigned long long acc[N_BEAMS * LSTRIDE];
    __shared__ int16_t b2a[N_BEAMS];
    const uint32_t f = blockIdx.y, tid = threadIdx.x;
    const uint32_t n = stats[f * SLAM_STAT_WORDS + stat_word];
    const uint32_t r0 = blockIdx.x * PIVOT_CHUNK;
    if (r0 >= n) return;
    // slotted rows: per ROWS_PER_BLOCK entries, or per tile where the fused kernel served the frame
    const uint32_t unit = (stats[f * SLAM_STAT_WORDS + SLAM_STAT_LAYOUT] & SLAM_LAYOUT_KEPT_TILES) ? (uint32_t)SLOT_ROWS : (uint32_t)ROWS_PER_BLOCK;
    const uint32_t r1 = min(n, r0 + PIVOT_CHUNK);
    for (int i = tid; i < N_BEAMS * LSTRIDE; i += 1024) acc[i] = 0;
    if (tid < N_BEAMS) b2a[tid] = beam_to_axis[tid];
    __syncthreads();
    const size_t fb = (size_t)f * row_cap;
    using V8 = typename v4<T8>::type;
    using V32 = typename v4<T32>::type;
    // NH groups of four entries per thread and step, loads first so all are in flight together (the slotted
    // layout leaves about a third of the lanes without rows, so each active one has to cover more latency)
    for (uint32_t base = r0 + tid * 4; base < r1; base += 4096u * NH) {
        uint32_t lim[NH];
        V8 u[NH], b[NH];
        V32 r[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const uint32_t i0 = base + 4096u * h;
            lim[h] = 0;
            if (i0 >= r1) continue;
            lim[h] = r1;                       // slotted rows: only the front of each block of entries is used
            if (slot_cnt) {
                const uint32_t T = i0 / unit;
                lim[h] = min(r1, T * unit + slot_cnt[(size_t)f * slot_stride + T]);
            }
            if (i0 >= lim[h]) continue;
            u[h] = *reinterpret_cast<const V8 *>(ue + fb + i0);
            b[h] = *reinterpret_cast<const V8 *>(bs + fb + i0);
            r[h] = *reinterpret_cast<const V32 *>(rss + fb + i0);
        }
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const uint32_t i0 = base + 4096u * h;
            if (i0 >= lim[h]) continue;
            const int64_t uu[4] = {(int64_t)u[h].x, (int64_t)u[h].y, (int64_t)u[h].z, (int64_t)u[h].w};
            const int64_t bb[4] = {(int64_t)b[h].x, (int64_t)b[h].y, (int64_t)b[h].z, (int64_t)b[h].w};
            const int64_t rr[4] = {(int64_t)r[h].x, (int64_t)r[h].y, (int64_t)r[h].z, (int64_t)r[h].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + k >= lim[h]) break;
                if (uu[k] < 0 || uu[k] > 63 || bb[k] < 0 || bb[k] > 63) continue;     // validate_beam_id (:68-72)
                if (rr[k] >= 0 && rr[k] < ROW_LIMIT) {
                    atomicAdd(&acc[bb[k] * LSTRIDE + uu[k]], (unsigned long long)rr[k] | (1ull << SHIFT));
                } else {                                                             // no room in the packed sum
                    const int c = b2a[uu[k]], rw = b2a[bb[k]];
                    if ((uint32_t)c >= (uint32_t)n_axis || (uint32_t)rw >= (uint32_t)n_axis) continue;
                    const size_t cell = (size_t)f * cell_stride + (beam_space ? (size_t)(bb[k] * N_BEAMS + uu[k]) : (size_t)(rw * n_axis + c));
                    atomicAdd(&gsum[cell], (unsigned long long)rr[k]);
                    atomicAdd(&gcnt[cell], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < N_BEAMS * N_BEAMS; i += 1024) {
        const int bb = i >> 6, uu = i & 63;
        const unsigned long long a = acc[bb * LSTRIDE + uu];
        if (!(a >> SHIFT)) continue;
        // unmapped beam (-1) dropped (:275); an index at or past n_axis cannot come from a valid table
        const int c = b2a[uu], rw = b2a[bb];
        if ((uint32_t)c >= (uint32_t)n_axis || (uint32_t)rw >= (uint32_t)n_axis) continue;
        const size_t cell = (size_t)f * cell_stride + (beam_space ? (size_t)i : (size_t)(rw * n_axis + c));
        atomicAdd(&gsum[cell], a & ((1ull << SHIFT) - 1));
        atomicAdd(&gcnt[cell], (uint32_t)(a >> SHIFT));
    }
}

// ---- pivot of real-valued RSS ---------------------------------------------------------------------------------
// pandas' pivot_table(aggfunc=mean) (compute_rssi_matrix, excel_heatmap_v3.py:297-299) sums every cell's values in
// row order with Kahan compensation and divides by the count; for values that are not integers the result depends
// on that order, so it is followed literally: k_cell_index gives every row its cell, a stable sort by cell
// (sort_cells) brings each cell's values together in row order, k_cell_bounds finds where each cell's run starts and
// ends, and one wave per cell adds its run up one value after the other.  Work grows with rows + cells, not with
// their product.
__global__ __launch_bounds__(256) void k_cell_index(const int32_t *ue, const int32_t *bs, const double *rss, uint32_t n,
                                                     const int16_t *beam_to_axis, int n_axis, uint32_t *cell)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t u = ue[i], b = bs[i];
    uint32_t c = (uint32_t)(n_axis * n_axis);                                 // no cell: sorts behind every cell
    if (u >= 0 && u <= 63 && b >= 0 && b <= 63 && !isnan(rss[i])) {          // validate_beam_id; NaN values are not counted
        const int cc = beam_to_axis[u], rr = beam_to_axis[b];
        if ((uint32_t)cc < (uint32_t)n_axis && (uint32_t)rr < (uint32_t)n_axis) c = (uint32_t)(rr * n_axis + cc);
    }
    cell[i] = c;
}

// sorted cells -> [begin, end) of every cell's run (cells without rows keep the zeros they were cleared to)
__global__ __launch_bounds__(256) void k_cell_bounds(const uint32_t *cell, uint32_t n, uint32_t cells, uint32_t *begin, uint32_t *end)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cell[i];
    if (c >= cells) return;
    if (i == 0 || cell[i - 1] != c) begin[c] = i;
    if (i + 1 == n || cell[i + 1] != c) end[c] = i + 1;
}

__global__ __launch_bounds__(256) void k_pivot_kahan(const double *val, const uint32_t *begin, const uint32_t *end, int cells,
                                                      double *sum, uint32_t *cnt, double *mean)
{
    const int mine = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (mine >= cells) return;
    const uint32_t lane = lane_id(), b0 = begin[mine], e0 = end[mine];
    double s = 0.0, comp = 0.0;
    for (uint32_t base = b0; base < e0; base += 64) {
        const uint32_t m = min(64u, e0 - base);
        const double v = lane < m ? val[base + lane] : 0.0;       // the cell's next values, in row order
        for (uint32_t j = 0; j < m; j++) {
            const double x = __longlong_as_double((long long)shfl64((uint64_t)__double_as_longlong(v), (int)j));
            const double y = x - comp;
            const double t = s + y;
            comp = (t - s) - y;
            if (comp != comp) comp = 0.0;                 // an infinite sum leaves no compensation
            s = t;
        }
    }
    if (lane == 0) {
        const uint32_t k = e0 - b0;
        if (sum) sum[mine] = s;
        if (cnt) cnt[mine] = k;
        mean[mine] = k ? s / (double)k : __longlong_as_double(0x7FF8000000000000ll);
    }
}

// ---- small-matrix device routines (one workgroup of 256 threads) --------------------------------------

__device__ __forceinline__ bool finite_d(double v) { return !(isnan(v) || isinf(v)); }

// Cell c of the normalised convolution of in (h x w, row stride w) with kern (ks x ks), edge padding
// (gaussian_blur_nan_aware :336-368, taps in the reference's row-major order).
__device__ __forceinline__ double blur_cell(const double *in, int h, int w, const double *kern, int ks, int c)
{
    const int p = ks / 2;
    const int i = c / w, j = c % w;
    double num = 0.0, den = 0.0;
    for (int a = 0; a < ks; a++) {
        int ii = i + a - p;
        ii = ii < 0 ? 0 : (ii >= h ? h - 1 : ii);
        for (int b = 0; b < ks; b++) {
            int jj = j + b - p;
            jj = jj < 0 ? 0 : (jj >= w ? w - 1 : jj);
            const double v = in[ii * w + jj];
            const bool fin = finite_d(v);
            const double wgt = kern[a * ks + b] * (fin ? 1.0 : 0.0);
            num += (fin ? v : 0.0) * wgt;
            den += wgt;
        }
    }
    return den > 1e-12 ? num / den : __longlong_as_double(0x7FF8000000000000ll);
}

// the whole matrix by one workgroup
__device__ void blur_block(const double *in, int h, int w, const double *kern, int ks, double *out)
{
    for (int c = threadIdx.x; c < h * w; c += blockDim.x) out[c] = blur_cell(in, h, w, kern, ks, c);
}

// finite min / max of m[0..n) across the block (blockDim.x a power of two <= 1024); sh needs 2*blockDim.x
// doubles.  Returns false when none is finite.
__device__ bool block_minmax(const double *m, int n, double *sh, double *mn, double *mx)
{
    const int T = blockDim.x;
    double lo = INFINITY, hi = -INFINITY;
    for (int c = threadIdx.x; c < n; c += T) {
        double v = m[c];
        if (finite_d(v)) { lo = fmin(lo, v); hi = fmax(hi, v); }
    }
    sh[threadIdx.x] = lo;
    sh[T + threadIdx.x] = hi;
    __syncthreads();
    for (int s = T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[threadIdx.x] = fmin(sh[threadIdx.x], sh[threadIdx.x + s]);
            sh[T + threadIdx.x] = fmax(sh[T + threadIdx.x], sh[T + threadIdx.x + s]);
        }
        __syncthreads();
    }
    *mn = sh[0];
    *mx = sh[T];
    __syncthreads();
    return *mn <= *mx;
}

// _edges (:430-440) of v[0..n) into e[0..n]
__device__ void edges_of(const double *v, int n, double *e)
{
    if (n == 1) { e[0] = v[0] - 0.5; e[1] = v[0] + 0.5; return; }
    for (int i = 1; i < n; i++) e[i] = (v[i - 1] + v[i]) / 2.0;
    e[0] = v[0] - (v[1] - v[0]) / 2.0;
    e[n] = v[n - 1] + (v[n - 1] - v[n - 2]) / 2.0;
}

// pixel -> cell index along one axis; flip = true samples from the top edge downwards
__device__ void axis_map(const double *e, int n, int npix, bool flip, int16_t *map)
{
    const double step = (e[n] - e[0]) / (double)npix;
    for (int p = threadIdx.x; p < npix; p += blockDim.x) {
        const double t = ((double)p + 0.5) * step;
        const double x = flip ? e[n] - t : e[0] + t;
        int lo = 0, hi = n + 1;                 // first edge > x
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (e[mid] <= x) lo = mid + 1; else hi = mid;
        }
        int c = lo - 1;
        c = c < 0 ? 0 : (c > n - 1 ? n - 1 : c);
        map[p] = (int16_t)c;
    }
}

// scale + colour of a compact h x w matrix `m` (row stride w) -> cells RGBA; returns limits used.
// Limits of the colour scale as the reference's figure ends up with them (excel_heatmap_v3.py:409-423, :449-457).
// Linear scale: pcolormesh gets no norm and no limits, so vmin / vmax given by the caller play no part and the scale
// spans the finite data.  Log scale: LogNorm over the data shifted by (1e-6 - min), given limits shifted alike;
// a limit that is not positive after the shift makes the norm fail ("Invalid vmin or vmax").  The colour bar then
// passes the limits through matplotlib's nonsingular(expander=0.1): an empty interval grows by a tenth of each end
// (or to +-0.1 around zero), and limits given in reverse come out as (0.9 * smaller, larger) -- the smaller end is
// expanded while both ends momentarily hold it.  Returns false for the invalid case.
__device__ bool scale_limits(int use_log, double vmin, double vmax, double mn, double mx, double *lo_out, double *hi_out)
{
    double lo, hi;
    if (use_log) {
        lo = isnan(vmin) ? (mn - mn + 1e-6) : (vmin - mn + 1e-6);
        hi = isnan(vmax) ? (mx - mn + 1e-6) : (vmax - mn + 1e-6);
        if (!(lo > 0.0) || !(hi > 0.0) || isinf(lo) || isinf(hi)) return false;
    } else {
        lo = mn;
        hi = mx;
    }
    if (hi < lo) {
        const double t = lo;
        lo = hi - 0.1 * fabs(hi);
        hi = t;
    } else {
        const double big = fmax(fabs(lo), fabs(hi));
        if (big < 2.2250738585072014e-287) { lo = -0.1; hi = 0.1; }
        else if (hi - lo <= big * 1e-15) { lo -= 0.1 * fabs(lo); hi += 0.1 * fabs(hi); }
    }
    *lo_out = lo;
    *hi_out = hi;
    return true;
}

// Cell colours for limits from scale_limits: index = floor(norm * 256) clipped to the table, holes transparent.
struct colour_scale { int use_log; double mn, lo, tlo, span; };

__device__ __forceinline__ colour_scale make_scale(int use_log, double mn, double lo, double hi)
{
    colour_scale s{use_log, mn, lo, 0.0, 0.0};
    if (use_log) {
        s.tlo = log10(lo);
        s.span = log10(hi) - s.tlo;
    } else {
        s.span = hi - lo;
    }
    return s;
}

__device__ __forceinline__ uchar4 colour_of(double v, const colour_scale &s, const uint8_t *lut)
{
    if (!finite_d(v)) return make_uchar4(255, 255, 255, 0);      // set_bad((1,1,1,0)) (:447)
    double nrm;
    if (s.use_log) nrm = (log10(v - s.mn + 1e-6) - s.tlo) / s.span;
    else nrm = (v - s.lo) / s.span;
    const double x = floor(nrm * 256.0);
    const int idx = isnan(x) ? 0 : (x < 0.0 ? 0 : (x > 255.0 ? 255 : (int)x));
    return make_uchar4(lut[idx * 3], lut[idx * 3 + 1], lut[idx * 3 + 2], 255);
}

__device__ void colour_block(const double *m, int h, int w, int use_log, double mn, double lo, double hi,
                             const uint8_t *lut, uint8_t *cells)
{
    const colour_scale s = make_scale(use_log, mn, lo, hi);
    for (int c = threadIdx.x; c < h * w; c += blockDim.x) reinterpret_cast<uchar4 *>(cells)[c] = colour_of(m[c], s, lut);
}

__device__ __forceinline__ void put_f64_stat(uint32_t *stats, int lo_word, double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    stats[lo_word] = (uint32_t)u;
    stats[lo_word + 1] = (uint32_t)(u >> 32);
}

// One workgroup per frame: sum/cnt -> mean grid -> compact -> blur -> scale -> cells + pixel maps.
__global__ __launch_bounds__(1024) void k_frame_matrix(render_params rp, render_ws ws, const double *kern, int ks,
                                                       double *matrix, double *blurred, uint32_t *stats)
{
    __shared__ double cm[MAX_CELLS];
    __shared__ double bm[MAX_CELLS];              // 64 KiB for the two grids: two workgroups fit a CU's LDS
    __shared__ double ang_r[MAX_AXIS], ang_c[MAX_AXIS], edge[MAX_AXIS + 1];
    __shared__ uint32_t rowp[MAX_AXIS], colp[MAX_AXIS];
    __shared__ int ridx[MAX_AXIS], cidx[MAX_AXIS];
    __shared__ int s_h, s_w;
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const int na = rp.n_axis;
    const unsigned long long *sum = ws.sum + (size_t)f * ws.cell_stride;
    const uint32_t *cnt = ws.cnt + (size_t)f * ws.cell_stride;
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    // full_grid: every entry of the axis table is a row and a column whether any row of data fell on it or not
    if (tid < MAX_AXIS) { rowp[tid] = rp.full_grid ? 1u : 0u; colp[tid] = rp.full_grid ? 1u : 0u; }
    __syncthreads();
    for (int c = tid; c < na * na; c += 1024) {
        const uint32_t k = cnt[c];
        const double mean = k ? (double)(long long)sum[c] / (double)k : __longlong_as_double(0x7FF8000000000000ll);
        if (matrix) matrix[(size_t)f * na * na + c] = mean;
        if (k) { rowp[c / na] = 1; colp[c % na] = 1; }
    }
    __syncthreads();
    if (tid < 64) {                                        // compact index lists of the axes that occur
        const bool rp_ = (int)tid < na && rowp[tid], cp_ = (int)tid < na && colp[tid];
        const unsigned long long rm = __ballot(rp_), cmk = __ballot(cp_);
        const unsigned long long below = (1ull << tid) - 1;
        if (rp_) { int k = __popcll(rm & below); ridx[k] = tid; ang_r[k] = rp.axis_angle[tid]; }
        if (cp_) { int k = __popcll(cmk & below); cidx[k] = tid; ang_c[k] = rp.axis_angle[tid]; }
        if (tid == 0) { s_h = __popcll(rm); s_w = __popcll(cmk); }
    }
    __syncthreads();
    const int h = s_h, w = s_w;
    if (tid == 0) { st[SLAM_STAT_N_AOD] = h; st[SLAM_STAT_N_AOA] = w; }
    if (h == 0 || w == 0) {
        if (tid == 0) st[SLAM_STAT_ALLNAN] = 1;
        return;
    }
    for (int c = tid; c < h * w; c += 1024) {
        const int src = ridx[c / w] * na + cidx[c % w];
        const uint32_t k = cnt[src];
        cm[c] = k ? (double)(long long)sum[src] / (double)k : __longlong_as_double(0x7FF8000000000000ll);
    }
    __syncthreads();
    const double *fin = cm;
    if (ks > 1) {
        blur_block(cm, h, w, kern, ks, bm);
        __syncthreads();
        fin = bm;
    }
    if (blurred)
        for (int c = tid; c < h * w; c += 1024) blurred[(size_t)f * na * na + c] = fin[c];
    double mn, mx;
    __syncthreads();
    const bool any = block_minmax(fin, h * w, fin == cm ? bm : cm, &mn, &mx);     // the grid not in use is the scratch
    uint8_t *cells = ws.cells + (size_t)f * ws.cell_stride * 4;
    if (!any) {
        if (tid == 0) st[SLAM_STAT_ALLNAN] = 1;
        return;
    }
    double lo, hi;
    if (!scale_limits(rp.use_log, rp.vmin, rp.vmax, mn, mx, &lo, &hi)) {
        if (tid == 0) st[SLAM_STAT_ALLNAN] = 2;
        return;
    }
    colour_block(fin, h, w, rp.use_log, mn, lo, hi, rp.lut, cells);
    if (tid == 0) {
        put_f64_stat(st, SLAM_STAT_VMIN_LO, lo);
        put_f64_stat(st, SLAM_STAT_VMAX_LO, hi);
        st[SLAM_STAT_ALLNAN] = 0;                  // a render that succeeds clears what an earlier one on these words left
    }
    if (tid == 0) edges_of(ang_c, w, edge);
    __syncthreads();
    axis_map(edge, w, rp.W, false, ws.colmap + (size_t)f * rp.W);
    __syncthreads();
    if (tid == 0) edges_of(ang_r, h, edge);
    __syncthreads();
    axis_map(edge, h, rp.H, true, ws.rowmap + (size_t)f * rp.H);
}

// tile[py][px] = cells[rowmap[py]][colmap[px]].  A thread owns 4 pixel columns (16 bytes per store) of a band of
// RASTER_BAND pixel rows: its cell columns are looked up once, and the four colours are fetched again only when the
// band moves to another cell row (a cell covers many pixel rows), so the kernel is a stream of 16-byte stores.
constexpr int RASTER_BAND = 16;
__global__ __launch_bounds__(256) void k_raster(const uint8_t *cells, const int16_t *colmap, const int16_t *rowmap,
                                                 int W, int H, const uint32_t *stats, uint8_t *tile, uint32_t cell_stride)
{
    const uint32_t f = blockIdx.z;
    const uint32_t px0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if ((int)px0 >= W) return;
    const uint32_t y0 = blockIdx.y * RASTER_BAND, y1 = min((uint32_t)H, y0 + RASTER_BAND);
    const uint32_t *st = stats + f * SLAM_STAT_WORDS;
    const bool blank = st[SLAM_STAT_ALLNAN] != 0;
    const int w = (int)st[SLAM_STAT_N_AOA];
    const uint32_t *c4 = reinterpret_cast<const uint32_t *>(cells) + (size_t)f * cell_stride;
    int col[4] = {0, 0, 0, 0};
    const int npx = min(4, W - (int)px0);
    if (!blank)
        for (int k = 0; k < npx; k++) col[k] = colmap[(size_t)f * W + px0 + k];
    const uint32_t none = 0x00FFFFFFu;                          // (255, 255, 255, 0): transparent white
    uint32_t out[4] = {none, none, none, none};
    int cur = -1;
    for (uint32_t py = y0; py < y1; py++) {
        if (!blank) {
            const int row = rowmap[(size_t)f * H + py];           // uniform over the workgroup
            if (row != cur) {
                cur = row;
                for (int k = 0; k < npx; k++) out[k] = c4[row * w + col[k]];
            }
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(tile) + ((size_t)f * H + py) * W + px0;
        if (npx == 4 && (W & 3) == 0) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
            for (int k = 0; k < npx; k++) dst[k] = out[k];
        }
    }
}

// ---- wide axis tables (65 .. 256 entries): the same steps as k_frame_matrix with the grids in global memory ------
// k_wide_mean -> k_wide_axes -> k_wide_compact -> k_wide_blur -> k_wide_colour; all but k_wide_axes run one cell per
// thread over grid (cells / 256, frames).

// finite doubles <-> u64 keys that order like the values (for atomicMin / atomicMax)
__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

// beam-pair accumulators (64 x 64 per frame, as the pivot leaves them for a wide table) onto the axis grid: one
// pass per frame; beams that share an axis entry add up there
__global__ __launch_bounds__(256) void k_wide_scatter(render_params rp, render_ws ws)
{
    const uint32_t f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;          // i = bs * 64 + ue
    const uint32_t k = ws.cnt[(size_t)f * MAX_CELLS + i];
    if (!k) return;
    const int rw = rp.beam_to_axis[i >> 6], c = rp.beam_to_axis[i & 63];
    if ((uint32_t)c >= (uint32_t)rp.n_axis || (uint32_t)rw >= (uint32_t)rp.n_axis) return;
    const size_t cell = (size_t)f * ws.cell_stride + (size_t)rw * rp.n_axis + c;
    atomicAdd(&ws.sum_w[cell], ws.sum[(size_t)f * MAX_CELLS + i]);
    atomicAdd(&ws.cnt_w[cell], k);
}

__global__ __launch_bounds__(256) void k_wide_mean(render_params rp, render_ws ws, double *matrix)
{
    const uint32_t f = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t na = (uint32_t)rp.n_axis;
    if (c >= na * na) return;
    const uint32_t k = ws.cnt_w[(size_t)f * ws.cell_stride + c];
    const double mean = k ? (double)(long long)ws.sum_w[(size_t)f * ws.cell_stride + c] / (double)k
                          : __longlong_as_double(0x7FF8000000000000ll);
    // with the full grid the compact matrix is the grid itself: it goes straight to where the blur reads
    (rp.full_grid ? ws.grid_a : ws.grid_b)[(size_t)f * ws.cell_stride + c] = mean;
    if (matrix) matrix[(size_t)f * na * na + c] = mean;
    if (k) {                                       // every writer stores the same value
        ws.present[(size_t)f * 2 * WIDE_AXIS + c / na] = 1u;
        ws.present[(size_t)f * 2 * WIDE_AXIS + WIDE_AXIS + c % na] = 1u;
    }
}

// One workgroup of 256 threads per frame: the axes that occur (or all of them), their index lists, _edges and the
// pixel -> cell maps; clears the frame's min / max keys.
__global__ __launch_bounds__(256) void k_wide_axes(render_params rp, render_ws ws, uint32_t *stats)
{
    __shared__ double ang_r[WIDE_AXIS], ang_c[WIDE_AXIS], edge[WIDE_AXIS + 1];
    __shared__ uint32_t sh[17];
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const int na = rp.n_axis;
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    const uint32_t *pres = ws.present + (size_t)f * 2 * WIDE_AXIS;
    int16_t *ridx = ws.axis_idx + (size_t)f * 2 * WIDE_AXIS, *cidx = ridx + WIDE_AXIS;
    const bool rp_ = (int)tid < na && (rp.full_grid || pres[tid]);
    const bool cp_ = (int)tid < na && (rp.full_grid || pres[WIDE_AXIS + tid]);
    uint32_t h, w;
    const uint32_t kr = block_excl_scan(rp_ ? 1u : 0u, sh, &h);
    const uint32_t kc = block_excl_scan(cp_ ? 1u : 0u, sh, &w);
    if (rp_) { ridx[kr] = (int16_t)tid; ang_r[kr] = rp.axis_angle[tid]; }
    if (cp_) { cidx[kc] = (int16_t)tid; ang_c[kc] = rp.axis_angle[tid]; }
    if (tid == 0) {
        st[SLAM_STAT_N_AOD] = h;
        st[SLAM_STAT_N_AOA] = w;
        st[SLAM_STAT_ALLNAN] = (h == 0 || w == 0) ? 1u : 0u;
    }
    if (h == 0 || w == 0) return;
    __syncthreads();
    if (tid == 0) edges_of(ang_c, (int)w, edge);
    __syncthreads();
    axis_map(edge, (int)w, rp.W, false, ws.colmap + (size_t)f * rp.W);
    __syncthreads();
    if (tid == 0) edges_of(ang_r, (int)h, edge);
    __syncthreads();
    axis_map(edge, (int)h, rp.H, true, ws.rowmap + (size_t)f * rp.H);
}

// compact mean matrix (h x w, row stride w) out of the full grid
__global__ __launch_bounds__(256) void k_wide_compact(render_params rp, render_ws ws, const uint32_t *stats)
{
    const uint32_t f = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t h = stats[f * SLAM_STAT_WORDS + SLAM_STAT_N_AOD], w = stats[f * SLAM_STAT_WORDS + SLAM_STAT_N_AOA];
    if (c >= h * w) return;
    const int16_t *ridx = ws.axis_idx + (size_t)f * 2 * WIDE_AXIS, *cidx = ridx + WIDE_AXIS;
    ws.grid_a[(size_t)f * ws.cell_stride + c] =
        ws.grid_b[(size_t)f * ws.cell_stride + (uint32_t)ridx[c / w] * (uint32_t)rp.n_axis + (uint32_t)cidx[c % w]];
}

// blur (or copy, ks <= 1) of the compact matrix, the finite minimum / maximum of the result.  A workgroup owns a
// 16 x 16 piece of the matrix; for kernels of up to WIDE_BLUR_KS taps a side it stages the piece with its halo (edge
// padding applied) and the weights in LDS and sums the taps in the same order as blur_cell.
constexpr int WIDE_BLUR_KS = 15, WIDE_BLUR_T = 16, WIDE_BLUR_TW = WIDE_BLUR_T + WIDE_BLUR_KS - 1;
constexpr int WIDE_BLUR_SLOTS = (WIDE_AXIS / WIDE_BLUR_T) * (WIDE_AXIS / WIDE_BLUR_T);       // pieces per frame at most
__global__ __launch_bounds__(256) void k_wide_blur(render_params rp, render_ws ws, const double *kern, int ks,
                                                    double *blurred, const uint32_t *stats)
{
    __shared__ double tile[WIDE_BLUR_TW * WIDE_BLUR_TW];
    __shared__ double wgt[WIDE_BLUR_KS * WIDE_BLUR_KS];
    const uint32_t f = blockIdx.z;
    const int h = (int)stats[f * SLAM_STAT_WORDS + SLAM_STAT_N_AOD], w = (int)stats[f * SLAM_STAT_WORDS + SLAM_STAT_N_AOA];
    __shared__ unsigned long long part[4][2];
    const int i0 = blockIdx.y * WIDE_BLUR_T, j0 = blockIdx.x * WIDE_BLUR_T;
    // every workgroup leaves the minimum / maximum of its piece (keys; none: ~0, 0) in its own slot: no atomics on a
    // word the whole frame shares
    unsigned long long *slot = ws.minmax + ((size_t)f * WIDE_BLUR_SLOTS + blockIdx.y * gridDim.x + blockIdx.x) * 2;
    if (i0 >= h || j0 >= w) {                                       // whole workgroup past the matrix
        if (threadIdx.x == 0) { slot[0] = ~0ull; slot[1] = 0ull; }
        return;
    }
    const double *in = ws.grid_a + (size_t)f * ws.cell_stride;
    const int li = threadIdx.x / WIDE_BLUR_T, lj = threadIdx.x % WIDE_BLUR_T, i = i0 + li, j = j0 + lj;
    const bool mine = i < h && j < w;
    const int c = i * w + j;
    double v = 0.0;
    if (ks > 1 && ks <= WIDE_BLUR_KS) {
        const int p = ks / 2, tw = WIDE_BLUR_T + ks - 1;
        for (int t = threadIdx.x; t < tw * tw; t += 256) {
            int ii = i0 - p + t / tw, jj = j0 - p + t % tw;
            ii = ii < 0 ? 0 : (ii >= h ? h - 1 : ii);
            jj = jj < 0 ? 0 : (jj >= w ? w - 1 : jj);
            tile[t] = in[ii * w + jj];
        }
        for (int t = threadIdx.x; t < ks * ks; t += 256) wgt[t] = kern[t];
        __syncthreads();
        double num = 0.0, den = 0.0;
        for (int a = 0; a < ks; a++)
            for (int b = 0; b < ks; b++) {
                const double x = tile[(li + a) * tw + lj + b];
                const bool fin = finite_d(x);
                const double wt = wgt[a * ks + b] * (fin ? 1.0 : 0.0);
                num += (fin ? x : 0.0) * wt;
                den += wt;
            }
        v = den > 1e-12 ? num / den : __longlong_as_double(0x7FF8000000000000ll);
    } else if (mine) {
        v = ks > 1 ? blur_cell(in, h, w, kern, ks, c) : in[c];
    }
    unsigned long long lo = ~0ull, hi = 0ull;
    if (mine) {
        ws.grid_b[(size_t)f * ws.cell_stride + c] = v;
        if (blurred) blurred[(size_t)f * rp.n_axis * rp.n_axis + c] = v;
        if (finite_d(v)) lo = hi = order_key(v);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long ol = shfl64(lo, (int)(lane_id() ^ d)), oh = shfl64(hi, (int)(lane_id() ^ d));
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if (lane_id() == 0) { part[threadIdx.x >> 6][0] = lo; part[threadIdx.x >> 6][1] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) {
            lo = part[k][0] < lo ? part[k][0] : lo;
            hi = part[k][1] > hi ? part[k][1] : hi;
        }
        slot[0] = lo;
        slot[1] = hi;
    }
}

__global__ __launch_bounds__(256) void k_wide_colour(render_params rp, render_ws ws, uint32_t *stats)
{
    const uint32_t f = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    uint32_t *st = stats + f * SLAM_STAT_WORDS;
    const uint32_t h = st[SLAM_STAT_N_AOD], w = st[SLAM_STAT_N_AOA];
    if (h == 0 || w == 0) return;
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    // the frame's minimum / maximum out of the blur's per-piece slots (every workgroup reduces them for itself)
    __shared__ unsigned long long red[4][2];
    const uint32_t bt = ((uint32_t)rp.n_axis + WIDE_BLUR_T - 1) / WIDE_BLUR_T, nslot = bt * bt;
    unsigned long long klo = ~0ull, khi = 0ull;
    if (threadIdx.x < nslot) {
        klo = ws.minmax[((size_t)f * WIDE_BLUR_SLOTS + threadIdx.x) * 2];
        khi = ws.minmax[((size_t)f * WIDE_BLUR_SLOTS + threadIdx.x) * 2 + 1];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long ol = shfl64(klo, (int)(lane_id() ^ d)), oh = shfl64(khi, (int)(lane_id() ^ d));
        klo = ol < klo ? ol : klo;
        khi = oh > khi ? oh : khi;
    }
    if (lane_id() == 0) { red[threadIdx.x >> 6][0] = klo; red[threadIdx.x >> 6][1] = khi; }
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        klo = red[k][0] < klo ? red[k][0] : klo;
        khi = red[k][1] > khi ? red[k][1] : khi;
    }
    if (klo > khi) {                                               // no finite cell
        if (first) st[SLAM_STAT_ALLNAN] = 1;
        return;
    }
    const double mn = key_value(klo), mx = key_value(khi);
    double lo, hi;
    if (!scale_limits(rp.use_log, rp.vmin, rp.vmax, mn, mx, &lo, &hi)) {
        if (first) st[SLAM_STAT_ALLNAN] = 2;
        return;
    }
    if (first) { put_f64_stat(st, SLAM_STAT_VMIN_LO, lo); put_f64_stat(st, SLAM_STAT_VMAX_LO, hi); }
    if (c >= h * w) return;
    const colour_scale s = make_scale(rp.use_log, mn, lo, hi);
    reinterpret_cast<uchar4 *>(ws.cells + (size_t)f * ws.cell_stride * 4)[c] =
        colour_of(ws.grid_b[(size_t)f * ws.cell_stride + c], s, rp.lut);
}

// ---- standalone stages on caller matrices ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur_only(const double *in, int h, int w, const double *kern, int ks,
                                                    double *out)
{
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;         // one cell per thread over the whole grid
    if (c < (size_t)h * w) out[c] = blur_cell(in, h, w, kern, ks, (int)c);
}

__global__ __launch_bounds__(256) void k_mean_only(const unsigned long long *sum, const uint32_t *cnt, int cells,
                                                    double *matrix)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cells) return;
    const uint32_t k = cnt[c];
    matrix[c] = k ? (double)(long long)sum[c] / (double)k : __longlong_as_double(0x7FF8000000000000ll);
}

__global__ __launch_bounds__(256) void k_render_only(const double *mat, int h, int w, const double *aod,
                                                      const double *aoa, int use_log, double vmin, double vmax,
                                                      const uint8_t *lut, uint8_t *cells, int16_t *colmap,
                                                      int16_t *rowmap, int W, int H, uint32_t *stats)
{
    __shared__ double red[512];
    __shared__ double edge[WIDE_AXIS + 1];
    const uint32_t tid = threadIdx.x;
    double mn, mx;
    if (tid == 0) { stats[SLAM_STAT_N_AOD] = h; stats[SLAM_STAT_N_AOA] = w; stats[SLAM_STAT_ALLNAN] = 0; }
    const bool any = block_minmax(mat, h * w, red, &mn, &mx);
    if (!any) {
        if (tid == 0) stats[SLAM_STAT_ALLNAN] = 1;
        return;
    }
    double lo, hi;
    if (!scale_limits(use_log, vmin, vmax, mn, mx, &lo, &hi)) {
        if (tid == 0) stats[SLAM_STAT_ALLNAN] = 2;
        return;
    }
    colour_block(mat, h, w, use_log, mn, lo, hi, lut, cells);
    if (tid == 0) { put_f64_stat(stats, SLAM_STAT_VMIN_LO, lo); put_f64_stat(stats, SLAM_STAT_VMAX_LO, hi); }
    if (tid == 0) edges_of(aoa, w, edge);
    __syncthreads();
    axis_map(edge, w, W, false, colmap);
    __syncthreads();
    if (tid == 0) edges_of(aod, h, edge);
    __syncthreads();
    axis_map(edge, h, H, true, rowmap);
}

}  // namespace

extern "C" uint32_t slam_cells_per_frame(int32_t n_axis)
{
    return n_axis <= SMALL_AXIS ? (uint32_t)MAX_CELLS : (uint32_t)n_axis * (uint32_t)n_axis;
}

void render_ws::plan(ws_plan &p, uint32_t F, int W, int H, int n_axis, size_t off[12])
{
    const size_t cells = slam_cells_per_frame(n_axis);
    const bool wide = n_axis > SMALL_AXIS;
    off[0] = p.add((size_t)F * (wide ? MAX_CELLS : cells) * 8);      // a wide table: beam pairs here, the axis grid in sum_w
    off[1] = p.add((size_t)F * (wide ? MAX_CELLS : cells) * 4);
    off[2] = p.add((size_t)F * cells * 4);
    off[3] = p.add((size_t)F * (size_t)(W > 0 ? W : 1) * 2);
    off[4] = p.add((size_t)F * (size_t)(H > 0 ? H : 1) * 2);
    off[5] = p.add(wide ? (size_t)F * cells * 8 : 0);
    off[6] = p.add(wide ? (size_t)F * cells * 8 : 0);
    off[7] = p.add(wide ? (size_t)F * 2 * WIDE_AXIS * 4 : 0);
    off[8] = p.add(wide ? (size_t)F * 2 * WIDE_AXIS * 2 : 0);
    off[9] = p.add(wide ? (size_t)F * 2 * 8 * 256 : 0);            // a minimum / maximum slot per 16 x 16 piece
    off[10] = p.add(wide ? (size_t)F * cells * 8 : 0);
    off[11] = p.add(wide ? (size_t)F * cells * 4 : 0);
}

void render_ws::bind(uint8_t *base, const size_t off[12], int n_axis)
{
    sum = (unsigned long long *)(base + off[0]);
    cnt = (uint32_t *)(base + off[1]);
    cells = base + off[2];
    colmap = (int16_t *)(base + off[3]);
    rowmap = (int16_t *)(base + off[4]);
    grid_a = (double *)(base + off[5]);
    grid_b = (double *)(base + off[6]);
    present = (uint32_t *)(base + off[7]);
    axis_idx = (int16_t *)(base + off[8]);
    minmax = (unsigned long long *)(base + off[9]);
    sum_w = (unsigned long long *)(base + off[10]);
    cnt_w = (uint32_t *)(base + off[11]);
    cell_stride = slam_cells_per_frame(n_axis);
}

template <typename T8, typename T32>
int pivot_accumulate(slam_ctx *ctx, const T8 *ue, const T8 *bs, const T32 *rss, uint32_t row_cap, uint32_t F,
                     const uint32_t *stats, int stat_word, const render_params &rp, const render_ws &ws,
                     const uint32_t *slot_cnt, uint32_t slot_stride)
{
    if (F == 0) return SLAM_OK;
    if (rp.n_axis < 1 || rp.n_axis > WIDE_AXIS) return slam_fail(ctx, SLAM_E_INVALID, "n_axis must be 1..%d", WIDE_AXIS);
    const bool wide = rp.n_axis > SMALL_AXIS;
    const uint32_t stride = wide ? (uint32_t)MAX_CELLS : ws.cell_stride;
    SLAM_TRY(slam_zero_async(ctx, ws.sum, (size_t)F * stride * 8));
    SLAM_TRY(slam_zero_async(ctx, ws.cnt, (size_t)F * stride * 4));
    constexpr uint32_t chunk = 1u << pivot_chunk<T32>::log2;
    dim3 grid((row_cap + chunk - 1) / chunk, F), block(1024);
    SLAM_LAUNCH(ctx, "pivot", (k_pivot<T8, T32>), grid, block, ue, bs, rss, row_cap, stats, stat_word,
                rp.beam_to_axis, rp.n_axis, stride, ws.sum, ws.cnt, slot_cnt, slot_stride, wide ? 1 : 0);
    if (wide) {          // beam pairs -> axis grid, once per frame
        SLAM_TRY(slam_zero_async(ctx, ws.sum_w, (size_t)F * ws.cell_stride * 8));
        SLAM_TRY(slam_zero_async(ctx, ws.cnt_w, (size_t)F * ws.cell_stride * 4));
        SLAM_LAUNCH(ctx, "wide_scatter", k_wide_scatter, dim3(MAX_CELLS / 256, F), dim3(256), rp, ws);
    }
    return SLAM_OK;
}

template int pivot_accumulate<uint8_t, uint32_t>(slam_ctx *, const uint8_t *, const uint8_t *, const uint32_t *,
                                                 uint32_t, uint32_t, const uint32_t *, int, const render_params &,
                                                 const render_ws &, const uint32_t *, uint32_t);
template int pivot_accumulate<int32_t, int32_t>(slam_ctx *, const int32_t *, const int32_t *, const int32_t *,
                                                uint32_t, uint32_t, const uint32_t *, int, const render_params &,
                                                const render_ws &, const uint32_t *, uint32_t);

// Gaussian kernel of gaussian_kernel (:320-333), built on the host into `k` (size*size doubles); returns size.
int slam_gaussian_kernel_host(double sigma, std::vector<double> &k)
{
    if (sigma <= 0) { k.assign(1, 1.0); return 1; }
    int size = (int)fmax(3.0, ceil(6.0 * sigma));
    if (size % 2 == 0) size++;
    const int c = size / 2;
    k.resize((size_t)size * size);
    double total = 0.0;
    for (int y = -c; y <= c; y++)
        for (int x = -c; x <= c; x++) {
            double v = exp(-(double)(x * x + y * y) / (2.0 * sigma * sigma));
            k[(size_t)(y + c) * size + (x + c)] = v;
            total += v;
        }
    for (auto &v : k) v /= total;
    return size;
}

static dim3 raster_grid(int W, int H, uint32_t F)
{
    return dim3((uint32_t)((W + 1023) / 1024), (uint32_t)((H + RASTER_BAND - 1) / RASTER_BAND), F);
}

int matrix_blur_render(slam_ctx *ctx, uint32_t F, const render_params &rp, const render_ws &ws, const double *kern,
                       int ks, double *matrix, double *blurred, uint8_t *tile, uint32_t *stats)
{
    if (F == 0) return SLAM_OK;
    if (rp.n_axis <= SMALL_AXIS) {
        SLAM_LAUNCH(ctx, "frame_matrix", k_frame_matrix, dim3(F), dim3(1024), rp, ws, kern, ks, matrix, blurred, stats);
    } else {
        const dim3 cgrid((uint32_t)(rp.n_axis * rp.n_axis + 255) / 256, F), blk(256);
        if (!rp.full_grid) SLAM_TRY(slam_zero_async(ctx, ws.present, (size_t)F * 2 * WIDE_AXIS * 4));
        SLAM_LAUNCH(ctx, "wide_mean", k_wide_mean, cgrid, blk, rp, ws, matrix);
        SLAM_LAUNCH(ctx, "wide_axes", k_wide_axes, dim3(F), blk, rp, ws, stats);
        if (!rp.full_grid) SLAM_LAUNCH(ctx, "wide_compact", k_wide_compact, cgrid, blk, rp, ws, stats);
        const uint32_t bt = (uint32_t)(rp.n_axis + WIDE_BLUR_T - 1) / WIDE_BLUR_T;
        SLAM_LAUNCH(ctx, "wide_blur", k_wide_blur, dim3(bt, bt, F), blk, rp, ws, kern, ks, blurred, stats);
        SLAM_LAUNCH(ctx, "wide_colour", k_wide_colour, cgrid, blk, rp, ws, stats);
    }
    if (tile)
        SLAM_LAUNCH(ctx, "raster", k_raster, raster_grid(rp.W, rp.H, F), dim3(256), ws.cells, ws.colmap, ws.rowmap, rp.W,
                    rp.H, stats, tile, ws.cell_stride);
    return SLAM_OK;
}

int raster_only(slam_ctx *ctx, const uint8_t *cells, const int16_t *colmap, const int16_t *rowmap, const uint32_t *stats,
                uint32_t F, int W, int H, uint8_t *tile, uint32_t cell_stride)
{
    if (F == 0) return SLAM_OK;
    SLAM_LAUNCH(ctx, "raster", k_raster, raster_grid(W, H, F), dim3(256), cells, colmap, rowmap, W, H, stats, tile,
                cell_stride);
    return SLAM_OK;
}

size_t pivot_kahan_ws::plan(ws_plan &p, uint32_t n, int n_axis)
{
    const size_t cells = (size_t)n_axis * n_axis;
    unsigned b = 1;
    while ((1ull << b) <= cells) b++;                     // keys 0 .. cells (the last one: no cell)
    bits = b;
    o_key = p.add((size_t)n * 4 + 4);
    o_key2 = p.add((size_t)n * 4 + 4);
    o_val2 = p.add((size_t)n * 8 + 8);
    o_bounds = p.add(cells * 2 * 4);
    tmp_bytes = n ? sort_cells_temp_bytes(n, bits) : 0;
    o_tmp = p.add(tmp_bytes + 16);
    return p.total;
}

int pivot_kahan(slam_ctx *ctx, const int32_t *ue, const int32_t *bs, const double *rss, uint32_t n,
                const int16_t *beam_to_axis, int n_axis, uint8_t *ws, const pivot_kahan_ws &w, double *sum, uint32_t *cnt,
                double *mean)
{
    const int cells = n_axis * n_axis;
    uint32_t *key = (uint32_t *)(ws + w.o_key), *key2 = (uint32_t *)(ws + w.o_key2), *begin = (uint32_t *)(ws + w.o_bounds),
             *end = begin + cells;
    double *val2 = (double *)(ws + w.o_val2);
    SLAM_TRY(slam_zero_async(ctx, begin, (size_t)cells * 2 * 4));
    if (n) {
        SLAM_LAUNCH(ctx, "cell_index", k_cell_index, dim3((n + 255) / 256), dim3(256), ue, bs, rss, n, beam_to_axis, n_axis, key);
        SLAM_TRY(sort_cells(ctx, ws + w.o_tmp, w.tmp_bytes, key, key2, rss, val2, n, w.bits));
        SLAM_LAUNCH(ctx, "cell_bounds", k_cell_bounds, dim3((n + 255) / 256), dim3(256), key2, n, (uint32_t)cells, begin, end);
    }
    SLAM_LAUNCH(ctx, "pivot_kahan", k_pivot_kahan, dim3((cells + 3) / 4), dim3(256), val2, begin, end, cells, sum, cnt, mean);
    return SLAM_OK;
}

int mean_only(slam_ctx *ctx, const unsigned long long *sum, const uint32_t *cnt, int cells, double *matrix)
{
    SLAM_LAUNCH(ctx, "mean", k_mean_only, dim3((cells + 255) / 256), dim3(256), sum, cnt, cells, matrix);
    return SLAM_OK;
}

int blur_only(slam_ctx *ctx, const double *in, int h, int w, const double *kern, int ks, double *out)
{
    const size_t cells = (size_t)h * w;
    SLAM_LAUNCH(ctx, "blur", k_blur_only, dim3((uint32_t)((cells + 255) / 256)), dim3(256), in, h, w, kern, ks, out);
    return SLAM_OK;
}

int render_only(slam_ctx *ctx, const double *mat, int h, int w, const double *aod, const double *aoa, int use_log,
                double vmin, double vmax, const uint8_t *lut, uint8_t *cells, uint8_t *tile, int W, int H,
                int16_t *colmap, int16_t *rowmap, uint32_t *stats)
{
    SLAM_LAUNCH(ctx, "render_cells", k_render_only, dim3(1), dim3(256), mat, h, w, aod, aoa, use_log, vmin, vmax,
                lut, cells, colmap, rowmap, W, H, stats);
    if (tile)      // the cells of one matrix: the stride between frames plays no part
        SLAM_LAUNCH(ctx, "raster", k_raster, raster_grid(W, H, 1), dim3(256), cells, colmap, rowmap, W, H, stats, tile, 0u);
    return SLAM_OK;
}
