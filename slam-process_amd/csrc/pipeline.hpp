// Internal stage launchers shared by the C-ABI entry points (host side).
#pragma once
#include "common.hpp"

// ---- record framing ----------------------------------------------------------------------------
constexpr int FRAME_RUN = 48;                    // data bytes owned by one lane
constexpr int FRAME_TILE = 64 * FRAME_RUN;       // 3072 data bytes per wave tile
constexpr int FRAME_TILE_DW = FRAME_TILE / 4 + 4;  // tile + 16 halo bytes, in dwords
constexpr int FRAME_SUMM_WORDS = 11;             // one packed word per entry offset 0..10
#ifndef SLAM_FRAME_TPW
#define SLAM_FRAME_TPW 4
#endif
constexpr int FRAME_TPW = SLAM_FRAME_TPW;        // tiles each wave walks per launch (2 / 3 / 4 / 8: 7.86 / 7.85 / 7.64 / 7.74 ms per 800 frames)
// single-pass framing: a wave's tile is one look-behind run + 63 runs of its own
constexpr int SLOT_TILE = 63 * FRAME_RUN;        // 3024 data bytes whose records one wave emits
constexpr int SLOT_ROWS = SLAM_TILE_SLOTS;       // row slots per such tile: ceil(3024 / 11) = 275, rounded up to 4
static_assert(SLOT_TILE == SLAM_TILE_BYTES && SLOT_ROWS % 4 == 0 && SLOT_ROWS * 11 >= SLOT_TILE + 10, "slot geometry");

struct frame_desc {             // prepared per frame: where its first token / byte is and how long it is
    uint64_t off;               // from frame_src::base
    uint32_t nb;                // data bytes
    uint32_t n_src;             // source bytes from off
};

enum { SRC_TEXT3 = 0, SRC_RAW = 1, SRC_DEC = 2 };

struct frame_src {
    const uint8_t *base;        // device: text (SRC_TEXT3) or data bytes (SRC_RAW) of all frames
    const uint64_t *off;        // device [F]
    const uint32_t *len;        // device [F]
    const uint8_t *dec;         // device: bytes decoded by pass 1 (frame f at f*dec_stride), SRC_DEC only
    uint64_t dec_stride;
    const frame_desc *fdesc;    // device [F], written by the prepare kernel
};

struct frame_ws {               // device scratch of the framing stage
    uint32_t *summ;             // [F * nt_max * 11]
    uint8_t *tile_entry;        // [F * nt_max]
    uint32_t *tile_off;         // [F * nt_max] records before each tile (two-pass frames); single-pass frames: the
    uint32_t *tile_off_d;       // [F * nt_max] discards before each tile ...  record / discard count of each workgroup
    uint8_t *dec;               // [F * dec_stride] decoded data bytes of text frames
    uint64_t dec_stride;
    static uint64_t dec_stride_for(uint32_t nt_max) { return (uint64_t)nt_max * FRAME_TILE + 256; }
    frame_desc *fdesc;          // [F]
    static void plan(ws_plan &p, uint32_t F, uint32_t nt_max, size_t off[6]);
    void bind(uint8_t *base, const size_t off[6], uint32_t nt_max);
};

// What the correction needs to know about neighbouring rows, produced per framing tile while the tile's records
// are still on chip (single-pass framing) or from the row arrays (k_tile_bits):
//   rmask   per 4 entries: group-start bits | baseline bits << 4 | is-a-row bits << 8 | FLAG == 1 bits << 12
//           (frame f at f*row_cap/4)
//   nbound / nbase  per tile: rows that start a group / are baselines (bit 31 of nbase: the tile's bits must
//           be recomputed from the row arrays because the row before its first row was not at hand)
//   pay     per tile up to BL_PAY_CAP baselines, any order: clk of the row before | bs << 32 | entry in tile << 40
constexpr uint32_t BL_PAY_CAP = SLOT_ROWS / 2 + 2;       // a baseline needs a FLAG 0 -> 1 step: at most every 2nd row
constexpr uint32_t TILE_NEEDS_BITS = 0x80000000u;
struct tile_bits {
    uint16_t *rmask = nullptr;
    uint32_t *nbound = nullptr, *nbase = nullptr;          // [F * stride]
    unsigned long long *pay = nullptr;                     // [F * stride * BL_PAY_CAP]
    uint32_t stride = 0;
};

// Decoded records, frame f at f*row_cap.  Two layouts:
//   dense   (slot_cnt == nullptr): records 0..VALID-1 in order;
//   slotted (slot_cnt != nullptr): the records that start in data bytes [t*SLOT_TILE, (t+1)*SLOT_TILE) occupy
//           slots [t*SLOT_ROWS, t*SLOT_ROWS + slot_cnt[f*slot_stride + t]); the rest of a tile's slots is unused.
struct rows_out {
    uint8_t *flag, *ue, *bs;
    uint32_t *rss, *clk;
    uint32_t row_cap;
    uint32_t *slot_cnt = nullptr;
    uint32_t slot_stride = 0;
    tile_bits bits;             // slotted layout only; bits.rmask == nullptr: not wanted
};

// Decode F frames: source -> records + stats words NBYTES / NOT_STRICT / VALID / DISCARDED.
// out.slot_cnt == nullptr: two passes, dense rows.  Otherwise one pass over the source into the slotted layout
// (frames whose look-behind run does not pin the scan position are redone by the two-pass kernels and
// described as full slots); stats also gets ROW_SLOTS and TWO_PASS.
struct fused_args;
int frame_decode(slam_ctx *ctx, int mode, const frame_src &src, uint32_t F, uint32_t max_src_len,
                 const frame_ws &ws, uint32_t nt_max, const rows_out &out, uint32_t *stats,
                 const fused_args *fused = nullptr, int force_two_pass = 0);
uint32_t frame_tiles_for(int mode, uint32_t max_src_len);
uint32_t slot_tiles_for(int mode, uint32_t max_src_len);
// After a dense (two-pass) frame_decode of the same source: data-byte indices of the discarded flag bytes, in scan
// order, frame f at f*pos_cap (stats DISCARDED of them).
int frame_rejects(slam_ctx *ctx, int mode, const frame_src &src, uint32_t F, const frame_ws &ws, uint32_t nt_max,
                  uint32_t *pos_out, uint32_t pos_cap);

// ---- correction ----------------------------------------------------------------------------------
constexpr int ROWS_PER_BLOCK = SLAM_KEPT_BLOCK;  // 256 threads x 4 rows
static_assert(ROWS_PER_BLOCK == 1024, "row kernels use 256 threads x 4 entries");

template <typename T8, typename T32>
struct rows_in_t {
    const T8 *flag, *ue, *bs;
    const T32 *rss, *clk;
    uint32_t row_cap;           // per-frame stride
    const uint32_t *slot_cnt = nullptr;   // slotted layout (see rows_out); nullptr = dense
    uint32_t slot_stride = 0;
    tile_bits bits;                       // slotted layout: per-tile bits (rmask == nullptr: use the per-block kernels)
    int bits_fresh = 0;                   // 1: the framing pass of this call filled `bits` (except flagged tiles / frames)
};

// Kept (corrected) rows.  slot_cnt == nullptr: dense, in order.  Otherwise the kept rows among the entries
// [T*ROWS_PER_BLOCK, (T+1)*ROWS_PER_BLOCK) of the input sit at the front of the same entries of the output,
// slot_cnt[f*slot_stride + T] of them.
template <typename T8, typename T32>
struct kept_out_t {
    T8 *ue, *bs;
    T32 *rss, *clk;
    uint32_t *slot_cnt = nullptr;
    uint32_t slot_stride = 0;
};

struct correct_ws {
    uint32_t *rt_nbound, *rt_nbase, *rt_nkept;   // [F * rt_max] per row-tile counts, then exclusive bases
    uint32_t *grp_row, *grp_bl, *grp_nd;         // [F * (row_cap + 1)] group tables
    int32_t *bl_clk, *bl_bs;                     // [F * bl_cap] baselines in row order
    uint32_t *bl_gid;                            // [F * bl_cap]
    uint32_t *srt_rb;                            // [F * bl_cap] per group: distinct clk_b mod cycle, ascending
    uint4 *srt_pay;                              // same order: (baseline index in group, clk_b div cycle, bs_b, 0)
    int32_t *corr;                               // [F * row_cap] corrected BS per row or "none"
    uint16_t *rmask;                             // [F * row_cap / 4] per 4 entries: group-start bits | baseline bits << 4 | row bits << 8 | FLAG == 1 bits << 12
    uint32_t *tmp_a, *tmp_b;                     // [F * bl_cap] scratch of the large-group sort
    uint32_t *defer;                             // [F * rt_max] workgroups of k_rows_finish left to its full build
    const uint32_t *gate = nullptr;              // [F] or null: serve only the frames whose word is nonzero (fused_ws::redo)
    uint32_t bl_cap;                             // baselines per frame the arrays can hold
    static uint32_t default_bl_cap(uint32_t row_cap) { return row_cap / 2 + 16; }
    static void plan(ws_plan &p, uint32_t F, uint32_t row_cap, uint32_t bl_cap, size_t off[16]);
    void bind(uint8_t *base, const size_t off[16], uint32_t bl_cap);
};

struct correct_params {
    int32_t cycle = 0, tol = 0, mod_base = 0;
    uint32_t div_m = 0, div_s1 = 0, div_s2 = 0;   // division by cycle as a multiply-high (set_cycle)
    uint32_t bias = 0;                            // multiple of cycle >= 2^30 added to clocks before dividing
    void set_cycle(int32_t c);
    int small_tol = 0;          // set by correct_rows: 0 <= 2*tol < cycle and mod_base a power of two
    int one_group = 0;          // 1: the whole table is one group (UE wraps ignored)
    int explicit_baselines = 0; // 1: baselines preloaded into bl_clk/bl_bs, count in stats BASELINES (implies one_group)
};

template <typename T8, typename T32>
int correct_rows(slam_ctx *ctx, const rows_in_t<T8, T32> &in, uint32_t F, const correct_ws &ws,
                 const correct_params &prm, const kept_out_t<T8, T32> &out, int32_t *gid_out,
                 uint32_t *stats);

// ---- fused decode + correction -------------------------------------------------------------------
// One kernel takes text to records AND corrected rows: a workgroup decodes FUSED_TILES consecutive framing tiles,
// keeps their rows in LDS, exchanges the baselines of the sweeps that cross its borders with the neighbouring
// workgroups (published through `pub`, signalled through `flag`) and corrects its rows there.  Kept rows of such a
// frame are slotted per tile like the decoded rows: the rows kept among tile t's entries sit at the front of the
// same entries of the kept arrays, slot_cnt[f*slot_stride + t] of them.  A frame the kernel cannot finish (see
// k_fused.hip) is flagged in `fail` and corrected by the separate kernels, which `redo` gates.
#ifndef SLAM_FUSED_TILES
#define SLAM_FUSED_TILES 16
#endif
constexpr int FUSED_TILES = SLAM_FUSED_TILES;    // framing tiles a workgroup decodes and corrects (tuning builds: 8 with four waves)
constexpr int FUSED_ENTRIES = FUSED_TILES * SLOT_ROWS;
constexpr int FUSED_PUB_WORDS = 256;             // 8-byte words a workgroup publishes: the baselines of its two border sweeps
constexpr int FUSED_CNT_WORDS = 4;               // per workgroup: valid, discarded, groups, baselines
struct fused_ws {
    uint32_t *flag = nullptr;                    // [F * wg_stride] the header word a workgroup publishes, 0 = not yet (zeroed before every launch)
    unsigned long long *pub = nullptr;           // [F * wg_stride * FUSED_PUB_WORDS]
    uint32_t *wg_cnt = nullptr;                  // [F * wg_stride * FUSED_CNT_WORDS]
    uint32_t *fail = nullptr;                    // [F] nonzero (SLAM_FUSED_GAVE_UP_* bits): the fused kernel gave the frame up (zeroed before every launch)
    uint32_t *redo = nullptr;                    // [F] nonzero: the separate correction kernels serve the frame
    uint32_t wg_stride = 0;
    static uint32_t wgs_for(uint32_t st_max) { return (st_max + FUSED_TILES - 1) / FUSED_TILES; }
    static void plan(ws_plan &p, uint32_t F, uint32_t st_max, size_t off[5]);
    void bind(uint8_t *base, const size_t off[5], uint32_t st_max);
};
bool fused_supported(const correct_params &prm);
struct fused_args {                              // what frame_decode needs to run the fused kernel instead of k_frame_onepass
    fused_ws fw;
    correct_params prm;
    kept_out_t<uint8_t, uint32_t> kout;
};
// the fused kernel over F frames (grid: workgroups of FUSED_TILES tiles x frames), and the per-frame wrap-up after the
// two-pass kernels have redone what it flagged: counters, layout bits, fw.redo
int fused_launch(slam_ctx *ctx, int mode, const frame_src &src, uint32_t F, uint32_t st_max, const rows_out &out,
                 const fused_args &fz, uint32_t *stats);
int fused_finish_launch(slam_ctx *ctx, int mode, uint32_t F, const fused_args &fz, uint32_t *stats);

// ---- pivot / blur / raster ---------------------------------------------------------------------
constexpr int SMALL_AXIS = 64;          // axis tables up to here: one workgroup per frame, grids in LDS (k_frame_matrix)
constexpr int WIDE_AXIS = SLAM_MAX_AXIS; // larger tables (up to 256 entries): grids in global memory, several kernels

struct render_params {
    const int16_t *beam_to_axis;    // device [64]
    const double *axis_angle;       // device [n_axis]
    int32_t n_axis;
    double sigma;
    int32_t use_log;
    double vmin, vmax;
    const uint8_t *lut;             // device [768]
    int32_t W, H;
    int32_t full_grid;              // 1: every entry of the axis table is a row and a column, occupied or not
};

struct render_ws {
    unsigned long long *sum;        // [F * cells]   cells = slam_cells_per_frame(n_axis) (wide tables: [F * 4096] beam pairs)
    uint32_t *cnt;                  // [F * cells]
    uint8_t *cells;                 // [F * cells * 4] RGBA per compact cell
    int16_t *colmap, *rowmap;       // [F * W], [F * H] pixel -> compact cell index
    // wide axis tables only
    double *grid_a, *grid_b;        // [F * cells] full-grid mean / compact mean / blurred, in turn
    uint32_t *present;              // [F * 2 * WIDE_AXIS] rows, then columns, that hold data
    int16_t *axis_idx;              // [F * 2 * WIDE_AXIS] compact row -> axis entry, compact column -> axis entry
    unsigned long long *minmax;     // [F * 256 * 2] per 16 x 16 piece: order-preserving keys of its finite minimum / maximum
    unsigned long long *sum_w;      // [F * cells] sums / counts on the axis grid (sum / cnt then hold the 64 x 64 beam pairs)
    uint32_t *cnt_w;
    uint32_t cell_stride;           // cells per frame
    static void plan(ws_plan &p, uint32_t F, int W, int H, int n_axis, size_t off[12]);
    void bind(uint8_t *base, const size_t off[12], int n_axis);
};

// slot_cnt != nullptr: the rows are slotted per ROWS_PER_BLOCK entries (kept_out_t) and stat_word counts entries
template <typename T8, typename T32>
int pivot_accumulate(slam_ctx *ctx, const T8 *ue, const T8 *bs, const T32 *rss, uint32_t row_cap,
                     uint32_t F, const uint32_t *stats, int stat_word, const render_params &rp,
                     const render_ws &ws, const uint32_t *slot_cnt = nullptr, uint32_t slot_stride = 0);
// sum/cnt -> mean matrix (full axis grid), compact + blur, scale, colours, index maps, tile.
// kern is the ks x ks Gaussian kernel on the device (ks == 1: no blur).
int matrix_blur_render(slam_ctx *ctx, uint32_t F, const render_params &rp, const render_ws &ws, const double *kern,
                       int ks, double *matrix, double *blurred, uint8_t *tile, uint32_t *stats);

// standalone small-matrix stages used by slam_pivot_mean / slam_blur_nan / slam_render (device pointers)
int mean_only(slam_ctx *ctx, const unsigned long long *sum, const uint32_t *cnt, int cells, double *matrix);
// real-valued RSS: per cell the Kahan sum in row order, the count and the mean.  The rows are brought together cell by
// cell with a stable sort (k_sort.hip); `ws` holds the scratch pivot_kahan_ws::plan laid out.
struct pivot_kahan_ws {
    size_t o_key = 0, o_key2 = 0, o_val2 = 0, o_bounds = 0, o_tmp = 0, tmp_bytes = 0;
    unsigned bits = 0;
    size_t plan(ws_plan &p, uint32_t n, int n_axis);
};
int pivot_kahan(slam_ctx *ctx, const int32_t *ue, const int32_t *bs, const double *rss, uint32_t n,
                const int16_t *beam_to_axis, int n_axis, uint8_t *ws, const pivot_kahan_ws &w, double *sum, uint32_t *cnt,
                double *mean);
size_t sort_cells_temp_bytes(uint32_t n, unsigned bits);
int sort_cells(slam_ctx *ctx, void *tmp, size_t tmp_bytes, const uint32_t *key_in, uint32_t *key_out, const double *val_in,
               double *val_out, uint32_t n, unsigned bits);
int blur_only(slam_ctx *ctx, const double *in, int h, int w, const double *kern, int ks, double *out);
int render_only(slam_ctx *ctx, const double *mat, int h, int w, const double *aod, const double *aoa, int use_log,
                double vmin, double vmax, const uint8_t *lut, uint8_t *cells, uint8_t *tile, int W, int H,
                int16_t *colmap, int16_t *rowmap, uint32_t *stats);
int raster_only(slam_ctx *ctx, const uint8_t *cells, const int16_t *colmap, const int16_t *rowmap, const uint32_t *stats,
                uint32_t F, int W, int H, uint8_t *tile, uint32_t cell_stride);
int slam_gaussian_kernel_host(double sigma, std::vector<double> &k);

// ---- dictionary correlation (k_dict.hip) -----------------------------------------------------------
// out[f] = phi_rx^T . m[f] . phi_tx, f64 row-major device arrays: phi_rx [n_ue x ga], phi_tx [n_bs x gd],
// m [n_frames x n_ue x n_bs], out [n_frames x ga x gd]; n_ue, n_bs <= 64.
int dict_correlate(slam_ctx *ctx, const double *phi_rx, int n_ue, int ga, const double *phi_tx, int n_bs, int gd,
                   const double *m, uint32_t n_frames, double *out);

// ---- general tokenizer ---------------------------------------------------------------------------
// text -> bytes with the reference's exact token rules; n_out_dev receives the count; ws holds
// tokenize_ws_bytes(n_text) bytes of scratch.
int tokenize_general(slam_ctx *ctx, const uint8_t *text_dev, size_t n_text, uint8_t *bytes_dev, uint32_t *n_out_dev,
                     uint8_t *ws);
size_t tokenize_ws_bytes(size_t n_text);
// The tokens that tokenizer skips, for the reference's per-token log lines: (index among all tokens, byte
// begin, byte end) per skipped token, in text order; n_dev[0] = how many there are.  `ws` must hold
// token_rejects_ws_bytes(n_text) bytes of scratch.
int token_rejects(slam_ctx *ctx, const uint8_t *text_dev, size_t n_text, uint32_t *idx_dev, uint32_t *begin_dev,
                  uint32_t *end_dev, uint32_t cap, uint32_t *n_dev, uint8_t *ws);
size_t token_rejects_ws_bytes(size_t n_text);
