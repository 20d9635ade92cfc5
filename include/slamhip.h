/*
 * slamhip.h -- C ABI of libslamhip.so, the MI355X (gfx950) implementation of the
 * slam-process decode -> correct -> raster hot path.
 *
 * The reference (lull21/slam-process) has no native interface: its hot path is
 * the module-level Python functions of serial_hex_to_excel_v3.py,
 * bs_beam_correction.py and excel_heatmap_v3.py.  Each entry point below names
 * the reference function(s) whose data path it replaces; the Python modules of
 * the same names in slam-process_amd/ bind them through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SLAM_E_* code;
 *     slam_last_error(ctx) gives the message of the last failure on that ctx;
 *   - all buffers are caller-owned, sizes are explicit, nothing is retained
 *     after return except the ctx's own device workspace;
 *   - plain entry points take HOST pointers and are synchronous;
 *     *_dev entry points take DEVICE pointers, enqueue on the ctx's stream and
 *     return without synchronising (call slam_sync);
 *   - a ctx is bound to one device and one stream and is not thread-safe.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_OK              0
#define SLAM_E_INVALID      -1   /* bad argument (NULL pointer, size out of range) */
#define SLAM_E_HIP          -2   /* a HIP runtime call failed */
#define SLAM_E_CAPACITY     -3   /* an output buffer is too small; required size is reported */
#define SLAM_E_EMPTY        -4   /* the stage produced no rows (the reference raises ValueError) */
#define SLAM_E_ALLNAN       -5   /* matrix has no finite cell (excel_heatmap_v3.py:405-406) */
#define SLAM_E_RANGE        -6   /* a value does not fit the device representation */
#define SLAM_E_LIMITS       -7   /* log scale with a limit at or below the data minimum: the reference's LogNorm
                                    (excel_heatmap_v3.py:413-418) fails with "Invalid vmin or vmax" */

typedef struct slam_ctx slam_ctx;

/* ---- context ------------------------------------------------------------------------- */
int slam_ctx_create(int device, slam_ctx **out);
int slam_ctx_destroy(slam_ctx *ctx);
/* Enqueue on an existing hipStream_t from now on (e.g. torch's current stream).  The handle is taken as is:
 * NULL is HIP's null stream, which is what torch's default stream is.  A new ctx uses a non-blocking stream
 * of its own; slam_use_own_stream goes back to it. */
int slam_set_stream(slam_ctx *ctx, void *hip_stream);
int slam_use_own_stream(slam_ctx *ctx);
int slam_sync(slam_ctx *ctx);
const char *slam_last_error(slam_ctx *ctx);
const char *slam_version(void);
/* Revision of this binary interface (struct layouts, argument lists, buffer layouts): a binding compares it with
 * the SLAM_ABI_REVISION it was written against before the first call. */
#define SLAM_ABI_REVISION 4
int slam_abi_revision(void);

/* ---- a1: hex text -> bytes ---------------------------------------------------------------
 * Replaces _read_bytes (serial_hex_to_excel_v3.py:18-36, regex :16): UTF-8 decode with
 * errors="ignore", split on Unicode whitespace, keep tokens of exactly two hex digits with an
 * optional lowercase "0x".  bytes_out needs room for n_text/2 + 1 bytes at most. */
int slam_hex_unpack(slam_ctx *ctx, const uint8_t *text, size_t n_text,
                    uint8_t *bytes_out, size_t cap, size_t *n_bytes);

/* Diagnostics of a1: the tokens _read_bytes skips, in text order — one per "跳过非16进制单字节token[idx]" log line
 * (serial_hex_to_excel_v3.py:22, :26-28).  index_out = the token's index among all tokens of the text (the
 * reference's enumerate index), [begin_out, end_out) = the bytes of the text from the token's first character
 * to the whitespace that ends it; decoding that span as UTF-8 with errors ignored gives the token.
 * n_rejects is always set to their number; SLAM_E_CAPACITY when it exceeds cap (nothing is copied then). */
int slam_token_rejects(slam_ctx *ctx, const uint8_t *text, size_t n_text,
                       uint32_t *index_out, uint32_t *begin_out, uint32_t *end_out, size_t cap, size_t *n_rejects);

/* ---- a2: bytes -> records ------------------------------------------------------------------
 * Replaces _parse_groups (serial_hex_to_excel_v3.py:41-103): the resynchronising scan for
 * 11-byte records FLAG UE BS CLK0..4 RSS0..2.  Outputs are struct-of-arrays in record order;
 * cap >= n/11 + 1 always suffices.  n_valid / n_discarded are the reference's two counters. */
int slam_parse_records(slam_ctx *ctx, const uint8_t *bytes, size_t n,
                       uint8_t *flag, uint8_t *ue, uint8_t *bs, uint32_t *rss, uint32_t *clk,
                       size_t cap, size_t *n_valid, size_t *n_discarded);

/* Diagnostics of a2: the byte indices at which _parse_groups discards a group, in scan order — one per "丢弃组"
 * log line (serial_hex_to_excel_v3.py:53-58 truncated tail, :62-87 failed tag checks).  The reason and the hex dump
 * of each line follow from the bytes at that index.  n_rejects equals the `discarded` counter; cap >= it. */
int slam_parse_rejects(slam_ctx *ctx, const uint8_t *bytes, size_t n,
                       uint32_t *pos_out, size_t cap, size_t *n_rejects);

/* a1+a2 fused: text straight to records (the text is never materialised as bytes when it has
 * the capture format "XX XX XX ..." with an optional leading "« "). */
int slam_parse_text(slam_ctx *ctx, const uint8_t *text, size_t n_text,
                    uint8_t *flag, uint8_t *ue, uint8_t *bs, uint32_t *rss, uint32_t *clk,
                    size_t cap, size_t *n_valid, size_t *n_discarded);

/* ---- a4-a7: BS beam correction -----------------------------------------------------------------
 * Replaces _detect_groups (:43-47), _identify_baselines (:50-61), _filter_group_corrected
 * (:140-178) and _correct_group (:79-116) of bs_beam_correction.py, over a whole table as
 * process_excel_filtered (:186-198) / process_excel (:125-134) drive them.
 *
 * Columns are int32 (|clk| < 2^30).  Any output pointer may be NULL.
 *   out_*        kept rows in order (UE_Beam, corrected BS_Beam, RSS, CLK), capacity n
 *   gid_out      group id per input row                                      [n]
 *   corrected    per input row: the corrected BS where a candidate won, else -1   [n]
 *   bl_clk/bl_bs/bl_gid  the baselines found, in order, capacity n/2 + 1
 * flags & SLAM_CORRECT_ONE_GROUP: the table is ONE group whatever its UE column holds (what the
 * reference's per-group functions see: their shift(1) never looks at UE).
 * If bl_in_n >= 0 the table is one group whose baselines are the given bl_in_clk/bl_in_bs (the
 * explicit `baselines` argument of _correct_group / _filter_group_corrected); none are detected. */
#define SLAM_CORRECT_ONE_GROUP 1u
int slam_correct(slam_ctx *ctx,
                 const int32_t *flag, const int32_t *ue, const int32_t *bs,
                 const int32_t *rss, const int32_t *clk, size_t n,
                 int32_t cycle, int32_t tol, int32_t mod_base, uint32_t flags,
                 const int32_t *bl_in_clk, const int32_t *bl_in_bs, int64_t bl_in_n,
                 int32_t *out_ue, int32_t *out_bs, int32_t *out_rss, int32_t *out_clk, size_t *n_kept,
                 int32_t *gid_out, int32_t *corrected,
                 int32_t *bl_clk, int32_t *bl_bs, int32_t *bl_gid, size_t *n_baselines,
                 size_t *n_groups);

/* ---- a9-a10: angle mapping + mean pivot ---------------------------------------------------------
 * Replaces map_angles (excel_heatmap_v3.py:264-280) + compute_rssi_matrix (:286-314).
 * beam_to_axis[64]: index of each beam's angle among the sorted distinct mapped angles, -1 if
 * unmapped (rows using it are dropped); n_axis <= SLAM_MAX_AXIS.  The reference's table has 64 entries, so
 * n_axis <= 64 there; a larger axis is the up-scaled table of slam_batch.full_grid, where beam b sits at entry
 * 4 * b of a 256-entry table.  Rows with a beam outside 0..63 are dropped.
 * sum/cnt: exact integer accumulators [n_axis*n_axis] (row = AoD axis of BS, col = AoA axis of UE).
 * matrix: mean as f64, NaN where cnt == 0, same layout.  rss values must be in [0, 2^31). */
#define SLAM_MAX_AXIS 256
int slam_pivot_mean(slam_ctx *ctx, const int32_t *ue, const int32_t *bs, const int32_t *rss, size_t n,
                    const int16_t *beam_to_axis, int n_axis,
                    int64_t *sum, int64_t *cnt, double *matrix);

/* The same pivot for RSS values that are not integers (the reference reads RSS with float(), excel_heatmap_v3.py:151-157,
 * and averages whatever it finds): pandas' pivot_table(aggfunc=mean) (:297-299) adds a cell's values in row order with
 * Kahan compensation and divides by their number, and so does this, bit for bit.  NaN values are not counted.
 * sum: the compensated sums; cnt, matrix as in slam_pivot_mean. */
int slam_pivot_mean_f64(slam_ctx *ctx, const int32_t *ue, const int32_t *bs, const double *rss, size_t n,
                        const int16_t *beam_to_axis, int n_axis,
                        double *sum, int64_t *cnt, double *matrix);

/* ---- a11-a12: NaN-aware Gaussian blur --------------------------------------------------------------
 * Replaces gaussian_kernel (:320-333) + gaussian_blur_nan_aware (:336-368): normalised
 * convolution, edge padding, NaN where the weight sum <= 1e-12.  h, w <= 4096; kernels of up to 127 taps
 * (sigma <= 21), SLAM_E_RANGE beyond.  sigma <= 0 copies. */
int slam_blur_nan(slam_ctx *ctx, const double *in, int h, int w, double sigma, double *out);

/* ---- a13: colour scale + mesh raster --------------------------------------------------------------
 * Replaces the data path of generate_heatmap (:396-456) after the blur: finite min/max,
 * Normalize / shifted LogNorm, 256-entry colormap lookup, and the pcolormesh of flat cells with
 * edges at midpoints (_edges :430-440), sampled into a data-only W x H RGBA tile (row 0 = largest
 * AoD); h, w <= SLAM_MAX_AXIS.  vmin/vmax (NaN = none given) count on the log scale only: the reference hands the linear mesh neither a
 * norm nor limits (:420-423, :449-456), so its linear scale always spans the finite data.  Empty and reversed
 * intervals come out as matplotlib's colour bar leaves them (expanded by a tenth).  Outputs (any may be NULL):
 * cells_rgba [h*w*4], tile_rgba [H*W*4], vmin_vmax_out[2] the limits used (after the log shift when use_log). */
int slam_render(slam_ctx *ctx, const double *mat, int h, int w,
                const double *aod, const double *aoa, int use_log, double vmin, double vmax,
                const uint8_t *lut256x3, uint8_t *cells_rgba, uint8_t *tile_rgba, int W, int H,
                double *vmin_vmax_out);

/* ---- next to the path: the dictionary correlation of the reference's path estimators -------------------
 * out[f] = Phi_RX^T . M[f] . Phi_TX in float64, for n_frames matrices at once: the line
 * `correlation = self.Phi_RX.T @ res_mat @ self.Phi_TX` that opens every round of LoSEstimator.estimate_los_path
 * (heatmap_gemini_v4.py:207-209) and NLoSEstimator.estimate_nlos_paths (:304; the same line in heatmap_gemini_v1 .. v3).
 * Row-major: phi_rx [n_ue x ga] and phi_tx [n_bs x gd] as construct_dictionary builds them (:182-198, :283-296; one row
 * per beam, one column per point of the search grid), m [n_frames x n_ue x n_bs] with UE rows and BS columns as
 * BeamDataProcessor.pivot_data leaves the matrix (:117-150), out [n_frames x ga x gd].  n_ue, n_bs <= 64 (beam ids are
 * 6 bit), any ga, gd.  Sums run on the f64 matrix cores in another order than NumPy's BLAS: results agree to about
 * 1e-13 of the largest element, not bit for bit.  The estimators' other steps (arg-max, NNLS, masks, clustering, plots)
 * are not part of this library. */
int slam_dict_correlate(slam_ctx *ctx, const double *phi_rx, int n_ue, int ga, const double *phi_tx, int n_bs, int gd,
                        const double *m, uint32_t n_frames, double *out);
/* the same on device pointers, asynchronous on the ctx's stream */
int slam_dict_correlate_dev(slam_ctx *ctx, const double *phi_rx_dev, int n_ue, int ga, const double *phi_tx_dev, int n_bs,
                            int gd, const double *m_dev, uint32_t n_frames, double *out_dev);

/* ---- batched, device-resident pipeline (the benchmark path) ----------------------------------------
 * F frames of hex text resident in device memory -> per frame: decoded records, corrected rows,
 * mean matrix, blurred matrix, RGBA tile.  Everything stays on the device; nothing synchronises.
 * All *_dev pointers are device pointers; optional outputs may be NULL. */
typedef struct slam_batch {
    /* inputs */
    const uint8_t *text_dev;      /* all frames' text; readable up to 32 bytes past the last frame */
    const uint64_t *text_off_dev; /* [F] byte offset of each frame's text (any alignment)        */
    const uint32_t *text_len_dev; /* [F] text bytes of each frame                                */
    uint32_t n_frames;            /* F, at most 65535 per call                                    */
    uint32_t max_text_len;        /* max over frames of text_len (host-known upper bound; a frame */
                                  /* whose length word is larger is cut at this many bytes)       */
    uint32_t src_is_bytes;        /* 0: text_dev holds hex text in the capture format;            */
                                  /* 1: it holds already-unpacked data bytes (slam_hex_unpack)    */
    int32_t cycle, tol, mod_base; /* CYCLE, TOL, MOD_BASE (bs_beam_correction.py:23-25)          */
    const int16_t *beam_to_axis_dev;  /* [64] axis entry of each beam, -1 = unmapped             */
    const double *axis_angle_dev;     /* [n_axis] sorted distinct angles                         */
    int32_t n_axis;                   /* 1 .. SLAM_MAX_AXIS                                      */
    int32_t full_grid;                /* 0: rows / columns of the matrix are the angles that occur, as     */
                                      /* compute_rssi_matrix (excel_heatmap_v3.py:301-307) builds them;    */
                                      /* 1: every entry of the axis table is a row and a column (the       */
                                      /* up-scaled table mode: no counterpart in the reference, whose      */
                                      /* validate_beam_id :68-72 admits 64 beams)                          */
    double blur_sigma;            /* --blur-sigma, default 1.0 (excel_heatmap_v3.py:601-603)      */
    int32_t use_log;              /* --logscale                                                   */
    double vmin, vmax;            /* NaN = none given; used on the log scale only (see slam_render) */
    const uint8_t *lut_dev;       /* [256*3]                                                      */
    int32_t tile_w, tile_h;
    /* outputs; row arrays have row_cap entries per frame, frame f at offset f*row_cap (16-byte aligned
     * bases).  Decoded records are written tile by tile: the records that start in data bytes
     * [t*SLAM_TILE_BYTES, (t+1)*SLAM_TILE_BYTES) of frame f, in order, occupy entries
     * [t*SLAM_TILE_SLOTS, t*SLAM_TILE_SLOTS + tile_cnt_dev[f*T + t]) with T = slam_row_tiles(max_text_len,
     * src_is_bytes); the other entries of a tile are unused.  Concatenating the tiles' used entries gives
     * the reference's record list.  Corrected rows (kept_*) use the same scheme over blocks of U entries:
     * the rows kept among entries [b*U, (b+1)*U) of the decoded arrays sit, in order, at the front of the same
     * entries of the kept arrays, kept_cnt_dev[f*KB + b] of them, KB = slam_kept_blocks(row_cap, max_text_len,
     * src_is_bytes).  U is a property of the frame: SLAM_KEPT_BLOCK, or SLAM_TILE_SLOTS (a block per tile, like
     * the decoded records) when bit SLAM_LAYOUT_KEPT_TILES of the frame's SLAM_STAT_LAYOUT word is set (frames
     * the fused decode + correct kernel served, see slam_set_fused). */
    uint32_t row_cap;             /* multiple of 16, >= T * SLAM_TILE_SLOTS                       */
    uint8_t *flag_dev, *ue_dev, *bs_dev;        /* decoded records                               */
    uint32_t *rss_dev, *clk_dev;
    uint32_t *tile_cnt_dev;       /* [F*T] records per tile (NULL: kept in the ctx workspace)     */
    uint8_t *kept_ue_dev, *kept_bs_dev;         /* corrected rows                                */
    uint32_t *kept_rss_dev, *kept_clk_dev;
    uint32_t *kept_cnt_dev;       /* [F*KB] kept rows per block (NULL: kept in the ctx workspace) */
    uint32_t *stats_dev;          /* [F*SLAM_STAT_WORDS], see SLAM_STAT_*                         */
    double *matrix_dev;           /* [F*n_axis*n_axis] mean, NaN = empty (full axis grid)         */
    double *blurred_dev;          /* [F*n_axis*n_axis] blurred compact matrix, row stride n_aoa   */
    uint8_t *tile_dev;            /* [F*tile_h*tile_w*4]                                          */
    /* optional: what the tile is painted from (NULL: kept in the ctx workspace).  tile[py][px] =
     * cells[rowmap[py] * N_AOA + colmap[px]]; with the frame's stats words these 20 KB reproduce the tile
     * bit for bit through slam_raster_dev, e.g. on the rank that collects results. */
    uint8_t *cells_dev;           /* [F*slam_cells_per_frame(n_axis)*4] RGBA per cell of the compact matrix */
    int16_t *colmap_dev;          /* [F*tile_w] pixel column -> cell column                       */
    int16_t *rowmap_dev;          /* [F*tile_h] pixel row -> cell row                             */
} slam_batch;

/* Cells reserved per frame in cells_dev and in slam_raster_dev: 4096 for axis tables of up to 64 entries,
 * n_axis * n_axis beyond. */
uint32_t slam_cells_per_frame(int32_t n_axis);

/* sizeof(slam_batch) as the library was built with: a binding compares it with its own mirror of the struct
 * before the first call (no counterpart in the reference, which has no binary interface). */
size_t slam_batch_sizeof(void);

#define SLAM_TILE_BYTES    3024   /* data bytes per decoded-record tile                           */
#define SLAM_TILE_SLOTS     276   /* entries reserved per tile in the decoded-record arrays        */
#define SLAM_KEPT_BLOCK    1024   /* entries per block of the corrected-row arrays (or SLAM_TILE_SLOTS: SLAM_LAYOUT_KEPT_TILES) */
/* Tiles per frame for frames of up to max_text_len text bytes (or data bytes when src_is_bytes). */
uint32_t slam_row_tiles(uint32_t max_text_len, uint32_t src_is_bytes);
/* Counts per frame in kept_cnt_dev: enough for either block size, max(ceil(row_cap / SLAM_KEPT_BLOCK), slam_row_tiles(...)). */
uint32_t slam_kept_blocks(uint32_t row_cap, uint32_t max_text_len, uint32_t src_is_bytes);

#define SLAM_STAT_WORDS      16
#define SLAM_STAT_NBYTES      0   /* data bytes decoded from the text                            */
#define SLAM_STAT_NOT_STRICT  1   /* 1 = text is not in the strict capture format: the frame comes back with no rows   */
                                  /* (VALID, DISCARDED, ROW_SLOTS, GROUPS, BASELINES, KEPT all 0, ALLNAN 1, tile blank) */
#define SLAM_STAT_VALID       2   /* reference's `valid`                                          */
#define SLAM_STAT_DISCARDED   3   /* reference's `discarded`                                      */
#define SLAM_STAT_GROUPS      4
#define SLAM_STAT_BASELINES   5
#define SLAM_STAT_KEPT        6
#define SLAM_STAT_N_AOD       7   /* rows of the compact matrix                                   */
#define SLAM_STAT_N_AOA       8   /* cols of the compact matrix                                   */
#define SLAM_STAT_ALLNAN      9   /* 1 = no finite cell, 2 = invalid log-scale limits; tile left transparent */
#define SLAM_STAT_VMIN_LO    10   /* vmin/vmax used, as two u32 halves of an f64 each             */
#define SLAM_STAT_VMIN_HI    11
#define SLAM_STAT_VMAX_LO    12
#define SLAM_STAT_VMAX_HI    13
#define SLAM_STAT_ROW_SLOTS  14   /* entries of the decoded-record arrays this frame spans         */
#define SLAM_STAT_LAYOUT     15   /* bits: how the frame's row arrays are laid out                  */
#define SLAM_LAYOUT_TWO_PASS     1u  /* decoded by the two-pass kernels: entries are dense (all tiles full but the last) */
#define SLAM_LAYOUT_KEPT_TILES   2u  /* corrected rows are slotted per tile (SLAM_TILE_SLOTS entries), not per SLAM_KEPT_BLOCK */
/* Diagnostic, bits 8 and up: why the fused decode + correct kernel handed the frame to the separate kernels (0: it
 * did not).  SLAM_FUSED_GAVE_UP_* << SLAM_LAYOUT_FUSED_WHY_SHIFT, several may be set. */
#define SLAM_LAYOUT_FUSED_WHY_SHIFT 8
#define SLAM_FUSED_GAVE_UP_TILE_LISTS   1u  /* more than 16 baselines or 8 sweep starts within one tile               */
#define SLAM_FUSED_GAVE_UP_UNSYNC       2u  /* a look-behind run does not pin the scan position (two-pass frame)       */
#define SLAM_FUSED_GAVE_UP_FIRST_ROW    4u  /* the record before a workgroup's first row was not at hand               */
#define SLAM_FUSED_GAVE_UP_WG_LISTS     8u  /* more than 7 sweep starts in 16 tiles, or a border sweep's list too long */
#define SLAM_FUSED_GAVE_UP_TIMEOUT     16u  /* a neighbouring workgroup's baselines did not arrive in time             */
#define SLAM_FUSED_GAVE_UP_NEIGHBOUR   32u  /* a neighbour gave up, or a sweep spans more workgroups than are walked   */
#define SLAM_FUSED_GAVE_UP_GROUP_SIZE  64u  /* a sweep of more than 256 baselines                                      */

/* Replaces, per frame, the chain main() -> _read_bytes -> _parse_groups
 * (serial_hex_to_excel_v3.py:153-154), process_excel_filtered's data path
 * (bs_beam_correction.py:186-198) and process_excel_to_heatmap's data path
 * (excel_heatmap_v3.py:519-545) without the xlsx hops between them.
 * stages: bit 0 decode, bit 1 correct, bit 2 pivot+blur+raster (later stages need earlier ones; a stage may be run
 * again on the same buffers, its counters are set, not added to; the decode stage starts a frame's stats afresh). */
int slam_process_frames_dev(slam_ctx *ctx, const slam_batch *batch, uint32_t stages);

/* How slam_process_frames_dev / _noalloc_dev run a call that asks for decode and correction together: 0 = two
 * kernels (records to memory, then read again by the correction), 1 = one fused kernel that corrects a workgroup's
 * rows while they are still on chip (frames it cannot finish -- a sweep of more than 256 baselines, more than 7
 * sweep starts within 16 tiles, a tolerance of half a cycle or more, a cycle of 2^24 or more ... -- are handed to
 * the separate kernels inside the same call).  Same records, same corrected rows, same counters either way; only
 * the block size of the corrected-row arrays differs per frame (SLAM_STAT_LAYOUT).  A new ctx takes its setting from
 * the environment variable SLAMHIP_FUSED (0 / 1) if set, else the library's default.  No counterpart in the
 * reference, which corrects a table with Python loops (bs_beam_correction.py:140-178). */
int slam_set_fused(slam_ctx *ctx, int mode);
int slam_get_fused(slam_ctx *ctx);

/* The same call for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference, which has no device memory to manage.  Where
 * slam_process_frames_dev takes its scratch (per-tile bit masks and counts, baseline lists, group and phase tables,
 * pivot accumulators, the blur weights) from the ctx workspace and grows that on demand, this entry point takes it from
 * a fixed array of SLAM_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds from the first slam_ctx_create
 * on it until the process ends.  It calls no hipMalloc / hipFree and copies nothing from the host (the blur weights
 * travel as a kernel argument); kernel timing is off for the call.  Same batch, same stages, same results.
 * Limits: SLAM_E_CAPACITY when the batch shape needs more scratch than the array holds (one 256 x 4096-record frame
 * needs about 54 MB; several smaller frames fit together), SLAM_E_RANGE for a blur of more than 21 taps
 * (sigma > 3.5).  All calls on one GPU share that GPU's array, whatever their ctx, stream or thread: they are ordered.
 * Each call makes its stream wait for the call issued before it on that GPU (an event per device, created with the
 * first ctx; the host side of a call runs under a mutex per device), so concurrent callers get the results they would
 * get alone, one after the other.  Not covered: a call captured into a HIP graph records no event and waits for none
 * (its replays are ordered by the stream they are launched on; keeping them apart from other users of the array is the
 * caller's business), and outputs left in the array (tile_cnt / kept_cnt / cells / colmap / rowmap passed as NULL)
 * live only until the next call on that GPU. */
#define SLAM_NOALLOC_SCRATCH_BYTES (64u << 20)
int slam_process_frames_noalloc_dev(slam_ctx *ctx, const slam_batch *batch, uint32_t stages);

/* Paint n_frames tiles from cell colours and pixel->cell maps as slam_process_frames_dev leaves them
 * (cells_dev / colmap_dev / rowmap_dev / stats_dev of slam_batch, same strides): the last step of
 * generate_heatmap's mesh (excel_heatmap_v3.py:442-446) on its own.  Device pointers, asynchronous. */
int slam_raster_dev(slam_ctx *ctx, const uint8_t *cells_dev, const int16_t *colmap_dev, const int16_t *rowmap_dev,
                    const uint32_t *stats_dev, uint32_t n_frames, int32_t tile_w, int32_t tile_h, uint8_t *tile_dev,
                    uint32_t cells_per_frame /* slam_cells_per_frame(n_axis) of the producing batch */);

/* Bytes of ctx workspace slam_process_frames_dev needs for this batch shape (it grows the
 * workspace itself on first use; callers that must not allocate later call this once up front).  Also uploads the
 * blur weights of batch->blur_sigma. */
int slam_reserve_workspace(slam_ctx *ctx, const slam_batch *batch);

/* Growing the workspace frees and reallocates it, and a blur sigma other than the reserved one re-uploads the
 * weights: both would silently invalidate every HIP graph captured from calls on this ctx (the captured kernels hold
 * workspace addresses).  While the ctx's stream is being captured such a call fails with SLAM_E_INVALID instead.
 * slam_pin_workspace(ctx, 1) extends that protection past the capture: until unpinned, any call on the ctx (the
 * host-pointer entry points included) that needs more workspace than is reserved fails with SLAM_E_CAPACITY rather
 * than reallocate.  A caller that replays captured graphs pins the ctx after capturing, or keeps a ctx of its own per
 * captured pipeline (slamhip.pipeline.FramePipeline does the latter).  No counterpart in the reference. */
int slam_pin_workspace(slam_ctx *ctx, int pinned);

/* Deterministic synthetic capture generator (device): frame `frame_index` of shape B rows x
 * N records into text_dev (needs slam_synth_text_len(B, N) bytes).  Byte-identical to the
 * NumPy generator slamhip.synth.make_frame_text. */
size_t slam_synth_text_len(uint32_t B, uint32_t N);
int slam_synth_frames_dev(slam_ctx *ctx, uint8_t *text_dev, uint64_t frame_stride,
                          uint32_t first_frame, uint32_t n_frames, uint32_t B, uint32_t N, uint64_t seed);

/* Device bandwidth probe used by the benchmark for the measured-copy ceiling: dst <- src, n bytes. */
int slam_copy_probe_dev(slam_ctx *ctx, void *dst_dev, const void *src_dev, size_t n);
/* Read-only probes shaped like the framing kernels' loads: every wave reads its own contiguous chunk of
 * src (n bytes) with 12 loads per lane.  mode 1: 16-byte loads at a 12-byte lane stride; 2: 12-byte loads;
 * 3: 16-byte loads at a 16-byte lane stride.  scratch_dev: at least 4 writable bytes.
 * Copy-like probes: modes 10/11/12 read n bytes and write n bytes to scratch_dev (plain / non-temporal stores /
 * non-temporal loads and stores); modes 20/21/22 read n bytes and write n/3 (the framing kernel's 3:1 mix). */
int slam_read_probe_dev(slam_ctx *ctx, void *scratch_dev, const void *src_dev, size_t n, int mode);

/* Timing of the kernels a ctx launched since slam_timing_reset: per kernel name, launches and
 * total milliseconds from HIP events recorded on the ctx's stream around each launch (only when
 * enabled; off by default).  names_out receives up to cap entries of 32 chars. */
int slam_timing_enable(slam_ctx *ctx, int on);
/* Time only the kernels named in the comma-separated list (names as slam_timing_read reports them, e.g.
 * "frame_onepass,frame_fused"); NULL or "": every kernel.  Two events per timed launch cost a pipeline of a few dozen
 * launches per batch about 4 % of its rate, so a throughput run times its dominant kernel only.  No counterpart in the
 * reference, which times nothing. */
int slam_timing_select(slam_ctx *ctx, const char *names);
int slam_timing_reset(slam_ctx *ctx);
int slam_timing_read(slam_ctx *ctx, char *names_out, uint32_t *launches_out, float *ms_out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
