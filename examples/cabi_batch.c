/* The batched, device-resident entry points of the C ABI from plain C: a batch of synthetic captures generated on the
 * device, slam_process_frames_dev on it, the same batch through slam_process_frames_noalloc_dev, the tiles repainted
 * from the compact results with slam_raster_dev, frame 0 against the host-pointer entry points, and the kernel timers.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/cabi_batch.c -o examples/cabi_batch \
 *       -Lslam-process_amd/slamhip -lslamhip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,'$ORIGIN/../slam-process_amd/slamhip' -Wl,-rpath,/opt/rocm/lib
 *   examples/cabi_batch            (exit status 0 and "batch ok" when every check holds)
 *
 * What one call replaces per frame: main() -> _read_bytes -> _parse_groups (serial_hex_to_excel_v3.py:153-154), the
 * data path of process_excel_filtered (bs_beam_correction.py:186-198) and of process_excel_to_heatmap
 * (excel_heatmap_v3.py:519-545). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "slamhip.h"

static slam_ctx *ctx;
static int checks;

#define CHECK(cond)                                                                              \
    do {                                                                                         \
        checks++;                                                                                \
        if (!(cond)) {                                                                           \
            fprintf(stderr, "line %d: %s does not hold (last error: %s)\n", __LINE__, #cond, slam_last_error(ctx)); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static void *dev_alloc(size_t n)
{
    void *p = NULL;
    if (hipMalloc(&p, n ? n : 1) != hipSuccess) { fprintf(stderr, "hipMalloc of %zu bytes failed\n", n); exit(2); }
    (void)hipMemset(p, 0, n ? n : 1);
    return p;
}

static void *to_host(const void *dev, size_t n)
{
    void *h = malloc(n ? n : 1);
    if (hipMemcpy(h, dev, n, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy to host failed\n"); exit(2); }
    return h;
}

enum { F = 6, B = 8, N = 512, W = 160, H = 96, NA = 64 };

/* every output buffer of one batch */
typedef struct outputs {
    uint8_t *flag, *ue, *bs, *kue, *kbs, *tile, *cells;
    uint32_t *rss, *clk, *krss, *kclk, *tile_cnt, *kept_cnt, *stats;
    int16_t *colmap, *rowmap;
    double *matrix, *blurred;
} outputs;

static void alloc_outputs(outputs *o, uint32_t row_cap, uint32_t T, uint32_t KB)
{
    const size_t R = (size_t)F * row_cap;
    o->flag = dev_alloc(R); o->ue = dev_alloc(R); o->bs = dev_alloc(R);
    o->rss = dev_alloc(R * 4); o->clk = dev_alloc(R * 4);
    o->kue = dev_alloc(R); o->kbs = dev_alloc(R); o->krss = dev_alloc(R * 4); o->kclk = dev_alloc(R * 4);
    o->tile_cnt = dev_alloc((size_t)F * T * 4); o->kept_cnt = dev_alloc((size_t)F * KB * 4);
    o->stats = dev_alloc((size_t)F * SLAM_STAT_WORDS * 4);
    o->matrix = dev_alloc((size_t)F * NA * NA * 8); o->blurred = dev_alloc((size_t)F * NA * NA * 8);
    o->tile = dev_alloc((size_t)F * W * H * 4);
    o->cells = dev_alloc((size_t)F * slam_cells_per_frame(NA) * 4);
    o->colmap = dev_alloc((size_t)F * W * 2); o->rowmap = dev_alloc((size_t)F * H * 2);
}

static void point_outputs(slam_batch *bt, const outputs *o)
{
    bt->flag_dev = o->flag; bt->ue_dev = o->ue; bt->bs_dev = o->bs; bt->rss_dev = o->rss; bt->clk_dev = o->clk;
    bt->kept_ue_dev = o->kue; bt->kept_bs_dev = o->kbs; bt->kept_rss_dev = o->krss; bt->kept_clk_dev = o->kclk;
    bt->tile_cnt_dev = o->tile_cnt; bt->kept_cnt_dev = o->kept_cnt; bt->stats_dev = o->stats;
    bt->matrix_dev = o->matrix; bt->blurred_dev = o->blurred; bt->tile_dev = o->tile;
    bt->cells_dev = o->cells; bt->colmap_dev = o->colmap; bt->rowmap_dev = o->rowmap;
}

int main(void)
{
    CHECK(slam_ctx_create(0, &ctx) == SLAM_OK);
    CHECK(slam_batch_sizeof() == sizeof(slam_batch));      /* this file's view of the struct is the library's */

    /* ---- the batch: F synthetic captures of B sweeps x N records, generated where they will be read */
    const size_t tl = slam_synth_text_len(B, N);
    const size_t stride = (tl + 255) / 256 * 256;
    uint8_t *text = dev_alloc(F * stride + 64);
    CHECK(slam_synth_frames_dev(ctx, text, stride, 0, F, B, N, 20260126ull) == SLAM_OK);
    uint64_t off_h[F];
    uint32_t len_h[F];
    for (int f = 0; f < F; f++) { off_h[f] = (uint64_t)f * stride; len_h[f] = (uint32_t)tl; }
    uint64_t *off = dev_alloc(sizeof off_h);
    uint32_t *len = dev_alloc(sizeof len_h);
    CHECK(hipMemcpy(off, off_h, sizeof off_h, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(len, len_h, sizeof len_h, hipMemcpyHostToDevice) == hipSuccess);

    /* beam table: 64 beams at 1.4 degree steps (read_mapping's dict, excel_heatmap_v3.py:187-258, as two arrays);
     * colour table: any 256 x 3 bytes */
    int16_t b2a_h[64];
    double ang_h[NA];
    uint8_t lut_h[768];
    for (int i = 0; i < 64; i++) { b2a_h[i] = (int16_t)i; ang_h[i] = -44.1 + 1.4 * i; }
    for (int i = 0; i < 256; i++) { lut_h[3 * i] = (uint8_t)i; lut_h[3 * i + 1] = (uint8_t)(255 - i); lut_h[3 * i + 2] = (uint8_t)(i / 2); }
    int16_t *b2a = dev_alloc(sizeof b2a_h);
    double *ang = dev_alloc(sizeof ang_h);
    uint8_t *lut = dev_alloc(sizeof lut_h);
    CHECK(hipMemcpy(b2a, b2a_h, sizeof b2a_h, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(ang, ang_h, sizeof ang_h, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(lut, lut_h, sizeof lut_h, hipMemcpyHostToDevice) == hipSuccess);

    const uint32_t T = slam_row_tiles((uint32_t)tl, 0);
    const uint32_t row_cap = (T * SLAM_TILE_SLOTS + 15u) & ~15u;
    const uint32_t KB = slam_kept_blocks(row_cap, (uint32_t)tl, 0);
    outputs a, b;
    alloc_outputs(&a, row_cap, T, KB);
    alloc_outputs(&b, row_cap, T, KB);

    slam_batch bt;
    memset(&bt, 0, sizeof bt);
    bt.text_dev = text; bt.text_off_dev = off; bt.text_len_dev = len;
    bt.n_frames = F; bt.max_text_len = (uint32_t)tl; bt.src_is_bytes = 0;
    bt.cycle = 61000; bt.tol = 500; bt.mod_base = 64;          /* CYCLE, TOL, MOD_BASE (bs_beam_correction.py:23-25) */
    bt.beam_to_axis_dev = b2a; bt.axis_angle_dev = ang; bt.n_axis = NA; bt.full_grid = 0;
    bt.blur_sigma = 1.0; bt.use_log = 0; bt.vmin = NAN; bt.vmax = NAN; bt.lut_dev = lut;
    bt.tile_w = W; bt.tile_h = H; bt.row_cap = row_cap;
    point_outputs(&bt, &a);

    /* ---- one call on the ctx workspace (reserved up front, so the call itself only enqueues kernels) */
    CHECK(slam_reserve_workspace(ctx, &bt) == SLAM_OK);
    CHECK(slam_timing_enable(ctx, 1) == SLAM_OK);
    CHECK(slam_process_frames_dev(ctx, &bt, 7) == SLAM_OK);
    CHECK(slam_sync(ctx) == SLAM_OK);
    CHECK(slam_timing_enable(ctx, 0) == SLAM_OK);

    uint32_t *st = to_host(a.stats, (size_t)F * SLAM_STAT_WORDS * 4);
    uint32_t *tc = to_host(a.tile_cnt, (size_t)F * T * 4);
    uint32_t *kc = to_host(a.kept_cnt, (size_t)F * KB * 4);
    uint8_t *tile_a = to_host(a.tile, (size_t)F * W * H * 4);
    for (int f = 0; f < F; f++) {
        const uint32_t *s = st + f * SLAM_STAT_WORDS;
        CHECK(s[SLAM_STAT_NOT_STRICT] == 0 && (s[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_TWO_PASS) == 0);
        CHECK(s[SLAM_STAT_NBYTES] == (tl + 1) / 3 - 1);                       /* the text starts with the 3-byte mark */
        CHECK(s[SLAM_STAT_VALID] > B * N * 9 / 10 && s[SLAM_STAT_VALID] <= B * N);
        CHECK(s[SLAM_STAT_GROUPS] == B);                                      /* one group per sweep (UE falls back) */
        CHECK(s[SLAM_STAT_BASELINES] > 0 && s[SLAM_STAT_KEPT] > 0 && s[SLAM_STAT_KEPT] < s[SLAM_STAT_VALID]);
        CHECK(s[SLAM_STAT_ALLNAN] == 0 && s[SLAM_STAT_N_AOD] > 0 && s[SLAM_STAT_N_AOA] == NA);
        uint32_t rows = 0, kept = 0;
        for (uint32_t t = 0; t < T; t++) rows += tc[f * T + t];
        const uint32_t ku = (s[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_KEPT_TILES) ? SLAM_TILE_SLOTS : SLAM_KEPT_BLOCK;
        for (uint32_t k = 0; k < (s[SLAM_STAT_ROW_SLOTS] + ku - 1) / ku; k++) kept += kc[f * KB + k];   /* the frame's blocks */
        CHECK(rows == s[SLAM_STAT_VALID] && kept == s[SLAM_STAT_KEPT]);       /* the slots' counts add up to the counters */
        size_t opaque = 0, clear = 0;                   /* a cell with no finite value within the blur's reach stays transparent */
        for (size_t px = 0; px < (size_t)W * H; px++) {
            const uint8_t alpha = tile_a[((size_t)f * W * H + px) * 4 + 3];
            opaque += alpha == 255;
            clear += alpha == 0;
        }
        CHECK(opaque + clear == (size_t)W * H && opaque > (size_t)W * H / 2);
    }

    /* ---- frame 0 through the host-pointer entry points: the same records, the same kept rows */
    {
        uint8_t *text0 = to_host(text, tl);
        const size_t cap = (size_t)B * N + 16;
        uint8_t *fl = malloc(cap), *ue = malloc(cap), *bs = malloc(cap);
        uint32_t *rs = malloc(cap * 4), *ck = malloc(cap * 4);
        size_t nv = 0, nd = 0;
        CHECK(slam_parse_text(ctx, text0, tl, fl, ue, bs, rs, ck, cap, &nv, &nd) == SLAM_OK);
        CHECK(nv == st[SLAM_STAT_VALID] && nd == st[SLAM_STAT_DISCARDED]);
        uint8_t *d_ue = to_host(a.ue, row_cap);
        uint32_t *d_ck = to_host(a.clk, (size_t)row_cap * 4), *d_rs = to_host(a.rss, (size_t)row_cap * 4);
        size_t i = 0;
        int same = 1;
        for (uint32_t t = 0; t < T; t++)                       /* concatenating the tiles' used entries gives the list */
            for (uint32_t k = 0; k < tc[t]; k++, i++) {
                const size_t e = (size_t)t * SLAM_TILE_SLOTS + k;
                same &= d_ue[e] == ue[i] && d_ck[e] == ck[i] && d_rs[e] == rs[i];
            }
        CHECK(i == nv && same);
        int32_t *c5[5];
        for (int c = 0; c < 5; c++) c5[c] = malloc(nv * 4);
        for (size_t r = 0; r < nv; r++) { c5[0][r] = fl[r]; c5[1][r] = ue[r]; c5[2][r] = bs[r]; c5[3][r] = (int32_t)rs[r]; c5[4][r] = (int32_t)ck[r]; }
        int32_t *k4[4];
        for (int c = 0; c < 4; c++) k4[c] = malloc(nv * 4);
        size_t nk = 0, nbl = 0, ng = 0;
        CHECK(slam_correct(ctx, c5[0], c5[1], c5[2], c5[3], c5[4], nv, 61000, 500, 64, 0, NULL, NULL, -1, k4[0], k4[1], k4[2], k4[3],
                           &nk, NULL, NULL, NULL, NULL, NULL, &nbl, &ng) == SLAM_OK);
        CHECK(nk == st[SLAM_STAT_KEPT] && nbl == st[SLAM_STAT_BASELINES] && ng == st[SLAM_STAT_GROUPS]);
        uint8_t *d_kbs = to_host(a.kbs, row_cap);
        uint32_t *d_kck = to_host(a.kclk, (size_t)row_cap * 4);
        i = 0;
        same = 1;
        /* blocks of the corrected-row arrays: SLAM_KEPT_BLOCK entries, or one tile's entries where the fused kernel served the frame */
        const uint32_t unit = (st[SLAM_STAT_LAYOUT] & SLAM_LAYOUT_KEPT_TILES) ? SLAM_TILE_SLOTS : SLAM_KEPT_BLOCK;
        for (uint32_t blk = 0; blk < KB && (size_t)blk * unit < row_cap; blk++)
            for (uint32_t k = 0; k < kc[blk]; k++, i++) {
                const size_t e = (size_t)blk * unit + k;
                same &= (int32_t)d_kbs[e] == k4[1][i] && (int32_t)d_kck[e] == k4[3][i];
            }
        CHECK(i == nk && same);
        free(text0); free(fl); free(ue); free(bs); free(rs); free(ck); free(d_ue); free(d_ck); free(d_rs); free(d_kbs); free(d_kck);
        for (int c = 0; c < 5; c++) free(c5[c]);
        for (int c = 0; c < 4; c++) free(k4[c]);
    }

    /* ---- the same batch on the library's fixed scratch: nothing allocated, the same bytes */
    point_outputs(&bt, &b);
    CHECK(slam_process_frames_noalloc_dev(ctx, &bt, 7) == SLAM_OK);
    CHECK(slam_sync(ctx) == SLAM_OK);
    {
        uint32_t *st_b = to_host(b.stats, (size_t)F * SLAM_STAT_WORDS * 4);
        uint8_t *tile_b = to_host(b.tile, (size_t)F * W * H * 4);
        double *m_a = to_host(a.matrix, (size_t)F * NA * NA * 8), *m_b = to_host(b.matrix, (size_t)F * NA * NA * 8);
        CHECK(memcmp(st, st_b, (size_t)F * SLAM_STAT_WORDS * 4) == 0);
        CHECK(memcmp(tile_a, tile_b, (size_t)F * W * H * 4) == 0);
        CHECK(memcmp(m_a, m_b, (size_t)F * NA * NA * 8) == 0);
        free(st_b); free(tile_b); free(m_a); free(m_b);
    }
    /* a batch beyond the fixed scratch is refused, not served from somewhere else */
    {
        slam_batch big = bt;
        big.max_text_len = 40u << 20;
        big.row_cap = (slam_row_tiles(big.max_text_len, 0) * SLAM_TILE_SLOTS + 15u) & ~15u;
        CHECK(slam_process_frames_noalloc_dev(ctx, &big, 1) == SLAM_E_CAPACITY);
    }

    /* ---- what a collecting rank does: the tiles again from the 20 KB per frame they were painted from */
    {
        uint8_t *again = dev_alloc((size_t)F * W * H * 4);
        CHECK(slam_raster_dev(ctx, a.cells, a.colmap, a.rowmap, a.stats, F, W, H, again, slam_cells_per_frame(NA)) == SLAM_OK);
        CHECK(slam_sync(ctx) == SLAM_OK);
        uint8_t *again_h = to_host(again, (size_t)F * W * H * 4);
        CHECK(memcmp(again_h, tile_a, (size_t)F * W * H * 4) == 0);
        free(again_h);
        (void)hipFree(again);
    }

    /* ---- refused arguments come back as codes with a message, nothing is launched */
    {
        slam_batch bad = bt;
        bad.row_cap = row_cap - 16;
        CHECK(slam_process_frames_dev(ctx, &bad, 7) == SLAM_E_INVALID && strlen(slam_last_error(ctx)) > 0);
        bad = bt;
        bad.n_axis = SLAM_MAX_AXIS + 1;
        CHECK(slam_process_frames_dev(ctx, &bad, 7) == SLAM_E_INVALID);
        bad = bt;
        bad.cycle = 0;
        CHECK(slam_process_frames_dev(ctx, &bad, 2) == SLAM_E_INVALID);
    }

    /* ---- the kernels of the first call, by the library's own event timers */
    {
        char names[64 * 32];
        uint32_t launches[64];
        float ms[64];
        const int n = slam_timing_read(ctx, names, launches, ms, 64);
        CHECK(n > 5 && n <= 64);
        int seen_framing = 0;
        for (int i = 0; i < n; i++) {
            printf("  %-20s %u launch(es) %.3f ms\n", names + 32 * i, launches[i], ms[i]);
            seen_framing |= strcmp(names + 32 * i, "frame_onepass") == 0 || strcmp(names + 32 * i, "frame_fused") == 0;
        }
        CHECK(seen_framing);
        CHECK(slam_timing_reset(ctx) == SLAM_OK);
    }

    free(st); free(tc); free(kc); free(tile_a);
    CHECK(slam_ctx_destroy(ctx) == SLAM_OK);
    printf("batch ok (%d checks)\n", checks);
    return 0;
}
