/* Every host-pointer entry point of the C ABI once, from plain C, on inputs small enough to check by hand.
 *
 *   gcc -O2 -Iinclude examples/cabi_tour.c -o examples/cabi_tour -Lslam-process_amd/slamhip -lslamhip -lm \
 *       -Wl,-rpath,'$ORIGIN/../slam-process_amd/slamhip'
 *   examples/cabi_tour            (exit status 0 and "tour ok" when every check holds)
 *
 * The expected values follow from the reference's rules: tokens and records from serial_hex_to_excel_v3.py:16-103,
 * the correction from bs_beam_correction.py:43-178, pivot / blur / colours from excel_heatmap_v3.py:264-456. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

/* one 11-byte record as capture text */
static size_t put_record(char *dst, int handover, int ue, int bs, uint32_t clk, uint32_t rss)
{
    uint8_t b[11] = {handover ? 0xCC : 0x33, (uint8_t)(ue & 63), (uint8_t)(0xC0 | (bs & 63))};
    for (int j = 0; j < 5; j++) b[3 + j] = (uint8_t)(0x40 | ((clk >> (6 * j)) & 63));
    for (int j = 0; j < 3; j++) b[8 + j] = (uint8_t)(0x80 | ((rss >> (6 * j)) & 63));
    size_t n = 0;
    for (int j = 0; j < 11; j++) n += (size_t)sprintf(dst + n, "%02X ", b[j]);
    return n;
}

int main(void)
{
    CHECK(slam_ctx_create(0, &ctx) == SLAM_OK);
    CHECK(strstr(slam_version(), "gfx950") != NULL);

    /* ---- a1: tokens.  "zz" and "0X41" are skipped (uppercase X is not the prefix), "0x41" and "4a" are kept */
    const char *loose = "33 zz 0x41\t4a 0X41\n7";
    uint8_t bytes[16];
    size_t nb = 0;
    CHECK(slam_hex_unpack(ctx, (const uint8_t *)loose, strlen(loose), bytes, sizeof bytes, &nb) == SLAM_OK);
    CHECK(nb == 3 && bytes[0] == 0x33 && bytes[1] == 0x41 && bytes[2] == 0x4A);
    uint32_t tidx[8], tbeg[8], tend[8];
    size_t nrej = 0;
    CHECK(slam_token_rejects(ctx, (const uint8_t *)loose, strlen(loose), tidx, tbeg, tend, 8, &nrej) == SLAM_OK);
    CHECK(nrej == 3 && tidx[0] == 1 && tidx[1] == 4 && tidx[2] == 5);
    CHECK(tbeg[0] == 3 && tend[0] == 5 && tbeg[2] == 19 && tend[2] == 20);
    CHECK(slam_token_rejects(ctx, (const uint8_t *)loose, strlen(loose), tidx, tbeg, tend, 2, &nrej) == SLAM_E_CAPACITY && nrej == 3);

    /* ---- a1 + a2: one sweep of 6 records, a handover in the middle that repeats the RSS of the row before, then a
     * second sweep (UE falls back to 0) of 2 records; stray flag bytes and a truncated record around them */
    char text[1024];
    size_t n = 0;
    n += (size_t)sprintf(text + n, "\xC2\xAB ");                       /* the capture tool's leading mark */
    n += (size_t)sprintf(text + n, "33 33 ");                          /* two stray flag bytes: two discards */
    const uint32_t c0 = 1000000;
    n += put_record(text + n, 0, 1, 7, c0, 20000);                     /* its BS is stale: corrected from the baseline */
    n += put_record(text + n, 0, 2, 7, c0 + 61000 + 40, 20001);
    n += put_record(text + n, 1, 3, 30, c0 + 2 * 61000, 20001);        /* handover: baseline (clk of the row before, 30) */
    n += put_record(text + n, 0, 4, 7, c0 + 3 * 61000 - 10, 20002);
    n += put_record(text + n, 0, 5, 7, c0 + 4 * 61000 + 900, 20003);   /* 860 off the grid: beyond TOL, dropped */
    n += put_record(text + n, 0, 6, 7, c0 - 61000 + 40, 20004);        /* one cycle before the baseline */
    n += put_record(text + n, 0, 0, 9, c0 + 9 * 61000, 20005);         /* UE drops: a new group without a baseline */
    n += put_record(text + n, 0, 1, 9, c0 + 10 * 61000, 20006);
    n += (size_t)sprintf(text + n, "CC 01 C2 41 ");                    /* a record cut short by the end: one discard */

    uint8_t flag[16], ue[16], bs[16];
    uint32_t rss[16], clk[16];
    size_t valid = 0, disc = 0;
    CHECK(slam_parse_text(ctx, (const uint8_t *)text, n, flag, ue, bs, rss, clk, 16, &valid, &disc) == SLAM_OK);
    CHECK(valid == 8 && disc == 3);
    CHECK(flag[2] == 1 && ue[2] == 3 && bs[2] == 30 && clk[2] == c0 + 2 * 61000 && rss[2] == 20001);
    CHECK(slam_parse_text(ctx, (const uint8_t *)text, n, flag, ue, bs, rss, clk, 4, &valid, &disc) == SLAM_E_CAPACITY && valid == 8);

    /* the same through the two separate steps, and the discard positions */
    uint8_t data[256];
    size_t nd = 0;
    CHECK(slam_hex_unpack(ctx, (const uint8_t *)text, n, data, sizeof data, &nd) == SLAM_OK && nd == 2 + 8 * 11 + 4);
    size_t v2 = 0, d2 = 0;
    CHECK(slam_parse_records(ctx, data, nd, flag, ue, bs, rss, clk, 16, &v2, &d2) == SLAM_OK && v2 == 8 && d2 == 3);
    uint32_t pos[8];
    size_t np_ = 0;
    CHECK(slam_parse_rejects(ctx, data, nd, pos, 8, &np_) == SLAM_OK);
    CHECK(np_ == 3 && pos[0] == 0 && pos[1] == 1 && pos[2] == 2 + 8 * 11);

    /* ---- a4 - a7: correction */
    int32_t col[5][16], out[4][16], gid[16], corr[16], blc[9], blb[9], blg[9];
    for (size_t i = 0; i < valid; i++) {
        col[0][i] = flag[i]; col[1][i] = ue[i]; col[2][i] = bs[i]; col[3][i] = (int32_t)rss[i]; col[4][i] = (int32_t)clk[i];
    }
    size_t kept = 0, nbl = 0, ngr = 0;
    CHECK(slam_correct(ctx, col[0], col[1], col[2], col[3], col[4], valid, 61000, 500, 64, 0, NULL, NULL, -1,
                       out[0], out[1], out[2], out[3], &kept, gid, corr, blc, blb, blg, &nbl, &ngr) == SLAM_OK);
    CHECK(ngr == 2 && nbl == 1 && blc[0] == (int32_t)(c0 + 61000 + 40) && blb[0] == 30 && blg[0] == 0);
    CHECK(gid[5] == 0 && gid[6] == 1 && gid[7] == 1);
    /* baseline clock = c0 + 61040: rows at -1, 0, +2 and -2 cycles from it qualify; the handover row and the row 860
     * off the grid do not; the second group has no baseline */
    CHECK(kept == 4);
    CHECK(out[1][0] == 29 && out[1][1] == 30 && out[1][2] == 32 && out[1][3] == 28);
    CHECK(out[0][3] == 6 && out[3][3] == (int32_t)(c0 - 61000 + 40));
    CHECK(corr[2] == -1 && corr[4] == -1 && corr[6] == -1 && corr[0] == 29);
    /* explicit baselines, one group (the per-group functions' use) */
    const int32_t ebc[1] = {(int32_t)c0}, ebb[1] = {10};
    CHECK(slam_correct(ctx, col[0], col[1], col[2], col[3], col[4], valid, 61000, 500, 64, 0, ebc, ebb, 1,
                       out[0], out[1], out[2], out[3], &kept, NULL, corr, NULL, NULL, NULL, &nbl, &ngr) == SLAM_OK);
    CHECK(nbl == 1 && corr[0] == 10 && corr[1] == 11 && corr[3] == 13 && corr[6] == 19 && corr[7] == 20 && corr[4] == -1);
    col[4][0] = 1 << 30;
    CHECK(slam_correct(ctx, col[0], col[1], col[2], col[3], col[4], valid, 61000, 500, 64, 0, NULL, NULL, -1,
                       NULL, NULL, NULL, NULL, &kept, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SLAM_E_RANGE);

    /* ---- a9 - a10: pivot over three angles; beam 5 is unmapped, beams 0 and 1 share an angle */
    int16_t b2a[64];
    memset(b2a, -1, sizeof b2a);
    b2a[0] = 0; b2a[1] = 0; b2a[2] = 1; b2a[3] = 2;
    const int32_t pue[6] = {0, 1, 2, 3, 5, 70}, pbs[6] = {2, 2, 3, 0, 2, 2}, prs[6] = {10, 21, 7, 9, 1000, 1000};
    int64_t sum[9], cnt[9];
    double mat[9];
    CHECK(slam_pivot_mean(ctx, pue, pbs, prs, 6, b2a, 3, sum, cnt, mat) == SLAM_OK);
    CHECK(sum[1 * 3 + 0] == 31 && cnt[1 * 3 + 0] == 2 && mat[1 * 3 + 0] == 15.5);       /* row = BS axis, col = UE axis */
    CHECK(sum[2 * 3 + 1] == 7 && cnt[0 * 3 + 2] == 1 && isnan(mat[0]) && cnt[1 * 3 + 1] == 0);
    b2a[9] = 3;
    CHECK(slam_pivot_mean(ctx, pue, pbs, prs, 6, b2a, 3, sum, cnt, mat) == SLAM_E_INVALID);
    b2a[9] = -1;
    /* RSS values that are not whole numbers: per cell the compensated sum in row order, as pandas' pivot mean.
     * 1e16 + 1 + 1 - 1e16: a plain left-to-right sum loses both ones (0), the compensated one keeps them (2) */
    const int32_t fue[7] = {0, 0, 0, 0, 2, 2, 5}, fbs[7] = {2, 2, 2, 2, 3, 3, 2};
    const double frs[7] = {1e16, 1.0, 1.0, -1e16, -70.25, NAN, 3.0};
    double fsum[9];
    CHECK(slam_pivot_mean_f64(ctx, fue, fbs, frs, 7, b2a, 3, fsum, cnt, mat) == SLAM_OK);
    CHECK(cnt[1 * 3 + 0] == 4 && fsum[1 * 3 + 0] == 2.0 && mat[1 * 3 + 0] == 0.5);
    CHECK(cnt[2 * 3 + 1] == 1 && mat[2 * 3 + 1] == -70.25 && isnan(mat[0]));                  /* the NaN value is not counted */
    /* an axis table of 256 entries (the up-scaled beam table): cells per frame of the batched buffers */
    CHECK(slam_cells_per_frame(64) == 4096 && slam_cells_per_frame(65) == 65 * 65 && slam_cells_per_frame(256) == 65536);
    /* a pinned workspace refuses to grow: a call that needs more than is reserved fails instead of reallocating */
    CHECK(slam_pin_workspace(ctx, 1) == SLAM_OK);
    {
        static double big[600 * 600], bigo[600 * 600];
        for (int i = 0; i < 600 * 600; i++) big[i] = 1.0;
        CHECK(slam_blur_nan(ctx, big, 600, 600, 1.0, bigo) == SLAM_E_CAPACITY);
        CHECK(slam_pin_workspace(ctx, 0) == SLAM_OK);
        CHECK(slam_blur_nan(ctx, big, 600, 600, 1.0, bigo) == SLAM_OK && fabs(bigo[300 * 600 + 300] - 1.0) < 1e-12);
    }

    /* ---- a11 - a12: blur.  A constant stays constant, NaN cells are filled from their neighbours, sigma <= 0 copies */
    double m[12], bl[12];
    for (int i = 0; i < 12; i++) m[i] = 4.0;
    m[5] = NAN;
    CHECK(slam_blur_nan(ctx, m, 3, 4, 1.0, bl) == SLAM_OK);
    for (int i = 0; i < 12; i++) CHECK(fabs(bl[i] - 4.0) < 1e-12);
    CHECK(slam_blur_nan(ctx, m, 3, 4, 0.0, bl) == SLAM_OK && isnan(bl[5]) && bl[4] == 4.0);
    CHECK(slam_blur_nan(ctx, m, 3, 4, 1e9, bl) == SLAM_E_RANGE);

    /* ---- a13: colours and the tile.  Linear scale over 1..3: the ends take the LUT's ends, NaN stays transparent */
    uint8_t lut[768];
    for (int i = 0; i < 256; i++) { lut[3 * i] = (uint8_t)i; lut[3 * i + 1] = (uint8_t)(255 - i); lut[3 * i + 2] = 7; }
    const double rm[4] = {1.0, 3.0, NAN, 2.0}, aod[2] = {-10.0, 10.0}, aoa[2] = {0.0, 5.0};
    uint8_t cells[16], tile[8 * 8 * 4];
    double lim[2];
    CHECK(slam_render(ctx, rm, 2, 2, aod, aoa, 0, NAN, NAN, lut, cells, tile, 8, 8, lim) == SLAM_OK);
    CHECK(lim[0] == 1.0 && lim[1] == 3.0);
    CHECK(cells[0] == 0 && cells[3] == 255 && cells[4] == 255 && cells[5] == 0 && cells[11] == 0 && cells[12] == 128);
    /* row 0 of the tile is the largest AoD: matrix row 1 -> (NaN, 2.0); the last tile row is matrix row 0 */
    CHECK(tile[3] == 0 && tile[7 * 4 + 0] == 128 && tile[(7 * 8 + 0) * 4 + 0] == 0 && tile[(7 * 8 + 7) * 4 + 0] == 255);
    const double allnan[4] = {NAN, NAN, NAN, NAN};
    CHECK(slam_render(ctx, allnan, 2, 2, aod, aoa, 0, NAN, NAN, lut, cells, NULL, 0, 0, lim) == SLAM_E_ALLNAN);
    CHECK(slam_render(ctx, rm, 2, 2, aod, aoa, 1, -5.0, NAN, lut, cells, NULL, 0, 0, lim) == SLAM_E_LIMITS);

    /* ---- next to the path: the estimators' correlation Phi_RX^T . M . Phi_TX (heatmap_gemini_v4.py:209).  A dictionary with a
     * single one picks a row / a column of the matrix into place; two beams with weights give the weighted sum */
    {
        double prx[4 * 6] = {0}, ptx[3 * 4] = {0}, mm[2 * 4 * 3], out[2 * 6 * 4];
        prx[2 * 6 + 5] = 1.0;                      /* beam 2 answers at grid point 5 */
        prx[0 * 6 + 1] = 0.5; prx[3 * 6 + 1] = 2.0;  /* grid point 1: half of beam 0 plus twice beam 3 */
        ptx[1 * 4 + 0] = 1.0;                      /* BS beam 1 answers at grid point 0 */
        ptx[2 * 4 + 3] = 3.0;
        for (int i = 0; i < 24; i++) mm[i] = (double)(i < 12 ? i : 100 - i);
        CHECK(slam_dict_correlate(ctx, prx, 4, 6, ptx, 3, 4, mm, 2, out) == SLAM_OK);
        CHECK(out[5 * 4 + 0] == mm[2 * 3 + 1] && out[5 * 4 + 3] == 3.0 * mm[2 * 3 + 2] && out[5 * 4 + 1] == 0.0);
        CHECK(out[1 * 4 + 0] == 0.5 * mm[0 * 3 + 1] + 2.0 * mm[3 * 3 + 1] && out[0] == 0.0);
        CHECK(out[24 + 5 * 4 + 0] == mm[12 + 2 * 3 + 1]);                               /* the second matrix of the stack */
        CHECK(slam_dict_correlate(ctx, prx, 65, 6, ptx, 3, 4, mm, 1, out) == SLAM_E_RANGE);
    }

    CHECK(slam_ctx_destroy(ctx) == SLAM_OK);
    printf("tour ok: %d checks\n", checks);
    return 0;
}
