/*
 * slam_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, sequential restatement of the reference's
 * parse -> correct -> pivot algorithms.  It exists to check the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in
 * the product package may link, load or call it.
 *
 * Every function cites the reference lines it restates.  The code is written
 * the obvious sequential way on purpose: it is the checker, not the product.
 *
 * Pinned against: the reference's own functions imported in the build
 * container, the 19 committed debugDoc/<name>.xlsx / <name>_filtered.xlsx
 * outputs, and the reference's embedded unit tests (see tests/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* a1. _read_bytes  (serial_hex_to_excel_v3.py:18-36, regex :16)              */
/* ------------------------------------------------------------------------- */

/* Py_UNICODE_ISSPACE: the set str.split() splits on (serial_hex_to_excel_v3.py:20). */
static int orc_is_space(uint32_t c)
{
    if (c < 0x80)
        return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20);
    return c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
           c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

static int orc_is_cont(uint8_t b) { return (b & 0xC0) == 0x80; }

/*
 * One step of CPython's UTF-8 decoder with errors="ignore"
 * (Path.read_text(encoding="utf-8", errors="ignore"), serial_hex_to_excel_v3.py:19).
 * Returns the number of input bytes consumed; *cp is the decoded code point or
 * 0xFFFFFFFF when the consumed bytes are dropped.
 * Error spans follow CPython: invalid start byte -> 1 byte; invalid
 * continuation at position k -> the k bytes before it; truncated sequence at
 * end of input -> everything that is left.
 */
static int64_t orc_utf8_step(const uint8_t *s, int64_t left, uint32_t *cp)
{
    uint8_t ch = s[0];
    *cp = 0xFFFFFFFFu;
    if (ch < 0x80) { *cp = ch; return 1; }
    if (ch < 0xC2) return 1;                       /* stray continuation / overlong C0,C1 */
    if (ch < 0xE0) {
        if (left < 2) return left;                 /* unexpected end */
        if (!orc_is_cont(s[1])) return 1;
        *cp = ((uint32_t)(ch & 0x1F) << 6) | (s[1] & 0x3F);
        return 2;
    }
    if (ch < 0xF0) {
        if (left < 2) return left;
        uint8_t c2 = s[1];
        if (!orc_is_cont(c2) || (c2 < 0xA0 ? ch == 0xE0 : ch == 0xED)) return 1;
        if (left < 3) return left;
        if (!orc_is_cont(s[2])) return 2;
        *cp = ((uint32_t)(ch & 0x0F) << 12) | ((uint32_t)(c2 & 0x3F) << 6) | (s[2] & 0x3F);
        return 3;
    }
    if (ch < 0xF5) {
        if (left < 2) return left;
        uint8_t c2 = s[1];
        if (!orc_is_cont(c2) || (c2 < 0x90 ? ch == 0xF0 : ch == 0xF4)) return 1;
        if (left < 3) return left;
        if (!orc_is_cont(s[2])) return 2;
        if (left < 4) return left;
        if (!orc_is_cont(s[3])) return 3;
        *cp = ((uint32_t)(ch & 0x07) << 18) | ((uint32_t)(c2 & 0x3F) << 12) |
              ((uint32_t)(s[2] & 0x3F) << 6) | (s[3] & 0x3F);
        return 4;
    }
    return 1;                                      /* F5..FF */
}

static int orc_hexval(uint32_t c)
{
    if (c >= '0' && c <= '9') return (int)(c - '0');
    if (c >= 'a' && c <= 'f') return (int)(c - 'a' + 10);
    if (c >= 'A' && c <= 'F') return (int)(c - 'A' + 10);
    return -1;
}

/*
 * text[n] -> out[] (capacity >= n/2 + 1).  Returns the number of bytes produced.
 * Token rule (serial_hex_to_excel_v3.py:16,22-35): a whitespace-separated
 * token is kept iff it is exactly two hex digits, optionally preceded by a
 * lowercase "0x".
 */
int64_t orc_read_bytes(const uint8_t *text, int64_t n, uint8_t *out)
{
    int64_t i = 0, n_out = 0;
    uint32_t tok[5];
    int tlen = 0;                                  /* code points in the current token, capped at 5 */
    while (i <= n) {
        uint32_t cp;
        int end_tok;
        if (i == n) { end_tok = 1; i++; }
        else {
            int64_t used = orc_utf8_step(text + i, n - i, &cp);
            i += used;
            if (cp == 0xFFFFFFFFu) continue;       /* dropped bytes join their neighbours */
            end_tok = orc_is_space(cp);
        }
        if (!end_tok) {
            if (tlen < 5) tok[tlen] = cp;
            if (tlen < 5) tlen++;
            continue;
        }
        if (tlen == 2) {
            int h = orc_hexval(tok[0]), l = orc_hexval(tok[1]);
            if (h >= 0 && l >= 0) out[n_out++] = (uint8_t)((h << 4) | l);
        } else if (tlen == 4 && tok[0] == '0' && tok[1] == 'x') {
            int h = orc_hexval(tok[2]), l = orc_hexval(tok[3]);
            if (h >= 0 && l >= 0) out[n_out++] = (uint8_t)((h << 4) | l);
        }
        tlen = 0;
    }
    return n_out;
}

/* ------------------------------------------------------------------------- */
/* a2. _parse_groups  (serial_hex_to_excel_v3.py:41-103)                      */
/* ------------------------------------------------------------------------- */

static int orc_top2(uint8_t v) { return (v >> 6) & 3; }   /* :10-11 */
static uint32_t orc_lsb6(uint8_t v) { return v & 0x3F; }  /* :13-14 */

/*
 * reason codes written to reject_reason[] (optional, may be NULL) mirror the
 * order of the reference's checks: 1 tail (<11 bytes left, :53-58), 2 UE tag
 * (:62-66), 3 BS tag (:67-71), 4 CLK tag (:73-79), 5 RSS tag (:81-87).
 */
void orc_parse_groups(const uint8_t *b, int64_t n,
                      uint8_t *flag, uint8_t *ue, uint8_t *bs, uint32_t *rss, uint32_t *clk,
                      int64_t *n_valid, int64_t *n_discarded,
                      int64_t *reject_pos, uint8_t *reject_reason, int64_t reject_cap)
{
    int64_t i = 0, valid = 0, disc = 0;
    while (i < n) {
        uint8_t f = b[i];
        if (f != 0xCC && f != 0x33) { i++; continue; }           /* :49-51, not counted */
        int reason = 0;
        if (i + 11 > n) reason = 1;
        else if (orc_top2(b[i + 1]) != 0) reason = 2;
        else if (orc_top2(b[i + 2]) != 3) reason = 3;
        else {
            for (int j = 3; j < 8 && !reason; j++) if (orc_top2(b[i + j]) != 1) reason = 4;
            for (int j = 8; j < 11 && !reason; j++) if (orc_top2(b[i + j]) != 2) reason = 5;
        }
        if (reason) {
            if (reject_pos && disc < reject_cap) { reject_pos[disc] = i; reject_reason[disc] = (uint8_t)reason; }
            disc++;
            if (reason == 1) break;                                  /* :58 */
            i++;
            continue;
        }
        uint32_t c = 0, r = 0;
        for (int j = 0; j < 5; j++) c |= orc_lsb6(b[i + 3 + j]) << (6 * j);   /* :93-95 */
        for (int j = 0; j < 3; j++) r |= orc_lsb6(b[i + 8 + j]) << (6 * j);   /* :97 */
        flag[valid] = (f == 0xCC) ? 1 : 0;                                   /* :89 */
        ue[valid] = (uint8_t)orc_lsb6(b[i + 1]);
        bs[valid] = (uint8_t)orc_lsb6(b[i + 2]);
        rss[valid] = r;
        clk[valid] = c;
        valid++;
        i += 11;
    }
    *n_valid = valid;
    *n_discarded = disc;
}

/* ------------------------------------------------------------------------- */
/* a4. _detect_groups  (bs_beam_correction.py:43-47)                          */
/* ------------------------------------------------------------------------- */

/* gid[i] = cumsum(boundary) - 1 with boundary = (ue[i-1] > ue[i]) or i == 0.  Returns #groups. */
int64_t orc_detect_groups(const int64_t *ue, int64_t n, int64_t *gid)
{
    int64_t g = -1;
    for (int64_t i = 0; i < n; i++) {
        if (i == 0 || ue[i - 1] > ue[i]) g++;
        gid[i] = g;
    }
    return g + 1;
}

/* ------------------------------------------------------------------------- */
/* a5. _identify_baselines  (bs_beam_correction.py:50-61)                     */
/* ------------------------------------------------------------------------- */

/*
 * Over rows [lo, hi) of ONE group (the shift(1) is within the group, so row lo
 * can never be a baseline).  Appends (CLK[i-1], BS[i]) in row order.
 */
static int64_t orc_baselines_of_group(const int64_t *flag, const int64_t *bs, const int64_t *rss,
                                      const int64_t *clk, int64_t lo, int64_t hi,
                                      int64_t *bl_clk, int64_t *bl_bs)
{
    int64_t nb = 0;
    for (int64_t i = lo + 1; i < hi; i++) {
        if (flag[i] == 1 && flag[i - 1] == 0 && rss[i] == rss[i - 1]) {
            bl_clk[nb] = clk[i - 1];
            bl_bs[nb] = bs[i];
            nb++;
        }
    }
    return nb;
}

/* All groups of a table; bl_gid[j] is the group of baseline j.  Returns the baseline count. */
int64_t orc_identify_baselines(const int64_t *flag, const int64_t *ue, const int64_t *bs,
                               const int64_t *rss, const int64_t *clk, int64_t n,
                               int64_t *bl_clk, int64_t *bl_bs, int64_t *bl_gid)
{
    int64_t nb = 0, lo = 0, g = 0;
    while (lo < n) {
        int64_t hi = lo + 1;
        while (hi < n && !(ue[hi - 1] > ue[hi])) hi++;
        int64_t k = orc_baselines_of_group(flag, bs, rss, clk, lo, hi, bl_clk + nb, bl_bs + nb);
        for (int64_t j = 0; j < k; j++) bl_gid[nb + j] = g;
        nb += k;
        lo = hi;
        g++;
    }
    return nb;
}

/* ------------------------------------------------------------------------- */
/* a6. candidate search shared by _correct_group / _filter_group_corrected    */
/*     (bs_beam_correction.py:95-104 and :152-161)                            */
/* ------------------------------------------------------------------------- */

static int64_t orc_pymod(int64_t a, int64_t m)           /* Python's % for m > 0 */
{
    int64_t r = a % m;
    return r < 0 ? r + m : r;
}

/*
 * Returns 1 and *corrected when some baseline has resid <= tol; the winner is
 * the smallest resid, earliest baseline on ties (stable sort, :103 / :160).
 * k = int(round(d / CYCLE)): float division, round-half-even (:98 / :155).
 */
static int orc_best_candidate(int64_t clk, const int64_t *bl_clk, const int64_t *bl_bs, int64_t nb,
                              int64_t cycle, int64_t tol, int64_t mod_base, int64_t *corrected)
{
    int found = 0;
    int64_t best_resid = 0;
    for (int64_t j = 0; j < nb; j++) {
        int64_t d = clk - bl_clk[j];
        int64_t k = (int64_t)nearbyint((double)d / (double)cycle);
        int64_t resid = llabs(d - k * cycle);
        if (resid <= tol) {
            if (!found || resid < best_resid) {
                found = 1;
                best_resid = resid;
                *corrected = orc_pymod(bl_bs[j] + k, mod_base);
            }
        }
    }
    return found;
}

/*
 * process_excel_filtered's data path (bs_beam_correction.py:186-198):
 * groups in order, per group _identify_baselines then _filter_group_corrected
 * (:140-178); kept rows concatenated in order.  Returns the kept-row count.
 * n_groups / n_baselines (optional) receive the totals.
 */
int64_t orc_filter_corrected(const int64_t *flag, const int64_t *ue, const int64_t *bs,
                             const int64_t *rss, const int64_t *clk, int64_t n,
                             int64_t cycle, int64_t tol, int64_t mod_base,
                             int64_t *out_ue, int64_t *out_bs, int64_t *out_rss, int64_t *out_clk,
                             int64_t *n_groups, int64_t *n_baselines)
{
    int64_t *bl_clk = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *bl_bs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t kept = 0, lo = 0, g = 0, nb_total = 0;
    while (lo < n) {
        int64_t hi = lo + 1;
        while (hi < n && !(ue[hi - 1] > ue[hi])) hi++;
        int64_t nb = orc_baselines_of_group(flag, bs, rss, clk, lo, hi, bl_clk, bl_bs);
        nb_total += nb;
        if (nb > 0) {
            for (int64_t i = lo; i < hi; i++) {
                if (flag[i] == 1) continue;                          /* :145-146 */
                int64_t c;
                if (orc_best_candidate(clk[i], bl_clk, bl_bs, nb, cycle, tol, mod_base, &c)) {
                    out_ue[kept] = ue[i];
                    out_bs[kept] = c;
                    out_rss[kept] = rss[i];
                    out_clk[kept] = clk[i];
                    kept++;
                }
            }
        }
        lo = hi;
        g++;
    }
    free(bl_clk);
    free(bl_bs);
    if (n_groups) *n_groups = g;
    if (n_baselines) *n_baselines = nb_total;
    return kept;
}

/*
 * process_excel's data path (bs_beam_correction.py:125-134, _correct_group :79-116):
 * every row keeps a value; corrected_bs[i] is the winner when one exists, else
 * the row's own BS.  was_corrected[i] = 1 only where a candidate won.
 */
void orc_correct_inplace(const int64_t *flag, const int64_t *ue, const int64_t *bs,
                         const int64_t *rss, const int64_t *clk, int64_t n,
                         int64_t cycle, int64_t tol, int64_t mod_base,
                         int64_t *corrected_bs, uint8_t *was_corrected)
{
    int64_t *bl_clk = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *bl_bs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t lo = 0;
    while (lo < n) {
        int64_t hi = lo + 1;
        while (hi < n && !(ue[hi - 1] > ue[hi])) hi++;
        int64_t nb = orc_baselines_of_group(flag, bs, rss, clk, lo, hi, bl_clk, bl_bs);
        for (int64_t i = lo; i < hi; i++) {
            int64_t c;
            corrected_bs[i] = bs[i];
            was_corrected[i] = 0;
            if (flag[i] == 1 || nb == 0) continue;
            if (orc_best_candidate(clk[i], bl_clk, bl_bs, nb, cycle, tol, mod_base, &c)) {
                corrected_bs[i] = c;
                was_corrected[i] = 1;
            }
        }
        lo = hi;
    }
    free(bl_clk);
    free(bl_bs);
}

/* ------------------------------------------------------------------------- */
/* a9+a10. map_angles + compute_rssi_matrix, integer part                     */
/*         (excel_heatmap_v3.py:264-280, :286-314)                            */
/* ------------------------------------------------------------------------- */

/*
 * beam_to_axis[id] (id 0..63) is the index of the beam's angle in the sorted
 * list of distinct mapped angles, or -1 when the beam is unmapped (such rows
 * are dropped, :275).  Beams outside 0..63 are dropped (:68-72, :166-168).
 * Accumulates the exact integer sum and count per (AoD axis index of BS,
 * AoA axis index of UE); the mean is sum/count in f64 (pivot_table mean).
 */
void orc_pivot(const int64_t *ue, const int64_t *bs, const int64_t *rss, int64_t n,
               const int16_t *beam_to_axis, int n_axis, int64_t *sum, int64_t *cnt)
{
    memset(sum, 0, sizeof(int64_t) * (size_t)n_axis * (size_t)n_axis);
    memset(cnt, 0, sizeof(int64_t) * (size_t)n_axis * (size_t)n_axis);
    for (int64_t i = 0; i < n; i++) {
        if (ue[i] < 0 || ue[i] > 63 || bs[i] < 0 || bs[i] > 63) continue;
        int c = beam_to_axis[ue[i]], r = beam_to_axis[bs[i]];
        if (c < 0 || r < 0) continue;
        sum[(int64_t)r * n_axis + c] += rss[i];
        cnt[(int64_t)r * n_axis + c] += 1;
    }
}
