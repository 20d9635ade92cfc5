// General hex tokenizer: arbitrary text -> bytes, with the reference's exact token rules.
//
// Replaces _read_bytes (serial_hex_to_excel_v3.py:18-36): read_text(encoding="utf-8",
// errors="ignore") -> str.split() -> keep tokens matching (?:0x)?[0-9a-fA-F]{2} (:16).
//
// What makes this parallel: with errors="ignore" the UTF-8 decoder restarts after every error span,
// and an error span or a multi-byte character only ever swallows continuation bytes (0x80..0xBF).
// So every byte's fate is decided by itself and at most the three bytes after it: continuation
// bytes, 0xC0/0xC1 and 0xF5..0xFF always vanish; a lead byte is one character when its sequence is
// complete and well-formed, else it vanishes; ASCII is itself.  Characters are compacted into a
// class stream (hex value / 'x' / other / space), then a token is judged from a 5-character window
// at each token start, and the kept bytes are compacted again.  Both compactions are
// count -> scan -> scatter over 4096-element tiles.

#include "pipeline.hpp"

namespace {

constexpr int TOK_TILE = 4096;          // elements per workgroup (256 threads x 16)
constexpr uint8_t C_X = 16, C_OTHER = 17, C_SPACE = 18, C_DROP = 0xFF;

__device__ __forceinline__ bool is_cont(uint32_t b) { return (b & 0xC0u) == 0x80u; }

__device__ __forceinline__ bool cp_is_space(uint32_t c)   // Py_UNICODE_ISSPACE
{
    if (c < 0x80) return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20);
    return c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 ||
           c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

// class of the character that starts at byte i, or C_DROP when byte i produces no character
__device__ __forceinline__ uint8_t classify(const uint8_t *t, size_t i, size_t n)
{
    const uint32_t ch = t[i];
    if (ch < 0x80) {
        if (cp_is_space(ch)) return C_SPACE;
        if (ch - '0' < 10u) return (uint8_t)(ch - '0');
        if (ch - 'a' < 6u) return (uint8_t)(ch - 'a' + 10);
        if (ch - 'A' < 6u) return (uint8_t)(ch - 'A' + 10);
        if (ch == 'x') return C_X;
        return C_OTHER;
    }
    if (ch < 0xC2 || ch > 0xF4) return C_DROP;
    const size_t left = n - i;
    uint32_t cp;
    if (ch < 0xE0) {
        if (left < 2 || !is_cont(t[i + 1])) return C_DROP;
        cp = ((ch & 0x1Fu) << 6) | (t[i + 1] & 0x3Fu);
    } else if (ch < 0xF0) {
        if (left < 3) return C_DROP;
        const uint32_t c2 = t[i + 1], c3 = t[i + 2];
        if (!is_cont(c2) || (c2 < 0xA0 ? ch == 0xE0 : ch == 0xED) || !is_cont(c3)) return C_DROP;
        cp = ((ch & 0x0Fu) << 12) | ((c2 & 0x3Fu) << 6) | (c3 & 0x3Fu);
    } else {
        if (left < 4) return C_DROP;
        const uint32_t c2 = t[i + 1], c3 = t[i + 2], c4 = t[i + 3];
        if (!is_cont(c2) || (c2 < 0x90 ? ch == 0xF0 : ch == 0xF4) || !is_cont(c3) || !is_cont(c4)) return C_DROP;
        cp = 0x10000;   // any supplementary-plane character: never whitespace, never hex
    }
    return cp_is_space(cp) ? C_SPACE : C_OTHER;
}

// ---- pass A/B: characters -> compact class stream -----------------------------------------------------
__global__ __launch_bounds__(256) void k_tok_count_chars(const uint8_t *t, size_t n, uint32_t *tile_cnt)
{
    __shared__ uint32_t sh[4];
    const size_t base = (size_t)blockIdx.x * TOK_TILE + (size_t)threadIdx.x * 16;
    uint32_t c = 0;
    for (int k = 0; k < 16; k++)
        if (base + k < n && classify(t, base + k, n) != C_DROP) c++;
    c = wave_sum(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// single workgroup: exclusive scan of tile counts in place; total to *total
__global__ __launch_bounds__(1024) void k_tok_scan(uint32_t *cnt, uint32_t n_tiles, uint32_t *total)
{
    __shared__ uint32_t sh[17];
    uint32_t carry = 0;
    for (uint32_t seg = 0; seg < n_tiles; seg += 1024) {
        uint32_t i = seg + threadIdx.x;
        uint32_t v = i < n_tiles ? cnt[i] : 0, tot;
        uint32_t ex = block_excl_scan(v, sh, &tot);
        if (i < n_tiles) cnt[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// `pos` (optional): the byte offset each character starts at, for the diagnostics pass below
__global__ __launch_bounds__(256) void k_tok_write_chars(const uint8_t *t, size_t n, const uint32_t *tile_off,
                                                          uint8_t *cls, uint32_t *pos)
{
    __shared__ uint32_t sh[17];
    const size_t base = (size_t)blockIdx.x * TOK_TILE + (size_t)threadIdx.x * 16;
    uint8_t c[16];
    uint32_t cnt = 0;
    for (int k = 0; k < 16; k++) {
        c[k] = base + k < n ? classify(t, base + k, n) : C_DROP;
        cnt += c[k] != C_DROP;
    }
    uint32_t tot;
    uint32_t o = tile_off[blockIdx.x] + block_excl_scan(cnt, sh, &tot);
    for (int k = 0; k < 16; k++)
        if (c[k] != C_DROP) {
            if (pos) pos[o] = (uint32_t)(base + k);
            cls[o++] = c[k];
        }
}

// ---- pass C/D: token starts -> bytes ----------------------------------------------------------------------
// value of the token starting at character k, or -1 when k starts no kept token
__device__ __forceinline__ int token_at(const uint8_t *cls, uint32_t k, uint32_t m)
{
    const uint8_t c0 = cls[k];
    if (c0 == C_SPACE) return -1;
    if (k > 0 && cls[k - 1] != C_SPACE) return -1;
    const uint8_t c1 = k + 1 < m ? cls[k + 1] : C_SPACE;
    const uint8_t c2 = k + 2 < m ? cls[k + 2] : C_SPACE;
    if (c0 < 16 && c1 < 16 && c2 == C_SPACE) return (c0 << 4) | c1;
    if (c0 == 0 && c1 == C_X) {
        const uint8_t c3 = k + 3 < m ? cls[k + 3] : C_SPACE;
        const uint8_t c4 = k + 4 < m ? cls[k + 4] : C_SPACE;
        if (c2 < 16 && c3 < 16 && c4 == C_SPACE) return (c2 << 4) | c3;
    }
    return -1;
}

__global__ __launch_bounds__(256) void k_tok_count_bytes(const uint8_t *cls, const uint32_t *m_dev, uint32_t *tile_cnt)
{
    __shared__ uint32_t sh[4];
    const uint32_t m = *m_dev;
    const uint32_t base = blockIdx.x * TOK_TILE + threadIdx.x * 16;
    uint32_t c = 0;
    for (int k = 0; k < 16; k++)
        if (base + k < m && token_at(cls, base + k, m) >= 0) c++;
    c = wave_sum(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_tok_write_bytes(const uint8_t *cls, const uint32_t *m_dev,
                                                          const uint32_t *tile_off, uint8_t *out)
{
    __shared__ uint32_t sh[17];
    const uint32_t m = *m_dev;
    const uint32_t base = blockIdx.x * TOK_TILE + threadIdx.x * 16;
    int v[16];
    uint32_t cnt = 0;
    for (int k = 0; k < 16; k++) {
        v[k] = base + k < m ? token_at(cls, base + k, m) : -1;
        cnt += v[k] >= 0;
    }
    uint32_t tot;
    uint32_t o = tile_off[blockIdx.x] + block_excl_scan(cnt, sh, &tot);
    for (int k = 0; k < 16; k++)
        if (v[k] >= 0) out[o++] = (uint8_t)v[k];
}

// ---- diagnostics: the tokens _read_bytes skips (:26-28), with their index among all tokens ---------------
__device__ __forceinline__ bool token_starts(const uint8_t *cls, uint32_t k)
{
    return cls[k] != C_SPACE && (k == 0 || cls[k - 1] == C_SPACE);
}

__global__ __launch_bounds__(256) void k_tok_count_rejects(const uint8_t *cls, const uint32_t *m_dev, uint32_t *cnt_tok,
                                                            uint32_t *cnt_rej)
{
    __shared__ uint32_t sh[8];
    const uint32_t m = *m_dev;
    const uint32_t base = blockIdx.x * TOK_TILE + threadIdx.x * 16;
    uint32_t ct = 0, cr = 0;
    for (int k = 0; k < 16; k++)
        if (base + k < m && token_starts(cls, base + k)) {
            ct++;
            cr += token_at(cls, base + k, m) < 0;
        }
    ct = wave_sum(ct);
    cr = wave_sum(cr);
    if (lane_id() == 0) { sh[threadIdx.x >> 6] = ct; sh[4 + (threadIdx.x >> 6)] = cr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt_tok[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
        cnt_rej[blockIdx.x] = sh[4] + sh[5] + sh[6] + sh[7];
    }
}

// One entry per skipped token, in text order: its index among all tokens and the byte span [begin, end) from
// its first character to the whitespace character that ends it (bytes the decoder ignores included).
__global__ __launch_bounds__(256) void k_tok_write_rejects(const uint8_t *cls, const uint32_t *pos, const uint32_t *m_dev,
                                                            uint32_t n_text, const uint32_t *off_tok, const uint32_t *off_rej,
                                                            uint32_t cap, uint32_t *out_idx, uint32_t *out_begin,
                                                            uint32_t *out_end)
{
    __shared__ uint32_t sh_t[17], sh_r[17];
    const uint32_t m = *m_dev;
    const uint32_t base = blockIdx.x * TOK_TILE + threadIdx.x * 16;
    uint32_t starts = 0, rej = 0;
    for (int k = 0; k < 16; k++)
        if (base + k < m && token_starts(cls, base + k)) {
            starts |= 1u << k;
            if (token_at(cls, base + k, m) < 0) rej |= 1u << k;
        }
    uint32_t tot;
    uint32_t ti = off_tok[blockIdx.x] + block_excl_scan(__popc(starts), sh_t, &tot);
    uint32_t ri = off_rej[blockIdx.x] + block_excl_scan(__popc(rej), sh_r, &tot);
    for (int k = 0; k < 16; k++) {
        if (!((starts >> k) & 1u)) continue;
        if ((rej >> k) & 1u) {
            if (ri < cap) {
                uint32_t e = base + k + 1;
                while (e < m && cls[e] != C_SPACE) e++;      // bounded by the class stream's length
                out_idx[ri] = ti;
                out_begin[ri] = pos[base + k];
                out_end[ri] = e < m ? pos[e] : n_text;
            }
            ri++;
        }
        ti++;
    }
}

}  // namespace

size_t token_rejects_ws_bytes(size_t n_text)
{
    const size_t tiles = (n_text + TOK_TILE - 1) / TOK_TILE + 1;
    return ((n_text + 255) & ~size_t(255)) + ((n_text * 4 + 255) & ~size_t(255)) + 3 * ((tiles * 4 + 255) & ~size_t(255)) + 256;
}

// Skipped tokens of a text: (token index, byte begin, byte end) triples into the three device arrays (room for
// `cap` each), their number to n_dev[0] (it may exceed cap: then only the first cap were written).
int token_rejects(slam_ctx *ctx, const uint8_t *text_dev, size_t n_text, uint32_t *idx_dev, uint32_t *begin_dev,
                  uint32_t *end_dev, uint32_t cap, uint32_t *n_dev, uint8_t *ws)
{
    const uint32_t tiles = (uint32_t)((n_text + TOK_TILE - 1) / TOK_TILE);
    if (tiles == 0) {
        SLAM_HIP(ctx, hipMemsetAsync(n_dev, 0, 4, ctx->stream));
        return SLAM_OK;
    }
    const size_t cnt_bytes = ((size_t)(tiles + 1) * 4 + 255) & ~size_t(255);
    uint8_t *cls = ws;
    uint32_t *pos = (uint32_t *)(ws + ((n_text + 255) & ~size_t(255)));
    uint32_t *cnt_a = (uint32_t *)((uint8_t *)pos + ((n_text * 4 + 255) & ~size_t(255)));
    uint32_t *cnt_t = (uint32_t *)((uint8_t *)cnt_a + cnt_bytes);
    uint32_t *cnt_r = (uint32_t *)((uint8_t *)cnt_t + cnt_bytes);
    uint32_t *m_dev = (uint32_t *)((uint8_t *)cnt_r + cnt_bytes);          // m, then the token total
    SLAM_LAUNCH(ctx, "tok_count_chars", k_tok_count_chars, dim3(tiles), dim3(256), text_dev, n_text, cnt_a);
    SLAM_LAUNCH(ctx, "tok_scan", k_tok_scan, dim3(1), dim3(1024), cnt_a, tiles, m_dev);
    SLAM_LAUNCH(ctx, "tok_write_chars", k_tok_write_chars, dim3(tiles), dim3(256), text_dev, n_text, cnt_a, cls, pos);
    SLAM_LAUNCH(ctx, "tok_count_rejects", k_tok_count_rejects, dim3(tiles), dim3(256), cls, m_dev, cnt_t, cnt_r);
    SLAM_LAUNCH(ctx, "tok_scan", k_tok_scan, dim3(1), dim3(1024), cnt_t, tiles, m_dev + 1);
    SLAM_LAUNCH(ctx, "tok_scan", k_tok_scan, dim3(1), dim3(1024), cnt_r, tiles, n_dev);
    SLAM_LAUNCH(ctx, "tok_write_rejects", k_tok_write_rejects, dim3(tiles), dim3(256), cls, pos, m_dev, (uint32_t)n_text,
                cnt_t, cnt_r, cap, idx_dev, begin_dev, end_dev);
    return SLAM_OK;
}

size_t tokenize_ws_bytes(size_t n_text)
{
    const size_t tiles = (n_text + TOK_TILE - 1) / TOK_TILE + 1;
    return ((n_text + 255) & ~size_t(255)) + 2 * ((tiles * 4 + 255) & ~size_t(255)) + 256;
}

int tokenize_general(slam_ctx *ctx, const uint8_t *text_dev, size_t n_text, uint8_t *bytes_dev, uint32_t *n_out_dev,
                     uint8_t *ws)
{
    const uint32_t tiles = (uint32_t)((n_text + TOK_TILE - 1) / TOK_TILE);
    if (tiles == 0) {
        SLAM_HIP(ctx, hipMemsetAsync(n_out_dev, 0, 4, ctx->stream));
        return SLAM_OK;
    }
    uint8_t *cls = ws;
    uint32_t *cnt_a = (uint32_t *)(ws + ((n_text + 255) & ~size_t(255)));
    uint32_t *cnt_b = (uint32_t *)((uint8_t *)cnt_a + (((size_t)(tiles + 1) * 4 + 255) & ~size_t(255)));
    uint32_t *m_dev = (uint32_t *)((uint8_t *)cnt_b + (((size_t)(tiles + 1) * 4 + 255) & ~size_t(255)));
    SLAM_LAUNCH(ctx, "tok_count_chars", k_tok_count_chars, dim3(tiles), dim3(256), text_dev, n_text, cnt_a);
    SLAM_LAUNCH(ctx, "tok_scan", k_tok_scan, dim3(1), dim3(1024), cnt_a, tiles, m_dev);
    SLAM_LAUNCH(ctx, "tok_write_chars", k_tok_write_chars, dim3(tiles), dim3(256), text_dev, n_text, cnt_a, cls, (uint32_t *)nullptr);
    // the class stream has at most n_text elements: the same tile count covers it
    SLAM_LAUNCH(ctx, "tok_count_bytes", k_tok_count_bytes, dim3(tiles), dim3(256), cls, m_dev, cnt_b);
    SLAM_LAUNCH(ctx, "tok_scan", k_tok_scan, dim3(1), dim3(1024), cnt_b, tiles, n_out_dev);
    SLAM_LAUNCH(ctx, "tok_write_bytes", k_tok_write_bytes, dim3(tiles), dim3(256), cls, m_dev, cnt_b, bytes_dev);
    return SLAM_OK;
}
