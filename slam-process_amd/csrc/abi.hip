// C-ABI entry points of libslamhip.so (declared in include/slamhip.h) and the context plumbing.

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "pipeline.hpp"

// ------------------------------------------------------------------------------------------------------
// fixed device scratch of the allocation-free entry point (slam_process_frames_noalloc_dev)
// ------------------------------------------------------------------------------------------------------
// Every GPU the library runs on holds its own copy of this array, at its own address, until the process ends.
// `scratch` itself is only the host-side handle of the symbol: host code never passes it to a kernel or to a
// memory routine.  The device address is looked up once per device (resolve_scratch, from slam_ctx_create with that
// device current) and kept in scratch_ptr[device]; kernels receive that pointer as an argument, exactly as they
// receive the ctx workspace in slam_process_frames_dev.
static __device__ __align__(256) unsigned char scratch[SLAM_NOALLOC_SCRATCH_BYTES];
constexpr int SCRATCH_MAX_DEVICES = 64;
static unsigned char *scratch_ptr[SCRATCH_MAX_DEVICES];
constexpr int NOALLOC_KMAX = 21;                          // blur taps per side the by-value kernel upload can carry
constexpr size_t NOALLOC_KERN_BYTES = 4096;               // front of the scratch: the Gaussian weights (21 * 21 doubles)
static_assert(NOALLOC_KMAX * NOALLOC_KMAX * sizeof(double) <= NOALLOC_KERN_BYTES, "kernel weights fit their piece");

// Calls on the fixed scratch are ordered per device: each waits, on its own stream, for the event the call before it
// recorded, and the host side of a call (wait, launches, record) runs under the device's mutex, so the record a call
// waits for is always that of the call issued just before it, whatever ctx, stream or thread issued it.
static hipEvent_t scratch_done[SCRATCH_MAX_DEVICES];
static std::mutex scratch_mutex[SCRATCH_MAX_DEVICES];
static std::mutex scratch_setup_mutex;

static hipError_t resolve_scratch(int device)              // `device` must be the current device
{
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(scratch_setup_mutex);
    if (scratch_ptr[device]) return hipSuccess;
    unsigned char *p = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    e = hipEventCreateWithFlags(&scratch_done[device], hipEventDisableTiming);
    if (e != hipSuccess) return e;
    scratch_ptr[device] = p;
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
int slam_fail(slam_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

extern "C" const char *slam_version(void) { return "slamhip 0.3 (gfx950)"; }
extern "C" int slam_abi_revision(void) { return SLAM_ABI_REVISION; }

// the library's default for slam_set_fused, unless the environment says otherwise
#ifndef SLAM_FUSED_DEFAULT
#define SLAM_FUSED_DEFAULT 0
#endif

extern "C" int slam_ctx_create(int device, slam_ctx **out)
{
    if (!out) return SLAM_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return SLAM_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return SLAM_E_HIP;
    // this device's copy of the fixed scratch array (devices beyond the table simply have no allocation-free path)
    if (device < SCRATCH_MAX_DEVICES && resolve_scratch(device) != hipSuccess) return SLAM_E_HIP;
    slam_ctx *c = new slam_ctx();
    c->device = device;
    const char *fe = getenv("SLAMHIP_FUSED"), *tp = getenv("SLAMHIP_FORCE_TWO_PASS");
    c->fused = fe && (fe[0] == '0' || fe[0] == '1') ? fe[0] - '0' : SLAM_FUSED_DEFAULT;
    c->force_two_pass = tp && tp[0] == '1';
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SLAM_E_HIP; }
    c->stream = c->own_stream;
    if (hipHostMalloc((void **)&c->pinned, 4096, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return SLAM_E_HIP;
    }
    *out = c;
    return SLAM_OK;
}

extern "C" int slam_ctx_destroy(slam_ctx *ctx)
{
    if (!ctx) return SLAM_E_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();          // not the ctx's stream: a caller's stream handle may be gone by now
    for (auto &p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->kern_dev) (void)hipFree(ctx->kern_dev);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SLAM_OK;
}

extern "C" int slam_set_stream(slam_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SLAM_E_INVALID;
    ctx->stream = (hipStream_t)hip_stream;          // NULL is HIP's null stream (e.g. torch's default stream)
    return SLAM_OK;
}

extern "C" int slam_use_own_stream(slam_ctx *ctx)
{
    if (!ctx) return SLAM_E_INVALID;
    ctx->stream = ctx->own_stream;
    return SLAM_OK;
}

extern "C" int slam_set_fused(slam_ctx *ctx, int mode)
{
    if (!ctx) return SLAM_E_INVALID;
    if (mode != 0 && mode != 1) return slam_fail(ctx, SLAM_E_INVALID, "slam_set_fused: mode must be 0 or 1");
    ctx->fused = mode;
    return SLAM_OK;
}

extern "C" int slam_get_fused(slam_ctx *ctx) { return ctx ? ctx->fused : SLAM_E_INVALID; }

extern "C" int slam_sync(slam_ctx *ctx)
{
    if (!ctx) return SLAM_E_INVALID;
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

extern "C" const char *slam_last_error(slam_ctx *ctx) { return ctx ? ctx->err : "null context"; }

// true while the ctx's stream is being captured into a graph: nothing may synchronise, allocate or free then, and
// addresses baked into captured kernel nodes must stay valid for the graph's lifetime
static bool stream_is_capturing(slam_ctx *ctx)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

int slam_ws_reserve(slam_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ws_size) return SLAM_OK;
    if (ctx->ws_pinned)
        return slam_fail(ctx, SLAM_E_CAPACITY,
                         "this call needs %zu bytes of workspace but the ctx's workspace (%zu bytes) is pinned by "
                         "slam_pin_workspace: growing it would invalidate graphs captured on it (use another ctx, or unpin)",
                         bytes, ctx->ws_size);
    if (stream_is_capturing(ctx))
        return slam_fail(ctx, SLAM_E_INVALID,
                         "the workspace would have to grow to %zu bytes while the stream is being captured: call "
                         "slam_reserve_workspace before the capture", bytes);
    SLAM_HIP(ctx, hipSetDevice(ctx->device));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) SLAM_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_size = 0;
    size_t want = bytes + bytes / 8 + (1 << 20);
    hipError_t e = hipMalloc((void **)&ctx->ws, want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc((void **)&ctx->ws, want);
    }
    if (e != hipSuccess) return slam_fail(ctx, SLAM_E_HIP, "hipMalloc of %zu workspace bytes failed: %s", want,
                                          hipGetErrorString(e));
    ctx->ws_size = want;
    return SLAM_OK;
}

void slam_ws_reset(slam_ctx *ctx) { ctx->ws_used = 0; }

void *slam_ws_take(slam_ctx *ctx, size_t bytes)
{
    size_t need = (bytes + 255) & ~size_t(255);
    if (ctx->ws_used + need > ctx->ws_size) return nullptr;
    void *p = ctx->ws + ctx->ws_used;
    ctx->ws_used += need;
    return p;
}

// Gaussian kernel cached on the device per ctx; re-uploaded only when sigma changes.
static int get_kernel(slam_ctx *ctx, double sigma, double **kern, int *ks)
{
    constexpr int KMAX = 127;
    if (!ctx->kern_dev) SLAM_HIP(ctx, hipMalloc((void **)&ctx->kern_dev, KMAX * KMAX * sizeof(double)));
    if (ctx->kern_size == 0 || ctx->kern_sigma != sigma) {
        // the size is checked before the kernel is built: size * size doubles for an arbitrary sigma is unbounded
        if (sigma != sigma) return slam_fail(ctx, SLAM_E_INVALID, "blur sigma is NaN");
        if (sigma > 0 && ceil(6.0 * sigma) > (double)KMAX)
            return slam_fail(ctx, SLAM_E_RANGE, "blur sigma %.3f needs a %.0f-tap kernel (max %d)", sigma, ceil(6.0 * sigma), KMAX);
        std::vector<double> k;
        int size = slam_gaussian_kernel_host(sigma, k);
        if (size > KMAX) return slam_fail(ctx, SLAM_E_RANGE, "blur sigma %.3f needs a %d-tap kernel (max %d)", sigma, size, KMAX);
        if (stream_is_capturing(ctx) || ctx->ws_pinned)      // the upload synchronises and rewrites what captured nodes read
            return slam_fail(ctx, SLAM_E_INVALID,
                             "blur sigma %.3f differs from the one reserved (%.3f): not while the stream is captured or the "
                             "workspace is pinned (slam_reserve_workspace with this batch first)", sigma, ctx->kern_sigma);
        SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SLAM_HIP(ctx, hipMemcpy(ctx->kern_dev, k.data(), k.size() * sizeof(double), hipMemcpyHostToDevice));
        ctx->kern_sigma = sigma;
        ctx->kern_size = size;
    }
    *kern = ctx->kern_dev;
    *ks = ctx->kern_size;
    return SLAM_OK;
}

// ---- timing ----------------------------------------------------------------------------------------------
int slam_timer_begin(slam_ctx *ctx, const char *name)
{
    if (!ctx->timing_only.empty() && ctx->timing_only.find("," + std::string(name) + ",") == std::string::npos) return -1;
    int slot = -1;
    for (size_t i = 0; i < ctx->timers.size(); i++)
        if (!strncmp(ctx->timers[i].name, name, 31)) { slot = (int)i; break; }
    if (slot < 0) {
        slam_timer_slot s;
        memset(&s, 0, sizeof(s));
        strncpy(s.name, name, 31);
        ctx->timers.push_back(s);
        slot = (int)ctx->timers.size() - 1;
    }
    slam_ctx::pending_ev p;
    p.slot = slot;
    p.a = p.b = nullptr;
    for (hipEvent_t *e : {&p.a, &p.b}) {
        if (!ctx->ev_pool.empty()) { *e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
        else if (hipEventCreate(e) != hipSuccess) {
            if (p.a) ctx->ev_pool.push_back(p.a);          // the launch goes untimed; nothing is lost
            return -1;
        }
    }
    (void)hipEventRecord(p.a, ctx->stream);
    ctx->pending.push_back(p);
    return (int)ctx->pending.size() - 1;
}

void slam_timer_end(slam_ctx *ctx, int token) { (void)hipEventRecord(ctx->pending[token].b, ctx->stream); }

static void timers_collect(slam_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->timers[p.slot].launches++;
            ctx->timers[p.slot].ms += ms;
        }
        ctx->ev_pool.push_back(p.a);
        ctx->ev_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

extern "C" int slam_timing_enable(slam_ctx *ctx, int on)
{
    if (!ctx) return SLAM_E_INVALID;
    if (!on) timers_collect(ctx);
    ctx->timing = on != 0;
    return SLAM_OK;
}

extern "C" int slam_timing_select(slam_ctx *ctx, const char *names)
{
    if (!ctx) return SLAM_E_INVALID;
    ctx->timing_only = names && names[0] ? "," + std::string(names) + "," : std::string();
    return SLAM_OK;
}

extern "C" int slam_timing_reset(slam_ctx *ctx)
{
    if (!ctx) return SLAM_E_INVALID;
    timers_collect(ctx);
    ctx->timers.clear();
    return SLAM_OK;
}

extern "C" int slam_timing_read(slam_ctx *ctx, char *names_out, uint32_t *launches_out, float *ms_out, int cap)
{
    if (!ctx) return SLAM_E_INVALID;
    timers_collect(ctx);
    int n = (int)std::min<size_t>(ctx->timers.size(), (size_t)std::max(cap, 0));
    for (int i = 0; i < n; i++) {
        if (names_out) memcpy(names_out + 32 * i, ctx->timers[i].name, 32);
        if (launches_out) launches_out[i] = ctx->timers[i].launches;
        if (ms_out) ms_out[i] = (float)ctx->timers[i].ms;
    }
    return (int)ctx->timers.size();
}

// ------------------------------------------------------------------------------------------------------
// helpers for the host-pointer entry points
// ------------------------------------------------------------------------------------------------------
#define REQUIRE_CTX(ctx)                                                                       \
    do {                                                                                       \
        if (!(ctx)) return SLAM_E_INVALID;                                                     \
        if (hipSetDevice((ctx)->device) != hipSuccess) return slam_fail((ctx), SLAM_E_HIP, "hipSetDevice failed"); \
    } while (0)

static uint32_t round16(uint64_t v) { return (uint32_t)((v + 15) & ~uint64_t(15)); }

namespace {
__global__ __launch_bounds__(256) void k_zero_words(uint32_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

int slam_zero_async(slam_ctx *ctx, void *dev, size_t bytes)
{
    const size_t n = bytes / 4;
    if (n == 0) return SLAM_OK;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 4096);
    SLAM_LAUNCH(ctx, "zero", k_zero_words, dim3(blocks), dim3(256), (uint32_t *)dev, n);
    return SLAM_OK;
}

static int h2d(slam_ctx *ctx, void *dst, const void *src, size_t n)
{
    if (n) SLAM_HIP(ctx, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, ctx->stream));
    return SLAM_OK;
}
static int d2h(slam_ctx *ctx, void *dst, const void *src, size_t n)
{
    if (n) SLAM_HIP(ctx, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, ctx->stream));
    return SLAM_OK;
}

// one frame on the device, decoded through the TEXT3 or RAW source; rows land in ws-carved arrays
struct one_frame {
    uint8_t *src;           // device copy of the input
    uint64_t *off;
    uint32_t *len;
    uint32_t *stats;
    rows_out rows;
    frame_ws fws;
    uint32_t nt_max;
};

static size_t one_frame_plan(ws_plan &p, size_t n_src, uint32_t row_cap, uint32_t nt_max, size_t off[16])
{
    off[0] = p.add(n_src + 32);
    off[1] = p.add(8);
    off[2] = p.add(4);
    off[3] = p.add(SLAM_STAT_WORDS * 4);
    off[4] = p.add(row_cap);
    off[5] = p.add(row_cap);
    off[6] = p.add(row_cap);
    off[7] = p.add((size_t)row_cap * 4);
    off[8] = p.add((size_t)row_cap * 4);
    frame_ws::plan(p, 1, nt_max, off + 9);
    return p.total;
}

static void one_frame_bind(one_frame &f, uint8_t *b, const size_t off[16], uint32_t row_cap, uint32_t nt_max)
{
    f.src = b + off[0];
    f.off = (uint64_t *)(b + off[1]);
    f.len = (uint32_t *)(b + off[2]);
    f.stats = (uint32_t *)(b + off[3]);
    f.rows.flag = b + off[4];
    f.rows.ue = b + off[5];
    f.rows.bs = b + off[6];
    f.rows.rss = (uint32_t *)(b + off[7]);
    f.rows.clk = (uint32_t *)(b + off[8]);
    f.rows.row_cap = row_cap;
    f.fws.bind(b, off + 9, nt_max);
    f.nt_max = nt_max;
}

static int set_one_frame_desc(slam_ctx *ctx, one_frame &f, uint32_t len)
{
    uint64_t *po = (uint64_t *)ctx->pinned;
    uint32_t *pl = ctx->pinned + 4;
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));         // pinned words are reused between calls
    *po = 0;
    *pl = len;
    SLAM_TRY(h2d(ctx, f.off, po, 8));
    SLAM_TRY(h2d(ctx, f.len, pl, 4));
    SLAM_HIP(ctx, hipMemsetAsync(f.stats, 0, SLAM_STAT_WORDS * 4, ctx->stream));
    return SLAM_OK;
}

static int read_stats(slam_ctx *ctx, const uint32_t *stats_dev, uint32_t out[SLAM_STAT_WORDS])
{
    SLAM_TRY(d2h(ctx, ctx->pinned + 16, stats_dev, SLAM_STAT_WORDS * 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->pinned + 16, SLAM_STAT_WORDS * 4);
    return SLAM_OK;
}

static int download_rows(slam_ctx *ctx, const rows_out &r, size_t n, uint8_t *flag, uint8_t *ue, uint8_t *bs,
                         uint32_t *rss, uint32_t *clk)
{
    if (flag) SLAM_TRY(d2h(ctx, flag, r.flag, n));
    if (ue) SLAM_TRY(d2h(ctx, ue, r.ue, n));
    if (bs) SLAM_TRY(d2h(ctx, bs, r.bs, n));
    if (rss) SLAM_TRY(d2h(ctx, rss, r.rss, n * 4));
    if (clk) SLAM_TRY(d2h(ctx, clk, r.clk, n * 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// a1: hex text -> bytes
// ------------------------------------------------------------------------------------------------------
extern "C" int slam_hex_unpack(slam_ctx *ctx, const uint8_t *text, size_t n_text, uint8_t *bytes_out, size_t cap,
                               size_t *n_bytes)
{
    REQUIRE_CTX(ctx);
    if ((!text && n_text) || !n_bytes) return slam_fail(ctx, SLAM_E_INVALID, "slam_hex_unpack: null argument");
    if (n_text > 0xFFFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "text larger than 4 GiB");
    *n_bytes = 0;
    if (n_text == 0) return SLAM_OK;
    ws_plan p;
    size_t o_text = p.add(n_text + 32), o_out = p.add(n_text / 2 + 32), o_cnt = p.add(4),
           o_tok = p.add(tokenize_ws_bytes(n_text));
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    SLAM_TRY(h2d(ctx, b + o_text, text, n_text));
    SLAM_TRY(tokenize_general(ctx, b + o_text, n_text, b + o_out, (uint32_t *)(b + o_cnt), b + o_tok));
    SLAM_TRY(d2h(ctx, ctx->pinned, b + o_cnt, 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t n = ctx->pinned[0];
    *n_bytes = n;
    if (n > cap || (n && !bytes_out))
        return slam_fail(ctx, SLAM_E_CAPACITY, "slam_hex_unpack: need room for %zu bytes, have %zu", n, cap);
    SLAM_TRY(d2h(ctx, bytes_out, b + o_out, n));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_token_rejects(slam_ctx *ctx, const uint8_t *text, size_t n_text, uint32_t *index_out,
                                  uint32_t *begin_out, uint32_t *end_out, size_t cap, size_t *n_rejects)
{
    REQUIRE_CTX(ctx);
    if ((!text && n_text) || !n_rejects) return slam_fail(ctx, SLAM_E_INVALID, "slam_token_rejects: null argument");
    if (n_text > 0xFFFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "text larger than 4 GiB");
    *n_rejects = 0;
    if (n_text == 0) return SLAM_OK;
    const uint32_t cap_dev = (uint32_t)std::min<size_t>(cap, n_text);      // a token takes at least one byte
    ws_plan p;
    size_t o_text = p.add(n_text + 32), o_cnt = p.add(4), o_out[3], o_tok = p.add(token_rejects_ws_bytes(n_text));
    for (int i = 0; i < 3; i++) o_out[i] = p.add((size_t)cap_dev * 4 + 4);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    SLAM_TRY(h2d(ctx, b + o_text, text, n_text));
    SLAM_TRY(token_rejects(ctx, b + o_text, n_text, (uint32_t *)(b + o_out[0]), (uint32_t *)(b + o_out[1]),
                           (uint32_t *)(b + o_out[2]), cap_dev, (uint32_t *)(b + o_cnt), b + o_tok));
    SLAM_TRY(d2h(ctx, ctx->pinned, b + o_cnt, 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n = ctx->pinned[0];
    *n_rejects = n;
    if (n > cap || (n && (!index_out || !begin_out || !end_out)))
        return slam_fail(ctx, SLAM_E_CAPACITY, "slam_token_rejects: need room for %zu tokens, have %zu", n, cap);
    uint32_t *outs[3] = {index_out, begin_out, end_out};
    for (int i = 0; i < 3; i++) SLAM_TRY(d2h(ctx, outs[i], b + o_out[i], n * 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// a2: bytes -> records, and a1+a2 fused
// ------------------------------------------------------------------------------------------------------
static int parse_common(slam_ctx *ctx, int mode, const uint8_t *src, size_t n_src, uint8_t *flag, uint8_t *ue,
                        uint8_t *bs, uint32_t *rss, uint32_t *clk, size_t cap, size_t *n_valid, size_t *n_disc)
{
    if ((!src && n_src) || !n_valid || !n_disc) return slam_fail(ctx, SLAM_E_INVALID, "parse: null argument");
    if (n_src > 0xFFFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "input larger than 4 GiB");
    *n_valid = 0;
    *n_disc = 0;
    if (n_src == 0) return SLAM_OK;
    const uint64_t nb_max = mode == SRC_TEXT3 ? (n_src + 1) / 3 : n_src;
    const uint32_t row_cap = round16(nb_max / 11 + 16);
    const uint32_t nt_max = frame_tiles_for(mode, (uint32_t)n_src);
    ws_plan p;
    size_t off[16];
    one_frame_plan(p, n_src, row_cap, nt_max, off);
    // room for the general tokenizer's output and scratch when the text is not in the strict format
    size_t o_gen = 0, o_gcnt = 0, o_gtok = 0;
    if (mode == SRC_TEXT3) {
        o_gen = p.add(n_src / 2 + 32);
        o_gcnt = p.add(4);
        o_gtok = p.add(tokenize_ws_bytes(n_src));
    }
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    one_frame fr;
    one_frame_bind(fr, ctx->ws, off, row_cap, nt_max);
    SLAM_TRY(h2d(ctx, fr.src, src, n_src));
    SLAM_TRY(set_one_frame_desc(ctx, fr, (uint32_t)n_src));
    frame_src fs{fr.src, fr.off, fr.len, nullptr, 0, nullptr};
    SLAM_TRY(frame_decode(ctx, mode, fs, 1, (uint32_t)n_src, fr.fws, nt_max, fr.rows, fr.stats));
    uint32_t st[SLAM_STAT_WORDS];
    SLAM_TRY(read_stats(ctx, fr.stats, st));
    if (mode == SRC_TEXT3 && st[SLAM_STAT_NOT_STRICT]) {
        // free-form text: exact tokenizer, then the same framing on the byte stream
        uint8_t *gen = ctx->ws + o_gen;
        uint32_t *gcnt = (uint32_t *)(ctx->ws + o_gcnt);
        SLAM_TRY(tokenize_general(ctx, fr.src, n_src, gen, gcnt, ctx->ws + o_gtok));
        SLAM_TRY(d2h(ctx, ctx->pinned, gcnt, 4));
        SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t nb = ctx->pinned[0];
        SLAM_TRY(set_one_frame_desc(ctx, fr, nb));
        frame_src gs{gen, fr.off, fr.len, nullptr, 0, nullptr};
        // the strict estimate (n/3 bytes) bounds the general count (n/2 at most -> tiles and rows still fit?)
        const uint32_t nt_gen = frame_tiles_for(SRC_RAW, nb);
        if (nt_gen > nt_max || (uint64_t)nb / 11 + 16 > row_cap)
            return slam_fail(ctx, SLAM_E_CAPACITY, "free-form text decoded to more bytes than planned");
        SLAM_TRY(frame_decode(ctx, SRC_RAW, gs, 1, nb, fr.fws, nt_max, fr.rows, fr.stats));
        SLAM_TRY(read_stats(ctx, fr.stats, st));
    }
    *n_valid = st[SLAM_STAT_VALID];
    *n_disc = st[SLAM_STAT_DISCARDED];
    if (*n_valid > cap)
        return slam_fail(ctx, SLAM_E_CAPACITY, "parse: need room for %zu records, have %zu", *n_valid, cap);
    return download_rows(ctx, fr.rows, *n_valid, flag, ue, bs, rss, clk);
}

extern "C" int slam_parse_records(slam_ctx *ctx, const uint8_t *bytes, size_t n, uint8_t *flag, uint8_t *ue,
                                  uint8_t *bs, uint32_t *rss, uint32_t *clk, size_t cap, size_t *n_valid,
                                  size_t *n_discarded)
{
    REQUIRE_CTX(ctx);
    return parse_common(ctx, SRC_RAW, bytes, n, flag, ue, bs, rss, clk, cap, n_valid, n_discarded);
}

extern "C" int slam_parse_rejects(slam_ctx *ctx, const uint8_t *bytes, size_t n, uint32_t *pos_out, size_t cap,
                                  size_t *n_rejects)
{
    REQUIRE_CTX(ctx);
    if ((!bytes && n) || !n_rejects) return slam_fail(ctx, SLAM_E_INVALID, "slam_parse_rejects: null argument");
    if (n > 0xFFFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "input larger than 4 GiB");
    *n_rejects = 0;
    if (n == 0) return SLAM_OK;
    const uint32_t row_cap = round16(n / 11 + 16);
    const uint32_t nt_max = frame_tiles_for(SRC_RAW, (uint32_t)n);
    ws_plan p;
    size_t off[16];
    one_frame_plan(p, n, row_cap, nt_max, off);
    const size_t o_pos = p.add((n + 16) * sizeof(uint32_t));          // every byte could be a discarded flag byte
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    one_frame fr;
    one_frame_bind(fr, ctx->ws, off, row_cap, nt_max);
    SLAM_TRY(h2d(ctx, fr.src, bytes, n));
    SLAM_TRY(set_one_frame_desc(ctx, fr, (uint32_t)n));
    frame_src fs{fr.src, fr.off, fr.len, nullptr, 0, nullptr};
    SLAM_TRY(frame_decode(ctx, SRC_RAW, fs, 1, (uint32_t)n, fr.fws, nt_max, fr.rows, fr.stats));
    uint32_t *pos_dev = (uint32_t *)(ctx->ws + o_pos);
    SLAM_TRY(frame_rejects(ctx, SRC_RAW, fs, 1, fr.fws, nt_max, pos_dev, (uint32_t)(n + 16)));
    uint32_t st[SLAM_STAT_WORDS];
    SLAM_TRY(read_stats(ctx, fr.stats, st));
    *n_rejects = st[SLAM_STAT_DISCARDED];
    if (*n_rejects > cap || (*n_rejects && !pos_out))
        return slam_fail(ctx, SLAM_E_CAPACITY, "slam_parse_rejects: need room for %zu positions, have %zu", *n_rejects, cap);
    SLAM_TRY(d2h(ctx, pos_out, pos_dev, *n_rejects * sizeof(uint32_t)));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_parse_text(slam_ctx *ctx, const uint8_t *text, size_t n_text, uint8_t *flag, uint8_t *ue,
                               uint8_t *bs, uint32_t *rss, uint32_t *clk, size_t cap, size_t *n_valid,
                               size_t *n_discarded)
{
    REQUIRE_CTX(ctx);
    return parse_common(ctx, SRC_TEXT3, text, n_text, flag, ue, bs, rss, clk, cap, n_valid, n_discarded);
}

// ------------------------------------------------------------------------------------------------------
// a4-a7: correction on int32 columns
// ------------------------------------------------------------------------------------------------------
extern "C" int slam_correct(slam_ctx *ctx, const int32_t *flag, const int32_t *ue, const int32_t *bs,
                            const int32_t *rss, const int32_t *clk, size_t n, int32_t cycle, int32_t tol,
                            int32_t mod_base, uint32_t flags, const int32_t *bl_in_clk, const int32_t *bl_in_bs, int64_t bl_in_n,
                            int32_t *out_ue, int32_t *out_bs, int32_t *out_rss, int32_t *out_clk, size_t *n_kept,
                            int32_t *gid_out, int32_t *corrected, int32_t *bl_clk, int32_t *bl_bs, int32_t *bl_gid,
                            size_t *n_baselines, size_t *n_groups)
{
    REQUIRE_CTX(ctx);
    if (n_kept) *n_kept = 0;
    if (n_baselines) *n_baselines = 0;
    if (n_groups) *n_groups = 0;
    if (n == 0) return SLAM_OK;
    if (!flag || !ue || !bs || !rss || !clk) return slam_fail(ctx, SLAM_E_INVALID, "slam_correct: null column");
    if (n > 0x7FFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "too many rows");
    if (cycle <= 0 || cycle > (1 << 30) || mod_base <= 0)
        return slam_fail(ctx, SLAM_E_INVALID, "cycle must be in 1..2^30 and mod_base positive");
    const bool explicit_bl = bl_in_n >= 0;
    if (explicit_bl && bl_in_n > 0 && (!bl_in_clk || !bl_in_bs))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_correct: null baseline arrays");
    if (bl_in_n > 0x7FFFFF00ll) return slam_fail(ctx, SLAM_E_RANGE, "too many baselines");
    for (size_t i = 0; i < n; i++)
        if (clk[i] <= -(1 << 30) || clk[i] >= (1 << 30))
            return slam_fail(ctx, SLAM_E_RANGE, "clk[%zu] = %d is outside (-2^30, 2^30)", i, clk[i]);
    for (int64_t i = 0; explicit_bl && i < bl_in_n; i++)
        if (bl_in_clk[i] <= -(1 << 30) || bl_in_clk[i] >= (1 << 30))
            return slam_fail(ctx, SLAM_E_RANGE, "baseline clk %d is outside (-2^30, 2^30)", bl_in_clk[i]);

    const uint32_t row_cap = round16(n + 16);
    const uint32_t bl_cap = std::max<uint32_t>(correct_ws::default_bl_cap(row_cap),
                                               explicit_bl ? (uint32_t)bl_in_n + 16 : 0);
    ws_plan p;
    size_t oc[5], ok[4], ostat = p.add(SLAM_STAT_WORDS * 4), ogid = p.add((size_t)row_cap * 4), cw[16];
    for (int i = 0; i < 5; i++) oc[i] = p.add((size_t)row_cap * 4);
    for (int i = 0; i < 4; i++) ok[i] = p.add((size_t)row_cap * 4);
    correct_ws::plan(p, 1, row_cap, bl_cap, cw);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    correct_ws ws;
    ws.bind(b, cw, bl_cap);
    uint32_t *stats = (uint32_t *)(b + ostat);
    const int32_t *cols[5] = {flag, ue, bs, rss, clk};
    for (int i = 0; i < 5; i++) SLAM_TRY(h2d(ctx, b + oc[i], cols[i], n * 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memset(ctx->pinned, 0, SLAM_STAT_WORDS * 4);
    ctx->pinned[SLAM_STAT_VALID] = (uint32_t)n;
    ctx->pinned[SLAM_STAT_ROW_SLOTS] = (uint32_t)n;
    if (explicit_bl) ctx->pinned[SLAM_STAT_BASELINES] = (uint32_t)bl_in_n;
    SLAM_TRY(h2d(ctx, stats, ctx->pinned, SLAM_STAT_WORDS * 4));
    if (explicit_bl && bl_in_n) {
        SLAM_TRY(h2d(ctx, ws.bl_clk, bl_in_clk, (size_t)bl_in_n * 4));
        SLAM_TRY(h2d(ctx, ws.bl_bs, bl_in_bs, (size_t)bl_in_n * 4));
    }
    rows_in_t<int32_t, int32_t> in{(int32_t *)(b + oc[0]), (int32_t *)(b + oc[1]), (int32_t *)(b + oc[2]),
                                   (int32_t *)(b + oc[3]), (int32_t *)(b + oc[4]), row_cap};
    kept_out_t<int32_t, int32_t> out{(int32_t *)(b + ok[0]), (int32_t *)(b + ok[1]), (int32_t *)(b + ok[2]),
                                     (int32_t *)(b + ok[3])};
    correct_params prm;
    prm.set_cycle(cycle);
    prm.tol = tol;
    prm.mod_base = mod_base;
    prm.one_group = (flags & SLAM_CORRECT_ONE_GROUP) ? 1 : 0;
    prm.explicit_baselines = explicit_bl ? 1 : 0;
    SLAM_TRY((correct_rows<int32_t, int32_t>(ctx, in, 1, ws, prm, out, gid_out ? (int32_t *)(b + ogid) : nullptr,
                                             stats)));
    uint32_t st[SLAM_STAT_WORDS];
    SLAM_TRY(read_stats(ctx, stats, st));
    const size_t k = st[SLAM_STAT_KEPT], nb = st[SLAM_STAT_BASELINES];
    if (n_kept) *n_kept = k;
    if (n_baselines) *n_baselines = nb;
    if (n_groups) *n_groups = st[SLAM_STAT_GROUPS];
    int32_t *outs[4] = {out_ue, out_bs, out_rss, out_clk};
    for (int i = 0; i < 4; i++)
        if (outs[i]) SLAM_TRY(d2h(ctx, outs[i], b + ok[i], k * 4));
    if (gid_out) SLAM_TRY(d2h(ctx, gid_out, b + ogid, n * 4));
    if (corrected) SLAM_TRY(d2h(ctx, corrected, ws.corr, n * 4));
    if (bl_clk) SLAM_TRY(d2h(ctx, bl_clk, ws.bl_clk, nb * 4));
    if (bl_bs) SLAM_TRY(d2h(ctx, bl_bs, ws.bl_bs, nb * 4));
    if (bl_gid) SLAM_TRY(d2h(ctx, bl_gid, ws.bl_gid, nb * 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// a9-a10: pivot
// ------------------------------------------------------------------------------------------------------
extern "C" int slam_pivot_mean(slam_ctx *ctx, const int32_t *ue, const int32_t *bs, const int32_t *rss, size_t n,
                               const int16_t *beam_to_axis, int n_axis, int64_t *sum, int64_t *cnt, double *matrix)
{
    REQUIRE_CTX(ctx);
    if (!beam_to_axis || n_axis < 1 || n_axis > SLAM_MAX_AXIS) return slam_fail(ctx, SLAM_E_INVALID, "slam_pivot_mean: bad axis table");
    if (n && (!ue || !bs || !rss)) return slam_fail(ctx, SLAM_E_INVALID, "slam_pivot_mean: null column");
    if (n > 0x7FFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "too many rows");
    for (int i = 0; i < 64; i++)
        if (beam_to_axis[i] >= n_axis)
            return slam_fail(ctx, SLAM_E_INVALID, "slam_pivot_mean: beam_to_axis[%d] = %d is not below n_axis = %d", i,
                             beam_to_axis[i], n_axis);
    const uint32_t row_cap = round16(n + 16);
    const int cells = n_axis * n_axis;
    ws_plan p;
    size_t oc[3], ostat = p.add(SLAM_STAT_WORDS * 4), ob2a = p.add(64 * sizeof(int16_t)), omat = p.add((size_t)cells * 8), rw[12];
    for (int i = 0; i < 3; i++) oc[i] = p.add((size_t)row_cap * 4);
    render_ws::plan(p, 1, 1, 1, n_axis, rw);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    render_ws ws;
    ws.bind(b, rw, n_axis);
    uint32_t *stats = (uint32_t *)(b + ostat);
    const int32_t *cols[3] = {ue, bs, rss};
    for (int i = 0; i < 3; i++) SLAM_TRY(h2d(ctx, b + oc[i], cols[i], n * 4));
    SLAM_TRY(h2d(ctx, b + ob2a, beam_to_axis, 64 * sizeof(int16_t)));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memset(ctx->pinned, 0, SLAM_STAT_WORDS * 4);
    ctx->pinned[SLAM_STAT_KEPT] = (uint32_t)n;
    SLAM_TRY(h2d(ctx, stats, ctx->pinned, SLAM_STAT_WORDS * 4));
    render_params rp{};
    rp.beam_to_axis = (const int16_t *)(b + ob2a);
    rp.n_axis = n_axis;
    SLAM_TRY((pivot_accumulate<int32_t, int32_t>(ctx, (int32_t *)(b + oc[0]), (int32_t *)(b + oc[1]),
                                                 (int32_t *)(b + oc[2]), row_cap, 1, stats, SLAM_STAT_KEPT, rp, ws)));
    // a wide table's accumulators sit on the axis grid in sum_w / cnt_w (sum / cnt then hold the beam pairs)
    const unsigned long long *gsum = n_axis > SMALL_AXIS ? ws.sum_w : ws.sum;
    const uint32_t *gcnt = n_axis > SMALL_AXIS ? ws.cnt_w : ws.cnt;
    SLAM_TRY(mean_only(ctx, gsum, gcnt, cells, (double *)(b + omat)));
    std::vector<uint32_t> c32(cells);
    if (sum) SLAM_TRY(d2h(ctx, sum, gsum, (size_t)cells * 8));
    SLAM_TRY(d2h(ctx, c32.data(), gcnt, (size_t)cells * 4));
    if (matrix) SLAM_TRY(d2h(ctx, matrix, b + omat, (size_t)cells * 8));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cnt) for (int i = 0; i < cells; i++) cnt[i] = c32[i];
    return SLAM_OK;
}

extern "C" int slam_pivot_mean_f64(slam_ctx *ctx, const int32_t *ue, const int32_t *bs, const double *rss, size_t n,
                                   const int16_t *beam_to_axis, int n_axis, double *sum, int64_t *cnt, double *matrix)
{
    REQUIRE_CTX(ctx);
    if (!beam_to_axis || n_axis < 1 || n_axis > SLAM_MAX_AXIS) return slam_fail(ctx, SLAM_E_INVALID, "slam_pivot_mean_f64: bad axis table");
    if (n && (!ue || !bs || !rss)) return slam_fail(ctx, SLAM_E_INVALID, "slam_pivot_mean_f64: null column");
    if (n > 0x7FFFFF00ull) return slam_fail(ctx, SLAM_E_RANGE, "too many rows");
    for (int i = 0; i < 64; i++)
        if (beam_to_axis[i] >= n_axis)
            return slam_fail(ctx, SLAM_E_INVALID, "slam_pivot_mean_f64: beam_to_axis[%d] = %d is not below n_axis = %d", i,
                             beam_to_axis[i], n_axis);
    const size_t cells = (size_t)n_axis * n_axis;
    ws_plan p;
    const size_t ou = p.add(n * 4 + 4), ob = p.add(n * 4 + 4), orr = p.add(n * 8 + 8),
                 ob2a = p.add(64 * sizeof(int16_t)), os = p.add(cells * 8), ok = p.add(cells * 4), om = p.add(cells * 8);
    pivot_kahan_ws kw;
    kw.plan(p, (uint32_t)n, n_axis);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    SLAM_TRY(h2d(ctx, b + ou, ue, n * 4));
    SLAM_TRY(h2d(ctx, b + ob, bs, n * 4));
    SLAM_TRY(h2d(ctx, b + orr, rss, n * 8));
    SLAM_TRY(h2d(ctx, b + ob2a, beam_to_axis, 64 * sizeof(int16_t)));
    SLAM_TRY(pivot_kahan(ctx, (const int32_t *)(b + ou), (const int32_t *)(b + ob), (const double *)(b + orr), (uint32_t)n,
                         (const int16_t *)(b + ob2a), n_axis, b, kw, (double *)(b + os), (uint32_t *)(b + ok),
                         (double *)(b + om)));
    std::vector<uint32_t> c32(cells);
    if (sum) SLAM_TRY(d2h(ctx, sum, b + os, cells * 8));
    SLAM_TRY(d2h(ctx, c32.data(), b + ok, cells * 4));
    if (matrix) SLAM_TRY(d2h(ctx, matrix, b + om, cells * 8));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cnt) for (size_t i = 0; i < cells; i++) cnt[i] = c32[i];
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// a11-a12: blur
// ------------------------------------------------------------------------------------------------------
extern "C" int slam_blur_nan(slam_ctx *ctx, const double *in, int h, int w, double sigma, double *out)
{
    REQUIRE_CTX(ctx);
    if (!in || !out || h < 1 || w < 1 || h > 4096 || w > 4096) return slam_fail(ctx, SLAM_E_INVALID, "slam_blur_nan: bad shape");
    const size_t bytes = (size_t)h * w * 8;
    if (sigma <= 0) { memmove(out, in, bytes); return SLAM_OK; }       // returns the input unchanged (:340-341)
    ws_plan p;
    size_t oi = p.add(bytes), oo = p.add(bytes);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    double *kern;
    int ks;
    SLAM_TRY(get_kernel(ctx, sigma, &kern, &ks));
    SLAM_TRY(h2d(ctx, b + oi, in, bytes));
    SLAM_TRY(blur_only(ctx, (double *)(b + oi), h, w, kern, ks, (double *)(b + oo)));
    SLAM_TRY(d2h(ctx, out, b + oo, bytes));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// a13: colour scale + raster
// ------------------------------------------------------------------------------------------------------
extern "C" int slam_render(slam_ctx *ctx, const double *mat, int h, int w, const double *aod, const double *aoa,
                           int use_log, double vmin, double vmax, const uint8_t *lut256x3, uint8_t *cells_rgba,
                           uint8_t *tile_rgba, int W, int H, double *vmin_vmax_out)
{
    REQUIRE_CTX(ctx);
    if (!mat || !aod || !aoa || !lut256x3 || h < 1 || w < 1 || h > SLAM_MAX_AXIS || w > SLAM_MAX_AXIS)
        return slam_fail(ctx, SLAM_E_INVALID, "slam_render: matrix must be 1..%d x 1..%d with axis lists and a LUT",
                         SLAM_MAX_AXIS, SLAM_MAX_AXIS);
    if (tile_rgba && (W < 1 || H < 1 || W > 16384 || H > 16384)) return slam_fail(ctx, SLAM_E_INVALID, "slam_render: bad tile size");
    if (!tile_rgba) { W = 1; H = 1; }
    ws_plan p;
    size_t om = p.add((size_t)h * w * 8), oa = p.add(SLAM_MAX_AXIS * 8), ob = p.add(SLAM_MAX_AXIS * 8), ol = p.add(768),
           oc = p.add((size_t)h * w * 4),
           ocm = p.add((size_t)W * 2), orm = p.add((size_t)H * 2), ost = p.add(SLAM_STAT_WORDS * 4),
           ot = p.add((size_t)W * H * 4);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    SLAM_TRY(h2d(ctx, b + om, mat, (size_t)h * w * 8));
    SLAM_TRY(h2d(ctx, b + oa, aod, (size_t)h * 8));
    SLAM_TRY(h2d(ctx, b + ob, aoa, (size_t)w * 8));
    SLAM_TRY(h2d(ctx, b + ol, lut256x3, 768));
    SLAM_HIP(ctx, hipMemsetAsync(b + ost, 0, SLAM_STAT_WORDS * 4, ctx->stream));
    SLAM_TRY(render_only(ctx, (double *)(b + om), h, w, (double *)(b + oa), (double *)(b + ob), use_log, vmin, vmax,
                         b + ol, b + oc, tile_rgba ? b + ot : nullptr, W, H, (int16_t *)(b + ocm), (int16_t *)(b + orm),
                         (uint32_t *)(b + ost)));
    uint32_t st[SLAM_STAT_WORDS];
    SLAM_TRY(read_stats(ctx, (uint32_t *)(b + ost), st));
    if (st[SLAM_STAT_ALLNAN] == 2) return slam_fail(ctx, SLAM_E_LIMITS, "Invalid vmin or vmax");
    if (st[SLAM_STAT_ALLNAN]) return slam_fail(ctx, SLAM_E_ALLNAN, "matrix has no finite cell");
    if (vmin_vmax_out) {
        uint64_t lo = (uint64_t)st[SLAM_STAT_VMIN_LO] | ((uint64_t)st[SLAM_STAT_VMIN_HI] << 32);
        uint64_t hi = (uint64_t)st[SLAM_STAT_VMAX_LO] | ((uint64_t)st[SLAM_STAT_VMAX_HI] << 32);
        memcpy(&vmin_vmax_out[0], &lo, 8);
        memcpy(&vmin_vmax_out[1], &hi, 8);
    }
    if (cells_rgba) SLAM_TRY(d2h(ctx, cells_rgba, b + oc, (size_t)h * w * 4));
    if (tile_rgba) SLAM_TRY(d2h(ctx, tile_rgba, b + ot, (size_t)W * H * 4));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// dictionary correlation of the path estimators
// ------------------------------------------------------------------------------------------------------
static int check_dict(slam_ctx *ctx, const void *phi_rx, int n_ue, int ga, const void *phi_tx, int n_bs, int gd, const void *m,
                      uint32_t n_frames, const void *out)
{
    if (n_ue < 1 || n_ue > 64 || n_bs < 1 || n_bs > 64)
        return slam_fail(ctx, SLAM_E_RANGE, "slam_dict_correlate: %d x %d beams, at most 64 a side", n_ue, n_bs);
    if (ga < 1 || gd < 1 || ga > 65535 * 64)
        return slam_fail(ctx, SLAM_E_RANGE, "slam_dict_correlate: search grid of %d x %d points", ga, gd);
    if (n_frames > 65535u) return slam_fail(ctx, SLAM_E_RANGE, "slam_dict_correlate: at most 65535 matrices per call, got %u", n_frames);
    if (!phi_rx || !phi_tx || (n_frames && (!m || !out))) return slam_fail(ctx, SLAM_E_INVALID, "slam_dict_correlate: NULL pointer");
    return SLAM_OK;
}

extern "C" int slam_dict_correlate_dev(slam_ctx *ctx, const double *phi_rx_dev, int n_ue, int ga, const double *phi_tx_dev, int n_bs,
                                       int gd, const double *m_dev, uint32_t n_frames, double *out_dev)
{
    REQUIRE_CTX(ctx);
    SLAM_TRY(check_dict(ctx, phi_rx_dev, n_ue, ga, phi_tx_dev, n_bs, gd, m_dev, n_frames, out_dev));
    return dict_correlate(ctx, phi_rx_dev, n_ue, ga, phi_tx_dev, n_bs, gd, m_dev, n_frames, out_dev);
}

extern "C" int slam_dict_correlate(slam_ctx *ctx, const double *phi_rx, int n_ue, int ga, const double *phi_tx, int n_bs, int gd,
                                   const double *m, uint32_t n_frames, double *out)
{
    REQUIRE_CTX(ctx);
    SLAM_TRY(check_dict(ctx, phi_rx, n_ue, ga, phi_tx, n_bs, gd, m, n_frames, out));
    if (n_frames == 0) return SLAM_OK;
    ws_plan p;
    const size_t ba = (size_t)n_ue * ga * 8, bb = (size_t)n_bs * gd * 8, bm = (size_t)n_frames * n_ue * n_bs * 8,
                 bo = (size_t)n_frames * ga * gd * 8;
    const size_t oa = p.add(ba), ob = p.add(bb), om = p.add(bm), oo = p.add(bo);
    SLAM_TRY(slam_ws_reserve(ctx, p.total));
    uint8_t *b = ctx->ws;
    SLAM_TRY(h2d(ctx, b + oa, phi_rx, ba));
    SLAM_TRY(h2d(ctx, b + ob, phi_tx, bb));
    SLAM_TRY(h2d(ctx, b + om, m, bm));
    SLAM_TRY(dict_correlate(ctx, (const double *)(b + oa), n_ue, ga, (const double *)(b + ob), n_bs, gd, (const double *)(b + om),
                            n_frames, (double *)(b + oo)));
    SLAM_TRY(d2h(ctx, out, b + oo, bo));
    SLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------------
// batched device-resident pipeline
// ------------------------------------------------------------------------------------------------------
struct batch_plan {
    size_t fw[6], cw[16], rw[12], tc, kc, tb[3], zw[5];
    uint32_t nt_max, st_max, bl_cap;
    int32_t n_axis;             // the batch's, or 64 where the batch leaves it unset (no render stage)
    size_t total;
};

static int plan_batch(slam_ctx *ctx, const slam_batch *bt, batch_plan &bp)
{
    if (!bt || !bt->text_dev || !bt->text_off_dev || !bt->text_len_dev || !bt->stats_dev)
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: text, offsets, lengths and stats are required");
    if (bt->n_frames > 65535u)          // frames are the y (z for the raster) dimension of the launch grids
        return slam_fail(ctx, SLAM_E_RANGE, "slam_batch: at most 65535 frames per call, got %u", bt->n_frames);
    if (bt->tile_dev && (bt->tile_w < 1 || bt->tile_h < 1))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: tile of %d x %d", bt->tile_w, bt->tile_h);
    if (bt->tile_dev && (int64_t)bt->tile_h > 16 * 65535ll)
        return slam_fail(ctx, SLAM_E_RANGE, "slam_batch: tile of %d x %d is beyond the raster kernel's grid", bt->tile_w, bt->tile_h);
    bp.st_max = slot_tiles_for(bt->src_is_bytes ? SRC_RAW : SRC_TEXT3, bt->max_text_len);
    if (bt->row_cap % 16 || (uint64_t)bt->row_cap < (uint64_t)bp.st_max * SLOT_ROWS)
        return slam_fail(ctx, SLAM_E_INVALID,
                         "slam_batch: row_cap must be a multiple of 16 and >= slam_row_tiles(...) * SLAM_TILE_SLOTS");
    ws_plan p;
    bp.tc = p.add((size_t)bt->n_frames * bp.st_max * sizeof(uint32_t));
    bp.kc = p.add((size_t)bt->n_frames * std::max((bt->row_cap + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, bp.st_max) * sizeof(uint32_t));
    bp.tb[0] = p.add((size_t)bt->n_frames * bp.st_max * sizeof(uint32_t));
    bp.tb[1] = p.add((size_t)bt->n_frames * bp.st_max * sizeof(uint32_t));
    bp.tb[2] = p.add((size_t)bt->n_frames * bp.st_max * BL_PAY_CAP * sizeof(unsigned long long));
    bp.nt_max = frame_tiles_for(bt->src_is_bytes ? SRC_RAW : SRC_TEXT3, bt->max_text_len);
    bp.bl_cap = correct_ws::default_bl_cap(bt->row_cap);
    // sized before the stages are known: a batch without a render stage may leave n_axis unset
    bp.n_axis = bt->n_axis >= 1 && bt->n_axis <= SLAM_MAX_AXIS ? bt->n_axis : SMALL_AXIS;
    frame_ws::plan(p, bt->n_frames, bp.nt_max, bp.fw);
    correct_ws::plan(p, bt->n_frames, bt->row_cap, bp.bl_cap, bp.cw);
    render_ws::plan(p, bt->n_frames, bt->tile_w, bt->tile_h, bp.n_axis, bp.rw);
    fused_ws::plan(p, bt->n_frames, bp.st_max, bp.zw);
    bp.total = p.total;
    return SLAM_OK;
}

extern "C" int slam_raster_dev(slam_ctx *ctx, const uint8_t *cells_dev, const int16_t *colmap_dev,
                               const int16_t *rowmap_dev, const uint32_t *stats_dev, uint32_t n_frames,
                               int32_t tile_w, int32_t tile_h, uint8_t *tile_dev, uint32_t cells_per_frame)
{
    REQUIRE_CTX(ctx);
    if (!cells_dev || !colmap_dev || !rowmap_dev || !stats_dev || !tile_dev || tile_w < 1 || tile_h < 1)
        return slam_fail(ctx, SLAM_E_INVALID, "slam_raster_dev: NULL pointer or empty tile");
    if (((uintptr_t)tile_dev & 15u) || ((uintptr_t)cells_dev & 3u))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_raster_dev: tile must be 16-byte aligned, cells 4-byte aligned");
    if (cells_per_frame == 0 || cells_per_frame > (uint32_t)SLAM_MAX_AXIS * SLAM_MAX_AXIS)
        return slam_fail(ctx, SLAM_E_INVALID, "slam_raster_dev: cells_per_frame must be slam_cells_per_frame(n_axis)");
    return raster_only(ctx, cells_dev, colmap_dev, rowmap_dev, stats_dev, n_frames, tile_w, tile_h, tile_dev, cells_per_frame);
}

extern "C" size_t slam_batch_sizeof(void) { return sizeof(slam_batch); }

extern "C" int slam_reserve_workspace(slam_ctx *ctx, const slam_batch *batch)
{
    REQUIRE_CTX(ctx);
    batch_plan bp;
    SLAM_TRY(plan_batch(ctx, batch, bp));
    double *kern;
    int ks;
    SLAM_TRY(get_kernel(ctx, batch->blur_sigma, &kern, &ks));
    return slam_ws_reserve(ctx, bp.total);
}

// argument checks shared by the two batched entry points
static int check_batch(slam_ctx *ctx, const slam_batch *bt, uint32_t stages)
{
    if ((stages & 1u) && (!bt->flag_dev || !bt->ue_dev || !bt->bs_dev || !bt->rss_dev || !bt->clk_dev))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: decoded-record arrays are required");
    if ((stages & 2u) && (!bt->kept_ue_dev || !bt->kept_bs_dev || !bt->kept_rss_dev || !bt->kept_clk_dev))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: corrected-row arrays are required");
    if ((stages & 4u) && (!bt->beam_to_axis_dev || !bt->axis_angle_dev || !bt->lut_dev || bt->n_axis < 1 ||
                          bt->n_axis > SLAM_MAX_AXIS || (bt->tile_dev && (bt->tile_w < 1 || bt->tile_h < 1))))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: render stage needs the axis table, LUT and tile size");
    if (((uintptr_t)bt->flag_dev | (uintptr_t)bt->ue_dev | (uintptr_t)bt->bs_dev | (uintptr_t)bt->rss_dev |
         (uintptr_t)bt->clk_dev) & 15u)
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: decoded-record arrays must be 16-byte aligned");
    if ((stages & 2u) && (bt->cycle <= 0 || bt->cycle > (1 << 30)))
        return slam_fail(ctx, SLAM_E_INVALID, "slam_batch: cycle must be in 1..2^30");
    return SLAM_OK;
}

// One batch with its scratch at `b` (bp.total bytes, laid out by plan_batch) and the ks x ks blur weights at `kern`:
// only kernel launches on the ctx's stream from here on.
static int run_batch(slam_ctx *ctx, const slam_batch *bt, uint32_t stages, const batch_plan &bp, uint8_t *b,
                     const double *kern, int ks)
{
    const uint32_t F = bt->n_frames;
    frame_ws fws;
    fws.bind(b, bp.fw, bp.nt_max);
    correct_ws cws;
    cws.bind(b, bp.cw, bp.bl_cap);
    render_ws rws;
    rws.bind(b, bp.rw, bp.n_axis);
    if (bt->cells_dev) rws.cells = bt->cells_dev;
    if (bt->colmap_dev) rws.colmap = bt->colmap_dev;
    if (bt->rowmap_dev) rws.rowmap = bt->rowmap_dev;
    uint32_t *tile_cnt = bt->tile_cnt_dev ? bt->tile_cnt_dev : (uint32_t *)(b + bp.tc);
    rows_out rows{bt->flag_dev, bt->ue_dev, bt->bs_dev, bt->rss_dev, bt->clk_dev, bt->row_cap, tile_cnt, bp.st_max};
    tile_bits tbits;
    tbits.rmask = cws.rmask;
    tbits.nbound = (uint32_t *)(b + bp.tb[0]);
    tbits.nbase = (uint32_t *)(b + bp.tb[1]);
    tbits.pay = (unsigned long long *)(b + bp.tb[2]);
    tbits.stride = bp.st_max;
    if (stages & 2u) rows.bits = tbits;          // the framing pass leaves the correction's per-tile bits
    const uint32_t kept_stride = std::max((bt->row_cap + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, bp.st_max);
    uint32_t *kept_cnt = bt->kept_cnt_dev ? bt->kept_cnt_dev : (uint32_t *)(b + bp.kc);
    correct_params prm;
    if (stages & 2u) {
        prm.set_cycle(bt->cycle);
        prm.tol = bt->tol;
        prm.mod_base = bt->mod_base;
    }
    kept_out_t<uint8_t, uint32_t> kout{bt->kept_ue_dev, bt->kept_bs_dev, bt->kept_rss_dev, bt->kept_clk_dev, kept_cnt, kept_stride};
    // decode and correction asked for together: one fused kernel, the separate correction kernels only for the frames it flags
    const bool fused = ctx->fused && (stages & 3u) == 3u && fused_supported(prm);
    fused_args fz;
    if (fused) {
        fz.fw.bind(b, bp.zw, bp.st_max);
        fz.prm = prm;
        fz.prm.small_tol = 1;
        fz.kout = kout;
        rows.bits = tile_bits();                 // the fused kernel keeps the per-tile bits on chip
        cws.gate = fz.fw.redo;
    }
    if (stages & 1u) {
        SLAM_TRY(slam_zero_async(ctx, bt->stats_dev, (size_t)F * SLAM_STAT_WORDS * 4));
        frame_src fs{bt->text_dev, bt->text_off_dev, bt->text_len_dev, nullptr, 0, nullptr};
        SLAM_TRY(frame_decode(ctx, bt->src_is_bytes ? SRC_RAW : SRC_TEXT3, fs, F, bt->max_text_len, fws, bp.nt_max,
                              rows, bt->stats_dev, fused ? &fz : nullptr, ctx->force_two_pass));
    }
    if (stages & 2u) {
        rows_in_t<uint8_t, uint32_t> in{bt->flag_dev, bt->ue_dev, bt->bs_dev, bt->rss_dev, bt->clk_dev, bt->row_cap,
                                        tile_cnt, bp.st_max};
        in.bits = tbits;
        in.bits_fresh = (stages & 1u) && !fused ? 1 : 0;
        SLAM_TRY((correct_rows<uint8_t, uint32_t>(ctx, in, F, cws, prm, kout, nullptr, bt->stats_dev)));
    }
    if (stages & 4u) {
        render_params rp{bt->beam_to_axis_dev, bt->axis_angle_dev, bt->n_axis, bt->blur_sigma, bt->use_log,
                         bt->vmin, bt->vmax, bt->lut_dev, bt->tile_w, bt->tile_h, bt->full_grid ? 1 : 0};
        SLAM_TRY((pivot_accumulate<uint8_t, uint32_t>(ctx, bt->kept_ue_dev, bt->kept_bs_dev, bt->kept_rss_dev,
                                                      bt->row_cap, F, bt->stats_dev, SLAM_STAT_ROW_SLOTS, rp, rws,
                                                      kept_cnt, kept_stride)));
        SLAM_TRY(matrix_blur_render(ctx, F, rp, rws, kern, ks, bt->matrix_dev, bt->blurred_dev, bt->tile_dev,
                                    bt->stats_dev));
    }
    return SLAM_OK;
}

extern "C" int slam_pin_workspace(slam_ctx *ctx, int pinned)
{
    if (!ctx) return SLAM_E_INVALID;
    ctx->ws_pinned = pinned != 0;
    return SLAM_OK;
}

extern "C" int slam_process_frames_dev(slam_ctx *ctx, const slam_batch *bt, uint32_t stages)
{
    REQUIRE_CTX(ctx);
    batch_plan bp;
    SLAM_TRY(plan_batch(ctx, bt, bp));
    if (bt->n_frames == 0) return SLAM_OK;
    SLAM_TRY(check_batch(ctx, bt, stages));
    SLAM_TRY(slam_ws_reserve(ctx, bp.total));
    double *kern = nullptr;
    int ks = 1;
    if (stages & 4u) SLAM_TRY(get_kernel(ctx, bt->blur_sigma, &kern, &ks));
    return run_batch(ctx, bt, stages, bp, ctx->ws, kern, ks);
}

// ---- the same call on the library's fixed device scratch: nothing is allocated ---------------------------
namespace {
struct kern_by_value { double w[NOALLOC_KMAX * NOALLOC_KMAX]; };
// The Gaussian weights travel as a launch argument (3.5 KB) and are written to the scratch by the kernel itself:
// no host-to-device copy, so the step is an ordinary kernel node for a caller capturing the stream.
__global__ __launch_bounds__(256) void k_store_weights(kern_by_value k, int n, double *dst)
{
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = k.w[i];
}
}  // namespace

extern "C" int slam_process_frames_noalloc_dev(slam_ctx *ctx, const slam_batch *bt, uint32_t stages)
{
    REQUIRE_CTX(ctx);
    batch_plan bp;
    SLAM_TRY(plan_batch(ctx, bt, bp));
    if (bt->n_frames == 0) return SLAM_OK;
    SLAM_TRY(check_batch(ctx, bt, stages));
    if (ctx->device >= SCRATCH_MAX_DEVICES || !scratch_ptr[ctx->device])
        return slam_fail(ctx, SLAM_E_HIP, "slam_process_frames_noalloc_dev: no fixed scratch on device %d", ctx->device);
    if (bp.total + NOALLOC_KERN_BYTES > (size_t)SLAM_NOALLOC_SCRATCH_BYTES)
        return slam_fail(ctx, SLAM_E_CAPACITY,
                         "slam_process_frames_noalloc_dev: this batch shape needs %zu bytes of scratch, the fixed array "
                         "holds %zu (fewer or shorter frames per call, or slam_process_frames_dev)",
                         bp.total + NOALLOC_KERN_BYTES, (size_t)SLAM_NOALLOC_SCRATCH_BYTES);
    uint8_t *base = scratch_ptr[ctx->device];                 // the device address, never the symbol's name
    double *kern = reinterpret_cast<double *>(base);
    int ks = 1;
    // One call at a time per device on the array: behind the call before (any ctx, any stream, any thread).  A call
    // that is being captured into a graph takes no part in this: its replays are ordered by the stream they are
    // launched on, and keeping them apart from other users of the array is the caller's business.
    std::lock_guard<std::mutex> order(scratch_mutex[ctx->device]);
    const bool captured = stream_is_capturing(ctx);
    if (!captured) SLAM_HIP(ctx, hipStreamWaitEvent(ctx->stream, scratch_done[ctx->device], 0));
    const bool timing = ctx->timing;                          // event timers create events: off for this call
    ctx->timing = false;
    int rc = SLAM_OK;
    if (stages & 4u) {
        const double sigma = bt->blur_sigma;
        kern_by_value kv = {};
        if (sigma != sigma) rc = slam_fail(ctx, SLAM_E_INVALID, "blur sigma is NaN");
        else if (sigma > 0) {
            // gaussian_kernel (excel_heatmap_v3.py:320-333) on the stack, same arithmetic as slam_gaussian_kernel_host
            if (ceil(6.0 * sigma) > (double)NOALLOC_KMAX) {
                rc = slam_fail(ctx, SLAM_E_RANGE, "slam_process_frames_noalloc_dev: blur sigma %.3f needs more than %d taps",
                               sigma, NOALLOC_KMAX);
            } else {
                int size = (int)fmax(3.0, ceil(6.0 * sigma));
                if (size % 2 == 0) size++;
                const int c = size / 2;
                double total = 0.0;
                for (int y = -c; y <= c; y++)
                    for (int x = -c; x <= c; x++) {
                        const double v = exp(-(double)(x * x + y * y) / (2.0 * sigma * sigma));
                        kv.w[(y + c) * size + (x + c)] = v;
                        total += v;
                    }
                for (int i = 0; i < size * size; i++) kv.w[i] /= total;
                ks = size;
            }
        } else {
            kv.w[0] = 1.0;
        }
        if (rc == SLAM_OK) {
            hipLaunchKernelGGL(k_store_weights, dim3(1), dim3(256), 0, ctx->stream, kv, ks * ks, kern);
            if (hipGetLastError() != hipSuccess) rc = slam_fail(ctx, SLAM_E_HIP, "launch of store_weights failed");
        }
    }
    if (rc == SLAM_OK) rc = run_batch(ctx, bt, stages, bp, base + NOALLOC_KERN_BYTES, kern, ks);
    ctx->timing = timing;
    // recorded whatever rc is: kernels of a call that failed half way may still be running on the array
    if (!captured && hipEventRecord(scratch_done[ctx->device], ctx->stream) != hipSuccess && rc == SLAM_OK)
        rc = slam_fail(ctx, SLAM_E_HIP, "slam_process_frames_noalloc_dev: hipEventRecord failed");
    return rc;
}
