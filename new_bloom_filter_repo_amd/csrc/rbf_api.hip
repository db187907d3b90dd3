// rbf_api.hip -- C ABI (include/rbf.h) over the gfx950 kernels.  Host side: argument checks,
// scratch management, launches on the context's single HIP stream, optional per-kernel timing.
#include "../../include/rbf.h"
#include "rbf_kernels_i64.h"
#include "rbf_kernels_s64.h"
#include "rbf_kernels_u64.h"
#include "rbf_kernels_noise.h"
#include "rbf_kernels_pack.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

using namespace rbf;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(RBF_EIO, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct Timed { int id; hipEvent_t a, b; };

struct rbf_ctx {
    int device = 0;
    uint32_t cus = 256;              // compute units of the device (hipDeviceAttributeMultiprocessorCount; MI355X: 256)
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // scratch (grown on demand, never shrunk)
    uint32_t *seg_cnt = nullptr;     size_t seg_cnt_cap = 0;
    uint64_t *seg_off = nullptr;     size_t seg_off_cap = 0;
    uint32_t *chunk_off = nullptr;   size_t chunk_off_cap = 0;   // k_chunk_offsets: where every compaction / expansion workgroup's witness bits start
    uint64_t *pass_words = nullptr;  size_t pass_words_cap = 0;
    uint32_t *partials = nullptr;    size_t partials_cap = 0;
    uint2 *ins_records = nullptr;    size_t ins_records_cap = 0;  // two-kernel insert: 8 bytes per set mask bit of the batch
    uint32_t *ins_counters = nullptr; size_t ins_counters_cap = 0; // ... and the records appended so far, per frame
    int no_two_phase = 0;            // 1 = tiled k_insert_tab even when the filter needs several LDS tiles
    int hash_positions = 0;          // 1 = k_insert_positions hashes the set positions itself whatever the table size
    int no_table_rewrite = 0;        // 1 = the query kernel never rewrites the hash table, sole holder or not
    uint64_t *ones_acc = nullptr;    size_t ones_acc_cap = 0;     // where the mask kernels count; k_finish_ones hands the counts out and re-zeroes it
    uint32_t *mask_ticket = nullptr;                              // the fused tail of the GOP mask kernel: workgroups done so far (zero between launches)
    int no_fused_finish = 0;                                      // 1 = always the separate k_finish_ones launch (rbf_ctx_option RBF_OPT_SEPARATE_FINISH)
    uint32_t insert_slices = 0;                                   // tuning (RBF_OPT_INSERT_SLICES): mask slices per frame of the single-tile insert, 0 = auto
    bool ones_acc_dirty = false;     // a call failed between the mask kernels and k_finish_ones
    uint32_t *qimage = nullptr;      size_t qimage_cap = 0;       // probe image of the batch's filters (FP64 query kernel)
    int32_t *thr_tab = nullptr;      size_t thr_tab_cap = 0;      // per-pair thresholds of the mask kernels
    uint64_t *pack_base = nullptr;   size_t pack_base_cap = 0;    // running record size between pack chunks
    int force_generic = 0;           // tests: 1 = never use the LDS fast path
    int single_buffer = 0;           // tests: 1 = fast query path without filter double-buffering
    uint32_t mask_chunks = 0;        // tuning: temporal chunks of the GOP mask kernel (0 = auto)
    int force_generic_mask_bits = 0; // tests: 1 = per-pixel threshold compare even for threshold 0
    int barrett_only = 0;            // tests/tuning: 1 = never take the FP64 reductions (mod_m_f64)
    int hash_rebuild = 0;            // 1 = run k_hash_table for every batch instead of taking the table the last query kernel wrote
    int no_hash_table = 0;           // 1 = the insert kernel hashes the set positions itself
    struct SharedHashTable *hash_shared = nullptr;                // the pixel-index hash table this context holds a reference to
    uint4 *hash_tab = nullptr;                                    // = hash_shared->table
    uint32_t tile_words = 0;         // tests/tuning: cap the LDS filter tile (dwords); forces the tiled kernels
    // host staging of encode_gop: device-visible pinned block [flag | ones...] the GPU publishes into
    uint64_t *ones_pinned = nullptr; size_t host_cap = 0;
    uint64_t *ones_mapped_dev = nullptr;     // device address of the same block
    uint64_t publish_token = 0;
    struct PendingGop {                      // between rbf_encode_gop_begin and rbf_encode_gop_finish
        bool active = false;
        uint64_t token = 0, n = 0; uint32_t pairs = 0; rbf_seeds seeds{};
        const void *masks_dev = nullptr; uint64_t mask_stride_bytes = 0;
        void *filters_dev = nullptr; uint64_t filter_stride_bytes = 0;
        void *witnesses_dev = nullptr; uint64_t witness_stride_bytes = 0; uint64_t *stats_dev = nullptr;
        bool has_skip = false;                   // ctx->run_skip[p] != 0: pair p crosses a keyframe and is not coded
    } gop;
    std::vector<uint8_t> run_skip;
    std::vector<rbf_filter_params> plan;
    std::vector<double> plan_k;
    // timing
    uint32_t timing = 0;             // bit k: bracket launches of kernel id k with HIP events
    std::vector<Timed> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[RBF_K_COUNT] = {0};
    uint64_t launches[RBF_K_COUNT] = {0};
};

static int grow(void **ptr, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return RBF_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(ptr, want));
    *cap = want;
    return RBF_OK;
}

// Every entry point starts here.
static int set_device(rbf_ctx *ctx)
{
    if (!ctx) return fail(RBF_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return RBF_OK;
}

struct LaunchTimer {
    rbf_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr; bool on; hipStream_t st;
    LaunchTimer(rbf_ctx *ctx, int kid) : c(ctx), id(kid), on((ctx->timing >> kid) & 1u), st(ctx->stream)
    {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        a = get(); b = get();
        if (!a || !b) { on = false; return; }
        (void)hipEventRecord(a, st);
    }
    ~LaunchTimer()
    {
        if (!on) return;
        (void)hipEventRecord(b, st);
        try { c->pending.push_back({id, a, b}); }                 // timing is best effort; nothing may throw across the C ABI
        catch (...) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    }
};

// ------------------------------------------------------------------------------------------
// The pixel-index hash table (k_hash_table, 26 bytes per pixel inside an allocation of 32: rbf_kernels_q64.h) depends on the device, the frame size and the seeds
// only, so the contexts of one process SHARE it: four pipelines coding 1080p GOPs gather from one 54 MB table that the
// 256 MB Infinity Cache can keep, instead of four private ones that it cannot (measured: 0.197 -> 0.18x ms per step).
// Built once by the first context that needs it (on its stream; the others make their streams wait for the `ready`
// event), freed when the last reference goes.
// ------------------------------------------------------------------------------------------
struct SharedHashTable {
    int device; uint64_t n; rbf_seeds seeds;
    uint4 *table; size_t bytes;
    hipEvent_t ready;
    int refs;
};
static std::mutex g_hash_mu;
static std::vector<SharedHashTable *> g_hash_tables;

static void hash_table_release(rbf_ctx *ctx)
{
    SharedHashTable *t = ctx->hash_shared;
    if (!t) return;
    (void)hipStreamSynchronize(ctx->stream);                      // my kernels no longer read it
    ctx->hash_shared = nullptr; ctx->hash_tab = nullptr;
    std::lock_guard<std::mutex> lk(g_hash_mu);
    if (--t->refs > 0) return;
    for (size_t i = 0; i < g_hash_tables.size(); ++i)
        if (g_hash_tables[i] == t) { g_hash_tables[i] = g_hash_tables.back(); g_hash_tables.pop_back(); break; }
    (void)hipEventDestroy(t->ready);
    (void)hipFree(t->table);
    delete t;
}

// The table of (ctx->device, n, seeds) in ctx->hash_tab, built if nobody has it yet.  false: no device memory (the caller hashes
// in the insert kernel instead).  *sole: this context is the only holder.
static bool hash_table_acquire(rbf_ctx *ctx, uint64_t n, const rbf_seeds &seeds, bool *built)
{
    *built = false;
    SharedHashTable *cur = ctx->hash_shared;
    if (cur && cur->n == n && cur->seeds.h1 == seeds.h1 && cur->seeds.h2 == seeds.h2 && cur->seeds.act == seeds.act) return true;
    hash_table_release(ctx);
    std::lock_guard<std::mutex> lk(g_hash_mu);
    for (SharedHashTable *t : g_hash_tables)
        if (t->device == ctx->device && t->n == n && t->seeds.h1 == seeds.h1 && t->seeds.h2 == seeds.h2 && t->seeds.act == seeds.act) {
            if (hipStreamWaitEvent(ctx->stream, t->ready, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            ++t->refs;
            ctx->hash_shared = t; ctx->hash_tab = t->table;
            return true;
        }
    SharedHashTable *t = new (std::nothrow) SharedHashTable{ctx->device, n, seeds, nullptr, ((size_t)n + QL_SEG_PIXELS) * 32, nullptr, 1};
    if (!t) return false;
    if (hipMalloc((void **)&t->table, t->bytes) != hipSuccess) { (void)hipGetLastError(); delete t; return false; }
    if (hipEventCreateWithFlags(&t->ready, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(t->table); delete t; return false; }
    const uint64_t segs = (n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
    {
        LaunchTimer timer(ctx, RBF_K_HASHTAB);
        hipLaunchKernelGGL(k_hash_table, dim3((uint32_t)((segs + HT_THREADS / WAVE - 1) / (HT_THREADS / WAVE))), dim3(HT_THREADS), 0, ctx->stream,
                           n, Seeds{seeds.h1, seeds.h2, seeds.act}, t->table);
    }
    // a table whose kernel never ran must not be published to the other contexts of the process
    if (hipGetLastError() != hipSuccess || hipEventRecord(t->ready, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipEventDestroy(t->ready); (void)hipFree(t->table); delete t;
        return false;
    }
    try { g_hash_tables.push_back(t); } catch (...) { (void)hipStreamSynchronize(ctx->stream); (void)hipEventDestroy(t->ready); (void)hipFree(t->table); delete t; return false; }
    ctx->hash_shared = t; ctx->hash_tab = t->table;
    *built = true;
    return true;
}


static int drain_timing(rbf_ctx *ctx)
{
    if (ctx->pending.empty()) return RBF_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto &t : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            ctx->total_ms[t.id] += ms;
            ctx->launches[t.id] += 1;
        }
        try { ctx->pool.push_back(t.a); } catch (...) { (void)hipEventDestroy(t.a); }
        try { ctx->pool.push_back(t.b); } catch (...) { (void)hipEventDestroy(t.b); }
    }
    ctx->pending.clear();
    return RBF_OK;
}

extern "C" {

int rbf_version(void) { return RBF_ABI_VERSION; }
const char *rbf_last_error(void) { return g_err; }

int rbf_device_count(int *count)
{
    if (!count) return fail(RBF_EINVAL, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(RBF_EIO, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return RBF_OK;
}

int rbf_ctx_create(int device, void *hip_stream, rbf_ctx **out)
{
    if (!out) return fail(RBF_EINVAL, "out is null");
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(RBF_EINVAL, "device %d out of range (have %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    rbf_ctx *c = new (std::nothrow) rbf_ctx();
    if (!c) return fail(RBF_ENOMEM, "out of host memory");
    c->device = device;
    {
        int cus = 0;                                               // workgroup counts are sized for THIS device
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = (uint32_t)cus;
    }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->owns_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(RBF_EIO, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->owns_stream = true;
    }
    *out = c;
    return RBF_OK;
}

int rbf_ctx_destroy(rbf_ctx *ctx)
{
    if (!ctx) return RBF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    if (ctx->ones_pinned) (void)hipHostFree(ctx->ones_pinned);
    if (ctx->seg_cnt) (void)hipFree(ctx->seg_cnt);
    if (ctx->seg_off) (void)hipFree(ctx->seg_off);
    if (ctx->chunk_off) (void)hipFree(ctx->chunk_off);
    if (ctx->pass_words) (void)hipFree(ctx->pass_words);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->ins_records) (void)hipFree(ctx->ins_records);
    if (ctx->ins_counters) (void)hipFree(ctx->ins_counters);
    if (ctx->qimage) (void)hipFree(ctx->qimage);
    if (ctx->ones_acc) (void)hipFree(ctx->ones_acc);
    if (ctx->mask_ticket) (void)hipFree(ctx->mask_ticket);
    hash_table_release(ctx);
    if (ctx->thr_tab) (void)hipFree(ctx->thr_tab);
    if (ctx->pack_base) (void)hipFree(ctx->pack_base);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RBF_OK;
}

int rbf_ctx_sync(rbf_ctx *ctx)
{
    if (int r = set_device(ctx)) return r;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RBF_OK;
}

int rbf_malloc(rbf_ctx *ctx, size_t bytes, void **out_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!out_dev) return fail(RBF_EINVAL, "out_dev is null");
    *out_dev = nullptr;
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(out_dev, bytes);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RBF_ENOMEM : RBF_EIO, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return RBF_OK;
}

int rbf_free(rbf_ctx *ctx, void *ptr_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!ptr_dev) return RBF_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(ptr_dev));
    return RBF_OK;
}

int rbf_memset(rbf_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    if (int r = set_device(ctx)) return r;
    if (bytes == 0) return RBF_OK;
    if (!dst_dev) return fail(RBF_EINVAL, "dst_dev is null");
    HIP_TRY(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return RBF_OK;
}

int rbf_memcpy_h2d(rbf_ctx *ctx, void *dst_dev, const void *src, size_t bytes)
{
    if (int r = set_device(ctx)) return r;
    if (bytes == 0) return RBF_OK;
    if (!dst_dev || !src) return fail(RBF_EINVAL, "null pointer");
    HIP_TRY(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RBF_OK;
}

int rbf_memcpy_d2h(rbf_ctx *ctx, void *dst, const void *src_dev, size_t bytes)
{
    if (int r = set_device(ctx)) return r;
    if (bytes == 0) return RBF_OK;
    if (!dst || !src_dev) return fail(RBF_EINVAL, "null pointer");
    HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RBF_OK;
}

int rbf_memcpy_d2d(rbf_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    if (int r = set_device(ctx)) return r;
    if (bytes == 0) return RBF_OK;
    if (!dst_dev || !src_dev) return fail(RBF_EINVAL, "null pointer");
    HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return RBF_OK;
}

int rbf_timing_enable(rbf_ctx *ctx, int on)
{
    if (int r = set_device(ctx)) return r;
    if (!on) { if (int r = drain_timing(ctx)) return r; }
    ctx->timing = on == 1 ? 0xFFFFFFFFu : (uint32_t)on;
    return RBF_OK;
}

int rbf_ctx_force_generic(rbf_ctx *ctx, int on)
{
    if (!ctx) return fail(RBF_EINVAL, "null context");
    ctx->force_generic = (on & 1) ? 1 : 0;
    ctx->single_buffer = (on & 2) ? 1 : 0;
    ctx->force_generic_mask_bits = (on & 4) ? 1 : 0;
    ctx->barrett_only = (on & 8) ? 1 : 0;
    ctx->hash_rebuild = (on & 16) ? 1 : 0;
    ctx->no_hash_table = (on & 32) ? 1 : 0;
    ctx->no_two_phase = (on & 128) ? 1 : 0;
    ctx->mask_chunks = (uint32_t)(on >> 8) & 0x1F;           // tuning knob, bits 8..12
    ctx->hash_positions = (on & (1 << 14)) ? 1 : 0;
    ctx->no_table_rewrite = (on & (1 << 15)) ? 1 : 0;
    ctx->tile_words = ((uint32_t)on >> 16) << 6;             // bits 16..31: LDS tile cap in units of 64 dwords
    return RBF_OK;
}

int rbf_ctx_option(rbf_ctx *ctx, int option, int64_t value)
{
    if (!ctx) return fail(RBF_EINVAL, "null context");
    switch (option) {
    case RBF_OPT_SEPARATE_FINISH: ctx->no_fused_finish = value ? 1 : 0; return RBF_OK;
    case RBF_OPT_INSERT_SLICES: ctx->insert_slices = value < 0 ? 0u : (uint32_t)value; return RBF_OK;
    default: return fail(RBF_EINVAL, "unknown option %d", option);
    }
}

int rbf_timing_reset(rbf_ctx *ctx)
{
    if (int r = set_device(ctx)) return r;
    if (int r = drain_timing(ctx)) return r;
    for (int k = 0; k < RBF_K_COUNT; ++k) { ctx->total_ms[k] = 0; ctx->launches[k] = 0; }
    return RBF_OK;
}

int rbf_timing_read(rbf_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches)
{
    if (int r = set_device(ctx)) return r;
    if (kernel_id < 0 || kernel_id >= RBF_K_COUNT) return fail(RBF_EINVAL, "kernel id %d", kernel_id);
    if (int r = drain_timing(ctx)) return r;
    if (total_ms) *total_ms = ctx->total_ms[kernel_id];
    if (launches) *launches = ctx->launches[kernel_id];
    return RBF_OK;
}

// ------------------------------------------------------------------------------------------
// host math
// ------------------------------------------------------------------------------------------
int rbf_optimal_params(uint64_t n, uint64_t ones, double *k_out, uint64_t *l_out)
{
    if (!k_out || !l_out) return fail(RBF_EINVAL, "null output");
    *k_out = 0.0; *l_out = 0;
    if (n == 0 || ones > n) return fail(RBF_EINVAL, "need 0 <= ones <= n, n > 0");
    // np.sum(uint8) / n : both operands become float64 (improved_video_compressor.py:211-212)
    const double p = (double)ones / (double)n;
    if (p <= 0.0001) return RBF_OK;                         // :174
    if (p >= 0.32453) return RBF_OK;                        // :177 (P_STAR)
    const double q = 1 - p;
    const double L = std::log(2.0);
    const double k = std::log2(q * std::pow(L, 2.0) / p);   // :185  (float ** int -> pow)
    if (std::isnan(k) || k <= 0) return RBF_OK;             // :188
    const double gamma = 1 / L;
    const double lf = p * (double)n * k * gamma;            // :193, left-to-right products
    const uint64_t l = (uint64_t)lf;                        // int(): truncation (lf > 0)
    *k_out = k > 0.1 ? k : 0.1;                             // max(0.1, k)
    *l_out = l > 1 ? l : 1;                                 // max(1, l)
    return RBF_OK;
}

int rbf_activation_threshold(double k_star, uint32_t *floor_k, uint64_t *threshold)
{
    if (!floor_k || !threshold) return fail(RBF_EINVAL, "null output");
    if (!(k_star >= 0.0) || k_star > 64.0) return fail(RBF_ERANGE, "k* = %g outside [0, 64]", k_star);
    const double fl = std::floor(k_star);
    *floor_k = (uint32_t)fl;
    const double pa = k_star - fl;                          // p_activation, :58
    if (!(pa > 0.0)) { *threshold = 0; return RBF_OK; }     // `x < 0.0` is never true
    // RN(h / (2^64-1)) >= pa  <=>  h / (2^64-1) > mid, mid = midpoint of pa and its predecessor
    // (no tie is possible: mid is dyadic with an odd numerator, 2^64-1 is odd).
    int ex;
    const double fr = std::frexp(pa, &ex);                  // pa = fr * 2^ex, fr in [0.5, 1)
    const uint64_t A = (uint64_t)std::ldexp(fr, 53);        // 2^52 <= A < 2^53
    const int s = ex - 53 - 2;                              // pa = 4A * 2^s
    const uint64_t N = (A == (1ull << 52)) ? 4 * A - 1 : 4 * A - 2;   // mid = N * 2^s
    const int sh = -s;                                      // pa < 1 -> sh >= 55
    if (sh >= 128) { *threshold = 1; return RBF_OK; }
    const unsigned __int128 X = (unsigned __int128)N * (unsigned __int128)0xFFFFFFFFFFFFFFFFull;
    *threshold = (uint64_t)(X >> sh) + 1;                   // floor(mid * (2^64-1)) + 1
    return RBF_OK;
}

int rbf_plan_batch(uint64_t n, const uint64_t *ones, uint32_t nframes, int guard_l_ge_n,
                   rbf_filter_params *params, double *k_out)
{
    if (!ones || !params) return fail(RBF_EINVAL, "null pointer");
    for (uint32_t f = 0; f < nframes; ++f) {
        double k = 0.0; uint64_t l = 0;
        if (int r = rbf_optimal_params(n, ones[f], &k, &l)) return r;
        // compress(): `p >= P_STAR` is already (0, 0) in _calculate_optimal_params (:177, :215)
        const bool skip = (l == 0) || (guard_l_ge_n && l >= n) || l > 0xFFFFFFFFull;
        params[f].m = 0; params[f].floor_k = 0; params[f].threshold = 0;
        if (!skip) {
            params[f].m = (uint32_t)l;
            if (int r = rbf_activation_threshold(k, &params[f].floor_k, &params[f].threshold)) return r;
        }
        if (k_out) k_out[f] = skip ? 0.0 : k;
    }
    return RBF_OK;
}

// ------------------------------------------------------------------------------------------
// shared argument plumbing
// ------------------------------------------------------------------------------------------
static int fill_table(const rbf_filter_params *params, uint32_t count, FrameTable *tab)
{
    memset(tab, 0, sizeof *tab);
    for (uint32_t f = 0; f < count; ++f) {
        if (params[f].floor_k > 64) return fail(RBF_ERANGE, "frame %u: floor_k %u > 64", f, params[f].floor_k);
        FrameDev &d = tab->f[f];
        d.m = params[f].m;
        d.floor_k = params[f].floor_k;
        d.T = params[f].threshold;
        d.M = params[f].m >= 2 ? (uint64_t)((((unsigned __int128)1) << 64) / params[f].m) : 0;
    }
    return RBF_OK;
}

static int check_frame_geometry(uint64_t n, uint32_t nframes, uint64_t mask_stride_bytes)
{
    if (n == 0 || n > 0xFFFFFFFFull) return fail(RBF_ERANGE, "n = %llu outside [1, 2^32-1]", (unsigned long long)n);
    if (nframes == 0) return fail(RBF_EINVAL, "nframes must be >= 1");
    if (mask_stride_bytes % 8 || mask_stride_bytes < ((n + 63) / 64) * 8)
        return fail(RBF_EINVAL, "mask stride %llu must be a multiple of 8 and >= %llu", (unsigned long long)mask_stride_bytes,
                    (unsigned long long)(((n + 63) / 64) * 8));
    return RBF_OK;
}

static inline uint64_t nseg_of(uint64_t n) { return (n + SEG_PIXELS - 1) / SEG_PIXELS; }

// Which kernels serve a batch: the LDS-resident fast path needs the largest filter of the batch
// (plus per-wave staging) to fit one workgroup's 160 KiB of LDS; otherwise the generic kernels
// probe / set the filter in global memory.
constexpr size_t LDS_LIMIT = 160 * 1024;
struct Plan {
    bool fast_insert;            // LDS partial-filter insert (any filter size, tiled when needed)
    int query_kind;              // 0 generic (global probes), 1 LDS whole filter, 2 LDS tiles (Barrett), 3 LDS tiles, FP64 (k_query_s64t)
    bool double_buffer, small_m;
    bool insert_tab;             // insert through the hash table + FP64 reductions (same size condition, any LDS fit)
    bool insert_two_phase;       // ... as k_insert_positions + k_insert_records (filters of more than one LDS tile, counts known on the host)
    bool f64_mod;                // every coded frame has F64MOD_M_MIN <= m <= F64MOD_M_MAX: reductions through the FP64 pipe
    uint32_t fwords_max, S /* slices of a coded frame */, per_tile /* sum of slices */, insert_group /* coded frames per insert launch */;
    SliceTable slices;
    uint32_t insert_tile_words, insert_tiles, query_tile_words;
    size_t insert_lds_bytes, query_lds_bytes;
    uint64_t nseg; uint32_t words_per_seg;
    uint32_t image_stride_words;  // row pitch of the probe image (dwords, multiple of 4)
};

constexpr size_t HASH_TABLE_CACHE_BYTES = (size_t)96 << 20;    // k_insert_positions: a pixel-index table larger than this is not worth gathering from (1440p, 118 MB: step 454 -> 425 us hashed; 2160p, 265 MB: insert 124 -> 97; re-measured with the 26-byte table of round 4: 1440p equal, 2160p 207 -> 192 Gpixel/s gathered, profiles/r04_bigtable.txt)
constexpr uint32_t MAX_INSERT_TILES = 7, MAX_QUERY_TILES = 3;     // measured crossovers, see make_plan
constexpr uint64_t STREAM_MIN_PIXEL_FRAMES = 64ull * 1920 * 1080;  // pixels x coded frames of a launch from which its one-shot data is moved with non-temporal accesses (rbf_kernels_lds.h, cache-policy note): 1080p from 64 frames, 2160p from 16

static Plan make_plan(const rbf_ctx *ctx, const rbf_filter_params *params, uint32_t nframes, uint64_t n, bool have_ones = false)
{
    Plan p{};
    uint32_t mmax = 0, active = 0;
    p.small_m = true;
    p.f64_mod = !ctx->barrett_only;
    bool sizes_f64 = true;
    for (uint32_t f = 0; f < nframes; ++f) {
        if (params[f].m && (params[f].m < F64MOD_M_MIN || params[f].m > F64MOD_M_MAX)) sizes_f64 = false;
        if (params[f].m > mmax) mmax = params[f].m;
        if (params[f].m) ++active;
        if (params[f].m == 1 || params[f].m > (1u << 30)) p.small_m = false;
        if (params[f].m && (params[f].m < F64MOD_M_MIN || params[f].m > F64MOD_M_MAX)) p.f64_mod = false;
    }
    p.fwords_max = (uint32_t)(((uint64_t)mmax + 31) / 32);
    const size_t fbytes = (size_t)((p.fwords_max + 3u) & ~3u) * 4;
    // insert: whole partial filter in LDS next to the per-wave queues, else tiles of the largest size that fits
    // the table-driven insert kernel has the larger per-wave queue; size the tiles for whichever may run
    const size_t queue_bytes = (size_t)IL_WAVES * (IT_WAVE_LDS_BYTES > IL_QUEUE * 4 ? IT_WAVE_LDS_BYTES : IL_QUEUE * 4);
    const uint32_t max_tile_words = (uint32_t)((LDS_LIMIT - queue_bytes) / 4) & ~3u;
    p.insert_tile_words = ((p.fwords_max + 3u) & ~3u) <= max_tile_words ? ((p.fwords_max + 3u) & ~3u) : max_tile_words;
    if (ctx->tile_words && ctx->tile_words < p.insert_tile_words) p.insert_tile_words = ctx->tile_words & ~3u;
    if (p.insert_tile_words < 4) p.insert_tile_words = 4;
    p.insert_tiles = (p.fwords_max + p.insert_tile_words - 1) / p.insert_tile_words;
    if (p.insert_tiles < 1) p.insert_tiles = 1;
    p.insert_lds_bytes = (size_t)p.insert_tile_words * 4 + queue_bytes;
    // Tiling re-hashes (insert) / re-probes (query) every key once per tile, so its cost grows with the
    // tile count while the global-memory kernels' does not.  Measured ps per (pixel, frame), 4K..16K frames:
    // insert tiled 3.1 / 5.2 / 6.3 / 10.4 at 3 / 5 / 7 / 10 tiles vs 7.8 generic; query tiled 5.8 / 11.2 at
    // 2 / 4 tiles vs 8.1-9.0 generic (profiles/r01_large_frames.txt).  Past the crossover the generic kernels run
    // (16K frames: 59 instead of 9 Gpixel/s).
    const bool auto_tiles = ctx->tile_words == 0;
    p.fast_insert = !ctx->force_generic && mmax > 0 && !(auto_tiles && p.insert_tiles > MAX_INSERT_TILES);
    // query
    // k_query_u64 (FP64 reductions, probe image) is double-buffered only; its buffers end with the SAFE dwords
    // ... and it addresses its outputs with 32-bit offsets (pass bytes: frame * nseg * 64 + ...): a chunk of frames x pixels >= 2^35 goes to
    // the tiled kernel, which keeps 64-bit row bases
    const bool u64_offsets_fit = (uint64_t)nframes * ((n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS) * (QL_SEG_PIXELS / 8) < (1ull << 32);
    if (2 * (fbytes + 16) + u64_geo_bytes(nframes) > LDS_LIMIT || ctx->single_buffer || !u64_offsets_fit) p.f64_mod = false;      // (filters that fit twice but leave no room for the frame records go to the tiled kernel: one buffer)
    const size_t qbytes = fbytes + (p.f64_mod ? 16 : 0);
    p.double_buffer = 2 * qbytes <= LDS_LIMIT && !ctx->single_buffer;
    p.query_kind = 0;
    if (!ctx->force_generic && mmax > 0) {
        if (qbytes <= LDS_LIMIT && !ctx->tile_words) {
            p.query_kind = 1;
            p.query_lds_bytes = (p.double_buffer ? 2 : 1) * qbytes;
        } else {
            p.query_kind = 2;
            p.query_tile_words = (uint32_t)(LDS_LIMIT / 4);
            if (ctx->tile_words && ctx->tile_words < p.query_tile_words) p.query_tile_words = ctx->tile_words & ~3u;
            if (p.query_tile_words < 4) p.query_tile_words = 4;
            p.query_lds_bytes = (size_t)(p.query_tile_words < ((p.fwords_max + 3u) & ~3u) ? p.query_tile_words : ((p.fwords_max + 3u) & ~3u)) * 4;
            if (auto_tiles && (p.fwords_max + p.query_tile_words - 1) / p.query_tile_words > MAX_QUERY_TILES) p.query_kind = 0;
        }
    }
    // Slices per frame so that one launch has about one workgroup per CU (ctx->cus; 256 on MI355X).  With many frames or several
    // tiles the quotient gets small (4K, 29 frames, 3 tiles: 2 slices -> 174 long workgroups), so the frames are
    // inserted in groups of `insert_group` coded frames, each group one launch with >= INSERT_SLICES slices per frame
    // (4K: 10 frames x 8 slices x 3 tiles = 240 workgroups).  Handing out uneven slices to use all 256 CUs was
    // measured and buys nothing: a launch lasts as long as its largest slice.
    // Filters of several tiles, set-bit counts known (rbf_encode_gop): one walk of the masks into position records, then the tiles
    // are filled from the records -- no queues in that kernel, so a tile may take all of LDS.
    p.insert_two_phase = p.fast_insert && sizes_f64 && !ctx->no_hash_table && !ctx->barrett_only && !ctx->no_two_phase && have_ones && p.insert_tiles > 1;
    if (p.insert_two_phase) {
        const uint32_t cap = (uint32_t)(LDS_LIMIT / 4) & ~3u;
        uint32_t tw = (p.fwords_max + 3u) & ~3u;
        if (tw > cap) { const uint32_t nt = (p.fwords_max + cap - 1) / cap; tw = (((p.fwords_max + nt - 1) / nt) + 3u) & ~3u; }
        if (ctx->tile_words && (ctx->tile_words & ~3u) < tw) tw = ctx->tile_words & ~3u;
        if (tw < 4) tw = 4;
        p.insert_tile_words = tw;
        p.insert_tiles = (p.fwords_max + tw - 1) / tw;
        p.insert_lds_bytes = (size_t)tw * 4;
    }
    constexpr uint32_t INSERT_SLICES = 8;
    const uint32_t units = ctx->cus / p.insert_tiles ? ctx->cus / p.insert_tiles : 1u;     // workgroups per tile layer
    uint32_t group = units / INSERT_SLICES;                                  // coded frames per launch
    if (group < 1) group = 1;
    if (group > active) group = active ? active : 1;
    uint32_t base = units / group;
    if (base < 1) base = 1;
    if (base > 32) base = 32;
    // Single-tile filters (every frame size up to 1080p): ONE insert launch for the whole batch -- up to 128 coded frames when several
    // GOPs ride in one block (rbf_encode_runs) -- of about one workgroup per CU: the largest power of two of slices per frame that fits.
    // An insert workgroup costs its filter tile twice over (zeroed, then written out as a partial the reduce kernel reads back), so fewer,
    // longer workgroups win: measured on 4 x 29 frames of 1080p (profiles/r05_sweep1.txt, r05_sweep2.txt) 8 slices in groups of 32 frames
    // (round 4) 155 us, 8 slices in one launch 135, 4 slices 117, 2 slices 115, 1 slice 212 (116 of 256 CUs), 3 slices 170 (a slice no
    // longer stays on one XCD); 2 x 29 frames: 4 slices 58, 2 slices 102; 29 frames: 8 slices 33, 4 slices 54, 16 slices 42.
    if (p.insert_tiles == 1 && !p.insert_two_phase && active) {
        group = active;
        base = 1;
        while (base * 2 * active <= units && base < 32) base *= 2;
        if (ctx->insert_slices) base = ctx->insert_slices > 32 ? 32 : ctx->insert_slices;      // RBF_OPT_INSERT_SLICES (tuning)
    }
    p.insert_group = group;
    p.S = base;
    p.per_tile = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        const uint32_t sf = params[f].m ? base : 0u;
        p.slices.n[f] = (uint8_t)sf;
        p.per_tile += sf;
    }
    // FP64 geometries that do not fit LDS twice (or whose tile size a test caps): k_query_s64t, one buffer of maximal tiles
    if (!ctx->force_generic && mmax > 0 && sizes_f64 && !ctx->barrett_only && !ctx->single_buffer && !(p.query_kind == 1 && p.f64_mod)) {
        const uint32_t cap = (uint32_t)((LDS_LIMIT - S64_GEO_BYTES) / 4 - 4) & ~3u;       // one buffer of tile_words + 4 dwords, k_query_s64t's geometry behind it
        uint32_t tw = (p.fwords_max + 3u) & ~3u;
        if (tw > cap) { const uint32_t nt = (p.fwords_max + cap - 1) / cap; tw = (((p.fwords_max + nt - 1) / nt) + 3u) & ~3u; }
        if (ctx->tile_words && (ctx->tile_words & ~3u) < tw) tw = ctx->tile_words & ~3u;
        if (tw < 4) tw = 4;
        p.query_kind = 3;
        p.query_tile_words = tw;
        p.query_lds_bytes = (size_t)(tw + 4) * 4;
        p.f64_mod = true;                                         // the probe image is needed
    }
    if (p.query_kind != 1 && p.query_kind != 3) p.f64_mod = false;
    p.insert_tab = p.fast_insert && sizes_f64 && !ctx->no_hash_table && !ctx->barrett_only;
    p.image_stride_words = (p.fwords_max + 3u) & ~3u;
    const uint32_t segpx = (p.query_kind == 1 || p.query_kind == 3) ? (uint32_t)QL_SEG_PIXELS : p.query_kind == 2 ? (uint32_t)TQ_SEG_PIXELS : (uint32_t)SEG_PIXELS;
    p.nseg = (n + segpx - 1) / segpx;
    p.words_per_seg = segpx / 64;
    return p;
}

// The FrameTable k_query_s64t reads (rbf_kernels_s64.h): COMPACTED over the coded frames -- entry j = j-th coded frame: m, M = bits of
// -1/m, floor_k = floor(k*) | c << 8 | frame index << 16 (c = coded thresholds below the frame's own), T = j-th smallest threshold.
// `empty`: bit f = frame f is not coded.
static FrameTable query_table_s64(const FrameTable &tab, uint32_t nframes, uint32_t *nactive, uint64_t (&empty)[2])
{
    FrameTable q;
    memset(&q, 0, sizeof q);
    empty[0] = empty[1] = 0;
    uint64_t sorted[MAX_BATCH];
    uint32_t coded = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        if (tab.f[f].m) sorted[coded++] = tab.f[f].T;
        else empty[f >> 6] |= 1ull << (f & 63);
    }
    std::sort(sorted, sorted + coded);
    uint32_t j = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        if (!tab.f[f].m) continue;
        const double ninv = -1.0 / (double)tab.f[f].m;
        q.f[j].m = tab.f[f].m;
        memcpy(&q.f[j].M, &ninv, 8);
        q.f[j].floor_k = tab.f[f].floor_k | ((uint32_t)(std::lower_bound(sorted, sorted + coded, tab.f[f].T) - sorted) << 8) | (f << 16);
        q.f[j].T = sorted[j];
        ++j;
    }
    *nactive = coded;
    return q;
}

static int allow_big_lds(const void *fn)
{
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    return RBF_OK;
}

// ------------------------------------------------------------------------------------------
// A1
// ------------------------------------------------------------------------------------------
// per-pair thresholds travel as kernel arguments (captured at launch) into a device table
struct ThrChunk {
    static constexpr uint32_t N = 256;
    int32_t v[N];
};
__global__ void k_store_thresholds(const ThrChunk c, int32_t *__restrict__ dst, uint32_t count)
{
    if (threadIdx.x < count) dst[threadIdx.x] = c.v[threadIdx.x];
}

// The tail of a residual-mask pass, ONE launch instead of a memset in front of the mask kernels, a copy kernel behind
// them and two more memsets (rocprofv3: the four small launches were ~25 us of a ~215 us step).  The mask kernels count
// into a context-owned accumulator that is zero whenever they start; block 0 hands the counts to the caller's array (and,
// for rbf_encode_gop, into the device-visible pinned block whose flag word the host spins on) and zeroes the accumulator
// again; every block clears its share of up to two output regions (the witness rows and the stats of the batch).
__global__ __launch_bounds__(256) void k_finish_ones(uint64_t *__restrict__ acc, uint64_t *__restrict__ ones, uint32_t count,
                                                     uint64_t *host_block /* nullable */, uint64_t token,
                                                     uint4 *__restrict__ clear_a, uint64_t quads_a, uint4 *__restrict__ clear_b, uint64_t quads_b)
{
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
            const uint64_t v = acc[i];
            ones[i] = v;
            acc[i] = 0;
            if (host_block) __hip_atomic_store(&host_block[1 + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (host_block) {
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&host_block[0], token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads_a; i += (uint64_t)gridDim.x * blockDim.x) clear_a[i] = z;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads_b; i += (uint64_t)gridDim.x * blockDim.x) clear_b[i] = z;
}

static int launch_finish_ones(rbf_ctx *ctx, uint64_t *ones_dev, uint32_t pairs, uint64_t *host_block, uint64_t token,
                              void *clear_a, size_t bytes_a, void *clear_b, size_t bytes_b)
{
    // regions that are not 16-byte shaped fall back to a memset (never the case for the library's own buffers)
    if (clear_a && (((uintptr_t)clear_a | bytes_a) & 15)) { HIP_TRY(hipMemsetAsync(clear_a, 0, bytes_a, ctx->stream)); clear_a = nullptr; bytes_a = 0; }
    if (clear_b && (((uintptr_t)clear_b | bytes_b) & 15)) { HIP_TRY(hipMemsetAsync(clear_b, 0, bytes_b, ctx->stream)); clear_b = nullptr; bytes_b = 0; }
    const uint64_t quads = bytes_a / 16 + bytes_b / 16;
    uint32_t blocks = (uint32_t)((quads + 256 * 4 - 1) / (256 * 4));       // ~4 stores per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_finish_ones, dim3(blocks), dim3(256), 0, ctx->stream, ctx->ones_acc, ones_dev, pairs, host_block, token,
                       (uint4 *)clear_a, (uint64_t)(bytes_a / 16), (uint4 *)clear_b, (uint64_t)(bytes_b / 16));
    HIP_TRY(hipGetLastError());
    ctx->ones_acc_dirty = false;
    return RBF_OK;
}

// every argument check of the mask stage, with no side effect (rbf_encode_gop_begin runs it before it touches the stream)
static int check_mask_args(const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes, uint32_t width, uint32_t height,
                           uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes, uint32_t sample_bytes, const int32_t *thr_floors,
                           const void *masks_dev, uint64_t mask_stride_bytes, const uint64_t *ones_dev)
{
    if (!frames_dev || !masks_dev || !ones_dev) return fail(RBF_EINVAL, "null device pointer");
    if (nframes < 2) return fail(RBF_EINVAL, "need at least 2 frames, got %u", nframes);
    if (width == 0 || height == 0) return fail(RBF_EINVAL, "empty frame %ux%u", width, height);
    if (sample_bytes != 1 && sample_bytes != 2) return fail(RBF_EINVAL, "sample_bytes must be 1 or 2, got %u", sample_bytes);
    if (pixel_stride_bytes < sample_bytes || pixel_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "pixel stride %u incompatible with %u-byte samples", pixel_stride_bytes, sample_bytes);
    if (row_pitch_bytes < (uint64_t)width * pixel_stride_bytes || row_pitch_bytes % sample_bytes) return fail(RBF_EINVAL, "row pitch %llu too small or misaligned", (unsigned long long)row_pitch_bytes);
    if (frame_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "frame stride misaligned");
    if (int r = check_frame_geometry((uint64_t)width * height, nframes - 1, mask_stride_bytes)) return r;
    if (thr_floors)
        for (uint32_t i = 0; i + 1 < nframes; ++i)
            if (thr_floors[i] < 0) return fail(RBF_EINVAL, "negative threshold %d for pair %u", thr_floors[i], i);
    return RBF_OK;
}

static int residual_mask_impl(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                              uint32_t nframes, uint32_t width, uint32_t height,
                              uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                              uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                              void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev, bool finish,
                              const MaskFinish *gop_tail = nullptr /* rbf_encode_gop: publish + clears; fused into the mask kernel when it covers the frame */,
                              const uint8_t *skip = nullptr /* rbf_encode_runs: skip[p] != 0 = pair p is not coded (zero row, zero count) */)
{
    if (int r = set_device(ctx)) return r;
    if (int r = check_mask_args(frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes,
                                thr_floors, masks_dev, mask_stride_bytes, ones_dev)) return r;
    const uint64_t n = (uint64_t)width * height;
    const uint32_t pairs = nframes - 1;
    if (ctx->ones_acc_cap < (size_t)pairs * 8) {
        if (ctx->ones_acc) { HIP_TRY(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->ones_acc); ctx->ones_acc = nullptr; ctx->ones_acc_cap = 0; }
        const size_t want = ((size_t)pairs + 64) * 8;
        HIP_TRY(hipMalloc((void **)&ctx->ones_acc, want));
        ctx->ones_acc_cap = want;
        ctx->ones_acc_dirty = true;
    }
    if (ctx->ones_acc_dirty) {                                    // a launch failed or was abandoned: counts AND tickets start from zero again
        HIP_TRY(hipMemsetAsync(ctx->ones_acc, 0, ctx->ones_acc_cap, ctx->stream));
        if (ctx->mask_ticket) HIP_TRY(hipMemsetAsync(ctx->mask_ticket, 0, (MASK_TICKETS + 1) * 4, ctx->stream));
    }
    ctx->ones_acc_dirty = true;                                   // until k_finish_ones has been enqueued
    uint64_t *const acc = ctx->ones_acc;
    const uint64_t nwords = (n + 63) / 64;
    bool fused = false;
    // Fast path: flat frames, 16-byte aligned, whole 1024-pixel segments; the generic kernel does the rest.
    uint64_t fast_segs = 0;
    const bool flat = row_pitch_bytes == (uint64_t)width * pixel_stride_bytes;
    const bool known = (sample_bytes == 1 && (pixel_stride_bytes == 1 || pixel_stride_bytes == 3)) ||
                       (sample_bytes == 2 && (pixel_stride_bytes == 2 || pixel_stride_bytes == 6));
    if (!ctx->force_generic && flat && known && frame_stride_bytes % 16 == 0 && ((uintptr_t)frames_dev % 16) == 0 &&
        (size_t)pairs * 4 <= 48 * 1024)
        fast_segs = n / 1024;
    // temporal chunks of the fast kernel: enough waves to fill the chip (>= ~32 per CU) without re-reading much
    MaskChunks mc{};
    uint32_t chunks = 1;
    bool skip_in_table = false;                // the chunk table cuts the block at its keyframes: the fast kernel needs no threshold trick
    if (fast_segs) {
        uint32_t coded = pairs;
        if (skip) { coded = 0; for (uint32_t i = 0; i < pairs; ++i) coded += skip[i] ? 0u : 1u; }
        chunks = ctx->mask_chunks ? ctx->mask_chunks : (uint32_t)((6500 + fast_segs - 1) / fast_segs);   // 4 at 1080p (measured best)
        if (chunks > coded) chunks = coded;
        if (chunks < 1) chunks = 1;
        uint32_t ppc = (coded + chunks - 1) / chunks;
        if (ppc < 1) ppc = 1;
        if (!skip) {
            chunks = (pairs + ppc - 1) / ppc;
            mc.ppc = ppc;
        } else if (nframes < MASK_CHUNK_SKIP) {
            uint32_t cnt = 0;
            bool fits = true;
            for (uint32_t a = 0; a < pairs && fits;) {
                uint32_t b = a;
                while (b < pairs && (skip[b] != 0) == (skip[a] != 0)) ++b;
                const uint32_t len = b - a;
                if (skip[a]) {
                    if (cnt >= MASK_MAX_CHUNKS) { fits = false; break; }
                    mc.first[cnt] = (uint16_t)a; mc.pairs[cnt] = (uint16_t)(len | MASK_CHUNK_SKIP); ++cnt;
                } else {
                    const uint32_t c = (len + ppc - 1) / ppc, per = (len + c - 1) / c;     // this run in c chunks of about ppc pairs
                    for (uint32_t x = a; x < b; x += per) {
                        if (cnt >= MASK_MAX_CHUNKS) { fits = false; break; }
                        mc.first[cnt] = (uint16_t)x; mc.pairs[cnt] = (uint16_t)(b - x < per ? b - x : per); ++cnt;
                    }
                }
                a = b;
            }
            if (fits && cnt) { mc.count = cnt; chunks = cnt; skip_in_table = true; }
        }
        if (skip && !skip_in_table) {           // (a block of more runs than the table holds) uniform chunks over everything, the skipped pairs through their thresholds
            mc = MaskChunks{};
            chunks = ctx->mask_chunks ? ctx->mask_chunks : (uint32_t)((6500 + fast_segs - 1) / fast_segs);
            if (chunks > pairs) chunks = pairs;
            ppc = (pairs + chunks - 1) / chunks;
            chunks = (pairs + ppc - 1) / ppc;
            mc.ppc = ppc;
        }
    }
    // Per-pair thresholds travel as kernel arguments into a device table.  A skipped pair that a kernel WITHOUT the chunk table sees
    // (the generic kernel behind a ragged frame tail or an unaligned layout; the fast kernel of a block with more runs than the table holds)
    // gets the threshold INT32_MAX: `abs(diff) > thr` is then never true -- a zero row and a zero count, like the table's.
    const bool generic_runs = fast_segs * 16 < nwords;
    const int32_t *thr_tab = nullptr, *thr_tab_fast = nullptr;
    if (thr_floors || (skip && (generic_runs || !skip_in_table))) {
        // Kernel arguments are captured at launch, so the caller's array is free as soon as we return.
        if (int r = grow((void **)&ctx->thr_tab, &ctx->thr_tab_cap, (size_t)pairs * 4)) return r;
        for (uint32_t base = 0; base < pairs; base += ThrChunk::N) {
            ThrChunk c{};
            const uint32_t cnt = pairs - base < ThrChunk::N ? pairs - base : ThrChunk::N;
            for (uint32_t i = 0; i < cnt; ++i) c.v[i] = (skip && skip[base + i]) ? 0x7FFFFFFF : thr_floors ? thr_floors[base + i] : thr_floor;
            hipLaunchKernelGGL(k_store_thresholds, dim3(1), dim3(ThrChunk::N), 0, ctx->stream, c, ctx->thr_tab + base, cnt);
        }
        HIP_TRY(hipGetLastError());
        thr_tab = ctx->thr_tab;
        if (thr_floors || (skip && !skip_in_table)) thr_tab_fast = ctx->thr_tab;
    }
    if (fast_segs) {
        const uint32_t bx = (uint32_t)((fast_segs + WG_WAVES - 1) / WG_WAVES);
        const size_t lds = (size_t)pairs * 4;
        // the pass's tail (counts out, clears) rides in this launch when it is the only mask launch of the pass
        MaskFinish fin{};
        if (gop_tail && !ctx->no_fused_finish && fast_segs * 16 == nwords && !(((uintptr_t)gop_tail->clear_a | (uintptr_t)gop_tail->clear_b) & 15)) {
            if (!ctx->mask_ticket) {
                HIP_TRY(hipMalloc((void **)&ctx->mask_ticket, (MASK_TICKETS + 1) * 4));
                if (hipError_t e = hipMemsetAsync(ctx->mask_ticket, 0, (MASK_TICKETS + 1) * 4, ctx->stream)) {      // never keep tickets that were not zeroed
                    (void)hipFree(ctx->mask_ticket); ctx->mask_ticket = nullptr;
                    return fail(RBF_EIO, "hipMemsetAsync(mask tickets): %s", hipGetErrorString(e));
                }
            }
            fin = *gop_tail;
            fin.enabled = 1; fin.count = pairs; fin.ticket = ctx->mask_ticket; fin.ones_out = ones_dev;
            fused = true;
        }
        LaunchTimer t(ctx, RBF_K_MASK);
#define RBF_MASK_GOP(S, PB, Z) hipLaunchKernelGGL((k_residual_mask_gop<S, PB, true, Z>), dim3(bx, chunks), dim3(WG_THREADS), lds, ctx->stream,   \
                               (const uint8_t *)frames_dev, frame_stride_bytes, nframes, fast_segs, thr_floor, thr_tab_fast, (uint16_t *)masks_dev, \
                               mask_stride_bytes / 2, acc, mc, fin)
#define RBF_MASK_GOP2(S, PB) do { if (thr0) RBF_MASK_GOP(S, PB, true); else RBF_MASK_GOP(S, PB, false); } while (0)
        const bool thr0 = !thr_tab_fast && thr_floor == 0 && !(ctx->force_generic_mask_bits);     // "luma changed": no per-pixel extraction
        if (sample_bytes == 1 && pixel_stride_bytes == 1) RBF_MASK_GOP2(uint8_t, 1);
        else if (sample_bytes == 1) RBF_MASK_GOP2(uint8_t, 3);
        else if (pixel_stride_bytes == 2) RBF_MASK_GOP2(uint16_t, 2);
        else RBF_MASK_GOP2(uint16_t, 6);
#undef RBF_MASK_GOP2
#undef RBF_MASK_GOP
    }
    const uint64_t first_word = fast_segs * 16;
    if (first_word < nwords) {
        const uint64_t rest = nwords - first_word;
        uint64_t bx = (rest + WG_WAVES * 16 - 1) / (WG_WAVES * 16);      // ~16 words per wave
        if (bx < 1) bx = 1;
        if (bx > 65535) bx = 65535;
        dim3 grid((uint32_t)bx, pairs), block(WG_THREADS);
        LaunchTimer t(ctx, RBF_K_MASK);
        if (sample_bytes == 1)
            hipLaunchKernelGGL(k_residual_mask<uint8_t>, grid, block, 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                               width, n, row_pitch_bytes, pixel_stride_bytes, thr_floor, thr_tab, (uint64_t *)masks_dev, mask_stride_bytes / 8, acc, first_word);
        else
            hipLaunchKernelGGL(k_residual_mask<uint16_t>, grid, block, 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                               width, n, row_pitch_bytes, pixel_stride_bytes, thr_floor, thr_tab, (uint64_t *)masks_dev, mask_stride_bytes / 8, acc, first_word);
    }
    HIP_TRY(hipGetLastError());
    if (fused) { ctx->ones_acc_dirty = false; return RBF_OK; }
    if (gop_tail) return launch_finish_ones(ctx, ones_dev, pairs, gop_tail->host_block, gop_tail->token, gop_tail->clear_a, (size_t)gop_tail->quads_a * 16,
                                            gop_tail->clear_b, (size_t)gop_tail->quads_b * 16);
    if (finish) return launch_finish_ones(ctx, ones_dev, pairs, nullptr, 0, nullptr, 0, nullptr, 0);
    return RBF_OK;
}

int rbf_residual_mask_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                            uint32_t nframes, uint32_t width, uint32_t height,
                            uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                            uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                            void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev)
{
    return residual_mask_impl(ctx, frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes,
                              thr_floor, thr_floors, masks_dev, mask_stride_bytes, ones_dev, true);
}

// ------------------------------------------------------------------------------------------
// A4 + A5
// ------------------------------------------------------------------------------------------
static int check_filter_strides(const rbf_filter_params *params, uint32_t nframes, uint64_t filter_stride_bytes)
{
    if (filter_stride_bytes % 8) return fail(RBF_EINVAL, "filter stride must be a multiple of 8");
    for (uint32_t f = 0; f < nframes; ++f) {
        const uint64_t need = (((uint64_t)params[f].m + 63) / 64) * 8;
        if (filter_stride_bytes < need) return fail(RBF_EINVAL, "frame %u: filter stride %llu < %llu", f, (unsigned long long)filter_stride_bytes, (unsigned long long)need);
    }
    return RBF_OK;
}

// The scan in front of the compaction (encode) / expansion (decode): the start of every workgroup's range of witness bits, from the
// segment pass counts the query kernel left (k_chunk_offsets, one workgroup per frame).
static int launch_chunk_offsets(rbf_ctx *ctx, const Plan &pl, uint32_t nframes, uint32_t nchunks, void *witnesses_dev = nullptr /* encode: zero the shared dwords */,
                                uint64_t witness_stride_bytes = 0)
{
    if (WG_THREADS % pl.words_per_seg) return fail(RBF_EINVAL, "segments of %u words do not tile a workgroup's chunk", pl.words_per_seg);
    if (int r = grow((void **)&ctx->chunk_off, &ctx->chunk_off_cap, (size_t)nframes * nchunks * 4)) return r;
    LaunchTimer t(ctx, RBF_K_SCAN);
    hipLaunchKernelGGL(k_chunk_offsets, dim3(nframes), dim3(CO_THREADS), 0, ctx->stream, (const uint32_t *)ctx->seg_cnt, pl.nseg,
                       (uint32_t)WG_THREADS / pl.words_per_seg, nchunks, ctx->chunk_off, (uint32_t *)witnesses_dev, witness_stride_bytes / 4);
    return RBF_OK;
}

// query launch shared by encode and decode
static int ensure_image(rbf_ctx *ctx, const Plan &pl, uint32_t nframes)
{
    return grow((void **)&ctx->qimage, &ctx->qimage_cap, (size_t)nframes * pl.image_stride_words * 4);
}

// image_ready: the probe image of this batch has already been written (k_filter_reduce does it on the encode side)
static int launch_query(rbf_ctx *ctx, const Plan &pl, uint64_t n, uint32_t nframes, const FrameTable &tab, const Seeds &sd,
                        const void *filters_dev, uint64_t filter_stride_bytes, bool image_ready, bool table_for_next = false,
                        bool quiet_passthrough = false /* frames with m == 0 are another launch's: do not write their (empty) outputs */)
{
    if ((pl.query_kind == 1 || pl.query_kind == 3) && pl.f64_mod) {
        if (int r = ensure_image(ctx, pl, nframes)) return r;
        if (!image_ready) {
            uint32_t bx = (pl.image_stride_words + WG_THREADS - 1) / WG_THREADS;
            hipLaunchKernelGGL(k_probe_image, dim3(bx, nframes), dim3(WG_THREADS), 0, ctx->stream, (const uint32_t *)filters_dev,
                               filter_stride_bytes / 4, ctx->qimage, (uint64_t)pl.image_stride_words);
        }
        filters_dev = ctx->qimage;
        filter_stride_bytes = (uint64_t)pl.image_stride_words * 4;
    }
    if (pl.query_kind == 3) {
        const uint64_t bx = (pl.nseg + QL_WAVES - 1) / QL_WAVES;
        LaunchTimer t(ctx, RBF_K_QUERY);
        uint32_t nactive; uint64_t empty[2];
        const FrameTable stab = query_table_s64(tab, nframes, &nactive, empty);
        if (quiet_passthrough) empty[0] = empty[1] = 0;
        int mode = 0;                                               // 0: every coded frame has floor(k*) 1 or 2; 1: 0, 1 or 2; 2: anything (all frames walk their probes per tile)
        for (uint32_t f = 0; f < nframes; ++f) {
            if (!tab.f[f].m) continue;
            if (tab.f[f].floor_k > 2) mode = 2;
            else if (tab.f[f].floor_k == 0 && mode < 1) mode = 1;
        }
        auto qkern = mode == 2 ? k_query_s64t<2> : mode == 1 ? k_query_s64t<1> : k_query_s64t<0>;
        if (int r = allow_big_lds((const void *)qkern)) return r;
        hipLaunchKernelGGL(qkern, dim3((uint32_t)bx), dim3(QL_THREADS), s64t_lds_bytes(pl.query_tile_words), ctx->stream,
                           n, nactive, stab, sd, (const uint32_t *)filters_dev, filter_stride_bytes / 4, pl.query_tile_words,
                           ctx->seg_cnt, pl.nseg, ctx->pass_words, empty[0], empty[1]);
    } else if (pl.query_kind == 1 && pl.f64_mod) {
        const uint64_t bx = (pl.nseg + QL_WAVES - 1) / QL_WAVES;
        LaunchTimer t(ctx, RBF_K_QUERY);
        // A context that is the pixel-index hash table's only holder has the kernel -- which hashes every index anyway -- write it again:
        // 54 MB of identical values whose only purpose is to be in the Infinity Cache when the next batch's insert gathers from them (one
        // pipeline: insert 47 -> 38 us, step 214 -> 209).  With several holders the table stays cached by being used.  (READING the hashes
        // from the table instead of computing them was measured in round 4: 73.2 instead of 74.7 us alone with the 32-byte entries of
        // that time, nothing in the step, 66 MB of extra traffic per launch -- not kept.)
        uint4 *table_out = nullptr;
        const SharedHashTable *sh = ctx->hash_shared;
        if (table_for_next && sh && sh->n == n && sh->seeds.h1 == sd.h1 && sh->seeds.h2 == sd.h2 && sh->seeds.act == sd.act && !ctx->no_hash_table) {
            bool sole;
            { std::lock_guard<std::mutex> lk(g_hash_mu); sole = sh->refs == 1; }
            if (sole && !ctx->no_table_rewrite) table_out = ctx->hash_tab;
        }
        // k_query_u64: coded frames ordered by floor(k*), 32-byte frame records in LDS behind the two image buffers
        uint32_t nactive; uint64_t empty[2]; U64Classes cls;
        const FrameTable utab = query_table_u64(tab, nframes, &nactive, &cls, empty);
        if (quiet_passthrough) empty[0] = empty[1] = 0;
        // the 111-register kernel (two waves of a neighbour pipeline's mask / compaction kernels fit next to it on every SIMD) unless the
        // batch has floor(k*) = 4 or 5, which only the 118-register one passes in rows
        const bool wide = cls.n[3] + cls.n[4] > 0;                 // (always the wide one: measured, no better -- profiles/r04_feed_sweep2.txt)
        auto kern64 = wide ? k_query_u64w : k_query_u64;
        if (int r = allow_big_lds((const void *)kern64)) return r;
        hipLaunchKernelGGL(kern64, dim3((uint32_t)bx), dim3(QL_THREADS), pl.query_lds_bytes + u64_geo_bytes(nactive), ctx->stream,
                           n, nactive, utab, cls, sd, (const uint32_t *)filters_dev, filter_stride_bytes / 4, pl.fwords_max,
                           ctx->seg_cnt, pl.nseg, ctx->pass_words, table_out, empty[0], empty[1]);
    } else if (pl.query_kind == 1) {
        auto kern = pl.double_buffer ? (pl.small_m ? k_query_lds<true, true> : k_query_lds<true, false>)
                                     : (pl.small_m ? k_query_lds<false, true> : k_query_lds<false, false>);
        if (int r = allow_big_lds((const void *)kern)) return r;
        const uint64_t bx = (pl.nseg + QL_WAVES - 1) / QL_WAVES;
        LaunchTimer t(ctx, RBF_K_QUERY);
        hipLaunchKernelGGL(kern, dim3((uint32_t)bx), dim3(QL_THREADS), pl.query_lds_bytes, ctx->stream,
                           n, nframes, tab, sd, (const uint32_t *)filters_dev, filter_stride_bytes / 4, pl.fwords_max,
                           ctx->seg_cnt, pl.nseg, ctx->pass_words);
    } else if (pl.query_kind == 2) {
        auto kern = pl.small_m ? k_query_tiled<true> : k_query_tiled<false>;
        if (int r = allow_big_lds((const void *)kern)) return r;
        const uint64_t bx = (pl.nseg + QL_WAVES - 1) / QL_WAVES;
        LaunchTimer t(ctx, RBF_K_QUERY);
        hipLaunchKernelGGL(kern, dim3((uint32_t)bx), dim3(QL_THREADS), pl.query_lds_bytes, ctx->stream,
                           n, nframes, tab, sd, (const uint32_t *)filters_dev, filter_stride_bytes / 4, pl.query_tile_words,
                           ctx->seg_cnt, pl.nseg, ctx->pass_words);
    } else {
        const uint64_t bx = (pl.nseg + WG_WAVES - 1) / WG_WAVES;
        LaunchTimer t(ctx, RBF_K_QUERY);
        hipLaunchKernelGGL(k_query, dim3((uint32_t)bx, nframes), dim3(WG_THREADS), 0, ctx->stream,
                           n, tab, sd, (const uint32_t *)filters_dev, filter_stride_bytes / 4, ctx->seg_cnt, pl.nseg, ctx->pass_words);
    }
    return RBF_OK;
}

// one chunk of at most MAX_BATCH frames (the geometry table rides in the kernel arguments)
// `quiet_passthrough` / `compact`: see encode_chunk (a batch split over the two kernel families runs this twice)
static int encode_chunk_pass(rbf_ctx *ctx, const void *masks_dev, uint64_t mask_stride_bytes,
                             uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                             const rbf_seeds *seeds,
                             void *filters_dev, uint64_t filter_stride_bytes,
                             void *witnesses_dev, uint64_t witness_stride_bytes,
                             uint64_t *stats_dev, bool outputs_zeroed, const uint64_t *ones_host /* nullable: set bits of every mask */,
                             bool quiet_passthrough, bool compact)
{
    uint64_t nrecords = 0;
    if (ones_host) for (uint32_t f = 0; f < nframes; ++f) if (params[f].m) nrecords += ones_host[f];
    Plan pl = make_plan(ctx, params, nframes, n, ones_host && nrecords < (1ull << 32));
    // The two-kernel insert needs 8 bytes per set mask bit.  Without that memory the batch is re-planned as if the counts
    // were unknown (queue-sized tiles, k_insert_tab / k_insert_lds): slower, not an error.
    if (pl.insert_two_phase &&
        (grow((void **)&ctx->ins_records, &ctx->ins_records_cap, (size_t)(nrecords ? nrecords : 1) * 8) ||
         grow((void **)&ctx->ins_counters, &ctx->ins_counters_cap, (size_t)MAX_BATCH * 4))) {
        (void)hipGetLastError();
        pl = make_plan(ctx, params, nframes, n, false);
    }
    FrameTable tab;
    if (int r = fill_table(params, nframes, &tab)) return r;
    uint32_t coded = 0;
    for (uint32_t f = 0; f < nframes; ++f) coded += params[f].m ? 1u : 0u;
    const bool stream_once = (uint64_t)coded * n >= STREAM_MIN_PIXEL_FRAMES;      // a block of several GOPs (or of large frames): its one-shot data must not evict the hash table and the probe images
    if (int r = grow((void **)&ctx->pass_words, &ctx->pass_words_cap, (size_t)nframes * pl.nseg * pl.words_per_seg * 8)) return r;
    if (int r = grow((void **)&ctx->seg_cnt, &ctx->seg_cnt_cap, (size_t)nframes * pl.nseg * 4 + 16)) return r;     // + 16: k_compact_witness reads the counts four to a load
    if (int r = grow((void **)&ctx->seg_off, &ctx->seg_off_cap, (size_t)nframes * pl.nseg * 8)) return r;
    const Seeds sd{seeds->h1, seeds->h2, seeds->act};
    const bool want_image = (pl.query_kind == 1 || pl.query_kind == 3) && pl.f64_mod;       // the reduce kernel also writes the FP64 query kernel's probe image
    if (want_image) if (int r = ensure_image(ctx, pl, nframes)) return r;
    uint32_t *image = want_image ? ctx->qimage : nullptr;

    if (!outputs_zeroed)                                           // (the witness rows need no clearing: k_chunk_offsets zeroes what the compaction shares)
        HIP_TRY(hipMemsetAsync(stats_dev, 0, (size_t)nframes * RBF_STATS_PER_FRAME * 8, ctx->stream));
    // ---- insert
    if (pl.fast_insert) {
        const uint64_t part_stride = (pl.fwords_max + 3u) & ~3ull;      // 16-byte rows for the reduce kernel
        if (int r = grow((void **)&ctx->partials, &ctx->partials_cap, (size_t)nframes * pl.S * part_stride * 4)) return r;
        // hash table of the pixel indices (k_hash_table): built for this batch, or kept from the last one when the
        // context was told to cache it; without device memory for it the insert kernel hashes for itself
        bool use_tab = pl.insert_tab;
        // A pixel-index table (an allocation of 32 B per pixel, process-wide, lives until the last context of its geometry goes) that would crowd the
        // 256 MB Infinity Cache is never built: the insert kernels hash their set positions on the spot instead (2160p, 265 MB:
        // 72 us against 96 with the gather) -- whichever insert kernel runs, with or without the masks' set-bit counts.
        const bool table_too_big = ((size_t)n + QL_SEG_PIXELS) * 32 > HASH_TABLE_CACHE_BYTES;
        bool hashed_positions = use_tab && (table_too_big || (pl.insert_two_phase && ctx->hash_positions));
        if (use_tab && hashed_positions && ctx->hash_shared &&
            (ctx->hash_shared->n != n || ctx->hash_shared->seeds.h1 != seeds->h1 || ctx->hash_shared->seeds.h2 != seeds->h2 || ctx->hash_shared->seeds.act != seeds->act))
            hash_table_release(ctx);                              // this context moved to a geometry that hashes: it no longer pins the old geometry's table
        if (use_tab && !hashed_positions) {
            bool built = false;
            if (!hash_table_acquire(ctx, n, *seeds, &built)) hashed_positions = true;        // no device memory for the table: hash instead
            else if (ctx->hash_rebuild && !built) {               // diagnostic: the table is rewritten (same values) for every batch
                const uint64_t segs = (n + QL_SEG_PIXELS - 1) / QL_SEG_PIXELS;
                LaunchTimer t(ctx, RBF_K_HASHTAB);
                hipLaunchKernelGGL(k_hash_table, dim3((uint32_t)((segs + HT_THREADS / WAVE - 1) / (HT_THREADS / WAVE))), dim3(HT_THREADS), 0, ctx->stream,
                                   n, sd, ctx->hash_tab);
            }
        }
        FrameTable itab = tab;                                     // k_insert_tab reads -1/m from the M field
        if (use_tab)
            for (uint32_t f = 0; f < nframes; ++f)
                if (itab.f[f].m) { const double ninv = -1.0 / (double)itab.f[f].m; memcpy(&itab.f[f].M, &ninv, 8); }
        auto ikern = pl.small_m ? k_insert_lds<true> : k_insert_lds<false>;
        if (int r = allow_big_lds((const void *)ikern)) return r;
        // one tile = the whole filter: the kernel without the in-tile test per probe
        const bool whole = pl.insert_tiles == 1;
        auto tkern = hashed_positions ? (whole ? k_insert_tab<true, true> : k_insert_tab<true, false>) : (whole ? k_insert_tab<false, true> : k_insert_tab<false, false>);
        if (int r = allow_big_lds((const void *)tkern)) return r;
        const bool two_phase = pl.insert_two_phase && use_tab;    // (its record memory was reserved above)
        if (two_phase) {
            // itab.floor_k / rtab.T carry the index of the frame's first record (the kernels' own use of those fields: none)
            uint64_t first = 0;
            for (uint32_t f = 0; f < nframes; ++f) {
                itab.f[f].floor_k = (uint32_t)first;
                if (params[f].m) first += ones_host[f];
            }
            HIP_TRY(hipMemsetAsync(ctx->ins_counters, 0, (size_t)nframes * 4, ctx->stream));
            const uint64_t groups = (((n + 7) >> 3) + IT_STEP_BYTES - 1) / IT_STEP_BYTES;
            uint64_t S1 = (uint64_t)ctx->cus * 4 / (nframes ? nframes : 1);           // ~4 workgroups of 4 waves per CU (2160p x 8: 96 us; 8 per CU: 115) ...
            if (S1 > groups / (IP_WAVES * 4)) S1 = groups / (IP_WAVES * 4);           // ... each wave with >= 4 steps
            S1 &= ~7ull;                                                              // a slice stays on one XCD across frames
            if (S1 < 1) S1 = 1;
            if (int r = allow_big_lds((const void *)k_insert_records)) return r;
            LaunchTimer t(ctx, RBF_K_INSERT);
            if (hashed_positions)
                hipLaunchKernelGGL((k_insert_positions<true>), dim3((uint32_t)S1, nframes), dim3(IP_THREADS), 0, ctx->stream,
                                   (const uint8_t *)masks_dev, mask_stride_bytes, n, itab, (const uint4 *)nullptr, sd, ctx->ins_records, ctx->ins_counters);
            else
                hipLaunchKernelGGL((k_insert_positions<false>), dim3((uint32_t)S1, nframes), dim3(IP_THREADS), 0, ctx->stream,
                                   (const uint8_t *)masks_dev, mask_stride_bytes, n, itab, (const uint4 *)ctx->hash_tab, sd, ctx->ins_records, ctx->ins_counters);
        }
        for (uint32_t f0 = 0; f0 < nframes;) {                    // groups of pl.insert_group coded frames
            SliceTable grp{};
            uint32_t per_tile = 0, coded = 0, f = f0;
            for (; f < nframes && coded < pl.insert_group; ++f) {
                grp.n[f] = pl.slices.n[f];
                per_tile += grp.n[f];
                coded += grp.n[f] ? 1u : 0u;
            }
            f0 = f;
            if (!per_tile) continue;
            LaunchTimer t(ctx, RBF_K_INSERT);
            if (two_phase) {
                FrameTable rtab = tab;
                uint64_t first = 0;
                for (uint32_t g = 0; g < nframes; ++g) { rtab.f[g].T = first; if (params[g].m) first += ones_host[g]; }
                hipLaunchKernelGGL(k_insert_records, dim3(per_tile * pl.insert_tiles), dim3(IL_THREADS), pl.insert_lds_bytes, ctx->stream,
                                   (const uint2 *)ctx->ins_records, (const uint32_t *)ctx->ins_counters, rtab, ctx->partials, part_stride,
                                   pl.insert_tile_words, grp, per_tile, pl.S);
            } else if (use_tab)
                hipLaunchKernelGGL(tkern, dim3(per_tile * pl.insert_tiles), dim3(IL_THREADS), pl.insert_lds_bytes, ctx->stream,
                                   (const uint8_t *)masks_dev, mask_stride_bytes, n, itab, hashed_positions ? (const uint4 *)nullptr : (const uint4 *)ctx->hash_tab, sd,
                                   ctx->partials, part_stride, pl.insert_tile_words, grp, per_tile, pl.S);
            else
                hipLaunchKernelGGL(ikern, dim3(per_tile * pl.insert_tiles), dim3(IL_THREADS), pl.insert_lds_bytes, ctx->stream,
                                   (const uint8_t *)masks_dev, mask_stride_bytes, n, tab, sd, ctx->partials, part_stride, pl.insert_tile_words,
                                   grp, per_tile, pl.S);
        }
        {
            LaunchTimer t(ctx, RBF_K_REDUCE);
            const uint64_t words = filter_stride_bytes / 4;
            const uint32_t vec_ok = (words % 4 == 0 && ((uintptr_t)filters_dev % 16) == 0) ? 1u : 0u;
            uint32_t bx = (uint32_t)((words + WG_THREADS * 4 - 1) / (WG_THREADS * 4));
            if (bx < 1) bx = 1;
            hipLaunchKernelGGL(stream_once ? k_filter_reduce<true> : k_filter_reduce<false>, dim3(bx, nframes), dim3(WG_THREADS), 0, ctx->stream,
                               (const uint32_t *)ctx->partials, part_stride, pl.S, pl.slices, tab, (uint32_t *)filters_dev, words, stats_dev, vec_ok,
                               image, (uint64_t)pl.image_stride_words);
        }
    } else {
        HIP_TRY(hipMemsetAsync(filters_dev, 0, (size_t)nframes * filter_stride_bytes, ctx->stream));
        const uint64_t nwords32 = (n + 31) / 32;
        uint64_t bx = (nwords32 + WG_THREADS - 1) / WG_THREADS;
        if (bx > 65535) bx = 65535;
        {
            LaunchTimer t(ctx, RBF_K_INSERT);
            hipLaunchKernelGGL(k_insert, dim3((uint32_t)bx, nframes), dim3(WG_THREADS), 0, ctx->stream,
                               (const uint32_t *)masks_dev, mask_stride_bytes / 4, n, tab, sd,
                               (uint32_t *)filters_dev, filter_stride_bytes / 4);
        }
        {
            LaunchTimer t(ctx, RBF_K_REDUCE);        // S = 1 in place: only counts the set bits
            SliceTable ones;
            memset(ones.n, 1, sizeof ones.n);
            const uint64_t words = filter_stride_bytes / 4;
            uint32_t bx2 = (uint32_t)((words + WG_THREADS * 4 - 1) / (WG_THREADS * 4));
            if (bx2 < 1) bx2 = 1;
            hipLaunchKernelGGL(k_filter_reduce<false>, dim3(bx2, nframes), dim3(WG_THREADS), 0, ctx->stream,
                               (const uint32_t *)filters_dev, words, 1u, ones, tab, (uint32_t *)filters_dev, words, stats_dev, 0u,
                               image, (uint64_t)pl.image_stride_words);
        }
    }
    // ---- query: pass word of every 64 positions + per-segment pass counts
    if (int r = launch_query(ctx, pl, n, nframes, tab, sd, filters_dev, filter_stride_bytes, want_image, pl.insert_tab, quiet_passthrough)) return r;
    // ---- witness: pext(mask, pass) of every word lands at its bit offset (scan fused in)
    if (compact) {
        const uint64_t words = pl.nseg * pl.words_per_seg;
        uint64_t bx = (words + WG_THREADS - 1) / WG_THREADS;
        if (bx < 1) bx = 1;
        if (int r = launch_chunk_offsets(ctx, pl, nframes, (uint32_t)bx, witnesses_dev, witness_stride_bytes)) return r;
        LaunchTimer t(ctx, RBF_K_STITCH);
        hipLaunchKernelGGL(stream_once ? k_compact_witness<true> : k_compact_witness<false>, dim3((uint32_t)bx, nframes), dim3(WG_THREADS), 0, ctx->stream,
                           ctx->pass_words, ctx->seg_cnt, pl.nseg, pl.words_per_seg,
                           (const uint64_t *)masks_dev, mask_stride_bytes / 8, n, (uint32_t *)witnesses_dev, witness_stride_bytes / 4, stats_dev, ctx->chunk_off);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

// One chunk of at most MAX_BATCH frames.  The FP64 kernels (hash-table insert, k_query_u64 / k_query_s64t) need EVERY coded filter of
// their launch inside F64MOD_M_MIN <= m <= F64MOD_M_MAX; a single nearly static frame (1080p: < ~0.2 % changed pixels) used to
// send its whole batch to the round-1 Barrett kernels.  A mixed batch is now coded in two passes over disjoint frame sets --
// first the out-of-range frames (Barrett kernels; they also write the empty outputs of every frame that is not theirs), then
// the in-range ones (FP64 kernels, told to leave the others' outputs alone) -- followed by one compaction over all frames.
// Only when both passes cut the frame into the same segments (pass bytes and segment counts are shared with the compaction).
static int encode_chunk(rbf_ctx *ctx, const void *masks_dev, uint64_t mask_stride_bytes,
                        uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                        const rbf_seeds *seeds,
                        void *filters_dev, uint64_t filter_stride_bytes,
                        void *witnesses_dev, uint64_t witness_stride_bytes,
                        uint64_t *stats_dev, bool outputs_zeroed, const uint64_t *ones_host)
{
    uint32_t in_range = 0, out_of_range = 0;
    for (uint32_t f = 0; f < nframes; ++f)
        if (params[f].m) ++((params[f].m >= F64MOD_M_MIN && params[f].m <= F64MOD_M_MAX) ? in_range : out_of_range);
    if (in_range && out_of_range && !ctx->force_generic && !ctx->barrett_only && !ctx->no_hash_table && !ctx->single_buffer) {
        rbf_filter_params small[MAX_BATCH], big[MAX_BATCH];
        for (uint32_t f = 0; f < nframes; ++f) {
            small[f] = big[f] = params[f];
            const bool fp64 = params[f].m >= F64MOD_M_MIN && params[f].m <= F64MOD_M_MAX;
            (fp64 ? small[f] : big[f]).m = 0;
        }
        const Plan ps = make_plan(ctx, small, nframes, n, false), pb = make_plan(ctx, big, nframes, n, ones_host != nullptr);
        const bool fp64_query = (pb.query_kind == 1 || pb.query_kind == 3) && pb.f64_mod;
        if (fp64_query && pb.insert_tab && ps.nseg == pb.nseg && ps.words_per_seg == pb.words_per_seg) {
            if (int r = encode_chunk_pass(ctx, masks_dev, mask_stride_bytes, n, nframes, small, seeds, filters_dev, filter_stride_bytes,
                                          witnesses_dev, witness_stride_bytes, stats_dev, outputs_zeroed, nullptr, false, false))
                return r;
            return encode_chunk_pass(ctx, masks_dev, mask_stride_bytes, n, nframes, big, seeds, filters_dev, filter_stride_bytes,
                                     witnesses_dev, witness_stride_bytes, stats_dev, true, ones_host, true, true);
        }
    }
    return encode_chunk_pass(ctx, masks_dev, mask_stride_bytes, n, nframes, params, seeds, filters_dev, filter_stride_bytes,
                             witnesses_dev, witness_stride_bytes, stats_dev, outputs_zeroed, ones_host, false, true);
}

static int encode_batch_impl(rbf_ctx *ctx, const void *masks_dev, uint64_t mask_stride_bytes,
                             uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                             const rbf_seeds *seeds,
                             void *filters_dev, uint64_t filter_stride_bytes,
                             void *witnesses_dev, uint64_t witness_stride_bytes,
                             uint64_t *stats_dev, bool outputs_zeroed, const uint64_t *ones_host = nullptr)
{
    if (int r = set_device(ctx)) return r;
    if (!masks_dev || !params || !seeds || !filters_dev || !witnesses_dev || !stats_dev) return fail(RBF_EINVAL, "null pointer");
    if (int r = check_frame_geometry(n, nframes, mask_stride_bytes)) return r;
    if (int r = check_filter_strides(params, nframes, filter_stride_bytes)) return r;
    if (witness_stride_bytes % 8 || witness_stride_bytes < ((n + 63) / 64) * 8) return fail(RBF_EINVAL, "witness stride too small or misaligned");
    for (uint32_t f0 = 0; f0 < nframes; f0 += MAX_BATCH) {
        const uint32_t cnt = nframes - f0 < (uint32_t)MAX_BATCH ? nframes - f0 : (uint32_t)MAX_BATCH;
        if (int r = encode_chunk(ctx, (const uint8_t *)masks_dev + (uint64_t)f0 * mask_stride_bytes, mask_stride_bytes, n, cnt, params + f0, seeds,
                                 (uint8_t *)filters_dev + (uint64_t)f0 * filter_stride_bytes, filter_stride_bytes,
                                 (uint8_t *)witnesses_dev + (uint64_t)f0 * witness_stride_bytes, witness_stride_bytes,
                                 stats_dev + (uint64_t)f0 * RBF_STATS_PER_FRAME, outputs_zeroed, ones_host ? ones_host + f0 : nullptr))
            return r;
    }
    return RBF_OK;
}

int rbf_bloom_encode_batch(rbf_ctx *ctx, const void *masks_dev, uint64_t mask_stride_bytes,
                           uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                           const rbf_seeds *seeds,
                           void *filters_dev, uint64_t filter_stride_bytes,
                           void *witnesses_dev, uint64_t witness_stride_bytes,
                           uint64_t *stats_dev)
{
    return encode_batch_impl(ctx, masks_dev, mask_stride_bytes, n, nframes, params, seeds, filters_dev, filter_stride_bytes,
                             witnesses_dev, witness_stride_bytes, stats_dev, false);
}

// ------------------------------------------------------------------------------------------
// A1, adaptive threshold: 5x5 median residual and its exact moments
// ------------------------------------------------------------------------------------------
int rbf_noise_moments_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                            uint32_t nframes, uint32_t width, uint32_t height,
                            uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                            uint32_t sample_bytes, int64_t *moments_dev, float *noise_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!frames_dev || !moments_dev) return fail(RBF_EINVAL, "null device pointer");
    if (nframes == 0 || nframes > 65535) return fail(RBF_EINVAL, "frame count %u out of range 1..65535", nframes);
    if (width == 0 || height == 0) return fail(RBF_EINVAL, "empty frame %ux%u", width, height);
    if (sample_bytes != 1 && sample_bytes != 2) return fail(RBF_EINVAL, "sample_bytes must be 1 or 2, got %u", sample_bytes);
    if (pixel_stride_bytes < sample_bytes || pixel_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "pixel stride %u incompatible with %u-byte samples", pixel_stride_bytes, sample_bytes);
    if (row_pitch_bytes < (uint64_t)width * pixel_stride_bytes || row_pitch_bytes % sample_bytes) return fail(RBF_EINVAL, "row pitch %llu too small or misaligned", (unsigned long long)row_pitch_bytes);
    if (frame_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "frame stride misaligned");
    if ((uint64_t)width * height >= (1ull << 32)) return fail(RBF_ERANGE, "frame of %llu pixels is too large", (unsigned long long)width * height);
    const uint32_t by = (height + NZ_TILE_H - 1) / NZ_TILE_H;
    if (by > 65535) return fail(RBF_ERANGE, "frame height %u too large", height);
    HIP_TRY(hipMemsetAsync(moments_dev, 0, (size_t)nframes * 2 * sizeof(int64_t), ctx->stream));
    dim3 grid((width + NZ_TILE_W - 1) / NZ_TILE_W, by, nframes), block(NZ_THREADS);
    LaunchTimer t(ctx, RBF_K_NOISE);
    if (sample_bytes == 1)
        hipLaunchKernelGGL(k_noise_moments<uint8_t>, grid, block, 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                           width, height, row_pitch_bytes, pixel_stride_bytes, (unsigned long long *)moments_dev, noise_dev);
    else
        hipLaunchKernelGGL(k_noise_moments<uint16_t>, grid, block, 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                           width, height, row_pitch_bytes, pixel_stride_bytes, (unsigned long long *)moments_dev, noise_dev);
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_bgr_to_gray_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes,
                          uint32_t width, uint32_t height, uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                          uint32_t sample_bytes, void *gray_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!frames_dev || !gray_dev) return fail(RBF_EINVAL, "null device pointer");
    if (nframes == 0 || nframes > 65535) return fail(RBF_EINVAL, "frame count %u out of range 1..65535", nframes);
    if (width == 0 || height == 0) return fail(RBF_EINVAL, "empty frame %ux%u", width, height);
    if (sample_bytes != 1 && sample_bytes != 2) return fail(RBF_EINVAL, "sample_bytes must be 1 or 2, got %u", sample_bytes);
    if (pixel_stride_bytes < 3 * sample_bytes || pixel_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "pixel stride %u holds fewer than 3 samples of %u bytes", pixel_stride_bytes, sample_bytes);
    if (row_pitch_bytes < (uint64_t)width * pixel_stride_bytes || row_pitch_bytes % sample_bytes) return fail(RBF_EINVAL, "row pitch too small or misaligned");
    if (frame_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "frame stride misaligned");
    const uint64_t n = (uint64_t)width * height;
    uint64_t bx = (n + 256 * 4 - 1) / (256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 8192) bx = 8192;
    LaunchTimer t(ctx, RBF_K_MASK);
    if (sample_bytes == 1)
        hipLaunchKernelGGL(k_bgr_to_gray<uint8_t>, dim3((uint32_t)bx, nframes), dim3(256), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                           width, n, row_pitch_bytes, pixel_stride_bytes, (uint8_t *)gray_dev);
    else
        hipLaunchKernelGGL(k_bgr_to_gray<uint16_t>, dim3((uint32_t)bx, nframes), dim3(256), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                           width, n, row_pitch_bytes, pixel_stride_bytes, (uint16_t *)gray_dev);
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_extract_luma_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes,
                           uint32_t width, uint32_t height, uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                           uint32_t sample_bytes, void *luma_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!frames_dev || !luma_dev) return fail(RBF_EINVAL, "null device pointer");
    if (nframes == 0 || nframes > 65535) return fail(RBF_EINVAL, "frame count %u out of range 1..65535", nframes);
    if (width == 0 || height == 0) return fail(RBF_EINVAL, "empty frame %ux%u", width, height);
    if (sample_bytes != 1 && sample_bytes != 2) return fail(RBF_EINVAL, "sample_bytes must be 1 or 2, got %u", sample_bytes);
    if (pixel_stride_bytes < sample_bytes || pixel_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "pixel stride %u incompatible with %u-byte samples", pixel_stride_bytes, sample_bytes);
    if (row_pitch_bytes < (uint64_t)width * pixel_stride_bytes || row_pitch_bytes % sample_bytes) return fail(RBF_EINVAL, "row pitch too small or misaligned");
    if (frame_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "frame stride misaligned");
    const uint64_t n = (uint64_t)width * height;
    uint64_t bx = (n + 256 * 4 - 1) / (256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 8192) bx = 8192;
    if (sample_bytes == 1)                                        // (untimed: an upload-time pass, not a kernel of the step)
        hipLaunchKernelGGL(k_extract_luma<uint8_t>, dim3((uint32_t)bx, nframes), dim3(256), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                           width, n, row_pitch_bytes, pixel_stride_bytes, (uint8_t *)luma_dev);
    else
        hipLaunchKernelGGL(k_extract_luma<uint16_t>, dim3((uint32_t)bx, nframes), dim3(256), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                           width, n, row_pitch_bytes, pixel_stride_bytes, (uint16_t *)luma_dev);
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

// The two halves of rbf_encode_gop (`..._masks()` -> host -> `..._blooms()`).  begin: every check, then the mask stage is
// enqueued and the call returns; finish: wait for the counts the mask kernel's last workgroup publishes into pinned host memory, the
// float64 parameter math, then insert / reduce / query / compaction are enqueued.  One GOP per context is between the two at a time;
// a caller with several contexts issues begin(k + 1) before finish(k), so that its thread never stands still while a mask kernel runs.
uint64_t rbf_filter_stride_min(uint64_t n)
{
    // l = int(p n k / ln 2) with k = log2((1 - p) ln(2)^2 / p) peaks at p = 0.13183 with l = 0.316053 n (improved_video_compressor.py:181-193)
    const uint64_t lmax = (uint64_t)(0.3161 * (double)n) + 2;
    return (lmax + 63) / 64 * 8;
}

int rbf_encode_runs_begin(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                          uint32_t nframes, uint32_t width, uint32_t height,
                          uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                          uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                          const uint8_t *run_starts, const rbf_seeds *seeds,
                          void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                          void *filters_dev, uint64_t filter_stride_bytes,
                          void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev)
{
    if (!ctx) return fail(RBF_EINVAL, "null context");
    if (ctx->gop.active) return fail(RBF_EINVAL, "rbf_encode_runs_begin / rbf_encode_gop_begin: the previous block of this context has not been finished (rbf_encode_gop_finish)");
    if (!filters_dev || !witnesses_dev || !stats_dev || !seeds) return fail(RBF_EINVAL, "null pointer");
    if (int r = set_device(ctx)) return r;
    // nothing below this block has run, and nothing of the caller's has been touched, when an argument is bad
    if (int r = check_mask_args(frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes,
                                thr_floors, masks_dev, mask_stride_bytes, ones_dev)) return r;
    const uint32_t pairs = nframes - 1;
    const uint64_t n = (uint64_t)width * height;
    if (witness_stride_bytes % 8 || witness_stride_bytes < ((n + 63) / 64) * 8) return fail(RBF_EINVAL, "witness stride too small or misaligned");
    if (filter_stride_bytes % 8) return fail(RBF_EINVAL, "filter stride must be a multiple of 8");
    // the filters are planned in the second half, from the counts: the stride has to cover whatever the planner can produce for n pixels
    if (filter_stride_bytes < rbf_filter_stride_min(n))
        return fail(RBF_EINVAL, "rbf_encode_runs_begin / rbf_encode_gop_begin: filter stride %llu < rbf_filter_stride_min(%llu) = %llu", (unsigned long long)filter_stride_bytes,
                    (unsigned long long)n, (unsigned long long)rbf_filter_stride_min(n));
    bool has_skip = false;
    if (run_starts) {
        try { ctx->run_skip.assign(pairs, 0); } catch (...) { return fail(RBF_ENOMEM, "out of host memory for %u pairs", pairs); }
        for (uint32_t p2 = 0; p2 < pairs; ++p2)
            if (run_starts[p2 + 1]) { ctx->run_skip[p2] = 1; has_skip = true; }
    }
    if (pairs > ctx->host_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->ones_pinned) HIP_TRY(hipHostFree(ctx->ones_pinned));
        ctx->ones_pinned = nullptr; ctx->host_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&ctx->ones_pinned, (size_t)(pairs + 17) * sizeof(uint64_t), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&ctx->ones_mapped_dev, ctx->ones_pinned, 0));
        ctx->ones_pinned[0] = 0;
        try {                                                    // the C ABI never throws
            ctx->plan.resize(pairs + 16);
            ctx->plan_k.resize(pairs + 16);
        } catch (...) {
            return fail(RBF_ENOMEM, "out of host memory for %u frame plans", pairs);
        }
        ctx->host_cap = pairs + 16;
    }
    // The GPU publishes the counts straight into host memory and clears the stats rows in the same pass -- inside the mask
    // kernel when it covers the whole frame, else through k_finish_ones -- so the only thing between the mask kernel and the
    // Bloom kernels is the host's float64 parameter math.  (The witness rows are not cleared any more: k_chunk_offsets zeroes the dwords
    // the compaction's workgroups share, the compaction writes everything else.)
    const uint64_t token = ++ctx->publish_token;
    MaskFinish tail{};
    tail.host_block = ctx->ones_mapped_dev; tail.token = token;
    tail.clear_a = nullptr; tail.quads_a = 0;
    tail.clear_b = (uint4 *)stats_dev;     tail.quads_b = (uint64_t)pairs * RBF_STATS_PER_FRAME * 8 / 16;
    if (int r = residual_mask_impl(ctx, frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes,
                                   pixel_stride_bytes, sample_bytes, thr_floor, thr_floors, masks_dev, mask_stride_bytes, ones_dev, false, &tail,
                                   has_skip ? ctx->run_skip.data() : nullptr))
        return r;
    rbf_ctx::PendingGop &g = ctx->gop;
    g.active = true; g.token = token; g.n = n; g.pairs = pairs; g.seeds = *seeds; g.has_skip = has_skip;
    g.masks_dev = masks_dev; g.mask_stride_bytes = mask_stride_bytes;
    g.filters_dev = filters_dev; g.filter_stride_bytes = filter_stride_bytes;
    g.witnesses_dev = witnesses_dev; g.witness_stride_bytes = witness_stride_bytes; g.stats_dev = stats_dev;
    return RBF_OK;
}

int rbf_encode_gop_poll(rbf_ctx *ctx, int *ready)
{
    if (!ctx || !ready) return fail(RBF_EINVAL, "null pointer");
    if (!ctx->gop.active) return fail(RBF_EINVAL, "rbf_encode_gop_poll: no GOP has been begun on this context");
    *ready = __atomic_load_n((volatile uint64_t *)ctx->ones_pinned, __ATOMIC_ACQUIRE) == ctx->gop.token ? 1 : 0;
    return RBF_OK;
}

int rbf_encode_gop_finish(rbf_ctx *ctx, rbf_filter_params *params_out, double *k_out)
{
    if (!ctx) return fail(RBF_EINVAL, "null context");
    if (!ctx->gop.active) return fail(RBF_EINVAL, "rbf_encode_gop_finish: no GOP has been begun on this context");
    if (int r = set_device(ctx)) return r;
    const rbf_ctx::PendingGop g = ctx->gop;
    ctx->gop.active = false;                                      // whatever happens below, the context is free for the next begin
    volatile uint64_t *flag = ctx->ones_pinned;
    for (uint64_t spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != g.token; ++spins) {
        if ((spins & 0xFFFF) == 0xFFFF) {                    // every ~65k polls make sure the stream is still alive
            hipError_t q = hipStreamQuery(ctx->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(RBF_EIO, "stream failed while waiting for the mask kernel: %s", hipGetErrorString(q));
            if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != g.token)
                return fail(RBF_EIO, "mask kernel finished without publishing its counts");
        }
        __builtin_ia32_pause();
    }
    if (g.has_skip)                                               // (zero by construction: a skipped pair's row is written as zeros and never counted)
        for (uint32_t p = 0; p < g.pairs; ++p) if (ctx->run_skip[p]) ctx->ones_pinned[1 + p] = 0;
    if (int r = rbf_plan_batch(g.n, ctx->ones_pinned + 1, g.pairs, 1, ctx->plan.data(), ctx->plan_k.data())) return r;
    if (params_out) {
        memcpy(params_out, ctx->plan.data(), (size_t)g.pairs * sizeof(rbf_filter_params));
        if (g.has_skip) for (uint32_t p = 0; p < g.pairs; ++p) if (ctx->run_skip[p]) params_out[p].floor_k = RBF_PAIR_SKIPPED;
    }
    if (k_out) memcpy(k_out, ctx->plan_k.data(), (size_t)g.pairs * sizeof(double));
    const int rc = encode_batch_impl(ctx, g.masks_dev, g.mask_stride_bytes, g.n, g.pairs, ctx->plan.data(), &g.seeds,
                                     g.filters_dev, g.filter_stride_bytes, g.witnesses_dev, g.witness_stride_bytes, g.stats_dev, true, ctx->ones_pinned + 1);
    return rc;
}

int rbf_encode_gop_begin(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                         uint32_t nframes, uint32_t width, uint32_t height,
                         uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                         uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                         const rbf_seeds *seeds,
                         void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                         void *filters_dev, uint64_t filter_stride_bytes,
                         void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev)
{
    return rbf_encode_runs_begin(ctx, frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes,
                                 thr_floor, thr_floors, nullptr, seeds, masks_dev, mask_stride_bytes, ones_dev, filters_dev, filter_stride_bytes,
                                 witnesses_dev, witness_stride_bytes, stats_dev);
}

int rbf_encode_runs(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                    uint32_t nframes, uint32_t width, uint32_t height,
                    uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                    uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                    const uint8_t *run_starts, const rbf_seeds *seeds,
                    void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                    void *filters_dev, uint64_t filter_stride_bytes,
                    void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev,
                    rbf_filter_params *params_out, double *k_out)
{
    if (int r = rbf_encode_runs_begin(ctx, frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes,
                                      thr_floor, thr_floors, run_starts, seeds, masks_dev, mask_stride_bytes, ones_dev, filters_dev, filter_stride_bytes,
                                      witnesses_dev, witness_stride_bytes, stats_dev))
        return r;
    return rbf_encode_gop_finish(ctx, params_out, k_out);
}

int rbf_encode_gop(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes,
                   uint32_t nframes, uint32_t width, uint32_t height,
                   uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                   uint32_t sample_bytes, int32_t thr_floor, const int32_t *thr_floors,
                   const rbf_seeds *seeds,
                   void *masks_dev, uint64_t mask_stride_bytes, uint64_t *ones_dev,
                   void *filters_dev, uint64_t filter_stride_bytes,
                   void *witnesses_dev, uint64_t witness_stride_bytes, uint64_t *stats_dev,
                   rbf_filter_params *params_out, double *k_out)
{
    if (int r = rbf_encode_gop_begin(ctx, frames_dev, frame_stride_bytes, nframes, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes,
                                     thr_floor, thr_floors, seeds, masks_dev, mask_stride_bytes, ones_dev, filters_dev, filter_stride_bytes,
                                     witnesses_dev, witness_stride_bytes, stats_dev))
        return r;
    return rbf_encode_gop_finish(ctx, params_out, k_out);
}

// ------------------------------------------------------------------------------------------
// exact-size record of a batch (what the multi-GPU gather moves)
// ------------------------------------------------------------------------------------------
uint64_t rbf_record_max_bytes(uint32_t nframes, uint64_t n)
{
    // header + per frame: a filter (or the passthrough mask) and a witness of at most n bits each
    return (uint64_t)(RECORD_HEADER_WORDS + RECORD_ROW_WORDS * (uint64_t)nframes) * 8 + (uint64_t)nframes * 2 * (((n + 63) / 64) * 8);
}

int rbf_pack_records(rbf_ctx *ctx, uint32_t nframes, uint64_t n, const rbf_filter_params *params, const double *k,
                     const void *masks_dev, uint64_t mask_stride_bytes,
                     const void *filters_dev, uint64_t filter_stride_bytes,
                     const void *witnesses_dev, uint64_t witness_stride_bytes,
                     const uint64_t *stats_dev, void *record_dev, uint64_t capacity_bytes)
{
    if (int r = set_device(ctx)) return r;
    if (!params || !masks_dev || !filters_dev || !witnesses_dev || !stats_dev || !record_dev) return fail(RBF_EINVAL, "null pointer");
    if (int r = check_frame_geometry(n, nframes, mask_stride_bytes)) return r;
    if (filter_stride_bytes % 8 || witness_stride_bytes % 8 || ((uintptr_t)record_dev % 8)) return fail(RBF_EINVAL, "strides and the record must be 8-byte aligned");
    const uint64_t header = (uint64_t)(RECORD_HEADER_WORDS + RECORD_ROW_WORDS * (uint64_t)nframes) * 8;
    if (capacity_bytes < header || capacity_bytes % 8) return fail(RBF_EINVAL, "record capacity %llu is smaller than the %llu-byte header or misaligned",
                                                                  (unsigned long long)capacity_bytes, (unsigned long long)header);
    for (uint32_t f = 0; f < nframes; ++f)
        if (params[f].m && ((uint64_t)params[f].m + 63) / 64 * 8 > filter_stride_bytes) return fail(RBF_EINVAL, "frame %u: filter of %u bits exceeds the filter stride", f, params[f].m);
    if (int r = grow((void **)&ctx->pack_base, &ctx->pack_base_cap, ((size_t)nframes / PACK_BATCH + 2) * 8)) return r;
    for (uint32_t first = 0; first < nframes; first += PACK_BATCH) {
        const uint32_t cnt = nframes - first < (uint32_t)PACK_BATCH ? nframes - first : (uint32_t)PACK_BATCH;
        PackTable tab{};
        for (uint32_t i = 0; i < cnt; ++i) {
            PackRow &r = tab.r[i];
            r.m = params[first + i].m; r.floor_k = params[first + i].floor_k; r.threshold = params[first + i].threshold;
            const double kv = k ? k[first + i] : 0.0;
            memcpy(&r.k_bits, &kv, 8);
        }
        LaunchTimer t(ctx, RBF_K_PACK);
        hipLaunchKernelGGL(k_pack_records, dim3(8, 2 * cnt), dim3(256), 0, ctx->stream, tab, first, cnt, nframes, n, stats_dev,
                           (const uint8_t *)masks_dev, mask_stride_bytes, (const uint8_t *)filters_dev, filter_stride_bytes,
                           (const uint8_t *)witnesses_dev, witness_stride_bytes, (uint64_t *)record_dev, capacity_bytes, ctx->pack_base);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

// ------------------------------------------------------------------------------------------
// A6
// ------------------------------------------------------------------------------------------
static int decode_chunk(rbf_ctx *ctx, const void *filters_dev, uint64_t filter_stride_bytes,
                        const void *witnesses_dev, uint64_t witness_stride_bytes,
                        uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                        const rbf_seeds *seeds,
                        void *masks_dev, uint64_t mask_stride_bytes)
{
    const Plan pl = make_plan(ctx, params, nframes, n);
    const uint32_t wps = pl.words_per_seg;                                            // pass words per segment
    FrameTable tab;
    if (int r = fill_table(params, nframes, &tab)) return r;
    if (WG_THREADS % wps) return fail(RBF_EINVAL, "segments of %u words do not tile a workgroup's chunk", wps);
    if (int r = grow((void **)&ctx->seg_cnt, &ctx->seg_cnt_cap, (size_t)nframes * pl.nseg * 4 + 16)) return r;      // + 16: the counts are read four to a load
    if (int r = grow((void **)&ctx->pass_words, &ctx->pass_words_cap, (size_t)nframes * pl.nseg * wps * 8)) return r;
    const Seeds sd{seeds->h1, seeds->h2, seeds->act};
    // mixed batch (see encode_chunk): the query runs twice over disjoint frame sets, the expansion once
    bool split = false;
    {
        uint32_t in_range = 0, out_of_range = 0;
        for (uint32_t f = 0; f < nframes; ++f)
            if (params[f].m) ++((params[f].m >= F64MOD_M_MIN && params[f].m <= F64MOD_M_MAX) ? in_range : out_of_range);
        if (in_range && out_of_range && !ctx->force_generic && !ctx->barrett_only && !ctx->single_buffer) {
            rbf_filter_params small[MAX_BATCH], big[MAX_BATCH];
            for (uint32_t f = 0; f < nframes; ++f) {
                small[f] = big[f] = params[f];
                const bool fp64 = params[f].m >= F64MOD_M_MIN && params[f].m <= F64MOD_M_MAX;
                (fp64 ? small[f] : big[f]).m = 0;
            }
            const Plan ps = make_plan(ctx, small, nframes, n), pb = make_plan(ctx, big, nframes, n);
            if ((pb.query_kind == 1 || pb.query_kind == 3) && pb.f64_mod && ps.nseg == pl.nseg && pb.nseg == pl.nseg &&
                ps.words_per_seg == wps && pb.words_per_seg == wps) {
                FrameTable ts, tb;
                if (int r = fill_table(small, nframes, &ts)) return r;
                if (int r = fill_table(big, nframes, &tb)) return r;
                if (int r = launch_query(ctx, ps, n, nframes, ts, sd, filters_dev, filter_stride_bytes, false)) return r;
                if (int r = launch_query(ctx, pb, n, nframes, tb, sd, filters_dev, filter_stride_bytes, false, false, true)) return r;
                split = true;
            }
        }
    }
    if (!split) if (int r = launch_query(ctx, pl, n, nframes, tab, sd, filters_dev, filter_stride_bytes, false)) return r;
    {   // one lane per 64-position word; the kernel sums the earlier segment counts itself (until round 4: k_scan_segments + k_expand_mask_p)
        const uint64_t bx = (pl.nseg * wps + WG_THREADS - 1) / WG_THREADS;
        if (int r = launch_chunk_offsets(ctx, pl, nframes, (uint32_t)bx)) return r;
        LaunchTimer t(ctx, RBF_K_EXPAND);
        hipLaunchKernelGGL(k_expand_mask, dim3((uint32_t)bx, nframes), dim3(WG_THREADS), 0, ctx->stream,
                           ctx->pass_words, ctx->seg_cnt, pl.nseg, wps, (const uint32_t *)witnesses_dev, witness_stride_bytes / 4,
                           (uint64_t *)masks_dev, mask_stride_bytes / 8, n, ctx->chunk_off);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_bloom_decode_batch(rbf_ctx *ctx, const void *filters_dev, uint64_t filter_stride_bytes,
                           const void *witnesses_dev, uint64_t witness_stride_bytes,
                           uint64_t n, uint32_t nframes, const rbf_filter_params *params,
                           const rbf_seeds *seeds,
                           void *masks_dev, uint64_t mask_stride_bytes)
{
    if (int r = set_device(ctx)) return r;
    if (!filters_dev || !witnesses_dev || !params || !seeds || !masks_dev) return fail(RBF_EINVAL, "null pointer");
    if (int r = check_frame_geometry(n, nframes, mask_stride_bytes)) return r;
    if (int r = check_filter_strides(params, nframes, filter_stride_bytes)) return r;
    if (witness_stride_bytes % 8) return fail(RBF_EINVAL, "witness stride must be a multiple of 8");
    for (uint32_t f0 = 0; f0 < nframes; f0 += MAX_BATCH) {
        const uint32_t cnt = nframes - f0 < (uint32_t)MAX_BATCH ? nframes - f0 : (uint32_t)MAX_BATCH;
        if (int r = decode_chunk(ctx, (const uint8_t *)filters_dev + (uint64_t)f0 * filter_stride_bytes, filter_stride_bytes,
                                 (const uint8_t *)witnesses_dev + (uint64_t)f0 * witness_stride_bytes, witness_stride_bytes, n, cnt,
                                 params + f0, seeds, (uint8_t *)masks_dev + (uint64_t)f0 * mask_stride_bytes, mask_stride_bytes))
            return r;
    }
    return RBF_OK;
}

// ------------------------------------------------------------------------------------------
// per-index surface
// ------------------------------------------------------------------------------------------
static int one_frame(const rbf_filter_params *p, FrameDev *d)
{
    if (!p) return fail(RBF_EINVAL, "params is null");
    if (p->m == 0) return fail(RBF_EINVAL, "filter length m must be >= 1");
    if (p->floor_k > 64) return fail(RBF_ERANGE, "floor_k %u > 64", p->floor_k);
    d->m = p->m; d->floor_k = p->floor_k; d->T = p->threshold;
    d->M = p->m >= 2 ? (uint64_t)((((unsigned __int128)1) << 64) / p->m) : 0;
    return RBF_OK;
}

static uint32_t index_grid(uint64_t count)
{
    uint64_t b = (count + WG_THREADS - 1) / WG_THREADS;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (uint32_t)b;
}

int rbf_filter_insert_indices(rbf_ctx *ctx, void *filter_dev, const rbf_filter_params *params,
                              const rbf_seeds *seeds, const uint32_t *indices_dev, uint64_t count)
{
    if (int r = set_device(ctx)) return r;
    if (!filter_dev || !seeds) return fail(RBF_EINVAL, "null pointer");
    FrameDev fd;
    if (int r = one_frame(params, &fd)) return r;
    if (count == 0) return RBF_OK;
    if (!indices_dev) return fail(RBF_EINVAL, "indices_dev is null");
    {
        LaunchTimer t(ctx, RBF_K_INDEX);
        hipLaunchKernelGGL(k_index_insert, dim3(index_grid(count)), dim3(WG_THREADS), 0, ctx->stream,
                           (uint32_t *)filter_dev, fd, Seeds{seeds->h1, seeds->h2, seeds->act}, indices_dev, count);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_filter_query_indices(rbf_ctx *ctx, const void *filter_dev, const rbf_filter_params *params,
                             const rbf_seeds *seeds, const uint32_t *indices_dev, uint64_t count,
                             uint8_t *out_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!filter_dev || !seeds) return fail(RBF_EINVAL, "null pointer");
    FrameDev fd;
    if (int r = one_frame(params, &fd)) return r;
    if (count == 0) return RBF_OK;
    if (!indices_dev || !out_dev) return fail(RBF_EINVAL, "null pointer");
    {
        LaunchTimer t(ctx, RBF_K_INDEX);
        hipLaunchKernelGGL(k_index_query, dim3(index_grid(count)), dim3(WG_THREADS), 0, ctx->stream,
                           (const uint32_t *)filter_dev, fd, Seeds{seeds->h1, seeds->h2, seeds->act}, indices_dev, count, out_dev);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_filter_insert_keys(rbf_ctx *ctx, void *filter_dev, const rbf_filter_params *params,
                           const rbf_seeds *seeds, uint32_t standard_k,
                           const uint8_t *keys_dev, const uint32_t *offsets_dev, uint64_t count)
{
    if (int r = set_device(ctx)) return r;
    if (!filter_dev || !seeds) return fail(RBF_EINVAL, "null pointer");
    FrameDev fd;
    if (int r = one_frame(params, &fd)) return r;
    if (standard_k > 64) return fail(RBF_ERANGE, "standard_k %u > 64", standard_k);
    if (count == 0) return RBF_OK;
    if (!keys_dev || !offsets_dev) return fail(RBF_EINVAL, "null key arrays");
    {
        LaunchTimer t(ctx, RBF_K_INDEX);
        hipLaunchKernelGGL(k_keys<true>, dim3(index_grid(count)), dim3(WG_THREADS), 0, ctx->stream,
                           (uint32_t *)filter_dev, fd, Seeds{seeds->h1, seeds->h2, seeds->act}, standard_k, keys_dev, offsets_dev, count, (uint8_t *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_filter_query_keys(rbf_ctx *ctx, const void *filter_dev, const rbf_filter_params *params,
                          const rbf_seeds *seeds, uint32_t standard_k,
                          const uint8_t *keys_dev, const uint32_t *offsets_dev, uint64_t count,
                          uint8_t *out_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!filter_dev || !seeds) return fail(RBF_EINVAL, "null pointer");
    FrameDev fd;
    if (int r = one_frame(params, &fd)) return r;
    if (standard_k > 64) return fail(RBF_ERANGE, "standard_k %u > 64", standard_k);
    if (count == 0) return RBF_OK;
    if (!keys_dev || !offsets_dev || !out_dev) return fail(RBF_EINVAL, "null pointer");
    {
        LaunchTimer t(ctx, RBF_K_INDEX);
        hipLaunchKernelGGL(k_keys<false>, dim3(index_grid(count)), dim3(WG_THREADS), 0, ctx->stream,
                           (uint32_t *)filter_dev, fd, Seeds{seeds->h1, seeds->h2, seeds->act}, standard_k, keys_dev, offsets_dev, count, out_dev);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

// ------------------------------------------------------------------------------------------
// A2 / A8
// ------------------------------------------------------------------------------------------
static int values_common(rbf_ctx *ctx, void *frame_dev, uint32_t width, uint32_t height,
                         uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes, uint32_t sample_bytes,
                         uint32_t channels, const void *mask_dev, void *values_dev, uint64_t *count_dev, bool scatter)
{
    if (int r = set_device(ctx)) return r;
    if (!frame_dev || !mask_dev || !values_dev) return fail(RBF_EINVAL, "null device pointer");
    if (width == 0 || height == 0) return fail(RBF_EINVAL, "empty frame");
    if (sample_bytes != 1 && sample_bytes != 2) return fail(RBF_EINVAL, "sample_bytes must be 1 or 2");
    if (channels == 0 || channels > 4) return fail(RBF_EINVAL, "channels must be 1..4");
    if (pixel_stride_bytes < channels * sample_bytes) return fail(RBF_EINVAL, "pixel stride smaller than channels*sample_bytes");
    if (row_pitch_bytes < (uint64_t)width * pixel_stride_bytes) return fail(RBF_EINVAL, "row pitch too small");
    const uint64_t n = (uint64_t)width * height;
    if (n > 0xFFFFFFFFull) return fail(RBF_ERANGE, "frame too large");
    const uint64_t nseg = nseg_of(n);
    if (int r = grow((void **)&ctx->seg_cnt, &ctx->seg_cnt_cap, (size_t)nseg * 4)) return r;
    if (int r = grow((void **)&ctx->seg_off, &ctx->seg_off_cap, (size_t)nseg * 8)) return r;
    const uint64_t bx = (nseg + WG_WAVES - 1) / WG_WAVES;
    LaunchTimer t(ctx, scatter ? RBF_K_SCATTER : RBF_K_GATHER);
    hipLaunchKernelGGL(k_mask_segment_counts, dim3((uint32_t)bx), dim3(WG_THREADS), 0, ctx->stream, (const uint64_t *)mask_dev, 0ull, n, ctx->seg_cnt, nseg);
    hipLaunchKernelGGL(k_scan_segments, dim3(1), dim3(1024), 0, ctx->stream, ctx->seg_cnt, ctx->seg_off, nseg, count_dev, 1u);
    if (sample_bytes == 1) {
        if (scatter) hipLaunchKernelGGL((k_values<uint8_t, true>), dim3((uint32_t)bx), dim3(WG_THREADS), 0, ctx->stream, (uint8_t *)frame_dev, width, n, row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)mask_dev, ctx->seg_off, nseg, (uint8_t *)values_dev);
        else hipLaunchKernelGGL((k_values<uint8_t, false>), dim3((uint32_t)bx), dim3(WG_THREADS), 0, ctx->stream, (uint8_t *)frame_dev, width, n, row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)mask_dev, ctx->seg_off, nseg, (uint8_t *)values_dev);
    } else {
        if (scatter) hipLaunchKernelGGL((k_values<uint16_t, true>), dim3((uint32_t)bx), dim3(WG_THREADS), 0, ctx->stream, (uint8_t *)frame_dev, width, n, row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)mask_dev, ctx->seg_off, nseg, (uint16_t *)values_dev);
        else hipLaunchKernelGGL((k_values<uint16_t, false>), dim3((uint32_t)bx), dim3(WG_THREADS), 0, ctx->stream, (uint8_t *)frame_dev, width, n, row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)mask_dev, ctx->seg_off, nseg, (uint16_t *)values_dev);
    }
    return RBF_OK;
}

int rbf_gather_values_batch(rbf_ctx *ctx, const void *frames_dev, uint64_t frame_stride_bytes, uint32_t nframes,
                            uint32_t width, uint32_t height, uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes,
                            uint32_t sample_bytes, uint32_t channels, const void *masks_dev, uint64_t mask_stride_bytes,
                            void *values_dev, uint64_t capacity_pixels, uint64_t *offsets_dev, uint64_t *uncovered_dev)
{
    if (int r = set_device(ctx)) return r;
    if (!frames_dev || !masks_dev || !values_dev || !offsets_dev) return fail(RBF_EINVAL, "null device pointer");
    if (nframes < 2) return fail(RBF_EINVAL, "need at least 2 frames, got %u", nframes);
    if (width == 0 || height == 0) return fail(RBF_EINVAL, "empty frame");
    if (sample_bytes != 1 && sample_bytes != 2) return fail(RBF_EINVAL, "sample_bytes must be 1 or 2");
    if (channels == 0 || channels > 4) return fail(RBF_EINVAL, "channels must be 1..4");
    if (pixel_stride_bytes < channels * sample_bytes || pixel_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "pixel stride incompatible with channels*sample_bytes");
    if (row_pitch_bytes < (uint64_t)width * pixel_stride_bytes || row_pitch_bytes % sample_bytes) return fail(RBF_EINVAL, "row pitch too small or misaligned");
    if (frame_stride_bytes % sample_bytes) return fail(RBF_EINVAL, "frame stride misaligned");
    const uint64_t n = (uint64_t)width * height;
    const uint32_t pairs = nframes - 1;
    if (pairs > 65535) return fail(RBF_ERANGE, "at most 65536 frames per call");
    if (int r = check_frame_geometry(n, pairs, mask_stride_bytes)) return r;
    const uint64_t nseg = nseg_of(n), nwords = (n + 63) / 64;
    if (int r = grow((void **)&ctx->seg_cnt, &ctx->seg_cnt_cap, (size_t)pairs * nseg * 4)) return r;
    if (int r = grow((void **)&ctx->seg_off, &ctx->seg_off_cap, (size_t)pairs * nseg * 8)) return r;
    if (int r = grow((void **)&ctx->pack_base, &ctx->pack_base_cap, ((size_t)pairs + 2) * 8)) return r;     // per-pair totals
    LaunchTimer t(ctx, RBF_K_GATHER);
    hipLaunchKernelGGL(k_mask_segment_counts, dim3((uint32_t)((nseg + WG_WAVES - 1) / WG_WAVES), pairs), dim3(WG_THREADS), 0, ctx->stream,
                       (const uint64_t *)masks_dev, mask_stride_bytes / 8, n, ctx->seg_cnt, nseg);
    hipLaunchKernelGGL(k_scan_segments, dim3(pairs), dim3(1024), 0, ctx->stream, ctx->seg_cnt, ctx->seg_off, nseg, ctx->pack_base, 1u);
    hipLaunchKernelGGL(k_frame_offsets, dim3(1), dim3(64), 0, ctx->stream, ctx->pack_base, offsets_dev, pairs);
    const dim3 grid((uint32_t)((nwords + WG_THREADS - 1) / WG_THREADS), pairs);
    if (sample_bytes == 1)
        hipLaunchKernelGGL(k_gather_words<uint8_t>, grid, dim3(WG_THREADS), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes, width, n,
                           row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)masks_dev, mask_stride_bytes / 8, ctx->seg_off, nseg,
                           offsets_dev, (uint8_t *)values_dev, capacity_pixels);
    else
        hipLaunchKernelGGL(k_gather_words<uint16_t>, grid, dim3(WG_THREADS), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes, width, n,
                           row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)masks_dev, mask_stride_bytes / 8, ctx->seg_off, nseg,
                           offsets_dev, (uint16_t *)values_dev, capacity_pixels);
    if (uncovered_dev) {
        HIP_TRY(hipMemsetAsync(uncovered_dev, 0, (size_t)pairs * 8, ctx->stream));
        uint64_t bx = (n + WG_THREADS * 8 - 1) / (WG_THREADS * 8);
        if (bx < 1) bx = 1;
        if (bx > 4096) bx = 4096;
        if (sample_bytes == 1)
            hipLaunchKernelGGL(k_uncovered_changes<uint8_t>, dim3((uint32_t)bx, pairs), dim3(WG_THREADS), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                               width, n, row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)masks_dev, mask_stride_bytes / 8, (unsigned long long *)uncovered_dev);
        else
            hipLaunchKernelGGL(k_uncovered_changes<uint16_t>, dim3((uint32_t)bx, pairs), dim3(WG_THREADS), 0, ctx->stream, (const uint8_t *)frames_dev, frame_stride_bytes,
                               width, n, row_pitch_bytes, pixel_stride_bytes, channels, (const uint64_t *)masks_dev, mask_stride_bytes / 8, (unsigned long long *)uncovered_dev);
    }
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_gather_values(rbf_ctx *ctx, const void *frame_dev, uint32_t width, uint32_t height,
                      uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes, uint32_t sample_bytes,
                      uint32_t channels, const void *mask_dev, void *values_dev, uint64_t *count_dev)
{
    if (!count_dev) return fail(RBF_EINVAL, "count_dev is null");
    if (int r = values_common(ctx, (void *)frame_dev, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes, channels, mask_dev, values_dev, count_dev, false)) return r;
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

int rbf_scatter_values(rbf_ctx *ctx, void *frame_dev, uint32_t width, uint32_t height,
                       uint64_t row_pitch_bytes, uint32_t pixel_stride_bytes, uint32_t sample_bytes,
                       uint32_t channels, const void *mask_dev, const void *values_dev)
{
    if (int r = values_common(ctx, frame_dev, width, height, row_pitch_bytes, pixel_stride_bytes, sample_bytes, channels, mask_dev, (void *)values_dev, nullptr, true)) return r;
    HIP_TRY(hipGetLastError());
    return RBF_OK;
}

}  // extern "C"
