// hark_api.hip -- context / table / result objects of the C ABI (include/hark.h)
// and the entry points that only orchestrate kernels from the k_*.hip units.
#include "hark_internal.h"
#include <stdarg.h>

int hark_fail(hark_context *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int hark_launch_failed(hark_context *ctx, hipError_t e, const char *launch_text, const char *file, int line)
{
    char name[160];
    size_t i = 0;
    while (launch_text[i] && i + 1 < sizeof name && !(launch_text[i] == '<' && launch_text[i + 1] == '<' && launch_text[i + 2] == '<')) { name[i] = launch_text[i]; i++; }
    name[i] = 0;
    const char *base = strrchr(file, '/');
    return hark_fail(ctx, HARK_EHIP, "launch of %s failed: %s (%s:%d)", name, hipGetErrorString(e), base ? base + 1 : file, line);
}

// ---- caching device allocator ---------------------------------------------------
// Every entry allocates its result columns and scratch; hipMalloc/hipFree cost
// 0.1-1 ms each and hipFree synchronises the device, which would dominate a
// sub-millisecond query.  Freed blocks are kept per context and handed out again
// (best fit within 25 %).  Reuse is safe without events because a context runs
// everything on ONE stream: a block freed by the host is only touched again by
// work enqueued later on that same stream.
static size_t pool_round(size_t bytes)
{
    if (bytes == 0) bytes = 16;
    const size_t g = bytes <= (1u << 20) ? 4096 : (2u << 20);
    return (bytes + g - 1) / g * g;
}

static void pool_trim(hark_context *ctx)
{
    for (auto &kv : ctx->pool_free) hipFree(kv.second);
    ctx->pool_free.clear();
    ctx->pool_cached = 0;
}

int hark_alloc(hark_context *ctx, void **out, size_t bytes)
{
    hark_device_guard guard__(ctx);
    *out = nullptr;
    const size_t size = pool_round(bytes);
    if (ctx) {
        auto it = ctx->pool_free.lower_bound(size);
        if (it != ctx->pool_free.end() && it->first <= size + size / 4 + 4096) {
            *out = it->second;
            ctx->pool_cached -= it->first;
            ctx->pool_live[*out] = it->first;
            ctx->pool_free.erase(it);
            return HARK_OK;
        }
    }
    hipError_t e = hipMalloc(out, size);
    if (e == hipErrorOutOfMemory && ctx && !ctx->pool_free.empty()) {     // give cached blocks back and retry
        (void)hipGetLastError();
        hipStreamSynchronize(ctx->stream);
        pool_trim(ctx);
        e = hipMalloc(out, size);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        (void)hipGetLastError();
        return hark_fail(ctx, e == hipErrorOutOfMemory ? HARK_ENOMEM : HARK_EHIP,
                         "hipMalloc(%zu bytes) failed: %s", size, hipGetErrorString(e));
    }
    if (ctx) ctx->pool_live[*out] = size;
    return HARK_OK;
}

void hark_free(hark_context *ctx, void *ptr)
{
    if (!ptr) return;
    if (!ctx) { hipFree(ptr); return; }
    auto it = ctx->pool_live.find(ptr);
    if (it == ctx->pool_live.end()) { hipFree(ptr); return; }              // not ours (should not happen)
    const size_t size = it->second;
    ctx->pool_live.erase(it);
    if (ctx->pool_cached + size > ctx->pool_limit) { hipStreamSynchronize(ctx->stream); hipFree(ptr); return; }
    ctx->pool_free.emplace(size, ptr);
    ctx->pool_cached += size;
}

int hark_read_words(hark_context *ctx, const void *dev, int64_t *host, int count)
{
    hark_device_guard guard__(ctx);
    if (count > 64) return hark_fail(ctx, HARK_EARG, "hark_read_words: count > 64");
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pin, dev, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->h_pin, (size_t)count * 8);
    return HARK_OK;
}

// A hipMemcpy into pageable memory is staged by the runtime in small pieces; results of
// a query are typically a few MB, so they go through two 8 MiB pinned buffers instead:
// the copy engine fills one while the host drains the other.
static constexpr size_t kBounce = (size_t)8 << 20;

int hark_d2h(hark_context *ctx, void *host, const void *dev, size_t bytes)
{
    hark_device_guard guard__(ctx);
    if (!bytes) return HARK_OK;
    // measured (tools/ingest_bench.py): the bounce buffers win for results of a few MB, the
    // runtime's own staging wins for hundreds of MB (12-24 vs 8-10 GB/s at 1 GiB)
    if (bytes <= 65536 || bytes > ((size_t)32 << 20)) {
        HIP_TRY(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return HARK_OK;
    }
    for (int i = 0; i < 2; i++) {
        if (!ctx->bounce[i]) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->bounce[i], kBounce));
        if (!ctx->bounce_ev[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->bounce_ev[i], hipEventDisableTiming));
    }
    const size_t nchunk = (bytes + kBounce - 1) / kBounce;
    const char *src = static_cast<const char *>(dev);
    char *dst = static_cast<char *>(host);
    for (size_t c = 0; c <= nchunk; c++) {
        if (c < nchunk) {
            const size_t len = c + 1 < nchunk ? kBounce : bytes - c * kBounce;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->bounce[c & 1], src + c * kBounce, len, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->bounce_ev[c & 1], ctx->stream));
        }
        if (c > 0) {
            const size_t q = c - 1, len = q + 1 < nchunk ? kBounce : bytes - q * kBounce;
            HIP_TRY(ctx, hipEventSynchronize(ctx->bounce_ev[q & 1]));
            memcpy(dst + q * kBounce, ctx->bounce[q & 1], len);
        }
    }
    return HARK_OK;
}

extern "C" {

// ---- pinned host blocks ---------------------------------------------------------------------------------------------
static size_t pin_round(size_t bytes) { const size_t g = bytes <= ((size_t)1 << 20) ? 65536 : (size_t)1 << 20; return (bytes + g - 1) / g * g; }

int hark_host_alloc(hark_context *ctx, void **out, size_t bytes)
{
    if (!ctx || !out) return HARK_EARG;
    hark_device_guard guard__(ctx);
    *out = nullptr;
    const size_t size = pin_round(bytes ? bytes : 1);
    auto it = ctx->pin_free.lower_bound(size);
    if (it != ctx->pin_free.end() && it->first <= 2 * size) {
        *out = it->second;
        ctx->pin_cached -= it->first;
        ctx->pin_live[*out] = it->first;
        ctx->pin_free.erase(it);
        return HARK_OK;
    }
    hipError_t e = hipHostMalloc(out, size);
    if (e != hipSuccess && !ctx->pin_free.empty()) {           // give the cached blocks back and try once more
        (void)hipGetLastError();
        for (auto &kv : ctx->pin_free) hipHostFree(kv.second);
        ctx->pin_free.clear(); ctx->pin_cached = 0;
        e = hipHostMalloc(out, size);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        (void)hipGetLastError();
        return hark_fail(ctx, HARK_ENOMEM, "hipHostMalloc(%zu bytes) failed: %s", size, hipGetErrorString(e));
    }
    ctx->pin_live[*out] = size;
    return HARK_OK;
}

int hark_host_free(hark_context *ctx, void *ptr)
{
    if (!ptr) return HARK_OK;
    if (!ctx) return hipHostFree(ptr) == hipSuccess ? HARK_OK : HARK_EHIP;   // the block outlived its context
    hark_device_guard guard__(ctx);
    auto it = ctx->pin_live.find(ptr);
    if (it == ctx->pin_live.end()) return hark_fail(ctx, HARK_EARG, "hark_host_free: not a live block of this context");
    const size_t size = it->second;
    ctx->pin_live.erase(it);
    if (ctx->pin_cached + size > ctx->pin_limit) { hipHostFree(ptr); return HARK_OK; }
    ctx->pin_free.emplace(size, ptr);
    ctx->pin_cached += size;
    return HARK_OK;
}

int hark_version(void) { return 103; }   // 1.02: context_trim, futhark_* veneer (futhark_compat.h), hash join, multi-aggregate pass

int hark_context_new(hark_context **out, int device)
{
    if (!out) return HARK_EARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HARK_EHIP;   // no GPU: fail loudly
    if (device < 0 || device >= ndev) return HARK_EARG;
    if (hipSetDevice(device) != hipSuccess) return HARK_EHIP;
    hark_context *ctx = new hark_context();
    ctx->device = device;
    if (const char *lim = getenv("HARK_POOL_LIMIT_MB")) { const long long mb = atoll(lim); if (mb >= 0) ctx->pool_limit = (size_t)mb << 20; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return HARK_EHIP; }
    ctx->stream = ctx->own_stream;
    if (hipMalloc((void **)&ctx->d_err, 64) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_pin, 65536) != hipSuccess) {
        hark_context_free(ctx);
        return HARK_EHIP;
    }
    hipMemset(ctx->d_err, 0, 64);
    *out = ctx;
    return HARK_OK;
}

void hark_context_free(hark_context *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    pool_trim(ctx);
    // cached pinned blocks go; blocks still in a caller's hands stay valid and are released by hark_host_free(NULL, p)
    for (auto &kv : ctx->pin_free) hipHostFree(kv.second);
    if (ctx->d_err) hipFree(ctx->d_err);
    if (ctx->h_pin) hipHostFree(ctx->h_pin);
    for (int i = 0; i < 2; i++) { if (ctx->bounce[i]) hipHostFree(ctx->bounce[i]); if (ctx->bounce_ev[i]) hipEventDestroy(ctx->bounce_ev[i]); }
    if (ctx->aux_event) hipEventDestroy(ctx->aux_event);
    if (ctx->main_event) hipEventDestroy(ctx->main_event);
    if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int hark_context_sync(hark_context *ctx)
{
    hark_device_guard guard__(ctx);
    if (!ctx) return HARK_EARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HARK_OK;
}

int hark_context_trim(hark_context *ctx)
{
    if (!ctx) return HARK_EARG;
    hark_device_guard guard__(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    pool_trim(ctx);
    for (auto &kv : ctx->pin_free) hipHostFree(kv.second);
    ctx->pin_free.clear(); ctx->pin_cached = 0;
    return HARK_OK;
}

const char *hark_context_get_error(hark_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int hark_context_set_stream(hark_context *ctx, void *hip_stream)
{
    hark_device_guard guard__(ctx);
    if (!ctx) return HARK_EARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // pooled blocks are ordered by ONE stream: drain the old one
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return HARK_OK;
}

// ---- device memory helpers -------------------------------------------------
int hark_dev_alloc(hark_context *ctx, void **out, int64_t bytes)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || bytes < 0) return HARK_EARG;
    return hark_alloc(ctx, out, (size_t)bytes);
}
int hark_dev_free(hark_context *ctx, void *dev)
{
    hark_device_guard guard__(ctx);
    if (!ctx) return HARK_EARG;
    if (!dev) return HARK_OK;
    hark_free(ctx, dev);
    return HARK_OK;
}
int hark_dev_upload(hark_context *ctx, void *dev, const void *host, int64_t bytes)
{
    hark_device_guard guard__(ctx);
    if (!ctx || bytes < 0 || (bytes && (!dev || !host))) return HARK_EARG;
    if (!bytes) return HARK_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HARK_OK;
}
int hark_dev_download(hark_context *ctx, void *host, const void *dev, int64_t bytes)
{
    hark_device_guard guard__(ctx);
    if (!ctx || bytes < 0 || (bytes && (!dev || !host))) return HARK_EARG;
    return hark_d2h(ctx, host, dev, (size_t)bytes);
}
int hark_op_zero(hark_context *ctx, void *dev, int64_t bytes)
{
    hark_device_guard guard__(ctx);
    if (!ctx || bytes < 0 || (bytes && !dev)) return HARK_EARG;
    if (!bytes) return HARK_OK;
    HIP_TRY(ctx, hipMemsetAsync(dev, 0, (size_t)bytes, ctx->stream));
    return HARK_OK;
}

// ---- tables ------------------------------------------------------------------
static bool dtype_ok(int d) { return d == HARK_I32 || d == HARK_U32 || d == HARK_F32 || d == HARK_I64; }

static void table_release(hark_context *ctx, hark_table *t)
{
    for (auto &c : t->cols) if (c.owned && c.data) hark_free(ctx, c.data);
    delete t;
}

int hark_table_new_columns(hark_context *ctx, hark_table **out, int64_t n, int64_t m,
                           const int32_t *dtypes, const void *const *host_cols)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out) return HARK_EARG;
    *out = nullptr;
    if (n < 0 || m < 0 || (m && (!dtypes || !host_cols))) return hark_fail(ctx, HARK_EARG, "table_new_columns: bad shape");
    hark_table *t = new hark_table();
    t->n = n; t->m = m;
    t->cols.resize((size_t)m);
    for (int64_t j = 0; j < m; j++) {
        if (!dtype_ok(dtypes[j]) || (n && !host_cols[j])) { table_release(ctx, t); return hark_fail(ctx, HARK_EARG, "table_new_columns: bad column %lld", (long long)j); }
        t->cols[j].dtype = dtypes[j];
        size_t bytes = (size_t)n * hark_dtype_size(dtypes[j]);
        int rc = hark_alloc(ctx, &t->cols[j].data, bytes);
        if (rc) { table_release(ctx, t); return rc; }
        if (bytes) {
            hipError_t e = hipMemcpyAsync(t->cols[j].data, host_cols[j], bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { table_release(ctx, t); return hark_fail(ctx, HARK_EHIP, "upload failed: %s", hipGetErrorString(e)); }
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host buffers may be released by the caller
    *out = t;
    return HARK_OK;
}

int hark_table_new_2d(hark_context *ctx, hark_table **out, const void *host, int dtype,
                      int64_t n, int64_t m, int64_t row_stride, int64_t col_stride)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out) return HARK_EARG;
    *out = nullptr;
    if (n < 0 || m < 0 || !dtype_ok(dtype) || (n && m && !host)) return hark_fail(ctx, HARK_EARG, "table_new_2d: bad argument");
    const size_t es = hark_dtype_size(dtype);
    std::vector<int32_t> dts((size_t)m, dtype);
    std::vector<const void *> ptrs((size_t)m, nullptr);
    std::vector<char> staging;
    const char *base = static_cast<const char *>(host);
    if (row_stride == 1 || n <= 1) {                 // F order: every column is already contiguous
        for (int64_t j = 0; j < m; j++) ptrs[j] = base + (size_t)j * (size_t)col_stride * es;
    } else {                                         // gather each column on the host (ingest, not the hot path)
        staging.resize((size_t)n * (size_t)m * es);
        for (int64_t j = 0; j < m; j++) {
            char *dst = staging.data() + (size_t)j * (size_t)n * es;
            const char *src = base + (size_t)j * (size_t)col_stride * es;
            if (es == 4) for (int64_t r = 0; r < n; r++) reinterpret_cast<uint32_t *>(dst)[r] = *reinterpret_cast<const uint32_t *>(src + (size_t)r * (size_t)row_stride * es);
            else for (int64_t r = 0; r < n; r++) reinterpret_cast<uint64_t *>(dst)[r] = *reinterpret_cast<const uint64_t *>(src + (size_t)r * (size_t)row_stride * es);
            ptrs[j] = dst;
        }
    }
    return hark_table_new_columns(ctx, out, n, m, dts.data(), ptrs.data());
}

int hark_table_from_device(hark_context *ctx, hark_table **out, int64_t n, int64_t m,
                           const int32_t *dtypes, void *const *dev_cols)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out) return HARK_EARG;
    *out = nullptr;
    if (n < 0 || m < 0 || (m && (!dtypes || !dev_cols))) return hark_fail(ctx, HARK_EARG, "table_from_device: bad shape");
    hark_table *t = new hark_table();
    t->n = n; t->m = m;
    t->cols.resize((size_t)m);
    for (int64_t j = 0; j < m; j++) {
        if (!dtype_ok(dtypes[j]) || (n && !dev_cols[j]) || (reinterpret_cast<uintptr_t>(dev_cols[j]) & 15u)) {
            delete t;
            return hark_fail(ctx, HARK_EARG, "table_from_device: column %lld is null, misaligned or of unknown dtype", (long long)j);
        }
        t->cols[j].dtype = dtypes[j]; t->cols[j].data = dev_cols[j]; t->cols[j].owned = false;
    }
    *out = t;
    return HARK_OK;
}

int hark_table_shape(const hark_table *t, int64_t *n, int64_t *m)
{
    if (!t) return HARK_EARG;
    if (n) *n = t->n;
    if (m) *m = t->m;
    return HARK_OK;
}
int hark_table_dtype(const hark_table *t, int64_t col) { return (!t || col < 0 || col >= t->m) ? -1 : t->cols[col].dtype; }
void *hark_table_column_device(const hark_table *t, int64_t col) { return (!t || col < 0 || col >= t->m) ? nullptr : t->cols[col].data; }

int hark_table_free(hark_context *ctx, hark_table *t)
{
    hark_device_guard guard__(ctx);
    if (!t) return HARK_OK;
    table_release(ctx, t);
    return HARK_OK;
}

// ---- results -----------------------------------------------------------------
int hark_result_shape(const hark_result *r, int64_t *n, int64_t *m)
{
    if (!r) return HARK_EARG;
    if (n) *n = r->n;
    if (m) *m = (int64_t)r->cols.size();
    return HARK_OK;
}
int hark_result_dtype(const hark_result *r, int64_t col) { return (!r || col < 0 || col >= (int64_t)r->cols.size()) ? -1 : r->cols[col].dtype; }
void *hark_result_column_device(const hark_result *r, int64_t col) { return (!r || col < 0 || col >= (int64_t)r->cols.size()) ? nullptr : r->cols[col].data; }

int hark_result_column(hark_context *ctx, const hark_result *r, int64_t col, void *host_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !r || col < 0 || col >= (int64_t)r->cols.size()) return HARK_EARG;
    size_t bytes = (size_t)r->n * hark_dtype_size(r->cols[col].dtype);
    if (!bytes) return HARK_OK;
    if (!host_out) return HARK_EARG;
    return hark_d2h(ctx, host_out, r->cols[col].data, bytes);
}

// The first `rows` rows of EVERY column in one go (LIMIT applied before PCIe): all copies are enqueued, the stream is
// drained once.  host_outs[j] receives rows x sizeof(dtype of column j) bytes.  (One call and one synchronisation per
// column cost ~25 us each: 0.1 ms of a 2.2 ms statement with five result columns.)
int hark_result_columns_prefix(hark_context *ctx, const hark_result *r, int64_t rows, void *const *host_outs)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !r || rows < 0 || (!host_outs && !r->cols.empty())) return HARK_EARG;
    if (rows > r->n) rows = r->n;
    if (rows == 0 || r->cols.empty()) return HARK_OK;
    size_t total = 0;
    for (auto &c : r->cols) total += ((size_t)rows * hark_dtype_size(c.dtype) + 15) & ~(size_t)15;
    if (total > 65536) {                                         // larger than the pinned scratch: column by column
        for (size_t j = 0; j < r->cols.size(); j++) HARK_TRY(hark_d2h(ctx, host_outs[j], r->cols[j].data, (size_t)rows * hark_dtype_size(r->cols[j].dtype)));
        return HARK_OK;
    }
    char *pin = reinterpret_cast<char *>(ctx->h_pin);
    size_t off = 0;
    for (auto &c : r->cols) {
        const size_t b = (size_t)rows * hark_dtype_size(c.dtype);
        HIP_TRY(ctx, hipMemcpyAsync(pin + off, c.data, b, hipMemcpyDeviceToHost, ctx->stream));
        off += (b + 15) & ~(size_t)15;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    off = 0;
    for (size_t j = 0; j < r->cols.size(); j++) {
        const size_t b = (size_t)rows * hark_dtype_size(r->cols[j].dtype);
        memcpy(host_outs[j], pin + off, b);
        off += (b + 15) & ~(size_t)15;
    }
    return HARK_OK;
}

// The first `rows` rows of every column into ONE pinned host block the caller owns afterwards (hark_host_free): the copies are
// enqueued back to back on the stream and the stream is drained once; column j starts at offsets[j] (64-byte aligned).  The
// copy engine writes the caller's memory directly -- no bounce buffer and no host memcpy (the bounce path drained 8-MiB
// pinned buffers with a single-threaded memcpy: 10-24 GB/s against ~50 for the DMA itself).
int hark_result_columns_pinned(hark_context *ctx, const hark_result *r, int64_t rows, void **host_block, int64_t *offsets)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !r || !host_block || rows < 0 || (!offsets && !r->cols.empty())) return HARK_EARG;
    *host_block = nullptr;
    if (rows > r->n) rows = r->n;
    size_t total = 0;
    for (size_t j = 0; j < r->cols.size(); j++) { offsets[j] = (int64_t)total; total += ((size_t)rows * hark_dtype_size(r->cols[j].dtype) + 63) & ~(size_t)63; }
    if (total == 0) return HARK_OK;
    void *blk = nullptr;
    HARK_TRY(hark_host_alloc(ctx, &blk, total));
    for (size_t j = 0; j < r->cols.size(); j++) {
        const size_t b = (size_t)rows * hark_dtype_size(r->cols[j].dtype);
        if (!b) continue;
        if (hipMemcpyAsync(static_cast<char *>(blk) + offsets[j], r->cols[j].data, b, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
            hipStreamSynchronize(ctx->stream); hark_host_free(ctx, blk);
            return hark_fail(ctx, HARK_EHIP, "result_columns_pinned: copy of column %zu failed: %s", j, hipGetErrorString(hipGetLastError()));
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        hark_host_free(ctx, blk);
        return hark_fail(ctx, HARK_EHIP, "result_columns_pinned: %s", hipGetErrorString(hipGetLastError()));
    }
    *host_block = blk;
    return HARK_OK;
}

// n contiguous bytes of device memory into a pinned host block the caller owns afterwards
int hark_dev_download_pinned(hark_context *ctx, const void *dev, size_t bytes, void **host_block)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !host_block || (bytes && !dev)) return HARK_EARG;
    *host_block = nullptr;
    if (!bytes) return HARK_OK;
    void *blk = nullptr;
    HARK_TRY(hark_host_alloc(ctx, &blk, bytes));
    if (hipMemcpyAsync(blk, dev, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        hark_host_free(ctx, blk);
        return hark_fail(ctx, HARK_EHIP, "dev_download_pinned: %s", hipGetErrorString(hipGetLastError()));
    }
    *host_block = blk;
    return HARK_OK;
}

int hark_result_free(hark_context *ctx, hark_result *r)
{
    hark_device_guard guard__(ctx);
    if (!r) return HARK_OK;
    for (auto &c : r->cols) if (c.owned && c.data) hark_free(ctx, c.data);
    hark_result_host_release(ctx, r);
    delete r;
    return HARK_OK;
}

// ---- raw operators -------------------------------------------------------------
int hark_op_gen_columns(hark_context *ctx, uint64_t seed, int64_t first_row, int64_t n,
                        uint32_t G, int32_t exact, float *p, int32_t *k, float *v)
{
    hark_device_guard guard__(ctx);
    if (!ctx) return HARK_EARG;
    return k_gen_columns(ctx, seed, first_row, n, G, exact, p, k, v);
}

// The measured ceiling the roofline fractions are quoted beside ("fraction of
// measured copy"): a pure read stream over up to three buffers AT ONCE (the shape of the fused kernels'
// input: three columns), one 1024-thread workgroup per CU, two 16-byte non-temporal loads in flight per
// buffer and lane.
__global__ __launch_bounds__(1024) void stream_read_kernel(const uint4 *__restrict__ b0, const uint4 *__restrict__ b1, const uint4 *__restrict__ b2,
                                                           int64_t nvec, unsigned long long *__restrict__ fold)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v *q0 = reinterpret_cast<const u4v *>(b0), *q1 = reinterpret_cast<const u4v *>(b1), *q2 = reinterpret_cast<const u4v *>(b2);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    u4v acc = {0u, 0u, 0u, 0u};
    for (; i + stride < nvec; i += 2 * stride) {
        const u4v a0 = __builtin_nontemporal_load(q0 + i), a1 = __builtin_nontemporal_load(q0 + i + stride);
        acc ^= a0 ^ a1;
        if (q1) { const u4v c0 = __builtin_nontemporal_load(q1 + i), c1 = __builtin_nontemporal_load(q1 + i + stride); acc ^= c0 ^ c1; }
        if (q2) { const u4v d0 = __builtin_nontemporal_load(q2 + i), d1 = __builtin_nontemporal_load(q2 + i + stride); acc ^= d0 ^ d1; }
    }
    for (; i < nvec; i += stride) {
        acc ^= __builtin_nontemporal_load(q0 + i);
        if (q1) acc ^= __builtin_nontemporal_load(q1 + i);
        if (q2) acc ^= __builtin_nontemporal_load(q2 + i);
    }
    unsigned long long x = ((unsigned long long)(acc.x ^ acc.z) << 32) | (acc.y ^ acc.w);
    for (int d = 32; d; d >>= 1) x ^= __shfl_xor(x, d, 64);
    if ((threadIdx.x & 63) == 0) atomicXor(fold, x);
}

int hark_op_stream_read(hark_context *ctx, const void *const *bufs, int32_t nbuf, int64_t bytes_each, uint64_t *fold_dev)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !bufs || nbuf < 1 || nbuf > 3 || bytes_each < 0 || !fold_dev || (bytes_each & 15)) return HARK_EARG;
    for (int j = 0; j < nbuf; j++) if (bytes_each && (!bufs[j] || ((uintptr_t)bufs[j] & 15))) return HARK_EARG;
    if (bytes_each == 0) return HARK_OK;
    stream_read_kernel<<<dim3((unsigned)ctx->num_cu), dim3(1024), 0, ctx->stream>>>(
        static_cast<const uint4 *>(bufs[0]), nbuf > 1 ? static_cast<const uint4 *>(bufs[1]) : nullptr,
        nbuf > 2 ? static_cast<const uint4 *>(bufs[2]) : nullptr, bytes_each / 16, reinterpret_cast<unsigned long long *>(fold_dev));
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

// The traffic mix of a partition producer as a plain stream (no scatter, no LDS): read three buffers, write
// `sixteenths`/16 of one buffer's volume contiguously, all non-temporal.  12 of 16 = the 3 B written per 12 B read
// of the headline path at 50 % selectivity with 6-byte pairs.  What this takes is the floor of any producer with
// that byte mix on this device (bench.py quotes the two-pass floor from it).
__global__ __launch_bounds__(1024) void stream_mix_kernel(const uint4 *__restrict__ b0, const uint4 *__restrict__ b1, const uint4 *__restrict__ b2,
                                                          int64_t nvec, uint4 *__restrict__ dst, int sixteenths)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v *q0 = reinterpret_cast<const u4v *>(b0), *q1 = reinterpret_cast<const u4v *>(b1), *q2 = reinterpret_cast<const u4v *>(b2);
    u4v *out = reinterpret_cast<u4v *>(dst);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto emit = [&](int64_t j, const u4v x) { if ((int)(j & 15) < sixteenths) __builtin_nontemporal_store(x, out + (j >> 4) * sixteenths + (j & 15)); };
    for (; i + stride < nvec; i += 2 * stride) {
        const u4v a0 = __builtin_nontemporal_load(q0 + i), a1 = __builtin_nontemporal_load(q0 + i + stride);
        const u4v c0 = __builtin_nontemporal_load(q1 + i), c1 = __builtin_nontemporal_load(q1 + i + stride);
        const u4v d0 = __builtin_nontemporal_load(q2 + i), d1 = __builtin_nontemporal_load(q2 + i + stride);
        emit(i, a0 ^ c0 ^ d0); emit(i + stride, a1 ^ c1 ^ d1);
    }
    for (; i < nvec; i += stride)
        emit(i, __builtin_nontemporal_load(q0 + i) ^ __builtin_nontemporal_load(q1 + i) ^ __builtin_nontemporal_load(q2 + i));
}

int hark_op_stream_mix(hark_context *ctx, const void *const *bufs, int64_t bytes_each, void *dst, int32_t sixteenths)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !bufs || bytes_each < 0 || (bytes_each & 255) || sixteenths < 0 || sixteenths > 16) return HARK_EARG;
    for (int j = 0; j < 3; j++) if (bytes_each && (!bufs[j] || ((uintptr_t)bufs[j] & 15))) return HARK_EARG;
    if (bytes_each && sixteenths && (!dst || ((uintptr_t)dst & 15))) return HARK_EARG;
    if (bytes_each == 0) return HARK_OK;
    stream_mix_kernel<<<dim3((unsigned)ctx->num_cu), dim3(1024), 0, ctx->stream>>>(
        static_cast<const uint4 *>(bufs[0]), static_cast<const uint4 *>(bufs[1]), static_cast<const uint4 *>(bufs[2]), bytes_each / 16,
        static_cast<uint4 *>(dst), sixteenths);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

int hark_op_filter_groupby_dense_f32(hark_context *ctx, hark_fgb_plan *plan,
                                     const float *p, int32_t cmp, float thr,
                                     const int32_t *k, const float *v, int64_t n)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !plan) return HARK_EARG;
    if (n < 0 || (n && !k)) return hark_fail(ctx, HARK_EARG, "filter_groupby: null key column");     // v == NULL: COUNT only
    if (plan->max_rows && n > plan->max_rows) return hark_fail(ctx, HARK_EARG, "filter_groupby: n exceeds the plan's max_rows");
    if (n > 0xFFFFFFFFll) return hark_fail(ctx, HARK_EARG, "filter_groupby: at most 2^32-1 rows per call (shard larger tables)");
    return k_fgb_dense_f32(ctx, plan, p, cmp, thr, k, v, n);
}

} // extern "C"
