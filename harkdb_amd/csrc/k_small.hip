// k_small.hip -- the reference's own two statements on tables of a few rows (BASELINE.json configs[0]: `select col1, col3`
// and `select col1, max(col3) ... group by col1` on the 7-row data.csv, README.md:42 / test.py:7).
//
// The general paths are built for 1e8..1e9 rows: a dozen launches and three to four host round trips (the number of groups,
// the key range, the result matrix) -- 70 us / 255 us through sql() for seven rows, where the reference's sequential C needs
// microseconds.  Tables of at most kSmallRows rows take ONE launch and ONE synchronisation instead:
//   * query_sel  (select.fut:17-23): the selected columns are written twice, as device columns (the result object other
//     operators may read) and as the row-major [n][k] matrix straight into a pinned host block (PCIe writes from the kernel:
//     no copy is enqueued);
//   * query_groupby (groupby.fut:51-62): ONE workgroup sorts (key, row) words in LDS (a bitonic network over 64-bit words:
//     stable, the row id is the low word), flags the heads of the runs (mk_flags, groupby.fut:26-33), numbers the groups
//     with a scan, folds every select column into an LDS table [group][column] with LDS atomics -- type_func's four
//     operators (groupby.fut:35-41) are associative and commutative on u32, so the table-order left fold of the
//     sequential backend and any other order give the same bits --, and writes the [G][s] matrix (leading key column,
//     ascending unsigned key) to the pinned block and to device columns; G comes back in the context's pinned scratch.
// hark_result_matrix_pinned hands the block over without touching the GPU again.
#include "hark_internal.h"

namespace {

constexpr int kSmallRows = 4096, kSmallThreads = 1024, kSmallCols = 32;
constexpr int kSmallTable = 16384;                       // words of the LDS group table: rows x result columns at most
struct SmallArgs { const uint32_t *src[kSmallCols]; int op[kSmallCols]; int s; };   // result column j: its source column and operator (group-by: column 0 is the key)

enum { SOP_PROD = 1, SOP_SUM = 2, SOP_MAX = 3, SOP_MIN = 4 };                        // type_func's opcodes; anything else is min (groupby.fut:41)

__global__ __launch_bounds__(256) void small_sel_kernel(SmallArgs a, int n, int stride, uint32_t *__restrict__ dev_cols /* [k][stride] */, uint32_t *__restrict__ host /* [n][k] */)
{
    const int total = n * a.s;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / a.s, j = i - r * a.s;
        const uint32_t x = a.src[j][r];
        host[i] = x; dev_cols[(size_t)j * stride + r] = x;
    }
}

__device__ __forceinline__ uint32_t sop_identity(int op) { return op == SOP_PROD ? 1u : op == SOP_SUM ? 0u : op == SOP_MAX ? 0u : 0xFFFFFFFFu; }

// LDS (dynamic): u64 sk[p2]; u16 gid[p2]; u32 table[G * s]
__global__ __launch_bounds__(kSmallThreads) void small_groupby_kernel(SmallArgs a, int n, int p2, int stride, uint32_t *__restrict__ dev_cols /* [s][stride] */,
                                                                      uint32_t *__restrict__ host /* [G][s] */, uint32_t *__restrict__ hdr /* [0] = G */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long *sk = reinterpret_cast<unsigned long long *>(lds_raw);
    uint16_t *gid = reinterpret_cast<uint16_t *>(sk + p2);
    uint32_t *table = reinterpret_cast<uint32_t *>(gid + p2);
    __shared__ uint32_t s_wave[kSmallThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = a.s;
    for (int i = tid; i < p2; i += kSmallThreads) sk[i] = i < n ? ((unsigned long long)a.src[0][i] << 32) | (unsigned)i : ~0ull;   // (the padding sorts behind every row)
    __syncthreads();
    // ---- stable sort by the unsigned key (rsort, groupby.fut:8-22): a bitonic network, every exchange moves the smaller word down
    auto exchange = [&](int i, int j) { const unsigned long long x = sk[i], y = sk[j]; if (x > y) { sk[i] = y; sk[j] = x; } };
    for (int k = 2; k <= p2; k <<= 1) {
        const int hk = k >> 1;
        for (int x = tid; x < (p2 >> 1); x += kSmallThreads) { const int blk = x / hk, off = x - blk * hk; exchange(blk * k + off, blk * k + k - 1 - off); }
        __syncthreads();
        for (int jj = hk >> 1; jj >= 1; jj >>= 1) {
            for (int x = tid; x < (p2 >> 1); x += kSmallThreads) { const int i = 2 * jj * (x / jj) + x % jj; exchange(i, i + jj); }
            __syncthreads();
        }
    }
    // ---- heads of the runs (mk_flags), numbered by a scan: four consecutive positions per thread
    const int per = (p2 + kSmallThreads - 1) / kSmallThreads, i0 = tid * per;
    uint32_t heads = 0;
    for (int q = 0; q < per; q++) { const int i = i0 + q; if (i < n && (i == 0 || (uint32_t)(sk[i - 1] >> 32) != (uint32_t)(sk[i] >> 32))) heads++; }
    uint32_t incl = heads;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - heads, G = 0;
    for (int w = 0; w < kSmallThreads / 64; w++) { const uint32_t x = s_wave[w]; if (w < wave) run += x; G += x; }
    for (int q = 0; q < per; q++) {
        const int i = i0 + q;
        if (i < n) {
            const bool head = i == 0 || (uint32_t)(sk[i - 1] >> 32) != (uint32_t)(sk[i] >> 32);
            if (head) {                                                  // the group's row of the table: its key, every operator's identity
                table[run * s] = (uint32_t)(sk[i] >> 32);
                for (int j = 1; j < s; j++) table[run * s + j] = sop_identity(a.op[j]);
                run++;
            }
            gid[i] = (uint16_t)(run - 1u);
        }
    }
    __syncthreads();
    // ---- fold (segmented_reduce with merge, groupby.fut:45-58)
    for (int x = tid; x < n * (s - 1); x += kSmallThreads) {
        const int i = x / (s - 1), j = 1 + x - i * (s - 1);
        const uint32_t v = a.src[j][(uint32_t)sk[i]], g = gid[i];
        uint32_t *slot = &table[g * s + j];
        switch (a.op[j]) {
        case SOP_SUM: atomicAdd(slot, v); break;
        case SOP_MAX: atomicMax(slot, v); break;
        case SOP_PROD: { uint32_t old = *slot, seen; do { seen = old; old = atomicCAS(slot, seen, seen * v); } while (old != seen); break; }
        default: atomicMin(slot, v); break;
        }
    }
    __syncthreads();
    for (int x = tid; x < (int)G * s; x += kSmallThreads) {
        const int g = x / s, j = x - g * s;
        const uint32_t v = table[x];
        host[x] = v; dev_cols[(size_t)j * stride + g] = v;
    }
    if (tid == 0) hdr[0] = G;
}

bool small_off() { return getenv("HARK_NO_SMALL") != nullptr; }

} // namespace

bool k_small_fits(const hark_table *db, int64_t result_cols)
{
    if (small_off() || db->n <= 0 || db->n > kSmallRows || result_cols < 1 || result_cols > kSmallCols || db->n * result_cols > kSmallTable) return false;
    return true;
}

// fills res (device columns in ONE block: column 0 owns it) and res->host_matrix; returns the usual codes
static int small_finish(hark_context *ctx, hark_result *res, uint32_t *dev, void *blk, int64_t n_alloc, int64_t rows, int64_t s, int dtype)
{
    res->n = rows; res->cols.resize((size_t)s);
    for (int64_t j = 0; j < s; j++) { res->cols[j].dtype = dtype; res->cols[j].data = dev + j * n_alloc; res->cols[j].owned = j == 0; }
    res->host_matrix = blk; res->host_rows = rows; res->host_cols = s;
    return HARK_OK;
}

int k_small_query_sel(hark_context *ctx, const hark_table *db, const int32_t *cols, int64_t k, hark_result *res)
{
    SmallArgs a{};
    a.s = (int)k;
    for (int64_t j = 0; j < k; j++) a.src[j] = static_cast<const uint32_t *>(db->cols[cols[j]].data);
    uint32_t *dev = nullptr; void *blk = nullptr;
    const size_t bytes = (size_t)db->n * (size_t)k * 4;
    const int64_t stride = (db->n + 15) & ~(int64_t)15;                   // device columns start on 64-byte boundaries (tables built over them ask for 16)
    HARK_TRY(hark_alloc(ctx, (void **)&dev, (size_t)stride * (size_t)k * 4));
    int rc = hark_host_alloc(ctx, &blk, bytes);
    if (rc) { hark_free(ctx, dev); return rc; }
    const int total = (int)(db->n * k);
    small_sel_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream>>>(a, (int)db->n, (int)stride, dev, static_cast<uint32_t *>(blk));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hark_free(ctx, dev); hark_host_free(ctx, blk); return hark_launch_failed(ctx, e, "small_sel_kernel<<<", __FILE__, __LINE__); }
    rc = small_finish(ctx, res, dev, blk, stride, db->n, k, HARK_I32);
    for (int64_t j = 0; j < k; j++) res->cols[j].dtype = db->cols[cols[j]].dtype;      // bit copies of 4-byte columns: i32 / u32 / f32 as they are
    return rc;
}

// cols[0] = the key column, cols[1..s) the select columns; ops[j] = type_func opcode of column j (1 prod, 2 sum, 3 max, else min)
int k_small_query_groupby(hark_context *ctx, const hark_table *db, const int32_t *cols, const int32_t *ops, int64_t s, hark_result *res, int64_t *G_out)
{
    SmallArgs a{};
    a.s = (int)s;
    for (int64_t j = 0; j < s; j++) { a.src[j] = static_cast<const uint32_t *>(db->cols[cols[j]].data); a.op[j] = ops[j]; }
    int p2 = 64;
    while (p2 < db->n) p2 <<= 1;
    const size_t lds = (size_t)p2 * 10 + (size_t)db->n * (size_t)s * 4;
    static bool attr_of[64] = {};                                          // once per device
    bool &attr_set = attr_of[ctx->device & 63];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&small_groupby_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallRows * 10 + kSmallTable * 4) != hipSuccess)
            return hark_fail(ctx, HARK_EHIP, "small_groupby: setting the dynamic LDS size failed");
        attr_set = true;
    }
    uint32_t *dev = nullptr; void *blk = nullptr;
    const size_t bytes = (size_t)db->n * (size_t)s * 4;
    const int64_t stride = (db->n + 15) & ~(int64_t)15;
    HARK_TRY(hark_alloc(ctx, (void **)&dev, (size_t)stride * (size_t)s * 4));
    int rc = hark_host_alloc(ctx, &blk, bytes);
    if (rc) { hark_free(ctx, dev); return rc; }
    uint32_t *hdr = reinterpret_cast<uint32_t *>(ctx->h_pin);              // pinned scratch of the context: the kernel writes G there
    small_groupby_kernel<<<dim3(1), dim3(kSmallThreads), lds, ctx->stream>>>(a, (int)db->n, p2, (int)stride, dev, static_cast<uint32_t *>(blk), hdr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hark_free(ctx, dev); hark_host_free(ctx, blk); return hark_launch_failed(ctx, e, "small_groupby_kernel<<<", __FILE__, __LINE__); }
    *G_out = (int64_t)hdr[0];
    return small_finish(ctx, res, dev, blk, stride, *G_out, s, HARK_U32);
}
