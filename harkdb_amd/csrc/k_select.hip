// k_select.hip -- projection and WHERE compaction.
//
//   hark_entry_query_sel   replaces futhark/select.fut:9-23 (`map (sel cols) db`):
//       the reference gathers k of m words out of every row-major row; with
//       one HBM buffer per column the projection is k coalesced column copies
//       (4 B read + 4 B written per selected cell, nothing else touched).
//   hark_entry_filter_sel  is the WHERE the reference only sketches
//       (select.fut:18 `-- let rows_to_keep = filter f db`): order-preserving
//       stream compaction = per-lane predicate -> wave64 ballot + popcount
//       prefix -> workgroup scan -> global offsets from a scan of per-tile
//       counts.  Indices come out ascending, bit-exact with a sequential
//       filter.
#include "hark_internal.h"
int k_exclusive_scan_u32(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, int64_t *total_host);
size_t k_scan_workspace_words(int64_t n);
int k_exclusive_scan_u32_dev(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, unsigned long long *total_dev,
                             unsigned long long *sums_ws);
int k_gather(hark_context *ctx, const void *src, int esz, const uint32_t *idx, void *dst, int64_t n);

namespace {

constexpr int kMaxCols = 32;            // columns handled per launch
constexpr int kTile = 4096;             // rows per workgroup tile (256 threads x 16 rows)
constexpr int kThreads = 256;

struct ColSet {
    const void *src[kMaxCols];
    void *dst[kMaxCols];
    int32_t esz[kMaxCols];              // 4 or 8
    int32_t ncols;
};

// ---- projection: coalesced per-column copy ---------------------------------
// NTL / NTS: non-temporal loads / stores; D: 16-byte loads in flight per lane.  The default (1, 1, 2) came out of an A/B of all
// eight variants on three boxes of the pool in one process each (tools/copy_ab.py, profiles/r04_copy_ab.log); HARK_COPY_VARIANT
// = "<ntl><nts><d>" (e.g. "002") selects another one for such A/B runs.
template <bool NTL, bool NTS, int D>
__global__ __launch_bounds__(256) void copy_columns_kernel(ColSet cs, int64_t n)
{
    const int col = blockIdx.y;
    const int64_t bytes = n * cs.esz[col];
    const int64_t nvec = bytes / 16;
    const uint4 *s4 = static_cast<const uint4 *>(cs.src[col]);
    uint4 *d4 = static_cast<uint4 *>(cs.dst[col]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto ld = [&](int64_t q) -> uint4 { return NTL ? ld_nt16(s4 + q) : s4[q]; };
    auto st = [&](int64_t q, uint4 v) { if (NTS) st_nt16(d4 + q, v); else d4[q] = v; };
    for (; i + (D - 1) * stride < nvec; i += D * stride) {      // D 16-byte loads in flight per lane
        uint4 a[D];
#pragma unroll
        for (int d = 0; d < D; d++) a[d] = ld(i + d * stride);
#pragma unroll
        for (int d = 0; d < D; d++) st(i + d * stride, a[d]);
    }
    for (; i < nvec; i += stride) st(i, ld(i));
    if (blockIdx.x == 0) {                            // tail bytes (n*esz not a multiple of 16)
        const uint32_t *s1 = static_cast<const uint32_t *>(cs.src[col]);
        uint32_t *d1 = static_cast<uint32_t *>(cs.dst[col]);
        for (int64_t w = nvec * 4 + threadIdx.x; w < bytes / 4; w += blockDim.x) d1[w] = s1[w];
    }
}

// ---- predicate ----------------------------------------------------------------
union Const64 { int64_t i; float f; uint32_t u; };

template <typename T>
__device__ __forceinline__ bool cmp_val(int op, T a, T b)
{
    switch (op) {
    case HARK_CMP_GT: return a > b;
    case HARK_CMP_GE: return a >= b;
    case HARK_CMP_LT: return a < b;
    case HARK_CMP_LE: return a <= b;
    case HARK_CMP_EQ: return a == b;
    default: return a != b;
    }
}

template <typename T>
__device__ __forceinline__ T const_as(Const64 c);
template <> __device__ __forceinline__ float const_as<float>(Const64 c) { return c.f; }
template <> __device__ __forceinline__ int32_t const_as<int32_t>(Const64 c) { return (int32_t)c.i; }
template <> __device__ __forceinline__ uint32_t const_as<uint32_t>(Const64 c) { return c.u; }
template <> __device__ __forceinline__ int64_t const_as<int64_t>(Const64 c) { return c.i; }

// 16 rows per thread as 4 groups of 4 consecutive rows; group g of thread t
// covers rows tile*kTile + (g*256 + t)*4 .. +3, so a wave reads 1 KiB per load.
// Returns the 16-bit survivor mask in THREAD-LOCAL order (bit g*4+j).
template <typename T>
__device__ __forceinline__ uint32_t eval_tile(const T *__restrict__ col, int64_t n, int64_t tile, int op, T c)
{
    uint32_t mask = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int64_t r = tile * kTile + ((int64_t)g * kThreads + threadIdx.x) * 4;
        if (r + 4 <= n) {
            T x[4];
            if constexpr (sizeof(T) == 4) {
                uint4 q = ld_nt16(col + r);
                memcpy(x, &q, 16);
            } else {
                uint4 q0 = ld_nt16(col + r), q1 = ld_nt16(col + r + 2);
                memcpy(x, &q0, 16); memcpy(x + 2, &q1, 16);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) mask |= (uint32_t)cmp_val<T>(op, x[j], c) << (g * 4 + j);
        } else {
            for (int j = 0; j < 4; j++) if (r + j < n) mask |= (uint32_t)cmp_val<T>(op, col[r + j], c) << (g * 4 + j);
        }
    }
    return mask;
}

// Pass 1: evaluate the predicate once.  Every thread keeps the 16-bit survivor
// mask of its 16 rows in `masks` (2 B per 16 rows = 0.125 B/row, so pass 2 never
// re-reads the predicate column) and the tile's survivor count goes to `counts`.
template <typename T>
__global__ __launch_bounds__(kThreads) void filter_count_kernel(const T *__restrict__ col, int64_t n, int op, Const64 c,
                                                                uint16_t *__restrict__ masks, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    const uint32_t mask = eval_tile<T>(col, n, tile, op, const_as<T>(c));
    masks[tile * kThreads + threadIdx.x] = (uint16_t)mask;
    uint32_t cnt = __popc(mask);
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) counts[tile] = s_cnt;
}

// Further conjuncts of a WHERE: AND this predicate into the masks pass 1 left and recount (a thread whose 16 rows
// are all gone already does not read the column again).
template <typename T>
__global__ __launch_bounds__(kThreads) void filter_and_kernel(const T *__restrict__ col, int64_t n, int op, Const64 c,
                                                              uint16_t *__restrict__ masks, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    uint32_t mask = masks[tile * kThreads + threadIdx.x];
    if (mask) {
        mask &= eval_tile<T>(col, n, tile, op, const_as<T>(c));
        masks[tile * kThreads + threadIdx.x] = (uint16_t)mask;
    }
    uint32_t cnt = __popc(mask);
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) counts[tile] = s_cnt;
}

// A conjunct that IS a linear survivor bitmask already (HARK_CMP_MASK: a predicate tree evaluated by hark_op_predicate_tree):
// the thread's 16 bits are four nibbles of it (r is a multiple of 4).  FIRST: it starts the tile masks, else it is ANDed in.
template <bool FIRST>
__global__ __launch_bounds__(kThreads) void filter_linear_mask_kernel(const uint8_t *__restrict__ lin, int64_t n, uint16_t *__restrict__ masks, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    uint32_t mask = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int64_t r = tile * kTile + ((int64_t)g * kThreads + threadIdx.x) * 4;
        if (r < n) mask |= ((uint32_t)(lin[r >> 3] >> (r & 4)) & 15u) << (g * 4);          // (bits past row n - 1 are zero in the linear mask)
    }
    if (!FIRST) mask &= masks[tile * kThreads + threadIdx.x];
    masks[tile * kThreads + threadIdx.x] = (uint16_t)mask;
    uint32_t cnt = __popc(mask);
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) counts[tile] = s_cnt;
}

// Linear survivor bitmask for the fused group-by kernels (k_fgb.hip, HARK_CMP_MASK): bit (r & 7) of byte (r >> 3).
// A thread owns 8 consecutive rows = one byte; with `and_in` the byte is ANDed into what an earlier conjunct wrote.
template <typename T>
__global__ __launch_bounds__(256) void pred_bitmask_kernel(const T *__restrict__ col, int64_t n, int op, Const64 c, uint8_t *__restrict__ mask, int and_in)
{
    const T cv = const_as<T>(c);
    const int64_t nbytes = (n + 7) / 8, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes; b += stride) {
        const int64_t r = b * 8;
        uint32_t old = and_in ? mask[b] : 0xFFu, m = 0;
        if (old) {
            if (r + 8 <= n) {
                T x[8];
                if constexpr (sizeof(T) == 4) {
                    const uint4 q0 = ld_nt16(col + r), q1 = ld_nt16(col + r + 4);
                    memcpy(x, &q0, 16); memcpy(x + 4, &q1, 16);
                } else {
#pragma unroll
                    for (int h = 0; h < 4; h++) { const uint4 q = ld_nt16(col + r + 2 * h); memcpy(x + 2 * h, &q, 16); }
                }
#pragma unroll
                for (int j = 0; j < 8; j++) m |= (uint32_t)cmp_val<T>(op, x[j], cv) << j;
            } else {
                for (int j = 0; j < 8; j++) if (r + j < n) m |= (uint32_t)cmp_val<T>(op, col[r + j], cv) << j;
            }
        }
        mask[b] = (uint8_t)(m & old);
    }
}

// a leaf that compares two columns of one dtype
template <typename T>
__global__ __launch_bounds__(256) void pred_bitmask_cols_kernel(const T *__restrict__ a, const T *__restrict__ b, int64_t n, int op, uint8_t *__restrict__ mask)
{
    const int64_t nbytes = (n + 7) / 8, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t y = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; y < nbytes; y += stride) {
        const int64_t r = y * 8;
        uint32_t m = 0;
        for (int j = 0; j < 8; j++) if (r + j < n) m |= (uint32_t)cmp_val<T>(op, a[r + j], b[r + j]) << j;
        mask[y] = (uint8_t)m;
    }
}

// masks as wholes: dst = a AND b / a OR b / NOT a (rows past n - 1 stay zero).  16 bytes per thread (the blocks have the slack).
__global__ __launch_bounds__(256) void mask_combine_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int64_t n, int op)
{
    const int64_t nbytes = (n + 7) / 8, nq = (nbytes + 15) / 16, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += stride) {
        const uint4 x = reinterpret_cast<const uint4 *>(a)[q];
        uint4 y = op == 2 ? uint4{0u, 0u, 0u, 0u} : reinterpret_cast<const uint4 *>(b)[q];
        if (op == 0) y = uint4{x.x & y.x, x.y & y.y, x.z & y.z, x.w & y.w};
        else if (op == 1) y = uint4{x.x | y.x, x.y | y.y, x.z | y.z, x.w | y.w};
        else y = uint4{~x.x, ~x.y, ~x.z, ~x.w};
        if (op == 2 && (q + 1) * 16 * 8 > n) {                         // the last words of a complement: no survivors past the table's end
            uint32_t w[4] = {y.x, y.y, y.z, y.w};
            for (int k = 0; k < 4; k++) {
                const int64_t first = (q * 16 + k * 4) * 8;            // the row of this word's bit 0
                if (first >= n) w[k] = 0u; else if (first + 32 > n) w[k] &= (1u << (n - first)) - 1u;
            }
            y = uint4{w[0], w[1], w[2], w[3]};
        }
        reinterpret_cast<uint4 *>(dst)[q] = y;
    }
}

// Pass 2: rank the survivors inside the tile (popcount prefix over lanes, waves
// and the four 1024-row groups), compact each output column through LDS and
// write it with unit-stride stores starting at the tile's global offset.
__global__ __launch_bounds__(kThreads) void filter_scatter_kernel(const uint16_t *__restrict__ masks, int64_t n,
                                                                  const int64_t *__restrict__ offsets,
                                                                  int64_t *__restrict__ row_index, ColSet cs)
{
    __shared__ uint64_t s_stage[kTile];               // 32 KiB: one compacted column of the tile
    __shared__ uint32_t s_wcnt[4][4];                 // [group][wave] survivors
    const int64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t mask = masks[tile * kThreads + threadIdx.x];
    uint32_t lane_excl[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t cnt = __popc((mask >> (g * 4)) & 15u);
        uint32_t incl = cnt;
        for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        lane_excl[g] = incl - cnt;
        if (lane == 63) s_wcnt[g][wave] = incl;
    }
    __syncthreads();
    // tile-local position of this thread's first survivor in each group
    uint32_t pos[4], run = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t x = s_wcnt[g][w]; if (w < wave) before += x; total += x; }
        pos[g] = run + before + lane_excl[g];
        run += total;
    }
    const uint32_t tile_total = run;
    const int64_t out0 = offsets[tile];
    // Columns are compacted one after the other through the LDS stage; the 16-byte loads of the NEXT 4-byte column are issued
    // before the current column is written out, so that they travel under that write-out (the row-index column, which
    // loads nothing, goes first: the first data column is fetched under it).
    const int ncols = cs.ncols + (row_index ? 1 : 0);
    uint4 pre[4];
    auto prefetch = [&](int c) -> bool {
        if (c < 0 || c >= cs.ncols || cs.esz[c] != 4) return false;
        const uint32_t *src = static_cast<const uint32_t *>(cs.src[c]);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int64_t r = tile * kTile + ((int64_t)g * kThreads + threadIdx.x) * 4;
            pre[g] = (((mask >> (g * 4)) & 15u) && r + 4 <= n) ? ld_nt16(src + r) : uint4{0u, 0u, 0u, 0u};
        }
        return true;
    };
    bool have = prefetch(0);
    for (int k = 0; k < ncols; k++) {
        const int cidx = row_index ? (k == 0 ? cs.ncols : k - 1) : k;      // the index first
        const bool is_index = row_index && cidx == cs.ncols;
        const int esz = is_index ? 8 : cs.esz[cidx];
        const int next = row_index ? k : k + 1;                             // the data column after this one
        uint32_t *s32 = reinterpret_cast<uint32_t *>(s_stage);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t m4 = (mask >> (g * 4)) & 15u;
            if (!m4) continue;
            const int64_t r = tile * kTile + ((int64_t)g * kThreads + threadIdx.x) * 4;
            uint32_t q = pos[g];
            if (is_index) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (m4 & (1u << j)) s_stage[q++] = (uint64_t)(r + j);
            } else if (esz == 4) {
                const uint32_t *src = static_cast<const uint32_t *>(cs.src[cidx]);
                uint32_t x[4];
                if (r + 4 <= n) { const uint4 v4 = have ? pre[g] : ld_nt16(src + r); x[0] = v4.x; x[1] = v4.y; x[2] = v4.z; x[3] = v4.w; }
                else for (int j = 0; j < 4; j++) x[j] = r + j < n ? src[r + j] : 0u;
#pragma unroll
                for (int j = 0; j < 4; j++) if (m4 & (1u << j)) s32[q++] = x[j];
            } else {
                const uint64_t *src = static_cast<const uint64_t *>(cs.src[cidx]);
#pragma unroll
                for (int j = 0; j < 4; j++) if (m4 & (1u << j)) s_stage[q++] = src[r + j];
            }
        }
        lds_barrier();
        have = is_index ? have : prefetch(next);                            // (under the index column the first data column is already on its way)
        if (esz == 4) {
            uint32_t *dst = static_cast<uint32_t *>(cs.dst[cidx]) + out0;
            for (uint32_t i = threadIdx.x; i < tile_total; i += kThreads) dst[i] = s32[i];
        } else {
            uint64_t *dst = (is_index ? reinterpret_cast<uint64_t *>(row_index) : static_cast<uint64_t *>(cs.dst[cidx])) + out0;
            for (uint32_t i = threadIdx.x; i < tile_total; i += kThreads) dst[i] = s_stage[i];
        }
        lds_barrier();
    }
}

// ---- row-major copy-out (futhark_values_*_2d) -----------------------------------
// out[r*m + j] = column j, converted to 4-byte or 8-byte integers.
__global__ __launch_bounds__(256) void interleave_kernel(ColSet cs, int64_t n, int out_esz, int sign_extend_mask, void *__restrict__ out)
{
    const int m = cs.ncols;
    const int64_t total = n * m;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / m; const int j = (int)(i - r * m);
        if (out_esz == 4) {
            uint32_t x = cs.esz[j] == 4 ? static_cast<const uint32_t *>(cs.src[j])[r] : (uint32_t)static_cast<const uint64_t *>(cs.src[j])[r];
            static_cast<uint32_t *>(out)[i] = x;
        } else {
            uint64_t x;
            if (cs.esz[j] == 8) x = static_cast<const uint64_t *>(cs.src[j])[r];
            else {
                uint32_t w = static_cast<const uint32_t *>(cs.src[j])[r];
                x = (sign_extend_mask >> j) & 1 ? (uint64_t)(int64_t)(int32_t)w : (uint64_t)w;
            }
            static_cast<uint64_t *>(out)[i] = x;
        }
    }
}

// out[r*m + j] = column sel[j] converted to the matrix's element type (what numpy's result_type of the selected columns
// gives: i32 / u32 / f32 as they are, mixed integers -> i64, anything with f32 and an integer or an i64 -> f64).
// One thread per output ROW: a row's m elements are stored side by side (m is small: a select list).
struct MatCols { const void *src[kMaxCols]; int32_t dtype[kMaxCols]; int32_t ncols; };
__global__ __launch_bounds__(256) void matrix_kernel(MatCols mc, int64_t n, int out_dtype, void *__restrict__ out, int row_stride /* elements of a matrix row (the launch's columns are a stripe of it) */)
{
    const int m = mc.ncols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        for (int j = 0; j < m; j++) {
            const int d = mc.dtype[j];
            if (out_dtype == HARK_F64) {
                double x;
                if (d == HARK_F32) x = (double)static_cast<const float *>(mc.src[j])[r];
                else if (d == HARK_I32) x = (double)static_cast<const int32_t *>(mc.src[j])[r];
                else if (d == HARK_U32) x = (double)static_cast<const uint32_t *>(mc.src[j])[r];
                else x = (double)static_cast<const int64_t *>(mc.src[j])[r];
                static_cast<double *>(out)[r * row_stride + j] = x;
            } else if (out_dtype == HARK_I64) {
                int64_t x;
                if (d == HARK_I32) x = (int64_t)static_cast<const int32_t *>(mc.src[j])[r];
                else if (d == HARK_U32) x = (int64_t)static_cast<const uint32_t *>(mc.src[j])[r];
                else x = static_cast<const int64_t *>(mc.src[j])[r];
                static_cast<int64_t *>(out)[r * row_stride + j] = x;
            } else {                                                // i32 / u32 / f32 matrices of columns of that very type: bit copies
                static_cast<uint32_t *>(out)[r * row_stride + j] = static_cast<const uint32_t *>(mc.src[j])[r];
            }
        }
    }
}

void result_release(hark_context *ctx, hark_result *r)
{
    for (auto &c : r->cols) if (c.owned && c.data) hark_free(ctx, c.data);
    hark_result_host_release(ctx, r);
    delete r;
}

int check_cols(hark_context *ctx, const hark_table *db, const int32_t *cols, int64_t k, const char *who)
{
    if (k < 0 || (k && !cols)) return hark_fail(ctx, HARK_EARG, "%s: bad column list", who);
    for (int64_t j = 0; j < k; j++)
        if (cols[j] < 0 || cols[j] >= db->m)
            return hark_fail(ctx, HARK_EBOUNDS, "%s: index %d out of bounds for a table with %lld columns", who, cols[j], (long long)db->m);
    return HARK_OK;
}

// ---- HAVING + ORDER BY + LIMIT k in two small kernels (top-k) -------------------------------------------------
// A G-row aggregation result that is filtered, fully radix-sorted and then cut to its first k rows costs a compaction
// (3 kernels + 2 host reads), 4 radix passes (12 kernels) and a gather per column -- 0.25 ms of launches for a statement
// whose scans take 1.7 ms.  For k <= kTopK the k best rows are SELECTED instead: every workgroup scans a slice, keeps
// the rows that pass the predicates as (order key, row) pairs and extracts its k smallest by repeated workgroup-wide
// minimum; one workgroup then extracts the k smallest of all candidates in order.  Order and ties are exactly the
// stable sort's: ascending 64-bit order key (the sort word of the column, inverted for DESC), then ascending row.
constexpr int kTopK = 64, kTopThreads = 256, kTopRows = 16;                  // rows per thread and slice pass
struct TopPreds { const void *col[8]; int dtype[8], cmp[8]; Const64 c[8]; int n; };
constexpr int kTopMaskType = 1000;                                           // "dtype" of a predicate that is a linear survivor bitmask (HARK_CMP_MASK)

__device__ __forceinline__ uint64_t order_key(const void *col, int dtype, int64_t r, uint64_t inv)
{
    uint64_t k;
    if (dtype == HARK_I64) k = static_cast<const uint64_t *>(col)[r] ^ 0x8000000000000000ull;
    else {
        uint32_t w = static_cast<const uint32_t *>(col)[r];
        if (dtype == HARK_I32) w ^= 0x80000000u;
        else if (dtype == HARK_F32) {
            if (w == 0x80000000u) w = 0u;
            if ((w & 0x7FFFFFFFu) > 0x7F800000u) w = 0xFFFFFFFFu;
            else w ^= (w & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
        }
        k = w;
        if (inv) return (uint64_t)(w ^ 0xFFFFFFFFu);                          // DESC flips the 32-bit word, as the radix sort's mask does
    }
    return k ^ inv;
}

// (order key, row) pairs compare lexicographically; row = ~0 marks "nothing"
struct TopPair { uint64_t key; uint64_t row; };
__device__ __forceinline__ bool pair_lt(TopPair a, TopPair b) { return a.key < b.key || (a.key == b.key && a.row < b.row); }
// stage 1: a slice of at most kTopThreads * kTopRows rows -> its k smallest (key, row) pairs, in order, at
// cand[block * k ..]; fewer than k: padded with row = ~0.  A thread evaluates the predicates and order keys of its
// kTopRows rows ONCE into registers (a bit per row says whether it is still in the race); the k rounds then compare
// registers only.
__global__ __launch_bounds__(kTopThreads) void topk_slice_kernel(TopPreds pr, const void *ocol, int odtype, uint64_t inv, int64_t n, int k,
                                                                 TopPair *__restrict__ cand)
{
    __shared__ TopPair s_w[kTopThreads / 64];
    const int64_t lo = (int64_t)blockIdx.x * ((int64_t)kTopThreads * kTopRows);
    const TopPair none{~0ull, ~0ull};
    uint64_t key[kTopRows];
    uint32_t alive = 0;
    // Every load below is unconditional (rows past the end read row n - 1) and the loops run predicate by predicate with
    // the dtype switch OUTSIDE the row loop: the 16 loads of a column are independent and go out back to back.  (With
    // `r < n && row_passes(...)` per row, each row's loads waited for the previous row's verdict: 47 us per 2^20 rows.)
#pragma unroll
    for (int j = 0; j < kTopRows; j++) alive |= (lo + (int64_t)j * kTopThreads + threadIdx.x < n ? 1u : 0u) << j;
    auto row_of = [&](int j) { const int64_t r = lo + (int64_t)j * kTopThreads + threadIdx.x; return r < n ? r : n - 1; };
    for (int q = 0; q < pr.n; q++) {
        uint32_t ok = 0;
        switch (pr.dtype[q]) {
        case kTopMaskType:
#pragma unroll
            for (int j = 0; j < kTopRows; j++) { const int64_t r = row_of(j); ok |= (uint32_t)((static_cast<const uint8_t *>(pr.col[q])[r >> 3] >> (r & 7)) & 1u) << j; }
            break;
        case HARK_F32:
#pragma unroll
            for (int j = 0; j < kTopRows; j++) ok |= (cmp_val<float>(pr.cmp[q], static_cast<const float *>(pr.col[q])[row_of(j)], pr.c[q].f) ? 1u : 0u) << j;
            break;
        case HARK_I32:
#pragma unroll
            for (int j = 0; j < kTopRows; j++) ok |= (cmp_val<int32_t>(pr.cmp[q], static_cast<const int32_t *>(pr.col[q])[row_of(j)], (int32_t)pr.c[q].i) ? 1u : 0u) << j;
            break;
        case HARK_U32:
#pragma unroll
            for (int j = 0; j < kTopRows; j++) ok |= (cmp_val<uint32_t>(pr.cmp[q], static_cast<const uint32_t *>(pr.col[q])[row_of(j)], pr.c[q].u) ? 1u : 0u) << j;
            break;
        default:
#pragma unroll
            for (int j = 0; j < kTopRows; j++) ok |= (cmp_val<int64_t>(pr.cmp[q], static_cast<const int64_t *>(pr.col[q])[row_of(j)], pr.c[q].i) ? 1u : 0u) << j;
            break;
        }
        alive &= ok;
    }
#pragma unroll
    for (int j = 0; j < kTopRows; j++) key[j] = order_key(ocol, odtype, row_of(j), inv);
    int t = 0;
    for (; t < k; t++) {
        TopPair mine = none;
#pragma unroll
        for (int j = 0; j < kTopRows; j++) {
            const TopPair c{key[j], (uint64_t)(lo + (int64_t)j * kTopThreads + threadIdx.x)};
            if (((alive >> j) & 1u) && (mine.row == ~0ull || pair_lt(c, mine))) mine = c;
        }
        // (row = ~0 marks "nothing": a real pair always wins against it, whatever its key)
        TopPair best = mine;
        for (int d = 32; d > 0; d >>= 1) {
            TopPair o; o.key = __shfl_xor(best.key, d, 64); o.row = __shfl_xor(best.row, d, 64);
            if (o.row != ~0ull && (best.row == ~0ull || pair_lt(o, best))) best = o;
        }
        const int wave = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_w[wave] = best;
        __syncthreads();
        best = s_w[0];
        for (int w = 1; w < kTopThreads / 64; w++) if (s_w[w].row != ~0ull && (best.row == ~0ull || pair_lt(s_w[w], best))) best = s_w[w];
        if (best.row == ~0ull) break;
#pragma unroll
        for (int j = 0; j < kTopRows; j++) if ((uint64_t)(lo + (int64_t)j * kTopThreads + threadIdx.x) == best.row) alive &= ~(1u << j);
        if (threadIdx.x == 0) cand[(int64_t)blockIdx.x * k + t] = best;
    }
    for (int u = t + (int)threadIdx.x; u < k; u += kTopThreads) cand[(int64_t)blockIdx.x * k + u] = none;
}

// stage 2: all candidates -> the k smallest row ids in order, and how many there are.  One workgroup; a thread holds
// its candidates in registers (<= kTopCand per thread: the host bounds the number of slices accordingly).
constexpr int kTopCand = 16;
__global__ __launch_bounds__(1024) void topk_merge_kernel(const TopPair *__restrict__ cand, int64_t ncand, int k, uint32_t *__restrict__ rows_out, int64_t *__restrict__ count_out)
{
    __shared__ TopPair s_w[1024 / 64];
    const TopPair none{~0ull, ~0ull};
    TopPair mine_all[kTopCand];
#pragma unroll
    for (int j = 0; j < kTopCand; j++) { const int64_t i = (int64_t)j * blockDim.x + threadIdx.x; mine_all[j] = i < ncand ? cand[i] : none; }
    int found = 0;
    for (int t = 0; t < k; t++) {
        TopPair best = none;
#pragma unroll
        for (int j = 0; j < kTopCand; j++) if (mine_all[j].row != ~0ull && (best.row == ~0ull || pair_lt(mine_all[j], best))) best = mine_all[j];
        for (int d = 32; d > 0; d >>= 1) {
            TopPair o; o.key = __shfl_xor(best.key, d, 64); o.row = __shfl_xor(best.row, d, 64);
            if (o.row != ~0ull && (best.row == ~0ull || pair_lt(o, best))) best = o;
        }
        const int wave = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_w[wave] = best;
        __syncthreads();
        best = s_w[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) if (s_w[w].row != ~0ull && (best.row == ~0ull || pair_lt(s_w[w], best))) best = s_w[w];
        if (best.row == ~0ull) break;
#pragma unroll
        for (int j = 0; j < kTopCand; j++) if (mine_all[j].row == best.row) mine_all[j] = none;
        if (threadIdx.x == 0) rows_out[t] = (uint32_t)best.row;
        found++;
    }
    if (threadIdx.x == 0) *count_out = found;
}

} // namespace

__global__ __launch_bounds__(256) void mask_and_bytes_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t nbytes)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += stride) dst[i] &= src[i];
}

static Const64 read_const(int dtype, const void *constant)
{
    Const64 c{}; c.i = 0;
    switch (dtype) {
    case HARK_F32: c.f = *static_cast<const float *>(constant); break;
    case HARK_I32: c.i = *static_cast<const int32_t *>(constant); break;
    case HARK_U32: c.u = *static_cast<const uint32_t *>(constant); break;
    default: c.i = *static_cast<const int64_t *>(constant); break;
    }
    return c;
}

// AND of n_preds (>= 1) predicates `db[:, where_cols[j]] <cmps[j]> *constants[j]` as a linear bitmask (pool block of
// (n + 7) / 8 + 16 bytes, caller frees): one pass over each predicate column, nothing else is read.
int k_predicate_bitmask(hark_context *ctx, const hark_table *db, int64_t n_preds, const int32_t *where_cols, const int32_t *cmps,
                        const void *const *constants, uint8_t **mask_out)
{
    *mask_out = nullptr;
    const int64_t n = db->n;
    uint8_t *mask = nullptr;
    HARK_TRY(hark_alloc(ctx, (void **)&mask, (size_t)((n + 7) / 8 + 16)));
    int64_t blocks = ((n + 7) / 8 + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
    if (blocks < 1) blocks = 1;
    for (int64_t j = 0; j < n_preds; j++) {
        const int and_in = j > 0;
        dim3 grid((unsigned)blocks), block(256);
        if (cmps[j] == HARK_CMP_MASK) {                                 // a conjunct that is a mask already (a predicate tree): constants[j] IS its device address
            const uint8_t *lin = static_cast<const uint8_t *>(constants[j]);
            if (!and_in) { if (hipMemcpyAsync(mask, lin, (size_t)((n + 7) / 8), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { hark_free(ctx, mask); return hark_fail(ctx, HARK_EHIP, "predicate mask: copy failed"); } }
            else mask_and_bytes_kernel<<<grid, block, 0, ctx->stream>>>(mask, lin, (n + 7) / 8);
            if (hipGetLastError() != hipSuccess) { hark_free(ctx, mask); return hark_fail(ctx, HARK_EHIP, "predicate mask: launch failed"); }
            continue;
        }
        const int dt = db->cols[where_cols[j]].dtype;
        const void *col = db->cols[where_cols[j]].data;
        const Const64 c = read_const(dt, constants[j]);
        switch (dt) {
        case HARK_F32: pred_bitmask_kernel<float><<<grid, block, 0, ctx->stream>>>(static_cast<const float *>(col), n, cmps[j], c, mask, and_in); break;
        case HARK_I32: pred_bitmask_kernel<int32_t><<<grid, block, 0, ctx->stream>>>(static_cast<const int32_t *>(col), n, cmps[j], c, mask, and_in); break;
        case HARK_U32: pred_bitmask_kernel<uint32_t><<<grid, block, 0, ctx->stream>>>(static_cast<const uint32_t *>(col), n, cmps[j], c, mask, and_in); break;
        default: pred_bitmask_kernel<int64_t><<<grid, block, 0, ctx->stream>>>(static_cast<const int64_t *>(col), n, cmps[j], c, mask, and_in); break;
        }
        if (hipGetLastError() != hipSuccess) { hark_free(ctx, mask); return hark_fail(ctx, HARK_EHIP, "predicate mask: launch failed"); }
    }
    *mask_out = mask;
    return HARK_OK;
}

int k_gather_columns(hark_context *ctx, const hark_table *db, const int32_t *cols, int64_t k, hark_result *res)
{
    res->n = db->n;
    res->cols.resize((size_t)k);
    for (int64_t j = 0; j < k; j++) {
        res->cols[j].dtype = db->cols[cols[j]].dtype;
        HARK_TRY(hark_alloc(ctx, &res->cols[j].data, (size_t)db->n * hark_dtype_size(res->cols[j].dtype)));
    }
    if (db->n == 0) return HARK_OK;
    for (int64_t j0 = 0; j0 < k; j0 += kMaxCols) {
        ColSet cs{};
        cs.ncols = (int)((k - j0 < kMaxCols) ? k - j0 : kMaxCols);
        for (int c = 0; c < cs.ncols; c++) {
            cs.src[c] = db->cols[cols[j0 + c]].data;
            cs.dst[c] = res->cols[j0 + c].data;
            cs.esz[c] = (int32_t)hark_dtype_size(res->cols[j0 + c].dtype);
        }
        int64_t blocks = (db->n * 8 / 16 + 255) / 256 / 2 + 1;
        int64_t cap = (int64_t)ctx->num_cu * 8 / cs.ncols + 1;
        if (blocks > cap) blocks = cap;
        const char *var = getenv("HARK_COPY_VARIANT");               // A/B runs only (tools/copy_ab.py)
        const int code = (var && strlen(var) == 3) ? (var[0] - '0') * 4 + (var[1] - '0') * 2 + (var[2] == '4' ? 1 : 0) : 6;
        const dim3 grid((unsigned)blocks, (unsigned)cs.ncols), blk(256);
        switch (code) {
        case 0: copy_columns_kernel<false, false, 2><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        case 1: copy_columns_kernel<false, false, 4><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        case 2: copy_columns_kernel<false, true, 2><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        case 3: copy_columns_kernel<false, true, 4><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        case 4: copy_columns_kernel<true, false, 2><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        case 5: copy_columns_kernel<true, false, 4><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        case 7: copy_columns_kernel<true, true, 4><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        default: copy_columns_kernel<true, true, 2><<<grid, blk, 0, ctx->stream>>>(cs, db->n); break;
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    return HARK_OK;
}

extern "C" {

int hark_entry_query_sel(hark_context *ctx, hark_result **out, const hark_table *db, const int32_t *cols, int64_t k)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    // select.fut:10 `row[i]` is bounds-checked per row: with zero rows nothing is evaluated.
    if (db->n > 0) HARK_TRY(check_cols(ctx, db, cols, k, "query_sel"));
    hark_result *res = new hark_result();
    if (db->n == 0) {
        res->n = 0; res->cols.resize((size_t)k);
        for (int64_t j = 0; j < k; j++) { res->cols[j].dtype = HARK_I32; res->cols[j].data = nullptr; res->cols[j].owned = false; }
        *out = res; return HARK_OK;
    }
    bool all32 = true;
    for (int64_t j = 0; j < k; j++) all32 = all32 && hark_dtype_size(db->cols[cols[j]].dtype) == 4;
    int rc = (all32 && k_small_fits(db, k)) ? k_small_query_sel(ctx, db, cols, k, res)      // a few rows: one launch, one synchronisation (k_small.hip)
                                              : k_gather_columns(ctx, db, cols, k, res);
    if (rc) { result_release(ctx, res); return rc; }
    *out = res;
    return HARK_OK;
}

int hark_op_predicate_bitmask(hark_context *ctx, const hark_table *db, int64_t n_preds, const int32_t *where_cols,
                              const int32_t *cmps, const void *const *constants, uint8_t *mask_dev)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !db || n_preds < 1 || n_preds > 16 || !where_cols || !cmps || !constants || (db->n && !mask_dev)) return HARK_EARG;
    HARK_TRY(check_cols(ctx, db, where_cols, n_preds, "predicate_bitmask"));
    for (int64_t j = 0; j < n_preds; j++)
        if (!constants[j] || cmps[j] < HARK_CMP_GT || cmps[j] > HARK_CMP_MASK) return hark_fail(ctx, HARK_EARG, "predicate_bitmask: bad predicate %lld", (long long)j);
    if (db->n == 0) return HARK_OK;
    uint8_t *tmp = nullptr;
    HARK_TRY(k_predicate_bitmask(ctx, db, n_preds, where_cols, cmps, constants, &tmp));
    hipError_t e = hipMemcpyAsync(mask_dev, tmp, (size_t)((db->n + 7) / 8), hipMemcpyDeviceToDevice, ctx->stream);
    hark_free(ctx, tmp);                                   // stream-ordered reuse: the copy above is enqueued first
    if (e != hipSuccess) return hark_fail(ctx, HARK_EHIP, "predicate_bitmask: copy failed");
    return HARK_OK;
}

// One node of an arithmetic expression inside an aggregate (`sum(a + b)`): out[i] = x[i] <op> y[i], an operand a 4- or 8-byte column or
// a constant.  Integer operands (and no division) stay integers -- i32 when both are 32-bit (wrapping, like the reference's u32
// arithmetic, groupby.fut:35-41), i64 otherwise --, anything else is computed and stored in f32.
namespace {
struct ExprOperand { const void *col; int32_t dtype; double c; };          // col == nullptr: the constant c
__device__ __forceinline__ int64_t expr_int(const ExprOperand &o, int64_t i)
{
    if (!o.col) return (int64_t)o.c;
    return o.dtype == HARK_I64 ? static_cast<const int64_t *>(o.col)[i] : o.dtype == HARK_U32 ? (int64_t)static_cast<const uint32_t *>(o.col)[i] : (int64_t)static_cast<const int32_t *>(o.col)[i];
}
__device__ __forceinline__ float expr_f32(const ExprOperand &o, int64_t i)
{
    if (!o.col) return (float)o.c;
    return o.dtype == HARK_F32 ? static_cast<const float *>(o.col)[i] : (float)expr_int(o, i);
}
__global__ __launch_bounds__(256) void column_binary_kernel(ExprOperand x, ExprOperand y, int op, int out_dtype, void *__restrict__ out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (out_dtype == HARK_F32) {
            const float a = expr_f32(x, i), b = expr_f32(y, i);
            static_cast<float *>(out)[i] = op == 0 ? a + b : op == 1 ? a - b : op == 2 ? a * b : a / b;
        } else {
            const uint64_t a = (uint64_t)expr_int(x, i), b = (uint64_t)expr_int(y, i), r = op == 0 ? a + b : op == 1 ? a - b : a * b;   // (unsigned: wrapping is defined)
            if (out_dtype == HARK_I64) static_cast<int64_t *>(out)[i] = (int64_t)r; else static_cast<int32_t *>(out)[i] = (int32_t)(uint32_t)r;
        }
    }
}
} // namespace

int hark_op_column_binary(hark_context *ctx, int64_t n, int32_t op, const void *x, int32_t x_dtype, double x_const, const void *y, int32_t y_dtype, double y_const,
                          int32_t out_dtype, void *out_dev)
{
    hark_device_guard guard__(ctx);
    if (!ctx || n < 0 || op < 0 || op > 3 || (n && !out_dev) || (out_dtype != HARK_I32 && out_dtype != HARK_I64 && out_dtype != HARK_F32)) return HARK_EARG;
    if (op == 3 && out_dtype != HARK_F32) return hark_fail(ctx, HARK_EARG, "column_binary: a division is computed in f32");
    if (n == 0) return HARK_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
    column_binary_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(ExprOperand{x, x_dtype, x_const}, ExprOperand{y, y_dtype, y_const}, op, out_dtype, out_dev, n);
    if (hipGetLastError() != hipSuccess) return hark_fail(ctx, HARK_EHIP, "column_binary: launch failed");
    return HARK_OK;
}

// A predicate TREE as a survivor bitmask: the nodes in postfix order, evaluated with a stack of masks.
//   kind[i] = HARK_PRED_CONST: db[:, a[i]] <b[i]> *constants[i]      (b: HARK_CMP_GT .. HARK_CMP_NE; the constant read as the column's dtype)
//             HARK_PRED_COLS : db[:, a[i]] <b[i] & 15> db[:, b[i] >> 4]   (two columns of ONE dtype)
//             HARK_PRED_AND / HARK_PRED_OR: the two masks on top of the stack;  HARK_PRED_NOT: the one on top
// One pass over a leaf's column(s) per leaf, 0.125 B/row per inner node.  The result ((n + 7) / 8 bytes at mask_dev, bits past row
// n - 1 zero) is a conjunct like any other: cmp = HARK_CMP_MASK, the constant pointer = mask_dev (hark_entry_filter_sel_and,
// hark_entry_filter_groupby*, hark_entry_topk, hark_op_predicate_bitmask).
int hark_op_predicate_tree(hark_context *ctx, const hark_table *db, int64_t n_nodes, const int32_t *kind, const int32_t *a, const int32_t *b,
                           const void *const *constants, uint8_t *mask_dev)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !db || n_nodes < 1 || n_nodes > 256 || !kind || !a || !b || !constants || (db->n && !mask_dev)) return HARK_EARG;
    const int64_t n = db->n;
    if (n == 0) return HARK_OK;
    const size_t bytes = (size_t)((n + 7) / 8 + 16);
    int64_t blocks = ((n + 7) / 8 + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(256);
    std::vector<uint8_t *> stack;
    int rc = HARK_OK;
    auto fail = [&](int code, const char *what) { if (!rc) rc = hark_fail(ctx, code, "predicate_tree: %s", what); };
    for (int64_t i = 0; i < n_nodes && !rc; i++) {
        if (kind[i] == HARK_PRED_CONST || kind[i] == HARK_PRED_COLS) {
            const int ca = a[i], op = kind[i] == HARK_PRED_CONST ? b[i] : (b[i] & 15), cb = kind[i] == HARK_PRED_COLS ? (b[i] >> 4) : 0;
            if (ca < 0 || ca >= db->m || cb < 0 || cb >= db->m) { fail(HARK_EBOUNDS, "column out of bounds"); break; }
            if (op < HARK_CMP_GT || op > HARK_CMP_NE) { fail(HARK_EARG, "unknown comparison"); break; }
            uint8_t *m = nullptr;
            rc = hark_alloc(ctx, (void **)&m, bytes);
            if (rc) break;
            stack.push_back(m);
            const int dt = db->cols[ca].dtype;
            const void *col = db->cols[ca].data;
            if (kind[i] == HARK_PRED_CONST) {
                if (!constants[i]) { fail(HARK_EARG, "a leaf without its constant"); break; }
                const Const64 c = read_const(dt, constants[i]);
                switch (dt) {
                case HARK_F32: pred_bitmask_kernel<float><<<grid, block, 0, ctx->stream>>>(static_cast<const float *>(col), n, op, c, m, 0); break;
                case HARK_I32: pred_bitmask_kernel<int32_t><<<grid, block, 0, ctx->stream>>>(static_cast<const int32_t *>(col), n, op, c, m, 0); break;
                case HARK_U32: pred_bitmask_kernel<uint32_t><<<grid, block, 0, ctx->stream>>>(static_cast<const uint32_t *>(col), n, op, c, m, 0); break;
                default: pred_bitmask_kernel<int64_t><<<grid, block, 0, ctx->stream>>>(static_cast<const int64_t *>(col), n, op, c, m, 0); break;
                }
            } else {
                if (db->cols[cb].dtype != dt) { fail(HARK_EUNSUPPORTED, "a comparison of two columns needs one dtype"); break; }
                const void *col2 = db->cols[cb].data;
                switch (dt) {
                case HARK_F32: pred_bitmask_cols_kernel<float><<<grid, block, 0, ctx->stream>>>(static_cast<const float *>(col), static_cast<const float *>(col2), n, op, m); break;
                case HARK_I32: pred_bitmask_cols_kernel<int32_t><<<grid, block, 0, ctx->stream>>>(static_cast<const int32_t *>(col), static_cast<const int32_t *>(col2), n, op, m); break;
                case HARK_U32: pred_bitmask_cols_kernel<uint32_t><<<grid, block, 0, ctx->stream>>>(static_cast<const uint32_t *>(col), static_cast<const uint32_t *>(col2), n, op, m); break;
                default: pred_bitmask_cols_kernel<int64_t><<<grid, block, 0, ctx->stream>>>(static_cast<const int64_t *>(col), static_cast<const int64_t *>(col2), n, op, m); break;
                }
            }
        } else if (kind[i] == HARK_PRED_AND || kind[i] == HARK_PRED_OR) {
            if (stack.size() < 2) { fail(HARK_EARG, "AND / OR with fewer than two masks on the stack"); break; }
            uint8_t *y = stack.back(); stack.pop_back();
            mask_combine_kernel<<<grid, block, 0, ctx->stream>>>(stack.back(), stack.back(), y, n, kind[i] == HARK_PRED_AND ? 0 : 1);
            hark_free(ctx, y);                                        // stream-ordered reuse: the kernel above is enqueued first
        } else if (kind[i] == HARK_PRED_NOT) {
            if (stack.empty()) { fail(HARK_EARG, "NOT with an empty stack"); break; }
            mask_combine_kernel<<<grid, block, 0, ctx->stream>>>(stack.back(), stack.back(), stack.back(), n, 2);
        } else fail(HARK_EARG, "unknown node kind");
        if (!rc && hipGetLastError() != hipSuccess) fail(HARK_EHIP, "launch failed");
    }
    if (!rc && stack.size() != 1) fail(HARK_EARG, "the nodes do not reduce to one mask");
    if (!rc && hipMemcpyAsync(mask_dev, stack.back(), (size_t)((n + 7) / 8), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) fail(HARK_EHIP, "copy failed");
    for (uint8_t *m : stack) hark_free(ctx, m);
    return rc;
}

int hark_entry_filter_sel(hark_context *ctx, hark_result **out, const hark_table *db, int32_t where_col, int32_t cmp,
                          const void *constant, const int32_t *cols, int64_t k, int32_t want_row_index)
{
    return hark_entry_filter_sel_and(ctx, out, db, 1, &where_col, &cmp, &constant, cols, k, want_row_index);
}

// HAVING (n_preds >= 0 predicates, AND) + ORDER BY key_col [DESC] + LIMIT k over a table, k <= 64, at most 2^32 rows: the
// first k rows of what hark_entry_filter_sel_and + hark_entry_sort would deliver (same order, same ties), columns `cols`.
int hark_entry_topk(hark_context *ctx, hark_result **out, const hark_table *db, int64_t n_preds, const int32_t *where_cols,
                    const int32_t *cmps, const void *const *constants, int32_t key_col, int32_t descending, int64_t k,
                    const int32_t *cols, int64_t ncols)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (n_preds < 0 || n_preds > 8 || (n_preds && (!where_cols || !cmps || !constants)) || k < 1 || k > kTopK || ncols < 0 || ncols > kMaxCols || db->n > 0xFFFFFFFFll)
        return hark_fail(ctx, HARK_EUNSUPPORTED, "topk: 0..8 predicates, 1 <= k <= %d, at most %d columns", kTopK, kMaxCols);
    if (key_col < 0 || key_col >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "topk: key column %d out of bounds", key_col);
    HARK_TRY(check_cols(ctx, db, cols, ncols, "topk"));
    HARK_TRY(check_cols(ctx, db, where_cols, n_preds, "topk(where)"));
    TopPreds pr{}; pr.n = (int)n_preds;
    for (int64_t j = 0; j < n_preds; j++) {
        if (!constants[j] || cmps[j] < HARK_CMP_GT || cmps[j] > HARK_CMP_MASK) return hark_fail(ctx, HARK_EARG, "topk: bad predicate %lld", (long long)j);
        if (cmps[j] == HARK_CMP_MASK) { pr.col[j] = constants[j]; pr.dtype[j] = kTopMaskType; pr.cmp[j] = cmps[j]; continue; }   // a linear survivor bitmask
        pr.col[j] = db->cols[where_cols[j]].data; pr.dtype[j] = db->cols[where_cols[j]].dtype; pr.cmp[j] = cmps[j];
        pr.c[j] = read_const(pr.dtype[j], constants[j]);
    }
    hark_result *res = new hark_result();
    res->cols.resize((size_t)ncols);
    for (int64_t j = 0; j < ncols; j++) { res->cols[(size_t)j].dtype = db->cols[cols[j]].dtype; res->cols[(size_t)j].data = nullptr; res->cols[(size_t)j].owned = true; }
    res->n = 0;
    const int64_t n = db->n;
    if (n == 0) { for (auto &c : res->cols) c.owned = false; *out = res; return HARK_OK; }
    const int64_t nblk = (n + (int64_t)kTopThreads * kTopRows - 1) / ((int64_t)kTopThreads * kTopRows);      // slices of 4096 rows
    if (nblk * k > (int64_t)kTopCand * 1024) {                             // the merge stage holds every candidate in registers
        for (auto &c : res->cols) c.owned = false;
        delete res;
        return hark_fail(ctx, HARK_EUNSUPPORTED, "topk: at most %lld rows for k = %lld", (long long)((int64_t)kTopCand * 1024 / k * kTopThreads * kTopRows), (long long)k);
    }
    TopPair *cand = nullptr; uint32_t *rows = nullptr; int64_t *count = nullptr;
    int rc = hark_alloc(ctx, (void **)&cand, (size_t)nblk * (size_t)k * sizeof(TopPair));
    if (!rc) rc = hark_alloc(ctx, (void **)&rows, (size_t)k * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&count, 16);
    int64_t found = 0;
    if (!rc) {
        hipStream_t st = ctx->stream;
        const uint64_t inv = descending ? ~0ull : 0ull;
        topk_slice_kernel<<<dim3((unsigned)nblk), dim3(kTopThreads), 0, st>>>(pr, db->cols[key_col].data, db->cols[key_col].dtype, inv, n, (int)k, cand);
        // a few thousand candidates (2^20 rows, k = 10: 2560) are merged by four waves: every round ends in a barrier
        // and a scan over one partial result per wave
        const int merge_threads = nblk * k <= (int64_t)kTopCand * 256 ? 256 : 1024;
        topk_merge_kernel<<<1, merge_threads, 0, st>>>(cand, nblk * k, (int)k, rows, count);
        if (hipGetLastError() != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "topk: launch failed");
        if (!rc) rc = hark_read_words(ctx, count, &found, 1);
    }
    if (!rc) {
        res->n = found;
        for (int64_t j = 0; j < ncols && !rc && found > 0; j++) {
            const int esz = (int)hark_dtype_size(res->cols[(size_t)j].dtype);
            rc = hark_alloc(ctx, &res->cols[(size_t)j].data, (size_t)found * esz);
            if (!rc) rc = k_gather(ctx, db->cols[cols[j]].data, esz, rows, res->cols[(size_t)j].data, found);
        }
        if (found == 0) for (auto &c : res->cols) c.owned = false;
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "topk: kernels failed");
    }
    hark_free(ctx, cand); hark_free(ctx, rows); hark_free(ctx, count);
    if (rc) { for (auto &c : res->cols) if (c.owned && c.data) hark_free(ctx, c.data); delete res; return rc; }
    *out = res;
    return HARK_OK;
}

int hark_entry_filter_sel_and(hark_context *ctx, hark_result **out, const hark_table *db, int64_t n_preds, const int32_t *where_cols,
                              const int32_t *cmps, const void *const *constants, const int32_t *cols, int64_t k, int32_t want_row_index)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (n_preds < 1 || n_preds > 16 || !where_cols || !cmps || !constants) return hark_fail(ctx, HARK_EARG, "filter_sel: 1..16 predicates");
    HARK_TRY(check_cols(ctx, db, cols, k, "filter_sel"));
    HARK_TRY(check_cols(ctx, db, where_cols, n_preds, "filter_sel(where)"));
    for (int64_t j = 0; j < n_preds; j++) {
        if (!constants[j]) return HARK_EARG;
        if (cmps[j] < HARK_CMP_GT || cmps[j] > HARK_CMP_MASK) return hark_fail(ctx, HARK_EARG, "filter_sel: unknown comparison %d", cmps[j]);
    }
    if (k > kMaxCols) return hark_fail(ctx, HARK_EUNSUPPORTED, "filter_sel: at most %d projected columns", kMaxCols);
    const int64_t n = db->n;
    const int64_t ntiles = (n + kTile - 1) / kTile;
    if (ntiles > 0x7FFFFFFF) return hark_fail(ctx, HARK_EARG, "filter_sel: table too large for one call");
    hark_result *res = new hark_result();
    const int extra = want_row_index ? 1 : 0;
    res->cols.resize((size_t)(k + extra));
    if (extra) res->cols[0].dtype = HARK_I64;
    for (int64_t j = 0; j < k; j++) res->cols[j + extra].dtype = db->cols[cols[j]].dtype;
    int64_t total = 0;
    uint32_t *counts = nullptr; int64_t *offsets = nullptr; uint16_t *masks = nullptr;
    int rc = HARK_OK;
    // When the result can be allocated for the worst case cheaply (every row survives: at most 4 GiB in all), count, scan and
    // scatter are enqueued back to back and the survivor count is read ONCE, at the end: the host round trip between the
    // scan and the scatter (a synchronisation, the allocations, a launch: ~25 us of a 0.45 ms statement) is gone.
    size_t row_bytes = 0;
    for (auto &col : res->cols) row_bytes += hark_dtype_size(col.dtype);
    const bool speculative = n > 0 && (size_t)n * row_bytes <= ((size_t)4 << 30) && !getenv("HARK_FILTER_NO_SPEC");
    const size_t scan_words = speculative ? k_scan_workspace_words(ntiles) : 0;
    if (n > 0) {
        rc = hark_alloc(ctx, (void **)&counts, (size_t)ntiles * sizeof(uint32_t));
        if (!rc) rc = hark_alloc(ctx, (void **)&offsets, (size_t)(ntiles + 2 + scan_words) * sizeof(int64_t));   // + the total, + scan scratch
        if (!rc) rc = hark_alloc(ctx, (void **)&masks, (size_t)ntiles * kThreads * sizeof(uint16_t));
        if (!rc) {
            hipStream_t st = ctx->stream;
            dim3 grid((unsigned)ntiles), block(kThreads);
            for (int64_t j = 0; j < n_preds; j++) {                     // conjunct 0 writes the masks, the others AND into them
                if (cmps[j] == HARK_CMP_MASK) {                          // a linear survivor bitmask (a predicate tree): constants[j] is its device address
                    if (j == 0) filter_linear_mask_kernel<true><<<grid, block, 0, st>>>(static_cast<const uint8_t *>(constants[j]), n, masks, counts);
                    else filter_linear_mask_kernel<false><<<grid, block, 0, st>>>(static_cast<const uint8_t *>(constants[j]), n, masks, counts);
                    continue;
                }
                const int wdt = db->cols[where_cols[j]].dtype;
                const void *wc = db->cols[where_cols[j]].data;
                const Const64 c = read_const(wdt, constants[j]);
                const int cmp = cmps[j];
#define HARK_FILTER_LAUNCH(T) do { if (j == 0) filter_count_kernel<T><<<grid, block, 0, st>>>(static_cast<const T *>(wc), n, cmp, c, masks, counts); \
                                   else filter_and_kernel<T><<<grid, block, 0, st>>>(static_cast<const T *>(wc), n, cmp, c, masks, counts); } while (0)
                switch (wdt) {
                case HARK_F32: HARK_FILTER_LAUNCH(float); break;
                case HARK_I32: HARK_FILTER_LAUNCH(int32_t); break;
                case HARK_U32: HARK_FILTER_LAUNCH(uint32_t); break;
                default: HARK_FILTER_LAUNCH(int64_t); break;
                }
#undef HARK_FILTER_LAUNCH
            }
            if (hipGetLastError() != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "filter_sel: launch failed");
            if (!rc && speculative) rc = k_exclusive_scan_u32_dev(ctx, counts, ntiles, nullptr, offsets, reinterpret_cast<unsigned long long *>(offsets + ntiles),
                                                                 reinterpret_cast<unsigned long long *>(offsets + ntiles + 2));
            else if (!rc) rc = k_exclusive_scan_u32(ctx, counts, ntiles, nullptr, offsets, &total);
        }
    }
    if (!rc) {
        res->n = total;
        for (auto &col : res->cols) {
            rc = hark_alloc(ctx, &col.data, (size_t)(speculative ? n : total) * hark_dtype_size(col.dtype));
            if (rc) break;
        }
    }
    if (!rc && (total > 0 || speculative)) {
        ColSet cs{};
        cs.ncols = (int)k;
        for (int64_t j = 0; j < k; j++) {
            cs.src[j] = db->cols[cols[j]].data; cs.dst[j] = res->cols[j + extra].data;
            cs.esz[j] = (int32_t)hark_dtype_size(res->cols[j + extra].dtype);
        }
        int64_t *ridx = extra ? static_cast<int64_t *>(res->cols[0].data) : nullptr;
        filter_scatter_kernel<<<dim3((unsigned)ntiles), dim3(kThreads), 0, ctx->stream>>>(masks, n, offsets, ridx, cs);
        if (hipGetLastError() != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "filter_sel: launch failed");
        if (!rc && speculative) { rc = hark_read_words(ctx, offsets + ntiles, &total, 1); res->n = total; }     // (drains the stream)
        else if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "filter_sel: kernel failed");
    }
    hark_free(ctx, counts); hark_free(ctx, offsets); hark_free(ctx, masks);
    if (rc) { result_release(ctx, res); return rc; }
    *out = res;
    return HARK_OK;
}

int hark_result_values_2d(hark_context *ctx, const hark_result *r, void *host_out, int dtype)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !r) return HARK_EARG;
    const int64_t m = (int64_t)r->cols.size();
    if (r->n == 0 || m == 0) return HARK_OK;
    if (!host_out) return HARK_EARG;
    if (m > kMaxCols) return hark_fail(ctx, HARK_EUNSUPPORTED, "values_2d: at most %d columns", kMaxCols);
    const int out_esz = (int)hark_dtype_size(dtype);
    ColSet cs{}; cs.ncols = (int)m;
    int sext = 0;
    for (int64_t j = 0; j < m; j++) {
        const int d = r->cols[j].dtype;
        if ((d == HARK_F32) != (dtype == HARK_F32))
            return hark_fail(ctx, HARK_EARG, "values_2d: cannot mix f32 and integer columns in one matrix");
        cs.src[j] = r->cols[j].data; cs.esz[j] = (int32_t)hark_dtype_size(d);
        if (d == HARK_I32) sext |= 1 << j;
    }
    void *tmp = nullptr;
    const size_t bytes = (size_t)r->n * (size_t)m * (size_t)out_esz;
    HARK_TRY(hark_alloc(ctx, &tmp, bytes));
    int64_t blocks = (r->n * m + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 8) blocks = (int64_t)ctx->num_cu * 8;
    interleave_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(cs, r->n, out_esz, sext, tmp);
    hipError_t e = hipGetLastError();
    int rc = e == hipSuccess ? hark_d2h(ctx, host_out, tmp, bytes) : hark_fail(ctx, HARK_EHIP, "values_2d: %s", hipGetErrorString(e));
    hark_free(ctx, tmp);
    return rc;
}

// The reference's result shape -- ONE row-major [rows][k] matrix (from_futhark, FutharkContext.py:66,71) -- of the selected
// result columns (repeats allowed), built on the device and copied once into a pinned host block the caller owns afterwards
// (hark_host_free).  out_dtype: HARK_I32 / HARK_U32 / HARK_F32 when every selected column has that type; HARK_I64 for integer
// columns; HARK_F64 for any mix.  (The host side used to download typed columns and interleave them with strided numpy stores:
// 3.4 x the kernels' time for a 2^20-group result.)
int hark_result_matrix_pinned(hark_context *ctx, const hark_result *r, const int32_t *cols, int64_t k, int64_t rows, int out_dtype, void **host_block)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !r || !host_block || k < 0 || (k && !cols) || rows < 0) return HARK_EARG;
    *host_block = nullptr;
    if (rows > r->n) rows = r->n;
    if (out_dtype < HARK_I32 || out_dtype > HARK_F64) return hark_fail(ctx, HARK_EARG, "result_matrix: bad element type %d", out_dtype);
    if (r->host_matrix && rows == r->host_rows && k == r->host_cols && k == (int64_t)r->cols.size() && (out_dtype == HARK_I32 || out_dtype == HARK_U32)) {
        bool as_is = true;                                       // the small-table paths wrote this very matrix already (k_small.hip)
        for (int64_t j = 0; j < k; j++) as_is = as_is && cols[j] == (int32_t)j && (r->cols[j].dtype == HARK_I32 || r->cols[j].dtype == HARK_U32);
        if (as_is) { *host_block = r->host_matrix; const_cast<hark_result *>(r)->host_matrix = nullptr; return HARK_OK; }
    }
    for (int64_t j = 0; j < k; j++) {
        if (cols[j] < 0 || cols[j] >= (int64_t)r->cols.size()) return hark_fail(ctx, HARK_EBOUNDS, "result_matrix: column %d of %zu", cols[j], r->cols.size());
        const int d = r->cols[cols[j]].dtype;
        const bool ok = out_dtype == HARK_F64 || (out_dtype == HARK_I64 && d != HARK_F32) || d == out_dtype ||
                        ((out_dtype == HARK_I32 || out_dtype == HARK_U32) && (d == HARK_I32 || d == HARK_U32));   // the reference's u32 view of i32 columns: bit copies
        if (!ok) return hark_fail(ctx, HARK_EARG, "result_matrix: a column of type %d does not convert to a matrix of type %d", d, out_dtype);
    }
    const size_t esz = (out_dtype == HARK_I64 || out_dtype == HARK_F64) ? 8 : 4;
    const size_t bytes = (size_t)rows * (size_t)k * esz;
    if (!bytes) return HARK_OK;
    // a matrix of more than 2 GiB is not pinned (nor doubled in device scratch): the caller takes typed columns and interleaves on the host
    if (bytes > ((size_t)2 << 30)) return hark_fail(ctx, HARK_EUNSUPPORTED, "result_matrix: %zu bytes exceed the pinned-block limit", bytes);
    void *tmp = nullptr, *blk = nullptr;
    HARK_TRY(hark_alloc(ctx, &tmp, bytes));
    int rc = hark_host_alloc(ctx, &blk, bytes);
    if (!rc) {
        int64_t blocks = (rows + 255) / 256;
        if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
        hipError_t e = hipSuccess;
        for (int64_t c0 = 0; c0 < k && e == hipSuccess; c0 += kMaxCols) {      // kMaxCols columns per launch, each launch its own stripe of the rows
            MatCols mc{}; mc.ncols = (int)std::min<int64_t>(kMaxCols, k - c0);
            for (int j = 0; j < mc.ncols; j++) { mc.src[j] = r->cols[cols[c0 + j]].data; mc.dtype[j] = r->cols[cols[c0 + j]].dtype; }
            matrix_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(mc, rows, out_dtype, static_cast<char *>(tmp) + (size_t)c0 * esz, (int)k);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(blk, tmp, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "result_matrix: %s", hipGetErrorString(e));
    }
    hark_free(ctx, tmp);
    if (rc) { if (blk) hark_host_free(ctx, blk); return rc; }
    *host_block = blk;
    return HARK_OK;
}

} // extern "C"
