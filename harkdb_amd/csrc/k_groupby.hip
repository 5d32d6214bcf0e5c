// k_groupby.hip -- GROUP BY over arbitrary (sparse) keys.
//
//   hark_entry_query_groupby   the reference's entry (main.fut:9 ->
//       groupby.fut:51-62): u32 keys in ascending UNSIGNED order, one u32
//       column per opcode of `type_func` (groupby.fut:35-41), a leading key
//       column.  The reference materialises rows, sorts them with 32 one-bit
//       passes and runs a segmented scan with `merge`; every opcode is
//       commutative and associative on u32 (wrapping *, +, max, min), so the
//       fold order does not show in the result and any reduction tree is
//       bit-exact with the sequential fold.
//   hark_entry_filter_groupby  the SQL-typed extension (WHERE + typed keys and
//       aggregates, COUNT/AVG), with the fused dense-key kernels of k_fgb.hip
//       behind it when the query has that shape.
//
// Device plan for both: stable argsort of the key column (k_sort.hip, 4 x
// 8-bit passes over (key, row id) only) -> head flags -> prefix sum = group
// ids -> per aggregate a gather through the permutation and a wave64 segmented
// scan whose run tails are combined into 64-bit accumulators with one atomic
// per (wave, group) -> typed finalisation.
#include "hark_internal.h"
#include <algorithm>

int k_argsort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending,
                     uint32_t **perm_out, uint32_t **sorted_words_out);
int k_gather(hark_context *ctx, const void *src, int esz, const uint32_t *idx, void *dst, int64_t n);
int k_argsort_i64_keys(hark_context *ctx, const void *col, int64_t n, uint32_t **perm_out, uint64_t **keys_out, const uint32_t *valcol, uint32_t **val_out, int *unique_out, bool *plain_out = nullptr, int8_t *msd_unfit = nullptr);
int k_sort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending, const uint32_t *payload,
                  uint32_t **vals_out, uint32_t **words_out);
int k_exclusive_scan_u32(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, int64_t *total_host);

namespace {

typedef unsigned long long u64;
enum { ACC_U64 = 0, ACC_I64 = 1, ACC_F64 = 2 };
enum { OP_SUM = 0, OP_PROD = 1, OP_MAX = 2, OP_MIN = 3 };

__device__ __forceinline__ u64 d2u(double d) { return (u64)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(u64 u) { return __longlong_as_double((long long)u); }

__device__ __forceinline__ u64 combine(int kind, int op, u64 a, u64 b)
{
    if (kind == ACC_F64) {
        const double x = u2d(a), y = u2d(b);
        switch (op) {
        case OP_SUM: return d2u(x + y);
        case OP_PROD: return d2u(x * y);
        case OP_MAX: return d2u(x > y ? x : y);
        default: return d2u(x < y ? x : y);
        }
    }
    switch (op) {
    case OP_SUM: return a + b;
    case OP_PROD: return a * b;
    case OP_MAX: return kind == ACC_I64 ? ((long long)a > (long long)b ? a : b) : (a > b ? a : b);
    default: return kind == ACC_I64 ? ((long long)a < (long long)b ? a : b) : (a < b ? a : b);
    }
}

__host__ __device__ inline u64 identity_of(int kind, int op)
{
    if (kind == ACC_F64) {
        const double v = op == OP_SUM ? 0.0 : op == OP_PROD ? 1.0 : op == OP_MAX ? -__builtin_huge_val() : __builtin_huge_val();
        u64 u; memcpy(&u, &v, 8); return u;
    }
    if (op == OP_SUM) return 0;
    if (op == OP_PROD) return 1;
    if (op == OP_MAX) return kind == ACC_I64 ? 0x8000000000000000ull : 0ull;
    return kind == ACC_I64 ? 0x7FFFFFFFFFFFFFFFull : ~0ull;
}

__device__ __forceinline__ void atomic_combine(int kind, int op, u64 *dst, u64 x)
{
    if (op == OP_SUM && kind != ACC_F64) { atomicAdd(dst, x); return; }
    if (op == OP_SUM) { unsafeAtomicAdd(reinterpret_cast<double *>(dst), u2d(x)); return; }
    u64 old = *dst, assumed;
    do {
        assumed = old;
        const u64 want = combine(kind, op, assumed, x);
        if (want == assumed) break;
        old = atomicCAS(dst, assumed, want);
    } while (old != assumed);
}

// value of row `r` of a column, widened to the accumulator representation
__device__ __forceinline__ u64 load_as_acc(const void *col, int dtype, uint32_t r, int kind)
{
    if (kind == ACC_F64) {                       // SUM/AVG/MIN/MAX of f32, AVG of integers
        switch (dtype) {
        case HARK_I32: return d2u((double)static_cast<const int32_t *>(col)[r]);
        case HARK_U32: return d2u((double)static_cast<const uint32_t *>(col)[r]);
        case HARK_F32: return d2u((double)static_cast<const float *>(col)[r]);
        default: return d2u((double)static_cast<const long long *>(col)[r]);
        }
    }
    switch (dtype) {
    case HARK_I32: return (u64)(long long)static_cast<const int32_t *>(col)[r];
    case HARK_U32: return (u64)static_cast<const uint32_t *>(col)[r];
    case HARK_F32: return d2u((double)static_cast<const float *>(col)[r]);
    default: return static_cast<const u64 *>(col)[r];
    }
}

__global__ __launch_bounds__(256) void unbias_u64_kernel(uint64_t *__restrict__ keys, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) keys[i] ^= 0x8000000000000000ull;
}

__global__ __launch_bounds__(256) void fill_u64_kernel(u64 *__restrict__ dst, int64_t n, u64 v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// flags[i] = first row of a run of equal keys (groupby.fut:26-33 mk_flags), on
// the key column gathered into sorted order (4- or 8-byte elements).
// f32 keys compare as the sort orders them (k_sort.hip sort_word_of): -0.0 == +0.0 and all NaNs are one key --
// raw bits would split a run of [0.0, -0.0, 0.0] (which a stable sort keeps interleaved) into three groups.
__device__ __forceinline__ uint32_t f32_key_class(uint32_t w)
{
    if ((w & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC00000u;
    return w == 0x80000000u ? 0u : w;
}

__global__ __launch_bounds__(256) void head_flags_kernel(const void *__restrict__ sorted_keys, int esz, int is_f32, int64_t n, uint32_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        bool head = i == 0;
        if (!head) {
            if (esz == 4 && is_f32) head = f32_key_class(static_cast<const uint32_t *>(sorted_keys)[i]) != f32_key_class(static_cast<const uint32_t *>(sorted_keys)[i - 1]);
            else if (esz == 4) head = static_cast<const uint32_t *>(sorted_keys)[i] != static_cast<const uint32_t *>(sorted_keys)[i - 1];
            else head = static_cast<const u64 *>(sorted_keys)[i] != static_cast<const u64 *>(sorted_keys)[i - 1];
        }
        flags[i] = head ? 1u : 0u;
    }
}

// seg[i] = (exclusive scan of flags)[i] + flags[i] - 1, in place over the scan;
// head rows also emit the group's key.
__global__ __launch_bounds__(256) void seg_ids_kernel(uint32_t *__restrict__ seg, const uint32_t *__restrict__ flags, int64_t n,
                                                      const void *__restrict__ sorted_keys, int esz, void *__restrict__ out_keys)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t f = flags[i], s = seg[i] + f - 1u;
        seg[i] = s;
        if (f) {
            if (esz == 4) static_cast<uint32_t *>(out_keys)[s] = static_cast<const uint32_t *>(sorted_keys)[i];
            else static_cast<u64 *>(out_keys)[s] = static_cast<const u64 *>(sorted_keys)[i];
        }
    }
}

// Segmented reduction of col[perm[i]] by seg[i] (seg ascending).  count_mode:
// every row contributes 1 (COUNT).
__global__ __launch_bounds__(256) void seg_reduce_kernel(const void *__restrict__ col, int dtype, const uint32_t *__restrict__ perm,
                                                         const uint32_t *__restrict__ seg, int64_t n, int kind, int op, int count_mode,
                                                         u64 *__restrict__ acc)
{
    const int lane = threadIdx.x & 63;
    const u64 ident = identity_of(kind, op);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n;
        uint32_t s = valid ? seg[i] : 0xFFFFFFFFu;
        u64 x = ident;
        if (valid) x = count_mode ? 1ull : load_as_acc(col, dtype, perm ? perm[i] : (uint32_t)i, kind);   // perm == null: col is already in sorted order
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 y = __shfl_up(x, d, 64);
            const uint32_t sy = __shfl_up(s, d, 64);
            if (lane >= d && sy == s) x = combine(kind, op, y, x);
        }
        // A run that begins AND ends inside this wave is the whole group (seg ascends): its total is stored
        // plainly over the identity acc was filled with; only groups that cross a wave boundary need the atomic
        // (with mostly distinct keys that turns one atomic per row into contiguous 8-byte stores).
        const uint32_t sn = __shfl_down(s, 1, 64), sp = __shfl_up(s, 1, 64);
        uint32_t edge = s;                                          // lane 0: the row before the wave, lane 63: the row after
        if (lane == 0) edge = (i > 0 && valid) ? seg[i - 1] : ~s;
        if (lane == 63) edge = (i + 1 < n) ? seg[i + 1] : ~s;
        const bool starts_run = lane == 0 || sp != s;
        const uint64_t starts = __ballot(starts_run);
        const bool lane0_is_head = __shfl((int)(edge != s), 0, 64) != 0;
        if (valid && (lane == 63 || sn != s)) {
            const uint64_t upto = lane == 63 ? starts : (starts & ((2ull << lane) - 1ull));
            const int run_start = 63 - __clzll((long long)upto);                 // lane 0 always starts a run
            const bool head_here = run_start > 0 || lane0_is_head;
            const bool tail_here = lane < 63 || edge != s;
            if (head_here && tail_here) acc[s] = x;
            else atomic_combine(kind, op, &acc[s], x);
        }
    }
}

// ---- the reference's u32 tail in two passes -----------------------------------------------------------------
// mk_flags -> scan -> tail scatter -> one segmented reduction per aggregate (groupby.fut:26-58) read the sorted rows six
// times and keep 64-bit accumulators for 32-bit results.  Over sorted u32 keys with u32 operators (sum / prod mod 2^32, max,
// min: associative and commutative, so the fold order does not show) two passes do: (1) run heads per tile of kSegTile rows,
// scanned into tile offsets; (2) every tile recomputes its heads, numbers its runs from its offset, emits the keys and reduces
// EVERY aggregate in one sweep with a wave-level segmented scan -- a run that lies inside one wave is stored plainly, one
// that crosses a wave boundary goes through a 32-bit atomic on the result (pre-filled with the operator's identity).
constexpr int kSegTile = 2048, kSegMaxAggs = 8;
struct SegAggs {
    const uint32_t *col[kSegMaxAggs];           // the aggregate's column: in sorted order (carried) or in table order (gathered through perm)
    uint32_t *out[kSegMaxAggs];                 // [G], pre-filled with the identity
    int32_t op[kSegMaxAggs];
    int32_t gathered[kSegMaxAggs];
    int32_t n;
};

__device__ __forceinline__ bool seg_is_head(const uint32_t *__restrict__ keys, int64_t i) { return i == 0 || keys[i] != keys[i - 1]; }

// the tile's keys (row j * 256 + thread of the tile in key[j]; rows past the end repeat the last key and are not heads) and
// which of them start a run: the key before comes from the neighbouring lane, lane 0 reads it
constexpr int kSegSteps = kSegTile / 256;
__device__ __forceinline__ uint32_t seg_tile_heads(const uint32_t *__restrict__ keys, int64_t n, int64_t base, uint32_t (&key)[kSegSteps])
{
    const int lane = threadIdx.x & 63;
    uint32_t before[kSegSteps];
#pragma unroll
    for (int j = 0; j < kSegSteps; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        key[j] = keys[i < n ? i : n - 1];
        before[j] = keys[lane == 0 && i > 0 && i < n ? i - 1 : 0];      // (one line per wave; other lanes read keys[0] and drop it)
    }
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < kSegSteps; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const uint32_t up = __shfl_up(key[j], 1, 64);
        const uint32_t prev = lane == 0 ? before[j] : up;
        if (i < n && (i == 0 || prev != key[j])) heads |= 1u << j;
    }
    return heads;
}

__global__ __launch_bounds__(256) void seg_count_kernel(const uint32_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0u;
    __syncthreads();
    uint32_t key[kSegSteps];
    uint32_t c = (uint32_t)__popc(seg_tile_heads(keys, n, (int64_t)blockIdx.x * kSegTile, key));
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
}

__device__ __forceinline__ uint32_t u32_combine(int op, uint32_t a, uint32_t b)
{
    return op == OP_SUM ? a + b : op == OP_PROD ? a * b : op == OP_MAX ? (a > b ? a : b) : (a < b ? a : b);
}

__device__ __forceinline__ void u32_atomic_combine(int op, uint32_t *dst, uint32_t x)
{
    if (op == OP_SUM) atomicAdd(dst, x);
    else if (op == OP_MAX) atomicMax(dst, x);
    else if (op == OP_MIN) atomicMin(dst, x);
    else { uint32_t old = *dst, assumed; do { assumed = old; old = atomicCAS(dst, assumed, assumed * x); } while (old != assumed); }
}

__global__ __launch_bounds__(256) void seg_fused_u32_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ perm, int64_t n,
                                                            const int64_t *__restrict__ offsets, uint32_t *__restrict__ out_keys, SegAggs aggs)
{
    __shared__ uint32_t s_w[kSegSteps][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kSegTile;
    // every load of the tile goes out before anything waits: keys, then the rows' ids / the carried column
    uint32_t key[kSegSteps], rowv[kSegSteps];
    const uint32_t heads = seg_tile_heads(keys, n, base, key);
    const uint32_t *first = perm ? perm : aggs.col[0];                      // perm: row ids; else: the (one) carried column in sorted order
#pragma unroll
    for (int j = 0; j < kSegSteps; j++) { const int64_t i = base + j * 256 + threadIdx.x; rowv[j] = aggs.n || perm ? first[i < n ? i : n - 1] : 0u; }
    uint32_t after[kSegSteps];                                               // lane 63: the key of the next wave's first row (does my run go on there?)
#pragma unroll
    for (int j = 0; j < kSegSteps; j++) { const int64_t i = base + j * 256 + threadIdx.x; after[j] = keys[lane == 63 && i + 1 < n ? i + 1 : 0]; }
    uint32_t incl[kSegSteps];
#pragma unroll
    for (int j = 0; j < kSegSteps; j++) {
        const uint64_t hb = __ballot((heads >> j) & 1u);
        incl[j] = (uint32_t)__popcll(hb & ((2ull << lane) - 1ull));
        if (lane == 0) s_w[j][wave] = (uint32_t)__popcll(hb);
    }
    __syncthreads();
    uint32_t run_base = (uint32_t)offsets[blockIdx.x];                      // run heads before this tile (= the number of the tile's first head)
#pragma unroll
    for (int j = 0; j < kSegSteps; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const bool valid = i < n, head = (heads >> j) & 1u;
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t x = s_w[j][w]; if (w < wave) before += x; total += x; }
        const uint32_t s = valid ? run_base + before + incl[j] - 1u : 0xFFFFFFFFu;     // the row's run: heads up to and including it, minus one
        run_base += total;
        if (head) out_keys[s] = key[j];
        if (__ballot(valid) == 0ull) continue;
        // the run's extent inside the wave: a run that starts and ends here is complete (the run numbers ascend)
        const uint32_t sn = __shfl_down(s, 1, 64), sp = __shfl_up(s, 1, 64);
        const bool last_of_run_here = valid && (lane == 63 || sn != s);
        // lane 63: does the run go on in the next wave?  The next wave's first row is row i + 1: not a head <=> same key
        const bool next_row_same = lane == 63 && i + 1 < n && after[j] == key[j];
        const uint64_t starts = __ballot(lane == 0 || sp != s);
        const uint64_t upto = lane == 63 ? starts : (starts & ((2ull << lane) - 1ull));
        const int run_start = 63 - __clzll((long long)upto);
        const bool lane0_is_head = __shfl((int)head, 0, 64) != 0;
        const bool complete = last_of_run_here && (run_start > 0 || lane0_is_head) && !next_row_same;
        // which scan steps join anything at all (bit d of `joins`: this lane takes the value 2^d lanes below; a step no lane
        // joins in is skipped by the whole wave -- with mostly distinct keys that is every step)
        uint32_t joins = 0, any = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t below = __shfl_up(s, 1 << q, 64);           // (every lane takes part: a lane that sat out would hand on nothing)
            const bool jn = lane >= (1 << q) && below == s;
            joins |= jn ? 1u << q : 0u;
            any |= __ballot(jn) != 0ull ? 1u << q : 0u;
        }
        for (int a = 0; a < aggs.n; a++) {
            const int op = aggs.op[a];
            uint32_t x = !valid ? 0u : aggs.gathered[a] ? aggs.col[a][rowv[j]] : rowv[j];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                if (!((any >> q) & 1u)) continue;
                const uint32_t y = __shfl_up(x, 1 << q, 64);
                if ((joins >> q) & 1u) x = u32_combine(op, y, x);
            }
            if (last_of_run_here) { if (complete) aggs.out[a][s] = x; else u32_atomic_combine(op, &aggs.out[a][s], x); }
        }
    }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *__restrict__ dst, int64_t n, uint32_t v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// out dtype conversions of the 64-bit accumulators
__global__ __launch_bounds__(256) void finalize_kernel(const u64 *__restrict__ acc, const u64 *__restrict__ cnt, int64_t G, int kind,
                                                       int out_dtype, int avg, void *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
        const u64 a = acc[g];
        if (avg) { static_cast<float *>(out)[g] = (float)(u2d(a) / (double)cnt[g]); continue; }
        switch (out_dtype) {
        case HARK_F32: static_cast<float *>(out)[g] = kind == ACC_F64 ? (float)u2d(a) : (float)(long long)a; break;
        case HARK_I64: static_cast<u64 *>(out)[g] = kind == ACC_F64 ? (u64)(long long)u2d(a) : a; break;
        default: static_cast<uint32_t *>(out)[g] = (uint32_t)a; break;     // I32 / U32: wrap to 32 bits
        }
    }
}

int grid_for(hark_context *ctx, int64_t n)
{
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

void result_release(hark_context *ctx, hark_result *r)
{
    for (auto &c : r->cols) if (c.owned && c.data) hark_free(ctx, c.data);
    hark_result_host_release(ctx, r);
    delete r;
}

struct AggSpec { int col; int op; int kind; int out_dtype; int count_mode; int avg; };

// Everything after argument checking, shared by both entries.  Keys of
// `key_dtype` are ordered by k_argsort_column's total order for that dtype.
int grouped_aggregate(hark_context *ctx, const hark_table *db, int key_col, int key_dtype,
                      const std::vector<AggSpec> &aggs, hark_result *res, int64_t *G_out)
{
    const int64_t n = db->n;
    const int kesz = (int)hark_dtype_size(key_dtype);
    uint32_t *perm = nullptr, *flags = nullptr, *seg = nullptr;
    void *sorted_keys = nullptr;
    u64 *acc = nullptr, *cnt = nullptr;
    int64_t G = 0;
    hipStream_t st = ctx->stream;
    // integer keys come back from the sorted sort words (no random gather of the key column)
    const bool from_words = key_dtype == HARK_U32 || key_dtype == HARK_I32;
    uint32_t *words = nullptr;
    // when every aggregate reads the same 4-byte column, that column travels with the keys through the sort and
    // the segmented reductions read it in order (no row-id permutation, no random gather per aggregate)
    int carry = -1;
    bool one_col = from_words;                              // the keys then come back from the sort words, not through row ids
    for (const AggSpec &a : aggs) {
        if (a.count_mode) continue;
        if (carry < 0) carry = a.col;
        one_col = one_col && a.col == carry && hark_dtype_size(db->cols[a.col].dtype) == 4;
    }
    bool carried = one_col && carry >= 0;
    int rc = HARK_OK;
    bool keys64 = false;
    if (key_dtype == HARK_I64 && n >= 4096) {
        // i64 keys: the sort hands back the sorted keys (biased by 2^63: equality tests and the emitted keys below undo it)
        // and, when every aggregate reads one 4-byte column, that column in sorted order -- no gather of either
        bool one64 = carry >= 0;
        for (const AggSpec &a : aggs) one64 = one64 && (a.count_mode || (a.col == carry && hark_dtype_size(db->cols[a.col].dtype) == 4));
        uint64_t *k64 = nullptr;
        uint32_t *val = nullptr;
        rc = k_argsort_i64_keys(ctx, db->cols[key_col].data, n, &perm, &k64, one64 ? static_cast<const uint32_t *>(db->cols[carry].data) : nullptr,
                                one64 ? &val : nullptr, nullptr, nullptr, &db->cols[key_col].msd_unfit);
        if (!rc) {
            HARK_LAUNCH_RC(ctx, rc, unbias_u64_kernel<<<grid_for(ctx, n), 256, 0, st>>>(k64, n));
            sorted_keys = k64; keys64 = true;
            if (one64 && val) { hark_free(ctx, perm); perm = val; carried = true; }      // `perm` now holds the column in sorted order
        } else hark_free(ctx, k64);
    } else
    rc = k_sort_column(ctx, db->cols[key_col].data, key_dtype, n, false,
                           carried ? static_cast<const uint32_t *>(db->cols[carry].data) : nullptr, &perm, from_words ? &words : nullptr);
    if (keys64) { }
    else if (!rc && from_words) { sorted_keys = words; words = nullptr; }       // integer keys: the sorted words ARE the sorted keys
    else if (!rc) {
        rc = hark_alloc(ctx, &sorted_keys, (size_t)n * kesz);
        if (!rc) rc = k_gather(ctx, db->cols[key_col].data, kesz, perm, sorted_keys, n);
    }
    hark_free(ctx, words);
    // u32 operators over 4-byte integer keys (the reference entry): heads, run numbers, keys and every aggregate in two passes
    bool u32_tail = from_words && aggs.size() <= (size_t)kSegMaxAggs && !getenv("HARK_GROUPBY_NO_FUSED_TAIL");
    for (const AggSpec &a : aggs)
        u32_tail = u32_tail && a.kind == ACC_U64 && a.out_dtype == HARK_U32 && !a.count_mode && !a.avg && hark_dtype_size(db->cols[a.col].dtype) == 4;
    if (!rc && u32_tail) {
        const int64_t ntiles = (n + kSegTile - 1) / kSegTile;
        uint32_t *counts = nullptr; int64_t *offsets = nullptr;
        rc = hark_alloc(ctx, (void **)&counts, (size_t)ntiles * 4);
        if (!rc) rc = hark_alloc(ctx, (void **)&offsets, (size_t)(ntiles + 1) * 8);
        if (!rc) {
            HARK_LAUNCH_RC(ctx, rc, seg_count_kernel<<<dim3((unsigned)ntiles), 256, 0, st>>>(static_cast<const uint32_t *>(sorted_keys), n, counts));
            if (!rc) rc = k_exclusive_scan_u32(ctx, counts, ntiles, nullptr, offsets, &G);
        }
        if (!rc) {
            res->n = G;
            res->cols.resize(1 + aggs.size());
            res->cols[0].dtype = key_dtype;
            rc = hark_alloc(ctx, &res->cols[0].data, (size_t)G * 4);
            SegAggs sa{};
            sa.n = (int32_t)aggs.size();
            for (size_t j = 0; j < aggs.size() && !rc; j++) {
                res->cols[1 + j].dtype = HARK_U32;
                rc = hark_alloc(ctx, &res->cols[1 + j].data, (size_t)G * 4);
                if (rc) break;
                sa.col[j] = carried ? perm : static_cast<const uint32_t *>(db->cols[aggs[j].col].data);
                sa.gathered[j] = carried ? 0 : 1;
                sa.out[j] = static_cast<uint32_t *>(res->cols[1 + j].data);
                sa.op[j] = aggs[j].op;
                HARK_LAUNCH_RC(ctx, rc, fill_u32_kernel<<<grid_for(ctx, G), 256, 0, st>>>(sa.out[j], G, (uint32_t)identity_of(ACC_U64, aggs[j].op)));
            }
            if (!rc) {
                HARK_LAUNCH_RC(ctx, rc, seg_fused_u32_kernel<<<dim3((unsigned)ntiles), 256, 0, st>>>(static_cast<const uint32_t *>(sorted_keys), carried ? nullptr : perm, n, offsets,
                                                                                                    static_cast<uint32_t *>(res->cols[0].data), sa));
            }
        }
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "groupby: kernels failed");
        hark_free(ctx, counts); hark_free(ctx, offsets); hark_free(ctx, perm); hark_free(ctx, sorted_keys);
        *G_out = G;
        return rc;
    }
    if (!rc) rc = hark_alloc(ctx, (void **)&flags, (size_t)n * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&seg, (size_t)n * 4);
    if (!rc) {
        HARK_LAUNCH_RC(ctx, rc, head_flags_kernel<<<grid_for(ctx, n), 256, 0, st>>>(sorted_keys, kesz, key_dtype == HARK_F32 ? 1 : 0, n, flags));
        if (!rc) rc = k_exclusive_scan_u32(ctx, flags, n, seg, nullptr, &G);
    }
    if (!rc) {
        res->n = G;
        res->cols.resize(1 + aggs.size());
        res->cols[0].dtype = key_dtype;
        rc = hark_alloc(ctx, &res->cols[0].data, (size_t)G * kesz);
        for (size_t j = 0; j < aggs.size() && !rc; j++) {
            res->cols[1 + j].dtype = aggs[j].out_dtype;
            rc = hark_alloc(ctx, &res->cols[1 + j].data, (size_t)G * hark_dtype_size(aggs[j].out_dtype));
        }
    }
    if (!rc) {
        HARK_LAUNCH_RC(ctx, rc, seg_ids_kernel<<<grid_for(ctx, n), 256, 0, st>>>(seg, flags, n, sorted_keys, kesz, res->cols[0].data));
        if (!rc) rc = hark_alloc(ctx, (void **)&acc, (size_t)G * 8);
        if (!rc) rc = hark_alloc(ctx, (void **)&cnt, (size_t)G * 8);
    }
    for (size_t j = 0; j < aggs.size() && !rc; j++) {
        const AggSpec &a = aggs[j];
        const int col_dtype = a.count_mode ? HARK_I32 : db->cols[a.col].dtype;
        const void *col = a.count_mode ? nullptr : carried ? static_cast<const void *>(perm) : db->cols[a.col].data;
        HARK_LAUNCH_RC(ctx, rc, fill_u64_kernel<<<grid_for(ctx, G), 256, 0, st>>>(acc, G, identity_of(a.kind, a.op)));
        HARK_LAUNCH_RC(ctx, rc, seg_reduce_kernel<<<grid_for(ctx, n), 256, 0, st>>>(col, col_dtype, carried ? nullptr : perm, seg, n, a.kind, a.op, a.count_mode, acc));
        if (a.avg) {
            HARK_LAUNCH_RC(ctx, rc, fill_u64_kernel<<<grid_for(ctx, G), 256, 0, st>>>(cnt, G, 0ull));
            HARK_LAUNCH_RC(ctx, rc, seg_reduce_kernel<<<grid_for(ctx, n), 256, 0, st>>>(nullptr, HARK_I32, perm, seg, n, ACC_U64, OP_SUM, 1, cnt));
        }
        HARK_LAUNCH_RC(ctx, rc, finalize_kernel<<<grid_for(ctx, G), 256, 0, st>>>(acc, cnt, G, a.kind, a.out_dtype, a.avg, res->cols[1 + j].data));
    }
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "groupby: kernels failed");
    hark_free(ctx, perm); hark_free(ctx, flags); hark_free(ctx, seg); hark_free(ctx, sorted_keys); hark_free(ctx, acc); hark_free(ctx, cnt);
    *G_out = G;
    return rc;
}

int kind_of(int dtype) { return dtype == HARK_F32 ? ACC_F64 : dtype == HARK_U32 ? ACC_U64 : ACC_I64; }

// defined at the end of this file: the fused dense-key kernels of k_fgb.hip behind the reference entry
int ref_groupby_dense(hark_context *ctx, const hark_table *view, const hark_table *stats_owner, int g_col,
                      const std::vector<AggSpec> &aggs, hark_result *res, int64_t *G_out, bool *used);

int ref_groupby_hash(hark_context *ctx, const hark_table *view, const hark_table *stats_owner, int g_col, const std::vector<AggSpec> &aggs,
                     hark_result *res, int64_t *G_out, bool *used);

} // namespace

extern "C" {

int hark_entry_query_groupby(hark_context *ctx, hark_result **out, const hark_table *db, int32_t g_col,
                             const int32_t *s_cols, int64_t ns, const int32_t *t_cols, int64_t nt)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (ns < 0 || nt < 0 || (ns && !s_cols) || (nt && !t_cols)) return hark_fail(ctx, HARK_EARG, "query_groupby: bad column lists");
    hark_result *res = new hark_result();
    if (db->n == 0) {                                   // segmented.fut:29: empty in, empty out
        res->n = 0; res->cols.resize((size_t)ns + 1);
        for (auto &c : res->cols) { c.dtype = HARK_U32; c.data = nullptr; c.owned = false; }
        *out = res; return HARK_OK;
    }
    // `row[i]` in keep_fun (groupby.fut:52) is bounds-checked for every row
    for (int64_t j = -1; j < ns; j++) {
        const int c = j < 0 ? g_col : s_cols[j];
        if (c < 0 || c >= db->m) { delete res; return hark_fail(ctx, HARK_EBOUNDS, "query_groupby: index %d out of bounds for a table with %lld columns", c, (long long)db->m); }
        if (db->cols[c].dtype != HARK_I32 && db->cols[c].dtype != HARK_U32) { delete res; return hark_fail(ctx, HARK_EUNSUPPORTED, "query_groupby: column %d is not a 32-bit integer column (groupby.fut:51 is u32)", c); }
    }
    std::vector<AggSpec> aggs;
    for (int64_t j = 0; j < ns; j++) {
        // opcode table of type_func (groupby.fut:35-41); anything else -> u32.min.
        // A missing opcode (nt < ns) is only an error if `merge` ever runs (checked below).
        const int t = j < nt ? t_cols[j] : 4;
        const int op = t == 1 ? OP_PROD : t == 2 ? OP_SUM : t == 3 ? OP_MAX : OP_MIN;
        aggs.push_back(AggSpec{s_cols[j], op, ACC_U64, HARK_U32, 0, 0});
    }
    if (k_small_fits(db, ns + 1)) {                       // a few rows: one launch, one synchronisation (k_small.hip)
        int32_t cols[33], ops[33];
        cols[0] = g_col; ops[0] = 0;
        for (int64_t j = 0; j < ns; j++) { cols[j + 1] = s_cols[j]; ops[j + 1] = j < nt ? t_cols[j] : 4; }
        int64_t Gs = 0;
        int rcs = k_small_query_groupby(ctx, db, cols, ops, ns + 1, res, &Gs);
        ctx->last_groupby_path = HARK_PATH_SMALL;
        if (!rcs && nt < ns && Gs < db->n)                // (as below: merge would index t_cols out of bounds)
            rcs = hark_fail(ctx, HARK_EBOUNDS, "query_groupby: %lld aggregate opcodes for %lld select columns", (long long)nt, (long long)ns);
        if (rcs) { result_release(ctx, res); return rcs; }
        *out = res;
        return HARK_OK;
    }
    // u32 view of every column: groupby.fut:51 types the whole table as u32
    hark_table view = *db;
    for (auto &c : view.cols) { c.owned = false; if (c.dtype == HARK_I32) c.dtype = HARK_U32; }
    int64_t G = 0;
    bool dense = false;
    ctx->last_groupby_path = HARK_PATH_NONE;
    ctx->last_groupby_window = 0;
    int rc = ref_groupby_dense(ctx, &view, db, g_col, aggs, res, &G, &dense);   // keys < 2^21: fused kernels, no sort
    if (!rc && dense) ctx->last_groupby_path = HARK_PATH_DENSE;
    if (!rc && !dense) { rc = ref_groupby_hash(ctx, &view, db, g_col, aggs, res, &G, &dense); if (!rc && dense) ctx->last_groupby_path = HARK_PATH_HASH; }   // sparse keys: LDS hash buckets
    if (!rc && !dense) { rc = grouped_aggregate(ctx, &view, g_col, HARK_U32, aggs, res, &G); if (!rc) ctx->last_groupby_path = HARK_PATH_SORT; }  // last resort: sort-based
    if (!rc && nt < ns && G < db->n)                    // some group has two rows: merge indexes t_cols[i-1] out of bounds
        rc = hark_fail(ctx, HARK_EBOUNDS, "query_groupby: %lld aggregate opcodes for %lld select columns", (long long)nt, (long long)ns);
    if (rc) { result_release(ctx, res); return rc; }
    *out = res;
    return HARK_OK;
}

} // extern "C"

// ---- SQL-typed GROUP BY on an already filtered table (used by hark_entry_filter_groupby)
int k_groupby_typed(hark_context *ctx, const hark_table *db, int32_t g_col, const int32_t *agg_cols,
                    const int32_t *agg_ops, int64_t n_aggs, hark_result *res)
{
    std::vector<AggSpec> aggs;
    for (int64_t j = 0; j < n_aggs; j++) {
        const int op = agg_ops[j];
        if (op == HARK_AGG_COUNT) { aggs.push_back(AggSpec{-1, OP_SUM, ACC_U64, HARK_I64, 1, 0}); continue; }
        const int c = agg_cols[j];
        if (c < 0 || c >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby: aggregate column %d out of bounds", c);
        const int dt = db->cols[c].dtype, kind = kind_of(dt);
        switch (op) {
        case HARK_AGG_SUM: aggs.push_back(AggSpec{c, OP_SUM, kind, dt == HARK_F32 ? HARK_F32 : HARK_I64, 0, 0}); break;
        case HARK_AGG_PROD: aggs.push_back(AggSpec{c, OP_PROD, kind, dt, 0, 0}); break;
        case HARK_AGG_MAX: aggs.push_back(AggSpec{c, OP_MAX, kind, dt, 0, 0}); break;
        case HARK_AGG_MIN: case HARK_AGG_KEY: aggs.push_back(AggSpec{c, OP_MIN, kind, dt, 0, 0}); break;
        case HARK_AGG_AVG: {
            // AVG = f64 sum / count, returned as f32
            AggSpec a{c, OP_SUM, ACC_F64, HARK_F32, 0, 1};
            aggs.push_back(a);
            break;
        }
        default: return hark_fail(ctx, HARK_EARG, "filter_groupby: unknown aggregate opcode %d", op);
        }
    }
    int64_t G = 0;
    return grouped_aggregate(ctx, db, g_col, db->cols[g_col].dtype, aggs, res, &G);
}

// ---------------------------------------------------------------------------
// hark_entry_filter_groupby: WHERE -> GROUP BY with SQL-typed outputs
// ---------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void minmax_u32_kernel(const uint32_t *__restrict__ col, int64_t n, int is_signed,
                                                         unsigned long long *__restrict__ out /* [0]=min, [1]=max (biased) */)
{
    // values are biased by 2^31 when signed so that one unsigned min/max serves both
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    const uint32_t bias = is_signed ? 0x80000000u : 0u;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t x = col[i] ^ bias;
        lo = x < lo ? x : lo; hi = x > hi ? x : hi;
    }
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t a = __shfl_down(lo, d, 64), b = __shfl_down(hi, d, 64);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], (unsigned long long)lo); atomicMax(&out[1], (unsigned long long)hi); }
}

int grid_for(hark_context *ctx, int64_t n);

// [min, max] of a 32-bit integer column (as signed for I32, unsigned otherwise); cached in the column.
int column_range(hark_context *ctx, const hark_table *t, int col, bool as_signed, int64_t *lo, int64_t *hi)
{
    const hark_column &c = t->cols[col];
    const int w = as_signed ? 1 : 0;
    if (!c.has_range[w] || t->n == 0) {
        unsigned long long *mm = nullptr;
        HARK_TRY(hark_alloc(ctx, (void **)&mm, 16));
        const unsigned long long init[2] = {~0ull, 0ull};
        int rc = hipMemcpyAsync(mm, init, 16, hipMemcpyHostToDevice, ctx->stream) == hipSuccess ? HARK_OK : hark_fail(ctx, HARK_EHIP, "stats upload failed");
        int64_t lohi[2] = {0, 0};
        if (!rc) {
            if (t->n > 0) HARK_LAUNCH_RC(ctx, rc, minmax_u32_kernel<<<grid_for(ctx, t->n), 256, 0, ctx->stream>>>(static_cast<const uint32_t *>(c.data), t->n, as_signed ? 1 : 0, mm));
            if (!rc) rc = hark_read_words(ctx, mm, lohi, 2);
        }
        hark_free(ctx, mm);
        if (rc) return rc;
        const int64_t bias = as_signed ? ((int64_t)1 << 31) : 0;
        c.range_min[w] = lohi[0] - bias; c.range_max[w] = lohi[1] - bias;
        c.has_range[w] = t->n > 0;
    }
    *lo = c.range_min[w]; *hi = c.range_max[w];
    return HARK_OK;
}

__global__ __launch_bounds__(256) void nonzero_flags_kernel(const unsigned long long *__restrict__ cnt, int64_t G, uint32_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) flags[g] = cnt[g] ? 1u : 0u;
}

constexpr int64_t kDenseMaxGroups = (int64_t)1 << 21;      // limit of the partition path of k_fgb.hip

// How one SQL aggregate maps onto a fused dense pass: accumulate operator, value transform,
// read-out kind (fgb_decode_kernel) and result dtype.
struct DensePass { int vop, xf, col, kind, out_dtype; bool count_only; };

// WHERE as an AND-list (n == 0: no filter); constants[j] is read as the dtype of column cols[j]
struct PredList { int64_t n; const int32_t *cols; const int32_t *cmps; const void *const *consts; };

bool dense_plan_for(int op, int dt, int col, DensePass *out)
{
    const bool f = dt == HARK_F32, i = dt == HARK_I32, u = dt == HARK_U32;
    if (op == HARK_AGG_COUNT) { *out = DensePass{-1, 0, -1, 11, HARK_I64, true}; return true; }
    if (!f && !i && !u) return false;                        // 8-byte columns: sort-based path
    switch (op) {
    case HARK_AGG_SUM: *out = f ? DensePass{0, 0, col, 0, HARK_F32, false} : i ? DensePass{5, 1, col, 5, HARK_I64, false} : DensePass{5, 0, col, 4, HARK_I64, false}; return true;
    case HARK_AGG_AVG: *out = f ? DensePass{0, 0, col, 7, HARK_F32, false} : i ? DensePass{5, 1, col, 9, HARK_F32, false} : DensePass{5, 0, col, 8, HARK_F32, false}; return true;
    case HARK_AGG_MAX: *out = f ? DensePass{2, 2, col, 3, HARK_F32, false} : i ? DensePass{2, 1, col, 2, HARK_I32, false} : DensePass{2, 0, col, 1, HARK_U32, false}; return true;
    case HARK_AGG_MIN: case HARK_AGG_KEY:
        *out = f ? DensePass{3, 2, col, 3, HARK_F32, false} : i ? DensePass{3, 1, col, 2, HARK_I32, false} : DensePass{3, 0, col, 1, HARK_U32, false}; return true;
    case HARK_AGG_PROD: if (f) return false; *out = DensePass{4, 0, col, 1, dt, false}; return true;      // wrapping 32-bit product
    default: return false;
    }
}

// Dense shape: 32-bit integer key with 0 <= key < 2^21 and aggregates over 4-byte columns.
// One fused pass (k_fgb.hip) per distinct (operator, column); an f32 predicate is fused into every
// pass, a predicate on another dtype compacts the referenced columns first.
// all_slots: the result has one row per key SLOT 0..G-1 (empty groups included: their COUNT is 0 and their other
// aggregates are unspecified) -- no group set, no scan, no host read; hark_entry_filter_groupby_topk selects from that.
// G_force > 0 (with all_slots): the key domain is [0, G_force) whatever THIS table's keys span -- a shard of a sharded table
// must lay its partial aggregates out over the domain of all shards (the slots are merged elementwise across GPUs); the
// caller has decided that the domain is dense enough.
int try_dense(hark_context *ctx, const hark_table *db, const PredList &preds,
              int32_t g_col, const int32_t *agg_cols, const int32_t *agg_ops, int64_t n_aggs, hark_result *res, bool *used, bool all_slots = false,
              int64_t G_force = 0)
{
    *used = false;
    const int kdt = db->cols[g_col].dtype;
    if (kdt != HARK_I32 && kdt != HARK_U32) return HARK_OK;
    std::vector<DensePass> plan_of((size_t)n_aggs);
    for (int64_t j = 0; j < n_aggs; j++) {
        const int c = agg_ops[j] == HARK_AGG_COUNT ? 0 : agg_cols[j];
        if (!dense_plan_for(agg_ops[j], db->cols[c].dtype, c, &plan_of[j])) return HARK_OK;
    }
    // column statistic: key range (computed once per column, then cached)
    int64_t kmin = 0, kmax = 0;
    if (db->n > 0) HARK_TRY(column_range(ctx, db, g_col, kdt == HARK_I32, &kmin, &kmax));
    int rc = HARK_OK;
    if (G_force > 0) {
        if (!all_slots || G_force > kDenseMaxGroups) return HARK_OK;
        if (kmin < 0 || kmax >= G_force) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_slots: keys span [%lld, %lld], outside [0, %lld)", (long long)kmin, (long long)kmax, (long long)G_force);
    } else if (kmin < 0 || kmax >= kDenseMaxGroups || kmax + 1 > 8 * db->n + 4096) return HARK_OK;
    const int64_t G = G_force > 0 ? G_force : kmax + 1;

    // WHERE: ONE f32 predicate rides in the kernels as it is (no extra pass over anything).  Several conjuncts, or a
    // predicate on another dtype, are evaluated once into a survivor bitmask (0.125 B/row) that every pass reads
    // instead of a predicate column -- no intermediate table is materialised.  (Measured, C5 with three aggregates of
    // three columns, 5e8 rows: the mask does NOT pay for a single f32 predicate even with three passes, 6.4 ms against
    // 5.6 ms -- the producer is bound by instruction issue, not by the 4 B/row of the predicate column.)
    const hark_table *src = db;
    const bool direct = preds.n == 1 && preds.cmps[0] != HARK_CMP_MASK && db->cols[preds.cols[0]].dtype == HARK_F32;
    uint8_t *mask = nullptr;
    if (preds.n >= 1 && !direct) HARK_TRY(k_predicate_bitmask(ctx, db, preds.n, preds.cols, preds.cmps, preds.consts, &mask));
    const float *p = direct ? static_cast<const float *>(db->cols[preds.cols[0]].data) : reinterpret_cast<const float *>(mask);
    const int cmp = direct ? preds.cmps[0] : HARK_CMP_MASK;
    const float thr = direct ? *static_cast<const float *>(preds.consts[0]) : 0.0f;
    const int32_t g2 = g_col;
    const int32_t *keys = static_cast<const int32_t *>(src->cols[g2].data);

    hark_fgb_plan *plan = nullptr;
    rc = k_fgb_plan_new_uncleared(ctx, &plan, src->n > 0 ? src->n : 1, G);          // (every pass below resets or initialises the accumulators itself)
    if (!rc) { plan->win_k = keys; plan->win_n = src->n; plan->win_verdict = src->cols[g2].key_clustered; }   // what an earlier statement found out about this key column
    uint32_t *flags = nullptr, *pos = nullptr;
    int64_t ngroups = -1;
    std::vector<char> done((size_t)n_aggs, 0);
    auto group_set = [&]() -> int {                            // the first pass also yields the group set
        if (ngroups >= 0) return HARK_OK;
        int r = HARK_OK;
        if (all_slots) ngroups = G;                            // pos stays NULL: the read-outs write slot g at row g
        else {
            r = hark_alloc(ctx, (void **)&flags, (size_t)G * 4);
            if (!r) r = hark_alloc(ctx, (void **)&pos, (size_t)G * 4);
            if (!r) {
                HARK_LAUNCH_RC(ctx, r, nonzero_flags_kernel<<<grid_for(ctx, G), 256, 0, ctx->stream>>>(plan->acc_cnt, G, flags));
                if (!r) r = k_exclusive_scan_u32(ctx, flags, G, pos, nullptr, &ngroups);
            }
        }
        if (!r) {
            res->n = ngroups;
            res->cols.resize((size_t)n_aggs + 1);
            res->cols[0].dtype = kdt;
            r = hark_alloc(ctx, &res->cols[0].data, (size_t)ngroups * 4);
            if (!r) r = hark_fgb_finish_typed(ctx, plan, 10, pos, res->cols[0].data);
            for (int64_t j = 0; j < n_aggs && !r; j++) {
                res->cols[(size_t)j + 1].dtype = plan_of[j].out_dtype;
                r = hark_alloc(ctx, &res->cols[(size_t)j + 1].data, (size_t)ngroups * hark_dtype_size(plan_of[j].out_dtype));
            }
        }
        return r;
    };
    auto run_pass = [&](int vop, int xf, const void *col) -> int {
        int r = hark_fgb_plan_set(plan, "vop", vop);
        if (!r) r = hark_fgb_plan_set(plan, "xform", xf);
        if (!r) r = hark_fgb_reset(ctx, plan);
        if (!r && src->n > 0) { r = k_fgb_dense_f32(ctx, plan, p, cmp, thr, keys, static_cast<const float *>(col), src->n); ctx->last_groupby_passes++; }
        if (!r) r = group_set();
        return r;
    };
    // ---- statistics pass: a column that needs two or more of {SUM/AVG, MIN, MAX} gets them (and COUNT) from ONE
    // producer + consumer pass (k_fgb_dense_stats); it declines (ran == false) for small G, > 4096 keys per bucket or
    // skewed data, and the separate passes below take over
    auto cls = [](const DensePass &d) { return d.count_only ? 0 : (d.vop == 0 || d.vop == 5) ? 1 : d.vop == 2 ? 2 : d.vop == 3 ? 4 : 0; };
    for (int64_t j = 0; j < n_aggs && !rc && src->n > 0; j++) {
        if (done[j] || !cls(plan_of[j])) continue;
        const int c = plan_of[j].col;
        int classes = 0;
        for (int64_t q = 0; q < n_aggs; q++) if (!done[q] && !plan_of[q].count_only && plan_of[q].col == c) classes |= cls(plan_of[q]);
        if (__builtin_popcount(classes) < 2) continue;
        const int dt = src->cols[c].dtype, vk = dt == HARK_F32 ? 0 : dt == HARK_I32 ? 1 : 2;
        bool ran = false;
        rc = hark_fgb_plan_set(plan, "vop", 0);
        if (!rc) rc = hark_fgb_plan_set(plan, "xform", 0);
        if (!rc) rc = k_fgb_dense_stats(ctx, plan, p, cmp, thr, keys, src->cols[c].data, src->n, vk, &ran);
        if (rc || !ran) continue;
        ctx->last_groupby_passes++;
        rc = group_set();
        for (int64_t q = 0; q < n_aggs && !rc; q++) {
            if (done[q]) continue;
            const int k = cls(plan_of[q]);
            if (plan_of[q].count_only) rc = hark_fgb_finish_typed_from(ctx, plan, 0, plan_of[q].kind, pos, res->cols[(size_t)q + 1].data);
            else if (k && plan_of[q].col == c) rc = hark_fgb_finish_typed_from(ctx, plan, k == 1 ? 0 : k == 2 ? 2 : 1, plan_of[q].kind, pos, res->cols[(size_t)q + 1].data);
            else continue;
            done[q] = 1;
        }
    }
    // ---- pair passes: two aggregates of two different columns from ONE producer + consumer pass (k_fgb_dense_pair:
    // 10-byte pairs).  Value 1 may be a SUM / AVG or a MAX / MIN, value 2 must be a MAX / MIN (a 32-bit LDS slot); a SUM is
    // paired first so that the MAX / MIN passes it would otherwise leave behind shrink to one.  Declines like the
    // statistics pass (small G, skew), and then the single passes below take over.
    auto same_pass = [&](int64_t a, int64_t b) { return !plan_of[b].count_only && plan_of[b].vop == plan_of[a].vop && plan_of[b].xf == plan_of[a].xf && plan_of[b].col == plan_of[a].col; };
    auto is_sum = [&](int64_t q) { return !plan_of[q].count_only && (plan_of[q].vop == 0 || plan_of[q].vop == 5); };
    auto is_ext = [&](int64_t q) { return !plan_of[q].count_only && (plan_of[q].vop == 2 || plan_of[q].vop == 3); };
    // ---- triple passes first (14-byte entries: value 1 a SUM / AVG when there is one, values 2 and 3 a MAX / MIN each, of
    // three different (operator, column) passes): BASELINE configs[4]'s SUM(a), MAX(b), MIN(c), COUNT(*) is ONE pass
    bool triples = true;
    while (triples && !rc && src->n > 0) {
        triples = false;
        int64_t j = -1, q1 = -1, q2 = -1;
        for (int64_t t = 0; t < n_aggs && j < 0; t++) if (!done[t] && is_sum(t)) j = t;
        for (int64_t t = 0; t < n_aggs && j < 0; t++) if (!done[t] && is_ext(t)) j = t;
        if (j < 0) break;
        for (int64_t t = 0; t < n_aggs && q1 < 0; t++) if (t != j && !done[t] && is_ext(t) && !same_pass(j, t)) q1 = t;
        for (int64_t t = 0; q1 >= 0 && t < n_aggs && q2 < 0; t++) if (t != j && t != q1 && !done[t] && is_ext(t) && !same_pass(j, t) && !same_pass(q1, t)) q2 = t;
        if (q2 < 0) break;
        // 14-byte entries in slabs sized for 8-byte ones (1.3 x 8 = 10.4 B per row) would overflow from ~70 % selectivity on:
        // a statement with a triple gets slabs for 14 B per row (9 GB per 5e8 rows instead of 5.2)
        if (plan->slack_pct < 230) rc = hark_fgb_plan_set(plan, "slack_pct", 230);
        if (rc) break;
        bool ran = false;
        rc = k_fgb_dense_multi(ctx, plan, p, cmp, thr, keys, src->cols[plan_of[j].col].data, plan_of[j].vop, plan_of[j].xf,
                               src->cols[plan_of[q1].col].data, plan_of[q1].vop, plan_of[q1].xf,
                               src->cols[plan_of[q2].col].data, plan_of[q2].vop, plan_of[q2].xf, src->n, &ran);
        if (rc || !ran) break;                                            // declined (geometry, skew, high selectivity): pairs and singles below
        ctx->last_groupby_passes++;
        rc = group_set();
        for (int64_t t = 0; t < n_aggs && !rc; t++) {
            if (done[t]) continue;
            if (plan_of[t].count_only || same_pass(j, t)) rc = hark_fgb_finish_typed(ctx, plan, plan_of[t].kind, pos, res->cols[(size_t)t + 1].data);
            else if (same_pass(q1, t)) rc = hark_fgb_finish_typed_from(ctx, plan, 1, plan_of[t].kind, pos, res->cols[(size_t)t + 1].data);
            else if (same_pass(q2, t)) rc = hark_fgb_finish_typed_from(ctx, plan, 2, plan_of[t].kind, pos, res->cols[(size_t)t + 1].data);
            else continue;
            done[t] = 1;
        }
        triples = true;
    }
    for (int turn = 0; turn < 2 && !rc && src->n > 0; turn++) {          // turn 0: (sum, max/min) pairs, turn 1: (max/min, max/min) pairs
        for (int64_t j = 0; j < n_aggs && !rc; j++) {
            if (done[j] || !(turn == 0 ? is_sum(j) : is_ext(j))) continue;
            int64_t q = -1;
            for (int64_t t = 0; t < n_aggs; t++) if (t != j && !done[t] && is_ext(t) && !same_pass(j, t)) { q = t; break; }
            if (q < 0) continue;
            bool ran = false;
            rc = k_fgb_dense_pair(ctx, plan, p, cmp, thr, keys, src->cols[plan_of[j].col].data, plan_of[j].vop, plan_of[j].xf,
                                  src->cols[plan_of[q].col].data, plan_of[q].vop, plan_of[q].xf, src->n, &ran);
            if (rc || !ran) { turn = 2; break; }                          // it declines for the whole statement alike
            ctx->last_groupby_passes++;
            rc = group_set();
            for (int64_t t = 0; t < n_aggs && !rc; t++) {
                if (done[t]) continue;
                if (plan_of[t].count_only || same_pass(j, t)) rc = hark_fgb_finish_typed(ctx, plan, plan_of[t].kind, pos, res->cols[(size_t)t + 1].data);
                else if (same_pass(q, t)) rc = hark_fgb_finish_typed_from(ctx, plan, 1, plan_of[t].kind, pos, res->cols[(size_t)t + 1].data);
                else continue;
                done[t] = 1;
            }
        }
    }
    for (int64_t j = 0; j < n_aggs && !rc; j++) {
        if (done[j] || plan_of[j].count_only) continue;
        rc = run_pass(plan_of[j].vop, plan_of[j].xf, src->cols[plan_of[j].col].data);
        for (int64_t q = j; q < n_aggs && !rc; q++)           // every aggregate this pass serves
            if (!done[q] && (plan_of[q].count_only || (plan_of[q].vop == plan_of[j].vop && plan_of[q].xf == plan_of[j].xf && plan_of[q].col == plan_of[j].col))) {
                rc = hark_fgb_finish_typed(ctx, plan, plan_of[q].kind, pos, res->cols[(size_t)q + 1].data);
                done[q] = 1;
            }
    }
    if (!rc && ngroups < 0) {                                 // COUNT only (or no aggregate, e.g. SELECT DISTINCT): no value column is
        rc = run_pass(0, 0, nullptr);                         // read at all; the partition carries bucket-local keys only
        for (int64_t q = 0; q < n_aggs && !rc; q++) rc = hark_fgb_finish_typed(ctx, plan, plan_of[q].kind, pos, res->cols[(size_t)q + 1].data);
    } else
        for (int64_t q = 0; q < n_aggs && !rc; q++)           // COUNTs that came before the first value pass
            if (!done[q]) rc = hark_fgb_finish_typed(ctx, plan, plan_of[q].kind, pos, res->cols[(size_t)q + 1].data);
    if (!rc && plan) rc = hark_fgb_check(ctx, plan);          // the sticky error word, once per statement (it also drains the stream)
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "filter_groupby: kernels failed");
    hark_free(ctx, flags); hark_free(ctx, pos);
    if (plan) {
        if (plan->win_k == keys && plan->win_verdict >= 0) src->cols[g2].key_clustered = (int8_t)plan->win_verdict;
        ctx->last_groupby_window = plan->win_rows > 0 ? 1 : plan->rot_rows > 0 ? 2 : 0;
        hark_fgb_plan_free(ctx, plan);
    }
    hark_free(ctx, mask);
    *used = rc == HARK_OK;
    return rc;
}

// Sparse 32-bit integer keys: the LDS hash-bucket pipeline of k_fgb.hip, one pass per distinct
// (operator, column), results brought into ascending key order (signed for I32).  A WHERE is
// fused into the hash producer (one f32 predicate as it is, anything else as a survivor bitmask).
int try_hash(hark_context *ctx, const hark_table *db, const PredList &preds,
             int32_t g_col, const int32_t *agg_cols, const int32_t *agg_ops, int64_t n_aggs, hark_result *res, bool *used)
{
    *used = false;
    const int kdt = db->cols[g_col].dtype;
    if ((kdt != HARK_I32 && kdt != HARK_U32) || db->n < ((int64_t)1 << 18)) return HARK_OK;
    std::vector<DensePass> plan_of((size_t)n_aggs);
    for (int64_t j = 0; j < n_aggs; j++) {
        const int c = agg_ops[j] == HARK_AGG_COUNT ? 0 : agg_cols[j];
        if (!dense_plan_for(agg_ops[j], db->cols[c].dtype, c, &plan_of[j])) return HARK_OK;
    }
    // WHERE rides in the hash producer (round 4; a compaction of the referenced columns before): ONE f32 predicate as it is,
    // anything else as a survivor bitmask evaluated once (0.125 B/row) -- only surviving rows are partitioned
    const hark_table *src = db;
    const int32_t g2 = g_col;
    const bool direct = preds.n == 1 && preds.cmps[0] != HARK_CMP_MASK && db->cols[preds.cols[0]].dtype == HARK_F32;
    uint8_t *mask = nullptr;
    if (preds.n >= 1 && !direct) HARK_TRY(k_predicate_bitmask(ctx, db, preds.n, preds.cols, preds.cmps, preds.consts, &mask));
    hark_row_pred pred{nullptr, 0, 0.0f};
    if (preds.n >= 1) pred = direct ? hark_row_pred{static_cast<const float *>(db->cols[preds.cols[0]].data), preds.cmps[0], *static_cast<const float *>(preds.consts[0])}
                                    : hark_row_pred{reinterpret_cast<const float *>(mask), HARK_CMP_MASK, 0.0f};
    const uint32_t *keys = static_cast<const uint32_t *>(src->cols[g2].data);
    int rc = HARK_OK;
    bool ok = src->n > 0;
    int64_t G = -1;
    // What an earlier statement WITHOUT a WHERE learnt about this key column stays with it, as in the reference entry (ADVICE r04:
    // a column the hash path does not fit -- too many distinct keys, skew -- paid producer + consumer + the sort path on every
    // statement): unfit -> no attempt; the table rounds its keys need -> no failed first round.  A WHERE changes which keys
    // occur, so filtered statements neither use nor leave a verdict (hark_table_invalidate_stats forgets it).
    const hark_column &kc = db->cols[g_col];
    const bool whole_column = preds.n == 0;
    if (whole_column && kc.hash_rounds < 0) { hark_free(ctx, mask); return HARK_OK; }
    uint32_t rounds = whole_column && kc.hash_rounds > 0 ? (uint32_t)kc.hash_rounds : 0u;
    int why = HARK_HASH_FITS;
    std::vector<char> done((size_t)n_aggs, 0);
    unsigned long long *accg = nullptr, *cntg = nullptr;
    hark_hash_part part;                                     // passes over one (column, value transform) share the hash partition
    auto run_pass = [&](int vop, int xf, const void *col) -> int {
        uint32_t *hk = nullptr, *perm = nullptr; unsigned long long *hv = nullptr, *hc = nullptr;
        int64_t Gj = 0;
        int r = k_fgb_hash_u32(ctx, keys, static_cast<const uint32_t *>(col), src->n, vop, xf, &hk, &hv, &hc, &Gj, &ok, &rounds, false, &part, &why,
                               preds.n >= 1 ? &pred : nullptr);
        if (!r && ok && Gj == 0) ok = false;                  // no row survives the WHERE: the generic path returns the typed empty result
        if (!r && ok) {
            if (G < 0) {
                G = Gj; res->n = G;
                res->cols.resize((size_t)n_aggs + 1);
                res->cols[0].dtype = kdt;
                r = hark_alloc(ctx, &res->cols[0].data, (size_t)G * 4);
                for (int64_t j = 0; j < n_aggs && !r; j++) {
                    res->cols[(size_t)j + 1].dtype = plan_of[j].out_dtype;
                    r = hark_alloc(ctx, &res->cols[(size_t)j + 1].data, (size_t)G * hark_dtype_size(plan_of[j].out_dtype));
                }
                if (!r) r = hark_alloc(ctx, (void **)&accg, (size_t)G * 8);
                if (!r) r = hark_alloc(ctx, (void **)&cntg, (size_t)G * 8);
            } else if (Gj != G) r = hark_fail(ctx, HARK_EHIP, "filter_groupby: inconsistent group counts between passes");
            if (!r) r = k_argsort_column(ctx, hk, kdt, G, false, &perm, nullptr);       // SQL order: signed for I32
            if (!r && G > 0) {
                r = k_gather(ctx, hk, 4, perm, res->cols[0].data, G);
                if (!r) r = k_gather(ctx, hv, 8, perm, accg, G);
                if (!r) r = k_gather(ctx, hc, 8, perm, cntg, G);
            }
        }
        hark_free(ctx, hk); hark_free(ctx, hv); hark_free(ctx, hc); hark_free(ctx, perm);
        return r;
    };
    // ---- statistics pass: a column that needs two or more of {SUM/AVG, MIN, MAX} gets them (and COUNT) from ONE consumer pass
    // over its pairs (fgb_agg_hash_stats_kernel) and ONE sort of the result keys, instead of a consumer pass and a sort each
    auto cls = [](const DensePass &d) { return d.count_only ? 0 : (d.vop == 0 || d.vop == 5) ? 1 : d.vop == 2 ? 2 : d.vop == 3 ? 4 : 0; };
    uint32_t rounds_stats = 0;
    unsigned long long *ming = nullptr, *maxg = nullptr;
    for (int64_t j = 0; j < n_aggs && !rc && ok; j++) {
        if (done[j] || !cls(plan_of[j])) continue;
        const int c = plan_of[j].col;
        int classes = 0;
        for (int64_t q = 0; q < n_aggs; q++) if (!done[q] && !plan_of[q].count_only && plan_of[q].col == c) classes |= cls(plan_of[q]);
        if (__builtin_popcount(classes) < 2) continue;
        const int dt = src->cols[c].dtype, vk = dt == HARK_F32 ? 0 : dt == HARK_I32 ? 1 : 2;
        uint32_t *hk = nullptr, *perm = nullptr; unsigned long long *hv = nullptr, *hc = nullptr, *hmin = nullptr, *hmax = nullptr;
        int64_t Gj = 0;
        rc = k_fgb_hash_u32(ctx, keys, static_cast<const uint32_t *>(src->cols[c].data), src->n, 0, 0, &hk, &hv, &hc, &Gj, &ok, &rounds_stats, false, &part, nullptr,
                            preds.n >= 1 ? &pred : nullptr, vk, &hmin, &hmax);
        if (!rc && ok && Gj == 0) ok = false;
        if (!rc && ok) {
            if (G < 0) {
                G = Gj; res->n = G;
                res->cols.resize((size_t)n_aggs + 1);
                res->cols[0].dtype = kdt;
                rc = hark_alloc(ctx, &res->cols[0].data, (size_t)G * 4);
                for (int64_t q = 0; q < n_aggs && !rc; q++) {
                    res->cols[(size_t)q + 1].dtype = plan_of[q].out_dtype;
                    rc = hark_alloc(ctx, &res->cols[(size_t)q + 1].data, (size_t)G * hark_dtype_size(plan_of[q].out_dtype));
                }
                if (!rc) rc = hark_alloc(ctx, (void **)&accg, (size_t)G * 8);
                if (!rc) rc = hark_alloc(ctx, (void **)&cntg, (size_t)G * 8);
            } else if (Gj != G) rc = hark_fail(ctx, HARK_EHIP, "filter_groupby: inconsistent group counts between passes");
            if (!rc && !ming) rc = hark_alloc(ctx, (void **)&ming, (size_t)G * 8);
            if (!rc && !maxg) rc = hark_alloc(ctx, (void **)&maxg, (size_t)G * 8);
            if (!rc) rc = k_argsort_column(ctx, hk, kdt, G, false, &perm, nullptr);       // SQL order: signed for I32
            if (!rc) {
                rc = k_gather(ctx, hk, 4, perm, res->cols[0].data, G);
                if (!rc) rc = k_gather(ctx, hv, 8, perm, accg, G);
                if (!rc) rc = k_gather(ctx, hc, 8, perm, cntg, G);
                if (!rc) rc = k_gather(ctx, hmin, 8, perm, ming, G);
                if (!rc) rc = k_gather(ctx, hmax, 8, perm, maxg, G);
            }
            for (int64_t q = 0; q < n_aggs && !rc; q++) {
                if (done[q]) continue;
                const int kq = cls(plan_of[q]);
                if (plan_of[q].count_only) rc = k_fgb_decode(ctx, accg, cntg, G, plan_of[q].kind, res->cols[(size_t)q + 1].data);
                else if (kq && plan_of[q].col == c) rc = k_fgb_decode(ctx, kq == 1 ? accg : kq == 2 ? maxg : ming, cntg, G, plan_of[q].kind, res->cols[(size_t)q + 1].data);
                else continue;
                done[q] = 1;
            }
        }
        hark_free(ctx, hk); hark_free(ctx, hv); hark_free(ctx, hc); hark_free(ctx, hmin); hark_free(ctx, hmax); hark_free(ctx, perm);
    }
    hark_free(ctx, ming); hark_free(ctx, maxg);
    std::vector<int64_t> order((size_t)n_aggs);
    for (int64_t j = 0; j < n_aggs; j++) order[(size_t)j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {      // ... so passes that can share one partition run back to back
        return plan_of[a].col != plan_of[b].col ? plan_of[a].col < plan_of[b].col : plan_of[a].xf < plan_of[b].xf; });
    for (int64_t oi = 0; oi < n_aggs && !rc && ok; oi++) {
        const int64_t j = order[(size_t)oi];
        if (done[j] || plan_of[j].count_only) continue;
        rc = run_pass(plan_of[j].vop, plan_of[j].xf, src->cols[plan_of[j].col].data);
        for (int64_t q = 0; q < n_aggs && !rc && ok; q++)
            if (!done[q] && (plan_of[q].count_only || (plan_of[q].vop == plan_of[j].vop && plan_of[q].xf == plan_of[j].xf && plan_of[q].col == plan_of[j].col))) {
                rc = k_fgb_decode(ctx, accg, cntg, G, plan_of[q].kind, res->cols[(size_t)q + 1].data);
                done[q] = 1;
            }
    }
    if (!rc && ok && G < 0) {                                  // COUNT only / no aggregate
        rc = run_pass(3, 0, keys);
        for (int64_t q = 0; q < n_aggs && !rc && ok; q++) rc = k_fgb_decode(ctx, accg, cntg, G, plan_of[q].kind, res->cols[(size_t)q + 1].data);
    }
    if (!rc && ok && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "filter_groupby: kernels failed");
    k_fgb_hash_part_free(ctx, &part);
    hark_free(ctx, accg); hark_free(ctx, cntg);
    hark_free(ctx, mask);
    if (!rc && whole_column && db->n < ((int64_t)1 << 32)) {
        if (ok && rounds > 0) kc.hash_rounds = (int32_t)rounds;
        else if (!ok && (why == HARK_HASH_NOFIT_DISTINCT || why == HARK_HASH_NOFIT_SKEW)) kc.hash_rounds = -1;
    }
    if (rc || !ok) {
        for (auto &c : res->cols) if (c.owned && c.data) hark_free(ctx, c.data);
        res->cols.clear(); res->n = 0;
        return rc;
    }
    *used = true;
    return HARK_OK;
}

} // namespace

// ---------------------------------------------------------------------------
// Aggregates of a FEW groups only (late materialisation for ORDER BY ... LIMIT k)
// ---------------------------------------------------------------------------
// A statement whose HAVING / ORDER BY / LIMIT leave k groups needs the aggregates that nobody filters or orders by for
// those k groups only.  The Python planner (harkdb_amd/context.py) therefore runs the full GROUP BY with the aggregates
// HAVING / ORDER BY mention, takes the k surviving keys, and asks this entry for the rest: one streaming pass that reads
// the predicate and key columns (8 B/row), tests every surviving row's key against an open-addressing set of the k keys
// in LDS (one or two LDS reads), and touches value columns only for member rows (k / G of the rows: nothing at
// k = 10, G = 2^20).  Per-workgroup accumulators in LDS (64-bit slot per (key, aggregate), a row count per key), merged
// with global atomics at the end, decoded by fgb_decode_kernel's kinds.  32-bit keys, 4-byte value columns,
// SUM / AVG / MIN / MAX / COUNT; anything else returns HARK_EUNSUPPORTED and the planner takes the one-phase path.
namespace {

constexpr int kSubMaxKeys = 1024, kSubMaxAggs = 8, kSubSlots = 4096, kSubThreads = 1024;
constexpr uint32_t kSubEmpty = 0xFFFFFFFFu;
struct SubAggs { const uint32_t *col[kSubMaxAggs]; int vop[kSubMaxAggs], xf[kSubMaxAggs]; int n; };

typedef unsigned long long su64;
__device__ __forceinline__ uint32_t sub_xf(int xf, uint32_t x)
{
    if (xf == 1) return x ^ 0x80000000u;
    if (xf == 2) { if (x == 0x80000000u) x = 0u; return x ^ ((x & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u); }
    return x;
}
__device__ __forceinline__ su64 sub_identity(int vop) { return vop == 3 ? 0xFFFFFFFFull : 0ull; }
// vop: 0 f32 sum (f64 slot), 5 u64 sum of the transformed word, 2 max / 3 min of the transformed word (low half)
__device__ __forceinline__ void sub_atomic(su64 *slot, int vop, uint32_t w, uint32_t raw)
{
    if (vop == 0) unsafeAtomicAdd(reinterpret_cast<double *>(slot), (double)__uint_as_float(raw));
    else if (vop == 5) atomicAdd(slot, (su64)w);
    else if (vop == 2) atomicMax(reinterpret_cast<uint32_t *>(slot), w);
    else atomicMin(reinterpret_cast<uint32_t *>(slot), w);
}
__device__ __forceinline__ void sub_merge(su64 *slot, int vop, su64 part)
{
    if (vop == 0) unsafeAtomicAdd(reinterpret_cast<double *>(slot), __longlong_as_double((long long)part));
    else if (vop == 5) atomicAdd(slot, part);
    else if (vop == 2) atomicMax(reinterpret_cast<uint32_t *>(slot), (uint32_t)part);
    else atomicMin(reinterpret_cast<uint32_t *>(slot), (uint32_t)part);
}

// pred: cmp = HARK_CMP_MASK -> p is a survivor bitmask (bit r & 7 of byte r >> 3); cmp < 0 -> no predicate; else f32 column
__global__ __launch_bounds__(kSubThreads) void subset_agg_kernel(const float *__restrict__ p, int cmp, float thr, const uint32_t *__restrict__ keys,
                                                                 int64_t n, const uint32_t *__restrict__ want, int nkeys, SubAggs aggs,
                                                                 su64 *__restrict__ gacc /* [naggs][nkeys] */, su64 *__restrict__ gcnt /* [nkeys] */)
{
    __shared__ uint32_t s_key[kSubSlots];
    __shared__ uint16_t s_id[kSubSlots];
    extern __shared__ __attribute__((aligned(16))) unsigned char sub_lds[];
    su64 *s_acc = reinterpret_cast<su64 *>(sub_lds);                              // [naggs][nkeys]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_acc + (size_t)aggs.n * nkeys);   // [nkeys]
    const int tid = threadIdx.x;
    for (int i = tid; i < kSubSlots; i += kSubThreads) s_key[i] = kSubEmpty;
    for (int i = tid; i < aggs.n * nkeys; i += kSubThreads) s_acc[i] = sub_identity(aggs.vop[i / nkeys]);
    for (int i = tid; i < nkeys; i += kSubThreads) s_cnt[i] = 0u;
    __syncthreads();
    auto hash = [](uint32_t k) { k ^= k >> 16; k *= 0x7FEB352Du; k ^= k >> 15; return (k * 0x846CA68Bu) >> 20; };     // 12 bits
    // the k keys are distinct: every thread inserts its own (claim a slot with a compare-and-swap; a stored key of
    // 0xFFFFFFFF is told from "empty" by its id being set)
    for (int i = tid; i < nkeys; i += kSubThreads) {
        const uint32_t k = want[i];
        uint32_t h = hash(k);
        if (k == kSubEmpty) continue;                                              // handled by the flag below
        while (atomicCAS(&s_key[h], kSubEmpty, k) != kSubEmpty) h = (h + 1) & (kSubSlots - 1);
        s_id[h] = (uint16_t)i;
    }
    __shared__ int s_all_ones;                                                     // index of the key 0xFFFFFFFF among the wanted ones, or -1
    if (tid == 0) s_all_ones = -1;
    __syncthreads();
    for (int i = tid; i < nkeys; i += kSubThreads) if (want[i] == kSubEmpty) s_all_ones = i;
    __syncthreads();
    const int all_ones = s_all_ones;
    // A row of a key that is not wanted -- nearly every row: a LIMIT leaves a handful of groups out of a million -- should cost
    // a few instructions, not a hash and a dependent LDS read (the kernel ran at 4.5 TB/s, bound by ~20 vector instructions
    // per surviving row).  Four 32-bit masks, one per 5-bit field of the key's low 20 bits, hold the bits the wanted keys set:
    // a key passes when all four of its fields do -- (k/32)^4 of the rows for k wanted keys: 1 % for ten.
    __shared__ uint32_t s_field[4];
    if (tid < 4) s_field[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < nkeys; i += kSubThreads) {
        const uint32_t k = want[i];
#pragma unroll
        for (int f = 0; f < 4; f++) atomicOr(&s_field[f], 1u << ((k >> (5 * f)) & 31u));
    }
    __syncthreads();
    const uint32_t f0 = s_field[0], f1 = s_field[1], f2 = s_field[2], f3 = s_field[3];
    auto maybe = [&](uint32_t k) -> bool { return ((f0 >> (k & 31u)) & (f1 >> ((k >> 5) & 31u)) & (f2 >> ((k >> 10) & 31u)) & (f3 >> ((k >> 15) & 31u)) & 1u) != 0u; };
    auto lookup = [&](uint32_t k) -> int {
        if (k == kSubEmpty) return all_ones;
        uint32_t h = hash(k);
        for (;;) {
            const uint32_t c = s_key[h];
            if (c == k) return (int)s_id[h];
            if (c == kSubEmpty) return -1;
            h = (h + 1) & (kSubSlots - 1);
        }
    };
    auto member_row = [&](int id, int64_t r) {
        atomicAdd(&s_cnt[id], 1u);
        for (int a = 0; a < aggs.n; a++) {
            if (aggs.vop[a] < 0) continue;                                         // COUNT: the row count is all it needs
            const uint32_t raw = aggs.col[a][r];
            sub_atomic(&s_acc[(size_t)a * nkeys + id], aggs.vop[a], sub_xf(aggs.xf[a], raw), raw);
        }
    };
    const int64_t nvec = n / 4, stride = (int64_t)gridDim.x * kSubThreads;
    // The KEY is tested first (field masks, then the set): the predicate is evaluated for the rows of wanted keys only -- a few
    // thousand scattered 4-byte reads instead of a second 4-byte-per-row stream beside the keys (the pass read 8 B/row before).
    auto survives = [&](int64_t r) -> bool {
        if (cmp < 0) return true;
        if (cmp == HARK_CMP_MASK) return (reinterpret_cast<const uint8_t *>(p)[r >> 3] >> (r & 7)) & 1u;
        const float f = p[r];
        switch (cmp) { case HARK_CMP_GT: return f > thr; case HARK_CMP_GE: return f >= thr; case HARK_CMP_LT: return f < thr;
                       case HARK_CMP_LE: return f <= thr; case HARK_CMP_EQ: return f == thr; default: return f != thr; }
    };
    auto row = [&](uint32_t k, int64_t r) {
        if (!maybe(k)) return;
        const int id = lookup(k);
        if (id >= 0 && survives(r)) member_row(id, r);
    };
    int64_t i = (int64_t)blockIdx.x * kSubThreads + tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {                               // four 16-byte loads in flight per lane
        const uint4 ka = ld_nt16(keys + 4 * i), kb = ld_nt16(keys + 4 * (i + stride)), kc = ld_nt16(keys + 4 * (i + 2 * stride)), kd = ld_nt16(keys + 4 * (i + 3 * stride));
        const uint32_t kk[16] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w, kc.x, kc.y, kc.z, kc.w, kd.x, kd.y, kd.z, kd.w};
#pragma unroll
        for (int j = 0; j < 16; j++) row(kk[j], 4 * (i + (j >> 2) * stride) + (j & 3));
    }
    for (; i < nvec; i += stride) {
        const uint4 ka = ld_nt16(keys + 4 * i);
        const uint32_t kk[4] = {ka.x, ka.y, ka.z, ka.w};
#pragma unroll
        for (int j = 0; j < 4; j++) row(kk[j], 4 * i + j);
    }
    if (blockIdx.x == 0) {                                                         // ragged tail (n % 4 rows)
        const int64_t r = nvec * 4 + tid;
        if (r < n) row(keys[r], r);
    }
    __syncthreads();
    for (int id = tid; id < nkeys; id += kSubThreads) {
        const uint32_t c = s_cnt[id];
        if (!c) continue;
        atomicAdd(&gcnt[id], (su64)c);
        for (int a = 0; a < aggs.n; a++) if (aggs.vop[a] >= 0) sub_merge(&gacc[(size_t)a * nkeys + id], aggs.vop[a], s_acc[(size_t)a * nkeys + id]);
    }
}

__global__ __launch_bounds__(256) void sub_fill_kernel(su64 *dst, int64_t n, su64 v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}

} // namespace

extern "C" int hark_entry_filter_groupby_subset(hark_context *ctx, hark_result **out, const hark_table *db, int64_t n_preds,
                                                const int32_t *where_cols, const int32_t *cmps, const void *const *constants, int32_t g_col,
                                                const void *keys_host, int64_t n_keys, const int32_t *agg_cols, const int32_t *agg_ops, int64_t n_aggs)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (n_preds < 0 || n_aggs < 1 || n_keys < 0 || (n_keys && !keys_host) || !agg_cols || !agg_ops) return hark_fail(ctx, HARK_EARG, "groupby_subset: bad arguments");
    if (g_col < 0 || g_col >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "groupby_subset: key column out of bounds");
    const int kdt = db->cols[g_col].dtype;
    if ((kdt != HARK_I32 && kdt != HARK_U32) || n_keys > kSubMaxKeys || n_aggs > kSubMaxAggs || n_keys * n_aggs > 8192)
        return hark_fail(ctx, HARK_EUNSUPPORTED, "groupby_subset: 32-bit keys, at most %d keys and %d aggregates", kSubMaxKeys, kSubMaxAggs);
    std::vector<DensePass> plan((size_t)n_aggs);
    SubAggs aggs; aggs.n = (int)n_aggs;
    for (int64_t j = 0; j < n_aggs; j++) {
        const int c = agg_ops[j] == HARK_AGG_COUNT ? 0 : agg_cols[j];
        if (c < 0 || c >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "groupby_subset: aggregate column out of bounds");
        if (agg_ops[j] == HARK_AGG_PROD || !dense_plan_for(agg_ops[j], db->cols[c].dtype, c, &plan[j]))
            return hark_fail(ctx, HARK_EUNSUPPORTED, "groupby_subset: SUM / AVG / MIN / MAX / COUNT over 4-byte columns");
        aggs.col[j] = plan[j].count_only ? nullptr : static_cast<const uint32_t *>(db->cols[c].data);
        aggs.vop[j] = plan[j].count_only ? -1 : plan[j].vop;
        aggs.xf[j] = plan[j].xf;
    }
    for (int64_t j = 0; j < n_preds; j++)
        if (where_cols[j] < 0 || where_cols[j] >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "groupby_subset: predicate column out of bounds");
    hark_result *res = new hark_result();
    res->n = n_keys; res->cols.resize((size_t)n_aggs);
    for (int64_t j = 0; j < n_aggs; j++) { res->cols[(size_t)j].dtype = plan[j].out_dtype; res->cols[(size_t)j].data = nullptr; res->cols[(size_t)j].owned = n_keys > 0; }
    if (n_keys == 0) { *out = res; return HARK_OK; }
    // WHERE exactly as try_dense: one f32 predicate rides along, anything else becomes a survivor bitmask
    const bool direct = n_preds == 1 && cmps[0] != HARK_CMP_MASK && db->cols[where_cols[0]].dtype == HARK_F32;
    uint8_t *mask = nullptr;
    uint32_t *want = nullptr; su64 *gacc = nullptr, *gcnt = nullptr;
    hipStream_t st = ctx->stream;
    int rc = HARK_OK;
    if (n_preds >= 1 && !direct) rc = k_predicate_bitmask(ctx, db, n_preds, where_cols, cmps, constants, &mask);
    if (!rc) rc = hark_alloc(ctx, (void **)&want, (size_t)n_keys * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&gacc, (size_t)n_keys * (size_t)n_aggs * 8);
    if (!rc) rc = hark_alloc(ctx, (void **)&gcnt, (size_t)n_keys * 8);
    for (int64_t j = 0; j < n_aggs && !rc; j++) rc = hark_alloc(ctx, &res->cols[(size_t)j].data, (size_t)n_keys * hark_dtype_size(plan[j].out_dtype));
    if (!rc && hipMemcpyAsync(want, keys_host, (size_t)n_keys * 4, hipMemcpyHostToDevice, st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "groupby_subset: key upload failed");
    if (!rc) {
        for (int64_t j = 0; j < n_aggs; j++)
            HARK_LAUNCH_RC(ctx, rc, sub_fill_kernel<<<1, 256, 0, st>>>(gacc + (size_t)j * n_keys, n_keys, plan[j].count_only ? 0ull : (plan[j].vop == 3 ? 0xFFFFFFFFull : 0ull)));
        HIP_TRY_RC(ctx, rc, hipMemsetAsync(gcnt, 0, (size_t)n_keys * 8, st));
        const float *p = direct ? static_cast<const float *>(db->cols[where_cols[0]].data) : reinterpret_cast<const float *>(mask);
        const int cmp = n_preds == 0 ? -1 : direct ? cmps[0] : HARK_CMP_MASK;
        const float thr = direct ? *static_cast<const float *>(constants[0]) : 0.0f;
        const size_t lds = (size_t)n_aggs * n_keys * 8 + (size_t)n_keys * 4 + 16;
        hipError_t he = hipSuccess;
        if (lds > 32 * 1024) he = hipFuncSetAttribute(reinterpret_cast<const void *>(&subset_agg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int64_t grid = (db->n / 4 + kSubThreads - 1) / kSubThreads;
        if (grid > ctx->num_cu) grid = ctx->num_cu;
        if (grid < 1) grid = 1;
        if (he == hipSuccess) HARK_LAUNCH_RC(ctx, rc, subset_agg_kernel<<<dim3((unsigned)grid), dim3(kSubThreads), lds, st>>>(p, cmp, thr, static_cast<const uint32_t *>(db->cols[g_col].data), db->n, want, (int)n_keys, aggs, gacc, gcnt));
        if (!rc && he != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "groupby_subset: setting the dynamic LDS size of subset_agg_kernel failed: %s", hipGetErrorString(he));
        for (int64_t j = 0; j < n_aggs && !rc; j++) rc = k_fgb_decode(ctx, gacc + (size_t)j * n_keys, gcnt, n_keys, plan[j].kind, res->cols[(size_t)j].data);
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "groupby_subset: kernels failed");
    }
    hark_free(ctx, mask); hark_free(ctx, want); hark_free(ctx, gacc); hark_free(ctx, gcnt);
    if (rc) { for (auto &c : res->cols) if (c.owned && c.data) hark_free(ctx, c.data); delete res; return rc; }
    *out = res;
    return HARK_OK;
}

extern "C" int hark_entry_filter_groupby(hark_context *ctx, hark_result **out, const hark_table *db, int32_t where_col, int32_t cmp,
                                         const void *constant, int32_t g_col, const int32_t *agg_cols, const int32_t *agg_ops,
                                         int64_t n_aggs)
{
    // where_col < 0 disables the filter
    return hark_entry_filter_groupby_and(ctx, out, db, where_col >= 0 ? 1 : 0, &where_col, &cmp, &constant, g_col, agg_cols, agg_ops, n_aggs);
}

extern "C" int hark_entry_filter_groupby_and(hark_context *ctx, hark_result **out, const hark_table *db, int64_t n_preds,
                                             const int32_t *where_cols, const int32_t *cmps, const void *const *constants,
                                             int32_t g_col, const int32_t *agg_cols, const int32_t *agg_ops, int64_t n_aggs)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (n_aggs < 0 || (n_aggs && (!agg_cols || !agg_ops))) return hark_fail(ctx, HARK_EARG, "filter_groupby: bad aggregate list");
    if (g_col < 0 || g_col >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby: group column %d out of bounds", g_col);
    if (n_preds < 0 || n_preds > 16 || (n_preds && (!where_cols || !cmps || !constants))) return hark_fail(ctx, HARK_EARG, "filter_groupby: 0..16 predicates");
    for (int64_t j = 0; j < n_preds; j++) {
        if (where_cols[j] < 0 || where_cols[j] >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby: where column %d out of bounds", where_cols[j]);
        if (!constants[j] || cmps[j] < HARK_CMP_GT || cmps[j] > HARK_CMP_MASK) return hark_fail(ctx, HARK_EARG, "filter_groupby: bad predicate");
    }
    const PredList preds{n_preds, where_cols, cmps, constants};
    for (int64_t j = 0; j < n_aggs; j++) {
        if (agg_ops[j] < HARK_AGG_KEY || agg_ops[j] > HARK_AGG_AVG) return hark_fail(ctx, HARK_EARG, "filter_groupby: unknown aggregate opcode %d", agg_ops[j]);
        if (agg_ops[j] != HARK_AGG_COUNT && (agg_cols[j] < 0 || agg_cols[j] >= db->m))
            return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby: aggregate column %d out of bounds", agg_cols[j]);
    }
    hark_result *res = new hark_result();
    int rc = HARK_OK;
    bool done = false;
    ctx->last_groupby_path = HARK_PATH_NONE;
    ctx->last_groupby_passes = 0;
    ctx->last_groupby_window = 0;
    if (db->n > 0) rc = try_dense(ctx, db, preds, g_col, agg_cols, agg_ops, n_aggs, res, &done);
    if (!rc && done) ctx->last_groupby_path = HARK_PATH_DENSE; else ctx->last_groupby_passes = 0;
    if (!rc && !done && db->n > 0) { rc = try_hash(ctx, db, preds, g_col, agg_cols, agg_ops, n_aggs, res, &done); if (!rc && done) ctx->last_groupby_path = HARK_PATH_HASH; }
    if (!rc && !done) {
        ctx->last_groupby_path = HARK_PATH_SORT;
        // generic path: compact the referenced columns, then sort-based typed aggregation
        for (auto &c : res->cols) if (c.owned && c.data) hark_free(ctx, c.data);
        res->cols.clear();
        const hark_table *src = db;
        hark_result *kept = nullptr;
        hark_table view;
        std::vector<int32_t> remap((size_t)db->m, -1), need;
        int32_t g2 = g_col;
        std::vector<int32_t> cols2(agg_cols, agg_cols + n_aggs);
        if (preds.n > 0 && db->n > 0) {
            auto want = [&](int c) { if (remap[c] < 0) { remap[c] = (int32_t)need.size(); need.push_back(c); } return remap[c]; };
            g2 = want(g_col);
            for (int64_t j = 0; j < n_aggs; j++) cols2[j] = agg_ops[j] == HARK_AGG_COUNT ? 0 : want(agg_cols[j]);
            rc = hark_entry_filter_sel_and(ctx, &kept, db, preds.n, preds.cols, preds.cmps, preds.consts, need.data(), (int64_t)need.size(), 0);
            if (!rc) {
                view.n = kept->n; view.m = (int64_t)kept->cols.size(); view.cols = kept->cols;
                for (auto &c : view.cols) c.owned = false;
                src = &view;
            }
        }
        if (!rc) {
            if (src->n == 0) {                       // typed empty result
                res->n = 0; res->cols.resize((size_t)n_aggs + 1);
                for (auto &c : res->cols) { c.dtype = HARK_I64; c.data = nullptr; c.owned = false; }
                res->cols[0].dtype = db->cols[g_col].dtype;
            } else rc = k_groupby_typed(ctx, src, g2, cols2.data(), agg_ops, n_aggs, res);
        }
        if (kept) hark_result_free(ctx, kept);
    }
    if (rc) { for (auto &c : res->cols) if (c.owned && c.data) hark_free(ctx, c.data); delete res; return rc; }
    *out = res;
    return HARK_OK;
}

// WHERE + GROUP BY + HAVING + ORDER BY + LIMIT k as ONE call for dense keys: the aggregates are read out per key slot
// (no group set: no flag / scan kernels, no host read), the k rows are selected from the slots with a non-zero count
// (hark_entry_topk's kernels), and only those k rows become the result [key, aggregates...].  Items: 0 = the key,
// j + 1 = aggregate j.  Ties in the order keep ascending key order, as a stable sort of the compacted groups would.
// HARK_EUNSUPPORTED (without a message): the keys are not dense or k is too large -- the caller composes the statement
// from hark_entry_filter_groupby_and + hark_entry_topk / filter / sort instead.
extern "C" int hark_entry_filter_groupby_topk(hark_context *ctx, hark_result **out, const hark_table *db, int64_t n_preds,
                                              const int32_t *where_cols, const int32_t *cmps, const void *const *constants,
                                              int32_t g_col, const int32_t *agg_cols, const int32_t *agg_ops, int64_t n_aggs,
                                              int64_t n_having, const int32_t *having_items, const int32_t *having_cmps, const void *const *having_consts,
                                              int32_t order_item, int32_t descending, int64_t k)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (n_aggs < 0 || n_aggs > 30 || (n_aggs && (!agg_cols || !agg_ops))) return hark_fail(ctx, HARK_EARG, "filter_groupby_topk: bad aggregate list");
    if (g_col < 0 || g_col >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_topk: group column %d out of bounds", g_col);
    if (n_preds < 0 || n_preds > 16 || (n_preds && (!where_cols || !cmps || !constants))) return hark_fail(ctx, HARK_EARG, "filter_groupby_topk: 0..16 predicates");
    for (int64_t j = 0; j < n_preds; j++) {
        if (where_cols[j] < 0 || where_cols[j] >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_topk: where column %d out of bounds", where_cols[j]);
        if (!constants[j] || cmps[j] < HARK_CMP_GT || cmps[j] > HARK_CMP_MASK) return hark_fail(ctx, HARK_EARG, "filter_groupby_topk: bad predicate");
    }
    for (int64_t j = 0; j < n_aggs; j++) {
        if (agg_ops[j] < HARK_AGG_KEY || agg_ops[j] > HARK_AGG_AVG) return hark_fail(ctx, HARK_EARG, "filter_groupby_topk: unknown aggregate opcode %d", agg_ops[j]);
        if (agg_ops[j] != HARK_AGG_COUNT && (agg_cols[j] < 0 || agg_cols[j] >= db->m))
            return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_topk: aggregate column %d out of bounds", agg_cols[j]);
    }
    if (n_having < 0 || n_having > 7 || (n_having && (!having_items || !having_cmps || !having_consts))) return hark_fail(ctx, HARK_EARG, "filter_groupby_topk: 0..7 HAVING conditions");
    for (int64_t j = 0; j < n_having; j++)
        if (having_items[j] < 0 || having_items[j] > n_aggs || !having_consts[j]) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_topk: HAVING item %d out of bounds", having_items[j]);
    if (order_item < 0 || order_item > n_aggs) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_topk: ORDER BY item %d out of bounds", order_item);
    if (k < 1 || db->n <= 0) return HARK_EUNSUPPORTED;
    // a COUNT tells the empty slots from the groups: appended when the statement has none
    std::vector<int32_t> cols2(agg_cols, agg_cols + n_aggs), ops2(agg_ops, agg_ops + n_aggs);
    int32_t count_item = -1;
    for (int64_t j = 0; j < n_aggs; j++) if (agg_ops[j] == HARK_AGG_COUNT) { count_item = (int32_t)j + 1; break; }
    if (count_item < 0) { cols2.push_back(0); ops2.push_back(HARK_AGG_COUNT); count_item = (int32_t)n_aggs + 1; }
    const PredList preds{n_preds, where_cols, cmps, constants};
    hark_result *slots = new hark_result();
    bool done = false;
    int rc = try_dense(ctx, db, preds, g_col, cols2.data(), ops2.data(), (int64_t)ops2.size(), slots, &done, true);
    if (!rc && !done) rc = HARK_EUNSUPPORTED;
    if (!rc && slots->cols[(size_t)count_item].dtype != HARK_I64) rc = hark_fail(ctx, HARK_EHIP, "filter_groupby_topk: COUNT is not an int64 column");
    if (!rc) {
        hark_table view;
        view.n = slots->n; view.m = (int64_t)slots->cols.size(); view.cols = slots->cols;
        for (auto &c : view.cols) c.owned = false;
        const int64_t zero = 0;
        std::vector<int32_t> pcols{count_item}, pcmps{HARK_CMP_GT}, outcols;
        std::vector<const void *> pconsts{&zero};
        for (int64_t j = 0; j < n_having; j++) { pcols.push_back(having_items[j]); pcmps.push_back(having_cmps[j]); pconsts.push_back(having_consts[j]); }
        for (int64_t j = 0; j <= n_aggs; j++) outcols.push_back((int32_t)j);
        (void)hipGetLastError();
        rc = hark_entry_topk(ctx, out, &view, (int64_t)pcols.size(), pcols.data(), pcmps.data(), pconsts.data(), order_item, descending, k, outcols.data(), (int64_t)outcols.size());
    }
    for (auto &c : slots->cols) if (c.owned && c.data) hark_free(ctx, c.data);
    delete slots;
    return rc;
}


// WHERE + GROUP BY over the dense key domain [0, G), one result row PER KEY SLOT (row g = key g): the partial aggregates
// of one shard of a row-range-sharded table, laid out so that the shards' results merge elementwise -- SUM and COUNT by
// addition, MIN / MAX by min / max -- with an RCCL all-reduce (harkdb_amd/dist.py, SURVEY.md 8(e) "GROUP BY, dense key
// domain").  Result: [aggregate 0, ..., aggregate n-1, COUNT(*)] (the trailing count is always there: a slot with count 0
// is an empty group and its other aggregates are unspecified).  HARK_EUNSUPPORTED when the shape is not the fused dense
// one (non-32-bit key, 8-byte value columns, G > 2^21).
extern "C" int hark_entry_filter_groupby_slots(hark_context *ctx, hark_result **out, const hark_table *db, int64_t n_preds,
                                               const int32_t *where_cols, const int32_t *cmps, const void *const *constants,
                                               int32_t g_col, int64_t G, const int32_t *agg_cols, const int32_t *agg_ops, int64_t n_aggs)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (n_aggs < 0 || n_aggs > 30 || (n_aggs && (!agg_cols || !agg_ops))) return hark_fail(ctx, HARK_EARG, "filter_groupby_slots: bad aggregate list");
    if (g_col < 0 || g_col >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_slots: group column %d out of bounds", g_col);
    if (G < 1) return hark_fail(ctx, HARK_EARG, "filter_groupby_slots: G must be positive");
    if (n_preds < 0 || n_preds > 16 || (n_preds && (!where_cols || !cmps || !constants))) return hark_fail(ctx, HARK_EARG, "filter_groupby_slots: 0..16 predicates");
    for (int64_t j = 0; j < n_preds; j++) {
        if (where_cols[j] < 0 || where_cols[j] >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_slots: where column %d out of bounds", where_cols[j]);
        if (!constants[j] || cmps[j] < HARK_CMP_GT || cmps[j] > HARK_CMP_MASK) return hark_fail(ctx, HARK_EARG, "filter_groupby_slots: bad predicate");
    }
    for (int64_t j = 0; j < n_aggs; j++) {
        if (agg_ops[j] < HARK_AGG_KEY || agg_ops[j] > HARK_AGG_AVG || agg_ops[j] == HARK_AGG_AVG || agg_ops[j] == HARK_AGG_PROD)
            return hark_fail(ctx, HARK_EARG, "filter_groupby_slots: SUM / MIN / MAX / COUNT only (AVG travels as SUM and COUNT), got opcode %d", agg_ops[j]);
        if (agg_ops[j] != HARK_AGG_COUNT && (agg_cols[j] < 0 || agg_cols[j] >= db->m))
            return hark_fail(ctx, HARK_EBOUNDS, "filter_groupby_slots: aggregate column %d out of bounds", agg_cols[j]);
    }
    std::vector<int32_t> cols2(agg_cols, agg_cols + n_aggs), ops2(agg_ops, agg_ops + n_aggs);
    cols2.push_back(0); ops2.push_back(HARK_AGG_COUNT);
    const PredList preds{n_preds, where_cols, cmps, constants};
    hark_result *slots = new hark_result();
    bool done = false;
    int rc = try_dense(ctx, db, preds, g_col, cols2.data(), ops2.data(), (int64_t)ops2.size(), slots, &done, true, G);
    if (!rc && !done) rc = HARK_EUNSUPPORTED;
    if (rc) { for (auto &c : slots->cols) if (c.owned && c.data) hark_free(ctx, c.data); delete slots; return rc; }
    // drop the key column (row g IS key g)
    if (slots->cols[0].owned && slots->cols[0].data) hark_free(ctx, slots->cols[0].data);
    slots->cols.erase(slots->cols.begin());
    *out = slots;
    return HARK_OK;
}


// ---------------------------------------------------------------------------
// Reference entry over dense keys: one fused pass per aggregate column
// ---------------------------------------------------------------------------
namespace {

// every result column of the dense path in one launch: column 0 = the keys of the non-empty slots, column j = vals[j - 1]
struct DenseEmit { const uint32_t *vals[kSegMaxAggs + 1]; uint32_t *out[kSegMaxAggs + 1]; int32_t n; };
__global__ __launch_bounds__(256) void dense_emit_all_u32_kernel(DenseEmit e, const unsigned long long *__restrict__ acc_cnt,
                                                                 const uint32_t *__restrict__ pos, int64_t G)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
        if (!acc_cnt[g]) continue;
        const uint32_t at = pos[g];
        for (int j = 0; j < e.n; j++) e.out[j][at] = e.vals[j] ? e.vals[j][g] : (uint32_t)g;
    }
}

int ref_groupby_dense(hark_context *ctx, const hark_table *view, const hark_table *stats_owner, int g_col,
                      const std::vector<AggSpec> &aggs, hark_result *res, int64_t *G_out, bool *used)
{
    *used = false;
    const int64_t n = view->n;
    const uint32_t *keys = static_cast<const uint32_t *>(view->cols[g_col].data);
    // column statistic: largest key (unsigned), cached in the column
    int64_t kmin = 0, kmax = 0;
    HARK_TRY(column_range(ctx, stats_owner, g_col, false, &kmin, &kmax));     // cached on the caller's table
    int rc = HARK_OK;
    const int64_t G = kmax + 1;
    if (G > kDenseMaxGroups || G > 8 * n + 4096) return HARK_OK;       // sparse or huge key domain: sort-based path

    hark_fgb_plan *plan = nullptr;
    HARK_TRY(k_fgb_plan_new_uncleared(ctx, &plan, n, G));                 // (every pass below resets or initialises the accumulators itself)
    const bool own_column = view->cols[g_col].data == stats_owner->cols[g_col].data && view->n == stats_owner->n;   // (a filtered view has its own rows)
    if (own_column) { plan->win_k = keys; plan->win_n = n; plan->win_verdict = stats_owner->cols[g_col].key_clustered; }
    std::vector<uint32_t *> vals(aggs.size(), nullptr);
    uint32_t *flags = nullptr, *pos = nullptr;
    int64_t ngroups = 0;
    const size_t runs = aggs.empty() ? 1 : aggs.size();                 // no aggregate: one pass just for the counts
    auto vop_of = [&](size_t j) { return aggs[j].op == OP_SUM ? 1 : aggs[j].op == OP_MAX ? 2 : aggs[j].op == OP_MIN ? 3 : 4; };
    std::vector<char> served(runs, 0);
    // statistics pass (k_fgb_dense_stats): two or more of {sum, max, min} of ONE column from one producer + consumer pass
    // that carries the column once (6-byte pairs; the 64-bit sum's low word is the reference's sum mod 2^32, groupby.fut:37).
    // Declines for small G, > 4096 keys per bucket or skew; the passes below then take over.
    for (size_t j = 0; j < aggs.size() && !rc; j++) {
        if (served[j] || vop_of(j) == 4) continue;
        int classes = 0;
        for (size_t t = 0; t < aggs.size(); t++) if (!served[t] && aggs[t].col == aggs[j].col && vop_of(t) != 4) classes |= 1 << vop_of(t);
        if (__builtin_popcount(classes) < 2) continue;
        bool ran = false;
        rc = hark_fgb_plan_set(plan, "vop", 0);
        if (!rc) rc = hark_fgb_plan_set(plan, "xform", 0);
        if (!rc) rc = k_fgb_dense_stats(ctx, plan, nullptr, 0, 0.0f, reinterpret_cast<const int32_t *>(keys), view->cols[aggs[j].col].data, n, 2, &ran);
        if (rc || !ran) break;
        int64_t blocks = (G + 255) / 256;
        if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
        for (size_t t = 0; t < aggs.size() && !rc; t++) {
            if (served[t] || aggs[t].col != aggs[j].col || vop_of(t) == 4) continue;
            rc = hark_alloc(ctx, (void **)&vals[t], (size_t)G * 4);
            if (!rc) rc = hark_fgb_finish_u32_of(ctx, plan, vop_of(t) == 1 ? 0 : vop_of(t) == 2 ? 2 : 1, vals[t]);
            served[t] = 1;
        }
    }
    // triple passes (k_fgb_dense_multi): a sum / max / min with two other max / min from ONE producer + consumer pass (14-byte
    // entries; every row survives here, so the plan's slabs are sized for them); declines for small G or skew
    while (!rc) {
        const size_t none = aggs.size();
        size_t j = none, q1 = none, q2 = none;
        auto ext = [&](size_t t) { return vop_of(t) == 2 || vop_of(t) == 3; };
        for (size_t t = 0; t < aggs.size() && j == none; t++) if (!served[t] && vop_of(t) == 1) j = t;
        for (size_t t = 0; t < aggs.size() && j == none; t++) if (!served[t] && ext(t)) j = t;
        for (size_t t = 0; j != none && t < aggs.size() && q1 == none; t++) if (t != j && !served[t] && ext(t)) q1 = t;
        for (size_t t = 0; q1 != none && t < aggs.size() && q2 == none; t++) if (t != j && t != q1 && !served[t] && ext(t)) q2 = t;
        if (q2 == none) break;
        if (plan->slack_pct < 230) rc = hark_fgb_plan_set(plan, "slack_pct", 230);
        if (rc) break;
        bool ran = false;
        rc = k_fgb_dense_multi(ctx, plan, nullptr, 0, 0.0f, reinterpret_cast<const int32_t *>(keys), view->cols[aggs[j].col].data, vop_of(j), 0,
                               view->cols[aggs[q1].col].data, vop_of(q1), 0, view->cols[aggs[q2].col].data, vop_of(q2), 0, n, &ran);
        if (rc || !ran) break;
        for (size_t t : {j, q1, q2}) if (!rc) rc = hark_alloc(ctx, (void **)&vals[t], (size_t)G * 4);
        if (!rc) rc = hark_fgb_finish_u32(ctx, plan, vals[j], nullptr);
        if (!rc) rc = hark_fgb_finish_u32_of(ctx, plan, 1, vals[q1]);
        if (!rc) rc = hark_fgb_finish_u32_of(ctx, plan, 2, vals[q2]);
        served[j] = served[q1] = served[q2] = 1;
    }
    // pair passes (k_fgb_dense_pair): a sum / max / min with another max / min from ONE producer + consumer pass; value 2
    // must be a max or min (a 32-bit LDS slot), products keep their own pass.  Declines for small G or skew.
    for (size_t j = 0; j < aggs.size() && !rc; j++) {
        if (served[j] || vop_of(j) == 4) continue;
        size_t q = aggs.size();
        for (size_t t = 0; t < aggs.size(); t++) if (t != j && !served[t] && (vop_of(t) == 2 || vop_of(t) == 3)) { q = t; break; }
        if (q == aggs.size()) continue;
        bool ran = false;
        rc = k_fgb_dense_pair(ctx, plan, nullptr, 0, 0.0f, reinterpret_cast<const int32_t *>(keys), view->cols[aggs[j].col].data, vop_of(j), 0,
                              view->cols[aggs[q].col].data, vop_of(q), 0, n, &ran);
        if (rc || !ran) break;
        rc = hark_alloc(ctx, (void **)&vals[j], (size_t)G * 4);
        if (!rc) rc = hark_alloc(ctx, (void **)&vals[q], (size_t)G * 4);
        if (!rc) rc = hark_fgb_finish_u32(ctx, plan, vals[j], nullptr);
        if (!rc) rc = hark_fgb_finish_u32_second(ctx, plan, vals[q]);
        served[j] = served[q] = 1;
    }
    for (size_t j = 0; j < runs && !rc; j++) {
        if (!aggs.empty() && served[j]) continue;
        const int vop = aggs.empty() ? 3 : vop_of(j);
        const uint32_t *col = aggs.empty() ? keys : static_cast<const uint32_t *>(view->cols[aggs[j].col].data);
        rc = hark_fgb_plan_set(plan, "vop", vop);
        if (!rc) rc = hark_fgb_reset(ctx, plan);
        if (!rc) rc = hark_op_groupby_dense_u32(ctx, plan, keys, col, n);
        if (!rc && !aggs.empty()) rc = hark_alloc(ctx, (void **)&vals[j], (size_t)G * 4);
        if (!rc) rc = hark_fgb_finish_u32(ctx, plan, aggs.empty() ? nullptr : vals[j], nullptr);
    }
    if (!rc) rc = hark_alloc(ctx, (void **)&flags, (size_t)G * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&pos, (size_t)G * 4);
    if (!rc) {
        HARK_LAUNCH_RC(ctx, rc, nonzero_flags_kernel<<<grid_for(ctx, G), 256, 0, ctx->stream>>>(plan->acc_cnt, G, flags));
        if (!rc) rc = k_exclusive_scan_u32(ctx, flags, G, pos, nullptr, &ngroups);
    }
    if (!rc) {
        res->n = ngroups;
        res->cols.resize(aggs.size() + 1);
        for (size_t j = 0; j <= aggs.size() && !rc; j++) {
            res->cols[j].dtype = HARK_U32;
            rc = hark_alloc(ctx, &res->cols[j].data, (size_t)ngroups * 4);
        }
        for (size_t j0 = 0; j0 <= aggs.size() && !rc; j0 += kSegMaxAggs + 1) {      // (kSegMaxAggs + 1 columns per launch)
            DenseEmit e{};
            for (size_t j = j0; j <= aggs.size() && e.n <= kSegMaxAggs; j++) {
                e.vals[e.n] = j == 0 ? nullptr : vals[j - 1];
                e.out[e.n++] = static_cast<uint32_t *>(res->cols[j].data);
            }
            HARK_LAUNCH_RC(ctx, rc, dense_emit_all_u32_kernel<<<grid_for(ctx, G), 256, 0, ctx->stream>>>(e, plan->acc_cnt, pos, G));
        }
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "query_groupby: kernels failed");
    }
    for (auto v : vals) hark_free(ctx, v);
    hark_free(ctx, flags); hark_free(ctx, pos);
    if (own_column && plan->win_k == keys && plan->win_verdict >= 0) stats_owner->cols[g_col].key_clustered = (int8_t)plan->win_verdict;
    ctx->last_groupby_window = plan->win_rows > 0 ? 1 : plan->rot_rows > 0 ? 2 : 0;
    hark_fgb_plan_free(ctx, plan);
    if (rc) { for (auto &c : res->cols) { if (c.owned && c.data) hark_free(ctx, c.data); } res->cols.clear(); return rc; }
    *G_out = ngroups;
    *used = true;
    return HARK_OK;
}

} // namespace


// ---------------------------------------------------------------------------
// Reference entry over sparse keys: hash partition + LDS hash tables, one pass per aggregate
// ---------------------------------------------------------------------------
namespace {

constexpr int64_t kHashMinRows = (int64_t)1 << 18;          // below this the sort-based path is as good

// out[i] = 32-bit word `word` (0 low, 1 high) of vals[perm[i]]
__global__ __launch_bounds__(256) void gather_word32_kernel(const unsigned long long *__restrict__ vals, const uint32_t *__restrict__ perm,
                                                            uint32_t *__restrict__ out, int64_t n, int word)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(vals);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = w[2 * (size_t)perm[i] + word];
}

int ref_groupby_hash(hark_context *ctx, const hark_table *view, const hark_table *stats_owner, int g_col, const std::vector<AggSpec> &aggs,
                     hark_result *res, int64_t *G_out, bool *used)
{
    *used = false;
    const int64_t n = view->n;
    if (n < kHashMinRows) return HARK_OK;
    // what an earlier call learnt about this key column (tables are immutable; hark_table_invalidate_stats forgets it): too many distinct keys -> no second attempt
    // (a failed one costs a partition pass and a sample round: 1.1 ms per 1e8 rows); the rounds it needs -> no failed first round
    const hark_column &kc = stats_owner->cols[g_col];
    if (kc.hash_rounds < 0) return HARK_OK;
    const uint32_t *keys = static_cast<const uint32_t *>(view->cols[g_col].data);
    const size_t runs = aggs.empty() ? 1 : aggs.size();
    int64_t G = -1;
    int rc = HARK_OK;
    bool ok = true;
    uint32_t rounds = kc.hash_rounds > 0 ? (uint32_t)kc.hash_rounds : 0u;   // table rounds the key column needs: found by the first pass, reused
    hark_hash_part part;                                     // aggregates of one column after another share the (key, value) partition
    std::vector<size_t> order(runs);
    for (size_t q = 0; q < runs; q++) order[q] = q;
    if (!aggs.empty()) std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return aggs[a].col < aggs[b].col; });   // ... so column by column
    bool first_pass = true;
    int why = HARK_HASH_FITS;
    auto vop_of = [&](size_t j) { return aggs.empty() ? 3 : aggs[j].op == OP_SUM ? 1 : aggs[j].op == OP_MAX ? 2 : aggs[j].op == OP_MIN ? 3 : 4; };
    // aggregates of one column come from ONE consumer pass, two or three operators at a time (fgb_agg_hash_ops_kernel), with
    // ONE sort of the result keys per pass; a single operator runs the same consumer with one slot per entry
    for (size_t oi = 0; oi < runs && !rc && ok; ) {
        size_t cnt = 1;
        if (!aggs.empty() && !getenv("HARK_NO_HASH_OPS_PASS"))
            while (cnt < 3 && oi + cnt < runs && aggs[order[oi + cnt]].col == aggs[order[oi]].col) cnt++;
        const size_t j0 = order[oi];
        const uint32_t *col = aggs.empty() ? keys : static_cast<const uint32_t *>(view->cols[aggs[j0].col].data);
        uint32_t ops = 0;
        for (size_t q = 0; q < cnt && cnt > 1; q++) ops |= (uint32_t)vop_of(order[oi + q]) << (8 * q);
        uint32_t *hk = nullptr, *perm = nullptr; unsigned long long *hv = nullptr, *hc = nullptr;
        int64_t Gj = 0;
        rc = k_fgb_hash_u32(ctx, keys, col, n, vop_of(j0), 0, &hk, &hv, &hc, &Gj, &ok, &rounds, true, &part, &why, nullptr, -1, nullptr, nullptr, ops);   // u32 operators, no counts
        if (!rc && ok) {
            if (G < 0) {
                G = Gj;
                res->n = G;
                res->cols.resize(aggs.size() + 1);
                for (auto &c : res->cols) { c.dtype = HARK_U32; c.data = nullptr; c.owned = true; }
                for (size_t q = 0; q <= aggs.size() && !rc; q++) rc = hark_alloc(ctx, &res->cols[q].data, (size_t)G * 4);
            } else if (Gj != G) rc = hark_fail(ctx, HARK_EHIP, "query_groupby: inconsistent group counts between passes");
            // the passes emit in table order of the hash buckets: bring every pass into ascending key order
            if (!rc) rc = k_argsort_column(ctx, hk, HARK_U32, G, false, &perm, nullptr);
            if (!rc && G > 0) {
                if (first_pass) rc = k_gather(ctx, hk, 4, perm, res->cols[0].data, G);
                first_pass = false;
                for (size_t q = 0; q < cnt && !rc && !aggs.empty(); q++)       // slot 0: low word of hv, slot 1: its high word, slot 2: low word of hc
                    HARK_LAUNCH_RC(ctx, rc, gather_word32_kernel<<<grid_for(ctx, G), 256, 0, ctx->stream>>>(q < 2 ? hv : hc, perm, static_cast<uint32_t *>(res->cols[order[oi + q] + 1].data), G, q == 1 ? 1 : 0));
            }
        }
        hark_free(ctx, hk); hark_free(ctx, hv); hark_free(ctx, hc); hark_free(ctx, perm);
        oi += cnt;
    }
    k_fgb_hash_part_free(ctx, &part);
    // the verdict stays with the key column (hark_internal.h: sticky until hark_table_invalidate_stats) -- unless nothing
    // was learnt about the column (row count outside the kernels' range)
    if (!rc && (ok || why == HARK_HASH_NOFIT_DISTINCT || why == HARK_HASH_NOFIT_SKEW)) kc.hash_rounds = ok ? (int32_t)rounds : -1;
    if (!rc && ok && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "query_groupby: kernels failed");
    if (rc || !ok) {
        for (auto &c : res->cols) if (c.owned && c.data) hark_free(ctx, c.data);
        res->cols.clear(); res->n = 0;
        return rc;
    }
    *G_out = G;
    *used = true;
    return HARK_OK;
}

} // namespace

// ---------------------------------------------------------------------------
// Multi-key GROUP BY (extension): the key columns are folded into ONE composite key column,
// ((k1 - min1) * span2 + (k2 - min2)) * span3 + ..., whose ascending order is the lexicographic order of the key
// tuple, so every single-key path above (dense, hash, sort-based) serves it unchanged.  The per-column [min, max]
// are the cached column statistics; the caller decodes the composite back into the key columns (G rows).
// ---------------------------------------------------------------------------
namespace {
constexpr int kMaxKeys = 4;
struct CompositeArgs { const uint32_t *col[kMaxKeys]; int is_signed[kMaxKeys]; long long mn[kMaxKeys]; long long span[kMaxKeys]; int nk; };

template <typename OUT>
__global__ __launch_bounds__(256) void composite_key_kernel(CompositeArgs a, int64_t n, OUT *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long acc = 0;
        for (int j = 0; j < a.nk; j++) {
            const uint32_t x = a.col[j][i];
            const long long v = a.is_signed[j] ? (long long)(int32_t)x : (long long)x;
            acc = acc * a.span[j] + (v - a.mn[j]);
        }
        out[i] = (OUT)acc;
    }
}
} // namespace

extern "C" int hark_context_last_groupby_path(const hark_context *ctx) { return ctx ? ctx->last_groupby_path : HARK_PATH_NONE; }
extern "C" int hark_context_last_groupby_passes(const hark_context *ctx) { return ctx ? ctx->last_groupby_passes : 0; }
extern "C" int hark_context_last_groupby_window(const hark_context *ctx) { return ctx ? ctx->last_groupby_window : 0; }
extern "C" int hark_context_last_join_path(const hark_context *ctx) { return ctx ? ctx->last_join_path : HARK_PATH_NONE; }

extern "C" int hark_table_invalidate_stats(hark_context *ctx, const hark_table *t, int32_t col)
{
    if (!ctx || !t) return HARK_EARG;
    if (col >= t->m) return hark_fail(ctx, HARK_EBOUNDS, "invalidate_stats: column %d out of bounds", col);
    for (int64_t j = 0; j < t->m; j++) if (col < 0 || j == col) t->cols[(size_t)j].invalidate_stats();
    return HARK_OK;
}

extern "C" int hark_table_column_range(hark_context *ctx, const hark_table *t, int32_t col, int64_t *lo, int64_t *hi)
{
    if (!ctx || !t || !lo || !hi) return HARK_EARG;
    if (col < 0 || col >= t->m) return hark_fail(ctx, HARK_EBOUNDS, "column_range: column %d out of bounds", col);
    const int dt = t->cols[col].dtype;
    if (dt != HARK_I32 && dt != HARK_U32) return hark_fail(ctx, HARK_EUNSUPPORTED, "column_range: column %d is not a 32-bit integer column", col);
    if (t->n == 0) return hark_fail(ctx, HARK_EARG, "column_range: the table has no rows");
    return column_range(ctx, t, col, dt == HARK_I32, lo, hi);
}

extern "C" int hark_table_composite_key(hark_context *ctx, const hark_table *t, const int32_t *cols, int64_t nk,
                                        void **out_dev, int32_t *out_dtype, int64_t *mins, int64_t *spans, int32_t given)
{
    if (!ctx || !t || !cols || !out_dev || !out_dtype || !mins || !spans) return HARK_EARG;
    *out_dev = nullptr;
    if (nk < 1 || nk > kMaxKeys) return hark_fail(ctx, HARK_EUNSUPPORTED, "composite key: 1..%d key columns", kMaxKeys);
    CompositeArgs a{};
    a.nk = (int)nk;
    long double total = 1.0L;
    for (int64_t j = 0; j < nk; j++) {
        const int c = cols[j];
        if (c < 0 || c >= t->m) return hark_fail(ctx, HARK_EBOUNDS, "composite key: column %d out of bounds", c);
        const int dt = t->cols[c].dtype;
        if (dt != HARK_I32 && dt != HARK_U32) return hark_fail(ctx, HARK_EUNSUPPORTED, "composite key: column %d is not a 32-bit integer column", c);
        int64_t lo = 0, hi = 0;
        if (given) {                                          // the caller's ranges (e.g. the all-rank ranges of a sharded table)
            if (spans[j] < 1) return hark_fail(ctx, HARK_EARG, "composite key: given span of key %lld is < 1", (long long)j);
            lo = mins[j]; hi = mins[j] + spans[j] - 1;
        } else if (t->n > 0) HARK_TRY(column_range(ctx, t, c, dt == HARK_I32, &lo, &hi));
        a.col[j] = static_cast<const uint32_t *>(t->cols[c].data);
        a.is_signed[j] = dt == HARK_I32;
        a.mn[j] = lo; a.span[j] = hi - lo + 1;
        mins[j] = lo; spans[j] = hi - lo + 1;
        total *= (long double)(hi - lo + 1);
    }
    if (total >= 4.0e18L) return hark_fail(ctx, HARK_EUNSUPPORTED, "composite key: the key ranges multiply to more than 2^62");
    const bool wide = total > 2147483647.0L;
    *out_dtype = wide ? HARK_I64 : HARK_I32;
    if (t->n == 0) return HARK_OK;
    HARK_TRY(hark_alloc(ctx, out_dev, (size_t)t->n * (wide ? 8 : 4)));
    int rc = HARK_OK;
    if (wide) HARK_LAUNCH_RC(ctx, rc, composite_key_kernel<long long><<<grid_for(ctx, t->n), 256, 0, ctx->stream>>>(a, t->n, static_cast<long long *>(*out_dev)));
    else HARK_LAUNCH_RC(ctx, rc, composite_key_kernel<int32_t><<<grid_for(ctx, t->n), 256, 0, ctx->stream>>>(a, t->n, static_cast<int32_t *>(*out_dev)));
    if (rc) { hark_free(ctx, *out_dev); *out_dev = nullptr; }
    return rc;
}
