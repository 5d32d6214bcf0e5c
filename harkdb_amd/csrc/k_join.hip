// k_join.hip -- inner equi-join on one u32 key per side, and ORDER BY.
//
//   hark_entry_join  replaces futhark/join.fut:52-75.  The reference tags and
//       concatenates both key columns, sorts the (key, tag, row) triples with
//       32 one-bit passes (join.fut:9-23, :58), finds key segments (:59-63) and
//       then runs a SEQUENTIAL loop over the distinct keys that builds each
//       cross product with partition + expand and concat-s it to the
//       accumulator (:64-68).  Output order: ascending unsigned key, then left
//       row id, then right row id.
//       Here: both sides are argsorted independently (stable, so row ids stay
//       ascending inside a key), every left row binary-searches its match
//       range in the sorted right keys (count phase), a prefix sum gives each
//       left row its output offset, and one thread per OUTPUT row finds its
//       left row by binary search in the offsets (write phase).  That yields
//       exactly the reference's order with no sequential step, and skewed keys
//       cost nothing extra because work is split by output row.
//   hark_entry_sort  ORDER BY one column (README.md:15 lists it; the reference
//       has no implementation): stable radix sort of the key's sort words; a u32 / i32
//       key column in the output comes back from the sorted words, a single other 4-byte
//       column travels with the keys as the payload, anything else is gathered through
//       the row-id permutation.
#include "hark_internal.h"

int k_argsort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending,
                     uint32_t **perm_out, uint32_t **sorted_words_out);
int k_gather(hark_context *ctx, const void *src, int esz, const uint32_t *idx, void *dst, int64_t n);
int k_argsort_i64_keys(hark_context *ctx, const void *col, int64_t n, uint32_t **perm_out, uint64_t **keys_out, const uint32_t *valcol, uint32_t **val_out, int *unique_out, bool *plain_out = nullptr, int8_t *msd_unfit = nullptr);
int k_argsort_i64_desc_tuples(hark_context *ctx, const void *col, int64_t n, uint32_t **perm_out, uint64_t **keys_out, const uint32_t *valcol, uint32_t **val_out, bool *done, uint64_t out_xor = 0, int8_t *msd_unfit = nullptr);
int k_sort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending, const uint32_t *payload,
                  uint32_t **vals_out, uint32_t **words_out);
int k_exclusive_scan_u32(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, int64_t *total_host);
// k_hjoin.hip: the partitioned path (probe side range-partitioned by splitters of the sorted build side, build slices in LDS)
int k_join_partitioned(hark_context *ctx, const void *lcol, bool k64, int64_t n, const void *rkeys, int64_t s, const uint32_t *lval, const uint32_t *rranked,
                       uint32_t **rank_out, uint32_t **lrow_out, uint32_t **cnt_out, uint32_t **lval_out, uint32_t **rval_out, int64_t *m_out, bool *used, bool *unique,
                       bool rows_needed, int build_unique);

namespace {

// Each thread owns kChunk consecutive (sorted) left keys: one binary search positions it in
// the sorted right keys, then it walks both sides forward -- sequential reads instead of
// one full binary search per left row.
constexpr int kJoinChunk = 8;
template <typename K>
__global__ __launch_bounds__(256) void join_count_kernel(const K *__restrict__ lkeys, int64_t n,
                                                         const K *__restrict__ rkeys, int64_t s,
                                                         uint32_t *__restrict__ lb_out, uint32_t *__restrict__ cnt_out)
{
    const int64_t nchunks = (n + kJoinChunk - 1) / kJoinChunk;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += stride) {
        const int64_t i0 = c * kJoinChunk, i1 = i0 + kJoinChunk < n ? i0 + kJoinChunk : n;
        const K first = lkeys[i0];
        int64_t lo = 0, hi = s;                       // lower bound of the chunk's first key
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rkeys[mid] < first) lo = mid + 1; else hi = mid; }
        int64_t lb = lo, ub = lo;
        K prev = first;
        {   // upper bound of the first key
            int64_t a = lo, b = s;
            while (a < b) { const int64_t mid = (a + b) >> 1; if (rkeys[mid] <= first) a = mid + 1; else b = mid; }
            ub = a;
        }
        for (int64_t i = i0; i < i1; i++) {
            const K key = lkeys[i];
            if (key != prev) {                        // left keys ascend: continue from the previous upper bound
                lb = ub;
                int steps = 0;
                while (lb < s && rkeys[lb] < key && steps < 16) { lb++; steps++; }
                if (lb < s && rkeys[lb] < key) {                     // far away: binary search the rest
                    int64_t a = lb, b = s;
                    while (a < b) { const int64_t mid = (a + b) >> 1; if (rkeys[mid] < key) a = mid + 1; else b = mid; }
                    lb = a;
                }
                ub = lb; steps = 0;
                while (ub < s && rkeys[ub] == key && steps < 16) { ub++; steps++; }
                if (ub < s && rkeys[ub] == key) {                    // a long run of equal right keys
                    int64_t a = ub, b = s;
                    while (a < b) { const int64_t mid = (a + b) >> 1; if (rkeys[mid] <= key) a = mid + 1; else b = mid; }
                    ub = a;
                }
                prev = key;
            }
            lb_out[i] = (uint32_t)lb;
            cnt_out[i] = (uint32_t)(ub - lb);
        }
    }
}

// One lane per matching left row writes its partner pairs itself while it has few of them (the common case: the
// stores of consecutive lanes are consecutive); a row with many partners is expanded by its whole wave, 64 pairs
// per step -- so neither the output size nor skewed keys unbalance the work.
__global__ __launch_bounds__(256) void join_expand_kernel(const int64_t *__restrict__ offs, const uint32_t *__restrict__ cnt, int64_t n,
                                                          const uint32_t *__restrict__ lb, const uint32_t *__restrict__ lperm,
                                                          const uint32_t *__restrict__ rperm, uint32_t *__restrict__ lrow,
                                                          uint32_t *__restrict__ rrow /* may be null */,
                                                          uint32_t *__restrict__ kpos /* may be null: position of the pair's key in the sorted build keys */,
                                                          const uint32_t *__restrict__ lval /* may be null: a left column's value per matching row ... */,
                                                          uint32_t *__restrict__ lval_out /* ... repeated for every pair of the row */)
{
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t base = ((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 6); base < n; base += nwaves * 64) {
        const int64_t i = base + lane;
        uint32_t c = 0, first = 0, left = 0, lv = 0;
        int64_t o = 0;
        if (i < n) { c = cnt[i]; o = offs[i]; first = lb[i]; left = lperm[i]; if (lval) lv = lval[i]; }
        if (c <= 8u) for (uint32_t j = 0; j < c; j++) {
            lrow[o + j] = left; if (rrow) rrow[o + j] = rperm[first + j]; if (kpos) kpos[o + j] = first + j; if (lval) lval_out[o + j] = lv;
        }
        unsigned long long big = __ballot(c > 8u);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1ull;
            const uint32_t cc = __shfl(c, src, 64), ff = __shfl(first, src, 64), ll = __shfl(left, src, 64), vv = __shfl(lv, src, 64);
            const int64_t oo = __shfl(o, src, 64);
            for (uint32_t j = lane; j < cc; j += 64) {
                lrow[oo + j] = ll; if (rrow) rrow[oo + j] = rperm[ff + j]; if (kpos) kpos[oo + j] = ff + j; if (lval) lval_out[oo + j] = vv;
            }
        }
    }
}


// ---- semi-join pre-filter ------------------------------------------------------------
// When the probe side is much larger than the build side, most probe rows usually have no partner.  A bitmap
// of hashed build keys (16 bits per build key, two bits set per key, false positives ~2 %) is tested for every probe row in TABLE
// order; the survivors' (key, row id) pairs are compacted, order kept, and only they are sorted and merged.
// False positives simply count zero partners, so the result is the same row for row.
__device__ __forceinline__ uint32_t jmix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t key_hash(uint32_t k) { return jmix32(k); }
__device__ __forceinline__ uint32_t key_hash(uint64_t k) { return jmix32((uint32_t)k ^ jmix32((uint32_t)(k >> 32))); }

// Two bits per key inside ONE 32-bit word (a blocked Bloom filter: one memory access per probe); with 16 bits of
// bitmap per build key about 2 % of the non-matching probe rows pass.
__device__ __forceinline__ uint32_t two_bits(uint32_t h)
{
    const uint32_t g = jmix32(h ^ 0x9E3779B9u);
    return (1u << (g & 31u)) | (1u << ((g >> 5) & 31u));
}

template <typename K>
__global__ __launch_bounds__(256) void bitmap_build_kernel(const K *__restrict__ keys, int64_t s, uint32_t *__restrict__ bitmap, uint32_t bitmask)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s; i += stride) {
        const uint32_t h = key_hash(keys[i]);
        atomicOr(&bitmap[(h & bitmask) >> 5], two_bits(h));
    }
}

constexpr int kSemiThreads = 256, kSemiTile = kSemiThreads * 16;       // a thread owns 4 groups of 4 consecutive rows
template <typename K>
__device__ __forceinline__ uint32_t semi_mask(const K *__restrict__ keys, int64_t n, int64_t tile, const uint32_t *__restrict__ bitmap, uint32_t bitmask)
{
    uint32_t mask = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int64_t r = tile * kSemiTile + ((int64_t)g * kSemiThreads + threadIdx.x) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (r + j < n) {
                const uint32_t h = key_hash(keys[r + j]), bits = two_bits(h);
                mask |= (uint32_t)((bitmap[(h & bitmask) >> 5] & bits) == bits) << (g * 4 + j);
            }
        }
    }
    return mask;
}

template <typename K>
__global__ __launch_bounds__(kSemiThreads) void semi_count_kernel(const K *__restrict__ keys, int64_t n, const uint32_t *__restrict__ bitmap, uint32_t bitmask,
                                                                  uint16_t *__restrict__ masks, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    const uint32_t mask = semi_mask<K>(keys, n, tile, bitmap, bitmask);
    masks[tile * kSemiThreads + threadIdx.x] = (uint16_t)mask;
    uint32_t cnt = __popc(mask);
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) counts[tile] = s_cnt;
}

template <typename K>
__global__ __launch_bounds__(kSemiThreads) void semi_scatter_kernel(const K *__restrict__ keys, int64_t n, const uint16_t *__restrict__ masks,
                                                                    const int64_t *__restrict__ offsets, K *__restrict__ out_keys, uint32_t *__restrict__ out_rows)
{
    __shared__ uint32_t s_wcnt[4][4];                 // [group][wave] survivors
    const int64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t mask = masks[tile * kSemiThreads + threadIdx.x];
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
    int64_t run = offsets[tile];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t x = s_wcnt[g][w]; if (w < wave) before += x; total += x; }
        const uint32_t m4 = (mask >> (g * 4)) & 15u;
        if (m4) {
            int64_t q = run + before + lane_excl[g];
            const int64_t r = tile * kSemiTile + ((int64_t)g * kSemiThreads + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) if (m4 & (1u << j)) { out_keys[q] = keys[r + j]; out_rows[q] = (uint32_t)(r + j); q++; }
        }
        run += total;
    }
}

__global__ __launch_bounds__(256) void gather_u32_via_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

// i64 keys gathered into sorted order, biased so that unsigned comparison is signed order
__global__ __launch_bounds__(256) void gather_biased_i64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm,
                                                                uint64_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[perm[i]] ^ 0x8000000000000000ull;
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

bool is_u32ish(int d) { return d == HARK_I32 || d == HARK_U32; }

} // namespace

extern "C" {

int hark_entry_join(hark_context *ctx, hark_result **out, const hark_table *db1, const hark_table *db2,
                    int32_t col1, int32_t col2, const int32_t *cols1, int64_t l, const int32_t *cols2, int64_t k)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db1 || !db2) return HARK_EARG;
    *out = nullptr;
    ctx->last_join_path = HARK_PATH_NONE;
    ctx->last_join_weighted = false;
    ctx->last_join_clustered = false;
    ctx->last_join_rotated = false;
    if (l < 0 || k < 0 || (l && !cols1) || (k && !cols2)) return hark_fail(ctx, HARK_EARG, "join: bad column lists");
    const int64_t n = db1->n, s = db2->n;
    // db1[:, col1] / db2[:, col2] (join.fut:55-56) are checked even for empty outputs when the side has rows
    if ((n > 0 && (col1 < 0 || col1 >= db1->m)) || (s > 0 && (col2 < 0 || col2 >= db2->m)))
        return hark_fail(ctx, HARK_EBOUNDS, "join: key column out of bounds");
    // join.fut:52 types both keys u32; i64 keys (BASELINE configs[3]) are an extension: signed key order
    const bool k64 = n > 0 && s > 0 && db1->cols[col1].dtype == HARK_I64 && db2->cols[col2].dtype == HARK_I64;
    if (!k64 && ((n > 0 && !is_u32ish(db1->cols[col1].dtype)) || (s > 0 && !is_u32ish(db2->cols[col2].dtype))))
        return hark_fail(ctx, HARK_EUNSUPPORTED, "join: key columns must both be 32-bit integers (join.fut:52 is u32) or both i64");
    if (n + s > 0xFFFFFFFFll) return hark_fail(ctx, HARK_EARG, "join: at most 2^32-1 rows in total");
    hark_result *res = new hark_result();
    res->n = 0; res->cols.resize((size_t)(l + k));
    for (auto &c : res->cols) { c.dtype = HARK_U32; c.data = nullptr; c.owned = false; }
    if (n == 0 || s == 0) { *out = res; return HARK_OK; }

    uint32_t *lperm = nullptr, *lkeys = nullptr, *rperm = nullptr, *rkeys = nullptr;
    uint32_t *lb = nullptr, *cnt = nullptr, *lrow = nullptr, *rrow = nullptr, *kpos = nullptr;
    int64_t *offs = nullptr;
    int64_t P = 0;
    hipStream_t st = ctx->stream;
    // keys as u32 whatever the declared signedness (join.fut:52 types both tables u32)
    uint64_t *lk64 = nullptr, *rk64 = nullptr;
    int rc = HARK_OK;
    int64_t nl = n;                                   // left rows that reach the sort + merge
    const void *lcol = db1->cols[col1].data, *rcol = db2->cols[col2].data;
    bool filtered = false, partitioned = false, unique = false;   // unique: partitioned path, all build keys distinct (one partner per survivor)
    // ONE non-key 4-byte output column of the BUILD side is brought into rank order up front and read off by the order
    // kernel for every output row (unique build keys), instead of P reads through the rank afterwards.  With i64 keys it
    // travels through the sort beside key and row id (no gather); with 32-bit keys it is gathered through the permutation.
    int rank_col = -1;
    for (int64_t j = 0; j < k && rank_col < 0; j++)
        if (cols2[j] >= 0 && cols2[j] < db2->m && cols2[j] != col2 && hark_dtype_size(db2->cols[cols2[j]].dtype) == 4 && !getenv("HARK_JOIN_NO_RANK_GATHER")) rank_col = cols2[j];
    if (!(n >= ((int64_t)1 << 18) && s >= 4096)) rank_col = -1;             // (the partitioned path's own thresholds)
    uint32_t *rranked = nullptr, *rval = nullptr;
    int build_unique = -1;                                     // 1: the sort saw that all build keys are distinct (no run lengths needed)
    // ---- the build side is sorted first: both paths need it (the probe keys' sample runs under it on the second stream)
    if (k_join_hot_prepare(ctx, lcol, k64, n, rcol, s) != HARK_OK) ctx->err.clear();       // (without it the sample runs in line)
    if (k64) rc = k_argsort_i64_keys(ctx, rcol, s, &rperm, &rk64,           // permutation + the sorted (biased) keys (+ the column) in one go
                                     rank_col >= 0 ? static_cast<const uint32_t *>(db2->cols[rank_col].data) : nullptr, rank_col >= 0 ? &rranked : nullptr, &build_unique);
    else {
        // 32-bit keys: when the rank-ordered column is the only thing the build side contributes besides its key, the sort
        // carries that column as its payload instead of the row ids (no permutation, no gather through it: 0.17 ms of a
        // 1e8 x 1e7 join); the sort-merge fallback below asks for the permutation if it turns out to be needed
        bool perm_free = rank_col >= 0 && !getenv("HARK_JOIN_PERM");
        for (int64_t j = 0; j < k; j++) perm_free = perm_free && (cols2[j] == rank_col || cols2[j] == col2);
        if (perm_free) rc = k_sort_column(ctx, rcol, HARK_U32, s, false, static_cast<const uint32_t *>(db2->cols[rank_col].data), &rranked, &rkeys);
        else rc = k_argsort_column(ctx, rcol, HARK_U32, s, false, &rperm, &rkeys);
    }
    // ---- partitioned path (k_hjoin.hip): matching probe rows as (rank in the sorted build side, left row), sorted
    // A non-key 4-byte output column of the PROBE side can travel with the probe rows through the partitioned path (in the
    // pad of its 16-byte i64 entries) instead of being gathered at the end: 6.25e7 random 4-byte reads over a 500-MB
    // column cost 1.2 ms of BASELINE configs[3]'s share, carrying it ~0.25 ms.
    int carry_col = -1;
    if (k64) for (int64_t j = 0; j < l && carry_col < 0; j++)
        if (cols1[j] >= 0 && cols1[j] < db1->m && cols1[j] != col1 && hark_dtype_size(db1->cols[cols1[j]].dtype) == 4 && !getenv("HARK_JOIN_NOCARRY")) carry_col = cols1[j];
    uint32_t *sval = nullptr, *lval_exp = nullptr;    // the carried column per matching probe row / per output pair
    if (!rc && rank_col >= 0 && !rranked) {
        rc = hark_alloc(ctx, (void **)&rranked, (size_t)s * 4);
        if (!rc) rc = k_gather(ctx, db2->cols[rank_col].data, 4, rperm, rranked, s);
    }
    // does any result column need the pairs' row ids or ranks?  Not when the carried probe-side column and the rank-ordered
    // build-side column are all that is selected (BASELINE configs[3]: the two row-id columns) -- with unique build keys the
    // order kernel then writes those two columns and nothing else
    bool rows_needed = false;
    for (int64_t j = 0; j < l; j++) rows_needed = rows_needed || cols1[j] != carry_col;
    for (int64_t j = 0; j < k; j++) rows_needed = rows_needed || cols2[j] != rank_col;
    if (!rc) {
        uint32_t *prank = nullptr, *plrow = nullptr, *pcnt = nullptr;
        int64_t M = 0;
        rc = k_join_partitioned(ctx, lcol, k64, n, k64 ? static_cast<const void *>(rk64) : static_cast<const void *>(rkeys), s,
                                carry_col >= 0 ? static_cast<const uint32_t *>(db1->cols[carry_col].data) : nullptr, rranked,
                                &prank, &plrow, &pcnt, &sval, &rval, &M, &partitioned, &unique, rows_needed, build_unique);
        ctx->last_join_path = !partitioned ? HARK_PATH_JOIN_SORTMERGE : ctx->last_join_clustered ? HARK_PATH_JOIN_CLUSTERED : ctx->last_join_rotated ? HARK_PATH_JOIN_PARTITIONED_ROTATED : ctx->last_join_weighted ? HARK_PATH_JOIN_PARTITIONED_WEIGHTED : HARK_PATH_JOIN_PARTITIONED;
        if (!rc && partitioned) {
            nl = M;
            lb = prank; lperm = plrow; cnt = pcnt;                 // freed with the other scratch below
            if (unique) P = M;                                     // no counts, no scan, no expansion: survivor i IS output row i
            else if (M > 0) {
                rc = hark_alloc(ctx, (void **)&offs, (size_t)M * 8);
                if (!rc) rc = k_exclusive_scan_u32(ctx, cnt, M, nullptr, offs, &P);
            }
        }
    }
    if (!rc && !partitioned && !rperm) {                       // the sort-merge path gathers through the permutation after all
        hark_free(ctx, rkeys); rkeys = nullptr;
        if (k64) { hark_free(ctx, rk64); rk64 = nullptr; rc = k_argsort_i64_keys(ctx, rcol, s, &rperm, &rk64, nullptr, nullptr, nullptr); }
        else rc = k_argsort_column(ctx, rcol, HARK_U32, s, false, &rperm, &rkeys);
    }
    if (!rc && !partitioned && n >= ((int64_t)1 << 20) && n >= 4 * s) {
        // semi-join pre-filter (see the kernels above): bitmap of 16 bits per build key, power of two, <= 2^31 bits
        uint32_t lgb = 16;
        while (lgb < 31 && ((int64_t)1 << lgb) < 16 * s) lgb++;
        const uint32_t bitmask = (uint32_t)(((uint64_t)1 << lgb) - 1u);
        const int64_t ntiles = (n + kSemiTile - 1) / kSemiTile;
        uint32_t *bitmap = nullptr, *tcnt = nullptr; uint16_t *masks = nullptr; int64_t *toffs = nullptr;
        void *ckeys = nullptr; uint32_t *crows = nullptr;
        rc = hark_alloc(ctx, (void **)&bitmap, ((size_t)1 << lgb) / 8);
        if (!rc) rc = hark_alloc(ctx, (void **)&tcnt, (size_t)ntiles * 4);
        if (!rc) rc = hark_alloc(ctx, (void **)&toffs, (size_t)ntiles * 8);
        if (!rc) rc = hark_alloc(ctx, (void **)&masks, (size_t)ntiles * kSemiThreads * 2);
        int64_t n1 = 0;
        if (!rc) {
            HIP_TRY_RC(ctx, rc, hipMemsetAsync(bitmap, 0, ((size_t)1 << lgb) / 8, st));
            if (k64) {
                HARK_LAUNCH_RC(ctx, rc, bitmap_build_kernel<uint64_t><<<grid_for(ctx, s), 256, 0, st>>>(static_cast<const uint64_t *>(rcol), s, bitmap, bitmask));
                HARK_LAUNCH_RC(ctx, rc, semi_count_kernel<uint64_t><<<dim3((unsigned)ntiles), kSemiThreads, 0, st>>>(static_cast<const uint64_t *>(lcol), n, bitmap, bitmask, masks, tcnt));
            } else {
                HARK_LAUNCH_RC(ctx, rc, bitmap_build_kernel<uint32_t><<<grid_for(ctx, s), 256, 0, st>>>(static_cast<const uint32_t *>(rcol), s, bitmap, bitmask));
                HARK_LAUNCH_RC(ctx, rc, semi_count_kernel<uint32_t><<<dim3((unsigned)ntiles), kSemiThreads, 0, st>>>(static_cast<const uint32_t *>(lcol), n, bitmap, bitmask, masks, tcnt));
            }
            if (!rc) rc = k_exclusive_scan_u32(ctx, tcnt, ntiles, nullptr, toffs, &n1);
        }
        if (!rc && n1 <= n / 2) {                      // otherwise the filter does not pay: sort the whole side
            filtered = true; nl = n1;
            if (n1 > 0) {
                rc = hark_alloc(ctx, &ckeys, (size_t)n1 * (k64 ? 8 : 4));
                if (!rc) rc = hark_alloc(ctx, (void **)&crows, (size_t)n1 * 4);
                if (!rc) {
                    if (k64) HARK_LAUNCH_RC(ctx, rc, semi_scatter_kernel<uint64_t><<<dim3((unsigned)ntiles), kSemiThreads, 0, st>>>(static_cast<const uint64_t *>(lcol), n, masks, toffs, static_cast<uint64_t *>(ckeys), crows));
                    else HARK_LAUNCH_RC(ctx, rc, semi_scatter_kernel<uint32_t><<<dim3((unsigned)ntiles), kSemiThreads, 0, st>>>(static_cast<const uint32_t *>(lcol), n, masks, toffs, static_cast<uint32_t *>(ckeys), crows));
                    if (rc) {}
                    else if (!k64) rc = k_sort_column(ctx, ckeys, HARK_U32, n1, false, crows, &lperm, &lkeys);   // the row ids travel with the keys
                    else {
                        uint32_t *pos = nullptr;                                                          // positions in the compacted arrays
                        rc = k_argsort_column(ctx, ckeys, HARK_I64, n1, false, &pos, nullptr);
                        if (!rc) rc = hark_alloc(ctx, (void **)&lperm, (size_t)n1 * 4);
                        if (!rc) rc = hark_alloc(ctx, (void **)&lk64, (size_t)n1 * 8);
                        if (!rc) {
                            HARK_LAUNCH_RC(ctx, rc, gather_u32_via_kernel<<<grid_for(ctx, n1), 256, 0, st>>>(crows, pos, lperm, n1));
                            HARK_LAUNCH_RC(ctx, rc, gather_biased_i64_kernel<<<grid_for(ctx, n1), 256, 0, st>>>(static_cast<const uint64_t *>(ckeys), pos, lk64, n1));
                            if (hipStreamSynchronize(st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: filter kernels failed");
                        }
                        hark_free(ctx, pos);
                    }
                }
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = hark_fail(ctx, HARK_EHIP, "join: filter kernels failed");
        hark_free(ctx, bitmap); hark_free(ctx, tcnt); hark_free(ctx, toffs); hark_free(ctx, masks); hark_free(ctx, ckeys); hark_free(ctx, crows);
    }
    if (!rc && !partitioned && !filtered) rc = k_argsort_column(ctx, lcol, k64 ? HARK_I64 : HARK_U32, n, false, &lperm, k64 ? nullptr : &lkeys);
    if (!rc && !partitioned && k64 && !filtered) {
        rc = hark_alloc(ctx, (void **)&lk64, (size_t)n * 8);
        if (!rc) HARK_LAUNCH_RC(ctx, rc, gather_biased_i64_kernel<<<grid_for(ctx, n), 256, 0, st>>>(static_cast<const uint64_t *>(lcol), lperm, lk64, n));
    }
    if (!rc && !partitioned && nl > 0) rc = hark_alloc(ctx, (void **)&lb, (size_t)nl * 4);
    if (!rc && !partitioned && nl > 0) rc = hark_alloc(ctx, (void **)&cnt, (size_t)nl * 4);
    if (!rc && !partitioned && nl > 0) rc = hark_alloc(ctx, (void **)&offs, (size_t)nl * 8);
    if (!rc && !partitioned && nl > 0) {
        if (k64) HARK_LAUNCH_RC(ctx, rc, join_count_kernel<uint64_t><<<grid_for(ctx, (nl + kJoinChunk - 1) / kJoinChunk), 256, 0, st>>>(lk64, nl, rk64, s, lb, cnt));
        else HARK_LAUNCH_RC(ctx, rc, join_count_kernel<uint32_t><<<grid_for(ctx, (nl + kJoinChunk - 1) / kJoinChunk), 256, 0, st>>>(lkeys, nl, rkeys, s, lb, cnt));
        if (!rc) rc = k_exclusive_scan_u32(ctx, cnt, nl, nullptr, offs, &P);
    }
    if (!rc && P > 0) {
        // select cols1 db1[r1,:] / select cols2 db2[r2,:] (join.fut:69-70) run only when there are pairs
        for (int64_t j = 0; j < l && !rc; j++)
            if (cols1[j] < 0 || cols1[j] >= db1->m) rc = hark_fail(ctx, HARK_EBOUNDS, "join: cols1[%lld] = %d out of bounds", (long long)j, cols1[j]);

        for (int64_t j = 0; j < k && !rc; j++)
            if (cols2[j] < 0 || cols2[j] >= db2->m) rc = hark_fail(ctx, HARK_EBOUNDS, "join: cols2[%lld] = %d out of bounds", (long long)j, cols2[j]);

        // Output positions ascend with the pair's RANK in the sorted build side, so anything that is a function of the rank is
        // read (almost) sequentially: the join key off the sorted build keys, and -- on the partitioned path -- every build
        // side column off a copy brought into rank order once (s random reads instead of P; the right row ids themselves are
        // then not needed).
        const bool by_rank = partitioned && !getenv("HARK_JOIN_NO_RANK_GATHER");
        if (!rc && !unique) rc = hark_alloc(ctx, (void **)&lrow, (size_t)P * 4);
        if (!rc && !by_rank) rc = hark_alloc(ctx, (void **)&rrow, (size_t)P * 4);
        bool key_out = false;
        for (int64_t j = 0; j < l + k && !rc; j++) key_out = key_out || (j < l ? cols1[j] == col1 : cols2[j - l] == col2);
        if (!rc && (key_out || by_rank) && !unique) rc = hark_alloc(ctx, (void **)&kpos, (size_t)P * 4);
        if (!rc && sval && !unique) rc = hark_alloc(ctx, (void **)&lval_exp, (size_t)P * 4);
        if (!rc && unique) {                               // (left row, rank) pairs are the output rows: the right row id is one gather away
            lrow = lperm; lperm = nullptr;
            kpos = lb; lb = nullptr;
            lval_exp = sval; sval = nullptr;
            if (!by_rank) rc = k_gather(ctx, rperm, 4, kpos, rrow, P);
        } else if (!rc) {
            HARK_LAUNCH_RC(ctx, rc, join_expand_kernel<<<grid_for(ctx, nl), 256, 0, st>>>(offs, cnt, nl, lb, lperm, rperm, lrow, rrow, kpos, sval, lval_exp));
        }
        res->n = P;
        int64_t carried_at = -1, ranked_at = -1;
        for (int64_t j = 0; j < l + k && !rc; j++) {
            const hark_table *t = j < l ? db1 : db2;
            const int c = j < l ? cols1[j] : cols2[j - l];
            res->cols[j].dtype = t->cols[c].dtype; res->cols[j].owned = true;
            const int esz = (int)hark_dtype_size(t->cols[c].dtype);
            if (j < l && c == carry_col && lval_exp) {                               // the carried column arrived with the rows: it IS the result column
                res->cols[j].data = lval_exp; lval_exp = nullptr; carried_at = j;
                continue;
            }
            if (j >= l && c == rank_col && rval) {                                   // read off in rank order by the order kernel
                res->cols[j].data = rval; rval = nullptr; ranked_at = j;
                continue;
            }
            rc = hark_alloc(ctx, &res->cols[j].data, (size_t)P * esz);
            const bool is_key = j < l ? c == col1 : c == col2;
            if (!rc && is_key && k64) HARK_LAUNCH_RC(ctx, rc, gather_biased_i64_kernel<<<grid_for(ctx, P), 256, 0, st>>>(rk64, kpos, static_cast<uint64_t *>(res->cols[j].data), P));   // XOR undoes the bias
            else if (!rc && is_key) rc = k_gather(ctx, rkeys, 4, kpos, res->cols[j].data, P);
            else if (!rc && j < l && c == carry_col && carried_at >= 0) {            // selected twice: a copy of the first
                if (hipMemcpyAsync(res->cols[j].data, res->cols[carried_at].data, (size_t)P * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: copy failed");
            } else if (!rc && j >= l && c == rank_col && ranked_at >= 0) {           // selected twice: a copy of the first
                if (hipMemcpyAsync(res->cols[j].data, res->cols[ranked_at].data, (size_t)P * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: copy failed");
            } else if (!rc && j >= l && by_rank) {
                void *ranked = c == rank_col ? rranked : nullptr;                    // the column in rank order: ranked[r] = column[rperm[r]]
                if (!ranked) {
                    rc = hark_alloc(ctx, &ranked, (size_t)s * esz);
                    if (!rc) rc = k_gather(ctx, t->cols[c].data, esz, rperm, ranked, s);
                }
                if (!rc) rc = k_gather(ctx, ranked, esz, kpos, res->cols[j].data, P);
                if (ranked != rranked) hark_free(ctx, ranked);                       // stream-ordered reuse
            }
            else if (!rc) rc = k_gather(ctx, t->cols[c].data, esz, j < l ? lrow : rrow, res->cols[j].data, P);
        }
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: kernels failed");
    }
    k_join_hot_release(ctx);
    hark_free(ctx, lk64); hark_free(ctx, rk64);
    hark_free(ctx, lperm); hark_free(ctx, lkeys); hark_free(ctx, rperm); hark_free(ctx, rkeys); hark_free(ctx, lb); hark_free(ctx, cnt); hark_free(ctx, offs); hark_free(ctx, lrow); hark_free(ctx, rrow); hark_free(ctx, kpos);
    hark_free(ctx, sval); hark_free(ctx, lval_exp); hark_free(ctx, rranked); hark_free(ctx, rval);
    if (rc) { result_release(ctx, res); return rc; }
    *out = res;
    return HARK_OK;
}

namespace {
__global__ __launch_bounds__(256) void unbias_i64_kernel(uint64_t *__restrict__ keys, int64_t n, uint64_t xorm)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) keys[i] ^= xorm;
}
} // namespace

int hark_entry_sort(hark_context *ctx, hark_result **out, const hark_table *db, int32_t key_col, int32_t descending,
                    const int32_t *cols, int64_t k)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out || !db) return HARK_EARG;
    *out = nullptr;
    if (k < 0 || (k && !cols)) return hark_fail(ctx, HARK_EARG, "sort: bad column list");
    if (key_col < 0 || key_col >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "sort: key column %d out of bounds", key_col);
    for (int64_t j = 0; j < k; j++)
        if (cols[j] < 0 || cols[j] >= db->m) return hark_fail(ctx, HARK_EBOUNDS, "sort: column %d out of bounds", cols[j]);
    hark_result *res = new hark_result();
    res->n = db->n; res->cols.resize((size_t)k);
    for (int64_t j = 0; j < k; j++) { res->cols[j].dtype = db->cols[cols[j]].dtype; res->cols[j].data = nullptr; res->cols[j].owned = db->n > 0; }
    if (db->n == 0) { *out = res; return HARK_OK; }
    // Output columns that ARE the (u32 / i32) key come straight from the sorted sort words; if exactly one other
    // 4-byte column is asked for it travels with the keys as the payload (no row ids, no gather); otherwise the
    // payload is the row id and every other column is gathered through it.
    const int kdt = db->cols[key_col].dtype;
    // Ascending i64 keys: the sort hands back the SORTED KEYS themselves (and one 4-byte column that travelled with them when
    // the high words differ: k_argsort_i64_keys) -- the key column and that column are then not gathered through the row ids
    // (two random reads per row: most of an ORDER BY on an i64 key).
    for (int attempt = 0; attempt < 1 && kdt == HARK_I64 && db->n >= 4096; attempt++) {      // (a block to leave: descending keys may decline)
        int carry64 = -1, others64 = 0;
        for (int64_t j = 0; j < k; j++) {
            if (cols[j] == key_col) continue;
            bool seen = false;
            for (int64_t q = 0; q < j; q++) seen = seen || cols[q] == cols[j];
            if (!seen) { others64++; carry64 = cols[j]; }
        }
        const bool carried64 = others64 == 1 && hark_dtype_size(db->cols[carry64].dtype) == 4;
        uint32_t *perm = nullptr, *val = nullptr;
        uint64_t *keys64 = nullptr;
        int rc;
        bool plain = false;
        if (descending) {                                   // the tuple passes on the complemented keys, or nothing (the general path below)
            bool done = false;
            rc = k_argsort_i64_desc_tuples(ctx, db->cols[key_col].data, db->n, &perm, &keys64,
                                           carried64 ? static_cast<const uint32_t *>(db->cols[carry64].data) : nullptr, carried64 ? &val : nullptr, &done, 0x7FFFFFFFFFFFFFFFull, &db->cols[key_col].msd_unfit);
            if (!rc && !done) break;
            plain = true;                                   // (the tuple path's last kernel wrote the plain keys)
        } else rc = k_argsort_i64_keys(ctx, db->cols[key_col].data, db->n, &perm, &keys64,
                                       carried64 ? static_cast<const uint32_t *>(db->cols[carry64].data) : nullptr, carried64 ? &val : nullptr, nullptr, &plain, &db->cols[key_col].msd_unfit);
        if (!rc && !plain) {                                // the permutation paths hand back biased keys
            HARK_LAUNCH_RC(ctx, rc, unbias_i64_kernel<<<grid_for(ctx, db->n), 256, 0, ctx->stream>>>(keys64, db->n, descending ? 0x7FFFFFFFFFFFFFFFull : 0x8000000000000000ull));
        }
        bool keys_taken = false, val_taken = false;
        for (int64_t j = 0; j < k && !rc; j++) {
            const int esz = (int)hark_dtype_size(res->cols[j].dtype);
            if (cols[j] == key_col && !keys_taken) { res->cols[j].data = keys64; keys_taken = true; continue; }
            if (carried64 && cols[j] == carry64 && val && !val_taken) { res->cols[j].data = val; val_taken = true; continue; }
            rc = hark_alloc(ctx, &res->cols[j].data, (size_t)db->n * esz);
            if (rc) break;
            hipError_t he = hipSuccess;
            if (cols[j] == key_col) he = hipMemcpyAsync(res->cols[j].data, keys64, (size_t)db->n * 8, hipMemcpyDeviceToDevice, ctx->stream);
            else if (carried64 && cols[j] == carry64 && val) he = hipMemcpyAsync(res->cols[j].data, val, (size_t)db->n * 4, hipMemcpyDeviceToDevice, ctx->stream);
            else rc = k_gather(ctx, db->cols[cols[j]].data, esz, perm, res->cols[j].data, db->n);
            if (he != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: copy failed");
        }
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: kernels failed");
        hark_free(ctx, perm);
        if (!keys_taken) hark_free(ctx, keys64);
        if (!val_taken) hark_free(ctx, val);
        if (rc) { result_release(ctx, res); return rc; }
        *out = res;
        return HARK_OK;
    }
    const bool key_from_words = kdt == HARK_U32 || kdt == HARK_I32;
    int carry = -1, others = 0;
    for (int64_t j = 0; j < k; j++) {
        if (cols[j] == key_col && key_from_words) continue;
        bool seen = false;
        for (int64_t q = 0; q < j; q++) seen = seen || cols[q] == cols[j];
        if (!seen) { others++; carry = cols[j]; }
    }
    const bool carried = others == 1 && kdt != HARK_I64 && hark_dtype_size(db->cols[carry].dtype) == 4;
    uint32_t *perm = nullptr, *words = nullptr;
    int rc = k_sort_column(ctx, db->cols[key_col].data, kdt, db->n, descending != 0,
                           carried ? static_cast<const uint32_t *>(db->cols[carry].data) : nullptr, &perm, key_from_words ? &words : nullptr);
    for (int64_t j = 0; j < k && !rc; j++) {
        const int esz = (int)hark_dtype_size(res->cols[j].dtype);
        if (carried && cols[j] == carry && perm) {                       // the sorted payload is the column (first use takes the buffer)
            bool first_use = true;
            for (int64_t q = 0; q < j; q++) first_use = first_use && cols[q] != carry;
            if (first_use) { res->cols[j].data = perm; continue; }
        }
        if (cols[j] == key_col && key_from_words && words) {             // the sorted keys are the column (first use takes the buffer)
            bool first_use = true;
            for (int64_t q = 0; q < j; q++) first_use = first_use && cols[q] != key_col;
            if (first_use) { res->cols[j].data = words; continue; }
        }
        rc = hark_alloc(ctx, &res->cols[j].data, (size_t)db->n * esz);
        if (rc) break;
        if (cols[j] == key_col && key_from_words) rc = hipMemcpyAsync(res->cols[j].data, words, (size_t)db->n * 4, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess
                                                          ? HARK_OK : hark_fail(ctx, HARK_EHIP, "sort: copy failed");
        else if (carried && cols[j] == carry) rc = hipMemcpyAsync(res->cols[j].data, perm, (size_t)db->n * 4, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess
                                                       ? HARK_OK : hark_fail(ctx, HARK_EHIP, "sort: copy failed");
        else rc = k_gather(ctx, db->cols[cols[j]].data, esz, perm, res->cols[j].data, db->n);
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: kernels failed");
    bool perm_taken = false;
    for (auto &c : res->cols) perm_taken = perm_taken || c.data == perm;
    if (!perm_taken) hark_free(ctx, perm);
    bool words_taken = false;
    for (auto &c : res->cols) words_taken = words_taken || (words && c.data == words);
    if (!words_taken) hark_free(ctx, words);
    if (rc) { result_release(ctx, res); return rc; }
    *out = res;
    return HARK_OK;
}

} // extern "C"
