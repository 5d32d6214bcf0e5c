// Internal declarations shared by the libhark.so translation units.
// gfx950 (MI355X) only: wave = 64 lanes, 256 CUs in 8 XCDs, 160 KiB LDS/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include <unordered_map>

#include "../../include/hark.h"

#define HARK_WAVE 64
#define HARK_NUM_CU 256

struct hark_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // the stream entries launch on
    // the join's early sample (k_hjoin.hip, k_join_hot_prepare): a second stream, its events, and the block prepared for the next partitioned join
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_event = nullptr, main_event = nullptr;
    void *join_prep = nullptr; const void *join_prep_col = nullptr; int64_t join_prep_n = 0, join_prep_s = 0;
    bool join_prep_tested = false;  // ... and the probe column's clustering test (k_cjoin.hip) was enqueued behind its sample
    std::string err;
    int32_t *d_err = nullptr;       // device-side sticky error word (bounds failures)
    int32_t *h_pin = nullptr;       // pinned scratch for small D2H reads (64 KiB: status words, LIMIT prefixes)
    int num_cu = HARK_NUM_CU;
    // pinned double buffer for device -> pageable-host copies (hark_d2h)
    char *bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    // caching allocator (hark_alloc / hark_free)
    std::multimap<size_t, void *> pool_free;
    std::unordered_map<void *, size_t> pool_live;
    // freed blocks kept for reuse up to pool_limit (HARK_POOL_LIMIT_MB; hark_context_trim gives them all back):
    // 64 GiB of 288: re-allocating multi-GB workspaces per query costs tens of ms (measured: C5 150 ms vs 2 ms with a 16 GiB limit)
    size_t pool_cached = 0, pool_limit = (size_t)64 << 30;
    // pinned HOST blocks handed to callers as the storage of downloaded results (hark_host_alloc / hark_host_free): pinning a
    // fresh 16-MiB buffer costs more than the copy into it, so freed blocks are kept (up to pin_limit) and reused
    std::multimap<size_t, void *> pin_free;
    std::unordered_map<void *, size_t> pin_live;
    size_t pin_cached = 0, pin_limit = (size_t)4 << 30;
    // diagnostic: which GROUP BY path served the last group-by entry (hark_context_last_groupby_path)
    int last_groupby_path = 0;
    int last_groupby_passes = 0;             // row passes of the last dense-path filter_groupby (hark_context_last_groupby_passes)
    int last_groupby_window = 0;             // ... the last dense GROUP BY: 1 took the window path (a key column sorted / clustered by the key), 2 the partition with rotated loads (hark_context_last_groupby_window)
    int last_join_path = 0;        // hark_context_last_join_path
    bool last_join_weighted = false;   // ... the partitioned path cut its buckets by the sampled probe rows' weight (k_hjoin.hip)
    bool last_join_overflow = false;   // ... the partitioned path gave up on a full slab (the caller falls back)
    bool join_rotate = false, last_join_rotated = false;   // ... the clustering test's third verdict for the join being run / the partition read with rotated loads
    bool last_join_clustered = false;  // ... the probe column was clustered by key: searched in row order (k_cjoin.hip)
};

// Every entry runs on the context's device whatever device the calling thread has current (a process may hold
// contexts on several GPUs, and torch changes the current device under us); restores the caller's device.
struct hark_device_guard {
    int prev = -1;
    explicit hark_device_guard(const hark_context *ctx)
    {
        int cur = -1;
        if (ctx && hipGetDevice(&cur) == hipSuccess && cur != ctx->device) { prev = cur; (void)hipSetDevice(ctx->device); }
    }
    ~hark_device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
    hark_device_guard(const hark_device_guard &) = delete;
    hark_device_guard &operator=(const hark_device_guard &) = delete;
};

struct hark_column {
    void *data = nullptr;
    int32_t dtype = HARK_I32;
    bool owned = true;
    // Column statistics, filled on first use and kept for the life of the table.  CONTRACT (include/hark.h,
    // hark_table_from_device): a table's columns are immutable while the table exists; a caller that rewrites a BORROWED
    // column (owned == false) calls hark_table_invalidate_stats, which clears everything below.  A stale range would
    // otherwise surface as a spurious HARK_EBOUNDS (dense paths check every key against it) or, under
    // hark_table_composite_key, alias two key tuples.
    // [min, max] of a 32-bit integer column:
    mutable bool has_range[2] = {false, false};            // [0] read as unsigned, [1] read as signed
    mutable int64_t range_min[2] = {0, 0}, range_max[2] = {0, 0};
    // ... and what the hash group-by learnt about it as a KEY column: 0 nothing yet, > 0 the table rounds its distinct keys
    // need, < 0 the LDS hash path does not fit this column -- more distinct keys than kMaxRoundsWorth rounds of tables hold,
    // or keys so skewed that a slab of the hash partition overflows; both are properties of (the key column, its row
    // count), not of the aggregate asked for, so the verdict is STICKY: the sort-based path is taken without another
    // attempt (a failed one costs a partition pass and a sample round: 1.1 ms per 1e8 rows) until the statistics are
    // invalidated.  Performance only: every path returns the same rows.
    mutable int32_t hash_rounds = 0;
    // ... and what the three-sweep sort of 64-bit keys (k_msort.hip) learnt: 1 = it gave up on this column (many copies of every key,
    // tight clusters), so the next ORDER BY / GROUP BY / JOIN on it starts with the tuple passes instead of losing the sweeps
    // again (1.7 ms per 1e8 rows).  Performance only.
    mutable int8_t msd_unfit = 0;
    // ... and what the fused GROUP BY's test (k_fgb.hip, fgb_cluster_test_kernel) found about it as a dense KEY column: -1 not
    // tested, 1 = rows an eighth of a batch apart are a few keys apart (a table kept in key order: the window path), 0 = they
    // scatter (the partition path).  One small launch and a synchronisation saved per statement.  Performance only.
    mutable int8_t key_clustered = -1;
    void invalidate_stats() const { has_range[0] = has_range[1] = false; hash_rounds = 0; msd_unfit = 0; key_clustered = -1; }
};

struct hark_table {
    int64_t n = 0, m = 0;
    std::vector<hark_column> cols;
};

struct hark_result {
    int64_t n = 0;
    std::vector<hark_column> cols;
    // the small-table paths (k_small.hip) deliver the row-major matrix of ALL columns in a pinned host block as well (a block of
    // hark_host_alloc): hark_result_matrix_pinned hands it over instead of launching anything; freed with the result otherwise
    void *host_matrix = nullptr;
    int64_t host_rows = 0, host_cols = 0;
};
static inline void hark_result_host_release(hark_context *ctx, hark_result *r) { if (r && r->host_matrix) { hark_host_free(ctx, r->host_matrix); r->host_matrix = nullptr; } }
int k_join_hot_prepare(hark_context *ctx, const void *lcol, bool k64, int64_t n, const void *rcol, int64_t s);
void k_join_hot_release(hark_context *ctx);
// k_cjoin.hip: the join of a probe column sorted / clustered by the key (same outputs as k_hjoin.hip's run_partitioned)
int k_cjoin_test(hark_context *ctx, hipStream_t st, const void *lcol, bool k64, int64_t n, const void *build, int64_t s, bool sorted_build);
int k_cjoin_verdict(hark_context *ctx);
int k_cjoin_run(hark_context *ctx, const void *lcol, bool k64, int64_t n, const void *rkeys, int64_t s, const uint32_t *runlen, const int32_t *flags,
                const uint32_t *lval, const uint32_t *rranked, uint32_t **rank_out, uint32_t **lrow_out, uint32_t **cnt_out, uint32_t **lval_out,
                uint32_t **rval_out, int64_t *m_out, bool *used, bool *dup_out, bool rows_needed, int64_t *general_out);
// k_small.hip: tables of a few rows in one launch and one synchronisation
bool k_small_fits(const hark_table *db, int64_t result_cols);
int k_small_query_sel(hark_context *ctx, const hark_table *db, const int32_t *cols, int64_t k, hark_result *res);
int k_small_query_groupby(hark_context *ctx, const hark_table *db, const int32_t *cols, const int32_t *ops, int64_t s, hark_result *res, int64_t *G_out);

static inline size_t hark_dtype_size(int dtype) { return dtype == HARK_I64 ? 8 : 4; }

int hark_fail(hark_context *ctx, int code, const char *fmt, ...);

#define HIP_TRY(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return hark_fail((ctx), HARK_EHIP, "%s failed: %s (%s:%d)", #call,          \
                             hipGetErrorString(e__), __FILE__, __LINE__);               \
    } while (0)

// ---- kernel launches --------------------------------------------------------------------------------------------------
// A launch that asks for more LDS or registers than a CU has, or for an impossible grid, is REFUSED when it is made; unchecked,
// it surfaces at the next synchronisation with a generic message (or as wrong results, when nothing synchronises before the
// output is read).  Every launch of the operator units goes through one of these two: the launch, then hipGetLastError(), and
// a failure names the kernel (the text of the launch up to its "<<<").  HARK_LAUNCH returns from the calling function;
// HARK_LAUNCH_RC launches only while `rc` is still HARK_OK and records the failure in it (functions that clean up at the end).
int hark_launch_failed(hark_context *ctx, hipError_t e, const char *launch_text, const char *file, int line);
#define HARK_LAUNCH(ctx, ...)                                                                                  \
    do {                                                                                                       \
        __VA_ARGS__;                                                                                           \
        const hipError_t le__ = hipGetLastError();                                                             \
        if (le__ != hipSuccess) return hark_launch_failed((ctx), le__, #__VA_ARGS__, __FILE__, __LINE__);      \
    } while (0)
#define HARK_LAUNCH_RC(ctx, rc, ...)                                                                           \
    do {                                                                                                       \
        if ((rc) == HARK_OK) {                                                                                 \
            __VA_ARGS__;                                                                                       \
            const hipError_t le__ = hipGetLastError();                                                         \
            if (le__ != hipSuccess) (rc) = hark_launch_failed((ctx), le__, #__VA_ARGS__, __FILE__, __LINE__);  \
        }                                                                                                      \
    } while (0)
// the same for the runtime calls of an rc-style function (hipMemsetAsync and friends)
#define HIP_TRY_RC(ctx, rc, call)                                                                              \
    do {                                                                                                       \
        if ((rc) == HARK_OK) {                                                                                 \
            const hipError_t e__ = (call);                                                                     \
            if (e__ != hipSuccess)                                                                             \
                (rc) = hark_fail((ctx), HARK_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
        }                                                                                                      \
    } while (0)

#define HARK_TRY(call)                 \
    do {                               \
        int rc__ = (call);             \
        if (rc__ != HARK_OK) return rc__; \
    } while (0)

// Device allocation that records failures in the context.
int hark_alloc(hark_context *ctx, void **out, size_t bytes);
void hark_free(hark_context *ctx, void *ptr);
// Reads `count` 8-byte words from the device after draining the stream.
int hark_read_words(hark_context *ctx, const void *dev, int64_t *host, int count);
// Device -> host copy of any size through pinned bounce buffers, synchronous on return.
int hark_d2h(hark_context *ctx, void *host, const void *dev, size_t bytes);

// ---- kernel launchers implemented in the k_*.hip units -------------------
// (all stream-ordered on ctx->stream; none allocates)

// k_fgb.hip
struct hark_fgb_plan {
    hark_context *ctx = nullptr;
    int64_t max_rows = 0, G = 0;
    int64_t algo = 0;          // 0 auto, 1 lds, 2 atomic, 3 partition
    int64_t chunk_rows = 0;    // partition path: rows per chunk (0 = auto)
    int64_t grid = 0;          // 0 = auto
    int64_t shift = 0;         // partition path: bucket = key >> shift
    int64_t P = 0;             // number of buckets
    int64_t nwg = 0;           // producer workgroups (= slabs per bucket)
    int64_t cap = 0;           // pairs per (bucket, workgroup) slab
    int64_t shift8 = 0, P8 = 0, cap8 = 0;   // second geometry over the same slab space: <= 128 buckets (one-word ring entries), 0 = none
    int64_t slack_pct = 0;     // slab capacity as % of the uniform share (0 = default)
    int64_t tile_rows = 0;
    int64_t period = 0;        // partition path: batches between ring sweeps (0 = the format's default)
    int64_t pairfmt = 0;       // partition pair format: 0 auto (compact), 1: 8-byte (key, value) pairs, 2: compact 6-byte units
    int64_t timing = 0;        // record HIP events around every kernel launch
    int64_t vop = 0;           // value operator: 0 f32 sum (f64 acc), 1..4 u32 sum/max/min/prod, 5 u32 -> u64 sum
    int64_t xform = 0;         // 0 none, 1 i32 -> ordered u32, 2 f32 -> ordered u32
    std::vector<hipEvent_t> ev; std::vector<int> ev_kind; size_t ev_used = 0;
    uint2 *pbuf = nullptr;     // [P][nwg][cap] (key, value-bits) pairs
    uint32_t *counts = nullptr;// [P][nwg] pairs in each slab
    double *acc_sum = nullptr; // [G]
    unsigned long long *acc_cnt = nullptr; // [G]
    unsigned long long *acc_min = nullptr, *acc_max = nullptr;   // [G] order words of the statistics pass (allocated on first use)
    int32_t *err = nullptr;    // device: [0] sticky error word, [1] the producers' batch counter, [2..3] pairs partitioned so far (u64, never reset)
    int64_t rows_fed = 0, rows_seen = 0, pairs_seen = 0;   // rows handed to the partition path / ... at the last error check / pairs at that check
    int64_t sel_pct = -1;      // share of the rows that survived the predicate between the last two checks (-1: not known): picks the geometry
    // the window path (k_fgb.hip, fgb_window_kernel: key columns sorted / clustered by the key)
    int64_t window = 0;        // knob: 0 by the test, 1 always, 2 never
    const void *win_k = nullptr; int64_t win_n = 0; int win_verdict = -1;   // the tested column and what the test said
    int64_t rot_rows = 0;      // rows the partition's producers read with rotated loads (verdict 2)
    int64_t win_rows = 0, win_rows_seen = 0, win_moves = 0; uint32_t win_outside_seen = 0;   // rows fed to it / at the last check; its counters then
};

int k_fgb_plan_new_uncleared(hark_context *ctx, hark_fgb_plan **out, int64_t max_rows, int64_t G);
int k_gen_columns(hark_context *ctx, uint64_t seed, int64_t first_row, int64_t n, uint32_t G,
                  int exact, float *p, int32_t *k, float *v);
int hark_fgb_finish_typed(hark_context *ctx, hark_fgb_plan *pl, int32_t kind, const uint32_t *pos, void *out);
int k_fgb_dense_f32(hark_context *ctx, hark_fgb_plan *plan, const float *p, int cmp, float thr,
                    const int32_t *k, const float *v, int64_t n);

// The hash partition of a (key column, value column) pair, kept between aggregates of the SAME value column (the pairs
// the producer writes carry the raw value bits: they depend neither on the operator nor on its order transform):
// zero-initialise, pass to every k_fgb_hash_u32 call, release with k_fgb_hash_part_free.
struct hark_hash_part { void *pbuf = nullptr; uint32_t *counts = nullptr; int64_t cap = 0, n = 0; const void *k = nullptr, *v = nullptr, *p = nullptr; int xf = 0; };
// a WHERE fused into a producer: f32 column `p` compared with `thr` (HARK_CMP_GT..NE), or a survivor bitmask (HARK_CMP_MASK)
struct hark_row_pred { const float *p; int cmp; float thr; };
void k_fgb_hash_part_free(hark_context *ctx, hark_hash_part *part);
// why k_fgb_hash_u32 reports *fits == false
enum { HARK_HASH_FITS = 0, HARK_HASH_NOFIT_DISTINCT = 1 /* too many distinct keys */, HARK_HASH_NOFIT_SKEW = 2 /* a partition slab overflowed */,
       HARK_HASH_NOFIT_ROWS = 3 /* n == 0 or n >= 2^32: nothing learnt about the column */ };
int k_fgb_hash_u32(hark_context *ctx, const uint32_t *k, const uint32_t *v, int64_t n, int vop, int xf,
                   uint32_t **keys_out, unsigned long long **vals_out, unsigned long long **cnts_out, int64_t *G_out, bool *fits,
                   uint32_t *rounds_hint, bool compact, hark_hash_part *part = nullptr, int *why_not = nullptr, const hark_row_pred *pred = nullptr,
                   int stats_vk = -1, unsigned long long **mins_out = nullptr, unsigned long long **maxs_out = nullptr, uint32_t ref_ops = 0);

int k_fgb_dense_stats(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr,
                      const int32_t *k, const void *v, int64_t n, int vk, bool *ran);
int hark_fgb_finish_typed_from(hark_context *ctx, hark_fgb_plan *pl, int32_t which, int32_t kind, const uint32_t *pos, void *out);
int hark_fgb_finish_u32_second(hark_context *ctx, hark_fgb_plan *pl, uint32_t *val_out);
int hark_fgb_finish_u32_of(hark_context *ctx, hark_fgb_plan *pl, int which, uint32_t *val_out);
int k_fgb_dense_pair(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr, const int32_t *k,
                     const void *v1, int vop1, int xf1, const void *v2, int vop2, int xf2, int64_t n, bool *ran);
int k_fgb_dense_multi(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr, const int32_t *k,
                      const void *v1, int vop1, int xf1, const void *v2, int vop2, int xf2, const void *v3, int vop3, int xf3, int64_t n, bool *ran);
int k_fgb_decode(hark_context *ctx, const unsigned long long *acc, const unsigned long long *cnt, int64_t G, int kind, void *out);

// k_select.hip
int k_predicate_bitmask(hark_context *ctx, const hark_table *db, int64_t n_preds, const int32_t *where_cols, const int32_t *cmps,
                        const void *const *constants, uint8_t **mask_out);
int k_groupby_typed(hark_context *ctx, const hark_table *db, int32_t g_col, const int32_t *agg_cols,
                    const int32_t *agg_ops, int64_t n_aggs, hark_result *res);
int k_gather_columns(hark_context *ctx, const hark_table *db, const int32_t *cols, int64_t k,
                     hark_result *res);

#ifdef __HIPCC__
// 16-byte non-temporal load / store: columns that a kernel streams through exactly once.  Measured on the fused
// group-by's single-pass kernel: 0.77 -> 0.85 of the HBM peak against plain loads (profiles/r02_notes.md).
__device__ __forceinline__ uint4 ld_nt16(const void *p)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v t = __builtin_nontemporal_load(static_cast<const u4v *>(p));
    return uint4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ void st_nt16(void *p, uint4 v)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u4v{v.x, v.y, v.z, v.w}, static_cast<u4v *>(p));
}

// Stores the compiler's s_waitcnt bookkeeping does not see.  gfx950 has ONE in-order counter (vmcnt) for loads and stores,
// and the compiler places the wait for a prefetched load by counting the memory instructions issued after it -- over all
// paths.  A store inside a conditional or a loop makes that count unknown, and the wait becomes vmcnt(0): the wave then
// sits out the prefetch it has just issued (measured: the join's bucket kernel streamed at 5.6 TB/s without its survivor
// stores and ran at 2.9 with them).  Issued from inline assembly the stores are not counted by the compiler; the
// hardware still counts them, which only makes a compiler-placed vmcnt(N) wait for a little more than it needs -- never
// for less: older operations retire first.  Use for fire-and-forget results only (nothing in the same kernel reads them
// back without a fence; the kernel's end completes them).
__device__ __forceinline__ void st_hidden_b32(void *p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_hidden_b64(void *p, uint2 v)
{
    const unsigned long long w = ((unsigned long long)v.y << 32) | v.x;
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(w) : "memory");
}
// (A store of MORE than 64 bits reads its data registers a few cycles after it issues: the ISA asks for wait states before a
// VALU instruction overwrites them.  The compiler's hazard recognizer inserts them for stores it knows; it cannot see into
// an asm statement, so the s_nop is part of it -- found the hard way in round 4: a 16-byte record store followed by the
// address arithmetic of the next store, which the register allocator had placed in the record's last two registers, wrote
// two address words into the record.)
__device__ __forceinline__ void st_hidden_nt_b128(void *p, uint4 v)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
}

// Loads with hand-placed waits, for software pipelines the compiler's bookkeeping cannot follow (it falls back to
// vmcnt(0) at a loop header that merges paths it cannot count): ld_hidden_* issues the load, the value MUST NOT be touched
// until wait_vm<N>(regs...) has passed them through an s_waitcnt -- N = the number of memory instructions issued after
// the load that may still be outstanding (loads AND stores, hidden or not: one in-order counter).  Keep each value in
// variables of its own between the two (no copies: the compiler would copy a register the load has not written yet).
typedef unsigned int hark_u4v __attribute__((ext_vector_type(4)));
typedef unsigned int hark_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ld_hidden_nt_b128(hark_u4v &dst, const void *p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld_hidden_nt_b64(hark_u2v &dst, const void *p) { asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm(hark_u4v &a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm(hark_u4v &a, hark_u2v &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm(hark_u4v &a, hark_u4v &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL
// address spaces: the compiler drains vmcnt before s_barrier, so a barrier inside a streaming loop waits
// for every global load and store the lanes still have in flight.  Where lanes exchange data through LDS
// only, lgkmcnt(0) is enough and the global traffic keeps flowing across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif
