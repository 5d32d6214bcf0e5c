// k_fgb.hip -- fused WHERE -> GROUP BY over a dense key domain (headline: SUM f32 + COUNT).
//
// Replaces, for the headline query of BASELINE.json
//     SELECT k, SUM(v), COUNT(*) FROM t WHERE p <cmp> thr GROUP BY k
// the reference pipeline  filter -> materialise (groupby.fut:52-53) -> 32 x
// 1-bit radix split of whole rows (groupby.fut:8-22) -> head flags (:26-33)
// -> segmented scan + tail scatter (segmented.fut:7-37).  That pipeline moves
// every row ~32 times; here each referenced byte is read from HBM once:
//
//   algorithmic bytes = 12 B/row (p, k, v)  +  16 B/group (key, sum, count).
//
// Accumulators are f64 sums + integer counts.  Measured on gfx950 (tools/
// ldsbench.hip): ds_add_f32 retires ~0.4 lanes/clk/CU, ds_add_f64 ~3.5,
// ds_add_u32 ~10, so sums are accumulated in double (which also makes the
// result independent of accumulation order to ~1e-16, far inside the 1e-5 bar)
// and rounded to f32 once at the end (hark_fgb_finish).
//
// Three device paths, chosen by the number of groups G:
//   LDS   (12 B x G fits a workgroup's LDS): every workgroup keeps a private
//         (sum,count) table in LDS (replicated per lane for tiny G to spread
//         same-address atomics), streams its rows with 16-byte loads and
//         flushes once with contiguous global atomics.
//   PART  (large G, e.g. 2^20): scattered global atomics retire ~21-27 G/s
//         on gfx950 (tools/ubench.hip: they execute at the memory side, one
//         64-byte request per lane), 20x too slow.  So rows are routed by key
//         range first.  PRODUCER: every bucket (key >> shift) owns a small
//         ring in LDS; a surviving row takes a slot with one ds_add_rtn_u32,
//         and only complete 128-byte lines leave the CU, appended to the
//         workgroup's OWN slab of that bucket (no global atomics, no partial
//         lines): units of 64 compact pairs (u16 bucket-local key + 32-bit
//         value, 6 B per pair) on the dense path, lines of 16 (key, value)
//         pairs in hash mode.  Heavy hitters are folded in a small LDS cache
//         instead.  The batch loop carries no run-time knobs and its barriers
//         order LDS only (it was instruction-issue bound).
//         CONSUMER: one workgroup per bucket folds the bucket's slabs into an
//         LDS-resident slice of the table and adds the slice to the global
//         table with plain stores (it owns the key range).
//   HASH  arbitrary u32 keys: the producer routes by the top bits of
//         mix32(key), one workgroup per bucket builds an open-addressing
//         table in LDS (fgb_agg_hash_kernel: value slot, row count and a
//         16-bit tag per entry; fgb_agg_hash_stats_kernel: five aggregates of
//         one column; fgb_agg_hash_ops_kernel: one to three of the reference's
//         u32 operators), in several rounds when a bucket holds more distinct
//         keys than a table.
//   ATOM  one global atomic pair per surviving row; the fallback for slab
//         overflow / G > 2^21 and a measured baseline.
//
// The same kernels serve other aggregates through a run-time "value operator"
// (u32 sum / max / min / product, u64 sum) and an order-preserving value
// transform (signed and float min/max); see vop_* below and try_dense() in
// k_groupby.hip.
#include "hark_internal.h"
#include <type_traits>

namespace {

constexpr int kVec = 4;                 // rows per 16-byte load
constexpr int kNoPred = -1;
// HARK_CMP_MASK: the predicate column is a precomputed 1-bit-per-row survivor bitmask (bit r & 7 of byte r >> 3:
// an AND-list of predicates on any dtypes, evaluated once by k_predicate_bitmask).  0.125 B/row instead of 4.
constexpr int kMaskPred = HARK_CMP_MASK;

// the four mask bits of rows r..r+3 (r a multiple of 4) as 1.0f / 0.0f "predicate values"
__device__ __forceinline__ float4 mask_nibble(const float *p, int64_t r)
{
    const uint32_t byte = reinterpret_cast<const uint8_t *>(p)[r >> 3], nib = (byte >> (r & 4)) & 15u;
    return float4{(float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)(nib >> 3)};
}
__device__ __forceinline__ float mask_bit(const float *p, int64_t r)
{
    return (float)((reinterpret_cast<const uint8_t *>(p)[r >> 3] >> (r & 7)) & 1u);
}

template <int OP>
__device__ __forceinline__ bool cmp_f32(float a, float b)
{
    if constexpr (OP == HARK_CMP_GT) return a > b;
    else if constexpr (OP == HARK_CMP_GE) return a >= b;
    else if constexpr (OP == HARK_CMP_LT) return a < b;
    else if constexpr (OP == HARK_CMP_LE) return a <= b;
    else if constexpr (OP == HARK_CMP_EQ) return a == b;
    else if constexpr (OP == HARK_CMP_NE) return a != b;
    else if constexpr (OP == kMaskPred) return a != 0.0f;
    else return true;                   // kNoPred
}

// ---- value operators of the accumulators ------------------------------------
// One 8-byte slot per group (+ a count).  F32SUM keeps a double; the u32 operators
// of the reference's type_func (groupby.fut:35-41: wrapping +, *, max, min) keep the
// value in the low word.  LDS forms: ds_add_f64 / ds_add_u32 / ds_max_u32 / ds_min_u32
// (native, 3.5-11 lanes/clk), product by an LDS compare-and-swap loop.
typedef unsigned long long u64;
enum { VOP_F32SUM = 0, VOP_U32SUM = 1, VOP_U32MAX = 2, VOP_U32MIN = 3, VOP_U32PROD = 4, VOP_U32SUM64 = 5 };
// Value transform applied to the raw 32 bits before accumulation, so that the unsigned
// operators also serve signed and floating-point columns: 1 = i32 -> order-preserving u32
// (x ^ 2^31), 2 = f32 -> order-preserving u32.  hark_fgb_finish_typed decodes.
enum { XF_NONE = 0, XF_I32_ORDER = 1, XF_F32_ORDER = 2 };
__device__ __forceinline__ uint32_t apply_xf(int xf, uint32_t x)
{
    if (xf == XF_I32_ORDER) return x ^ 0x80000000u;
    if (xf == XF_F32_ORDER) { if (x == 0x80000000u) x = 0u; return x ^ ((x & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u); }
    return x;
}

__host__ __device__ inline u64 vop_identity(int vop)
{
    return vop == VOP_U32MIN ? 0xFFFFFFFFull : vop == VOP_U32PROD ? 1ull : 0ull;    // 0.0 has all-zero bits
}

__device__ __forceinline__ u64 vop_merge(int vop, u64 a, u64 b)
{
    switch (vop) {
    case VOP_F32SUM: return (u64)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    case VOP_U32SUM: return (uint32_t)((uint32_t)a + (uint32_t)b);
    case VOP_U32MAX: return (uint32_t)a > (uint32_t)b ? (uint32_t)a : (uint32_t)b;
    case VOP_U32MIN: return (uint32_t)a < (uint32_t)b ? (uint32_t)a : (uint32_t)b;
    case VOP_U32SUM64: return a + b;
    default: return (uint32_t)((uint32_t)a * (uint32_t)b);
    }
}

// combine the value `x` (raw 32 bits of the column element) into an 8-byte slot with atomics;
// works on LDS and on global memory (the compiler picks ds_* or global_* from the address space)
template <int VOP>
__device__ __forceinline__ void vop_atomic(u64 *slot, uint32_t x)
{
    uint32_t *lo = reinterpret_cast<uint32_t *>(slot);
    if constexpr (VOP == VOP_F32SUM) unsafeAtomicAdd(reinterpret_cast<double *>(slot), (double)__uint_as_float(x));
    else if constexpr (VOP == VOP_U32SUM) atomicAdd(lo, x);
    else if constexpr (VOP == VOP_U32MAX) atomicMax(lo, x);
    else if constexpr (VOP == VOP_U32MIN) atomicMin(lo, x);
    else if constexpr (VOP == VOP_U32SUM64) atomicAdd(slot, (u64)x);            // ds_add_u64 / global_atomic_add_x2
    else {
        uint32_t old = *lo, assumed;
        do { assumed = old; old = atomicCAS(lo, assumed, assumed * x); } while (old != assumed);
    }
}

// same with an already-accumulated partial (slot value `part`) instead of a column element
template <int VOP>
__device__ __forceinline__ void vop_atomic_partial(u64 *slot, u64 part)
{
    if constexpr (VOP == VOP_F32SUM) unsafeAtomicAdd(reinterpret_cast<double *>(slot), __longlong_as_double((long long)part));
    else if constexpr (VOP == VOP_U32SUM64) atomicAdd(slot, part);
    else vop_atomic<VOP>(slot, (uint32_t)part);
}

__device__ __forceinline__ void vop_atomic_partial_rt(int vop, u64 *slot, u64 part)
{
    switch (vop) {
    case VOP_F32SUM: vop_atomic_partial<VOP_F32SUM>(slot, part); break;
    case VOP_U32SUM: vop_atomic_partial<VOP_U32SUM>(slot, part); break;
    case VOP_U32MAX: vop_atomic_partial<VOP_U32MAX>(slot, part); break;
    case VOP_U32MIN: vop_atomic_partial<VOP_U32MIN>(slot, part); break;
    case VOP_U32SUM64: vop_atomic_partial<VOP_U32SUM64>(slot, part); break;
    default: vop_atomic_partial<VOP_U32PROD>(slot, part); break;
    }
}

__device__ __forceinline__ void vop_atomic_rt(int vop, u64 *slot, uint32_t x)
{
    switch (vop) {
    case VOP_F32SUM: vop_atomic<VOP_F32SUM>(slot, x); break;
    case VOP_U32SUM: vop_atomic<VOP_U32SUM>(slot, x); break;
    case VOP_U32MAX: vop_atomic<VOP_U32MAX>(slot, x); break;
    case VOP_U32MIN: vop_atomic<VOP_U32MIN>(slot, x); break;
    case VOP_U32SUM64: vop_atomic<VOP_U32SUM64>(slot, x); break;
    default: vop_atomic<VOP_U32PROD>(slot, x); break;
    }
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void gen_columns_kernel(uint64_t seed, int64_t first_row, int64_t n, uint32_t G, int pow2,
                                   int exact, float *__restrict__ p, int32_t *__restrict__ k,
                                   float *__restrict__ v)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = splitmix64(seed + (uint64_t)(first_row + i));
        if (k) k[i] = pow2 ? (int32_t)(h & (uint64_t)(G - 1)) : (int32_t)(h % (uint64_t)G);
        if (p) p[i] = (float)((h >> 20) & 0xFFFFFF) * (1.0f / 16777216.0f);
        if (v) v[i] = exact ? (float)((h >> 44) & 15) : (float)((h >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
    }
}

// ---------------------------------------------------------------------------
// LDS-privatised path
// ---------------------------------------------------------------------------
// Dynamic LDS: u64 s_sum[G << RL] (8-byte value slots); uint32 s_cnt[G << RL].  The
// replica of a key used by lane l is (key << RL) | (l & (R-1)).
// FSUM: the f32-sum operator is compiled in (the headline path: a run-time operator switch in
// the row loop costs 7 % there); otherwise the operator is the wave-uniform run-time `vop`.
// VM: 0 = run-time value operator, 1 = f32 sum compiled in, 2 = COUNT only (no value column is read).
template <int OP, int VM>
__global__ __launch_bounds__(1024) void fgb_lds_kernel(
    const float *__restrict__ p, const int32_t *__restrict__ k, const float *__restrict__ v,
    int64_t n, float thr, int G, int RL, u64 *__restrict__ gsum,
    unsigned long long *__restrict__ gcnt, int32_t *__restrict__ err, int xf, int vop_rt)
{
    constexpr bool FSUM = VM == 1, CNT = VM == 2;
    const int VOP = (FSUM || CNT) ? (int)VOP_F32SUM : vop_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int slots = G << RL;
    u64 *s_sum = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * slots);
    for (int i = threadIdx.x; i < slots; i += blockDim.x) { s_sum[i] = vop_identity(VOP); s_cnt[i] = 0u; }
    __syncthreads();

    const uint32_t rep = threadIdx.x & ((1u << RL) - 1u);
    const int64_t nvec = n / kVec;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    const int4 *k4 = reinterpret_cast<const int4 *>(k);
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    bool bad = false;

    auto row = [&](float pv, int32_t key, float val) {
        if (cmp_f32<OP>(pv, thr)) {
            if ((uint32_t)key < (uint32_t)G) {
                uint32_t s = ((uint32_t)key << RL) | rep;
                if constexpr (FSUM) vop_atomic<VOP_F32SUM>(&s_sum[s], __float_as_uint(val));      // ds_add_f64
                else if constexpr (!CNT) vop_atomic_rt(VOP, &s_sum[s], apply_xf(xf, __float_as_uint(val)));   // ds_{add,max,min}_u32 / ds_add_u64
                atomicAdd(&s_cnt[s], 1u);                           // ds_add_u32
            } else bad = true;
        }
    };

    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // D independent 16-byte loads per column in flight per lane, streamed once: non-temporal (measured on one box against
    // plain loads: 1.96 ms -> 1.76 ms per 1e9 rows, 0.77 -> 0.85 of peak; three or four loads in flight are slower)
    constexpr int D = 2;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef int i4v __attribute__((ext_vector_type(4)));
    auto ldp = [&](int64_t q) -> float4 {
        if (OP == kNoPred) return float4{0, 0, 0, 0};
        if (OP == kMaskPred) return mask_nibble(p, q * kVec);
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p4 + q));
        return float4{t.x, t.y, t.z, t.w};
    };
    auto ldk = [&](int64_t q) -> int4 {
        const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(k4 + q));
        return int4{t.x, t.y, t.z, t.w};
    };
    auto ldv = [&](int64_t q) -> float4 {
        if (CNT) return float4{0, 0, 0, 0};
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(v4 + q));
        return float4{t.x, t.y, t.z, t.w};
    };
    for (; i + (D - 1) * stride < nvec; i += D * stride) {
        float4 pa[D], va[D]; int4 ka[D];
#pragma unroll
        for (int d = 0; d < D; d++) { pa[d] = ldp(i + d * stride); ka[d] = ldk(i + d * stride); va[d] = ldv(i + d * stride); }
#pragma unroll
        for (int d = 0; d < D; d++) {
            row(pa[d].x, ka[d].x, va[d].x); row(pa[d].y, ka[d].y, va[d].y); row(pa[d].z, ka[d].z, va[d].z); row(pa[d].w, ka[d].w, va[d].w);
        }
    }
    for (; i < nvec; i += stride) {
        const float4 pa = ldp(i), va = ldv(i);
        const int4 ka = ldk(i);
        row(pa.x, ka.x, va.x); row(pa.y, ka.y, va.y); row(pa.z, ka.z, va.z); row(pa.w, ka.w, va.w);
    }
    // ragged tail (n % 4 rows) by the first lanes of block 0
    if (blockIdx.x == 0) {
        int64_t t = nvec * kVec + threadIdx.x;
        if (t < n) row(OP == kNoPred ? 0.0f : OP == kMaskPred ? mask_bit(p, t) : p[t], k[t], CNT ? 0.0f : v[t]);
    }
    if (bad) *err = HARK_EBOUNDS;
    __syncthreads();

    const int R = 1 << RL;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        u64 s = vop_identity(VOP); uint32_t c = 0;
        for (int r = 0; r < R; r++) { s = vop_merge(VOP, s, s_sum[(g << RL) + r]); c += s_cnt[(g << RL) + r]; }
        if (c) {
            if constexpr (FSUM) vop_atomic_partial<VOP_F32SUM>(&gsum[g], s);   // contiguous global_atomic_add_f64
            else if constexpr (!CNT) vop_atomic_partial_rt(VOP, &gsum[g], s);
            atomicAdd(&gcnt[g], (unsigned long long)c);
        }
    }
}

// ---------------------------------------------------------------------------
// Window path: key columns that are SORTED or CLUSTERED (a table kept in key order)
// ---------------------------------------------------------------------------
// The partition path below routes every surviving row through per-bucket LDS rings sized for keys that SCATTER over the
// buckets; consecutive rows of a sorted key column all fall into ONE bucket, whose ring of ~100 entries is swept ~40 times
// per batch of 4096 rows: the headline statement took 53 ms per 1e9 rows on sorted keys against 3.1 ms on shuffled ones,
// 121 against 3.4 without a predicate (tools/groupby_cluster_probe.py).  What hurts there is what makes such columns
// cheap: the rows of a batch hold a handful of neighbouring keys.  So every workgroup takes a CONTIGUOUS stretch of the
// table and aggregates batch after batch into a WINDOW of kWinKeys consecutive keys in LDS (fgb_lds_kernel's slots: a double
// or a word per key + a count, kWinRep replicas -- the lanes of a wave mostly hold the SAME key), which follows the keys:
// rows inside the window are added there; when many rows of a batch lie outside, the window is added to the global accumulators
// (one global atomic per key it holds) and put around a key the wave with the most waiting rows picked.  A FEW rows outside --
// late rows in a table that is in order otherwise -- go to the global accumulators one by one.  One pass, 12 B/row, no pair is
// written.  A batch whose rows lie in more clusters than a batch gets turns sends the rest to global atomics too (correct,
// slow): the caller picks this path only for columns whose rows an eighth of a batch apart are a few keys apart
// (fgb_cluster_test_kernel), reads back how many rows went that way, and drops the verdict when it was more than a third.
constexpr int kWinKeys = 1024, kWinRL = 3, kWinRep = 1 << kWinRL, kWinNear = kWinKeys / 4, kWinFar = 512, kWinTurns = 4, kWinStray = 256;

// stat[0]: times a window moved, stat[1]: rows that went to global atomics
template <int OP, int VM>
__global__ __launch_bounds__(1024) void fgb_window_kernel(
    const float *__restrict__ p, const int32_t *__restrict__ k, const float *__restrict__ v,
    int64_t n, float thr, int64_t G, u64 *__restrict__ gsum, unsigned long long *__restrict__ gcnt,
    int32_t *__restrict__ err, int xf, int vop_rt, uint32_t *__restrict__ stat)
{
    constexpr bool FSUM = VM == 1, CNT = VM == 2;
    const int VOP = (FSUM || CNT) ? (int)VOP_F32SUM : vop_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int SLOTS = kWinKeys << kWinRL;
    u64 *s_sum = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * SLOTS);
    __shared__ uint32_t s_out[3];                                        // rows of a turn outside the window (three sets of words in rotation, see below)
    __shared__ unsigned long long s_best[3];                             // ... and the wave with the most of them: rows << 32 | one of their keys
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < SLOTS; i += 1024) { s_sum[i] = vop_identity(VOP); s_cnt[i] = 0u; }
    if (tid < 3) { s_out[tid] = 0u; s_best[tid] = 0ull; }
    __syncthreads();
    const uint32_t rep = (uint32_t)tid & (uint32_t)(kWinRep - 1);
    const int64_t nvec = n / kVec;
    const int64_t per = ((nvec + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;      // 16-byte groups per workgroup: whole batches
    const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nvec ? q0 + per : nvec;
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    const int4 *k4 = reinterpret_cast<const int4 *>(k);
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef int i4v __attribute__((ext_vector_type(4)));
    auto ldp = [&](int64_t q) -> float4 {
        if (OP == kNoPred) return float4{0, 0, 0, 0};
        if (OP == kMaskPred) return mask_nibble(p, q * kVec);
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p4 + q));
        return float4{t.x, t.y, t.z, t.w};
    };
    auto ldk = [&](int64_t q) -> int4 {
        const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(k4 + q));
        return int4{t.x, t.y, t.z, t.w};
    };
    auto ldv = [&](int64_t q) -> float4 {
        if (CNT) return float4{0, 0, 0, 0};
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(v4 + q));
        return float4{t.x, t.y, t.z, t.w};
    };
    bool bad = false, open = false;                                       // open: a window exists (the first batch makes one)
    int base = 0;                                                         // the window [base, base + kWinKeys)
    uint32_t moved = 0, outside = 0;
    // the window -> global accumulators: a key per thread, all of them (every thread calls it, behind a turn's barrier -- the rows of that turn
    // are in --; ends with a barrier).  (Tracking the touched range cost a reduction and a barrier more per move than reading 1024 keys does.)
    auto write_out = [&]() {
        for (int g = tid; g < kWinKeys; g += 1024) {
            u64 sacc = vop_identity(VOP); uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < kWinRep; r++) {
                const int slot = (g << kWinRL) + r;
                const uint32_t cr = s_cnt[slot];
                if (cr) { sacc = vop_merge(VOP, sacc, s_sum[slot]); c += cr; s_sum[slot] = vop_identity(VOP); s_cnt[slot] = 0u; }
            }
            if (c) {
                if constexpr (FSUM) vop_atomic_partial<VOP_F32SUM>(&gsum[base + g], sacc);
                else if constexpr (!CNT) vop_atomic_partial_rt(VOP, &gsum[base + g], sacc);
                atomicAdd(&gcnt[base + g], (unsigned long long)c);
            }
        }
        __syncthreads();
    };
    auto to_global = [&](int key, uint32_t x) {                           // a row the window does not hold: straight to the accumulators
        if constexpr (FSUM) vop_atomic<VOP_F32SUM>(&gsum[key], x);
        else if constexpr (!CNT) vop_atomic_rt(VOP, &gsum[key], apply_xf(xf, x));
        atomicAdd(&gcnt[key], 1ull);
        outside++;
    };
    float4 pn = float4{0, 0, 0, 0}, vn = pn;
    int4 kn = int4{0, 0, 0, 0};
    if (q0 + tid < q1) { pn = ldp(q0 + tid); kn = ldk(q0 + tid); vn = ldv(q0 + tid); }
    int ph = 0;                                                           // the words this turn publishes in
    for (int64_t qb = q0; qb < q1; qb += 1024) {
        const int64_t q = qb + tid;
        const bool have = q < q1;
        const float4 pa = pn, va = vn;
        const int4 ka = kn;
        if (q + 1024 < q1) { pn = ldp(q + 1024); kn = ldk(q + 1024); vn = ldv(q + 1024); }      // the next batch travels while this one is added
        const float pv[4] = {pa.x, pa.y, pa.z, pa.w}, vv[4] = {va.x, va.y, va.z, va.w};
        const int kk[4] = {ka.x, ka.y, ka.z, ka.w};
        uint32_t live = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (have && cmp_f32<OP>(pv[j], thr)) {
                if ((uint32_t)kk[j] < (uint64_t)G) live |= 1u << j;
                else bad = true;
            }
        // A turn: the rows inside the window are added; the others are counted (and one of their keys per wave remembered).  None
        // outside -- the usual batch of a column in key order --: done, one barrier.  A few (kWinStray: stray keys in a column that
        // is in order otherwise, late rows; the first version moved the window to the SMALLEST waiting key and spent its turns on
        // the strays: 55 ms per 1e9 rows with one row in a thousand out of place): straight to the global accumulators.  Many: the
        // keys have moved on -- the window is written out and put where one of the waiting rows is (most of them are its
        // neighbours), up to kWinTurns times per batch (a batch that straddles clusters: sorted runs in shuffled order); what still
        // waits after the last turn goes to the global accumulators too.
        // (the words: a turn publishes in set ph, reads it behind the barrier, and clears the set of the turn before it -- last read
        // before this barrier, next written behind the next one)
        for (int turn = 0;; turn++) {
            const uint32_t n0 = (uint32_t)__popc(live);                  // this lane's rows at the turn's start
            if constexpr (FSUM || CNT) {
                // a wave whose surviving rows all hold ONE key (a sorted column: ~950 rows per key at the headline's sizes) adds them up
                // in registers and touches the window once, instead of 64 lanes queueing at the key's eight replicas
                const unsigned long long lv = __ballot(live != 0u);
                if (open && lv != 0ull) {
                    const int mine = kk[live ? __ffs((int)live) - 1 : 0];
                    const int k0 = __shfl(mine, __ffsll((long long)lv) - 1, 64);
                    bool same = true;
#pragma unroll
                    for (int j = 0; j < 4; j++) if ((live & (1u << j)) && kk[j] != k0) same = false;
                    if (__ballot(!same) == 0ull && (uint32_t)(k0 - base) < (uint32_t)kWinKeys) {
                        double part = 0.0;
                        uint32_t c = 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) if (live & (1u << j)) { part += (double)vv[j]; c++; }
#pragma unroll
                        for (int d = 32; d > 0; d >>= 1) {
                            const long long pb = __double_as_longlong(part);
                            const int lo2 = __shfl_xor((int)(uint32_t)pb, d, 64), hi2 = __shfl_xor((int)(uint32_t)((unsigned long long)pb >> 32), d, 64);
                            part += __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi2 << 32) | (uint32_t)lo2));
                            c += (uint32_t)__shfl_xor((int)c, d, 64);
                        }
                        if (lane == 0) {
                            const int off = k0 - base;
                            const uint32_t slot = ((uint32_t)off << kWinRL) | rep;
                            if constexpr (FSUM) unsafeAtomicAdd(reinterpret_cast<double *>(&s_sum[slot]), part);
                            atomicAdd(&s_cnt[slot], c);
                        }
                        live = 0;
                    }
                }
            }
            uint32_t nout = 0;
            int cand = -1;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (live & (1u << j)) {
                    const uint32_t off = (uint32_t)(kk[j] - base);
                    if (open && off < (uint32_t)kWinKeys) {
                        const uint32_t slot = (off << kWinRL) | rep;
                        const uint32_t x = __float_as_uint(vv[j]);
                        if constexpr (FSUM) vop_atomic<VOP_F32SUM>(&s_sum[slot], x);
                        else if constexpr (!CNT) vop_atomic_rt(VOP, &s_sum[slot], apply_xf(xf, x));
                        atomicAdd(&s_cnt[slot], 1u);
                        live &= ~(1u << j);
                    } else { nout++; if (cand < 0) cand = kk[j]; }
                }
            const unsigned long long pend = __ballot(nout != 0u);
            if (pend != 0ull) {                                           // (wave-uniform) the wave's waiting rows, and the key of a lane in the middle that has any
                uint32_t both = (n0 << 16) | nout;                        // (rows of the wave at the turn's start, rows that wait now: <= 256 each)
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) both += (uint32_t)__shfl_xor((int)both, d, 64);
                const uint32_t wsum = both & 0xFFFFu, wall = both >> 16;
                const int mid = (pend >> 32) != 0ull ? 31 + __ffsll((long long)(pend >> 32)) : 63 - __clzll((long long)pend);   // the first such lane from 32 on, else the last below
                const int c0 = __shfl(cand, mid, 64);
                // A wave MOST of whose rows wait says "the keys have moved on" (rows << 32 | key: the largest word is the wave with the most
                // of them); a wave with a few waiting rows holds strays, however many such waves there are (a fifth of all rows out of
                // place: 800 strays per batch, 50 per wave -- moving the window after them cost 57 ms per 1e9 rows)
                if (lane == 0) { atomicAdd(&s_out[ph], wsum); if (2u * wsum >= wall) atomicMax(&s_best[ph], ((unsigned long long)wsum << 32) | (uint32_t)c0); }
            }
            __syncthreads();
            const uint32_t tout = s_out[ph];                              // (one word in the usual case: nothing waits)
            const unsigned long long best = s_best[ph];
            const int pick = (int)(uint32_t)best;
            const int clr = ph == 0 ? 2 : ph - 1;
            ph = ph == 2 ? 0 : ph + 1;
            if (tid == 0) { s_out[clr] = 0u; s_best[clr] = 0ull; }
            if (tout == 0u) break;
            if (tout <= (uint32_t)kWinStray || best == 0ull || turn == kWinTurns - 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (live & (1u << j)) to_global(kk[j], __float_as_uint(vv[j]));
                break;
            }
            if (open) write_out();
            base = max(0, pick - kWinKeys / 2);                           // (the picked key lies somewhere inside its cluster: the window around it)
            open = true; moved++;
        }
    }
    __syncthreads();
    if (open) write_out();
    // ragged tail (n % 4 rows): the first lanes of workgroup 0, straight to the accumulators
    if (blockIdx.x == 0) {
        const int64_t t = nvec * kVec + tid;
        if (t < n && cmp_f32<OP>(OP == kNoPred ? 0.0f : OP == kMaskPred ? mask_bit(p, t) : p[t], thr)) {
            const int key = k[t];
            if ((uint32_t)key < (uint64_t)G) {
                const uint32_t x = CNT ? 0u : __float_as_uint(v[t]);
                if constexpr (FSUM) vop_atomic<VOP_F32SUM>(&gsum[key], x);
                else if constexpr (!CNT) vop_atomic_rt(VOP, &gsum[key], apply_xf(xf, x));
                atomicAdd(&gcnt[key], 1ull);
            } else bad = true;
        }
    }
    if (bad) *err = HARK_EBOUNDS;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) outside += __shfl_xor(outside, d, 64);
    if (lane == 0 && outside) atomicAdd(&stat[1], outside);
    if (tid == 0 && moved) atomicAdd(&stat[0], moved);
}

// The window path for the passes that keep several aggregates per group (k_fgb_dense_stats: SUM / COUNT / MIN / MAX of one column;
// k_fgb_dense_multi: two or three columns): per key and replica a 64-bit slot a0 (operator vop0 on column c0), the row count, and two
// 32-bit slots a1 / a2 (MAX or MIN of the order words of columns c1 / c2; c2 may be null) -- 20 B x 1024 keys x 4 replicas = 80 KiB.
// The operators are run-time values here (these passes are not the headline's).  Same windows, same turns as fgb_window_kernel.
struct WinAgg { const uint32_t *c0, *c1, *c2; int vop0, xf0, vop1, xf1, vop2, xf2; u64 *g0, *g1, *g2; };
constexpr int kWinXRL = 2, kWinXRep = 1 << kWinXRL;

template <int OP>
__global__ __launch_bounds__(1024) void fgb_windowx_kernel(
    const float *__restrict__ p, const int32_t *__restrict__ k, int64_t n, float thr, int64_t G, const WinAgg A,
    unsigned long long *__restrict__ gcnt, int32_t *__restrict__ err, uint32_t *__restrict__ stat)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int SLOTS = kWinKeys << kWinXRL;
    u64 *s_a0 = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * SLOTS), *s_a1 = s_cnt + SLOTS, *s_a2 = s_a1 + SLOTS;
    __shared__ uint32_t s_out[3];                                        // (the turns' words: as in fgb_window_kernel)
    __shared__ unsigned long long s_best[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool has2 = A.c2 != nullptr;
    const u64 id0 = vop_identity(A.vop0);
    const uint32_t id1 = (uint32_t)vop_identity(A.vop1), id2 = (uint32_t)vop_identity(A.vop2);
    for (int i = tid; i < SLOTS; i += 1024) { s_a0[i] = id0; s_cnt[i] = 0u; s_a1[i] = id1; s_a2[i] = id2; }
    if (tid < 3) { s_out[tid] = 0u; s_best[tid] = 0ull; }
    __syncthreads();
    const uint32_t rep = (uint32_t)tid & (uint32_t)(kWinXRep - 1);
    const int64_t nvec = n / kVec;
    const int64_t per = ((nvec + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;
    const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nvec ? q0 + per : nvec;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    auto ldp = [&](int64_t q) -> float4 {
        if (OP == kNoPred) return float4{0, 0, 0, 0};
        if (OP == kMaskPred) return mask_nibble(p, q * kVec);
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p) + q);
        return float4{t.x, t.y, t.z, t.w};
    };
    auto ldu = [&](const uint32_t *c, int64_t q) -> u4v { return __builtin_nontemporal_load(reinterpret_cast<const u4v *>(c) + q); };
    const bool same1 = A.c1 == A.c0, same2 = A.c2 == A.c0;
    bool bad = false, open = false;
    int base = 0;
    uint32_t moved = 0, outside = 0;
    auto global_row = [&](int key, uint32_t x0, uint32_t x1, uint32_t x2) {
        vop_atomic_rt(A.vop0, &A.g0[key], apply_xf(A.xf0, x0));
        const u64 w1 = apply_xf(A.xf1, x1);
        if (A.vop1 == VOP_U32MIN) atomicMin(&A.g1[key], w1); else atomicMax(&A.g1[key], w1);
        if (has2) { const u64 w2 = apply_xf(A.xf2, x2); if (A.vop2 == VOP_U32MIN) atomicMin(&A.g2[key], w2); else atomicMax(&A.g2[key], w2); }
        atomicAdd(&gcnt[key], 1ull);
    };
    auto write_out = [&]() {                                              // (all 1024 keys, one per thread: see fgb_window_kernel)
        for (int g = tid; g < kWinKeys; g += 1024) {
            u64 a0 = id0; uint32_t c = 0, a1 = id1, a2 = id2;
#pragma unroll
            for (int r = 0; r < kWinXRep; r++) {
                const int slot = (g << kWinXRL) + r;
                const uint32_t cr = s_cnt[slot];
                if (cr) {
                    a0 = vop_merge(A.vop0, a0, s_a0[slot]); c += cr;
                    a1 = A.vop1 == VOP_U32MIN ? min(a1, s_a1[slot]) : max(a1, s_a1[slot]);
                    a2 = A.vop2 == VOP_U32MIN ? min(a2, s_a2[slot]) : max(a2, s_a2[slot]);
                    s_a0[slot] = id0; s_cnt[slot] = 0u; s_a1[slot] = id1; s_a2[slot] = id2;
                }
            }
            if (c) {
                vop_atomic_partial_rt(A.vop0, &A.g0[base + g], a0);
                if (A.vop1 == VOP_U32MIN) atomicMin(&A.g1[base + g], (u64)a1); else atomicMax(&A.g1[base + g], (u64)a1);
                if (has2) { if (A.vop2 == VOP_U32MIN) atomicMin(&A.g2[base + g], (u64)a2); else atomicMax(&A.g2[base + g], (u64)a2); }
                atomicAdd(&gcnt[base + g], (unsigned long long)c);
            }
        }
        __syncthreads();
    };
    int ph = 0;
    for (int64_t qb = q0; qb < q1; qb += 1024) {
        const int64_t q = qb + tid;
        const bool have = q < q1;
        float4 pa = float4{0, 0, 0, 0};
        u4v ka = {0u, 0u, 0u, 0u}, x0 = ka, x1 = ka, x2 = ka;
        if (have) {                                                       // (no prefetch of the next batch here: five vectors more ran into scratch, 0.54 -> 0.65 ms per 1e8 rows)
            pa = ldp(q); ka = ldu(reinterpret_cast<const uint32_t *>(k), q); x0 = ldu(A.c0, q);
            x1 = same1 ? x0 : ldu(A.c1, q);
            if (has2) x2 = same2 ? x0 : ldu(A.c2, q);
        }
        const float pv[4] = {pa.x, pa.y, pa.z, pa.w};
        const int kk[4] = {(int)ka.x, (int)ka.y, (int)ka.z, (int)ka.w};
        const uint32_t v0[4] = {x0.x, x0.y, x0.z, x0.w}, v1[4] = {x1.x, x1.y, x1.z, x1.w}, v2[4] = {x2.x, x2.y, x2.z, x2.w};
        uint32_t live = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (have && cmp_f32<OP>(pv[j], thr)) {
                if ((uint32_t)kk[j] < (uint64_t)G) live |= 1u << j;
                else bad = true;
            }
        for (int turn = 0;; turn++) {                                     // (the turns: see fgb_window_kernel)
            const uint32_t n0 = (uint32_t)__popc(live);
            // a wave whose surviving rows all hold ONE key folds them in registers and touches the window once
            const unsigned long long lv = __ballot(live != 0u);
            if (open && lv != 0ull && A.vop0 != VOP_U32PROD) {
                const int mine = kk[live ? __ffs((int)live) - 1 : 0];
                const int k0 = __shfl(mine, __ffsll((long long)lv) - 1, 64);
                bool same = true;
#pragma unroll
                for (int j = 0; j < 4; j++) if ((live & (1u << j)) && kk[j] != k0) same = false;
                if (__ballot(!same) == 0ull && (uint32_t)(k0 - base) < (uint32_t)kWinKeys) {
                    u64 part = id0;
                    uint32_t c = 0, e1 = id1, e2 = id2;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (live & (1u << j)) {
                            const uint32_t x = apply_xf(A.xf0, v0[j]);
                            part = vop_merge(A.vop0, part, A.vop0 == VOP_F32SUM ? (u64)__double_as_longlong((double)__uint_as_float(x)) : (u64)x);
                            const uint32_t w1 = apply_xf(A.xf1, v1[j]);
                            e1 = A.vop1 == VOP_U32MIN ? min(e1, w1) : max(e1, w1);
                            if (has2) { const uint32_t w2 = apply_xf(A.xf2, v2[j]); e2 = A.vop2 == VOP_U32MIN ? min(e2, w2) : max(e2, w2); }
                            c++;
                        }
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) {
                        const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)part, d, 64), phi = (uint32_t)__shfl_xor((int)(uint32_t)(part >> 32), d, 64);
                        part = vop_merge(A.vop0, part, ((u64)phi << 32) | plo);
                        const uint32_t o1 = (uint32_t)__shfl_xor((int)e1, d, 64), o2 = (uint32_t)__shfl_xor((int)e2, d, 64);
                        e1 = A.vop1 == VOP_U32MIN ? min(e1, o1) : max(e1, o1);
                        e2 = A.vop2 == VOP_U32MIN ? min(e2, o2) : max(e2, o2);
                        c += (uint32_t)__shfl_xor((int)c, d, 64);
                    }
                    if (lane == 0) {
                        const int off = k0 - base;
                        const uint32_t slot = ((uint32_t)off << kWinXRL) | rep;
                        vop_atomic_partial_rt(A.vop0, &s_a0[slot], part);
                        if (A.vop1 == VOP_U32MIN) atomicMin(&s_a1[slot], e1); else atomicMax(&s_a1[slot], e1);
                        if (has2) { if (A.vop2 == VOP_U32MIN) atomicMin(&s_a2[slot], e2); else atomicMax(&s_a2[slot], e2); }
                        atomicAdd(&s_cnt[slot], c);
                    }
                    live = 0;
                }
            }
            uint32_t nout = 0;
            int cand = -1;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (live & (1u << j)) {
                    const uint32_t off = (uint32_t)(kk[j] - base);
                    if (open && off < (uint32_t)kWinKeys) {
                        const uint32_t slot = (off << kWinXRL) | rep;
                        vop_atomic_rt(A.vop0, &s_a0[slot], apply_xf(A.xf0, v0[j]));
                        const uint32_t w1 = apply_xf(A.xf1, v1[j]);
                        if (A.vop1 == VOP_U32MIN) atomicMin(&s_a1[slot], w1); else atomicMax(&s_a1[slot], w1);
                        if (has2) { const uint32_t w2 = apply_xf(A.xf2, v2[j]); if (A.vop2 == VOP_U32MIN) atomicMin(&s_a2[slot], w2); else atomicMax(&s_a2[slot], w2); }
                        atomicAdd(&s_cnt[slot], 1u);
                        live &= ~(1u << j);
                    } else { nout++; if (cand < 0) cand = kk[j]; }
                }
            const unsigned long long pend = __ballot(nout != 0u);
            if (pend != 0ull) {                                           // (wave-uniform) the wave's waiting rows, and the key of a lane in the middle that has any
                uint32_t both = (n0 << 16) | nout;                        // (rows of the wave at the turn's start, rows that wait now: <= 256 each)
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) both += (uint32_t)__shfl_xor((int)both, d, 64);
                const uint32_t wsum = both & 0xFFFFu, wall = both >> 16;
                const int mid = (pend >> 32) != 0ull ? 31 + __ffsll((long long)(pend >> 32)) : 63 - __clzll((long long)pend);   // the first such lane from 32 on, else the last below
                const int c0 = __shfl(cand, mid, 64);
                // A wave MOST of whose rows wait says "the keys have moved on" (rows << 32 | key: the largest word is the wave with the most
                // of them); a wave with a few waiting rows holds strays, however many such waves there are (a fifth of all rows out of
                // place: 800 strays per batch, 50 per wave -- moving the window after them cost 57 ms per 1e9 rows)
                if (lane == 0) { atomicAdd(&s_out[ph], wsum); if (2u * wsum >= wall) atomicMax(&s_best[ph], ((unsigned long long)wsum << 32) | (uint32_t)c0); }
            }
            __syncthreads();
            const uint32_t tout = s_out[ph];                              // (one word in the usual case: nothing waits)
            const unsigned long long best = s_best[ph];
            const int pick = (int)(uint32_t)best;
            const int clr = ph == 0 ? 2 : ph - 1;
            ph = ph == 2 ? 0 : ph + 1;
            if (tid == 0) { s_out[clr] = 0u; s_best[clr] = 0ull; }
            if (tout == 0u) break;
            if (tout <= (uint32_t)kWinStray || best == 0ull || turn == kWinTurns - 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (live & (1u << j)) { global_row(kk[j], v0[j], v1[j], v2[j]); outside++; }
                break;
            }
            if (open) write_out();
            base = max(0, pick - kWinKeys / 2);
            open = true; moved++;
        }
    }
    __syncthreads();
    if (open) write_out();
    if (blockIdx.x == 0) {                                                // ragged tail (n % 4 rows)
        const int64_t t = nvec * kVec + tid;
        if (t < n && cmp_f32<OP>(OP == kNoPred ? 0.0f : OP == kMaskPred ? mask_bit(p, t) : p[t], thr)) {
            const int key = k[t];
            if ((uint32_t)key < (uint64_t)G) global_row(key, A.c0[t], A.c1[t], has2 ? A.c2[t] : 0u);
            else bad = true;
        }
    }
    if (bad) *err = HARK_EBOUNDS;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) outside += __shfl_xor(outside, d, 64);
    if (lane == 0 && outside) atomicAdd(&stat[1], outside);
    if (tid == 0 && moved) atomicAdd(&stat[0], moved);
}

// ... and, failing that: do NEIGHBOURING rows (kRotClose apart: four lanes) share a bucket of the partition (within kRotNear keys: a
// quarter of the smallest bucket)?  Sorted files one behind the other (every block spans the whole key range: one row per key and
// block, nothing for a window to add up), sorted runs of a few hundred rows: a batch of 4096 consecutive rows then falls into one or a
// few rings (28 ms per 1e9 rows against 3.1 shuffled).  out[0] = 2: the producer reads every batch from 64 places (fgb_part_kernel's
// `rot`): 4.7 ms (emulated by regrouping the rows, tools/groupby_cluster_probe.py rot64:blocks1000000).
constexpr int kRotClose = 16, kRotNear = 1024;
// Are rows kWinFar rows (an eighth of a batch) apart a few keys apart?  1024 such pairs, and as many of rows anywhere apart (a column of
// few distinct keys is close to itself everywhere -- and needs no window).  out[0] = 1: clustered, [1] / [2]: the close pairs.
// `out` may be pinned host memory.
__global__ __launch_bounds__(1024) void fgb_cluster_test_kernel(const int32_t *__restrict__ k, int64_t n, unsigned long long *out)
{
    __shared__ int s_a[1024];
    __shared__ uint32_t s_c[4];
    const int tid = threadIdx.x;
    if (tid < 4) s_c[tid] = 0u;
    const int64_t r = (int64_t)(((uint64_t)mix32(0x9E3779B9u + (uint32_t)tid) * (uint64_t)(n - kWinFar)) >> 32);
    const int ka = k[r], kb = k[r + kWinFar], kc = k[r + kRotClose];
    s_a[tid] = ka;
    __syncthreads();
    const int kf = s_a[(tid + 512) & 1023];
    const bool near = abs((int64_t)ka - (int64_t)kb) <= kWinNear, far = abs((int64_t)ka - (int64_t)kf) <= kWinNear;
    const bool close = abs((int64_t)ka - (int64_t)kc) <= kRotNear, cfar = abs((int64_t)ka - (int64_t)kf) <= kRotNear;
    const unsigned long long m0 = __ballot(near), m1 = __ballot(far), m2 = __ballot(close), m3 = __ballot(cfar);
    if ((tid & 63) == 0) { atomicAdd(&s_c[0], __popcll(m0)); atomicAdd(&s_c[1], __popcll(m1)); atomicAdd(&s_c[2], __popcll(m2)); atomicAdd(&s_c[3], __popcll(m3)); }
    __syncthreads();
    if (tid == 0) {
        const uint32_t a = s_c[0], b = s_c[1], c = s_c[2], d = s_c[3];
        out[1] = a; out[2] = b; out[3] = c; out[4] = d;
        // three of eight pairs close (runs of 1024 sorted rows: half of them), and not because everything is: the window path; else
        // three of four NEIGHBOURING pairs in one bucket: the partition with rotated loads
        out[0] = (a >= 384u && b < 256u) ? 1ull : (c >= 768u && d < 256u) ? 2ull : 0ull;
        __threadfence_system();                                           // (`out` may be host memory)
    }
}

// ---------------------------------------------------------------------------
// Global-atomic path (fallback + baseline)
// ---------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void fgb_atomic_kernel(
    const float *__restrict__ p, const int32_t *__restrict__ k, const float *__restrict__ v,
    int64_t n, float thr, int64_t G, u64 *__restrict__ gsum,
    unsigned long long *__restrict__ gcnt, int32_t *__restrict__ err, int vop, int xf)
{
    const int64_t nvec = n / kVec;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    const int4 *k4 = reinterpret_cast<const int4 *>(k);
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    bool bad = false;
    auto row = [&](float pv, int32_t key, float val) {
        if (cmp_f32<OP>(pv, thr)) {
            if (key >= 0 && (int64_t)key < G) {
                if (v) vop_atomic_rt(vop, &gsum[key], apply_xf(xf, __float_as_uint(val)));      // v == null: COUNT only
                atomicAdd(&gcnt[key], 1ull);
            } else bad = true;
        }
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4 pa = OP == kNoPred ? float4{0, 0, 0, 0} : OP == kMaskPred ? mask_nibble(p, i * kVec) : p4[i];
        int4 ka = k4[i];
        float4 va = v ? v4[i] : float4{0, 0, 0, 0};
        row(pa.x, ka.x, va.x); row(pa.y, ka.y, va.y); row(pa.z, ka.z, va.z); row(pa.w, ka.w, va.w);
    }
    if (blockIdx.x == 0) {
        int64_t t = nvec * kVec + threadIdx.x;
        if (t < n) row(OP == kNoPred ? 0.0f : OP == kMaskPred ? mask_bit(p, t) : p[t], k[t], v ? v[t] : 0.0f);
    }
    if (bad) *err = HARK_EBOUNDS;
}

// ---------------------------------------------------------------------------
// Partition path, PRODUCER: route surviving (key,value) pairs by key range
// ---------------------------------------------------------------------------
// Software write-combining.  Every bucket owns a small circular queue in LDS
// (kQ pairs).  A batch of rows is enqueued with one returning LDS atomic per
// surviving row (ds_add_rtn_u32 on the bucket's fill count), then 8-lane groups
// sweep the buckets and store every complete 128-byte line (16 pairs, one
// dwordx4 per lane) to the workgroup's slab of that bucket.  Only whole, aligned
// lines ever leave the CU (appending the ~16-pair runs of a tile directly cost
// 2.2x the time of the same stores made contiguous: half-written lines get
// evicted), there is no sort of the tile, and two barriers per batch.
// A pair that finds its queue full stays in its lane and retries after the
// flush, so skewed keys slow the producer down but cannot overflow LDS; a full
// SLAB (heavier skew than the plan's slack) falls back to direct atomics.
constexpr int kLine = 16;                                // pairs per 128-byte line
constexpr int kQ = 64;                                   // queue capacity per bucket (pairs)
constexpr int kPartThreads = 1024;
constexpr int kBatchRows = kPartThreads * kVec;          // 4096 rows per batch
constexpr int kMaxBuckets = 256;                         // 256 * kQ * 8 B = 128 KiB of queues (one workgroup per CU)
constexpr int kTileRows = 8192;                          // chunk granularity (multiple of kBatchRows)

// Heavy-hitter cache: a direct-mapped LDS table of kHot (key, partial value, count) entries per
// workgroup.  A surviving row whose key owns its slot is folded there and never enters a queue;
// a slot is claimed by the first key that hashes to it.  With uniform keys it costs one LDS read
// per row; with skewed keys (tools/skew_bench.py: one hot key ran 700x slower without it) the hot
// keys are absorbed on chip.  Entries are added to the global table once, at the end of the kernel.
constexpr int kHotBits = 9, kHot = 1 << kHotBits;
constexpr uint32_t kHotEmpty = 0xFFFFFFFFu;
constexpr int kHotProbeBatches = 2;                      // a workgroup keeps the cache on only if > 1/16 of its first rows hit it
constexpr int kFlushPeriod = 4;                          // batches between queue sweeps (64-pair queues, ~8 new pairs per bucket and batch)
constexpr int kRetryRounds = 8;                          // queue-full retries per batch before direct atomics
constexpr int kRetryRoundsHash = 160;                    // hash mode has no atomics fallback: drain a hot bucket (4096 rows / 32 per round)
constexpr int kErrOverflow = 100;                        // device error word: a slab or a hash table overflowed (host picks another path)
// Compact pair format (dense keys): a pair is 6 bytes, a 16-bit bucket-local key and the 32-bit value,
// stored as UNITS of 64 pairs = 256 B of values followed by 128 B of keys -- three whole 128-byte lines, so
// the write combining keeps its whole-line property (a 64-B + 32-B split of 16 pairs did not).  The producer
// is at the HBM ceiling of its traffic mix, so 25 % fewer partition bytes are worth their price in LDS: rings
// of 96 pairs (one unit + 32 of headroom) per bucket, swept every second batch.
// (nwg x slab bytes is a multiple of 64 KiB, so all 256 slabs a workgroup sweeps share their low address bits; skewing the
// bucket rows against each other was measured flat, profiles/r03_bucket_skew.log)
constexpr int kU = 64;                                   // pairs per unit
constexpr int kUnitBytes = kU * 6;                       // 384
constexpr int kQ6 = 96;                                  // ring capacity per bucket (pairs)
constexpr int kFlushPeriod6 = 2;                         // batches between sweeps (every batch once a ring was found full)
// COUNT-only queries (no value column at all): the partition carries bucket-local keys only, units of 64 keys = one
// 128-byte line, rings of 128 keys; the producer does not even read a value column (8 B/row instead of 12).
constexpr int kQ2 = 128;
constexpr int kUnit2Bytes = kU * 2;                      // 128
// FMT 3 (round 3 A/B, pairfmt = 3): the same 6-byte units in HBM, but a ring entry is ONE 8-byte LDS word (value,
// bucket-local key) -- one ds_write_b64 per surviving row instead of a b32 and a b16 -- in at most 128 buckets of 8192
// keys: 128 rings of 144 entries (a unit + 80 of headroom) are 144 KiB.  Measured flat against FMT 1 (profiles/
// r03_notes.md): the ring stores were never what the producer waits for.
constexpr int kQ8e = 144;
constexpr int kMaxBuckets8e = 128;
constexpr int kFlushPeriod8e = 4;
// ---- batches of a producer launch are handed out first come, first served ---------------------------------------
// The XCDs of a card do not stream at the same rate: under a fixed assignment (batch = wg + j * nwg) the workgroup durations
// of one launch spread by 6-10 % (per-XCD means differ by up to 6 %, profiles/r02_notes.md 12), and the kernel lasts as long as
// its slowest workgroup.  A workgroup's step j works on batch seq(j): steps 0..3 are fixed (wg + j * nwg); from step 4 on
// thread 0 draws kDraw consecutive batches at a time from a device counter (the word behind the plan's error word, zeroed
// before every launch) -- asked for at step j, parked in an LDS ring at step j + 1, read by all threads from step j + 2 on,
// when the loads of those batches are issued: two barriers of slack, and the returned value is not touched for a whole
// batch, so nobody waits for the atomic.  One draw per batch cost 0.5 us per step (256 workgroups in step on one word);
// four batches per draw: producer 2.59 -> 2.49 ms on a fast card, 2.94 -> 2.58..2.77 on a slow one.
constexpr int kDraw = 4, kSeqRing = 8;
struct BatchSeq {
    uint32_t *ring;                                      // LDS [kSeqRing]
    uint32_t *ctr;
    uint64_t nbatch, first_drawn;
    uint32_t asked;
    int zero;
    // call before a barrier that precedes step 0
    __device__ __forceinline__ void init(uint32_t *ring_lds, int32_t *err, int64_t nb, int wg, int nwg, int tid)
    {
        ring = ring_lds; ctr = reinterpret_cast<uint32_t *>(err) + 1; nbatch = (uint64_t)nb; first_drawn = 4u * (uint64_t)nwg; asked = 0u;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));     // an opaque zero (see draw)
        if (tid < 2) ring[2 + tid] = (uint32_t)min((uint64_t)wg + (uint64_t)(2 + tid) * (uint64_t)nwg, nbatch);
    }
    // the batch whose loads are issued at step j (= the batch of step j + 2)
    __device__ __forceinline__ int64_t to_load(int j) const { return (int64_t)ring[(j + 2) & (kSeqRing - 1)]; }
    // thread 0, once per step, before the step's first barrier
    __device__ __forceinline__ void draw(int j)
    {
        if ((j & (kDraw - 1)) == 1) {                    // the numbers of steps j + 3 .. j + 6, asked for at step j - 1
#pragma unroll
            for (int q = 0; q < kDraw; q++) {
                const uint64_t id = (uint64_t)kDraw * asked + (uint64_t)q + first_drawn;
                ring[(j + 3 + q) & (kSeqRing - 1)] = id < nbatch ? (uint32_t)id : (uint32_t)nbatch;
            }
        }
        // (the offset is an opaque zero: with a provably uniform address the compiler's atomic optimizer wraps the atomic in
        // a wave reduction that reads the returned value -- and waits for it -- on the spot)
        if ((j & (kDraw - 1)) == 0) asked = atomicAdd(ctr + zero, 1u);
    }
};

static size_t part_lds_bytes(int P, int fmt)             // fmt: 0 = 8-byte pairs, 1 = compact 6-byte pairs, 2 = keys only
{
    const size_t q = fmt == 1 ? (size_t)6 * P * kQ6 : fmt == 2 ? (size_t)2 * P * kQ2 : fmt == 3 ? (size_t)8 * P * kQ8e
                   : sizeof(uint2) * (size_t)P * (P > kMaxBuckets ? kQ / 2 : kQ);
    return q + sizeof(int) * 2 * (size_t)P + 8 + (size_t)kHot * 16 + 64;      // ... + h_stat[8] + the ring of batch numbers seq[8]
}

// Workgroup-wide OR through one LDS word and ONE lds_barrier: three slots used in rotation, the next
// one cleared before the barrier (its last readers passed the previous barrier already).
__device__ __forceinline__ bool wg_or(bool pred, uint32_t *flags, int &phase)
{
    const int s = phase;
    phase = s == 2 ? 0 : s + 1;
    if (__ballot(pred) != 0ull && (threadIdx.x & 63) == 0) flags[s] = 1u;
    if (threadIdx.x == 0) flags[phase] = 0u;
    lds_barrier();
    return flags[s] != 0u;
}

// MODE 0: dense keys, f32 sum, no value transform (the headline path: nothing of the operator is decided at
// run time); MODE 1: dense keys, any value operator / transform (wave-uniform run-time switches);
// MODE 2: hash mode -- bucket = top bits of mix32(key), any u32 keys, no dense table behind the slabs.
// The kernel is bound by instruction issue (16 waves per CU, every wave-instruction costs four cycles of
// its SIMD; tools/fgb_ablate.py with synthetic rows and no memory traffic runs at 80 % of the full time),
// so the batch loop carries no run-time knobs: a bucket's queue state is ONE word (head << 16 | count,
// one returning LDS atomic hands a row its slot), keys are range-checked as unsigned 32-bit, and the
// ragged-end tests only run in the last batch.
// one surviving row straight into the global table (slab / ring overflow fallback); vop < 0: COUNT only
__device__ __noinline__ void fgb_direct_row(u64 *gsum, unsigned long long *gcnt, uint32_t key, uint32_t vb, int vop)
{
    if (vop >= 0) vop_atomic_rt(vop, &gsum[key], vb);
    atomicAdd(&gcnt[key], 1ull);
}

template <int OP, int MODE, int FMT, bool ROT = false /* rotated loads compiled in: see `rot` (the plain instantiation keeps its batch numbers in scalar registers; a run-time test alone moved the addresses into vector registers: +1 % on the headline) */>
__global__ __launch_bounds__(kPartThreads) void fgb_part_kernel(
    const float *__restrict__ p, const int32_t *__restrict__ k, const float *__restrict__ v,
    int64_t row0, int64_t row1, float thr, int64_t G, int shift, int P,
    uint2 *__restrict__ pbuf, uint32_t *__restrict__ counts, uint32_t cap,
    u64 *__restrict__ gsum, unsigned long long *__restrict__ gcnt, int32_t *__restrict__ err, int period_knob, int vop_rt, int xf_rt,
    int hash_bits, int strict /* dense mode without a fallback: no heavy-hitter cache, a full slab or ring reports kErrOverflow
                                (the statistics pass: the global table cannot take single rows) */,
    int64_t rot = 0 /* > 0: the 64 sixteen-lane groups of the workgroup read their 64 rows of 64 DIFFERENT batches, group g those of batch
                       + g * rot (mod the full batches): for key columns whose neighbouring rows share a bucket but that are no case for the
                       window path -- sorted files one behind the other -- a batch then reaches 64 buckets instead of one (see
                       fgb_cluster_test_kernel).  Every row is still read exactly once: for a fixed group the map is a rotation of the batches. */)
{
    constexpr bool HASH = MODE == 2, C6 = FMT == 1, K2 = FMT == 2, C8 = FMT == 3;
    const int vop = MODE == 0 ? (int)VOP_F32SUM : vop_rt;
    const int xf = MODE == 0 ? 0 : xf_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    static_assert(!((C6 || K2 || C8) && MODE == 2), "compact pairs carry bucket-local keys: dense mode only");
    uint2 *q8e = reinterpret_cast<uint2 *>(lds_raw);                   // [P][kQ8e] (value, bucket-local key)   FMT 3
    uint2 *queue = reinterpret_cast<uint2 *>(lds_raw);                 // [P][q8]           8-byte pairs
    const int q8 = P > kMaxBuckets ? kQ / 2 : kQ;                      // ring capacity of the 8-byte format: 64 pairs, 32 with 257..512 buckets (hash mode)
    uint32_t *qv = reinterpret_cast<uint32_t *>(lds_raw);              // [P][kQ6] values   compact format
    uint16_t *qk = reinterpret_cast<uint16_t *>(qv + (size_t)P * kQ6); // [P][kQ6] bucket-local keys
    uint16_t *qk2 = reinterpret_cast<uint16_t *>(lds_raw);            // [P][kQ2] bucket-local keys   keys-only format
    uint32_t *s_w = C6 ? reinterpret_cast<uint32_t *>(qk + (size_t)P * kQ6)
                  : C8 ? reinterpret_cast<uint32_t *>(q8e + (size_t)P * kQ8e)
                  : K2 ? reinterpret_cast<uint32_t *>(qk2 + (size_t)P * kQ2)
                       : reinterpret_cast<uint32_t *>(queue + (size_t)P * q8);   // [P] ring index of the oldest pair << 16 | pairs queued
    int *s_lcur = reinterpret_cast<int *>(s_w + P);                    // [P] lines already stored in this workgroup's slab
    u64 *h_val = reinterpret_cast<u64 *>(s_lcur + P + ((2 * P) & 1)); // [kHot] heavy-hitter partial values (8-byte aligned)
    uint32_t *h_key = reinterpret_cast<uint32_t *>(h_val + kHot);     // [kHot] owning key or kHotEmpty
    uint32_t *h_cnt = h_key + kHot;                                    // [kHot]
    uint32_t *h_stat = h_cnt + kHot;                                   // [0] hits, [1] surviving rows seen, [4..6] wg_or slots
    const int tid = threadIdx.x;
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int64_t nbatch = (row1 - row0 + kBatchRows - 1) / kBatchRows;
    const int cap_lines = (int)(cap / kLine) - 1;                      // the last line is kept for the final partial flush
    const int cap_units = (int)((size_t)cap * 8 / (K2 ? kUnit2Bytes : kUnitBytes)) - 1;   // compact formats: same slab bytes, the last unit for the final partial flush
    unsigned char *slab6 = reinterpret_cast<unsigned char *>(pbuf) + (size_t)wg * ((size_t)cap * 8);    // + b * nwg * cap * 8
    const uint32_t kmask = (1u << shift) - 1u;
    auto wrap6 = [](int x) { return x >= kQ6 ? x - kQ6 : x; };
    auto wrap8 = [](int x) { return x >= kQ8e ? x - kQ8e : x; };
    int period = period_knob > 0 ? period_knob : C8 ? kFlushPeriod8e : ((C6 || K2) ? kFlushPeriod6 : HASH ? 2 : kFlushPeriod);   // hash mode: every row is enqueued (16 or 8 per bucket and batch)
    const uint32_t Gu = (uint32_t)G;                                   // G <= 2^31: one unsigned compare rejects negative keys too
    bool bad = false, overflow = false;
    for (int b = tid; b < P; b += kPartThreads) { s_w[b] = 0u; s_lcur[b] = 0; }
    for (int h = tid; h < kHot; h += kPartThreads) { h_val[h] = vop_identity(vop); h_key[h] = kHotEmpty; h_cnt[h] = 0u; }
    if (tid < 8) h_stat[tid] = 0u;
    uint32_t *or_flags = h_stat + 4;
    BatchSeq seq;
    seq.init(h_stat + 8, err, nbatch, wg, nwg, tid);
    int or_phase = 0;
    int batches_done = 0, since_sweep = 0, n_full = 0;
    bool hot_on = !HASH && !strict;                                    // workgroup-uniform; switched off after the probe unless keys repeat
    __syncthreads();

    // (hash mode partitions the RAW value bits: the pairs then serve every aggregate of the column, whatever its order
    // transform -- the hash consumers apply it)
    auto vbits_of = [&](float x) -> uint32_t { return (MODE == 0 || HASH) ? __float_as_uint(x) : apply_xf(xf, __float_as_uint(x)); };
    // (a call, not inlined: inlined, the compiler hoisted the fallback's f32->f64 conversions and 64-bit address
    // arithmetic of all four rows out of the retry loop into the hot path of every batch)
    auto direct = [&](uint32_t key, uint32_t vb) { fgb_direct_row(gsum, gcnt, key, vb, K2 ? -1 : (MODE == 0 ? (int)VOP_F32SUM : vop)); };

    auto load = [&](int64_t batch, float4 &pr, int4 &kr, float4 &vr) {
        const int64_t r = row0 + batch * kBatchRows + (int64_t)tid * kVec;
        if (r + kVec <= row1) {                                        // streamed once: non-temporal loads
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef int i4v __attribute__((ext_vector_type(4)));
            if (OP == kMaskPred) {                                     // r is a multiple of 4: one byte holds the lane's four bits,
                const uint32_t byte = reinterpret_cast<const uint8_t *>(p)[r >> 3];      // which travel as an integer in pr.x
                pr = float4{__uint_as_float((byte >> (r & 4)) & 15u), 0, 0, 0};
            }
            else if (OP != kNoPred) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p + r)); pr = float4{t.x, t.y, t.z, t.w}; }
            else pr = float4{0, 0, 0, 0};
            const i4v tk = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(k + r)); kr = int4{tk.x, tk.y, tk.z, tk.w};
            if (K2) vr = float4{0, 0, 0, 0};                                 // COUNT only: the value column is not read
            else { const f4v tv = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(v + r)); vr = float4{tv.x, tv.y, tv.z, tv.w}; }
        } else {                                                       // ragged end of the table
            float pp[4] = {0, 0, 0, 0}; int kk[4] = {0, 0, 0, 0}; float vv[4] = {0, 0, 0, 0};
            for (int j = 0; j < kVec; j++) if (r + j < row1) {
                pp[j] = OP == kNoPred ? 0.0f : OP == kMaskPred ? mask_bit(p, r + j) : p[r + j]; kk[j] = k[r + j]; vv[j] = K2 ? 0.0f : v[r + j];
            }
            if (OP == kMaskPred) pp[0] = __uint_as_float((uint32_t)(pp[0] != 0.0f) | ((uint32_t)(pp[1] != 0.0f) << 1) | ((uint32_t)(pp[2] != 0.0f) << 2) | ((uint32_t)(pp[3] != 0.0f) << 3));
            pr = float4{pp[0], pp[1], pp[2], pp[3]}; kr = int4{kk[0], kk[1], kk[2], kk[3]}; vr = float4{vv[0], vv[1], vv[2], vv[3]};
        }
    };

    auto process = [&](int64_t batch, const float4 pr, const int4 kr, const float4 vr, const bool flush_now) {
        const float pv[4] = {pr.x, pr.y, pr.z, pr.w};
        const int kv[4] = {kr.x, kr.y, kr.z, kr.w};
        const float vv[4] = {vr.x, vr.y, vr.z, vr.w};
        const int64_t bend = row0 + (batch + 1) * kBatchRows;          // workgroup-uniform
        uint32_t pending = 0;
        if (OP == kMaskPred) {                                           // the survivor bits themselves (rows past the end are 0 bits)
            pending = __float_as_uint(pv[0]);
        } else if (bend <= row1) {
#pragma unroll
            for (int j = 0; j < kVec; j++) if (cmp_f32<OP>(pv[j], thr)) pending |= 1u << j;
        } else {
            const int64_t r = row0 + batch * kBatchRows + (int64_t)tid * kVec;
#pragma unroll
            for (int j = 0; j < kVec; j++) if (r + j < row1 && cmp_f32<OP>(pv[j], thr)) pending |= 1u << j;
        }
        if (!HASH) {
            uint32_t inr = 0;
#pragma unroll
            for (int j = 0; j < kVec; j++) inr |= (uint32_t)((uint32_t)kv[j] < Gu) << j;
            bad |= (pending & ~inr) != 0u;
            pending &= inr;
        }
        // ---- heavy hitters: rows whose key owns its cache slot are folded in LDS right here
        if (!HASH && hot_on) {
            const bool probing = batches_done < kHotProbeBatches;
            uint32_t seen = __popc(pending), hits = 0;
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                if (pending & (1u << j)) {
                    const uint32_t key = (uint32_t)kv[j], h = (key * 0x9E3779B1u) >> (32 - kHotBits);
                    uint32_t owner = h_key[h];
                    bool claimed = false;
                    if (owner == kHotEmpty) { owner = atomicCAS(&h_key[h], kHotEmpty, key); if (owner == kHotEmpty) { owner = key; claimed = true; } }
                    if (owner == key) {
                        if (!K2) vop_atomic_rt(vop, &h_val[h], vbits_of(vv[j]));
                        atomicAdd(&h_cnt[h], 1u);
                        pending &= ~(1u << j);
                        hits += claimed ? 0u : 1u;                      // a claim is not evidence of skew, a repeat is
                    }
                }
            }
            if (probing) {                                                            // wave-aggregated statistics of the probe phase
                for (int d = 32; d > 0; d >>= 1) { seen += __shfl_down(seen, d, 64); hits += __shfl_down(hits, d, 64); }
                if ((tid & 63) == 0) { atomicAdd(&h_stat[0], hits); atomicAdd(&h_stat[1], seen); }
            }
        }
        batches_done++;
        bool again;
        int rounds = 0;
        do {
            // ---- enqueue: one returning LDS atomic per surviving row gives it the queue position and the head
            // the four returning atomics are issued back to back (one LDS round trip per batch instead of four)
            uint32_t olds[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                olds[j] = 0u;
                if (pending & (1u << j)) {
                    const uint32_t key = (uint32_t)kv[j], b = HASH ? mix32(key) >> (32 - hash_bits) : key >> shift;
                    olds[j] = atomicAdd(&s_w[b], 1u);
                }
            }
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                if (pending & (1u << j)) {
                    const uint32_t key = (uint32_t)kv[j], b = HASH ? mix32(key) >> (32 - hash_bits) : key >> shift;
                    const uint32_t old = olds[j], pos = old & 0xFFFFu;
                    if (C6) {
                        if (pos < (uint32_t)kQ6) {
                            const int at = (int)b * kQ6 + wrap6((int)(old >> 16) + (int)pos);
                            qv[at] = vbits_of(vv[j]); qk[at] = (uint16_t)(key & kmask);
                            pending &= ~(1u << j);
                        } else atomicSub(&s_w[b], 1u);
                    } else if (C8) {
                        if (pos < (uint32_t)kQ8e) {
                            q8e[(int)b * kQ8e + wrap8((int)(old >> 16) + (int)pos)] = uint2{vbits_of(vv[j]), key & kmask};   // one ds_write_b64
                            pending &= ~(1u << j);
                        } else atomicSub(&s_w[b], 1u);
                    } else if (K2) {
                        if (pos < (uint32_t)kQ2) {
                            qk2[b * kQ2 + (((old >> 16) + pos) & (kQ2 - 1))] = (uint16_t)(key & kmask);
                            pending &= ~(1u << j);
                        } else atomicSub(&s_w[b], 1u);
                    } else
                    if (pos < (uint32_t)q8) {
                        queue[b * q8 + (((old >> 16) + pos) & (q8 - 1))] = uint2{HASH ? mix32(key) : key, vbits_of(vv[j])};   // hash mode: the MIXED key travels (a bijection: the consumers
                                                                                                                              // need only it, and unmix32 at emit time)
                        pending &= ~(1u << j);
                    } else atomicSub(&s_w[b], 1u);                      // queue full: retry after the flush
                }
            }
            // one barrier orders the enqueues before the sweep and tells whether any queue was full
            const bool full = wg_or(pending != 0, or_flags, or_phase);
            if (C8 && full && ++n_full * 8 > batches_done && period > 1) { period--; n_full = 0; }
            if ((C6 || K2) && full && ++n_full * 8 > batches_done) period = 1;    // workgroup-uniform: rings overflow in more than 1/8 of the batches (32 pairs
                                                                          // of headroom are too few for this selectivity / skew): sweep every batch
            if (!(full || flush_now)) break;
            // ---- flush (compact): 8 lanes per bucket store its complete unit, 3 x 16 bytes per lane
            if (C6) {
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                for (int b = tid >> 3; b < P; b += kPartThreads / 8) {
                    const uint32_t w = s_w[b];
                    const int cnt = (int)(w & 0xFFFFu);
                    if (cnt >= kU) {
                        const int i = tid & 7, head = (int)(w >> 16), lc = s_lcur[b];
                        const int iv0 = wrap6(head + 4 * i), iv1 = wrap6(head + 32 + 4 * i), ik = wrap6(head + 8 * i);
                        const uint4 v0 = *reinterpret_cast<const uint4 *>(&qv[b * kQ6 + iv0]);
                        const uint4 v1 = *reinterpret_cast<const uint4 *>(&qv[b * kQ6 + iv1]);
                        const uint4 k0 = *reinterpret_cast<const uint4 *>(&qk[b * kQ6 + ik]);
                        if (lc < cap_units) {
                            unsigned char *dst = slab6 + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)lc * kUnitBytes;
                            __builtin_nontemporal_store(u4v{v0.x, v0.y, v0.z, v0.w}, reinterpret_cast<u4v *>(dst + 16 * i));
                            __builtin_nontemporal_store(u4v{v1.x, v1.y, v1.z, v1.w}, reinterpret_cast<u4v *>(dst + 128 + 16 * i));
                            __builtin_nontemporal_store(u4v{k0.x, k0.y, k0.z, k0.w}, reinterpret_cast<u4v *>(dst + 256 + 16 * i));
                        } else if (strict) {
                            overflow = true;
                        } else {                                           // slab full: direct atomics
                            const uint32_t kb = (uint32_t)b << shift;
                            const uint32_t va[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                            for (int q = 0; q < 8; q++) direct(kb | qk[b * kQ6 + (q < 4 ? iv0 + q : iv1 + q - 4)], va[q]);
                        }
                        if (i == 0) {
                            s_w[b] = ((uint32_t)wrap6(head + kU) << 16) | (uint32_t)(cnt - kU);
                            s_lcur[b] = min(lc + 1, cap_units);
                        }
                    }
                }
            } else if (C8) {
                // ---- flush (8-byte ring entries -> the same 384-byte unit): lane i of 8 takes entries 4i..4i+3 and 32+4i..35+4i
                // (four ds_read_b128), stores their values as the 16-byte pieces i of the unit's two value lines and their
                // keys as the 8-byte pieces i and 8 + i of the key line
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                for (int b = tid >> 3; b < P; b += kPartThreads / 8) {
                    const uint32_t w = s_w[b];
                    int cnt = (int)(w & 0xFFFFu), head = (int)(w >> 16), lc = s_lcur[b];        // head and kQ8e are multiples of 16
                    if (cnt < kU) continue;
                    const int i = tid & 7;
                    // a ring holds up to two complete units (144 entries): both leave now, so that fewer than a unit stays behind
                    // (the final partial flush relies on it)
                    for (; cnt >= kU; cnt -= kU, head = wrap8(head + kU), lc = min(lc + 1, cap_units)) {
                        const uint2 *r0 = &q8e[b * kQ8e + wrap8(head + 4 * i)], *r1 = &q8e[b * kQ8e + wrap8(head + 32 + 4 * i)];
                        const uint4 a0 = *reinterpret_cast<const uint4 *>(r0), a1 = *reinterpret_cast<const uint4 *>(r0 + 2);
                        const uint4 c0 = *reinterpret_cast<const uint4 *>(r1), c1 = *reinterpret_cast<const uint4 *>(r1 + 2);
                        const uint32_t ka = a0.y | (a0.w << 16), kb2 = a1.y | (a1.w << 16), kc = c0.y | (c0.w << 16), kd = c1.y | (c1.w << 16);
                        if (lc < cap_units) {
                            unsigned char *dst = slab6 + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)lc * kUnitBytes;
                            __builtin_nontemporal_store(u4v{a0.x, a0.z, a1.x, a1.z}, reinterpret_cast<u4v *>(dst + 16 * i));
                            __builtin_nontemporal_store(u4v{c0.x, c0.z, c1.x, c1.z}, reinterpret_cast<u4v *>(dst + 128 + 16 * i));
                            __builtin_nontemporal_store(u2v{ka, kb2}, reinterpret_cast<u2v *>(dst + 256 + 8 * i));
                            __builtin_nontemporal_store(u2v{kc, kd}, reinterpret_cast<u2v *>(dst + 320 + 8 * i));
                        } else if (strict) {
                            overflow = true;
                        } else {                                           // slab full: direct atomics
                            const uint32_t kb = (uint32_t)b << shift;
                            direct(kb | a0.y, a0.x); direct(kb | a0.w, a0.z); direct(kb | a1.y, a1.x); direct(kb | a1.w, a1.z);
                            direct(kb | c0.y, c0.x); direct(kb | c0.w, c0.z); direct(kb | c1.y, c1.x); direct(kb | c1.w, c1.z);
                        }
                    }
                    // (the 8 lanes of a bucket run in lockstep inside one wave: lane 0's update follows every lane's ring reads)
                    if (i == 0) { s_w[b] = ((uint32_t)head << 16) | (uint32_t)cnt; s_lcur[b] = lc; }
                }
            } else if (K2) {
                // ---- flush (keys only): 8 lanes per bucket store its complete unit of 64 keys, one 128-byte line
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                for (int b = tid >> 3; b < P; b += kPartThreads / 8) {
                    const uint32_t w = s_w[b];
                    const int cnt = (int)(w & 0xFFFFu);
                    if (cnt >= kU) {
                        const int i = tid & 7, head = (int)(w >> 16), lc = s_lcur[b];
                        const int ik = (head + 8 * i) & (kQ2 - 1);
                        const uint4 k0 = *reinterpret_cast<const uint4 *>(&qk2[b * kQ2 + ik]);
                        if (lc < cap_units) {
                            unsigned char *dst = slab6 + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)lc * kUnit2Bytes;
                            __builtin_nontemporal_store(u4v{k0.x, k0.y, k0.z, k0.w}, reinterpret_cast<u4v *>(dst + 16 * i));
                        } else {                                           // slab full: direct atomics
                            const uint32_t kb = (uint32_t)b << shift;
#pragma unroll
                            for (int q = 0; q < 8; q++) direct(kb | qk2[b * kQ2 + ik + q], 0u);
                        }
                        if (i == 0) {
                            s_w[b] = ((uint32_t)((head + kU) & (kQ2 - 1)) << 16) | (uint32_t)(cnt - kU);
                            s_lcur[b] = min(lc + 1, cap_units);
                        }
                    }
                }
            } else
            // ---- flush: 8 lanes per bucket store its complete lines, 16 bytes per lane
            for (int b = tid >> 3; b < P; b += kPartThreads / 8) {
                const uint32_t w = s_w[b];
                const int cnt = (int)(w & 0xFFFFu), lines = cnt / kLine;
                if (lines) {
                    const int i = tid & 7, head = (int)(w >> 16), lc = s_lcur[b];
                    for (int q = 0; q < lines; q++) {
                        const uint4 two = *reinterpret_cast<const uint4 *>(&queue[b * q8 + ((head + q * kLine) & (q8 - 1)) + 2 * i]);
                        if (lc + q < cap_lines) {
                            uint4 *dst = reinterpret_cast<uint4 *>(&pbuf[((size_t)b * nwg + wg) * cap + (size_t)(lc + q) * kLine + 2 * i]);
                            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(u4v{two.x, two.y, two.z, two.w}, reinterpret_cast<u4v *>(dst));
                        } else if (HASH) {                                 // slab full, no dense table to fall back to
                            overflow = true;
                        } else {                                           // slab full: direct atomics
                            direct(two.x, two.y); direct(two.z, two.w);
                        }
                    }
                    if (i == 0) {
                        s_w[b] = ((uint32_t)((head + lines * kLine) & (q8 - 1)) << 16) | (uint32_t)(cnt - lines * kLine);
                        s_lcur[b] = min(lc + lines, cap_lines);
                    }
                }
            }
            since_sweep = 0;
            if (++rounds >= ((HASH || strict) ? kRetryRoundsHash : kRetryRounds)) {   // bounded: leftovers go through direct atomics
                if ((HASH || strict) && pending) { overflow = true; pending = 0; }
#pragma unroll
                for (int j = 0; j < kVec; j++)
                    if (pending & (1u << j)) direct((uint32_t)kv[j], vbits_of(vv[j]));
                pending = 0;
            }
            again = wg_or(pending != 0, or_flags, or_phase);
        } while (again);
        if (!HASH && batches_done == kHotProbeBatches) {
            if (h_stat[0] * 16u < h_stat[1]) {                            // workgroup-uniform: keys are not skewed
                // stop probing, and hand the few rows the cache absorbed to the global table now, while
                // the rest of the chip is still streaming, instead of at the tail of the kernel
                __syncthreads();
                for (int h = tid; h < kHot; h += kPartThreads) {
                    const uint32_t c = h_cnt[h];
                    if (c) { if (!K2) vop_atomic_partial_rt(vop, &gsum[h_key[h]], h_val[h]); atomicAdd(&gcnt[h_key[h]], (unsigned long long)c); h_cnt[h] = 0u; }
                }
                hot_on = false;
                __syncthreads();
            }
        }
    };

    // two batches of loads stay in flight per lane while a batch is enqueued and flushed.  The loop only ever sees FULL batches,
    // whose loads are three (two, one) unconditional vector loads per lane: the ragged last batch of a chunk -- conditional
    // scalar loads, a number the compiler cannot count -- is peeled out of the loop (inside it, the merge of the two paths made the
    // compiler drain vmcnt right after the prefetch in the no-predicate instantiation: nothing in flight while a batch was processed).
    const int64_t nfullb = (row1 - row0) / kBatchRows;                 // batches 0 .. nfullb - 1 are full; batch nfullb, if it exists, is the ragged one
    auto load_full = [&](int64_t batch, float4 &pr, int4 &kr, float4 &vr) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef int i4v __attribute__((ext_vector_type(4)));
        if constexpr (ROT) { batch += (int64_t)(tid >> 4) * rot; if (batch >= nfullb) batch -= nfullb; }   // (full batches only: see `rot`)
        const int64_t r = row0 + batch * kBatchRows + (int64_t)tid * kVec;
        if (OP == kMaskPred) {
            const uint32_t byte = reinterpret_cast<const uint8_t *>(p)[r >> 3];
            pr = float4{__uint_as_float((byte >> (r & 4)) & 15u), 0, 0, 0};
        }
        else if (OP != kNoPred) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p + r)); pr = float4{t.x, t.y, t.z, t.w}; }
        else pr = float4{0, 0, 0, 0};
        const i4v tk = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(k + r)); kr = int4{tk.x, tk.y, tk.z, tk.w};
        if (K2) vr = float4{0, 0, 0, 0};
        else { const f4v tv = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(v + r)); vr = float4{tv.x, tv.y, tv.z, tv.w}; }
    };
    float4 pA, vA, pB, vB; int4 kA, kB;
    int64_t idA = wg, idB = (int64_t)wg + nwg;                        // the batches whose rows sit in the A and B registers
    if (idA < nfullb) load_full(idA, pA, kA, vA);
    if (idB < nfullb) load_full(idB, pB, kB, vB);
    for (int j = 0; idA < nfullb; j++) {
        const int64_t batch = idA;
        const float4 pr = pA, vr = vA; const int4 kr = kA;
        pA = pB; vA = vB; kA = kB;
        idA = idB;
        idB = seq.to_load(j);                                           // parked at least one barrier ago
        if (tid == 0) seq.draw(j);
        if (idB < nfullb) load_full(idB, pB, kB, vB);
        process(batch, pr, kr, vr, ++since_sweep >= period || idA >= nbatch);   // sweep every period-th batch and on the workgroup's last one
    }
    if (idA < nbatch) {                                                 // the ragged batch (ids ascend: it is this workgroup's last)
        load(idA, pA, kA, vA);
        process(idA, pA, kA, vA, true);
    }
    // ---- final flush: what is left (< kLine pairs per bucket) goes out as one partial line
    uint32_t my_pairs = 0;                                              // pairs this workgroup partitioned: the host learns the predicate's selectivity from the sum
    for (int b = tid; b < P; b += kPartThreads) {
        const uint32_t w = s_w[b];
        const int l = (int)(w & 0xFFFFu), head = (int)(w >> 16);
        my_pairs += (uint32_t)(s_lcur[b] * ((C6 || C8 || K2) ? kU : kLine) + l);
        if (C6) {                                                       // l < kU: the last sweep took every complete unit
            unsigned char *dst = slab6 + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)s_lcur[b] * kUnitBytes;
            for (int j = 0; j < l; j++) {
                const int at = b * kQ6 + wrap6(head + j);
                reinterpret_cast<uint32_t *>(dst)[j] = qv[at]; reinterpret_cast<uint16_t *>(dst + 4 * kU)[j] = qk[at];
            }
            counts[(size_t)b * nwg + wg] = (uint32_t)(s_lcur[b] * kU + l);
            continue;
        }
        if (C8) {                                                       // l < kU (see above)
            unsigned char *dst = slab6 + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)s_lcur[b] * kUnitBytes;
            for (int j = 0; j < l; j++) {
                const uint2 e = q8e[b * kQ8e + wrap8(head + j)];
                reinterpret_cast<uint32_t *>(dst)[j] = e.x; reinterpret_cast<uint16_t *>(dst + 4 * kU)[j] = (uint16_t)e.y;
            }
            counts[(size_t)b * nwg + wg] = (uint32_t)(s_lcur[b] * kU + l);
            continue;
        }
        if (K2) {
            unsigned char *dst = slab6 + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)s_lcur[b] * kUnit2Bytes;
            for (int j = 0; j < l; j++) reinterpret_cast<uint16_t *>(dst)[j] = qk2[b * kQ2 + ((head + j) & (kQ2 - 1))];
            counts[(size_t)b * nwg + wg] = (uint32_t)(s_lcur[b] * kU + l);
            continue;
        }
        const size_t base = ((size_t)b * nwg + wg) * cap + (size_t)s_lcur[b] * kLine;
        for (int j = 0; j < l; j++) pbuf[base + j] = queue[b * q8 + ((head + j) & (q8 - 1))];
        counts[(size_t)b * nwg + wg] = (uint32_t)(s_lcur[b] * kLine + l);
    }
    // ---- the heavy hitters join the global table (one atomic pair per occupied entry)
    if (!HASH)
    for (int h = tid; h < kHot; h += kPartThreads) {
        const uint32_t c = h_cnt[h];
        if (c) { if (!K2) vop_atomic_partial_rt(vop, &gsum[h_key[h]], h_val[h]); atomicAdd(&gcnt[h_key[h]], (unsigned long long)c); }
    }
    if (bad) *err = HARK_EBOUNDS;
    if (overflow) *err = kErrOverflow;
    if (!HASH) {
        for (int d = 32; d > 0; d >>= 1) my_pairs += __shfl_down(my_pairs, d, 64);
        if ((tid & 63) == 0 && my_pairs) atomicAdd(reinterpret_cast<unsigned long long *>(err + 2), (unsigned long long)my_pairs);
    }
}

// ---------------------------------------------------------------------------
// Partition path, CONSUMER: one workgroup folds one bucket into its table slice
// ---------------------------------------------------------------------------
// LDS: u64 s_sum[KPB] (8-byte value slots); uint32 s_cnt[KPB], KPB = 1 << shift keys per bucket.
template <int VOP>
__global__ __launch_bounds__(1024) void fgb_agg_kernel(
    const uint2 *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, int shift,
    int64_t G, u64 *__restrict__ gsum, unsigned long long *__restrict__ gcnt)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    auto ld = [&](const uint4 *q) -> uint4 {
        const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(q));   // pairs are read exactly once
        return uint4{t.x, t.y, t.z, t.w};
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int KPB = 1 << shift;
    u64 *s_sum = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * KPB);
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) { s_sum[i] = vop_identity(VOP); s_cnt[i] = 0u; }
    __syncthreads();
    const uint32_t mask = (uint32_t)KPB - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    auto add = [&](uint32_t key, uint32_t vbits) {
        vop_atomic<VOP>(&s_sum[key & mask], vbits);
        atomicAdd(&s_cnt[key & mask], 1u);
    };
    // each wave walks whole slabs: 2 pairs (16 bytes) per lane per load, 2 loads in flight
    for (int w = wave; w < nwg; w += nwaves) {
        const uint32_t count = min(counts[(size_t)b * nwg + w], cap);
        const uint2 *src = pbuf + ((size_t)b * nwg + w) * cap;         // cap is even -> 16-byte aligned
        const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
        const uint32_t n2 = count / 2;
        uint32_t i = lane;
        for (; i + 64 < n2; i += 128) {
            const uint4 q0 = ld(src4 + i), q1 = ld(src4 + i + 64);
            add(q0.x, q0.y); add(q0.z, q0.w); add(q1.x, q1.y); add(q1.z, q1.w);
        }
        for (; i < n2; i += 64) { const uint4 q = ld(src4 + i); add(q.x, q.y); add(q.z, q.w); }
        if ((count & 1u) && lane == 0) { const uint2 q = src[count - 1]; add(q.x, q.y); }
    }
    __syncthreads();
    const int64_t kbase = (int64_t)b << shift;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) {
        const uint32_t c = s_cnt[i];
        if (c && kbase + i < G) {                           // this workgroup owns [kbase, kbase+KPB)
            gsum[kbase + i] = vop_merge(VOP, gsum[kbase + i], s_sum[i]);
            gcnt[kbase + i] += (unsigned long long)c;
        }
    }
}

// Consumer of the compact format: units of 64 pairs (256 B of values, 128 B of 16-bit local keys).
// A 16-lane group takes a unit: 16 bytes of values + 8 bytes of keys per lane, two units in flight.
// split = 1: workgroup b owns bucket b and its key range (plain read-modify-write of the global table); split = 2 (at most
// 128 buckets): workgroups 2b and 2b + 1 share bucket b -- even / odd slabs -- so that all 256 CUs work, and merge with
// contiguous global atomics.
template <int VOP>
__global__ __launch_bounds__(1024) void fgb_agg6_kernel(
    const unsigned char *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, int shift,
    int64_t G, u64 *__restrict__ gsum, unsigned long long *__restrict__ gcnt, int split)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int KPB = 1 << shift;
    u64 *s_sum = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * KPB);
    const int b = split > 1 ? (int)blockIdx.x / split : (int)blockIdx.x, half = split > 1 ? (int)blockIdx.x % split : 0;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) { s_sum[i] = vop_identity(VOP); s_cnt[i] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t max_pairs = (uint32_t)((size_t)cap * 8 / kUnitBytes) * kU;
    auto add = [&](uint32_t key, uint32_t vbits) {
        vop_atomic<VOP>(&s_sum[key], vbits);
        atomicAdd(&s_cnt[key], 1u);
    };
    auto add4 = [&](const u4v v, const u2v kk) {
        add(kk.x & 0xFFFFu, v.x); add(kk.x >> 16, v.y); add(kk.y & 0xFFFFu, v.z); add(kk.y >> 16, v.w);
    };
    const int piece = lane & 15, sub = lane >> 4;
    for (int w = wave * split + half; w < nwg; w += nwaves * split) {
        const uint32_t count = min(counts[(size_t)b * nwg + w], max_pairs);
        const unsigned char *src = pbuf + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)w * ((size_t)cap * 8);
        const uint32_t units = count / kU, rem = count % kU;
        uint32_t u = sub;
        for (; u + 4 < units; u += 8) {
            const unsigned char *a = src + (size_t)u * kUnitBytes, *c = a + 4 * kUnitBytes;
            const u4v va = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + 16 * piece));
            const u2v ka = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(a + 4 * kU + 8 * piece));
            const u4v vc = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(c + 16 * piece));
            const u2v kc = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(c + 4 * kU + 8 * piece));
            add4(va, ka); add4(vc, kc);
        }
        for (; u < units; u += 4) {
            const unsigned char *a = src + (size_t)u * kUnitBytes;
            const u4v va = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + 16 * piece));
            const u2v ka = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(a + 4 * kU + 8 * piece));
            add4(va, ka);
        }
        if ((uint32_t)lane < rem) {
            const unsigned char *a = src + (size_t)units * kUnitBytes;
            add(reinterpret_cast<const uint16_t *>(a + 4 * kU)[lane], reinterpret_cast<const uint32_t *>(a)[lane]);
        }
    }
    __syncthreads();
    const int64_t kbase = (int64_t)b << shift;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) {
        const uint32_t c = s_cnt[i];
        if (c && kbase + i < G) {
            if (split > 1) {                                // two workgroups share [kbase, kbase+KPB)
                vop_atomic_partial<VOP>(&gsum[kbase + i], s_sum[i]);
                atomicAdd(&gcnt[kbase + i], (unsigned long long)c);
            } else {                                        // this workgroup owns [kbase, kbase+KPB)
                gsum[kbase + i] = vop_merge(VOP, gsum[kbase + i], s_sum[i]);
                gcnt[kbase + i] += (unsigned long long)c;
            }
        }
    }
}

// Consumer of the keys-only format (COUNT): units of 64 bucket-local 16-bit keys = one 128-byte line; an 8-lane group
// takes a unit (16 bytes = 8 keys per lane), two units in flight per group.
__global__ __launch_bounds__(1024) void fgb_agg2_kernel(
    const unsigned char *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, int shift,
    int64_t G, unsigned long long *__restrict__ gcnt)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int KPB = 1 << shift;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw);
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) s_cnt[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t max_keys = (uint32_t)((size_t)cap * 8 / kUnit2Bytes) * kU;
    auto add8 = [&](const u4v q) {
        atomicAdd(&s_cnt[q.x & 0xFFFFu], 1u); atomicAdd(&s_cnt[q.x >> 16], 1u);
        atomicAdd(&s_cnt[q.y & 0xFFFFu], 1u); atomicAdd(&s_cnt[q.y >> 16], 1u);
        atomicAdd(&s_cnt[q.z & 0xFFFFu], 1u); atomicAdd(&s_cnt[q.z >> 16], 1u);
        atomicAdd(&s_cnt[q.w & 0xFFFFu], 1u); atomicAdd(&s_cnt[q.w >> 16], 1u);
    };
    const int piece = lane & 7, sub = lane >> 3;
    for (int w = wave; w < nwg; w += nwaves) {
        const uint32_t count = min(counts[(size_t)b * nwg + w], max_keys);
        const unsigned char *src = pbuf + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)w * ((size_t)cap * 8);
        const uint32_t units = count / kU, rem = count % kU;
        uint32_t u = sub;
        for (; u + 8 < units; u += 16) {
            const u4v a = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(src + (size_t)u * kUnit2Bytes + 16 * piece));
            const u4v c = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(src + (size_t)(u + 8) * kUnit2Bytes + 16 * piece));
            add8(a); add8(c);
        }
        for (; u < units; u += 8) add8(__builtin_nontemporal_load(reinterpret_cast<const u4v *>(src + (size_t)u * kUnit2Bytes + 16 * piece)));
        if ((uint32_t)lane < rem) atomicAdd(&s_cnt[reinterpret_cast<const uint16_t *>(src + (size_t)units * kUnit2Bytes)[lane]], 1u);
    }
    __syncthreads();
    const int64_t kbase = (int64_t)b << shift;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) {
        const uint32_t c = s_cnt[i];
        if (c && kbase + i < G) gcnt[kbase + i] += (unsigned long long)c;     // this workgroup owns [kbase, kbase+KPB)
    }
}

// Statistics consumer (SUM / COUNT / AVG + MIN + MAX of ONE value column in one pass): the pairs carry the raw value
// bits, the bucket's LDS slice holds 20 B per key -- 64-bit sum slot, row count, smallest and largest order word.
// VK: 0 = f32 values (f64 sum), 1 = i32 (sum of the biased values, like XF_I32_ORDER), 2 = u32.
template <int VK>
__global__ __launch_bounds__(1024) void fgb_agg6_stats_kernel(
    const unsigned char *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, int shift,
    int64_t G, u64 *__restrict__ gsum, unsigned long long *__restrict__ gcnt, u64 *__restrict__ gmin, u64 *__restrict__ gmax)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int KPB = 1 << shift;
    u64 *s_sum = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * KPB);
    uint32_t *s_min = s_cnt + KPB, *s_max = s_min + KPB;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) { s_sum[i] = 0ull; s_cnt[i] = 0u; s_min[i] = 0xFFFFFFFFu; s_max[i] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t max_pairs = (uint32_t)((size_t)cap * 8 / kUnitBytes) * kU;
    auto add = [&](uint32_t key, uint32_t raw) {
        uint32_t w;
        if constexpr (VK == 0) { vop_atomic<VOP_F32SUM>(&s_sum[key], raw); w = apply_xf(2, raw); }
        else if constexpr (VK == 1) { w = raw ^ 0x80000000u; atomicAdd(&s_sum[key], (u64)w); }
        else { w = raw; atomicAdd(&s_sum[key], (u64)w); }
        atomicAdd(&s_cnt[key], 1u);
        atomicMin(&s_min[key], w);
        atomicMax(&s_max[key], w);
    };
    auto add4 = [&](const u4v v, const u2v kk) {
        add(kk.x & 0xFFFFu, v.x); add(kk.x >> 16, v.y); add(kk.y & 0xFFFFu, v.z); add(kk.y >> 16, v.w);
    };
    const int piece = lane & 15, sub = lane >> 4;
    for (int w = wave; w < nwg; w += nwaves) {
        const uint32_t count = min(counts[(size_t)b * nwg + w], max_pairs);
        const unsigned char *src = pbuf + (size_t)b * ((size_t)nwg * ((size_t)cap * 8)) + (size_t)w * ((size_t)cap * 8);
        const uint32_t units = count / kU, rem = count % kU;
        for (uint32_t u = sub; u < units; u += 4) {
            const unsigned char *a = src + (size_t)u * kUnitBytes;
            const u4v va = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + 16 * piece));
            const u2v ka = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(a + 4 * kU + 8 * piece));
            add4(va, ka);
        }
        if ((uint32_t)lane < rem) {
            const unsigned char *a = src + (size_t)units * kUnitBytes;
            add(reinterpret_cast<const uint16_t *>(a + 4 * kU)[lane], reinterpret_cast<const uint32_t *>(a)[lane]);
        }
    }
    __syncthreads();
    const int64_t kbase = (int64_t)b << shift;
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) {
        const uint32_t c = s_cnt[i];
        if (c && kbase + i < G) {                           // this workgroup owns [kbase, kbase+KPB)
            gsum[kbase + i] = VK == 0 ? vop_merge(VOP_F32SUM, gsum[kbase + i], s_sum[i]) : gsum[kbase + i] + s_sum[i];
            gcnt[kbase + i] += (unsigned long long)c;
            const u64 mn = gmin[kbase + i], mx = gmax[kbase + i];
            gmin[kbase + i] = mn < (u64)s_min[i] ? mn : (u64)s_min[i];
            gmax[kbase + i] = mx > (u64)s_max[i] ? mx : (u64)s_max[i];
        }
    }
}

// ---------------------------------------------------------------------------
// Two or three aggregates of DIFFERENT columns in one pass ("pair pass", "triple pass")
// ---------------------------------------------------------------------------
// One producer + consumer pass per (operator, column) re-reads the predicate and key columns every time
// (BASELINE configs[4] with three aggregates: three passes).  Here an entry carries the raw 32 bits of NV = 2 or 3 value
// columns and a 16-bit bucket-local key: 10 or 14 bytes, units of 64 entries = NV x 256 B + 128 B (whole lines), in
// HALF as many buckets as the one-value passes use (<= 128 buckets of <= 8192 keys).  Pairs: rings of 112 entries (a unit
// + 48 of headroom) are 140 KiB of LDS, swept every other batch.  (A first version with 256 buckets, units of 32 pairs =
// 2.5 lines and 24 pairs of headroom ran at 2.44 ms per 5e8 rows against 1.47 ms for a one-value pass: twice the sweeps,
// half-line stores, rings overflowing.)  Triples: 128 rings of 14-byte entries leave 88 entries per ring (154 KiB) -- a
// unit + 24 of headroom for the ~16 arrivals of a batch at 50 % selectivity: swept after EVERY batch, and ~7 % of the
// batches see some ring overflow and take a second round.
// The consumer keeps per key a 64-bit slot for value 1 (any operator), the row count, and a 32-bit slot for each further
// value, which therefore must be a MAX or MIN (order words; a MIN is kept as the MAX of the inverted word, so the kernels
// are instantiated per operator of value 1 only) -- 16 B x 8192 keys = 128 KiB, or 20 B x 8192 = ALL 160 KiB of a CU
// for triples (a workgroup may allocate exactly that on gfx950; checked with a probe) -- and TWO workgroups share a
// bucket (even / odd slabs) so that all 256 CUs work; they merge into the global table with contiguous atomics.
// Like the statistics pass these have no single-row fallback: skew that overflows a ring or a slab reports
// kErrOverflow and the caller runs the separate passes.
constexpr int kU10 = 64;                                 // entries per unit
constexpr int kPairBuckets = 128;
template <int NV> struct MultiGeo {
    static constexpr int Q = NV == 2 ? 112 : 88;         // ring capacity per bucket (entries); multiples of 8 keep 16-byte pieces whole across the wrap
    static constexpr int entry_bytes = 4 * NV + 2;
    static constexpr int unit_bytes = kU10 * entry_bytes; // 640 / 896: value r at 256 r, the keys behind the values
    static constexpr int first_period = NV == 2 ? 2 : 1; // batches between sweeps
};
static size_t partv_lds_bytes(int P, int nv) { return (size_t)(4 * nv + 2) * P * (nv == 2 ? MultiGeo<2>::Q : MultiGeo<3>::Q) + sizeof(uint32_t) * 2 * (size_t)P + 32 + 4 * kSeqRing; }
static int multi_unit_bytes(int nv) { return kU10 * (4 * nv + 2); }

template <int OP, int NV, bool ROT = false>
__global__ __launch_bounds__(kPartThreads) void fgb_partv_kernel(
    const float *__restrict__ p, const int32_t *__restrict__ k, const uint32_t *__restrict__ v1, const uint32_t *__restrict__ v2, const uint32_t *__restrict__ v3,
    int64_t row0, int64_t row1, float thr, int64_t G, int shift, int P,
    unsigned char *__restrict__ pbuf, uint32_t *__restrict__ counts, size_t slab_bytes, int32_t *__restrict__ err, int64_t rot /* as fgb_part_kernel's */)
{
    constexpr int Q = MultiGeo<NV>::Q, kUnitBytes = MultiGeo<NV>::unit_bytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *qv = reinterpret_cast<uint32_t *>(lds_raw);                  // [NV][P][Q] raw value bits
    uint16_t *qk = reinterpret_cast<uint16_t *>(qv + (size_t)NV * P * Q);  // [P][Q] bucket-local keys
    uint32_t *s_w = reinterpret_cast<uint32_t *>(qk + (size_t)P * Q);      // [P] ring index of the oldest entry << 16 | entries queued
    int *s_lcur = reinterpret_cast<int *>(s_w + P);                        // [P] units already stored in this workgroup's slab
    uint32_t *or_flags = reinterpret_cast<uint32_t *>(s_lcur + P);         // [3] wg_or slots
    const int tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const int PQ = P * Q;
    const int64_t nbatch = (row1 - row0 + kBatchRows - 1) / kBatchRows;
    const int cap_units = (int)(slab_bytes / kUnitBytes) - 1;              // the last unit for the final partial flush
    const uint32_t kmask = (1u << shift) - 1u, Gu = (uint32_t)G;
    auto slab_of = [&](int b) -> unsigned char * { return pbuf + ((size_t)b * nwg + wg) * slab_bytes; };
    auto wrap = [](int x) { return x >= Q ? x - Q : x; };
    for (int b = tid; b < P; b += kPartThreads) { s_w[b] = 0u; s_lcur[b] = 0; }
    if (tid < 4) or_flags[tid] = 0u;
    BatchSeq seq;                                                          // batches first come, first served (see BatchSeq)
    seq.init(or_flags + 8, err, nbatch, wg, nwg, tid);
    int or_phase = 0, period = MultiGeo<NV>::first_period, since_sweep = 0, n_full = 0, batches_done = 0;
    bool bad = false, overflow = false;
    __syncthreads();

    struct Rows { float4 p; int4 k; uint4 v[NV]; };
    const int64_t nfullv = (row1 - row0) / kBatchRows;
    auto load_full = [&](int64_t batch, Rows &r) {                        // a batch of kBatchRows rows: unconditional vector loads
        if constexpr (ROT) { batch += (int64_t)(tid >> 4) * rot; if (batch >= nfullv) batch -= nfullv; }
        const int64_t rb = row0 + batch * kBatchRows;                      // workgroup-uniform (per sixteen lanes with rotated loads)
        const uint32_t lo = (uint32_t)tid * kVec;
        if (OP == kMaskPred) {
            const uint32_t byte = (reinterpret_cast<const uint8_t *>(p) + (rb >> 3))[lo >> 3];
            r.p = float4{__uint_as_float((byte >> (lo & 4u)) & 15u), 0, 0, 0};
        } else if (OP != kNoPred) { const uint4 t = ld_nt16(p + rb + lo); r.p = float4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)}; }
        else r.p = float4{0, 0, 0, 0};
        const uint4 tk = ld_nt16(k + rb + lo); r.k = int4{(int)tk.x, (int)tk.y, (int)tk.z, (int)tk.w};
        r.v[0] = ld_nt16(v1 + rb + lo); r.v[1] = ld_nt16(v2 + rb + lo);
        if constexpr (NV == 3) r.v[2] = ld_nt16(v3 + rb + lo);
    };
    auto load = [&](int64_t batch, Rows &r) {
        const int64_t rb = row0 + batch * kBatchRows;                      // workgroup-uniform
        const uint32_t lo = (uint32_t)tid * kVec;
        if (rb + kBatchRows <= row1) load_full(batch, r);
        else {                                                             // ragged end of the table
            const int64_t r0 = rb + lo;
            float pp[4] = {0, 0, 0, 0}; int kk[4] = {0, 0, 0, 0}; uint32_t vv[NV][4] = {};
            for (int j = 0; j < kVec; j++) if (r0 + j < row1) {
                pp[j] = OP == kNoPred ? 0.0f : OP == kMaskPred ? mask_bit(p, r0 + j) : p[r0 + j]; kk[j] = k[r0 + j]; vv[0][j] = v1[r0 + j]; vv[1][j] = v2[r0 + j];
                if constexpr (NV == 3) vv[2][j] = v3[r0 + j];
            }
            if (OP == kMaskPred) pp[0] = __uint_as_float((uint32_t)(pp[0] != 0.0f) | ((uint32_t)(pp[1] != 0.0f) << 1) | ((uint32_t)(pp[2] != 0.0f) << 2) | ((uint32_t)(pp[3] != 0.0f) << 3));
            r.p = float4{pp[0], pp[1], pp[2], pp[3]}; r.k = int4{kk[0], kk[1], kk[2], kk[3]};
#pragma unroll
            for (int c = 0; c < NV; c++) r.v[c] = uint4{vv[c][0], vv[c][1], vv[c][2], vv[c][3]};
        }
    };

    // every complete unit of every ring leaves for the workgroup's slabs: 8 lanes per bucket, 16 bytes per lane and piece
    auto sweep = [&]() {
        for (int b = tid >> 3; b < P; b += kPartThreads / 8) {
            const uint32_t w = s_w[b];
            const int cnt = (int)(w & 0xFFFFu);
            if (cnt >= kU10) {
                const int i = tid & 7, head = (int)(w >> 16), lc = s_lcur[b];
                const int iv0 = wrap(head + 4 * i), iv1 = wrap(head + 32 + 4 * i), ik = wrap(head + 8 * i);
                if (lc < cap_units) {
                    unsigned char *dst = slab_of(b) + (size_t)lc * kUnitBytes;
#pragma unroll
                    for (int c = 0; c < NV; c++) {                         // one value ring at a time: 8 data registers live, not 8 NV
                        const uint4 a0 = *reinterpret_cast<const uint4 *>(&qv[c * PQ + b * Q + iv0]);
                        const uint4 a1 = *reinterpret_cast<const uint4 *>(&qv[c * PQ + b * Q + iv1]);
                        st_nt16(dst + 256 * c + 16 * i, a0); st_nt16(dst + 256 * c + 128 + 16 * i, a1);
                        if constexpr (NV == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    st_nt16(dst + 256 * NV + 16 * i, *reinterpret_cast<const uint4 *>(&qk[b * Q + ik]));
                } else overflow = true;
                if (i == 0) {
                    s_w[b] = ((uint32_t)wrap(head + kU10) << 16) | (uint32_t)(cnt - kU10);
                    s_lcur[b] = min(lc + 1, cap_units);
                }
            }
        }
    };

    auto process = [&](int64_t batch, const Rows &r, const bool flush_now) {
        const float pv[4] = {r.p.x, r.p.y, r.p.z, r.p.w};
        const int kv[4] = {r.k.x, r.k.y, r.k.z, r.k.w};
        uint32_t vv[NV][4];
#pragma unroll
        for (int c = 0; c < NV; c++) { vv[c][0] = r.v[c].x; vv[c][1] = r.v[c].y; vv[c][2] = r.v[c].z; vv[c][3] = r.v[c].w; }
        const int64_t bend = row0 + (batch + 1) * kBatchRows;
        bool sv[kVec];
        if (OP == kMaskPred) {
            const uint32_t nib = __float_as_uint(pv[0]);
#pragma unroll
            for (int j = 0; j < kVec; j++) sv[j] = (nib >> j) & 1u;
        } else if (bend <= row1) {
#pragma unroll
            for (int j = 0; j < kVec; j++) sv[j] = cmp_f32<OP>(pv[j], thr);
        } else {
            const int64_t r0 = row0 + batch * kBatchRows + (int64_t)tid * kVec;
#pragma unroll
            for (int j = 0; j < kVec; j++) sv[j] = r0 + j < row1 && cmp_f32<OP>(pv[j], thr);
        }
#pragma unroll
        for (int j = 0; j < kVec; j++) { const bool inr = (uint32_t)kv[j] < Gu; bad |= sv[j] && !inr; sv[j] = sv[j] && inr; }
        batches_done++;
        bool again;
        int rounds = 0;
        do {
            uint32_t olds[kVec];                                           // the returning atomics of a lane back to back
#pragma unroll
            for (int j = 0; j < kVec; j++) { olds[j] = 0u; if (sv[j]) olds[j] = atomicAdd(&s_w[(uint32_t)kv[j] >> shift], 1u); }
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                if (sv[j]) {
                    const uint32_t key = (uint32_t)kv[j], b = key >> shift, old = olds[j], pos = old & 0xFFFFu;
                    if (pos < (uint32_t)Q) {
                        const int at = (int)__umul24(b, (uint32_t)Q) + wrap((int)(old >> 16) + (int)pos);
#pragma unroll
                        for (int c = 0; c < NV; c++) qv[c * PQ + at] = vv[c][j];
                        qk[at] = (uint16_t)(key & kmask);
                        sv[j] = false;
                    } else atomicSub(&s_w[b], 1u);                         // ring full: retry after the sweep
                }
            }
            const bool full = wg_or(sv[0] | sv[1] | sv[2] | sv[3], or_flags, or_phase);
            if (full && ++n_full * 8 > batches_done) period = 1;          // rings overflow often (selectivity / skew): sweep every batch
            if (!(full || flush_now)) break;
            sweep();
            since_sweep = 0;
            if (++rounds >= kRetryRoundsHash) { if (sv[0] | sv[1] | sv[2] | sv[3]) overflow = true; sv[0] = sv[1] = sv[2] = sv[3] = false; }
            again = wg_or(sv[0] | sv[1] | sv[2] | sv[3], or_flags, or_phase);
        } while (again);
    };

    // (as in fgb_part_kernel: the loop sees full batches only -- unconditional vector loads the compiler can count --, the ragged
    // last batch of a chunk is peeled out)
    const int64_t nfullb = (row1 - row0) / kBatchRows;
    Rows A, B;
    int64_t idA = wg, idB = (int64_t)wg + nwg;
    if (idA < nfullb) load_full(idA, A);
    if (idB < nfullb) load_full(idB, B);
    for (int j = 0; idA < nfullb; j++) {
        const int64_t batch = idA;
        const Rows cur = A;
        A = B;
        idA = idB;
        idB = seq.to_load(j);
        if (tid == 0) seq.draw(j);
        if (idB < nfullb) load_full(idB, B);
        process(batch, cur, ++since_sweep >= period || idA >= nbatch);
    }
    if (idA < nbatch) {                                                 // the ragged batch (ids ascend: it is this workgroup's last)
        load(idA, A);
        process(idA, A, true);
    }
    // what is left (< kU10 entries per bucket) goes out as one partial unit
    for (int b = tid; b < P; b += kPartThreads) {
        const uint32_t w = s_w[b];
        const int l = (int)(w & 0xFFFFu), head = (int)(w >> 16);
        unsigned char *dst = slab_of(b) + (size_t)s_lcur[b] * kUnitBytes;
        for (int j = 0; j < l; j++) {
            const int at = b * Q + wrap(head + j);
#pragma unroll
            for (int c = 0; c < NV; c++) reinterpret_cast<uint32_t *>(dst + 256 * c)[j] = qv[c * PQ + at];
            reinterpret_cast<uint16_t *>(dst + 256 * NV)[j] = qk[at];
        }
        counts[(size_t)b * nwg + wg] = (uint32_t)(s_lcur[b] * kU10 + l);
    }
    if (bad) *err = HARK_EBOUNDS;
    if (overflow) *err = kErrOverflow;
}

// Consumer of the pair / triple pass: two workgroups per bucket, LDS slice of 16 / 20 B per key (64-bit slot of value 1,
// row count, one 32-bit slot per further value).  VOP1: any of F32SUM / U32SUM64 / U32SUM / U32MAX / U32MIN (the last two
// in the low word); values 2 and 3: MAX (inv = 0) or MIN (inv = 0xFFFFFFFF: the slot keeps the MAX of the inverted order
// word).  xf1..xf3: order transforms of the raw bits (apply_xf).
template <int VOP1, int NV>
__global__ __launch_bounds__(1024) void fgb_aggv_kernel(
    const unsigned char *__restrict__ pbuf, const uint32_t *__restrict__ counts, size_t slab_bytes, int nwg, int shift,
    int64_t G, u64 *__restrict__ gsum, unsigned long long *__restrict__ gcnt, u64 *__restrict__ g2, u64 *__restrict__ g3,
    int xf1, int xf2, int xf3, uint32_t inv2, uint32_t inv3)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    constexpr int kUnitBytes = MultiGeo<NV>::unit_bytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int KPB = 1 << shift;
    u64 *s_a = reinterpret_cast<u64 *>(lds_raw);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw + sizeof(u64) * KPB);
    uint32_t *s_b = s_cnt + KPB;
    uint32_t *s_c = s_b + KPB;                                         // NV == 3 only
    const int b = blockIdx.x >> 1, half = blockIdx.x & 1;              // two workgroups per bucket: even / odd slabs
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) { s_a[i] = vop_identity(VOP1); s_cnt[i] = 0u; s_b[i] = 0u; if constexpr (NV == 3) s_c[i] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t max_entries = (uint32_t)(slab_bytes / kUnitBytes) * kU10;
    auto add = [&](uint32_t key, uint32_t ra, uint32_t rb, uint32_t rc) {
        vop_atomic<VOP1>(&s_a[key], VOP1 == VOP_F32SUM ? ra : apply_xf(xf1, ra));
        atomicAdd(&s_cnt[key], 1u);
        atomicMax(&s_b[key], apply_xf(xf2, rb) ^ inv2);
        if constexpr (NV == 3) atomicMax(&s_c[key], apply_xf(xf3, rc) ^ inv3);
    };
    const int piece = lane & 15, sub = lane >> 4;                     // 16 lanes per unit (16 B of each value, 8 B of keys per lane), 4 units per wave and step
    for (int w = 2 * wave + half; w < nwg; w += 2 * nwaves) {
        const uint32_t count = min(counts[(size_t)b * nwg + w], max_entries);
        const unsigned char *src = pbuf + ((size_t)b * nwg + w) * slab_bytes;
        const uint32_t units = count / kU10, rem = count % kU10;
        for (uint32_t u = sub; u < units; u += 4) {
            const unsigned char *a = src + (size_t)u * kUnitBytes;
            const u4v va = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + 16 * piece));
            const u4v vb = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + 256 + 16 * piece));
            u4v vc = {0u, 0u, 0u, 0u};
            if constexpr (NV == 3) vc = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + 512 + 16 * piece));
            const u2v kk = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(a + 256 * NV + 8 * piece));
            add(kk.x & 0xFFFFu, va.x, vb.x, vc.x); add(kk.x >> 16, va.y, vb.y, vc.y); add(kk.y & 0xFFFFu, va.z, vb.z, vc.z); add(kk.y >> 16, va.w, vb.w, vc.w);
        }
        if ((uint32_t)lane < rem) {
            const unsigned char *a = src + (size_t)units * kUnitBytes;
            add(reinterpret_cast<const uint16_t *>(a + 256 * NV)[lane], reinterpret_cast<const uint32_t *>(a)[lane], reinterpret_cast<const uint32_t *>(a + 256)[lane],
                NV == 3 ? reinterpret_cast<const uint32_t *>(a + 512)[lane] : 0u);
        }
    }
    __syncthreads();
    // the two workgroups of a bucket meet in the global table: contiguous atomics (one per touched key and accumulator)
    const int64_t kbase = (int64_t)b << shift;
    auto ext = [](u64 *slot, uint32_t w, uint32_t inv) {              // the order word lives in the low half of the 64-bit slot
        uint32_t *lo = reinterpret_cast<uint32_t *>(slot);
        if (inv) atomicMin(lo, w ^ inv); else atomicMax(lo, w);
    };
    for (int i = threadIdx.x; i < KPB; i += blockDim.x) {
        const uint32_t c = s_cnt[i];
        if (c && kbase + i < G) {
            vop_atomic_partial<VOP1>(&gsum[kbase + i], s_a[i]);
            atomicAdd(&gcnt[kbase + i], (unsigned long long)c);
            ext(&g2[kbase + i], s_b[i], inv2);
            if constexpr (NV == 3) ext(&g3[kbase + i], s_c[i], inv3);
        }
    }
}

// A wave's walk over its slabs of a bucket (every nwaves-th slab; 8-byte (key, value) pairs): ONE stream of steps of eight
// pairs per lane (four 16-byte pieces), with the NEXT step's loads in flight while a step is probed -- across slab
// boundaries too (the slabs' counts sit in the lanes of the wave).  The hash consumers' probes start with dependent LDS
// reads: with the loads issued only at the top of a step the kernels ran at the latency of a global load (2.1 TB/s on 30-KB
// slabs: 64 KB in flight per CU at best, nothing in flight while a step is probed), whatever was done to the probes
// themselves.  f(keys[8], values[8], live mask, full) -> false stops the walk (a table overflowed); full (wave-uniform):
// every pair of the step exists.
template <typename F>
__device__ __forceinline__ void walk_pair_slabs(const uint2 *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, uint32_t b, F &&f)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    // Round 5.  Rounds 3-4 walked with two register sets in turn so that the next step's loads would be in flight while a step
    // is probed -- and the generated code drained vmcnt(0) at the loop header, before every step (with the wave number a per-lane
    // value to the compiler every `break` was a divergent branch and the loads sat under exec masks it could not count; found by
    // scanning the generated code for drains right behind loads, after the partition producer's no-predicate instantiation had
    // shown the same disease).  Now the cursor is a SCALAR (the wave number through readfirstlane, the slabs' counts through
    // readlane) and the walk is the producer's pattern: take the arrived step out of its registers, issue the next step's loads,
    // probe.  Interleaved on one box the two walks run the hash consumers at the same speed (4.0-4.1 ms per sparse statement,
    // 0.91 ms for the reference entry, either way): 16 waves per CU hide a wave's load latency already and the kernels wait for
    // the LDS -- kept because it is the simpler code and says what it does.  (Loads from inline assembly with hand-placed waits
    // were tried first and refused by tools/check_hidden_loads.py: the register allocator copied in-flight registers at the
    // loop's back-edge.)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = blockDim.x >> 6;
    const int nslab = wave < nwg ? (nwg - wave + nwaves - 1) / nwaves : 0;          // <= 64 (at most 1024 partition workgroups)
    const uint32_t mycount = lane < nslab ? min(counts[(size_t)b * nwg + wave + lane * nwaves], cap) : 0u;
    struct Cursor { int j; uint32_t i0, n2, count; };
    auto seek = [&](Cursor &c) {                                          // the first step at or after c that has pairs
        while (c.j < nslab && c.i0 >= c.n2) {
            c.j++; c.i0 = 0u;
            c.count = c.j < nslab ? (uint32_t)__builtin_amdgcn_readlane((int)mycount, __builtin_amdgcn_readfirstlane(c.j)) : 0u;
            c.n2 = (c.count + 1u) / 2u;                                   // 16-byte pieces (cap is even: the last piece exists)
        }
    };
    auto fetch = [&](const Cursor &c, u4v (&q)[4]) {                      // unconditional (clamped addresses): a countable number of loads per step
        const bool live = c.j < nslab;
        const u4v *src4 = reinterpret_cast<const u4v *>(pbuf + ((size_t)b * nwg + wave % nwg + (size_t)(live ? c.j : 0) * nwaves) * cap);
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t i = c.i0 + 64u * k + lane; q[k] = __builtin_nontemporal_load(src4 + ((live && i < c.n2) ? i : 0u)); }
    };
    auto step = [&](const u4v (&q)[4], const Cursor &c) -> bool {
        uint32_t key[8], vb[8], live = 0xFFu;
        const bool full = 2u * c.i0 + 512u <= c.count;                    // (scalar) every pair of the step exists: all but a slab's last step
#pragma unroll
        for (int k = 0; k < 4; k++) { key[2 * k] = q[k].x; vb[2 * k] = q[k].y; key[2 * k + 1] = q[k].z; vb[2 * k + 1] = q[k].w; }
        if (!full) {
            live = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = c.i0 + 64u * k + lane;
                if (2u * i < c.count) live |= 1u << (2 * k);
                if (2u * i + 1u < c.count) live |= 1u << (2 * k + 1);
            }
        }
        return f(key, vb, live, full);
    };
    Cursor c{-1, 0u, 0u, 0u};
    seek(c);
    u4v nx[4];
    fetch(c, nx);
    while (c.j < nslab) {
        const Cursor cur = c;
        const u4v q[4] = {nx[0], nx[1], nx[2], nx[3]};                    // the step that has arrived
        c.i0 += 256u; seek(c);
        fetch(c, nx);                                                     // ... and the next one, in flight while this one is probed
        if (!step(q, cur)) break;
    }
}

// ---------------------------------------------------------------------------
// Hash flavour of the consumer: arbitrary u32 keys ("LDS-staged hash buckets")
// ---------------------------------------------------------------------------
// The producer routed pairs by the top bits of mix32(key).  One workgroup per bucket builds an open-addressing table in
// LDS.  mix32 is a bijection and every key of bucket b has b in the top log2(P) bits of mix32(key), so the LOW bits of
// mix32(key) identify the key inside its bucket: an entry is a 32-bit tag (those bits | 2^31 = occupied; 0 = empty, claimed
// with ds_cmpst_b32), a 32-bit row count and the 8-byte value slot -- all 32-bit LDS atomics beside the value's own
// (round 3 kept key | 2^32 and the count in ONE 64-bit word: a 64-bit read, a 64-bit compare-and-swap and a 64-bit add per
// row; the consumer took 2.0 ms per 5e8 pairs against 0.5 ms for the dense one).  The key is rebuilt with unmix32 at emit
// time.  A bucket with more distinct keys than the table holds is processed in R rounds, round r taking the keys with
// mix32(key ^ salt) % R == r (the slabs are re-read, the table is emitted after every round).
// tags per group: 4 (one ds_read_b64 per probe: 9 LDS cycles per 64 lanes against 25 for a ds_read_b128, tools/hashlds.hip) or 8.
// Four win where a round's keys leave the groups mostly empty (0.7 / 1.1 keys per group for the typed and the statistics
// consumers: 970 against 1048 us and 1238 against 1323 us per 5e8 pairs); the reference entry's consumer runs its groups at 1.0
// keys per four slots -- two or three displaced pairs in every step -- and is faster with eight (301 against 326 us per 1e8).
#ifndef HARK_TAGW
#define HARK_TAGW 4
#endif
#ifndef HARK_TAGW_OPS
#define HARK_TAGW_OPS 8
#endif
constexpr int kTagW = HARK_TAGW, kTagWOps = HARK_TAGW_OPS;
constexpr int kHashGroups = kTagW == 4 ? 2816 : 1024, kHashCap = kHashGroups * kTagW;   // entries of 14 B (value slot, count, 16-bit tag): 154 / 112 KiB of LDS
constexpr int kHashFill = 3072;                          // distinct keys per round and bucket (load <= 0.375: probe chains stay short;
                                                         // a wave waits for its longest chain, so the load factor is what matters)
__device__ __forceinline__ uint32_t unmix32(uint32_t y)
{
    y ^= y >> 16; y *= 0x43021123u; y ^= (y >> 15) ^ (y >> 30); y *= 0x1D69E2A5u; y ^= y >> 16;
    return y;
}

// ---- 16-bit tags, eight to a 16-byte group: ONE ds_read_b128 per probe -------------------------------------------------
// (fgb_agg_hash_kernel's comment has the measurements.)  x = the low `lowbits` (<= 24) bits of mix32(key) = the key inside its
// bucket.  Its home group is g = x * groups >> lowbits (any group count; two 24-bit multiplies and a funnel shift -- a first
// version divided by the identities per group: umulhi, two 32-bit multiplies and a correction per pair), its remainder
// tv = bits [shift, lowbits) of the product's low part, where 2^shift <= groups: inside a group the product steps by `groups`
// from key to key, so no two keys of a group share a tv.  A slot holds d << rembits | tv, d <= maxdisp (<= 3 for groups of 8, <= 7 for groups of 4) = how many groups
// past its home the key lives (the first group that had room when it came; all before it are full of other keys, and stay
// so: slots only ever go from empty to occupied); 0xFFFF = an empty slot (the one tag that spells 0xFFFF is never stored: its
// key overflows one group early).  Slot s of a group is half s >> 2 of word s & 3 (what the branch-free search hands back).
template <int TW>
struct TagGroups {
    static constexpr uint32_t W = (uint32_t)TW;
    static_assert(W == 4 || W == 8, "a group is one 8- or 16-byte LDS read");
    typedef typename std::conditional<TW == 4, uint2, uint4>::type Group;
    uint32_t *tagw;                                          // LDS [groups * W / 2]: two tags per word, 0xFFFF = empty
    uint32_t groups, lowbits, shift, rembits, maxdisp;
    static constexpr uint32_t kEmpty2 = 0xFFFFFFFFu;
    __device__ __forceinline__ void init(uint32_t *lds_words, uint32_t ngroups, int lowbits_)
    {
        tagw = lds_words; groups = ngroups; lowbits = (uint32_t)lowbits_;
        shift = 31u - (uint32_t)__clz((int)ngroups);         // groups >= 512 and lowbits <= 24 (>= 256 buckets: checked where the consumers are launched): rembits <= 15
        rembits = lowbits - shift;
        const uint32_t room = (1u << (16u - rembits)) - 1u, most = W == 4 ? 7u : 3u;
        maxdisp = room < most ? room : most;
    }
    // group and tv * 0x10001 (the tag of a key at home, in both halves of a word)
    __device__ __forceinline__ void home(uint32_t x, uint32_t &g, uint32_t &t2) const
    {
        const u64 pr = (u64)(x & 0xFFFFFFu) * (u64)(groups & 0xFFFFFFu);        // v_mul_u32_u24 + v_mul_hi_u32_u24
        const uint32_t lo = (uint32_t)pr, hi = (uint32_t)(pr >> 32);
        g = __builtin_amdgcn_alignbit(hi, lo, lowbits);
        t2 = __builtin_amdgcn_ubfe(lo, shift, rembits) * 0x10001u;
    }
    __device__ __forceinline__ Group read(uint32_t g) const { return reinterpret_cast<const Group *>(tagw)[g]; }
    // Where in a group is the tag of t2's halves?  A code p: the low bits = the word, p >> 4 = the half; -1: nowhere.  Branch-free:
    // v_pk_min_u16(x, 1) is 1 per non-zero half (the compiler turns the same thing written in C into compares and selects:
    // inline assembly), the words' indicators are packed into one, inverted, and the lowest set bit is the slot -- 7 / 14
    // instructions where a chain of compares took ~30 and four levels of divergent branches per pair.
    static __device__ __forceinline__ uint32_t nz16(uint32_t x) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "s"(0x00010001u)); return r; }
    static __device__ __forceinline__ int find(const uint2 &q, uint32_t t2)
    {
        const uint32_t n = nz16(q.x ^ t2) | (nz16(q.y ^ t2) << 1);
        int p; asm("v_ffbl_b32 %0, %1" : "=v"(p) : "v"(n ^ 0x00030003u));   // -1 when no bit is set
        return p;
    }
    static __device__ __forceinline__ int find(const uint4 &q, uint32_t t2)
    {
        const uint32_t n = nz16(q.x ^ t2) | (nz16(q.y ^ t2) << 1) | (nz16(q.z ^ t2) << 2) | (nz16(q.w ^ t2) << 3);
        int p; asm("v_ffbl_b32 %0, %1" : "=v"(p) : "v"(n ^ 0x000F000Fu));
        return p;
    }
    // slot s of a group is half s / (W / 2) of word s % (W / 2)
    static __device__ __forceinline__ uint32_t slot_of(int p) { return W == 4 ? (((uint32_t)p & 1u) | ((uint32_t)p >> 3)) : (((uint32_t)p & 3u) | ((uint32_t)p >> 2)); }
    static __device__ __forceinline__ uint32_t word_of(const uint2 &q, uint32_t w) { return w == 0 ? q.x : q.y; }
    static __device__ __forceinline__ uint32_t word_of(const uint4 &q, uint32_t w) { return w == 0 ? q.x : w == 1 ? q.y : w == 2 ? q.z : q.w; }
    static __device__ __forceinline__ void set_word(uint2 &q, uint32_t w, uint32_t v) { if (w == 0) q.x = v; else q.y = v; }
    static __device__ __forceinline__ void set_word(uint4 &q, uint32_t w, uint32_t v) { if (w == 0) q.x = v; else if (w == 1) q.y = v; else if (w == 2) q.z = v; else q.w = v; }
    // The slot of a key that its home group's read did not show: found further on, or claimed -- the first empty slot of the
    // first group with room, with a compare-and-swap on the WORD that holds it.  A lane that loses learns the word's new content
    // from the compare-and-swap itself and looks again, so two lanes with one key end up in one slot (a word changes at most
    // twice: <= W failures per group).  (A stale view would do -- slots only go from empty to occupied and every lane takes the
    // first empty slot in ONE order, so a lane reaches a slot only after it has seen every slot before it occupied in a view
    // the compare-and-swap refreshed -- and handing the probe's read in saves a round trip per pass, but picking it out of the
    // registers costs more issue slots than the round trip costs a kernel that waits for the LDS pipe, not for latency:
    // measured 2-5 % slower.)  Returns the slot; -1: every group the key may live in is full of other keys.  A claim adds one
    // to *used (nobody waits for the sum: the step looks at it once).
    __device__ __forceinline__ int locate(uint32_t tv, uint32_t g, uint32_t *used, bool &claimed) const
    {
        for (uint32_t d = 0; d <= maxdisp; d++) {
            uint32_t gg = g + d; if (gg >= groups) gg -= groups;
            const uint32_t t = tv | (d << rembits), t2 = t * 0x10001u;
            if (t == 0xFFFFu) break;                         // spells "empty"
            Group q = read(gg);
            for (int tries = 0; tries < 16; tries++) {
                const int at = find(q, t2);
                if (at >= 0) return (int)(W * gg + slot_of(at));
                const int e = find(q, kEmpty2);
                if (e < 0) break;                            // full of other keys: the next group
                const uint32_t w = (uint32_t)e & (W / 2 - 1u), oldw = word_of(q, w);
                const uint32_t got = atomicCAS(&tagw[(W / 2) * gg + w], oldw, oldw ^ ((t ^ 0xFFFFu) << ((uint32_t)e & 16u)));   // ds_cmpst_rtn_b32
                if (got == oldw) { atomicAdd(used, 1u); claimed = true; return (int)(W * gg + slot_of(e)); }
                set_word(q, w, got);
            }
        }
        return -1;
    }
    // the tag in a slot (0xFFFF: empty)
    __device__ __forceinline__ uint32_t tag_at(uint32_t slot) const
    {
        const uint32_t s = slot & (W - 1u);
        return (tagw[(W / 2) * (slot / W) + (s & (W / 2 - 1u))] >> (16u * (s / (W / 2)))) & 0xFFFFu;
    }
    // the key's low bits x back from its slot and tag (emit time): the one x whose product lies in the tag's window
    __device__ __forceinline__ uint32_t identity(uint32_t slot, uint32_t tag) const
    {
        const uint32_t d = tag >> rembits, tv = tag & ((1u << rembits) - 1u);
        uint32_t g = slot / W; g = g >= d ? g - d : g + groups - d;
        const u64 a = ((u64)g << lowbits) | ((u64)tv << shift);
        return (uint32_t)((a + groups - 1u) / groups);
    }
};

// One step of a hash consumer: eight pairs per lane against a TagGroups table -- the home groups' reads all in flight, then
// hit(slot, value) for the keys they show; what they do not show (new keys, displaced keys) is located one pair per pass, each
// lane its own first (the number of passes is the LARGEST number of misses any lane has -- one, seldom two -- not the number
// of pair positions at which some lane missed).  A round's budget of distinct keys (*used <= fill) is looked at once per step
// that claimed.  FULL (wave-uniform): every pair of the step is live and the round takes every key, so no pair carries a
// predicate of its own.  Sets overflow when a key found no room or the budget is spent (the round is void then).
// BATCH: the hits of a step are handed over together, the located pairs among them -- hit(slots[8], values[8], mask of the pairs
// with a slot) -- for a consumer whose operators are run-time values (the reference entry's): it branches on an operator once
// per step, not once per pair.
template <bool FULL, bool BATCH = false, typename TG, typename HIT>
__device__ __forceinline__ void tag_probe_step(const TG &tg, const uint32_t (&key)[8], const uint32_t (&vb)[8], uint32_t live,
                                               uint32_t lowmask, uint32_t Rmask, uint32_t r, uint32_t *used, uint32_t fill, bool &overflow, HIT &&hit)
{
    uint32_t t2[8], g[8]; typename TG::Group q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        tg.home(key[j] & lowmask, g[j], t2[j]);                               // the producer wrote mix32(key)
        if (!FULL && Rmask && (mix32(key[j] ^ 0x9E3779B9u) & Rmask) != r) live &= ~(1u << j);   // not this round's share of the key space
    }
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = tg.read(g[j]);                        // eight independent LDS reads in flight
    uint32_t miss = 0, slots[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = TG::find(q[j], t2[j]);
        miss |= (uint32_t)p & (0x100u << j);                                  // p = -1: every bit (a found p is below 32)
        if constexpr (BATCH) slots[j] = TG::W * g[j] + TG::slot_of(p);
        else if (FULL ? p >= 0 : (p >= 0 && ((live >> j) & 1u))) hit(TG::W * g[j] + TG::slot_of(p), vb[j]);
    }
    miss >>= 8;
    uint32_t ok = (FULL ? 0xFFu : live) & ~miss;                              // (BATCH) the pairs whose slot is known
    if (!FULL) miss &= live;
    if (__any(miss != 0u)) {
        bool claimed = false;
        do {
            const int first = miss ? __ffs((int)miss) - 1 : -1;
            uint32_t tj = 0u, gj = 0u, vj = 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) if (j == first) { tj = t2[j] & 0xFFFFu; gj = g[j]; if constexpr (!BATCH) vj = vb[j]; }
            if (miss) {
                const int slot = tg.locate(tj, gj, used, claimed);
                if (slot < 0) overflow = true;
                else if constexpr (BATCH) {                                   // joins the step's hits
#pragma unroll
                    for (int j = 0; j < 8; j++) if (j == first) slots[j] = (uint32_t)slot;
                    ok |= 1u << first;
                } else hit((uint32_t)slot, vj);
                miss &= miss - 1u;
            }
        } while (__any(miss != 0u));
        if (__any(claimed) && *used > fill) overflow = true;
    }
    if constexpr (BATCH) hit(slots, vb, ok);
}

template <int VOP>
__global__ __launch_bounds__(1024) void fgb_agg_hash_kernel(
    const uint2 *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, uint32_t Rmask, uint32_t r,
    uint32_t *__restrict__ out_key, u64 *__restrict__ out_val, u64 *__restrict__ out_cnt, unsigned long long *__restrict__ out_cursor,
    unsigned long long out_cap, int32_t *__restrict__ err, int xf /* order transform of the raw value bits (the pairs carry them raw) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u64 *t_val = reinterpret_cast<u64 *>(lds_raw);                       // [kHashCap]
    uint32_t *t_cnt = reinterpret_cast<uint32_t *>(t_val + kHashCap);    // [kHashCap] rows
    uint32_t *t_tagw = t_cnt + kHashCap;                                 // [kHashCap / 2] 16-bit tags, two per word (TagGroups; 0xFFFF = empty)
    __shared__ uint32_t s_used, s_emit;
    __shared__ unsigned long long s_base;
    const uint32_t b = blockIdx.x;
    const int lowbits = 33 - __ffs((int)gridDim.x);                      // P = gridDim.x buckets, a power of two: bucket = mix32(key) >> lowbits
    const uint32_t lowmask = (1u << lowbits) - 1u;
    for (int i = threadIdx.x; i < kHashCap; i += blockDim.x) { t_cnt[i] = 0u; t_val[i] = vop_identity(VOP); }
    for (int i = threadIdx.x; i < kHashCap / 2; i += blockDim.x) t_tagw[i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) { s_used = 0u; s_emit = 0u; }
    __syncthreads();
    bool overflow = false;
    // What the kernel costs (profiles/r04_notes.md 3, r05_notes.md 2): the stream of pairs 0.30 ms per 2.5e8; the rest is the LDS and
    // the instructions around it.  Round 4's counters of the 32-bit-tag version: LDS arrays busy 72 % of the kernel's cycles, 59 %
    // of them bank-conflict cycles.  tools/hashlds.hip prices the LDS work of a probe alone, per 64 pairs at random addresses:
    // two ds_read_b128 (eight 32-bit tags: round 4) 60 cycles, ONE ds_read_b128 25, one ds_read_b64 9; ds_add_f64 21, ds_add_u32
    // 7 -- and the costs add up (b128 + f64 + u32 = 53, b64 + f64 + u32 = 39 = 0.25 ms per 2.5e8 pairs: below the stream).
    // Hence 16-BIT tags, FOUR to an 8-byte group (TagGroups: the layout, the branch-free search and what happens to a key
    // whose home group is full of other keys); 2816 groups, so a round's <= 3072 keys (2048 of a 2^20-key table in 512
    // buckets) leave a group 0.7 keys on average: five or more in 0.09 % of the groups.  A probe finds its key in the
    // home group without a branch; anything else -- a new key, a displaced key -- takes the slow path, half a pair per step of
    // 512 once the table is built.  The sequence of the round: 32-bit tags in groups of eight 1.465 ms per 5e8 pairs; 16-bit
    // tags, eight to a group 1.15; the branch-free search and a multiplicative home group (~30 vector instructions per pair
    // where there were ~60) 1.04; groups of four 0.97.
    constexpr int kNP = 8;
    constexpr uint32_t kGroups = (uint32_t)kHashGroups;
    TagGroups<kTagW> tg;
    tg.init(t_tagw, kGroups, lowbits);
    auto hit = [&](uint32_t slot, uint32_t vbits) { vop_atomic<VOP>(&t_val[slot], VOP == VOP_F32SUM ? vbits : apply_xf(xf, vbits)); atomicAdd(&t_cnt[slot], 1u); };
    walk_pair_slabs(pbuf, counts, cap, nwg, b, [&](const uint32_t (&key)[kNP], const uint32_t (&vb)[kNP], uint32_t live, bool full) -> bool {
        if (full && !Rmask) tag_probe_step<true>(tg, key, vb, live, lowmask, Rmask, r, &s_used, (uint32_t)kHashFill, overflow, hit);
        else tag_probe_step<false>(tg, key, vb, live, lowmask, Rmask, r, &s_used, (uint32_t)kHashFill, overflow, hit);
        return !__any(overflow);                                             // an overflowed round is void anyway: stop reading
    });
    if (overflow) *err = kErrOverflow;
    __syncthreads();
    // emit the occupied entries: reserve a range of the output with one global atomic per workgroup
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < kHashCap; i += blockDim.x) mine += t_cnt[i] ? 1u : 0u;
    uint32_t pos = mine ? atomicAdd(&s_emit, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_emit ? atomicAdd(out_cursor, (unsigned long long)s_emit) : 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < kHashCap; i += blockDim.x) {
        const uint32_t c = t_cnt[i];
        if (!c) continue;
        const unsigned long long o = s_base + pos++;
        if (o < out_cap) {
            out_key[o] = unmix32((b << lowbits) | tg.identity((uint32_t)i, tg.tag_at((uint32_t)i))); out_val[o] = t_val[i]; out_cnt[o] = (u64)c;
        }
        else *err = kErrOverflow;
    }
}

// Statistics flavour of the hash consumer: SUM / AVG / COUNT + MIN + MAX of ONE value column from one pass over the pairs
// (the hash counterpart of fgb_agg6_stats_kernel): an entry is a tag, a row count, a 64-bit sum slot and the smallest and
// largest ORDER word of the raw values; 2560 distinct keys per round and bucket (2048 would put 2^20 keys in 512 buckets exactly
// on the edge: a failed first round, a sample round and two rounds -- measured 4.66 ms against 4.0 ms for three separate
// passes).  Rounds 3-4: 32-bit tags, 24 bytes per entry, 5120 entries in 640 groups of eight (120 KiB).
// VK: 0 = f32 values (f64 sum), 1 = i32 (sum of the biased values, like XF_I32_ORDER), 2 = u32.
// (round 5: 16-bit tags -- TagGroups, 22 bytes per entry -- 7424 entries in the same 160 KiB instead of 5120, in 1856 groups of four:
// 1.1 keys per group for 2^20 keys in 512 buckets, and one ds_read_b64 per probe.)
constexpr int kHashSGroups = 928 * 8 / kTagW, kHashSCap = kHashSGroups * kTagW, kHashSFill = 2560;
template <int VK>
__global__ __launch_bounds__(1024) void fgb_agg_hash_stats_kernel(
    const uint2 *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, uint32_t Rmask, uint32_t r,
    uint32_t *__restrict__ out_key, u64 *__restrict__ out_sum, u64 *__restrict__ out_cnt, u64 *__restrict__ out_min, u64 *__restrict__ out_max,
    unsigned long long *__restrict__ out_cursor, unsigned long long out_cap, int32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u64 *t_sum = reinterpret_cast<u64 *>(lds_raw);                       // [kHashSCap]
    uint32_t *t_cnt = reinterpret_cast<uint32_t *>(t_sum + kHashSCap);   // [kHashSCap]
    uint32_t *t_min = t_cnt + kHashSCap, *t_max = t_min + kHashSCap;     // [kHashSCap] each
    uint32_t *t_tagw = t_max + kHashSCap;                                // [kHashSCap / 2] 16-bit tags (TagGroups; 20 x 7424 bytes in: 16-byte aligned)
    __shared__ uint32_t s_used, s_emit;
    __shared__ unsigned long long s_base;
    const uint32_t b = blockIdx.x;
    const int lowbits = 33 - __ffs((int)gridDim.x);
    const uint32_t lowmask = (1u << lowbits) - 1u;
    for (int i = threadIdx.x; i < kHashSCap; i += blockDim.x) { t_cnt[i] = 0u; t_sum[i] = 0ull; t_min[i] = 0xFFFFFFFFu; t_max[i] = 0u; }
    for (int i = threadIdx.x; i < kHashSCap / 2; i += blockDim.x) t_tagw[i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) { s_used = 0u; s_emit = 0u; }
    __syncthreads();
    bool overflow = false;
    constexpr int kNP = 8;
    TagGroups<kTagW> tg;
    tg.init(t_tagw, (uint32_t)kHashSGroups, lowbits);
    auto hit = [&](uint32_t slot, uint32_t raw) {
        uint32_t w;
        if constexpr (VK == 0) { unsafeAtomicAdd(reinterpret_cast<double *>(&t_sum[slot]), (double)__uint_as_float(raw)); w = apply_xf(XF_F32_ORDER, raw); }
        else if constexpr (VK == 1) { w = apply_xf(XF_I32_ORDER, raw); atomicAdd(&t_sum[slot], (u64)w); }
        else { w = raw; atomicAdd(&t_sum[slot], (u64)w); }
        atomicMin(&t_min[slot], w); atomicMax(&t_max[slot], w);
        atomicAdd(&t_cnt[slot], 1u);
    };
    walk_pair_slabs(pbuf, counts, cap, nwg, b, [&](const uint32_t (&key)[kNP], const uint32_t (&vb)[kNP], uint32_t live, bool full) -> bool {
        if (full && !Rmask) tag_probe_step<true>(tg, key, vb, live, lowmask, Rmask, r, &s_used, (uint32_t)kHashSFill, overflow, hit);
        else tag_probe_step<false>(tg, key, vb, live, lowmask, Rmask, r, &s_used, (uint32_t)kHashSFill, overflow, hit);
        return !__any(overflow);
    });
    if (overflow) *err = kErrOverflow;
    __syncthreads();
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < kHashSCap; i += blockDim.x) mine += t_cnt[i] ? 1u : 0u;
    uint32_t pos = mine ? atomicAdd(&s_emit, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_emit ? atomicAdd(out_cursor, (unsigned long long)s_emit) : 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < kHashSCap; i += blockDim.x) {
        const uint32_t c = t_cnt[i];
        if (!c) continue;
        const unsigned long long o = s_base + pos++;
        if (o < out_cap) {
            out_key[o] = unmix32((b << lowbits) | tg.identity((uint32_t)i, tg.tag_at((uint32_t)i)));
            out_sum[o] = t_sum[i]; out_cnt[o] = (u64)c; out_min[o] = (u64)t_min[i]; out_max[o] = (u64)t_max[i];
        } else *err = kErrOverflow;
    }
}

// One, two or three of the reference's u32 operators (groupby.fut:35-41) over ONE value column from one pass over its pairs:
// an entry is a 16-bit tag and one 32-bit slot per operator -- 6, 10 or 14 B in 3408 / 2040 / 1456 groups of eight (160 KiB) --
// and a round takes 8192 / 6144 / 4608 distinct keys per bucket (2^21 keys in 512 buckets: ONE round, two keys per group),
// instead of one consumer pass and one sort of the result keys per operator.  ops: operator of slot j in byte j
// (VOP_U32SUM / MAX / MIN / PROD).  Emits slot 0 | slot 1 << 32 as the value word and slot 2 as the count word (one
// operator: the value word only).  (Rounds 3-4 kept a single operator in 8-byte entries, value << 32 | tag, with one 64-bit
// atomic per row and two ds_read_b128 per probe: 435 us per 1e8 pairs against 245 for this kernel with one slot per entry.)
template <int NOPS> struct HashOpsGeo { static constexpr int groups = (NOPS == 1 ? 3408 : NOPS == 2 ? 2040 : 1456) * 8 / kTagWOps, cap = groups * kTagWOps, fill = NOPS == 1 ? 8192 : NOPS == 2 ? 6144 : 4608; };
__device__ __forceinline__ void op32_atomic(int vop, uint32_t *slot, uint32_t x)
{
    if (vop == VOP_U32SUM) atomicAdd(slot, x);
    else if (vop == VOP_U32MAX) atomicMax(slot, x);
    else if (vop == VOP_U32MIN) atomicMin(slot, x);
    else { uint32_t old = *slot, assumed; do { assumed = old; old = atomicCAS(slot, assumed, assumed * x); } while (old != assumed); }
}
template <int NOPS>
__global__ __launch_bounds__(1024) void fgb_agg_hash_ops_kernel(
    const uint2 *__restrict__ pbuf, const uint32_t *__restrict__ counts, uint32_t cap, int nwg, uint32_t Rmask, uint32_t r,
    uint32_t *__restrict__ out_key, u64 *__restrict__ out_val, u64 *__restrict__ out_cnt,
    unsigned long long *__restrict__ out_cursor, unsigned long long out_cap, int32_t *__restrict__ err, uint32_t ops)
{
    constexpr int kCap = HashOpsGeo<NOPS>::cap, kGroups = HashOpsGeo<NOPS>::groups, kFill = HashOpsGeo<NOPS>::fill;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *t_tagw = reinterpret_cast<uint32_t *>(lds_raw);            // [kCap / 2] 16-bit tags (TagGroups), 0xFFFF = empty
    uint32_t *t_a = t_tagw + kCap / 2, *t_b = t_a + kCap, *t_c = t_b + kCap;   // [kCap] each; t_c with three operators only
    __shared__ uint32_t s_used, s_emit;
    __shared__ unsigned long long s_base;
    const uint32_t b = blockIdx.x;
    const int lowbits = 33 - __ffs((int)gridDim.x);
    const uint32_t lowmask = (1u << lowbits) - 1u;
    const int op_a = (int)(ops & 255u), op_b = (int)((ops >> 8) & 255u), op_c = (int)((ops >> 16) & 255u);
    const uint32_t id_a = (uint32_t)vop_identity(op_a), id_b = (uint32_t)vop_identity(op_b), id_c = (uint32_t)vop_identity(op_c);
    for (int i = threadIdx.x; i < kCap; i += blockDim.x) { t_a[i] = id_a; if constexpr (NOPS >= 2) t_b[i] = id_b; if constexpr (NOPS == 3) t_c[i] = id_c; }
    for (int i = threadIdx.x; i < kCap / 2; i += blockDim.x) t_tagw[i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) { s_used = 0u; s_emit = 0u; }
    __syncthreads();
    bool overflow = false;
    constexpr int kNP = 8;
    TagGroups<kTagWOps> tg;
    tg.init(t_tagw, (uint32_t)kGroups, lowbits);                         // 2040 / 1456 groups of eight
    // The operators are run-time values: a step's hits are applied operator by operator, one (uniform) branch per operator and
    // step -- with a branch chain per pair and operator the kernel took 301 us per 1e8 pairs, with the operators compiled in 252.
    struct Hit {
        uint32_t *t_a, *t_b, *t_c; int op_a, op_b, op_c;
        static __device__ __forceinline__ void all(int vop, uint32_t *t, const uint32_t (&slots)[8], const uint32_t (&x)[8], uint32_t ok)
        {
            if (vop == VOP_U32SUM) {
#pragma unroll
                for (int j = 0; j < 8; j++) if ((ok >> j) & 1u) atomicAdd(&t[slots[j]], x[j]);
            } else if (vop == VOP_U32MAX) {
#pragma unroll
                for (int j = 0; j < 8; j++) if ((ok >> j) & 1u) atomicMax(&t[slots[j]], x[j]);
            } else if (vop == VOP_U32MIN) {
#pragma unroll
                for (int j = 0; j < 8; j++) if ((ok >> j) & 1u) atomicMin(&t[slots[j]], x[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) if ((ok >> j) & 1u) op32_atomic(vop, &t[slots[j]], x[j]);
            }
        }
        __device__ __forceinline__ void operator()(const uint32_t (&slots)[8], const uint32_t (&x)[8], uint32_t ok) const   // a step's hits
        {
            all(op_a, t_a, slots, x, ok);
            if constexpr (NOPS >= 2) all(op_b, t_b, slots, x, ok);
            if constexpr (NOPS == 3) all(op_c, t_c, slots, x, ok);
        }
    };
    const Hit hit{t_a, t_b, t_c, op_a, op_b, op_c};
    walk_pair_slabs(pbuf, counts, cap, nwg, b, [&](const uint32_t (&key)[kNP], const uint32_t (&vb)[kNP], uint32_t live, bool full) -> bool {
        if (full && !Rmask) tag_probe_step<true, true>(tg, key, vb, live, lowmask, Rmask, r, &s_used, (uint32_t)kFill, overflow, hit);
        else tag_probe_step<false, true>(tg, key, vb, live, lowmask, Rmask, r, &s_used, (uint32_t)kFill, overflow, hit);
        return !__any(overflow);
    });
    if (overflow) *err = kErrOverflow;
    __syncthreads();
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < kCap; i += blockDim.x) mine += tg.tag_at((uint32_t)i) != 0xFFFFu ? 1u : 0u;
    uint32_t pos = mine ? atomicAdd(&s_emit, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_emit ? atomicAdd(out_cursor, (unsigned long long)s_emit) : 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < kCap; i += blockDim.x) {
        const uint32_t t = tg.tag_at((uint32_t)i);
        if (t == 0xFFFFu) continue;
        const unsigned long long o = s_base + pos++;
        if (o < out_cap) {
            out_key[o] = unmix32((b << lowbits) | tg.identity((uint32_t)i, t));
            if constexpr (NOPS == 1) out_val[o] = (u64)t_a[i];                 // (no count array with one operator)
            else { out_val[o] = (u64)t_a[i] | ((u64)t_b[i] << 32); out_cnt[o] = NOPS == 3 ? (u64)t_c[i] : 0ull; }
        } else *err = kErrOverflow;
    }
}

__global__ __launch_bounds__(256) void fgb_finish_kernel(const double *__restrict__ acc_sum, const unsigned long long *__restrict__ acc_cnt,
                                                         int64_t G, float *__restrict__ sum_out, int64_t *__restrict__ cnt_out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
        if (sum_out) sum_out[g] = (float)acc_sum[g];
        if (cnt_out) cnt_out[g] = (int64_t)acc_cnt[g];
    }
}

__global__ __launch_bounds__(256) void fgb_finish_u32_kernel(const u64 *__restrict__ acc, const unsigned long long *__restrict__ acc_cnt,
                                                             int64_t G, uint32_t *__restrict__ val_out, int64_t *__restrict__ cnt_out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
        if (val_out) val_out[g] = (uint32_t)acc[g];
        if (cnt_out) cnt_out[g] = (int64_t)acc_cnt[g];
    }
}

// the four accumulators of a statistics pass: sums, counts and maxima 0, minima all ones (order words)
__global__ __launch_bounds__(256) void fgb_init_stats_kernel(u64 *__restrict__ sum, unsigned long long *__restrict__ cnt, u64 *__restrict__ mn, u64 *__restrict__ mx, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { sum[i] = 0ull; cnt[i] = 0ull; mn[i] = 0xFFFFFFFFull; mx[i] = 0ull; }
}

__global__ __launch_bounds__(256) void fgb_fill_kernel(u64 *__restrict__ dst, int64_t n, u64 v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// Typed read-out of the accumulators.  kind: 0 f32 <- f64 sum | 1 u32 <- low word | 2 i32 <- ordered u32 |
// 3 f32 <- ordered u32 | 4 i64 <- u64 sum | 5 i64 <- u64 sum of ordered i32 (minus count * 2^31) |
// 7/8/9 f32 average from the sums of kinds 0/4/5.  pos != NULL compacts: only groups with a
// non-zero count are written, at out[pos[g]]; key_out (optional) receives g there.
__global__ __launch_bounds__(256) void fgb_decode_kernel(const u64 *__restrict__ acc, const unsigned long long *__restrict__ cnt, int64_t G, int kind,
                                                         const uint32_t *__restrict__ pos, void *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += stride) {
        const unsigned long long c = cnt[g];
        if (pos && !c) continue;
        const int64_t o = pos ? (int64_t)pos[g] : g;
        const u64 a = acc[g];
        const uint32_t lo = (uint32_t)a;
        switch (kind) {
        case 0: static_cast<float *>(out)[o] = (float)__longlong_as_double((long long)a); break;
        case 1: static_cast<uint32_t *>(out)[o] = lo; break;
        case 2: static_cast<uint32_t *>(out)[o] = lo ^ 0x80000000u; break;
        case 3: static_cast<uint32_t *>(out)[o] = (lo & 0x80000000u) ? (lo ^ 0x80000000u) : ~lo; break;
        case 4: static_cast<long long *>(out)[o] = (long long)a; break;
        case 5: static_cast<long long *>(out)[o] = (long long)(a - (c << 31)); break;
        case 7: static_cast<float *>(out)[o] = (float)(__longlong_as_double((long long)a) / (double)c); break;
        case 8: static_cast<float *>(out)[o] = (float)((double)a / (double)c); break;
        case 9: static_cast<float *>(out)[o] = (float)((double)(long long)(a - (c << 31)) / (double)c); break;
        case 10: static_cast<uint32_t *>(out)[o] = (uint32_t)g; break;                    // the key itself
        default: static_cast<long long *>(out)[o] = (long long)c; break;                   // 11: the count
        }
    }
}

template <typename F>
int dispatch_vop(int vop, F &&f)
{
    switch (vop) {
    case VOP_F32SUM: return f(std::integral_constant<int, VOP_F32SUM>{});
    case VOP_U32SUM: return f(std::integral_constant<int, VOP_U32SUM>{});
    case VOP_U32MAX: return f(std::integral_constant<int, VOP_U32MAX>{});
    case VOP_U32MIN: return f(std::integral_constant<int, VOP_U32MIN>{});
    case VOP_U32PROD: return f(std::integral_constant<int, VOP_U32PROD>{});
    case VOP_U32SUM64: return f(std::integral_constant<int, VOP_U32SUM64>{});
    default: return HARK_EARG;
    }
}

template <typename F>
int dispatch_op(int cmp, bool has_pred, F &&f)
{
    if (!has_pred) return f(std::integral_constant<int, kNoPred>{});
    switch (cmp) {
    case HARK_CMP_GT: return f(std::integral_constant<int, HARK_CMP_GT>{});
    case HARK_CMP_GE: return f(std::integral_constant<int, HARK_CMP_GE>{});
    case HARK_CMP_LT: return f(std::integral_constant<int, HARK_CMP_LT>{});
    case HARK_CMP_LE: return f(std::integral_constant<int, HARK_CMP_LE>{});
    case HARK_CMP_EQ: return f(std::integral_constant<int, HARK_CMP_EQ>{});
    case HARK_CMP_NE: return f(std::integral_constant<int, HARK_CMP_NE>{});
    case HARK_CMP_MASK: return f(std::integral_constant<int, HARK_CMP_MASK>{});
    default: return HARK_EARG;
    }
}

// Optional live timing: HIP events recorded on the launch stream around every kernel of the path.
struct TimedLaunch {
    hark_fgb_plan *pl; hipStream_t st; int kind; size_t slot = 0; bool on = false;
    TimedLaunch(hark_fgb_plan *pl_, hipStream_t st_, int kind_) : pl(pl_), st(st_), kind(kind_)
    {
        if (!pl->timing) return;
        if (pl->ev_used + 2 > pl->ev.size()) {
            for (int i = 0; i < 2; i++) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; pl->ev.push_back(e); }
            pl->ev_kind.resize(pl->ev.size() / 2);
        }
        slot = pl->ev_used; pl->ev_used += 2; pl->ev_kind[slot / 2] = kind; on = true;
        hipEventRecord(pl->ev[slot], st);
    }
    ~TimedLaunch() { if (on) hipEventRecord(pl->ev[slot + 1], st); }
};

constexpr int64_t kLdsTableBudget = 159 * 1024;  // LDS path: 12 B per group -> G <= 13568 (one 1024-thread workgroup per CU; measured 2.0 ms per 1e9 rows at G = 13000 against 3.5 ms through the partition path)
constexpr int64_t kAggTableBudget = 96 * 1024;   // consumer: 12 B per key of a bucket

} // namespace

int k_gen_columns(hark_context *ctx, uint64_t seed, int64_t first_row, int64_t n, uint32_t G,
                  int exact, float *p, int32_t *k, float *v)
{
    if (n < 0 || G == 0) return hark_fail(ctx, HARK_EARG, "gen_columns: n < 0 or G == 0");
    if (n == 0) return HARK_OK;
    int pow2 = (G & (G - 1)) == 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > ctx->num_cu * 16) blocks = ctx->num_cu * 16;
    gen_columns_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(seed, first_row, n, G, pow2, exact, p, k, v);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

static void plan_drop_partition(hark_fgb_plan *pl)
{
    if (pl->pbuf) { hark_free(pl->ctx, pl->pbuf); pl->pbuf = nullptr; }
    if (pl->counts) { hark_free(pl->ctx, pl->counts); pl->counts = nullptr; }
}

static int fgb_plan_new(hark_context *ctx, hark_fgb_plan **out, int64_t max_rows, int64_t G, bool clear);
int hark_fgb_plan_new(hark_context *ctx, hark_fgb_plan **out, int64_t max_rows, int64_t G) { return fgb_plan_new(ctx, out, max_rows, G, true); }
// ... for callers that reset or initialise the accumulators before every pass anyway (the statement entries of k_groupby.hip): without
// the two clears of 8 B x G (13 us per statement at 2^20 groups)
int k_fgb_plan_new_uncleared(hark_context *ctx, hark_fgb_plan **out, int64_t max_rows, int64_t G) { return fgb_plan_new(ctx, out, max_rows, G, false); }

static int fgb_plan_new(hark_context *ctx, hark_fgb_plan **out, int64_t max_rows, int64_t G, bool clear)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !out) return HARK_EARG;
    *out = nullptr;
    if (max_rows < 0 || G <= 0 || G > (int64_t)1 << 31)
        return hark_fail(ctx, HARK_EARG, "fgb_plan_new: need max_rows >= 0 and 0 < G <= 2^31 (got %lld, %lld)",
                         (long long)max_rows, (long long)G);
    hark_fgb_plan *pl = new hark_fgb_plan();
    pl->max_rows = max_rows; pl->G = G; pl->ctx = ctx;
    pl->tile_rows = kTileRows;
    int rc = hark_alloc(ctx, (void **)&pl->err, 32);                  // [0] sticky error word, [1] the producers' batch counter, [2..3] pairs, [4..5] the window path's counters
    if (!rc) rc = hark_alloc(ctx, (void **)&pl->acc_sum, (size_t)G * sizeof(double));
    if (!rc) rc = hark_alloc(ctx, (void **)&pl->acc_cnt, (size_t)G * sizeof(unsigned long long));
    if (rc) { hark_fgb_plan_free(ctx, pl); return rc; }
    HIP_TRY_RC(ctx, rc, hipMemsetAsync(pl->err, 0, 32, ctx->stream));
    if (clear) {
        HIP_TRY_RC(ctx, rc, hipMemsetAsync(pl->acc_sum, 0, (size_t)G * sizeof(double), ctx->stream));
        HIP_TRY_RC(ctx, rc, hipMemsetAsync(pl->acc_cnt, 0, (size_t)G * sizeof(unsigned long long), ctx->stream));
    }
    if (rc) { hark_fgb_plan_free(ctx, pl); return rc; }
    *out = pl;
    return HARK_OK;
}

int hark_fgb_plan_free(hark_context *ctx, hark_fgb_plan *pl)
{
    hark_device_guard guard__(ctx);
    if (!pl) return HARK_OK;
    if (ctx) hipStreamSynchronize(ctx->stream);
    plan_drop_partition(pl);
    hark_free(ctx, pl->err);
    hark_free(ctx, pl->acc_sum);
    hark_free(ctx, pl->acc_cnt);
    hark_free(ctx, pl->acc_min); hark_free(ctx, pl->acc_max);
    for (auto e : pl->ev) hipEventDestroy(e);
    delete pl;
    return HARK_OK;
}

int hark_fgb_plan_set(hark_fgb_plan *pl, const char *key, int64_t value)
{
    if (!pl || !key) return HARK_EARG;
    if (!strcmp(key, "algo")) { if (value < 0 || value > 3) return HARK_EARG; pl->algo = value; }
    else if (!strcmp(key, "chunk_rows")) { if (value < 0) return HARK_EARG; pl->chunk_rows = value; }
    else if (!strcmp(key, "grid")) { if (value < 0 || value > 65535) return HARK_EARG; pl->grid = value; }
    else if (!strcmp(key, "shift")) { if (value < 0 || value > 13) return HARK_EARG; pl->shift = value; }
    else if (!strcmp(key, "slack_pct")) { if (value < 0 || value > 10000) return HARK_EARG; pl->slack_pct = value; }
    else if (!strcmp(key, "timing")) { pl->timing = value; return HARK_OK; }
    else if (!strcmp(key, "vop")) { if (value < 0 || value > 5) return HARK_EARG; pl->vop = value; return HARK_OK; }   // reset afterwards
    else if (!strcmp(key, "xform")) { if (value < 0 || value > 2) return HARK_EARG; pl->xform = value; return HARK_OK; }
    else if (!strcmp(key, "pairfmt")) { if (value < 0 || value > 3) return HARK_EARG; pl->pairfmt = value; }   // 0 auto, 1: 8-byte pairs, 2: compact 6-byte units, 3: 6-byte units from 8-byte ring entries (<= 128 buckets)
    else if (!strcmp(key, "window")) { if (value < 0 || value > 3) return HARK_EARG; pl->window = value; return HARK_OK; }   // 0: by the test, 1: always the window path, 2: never (nor rotated loads), 3: always the partition with rotated loads
    else if (!strcmp(key, "period")) { if (value < 0 || value > 15) return HARK_EARG; pl->period = value; return HARK_OK; }   // batches between ring sweeps (0 = default); any value gives the same result
    else return HARK_EARG;
    plan_drop_partition(pl);     // partition geometry depends on the knobs
    return HARK_OK;
}

int hark_fgb_reset(hark_context *ctx, hark_fgb_plan *pl)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl) return HARK_EARG;
    const u64 ident = vop_identity((int)pl->vop);
    if (ident == 0) HIP_TRY(ctx, hipMemsetAsync(pl->acc_sum, 0, (size_t)pl->G * sizeof(double), ctx->stream));
    else {
        int64_t blocks = (pl->G + 255) / 256;
        if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
        fgb_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(reinterpret_cast<u64 *>(pl->acc_sum), pl->G, ident);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemsetAsync(pl->acc_cnt, 0, (size_t)pl->G * sizeof(unsigned long long), ctx->stream));
    return HARK_OK;
}

int hark_fgb_acc_device(hark_fgb_plan *pl, void **sum_f64, void **count_i64)
{
    if (!pl) return HARK_EARG;
    if (sum_f64) *sum_f64 = pl->acc_sum;
    if (count_i64) *count_i64 = pl->acc_cnt;
    return HARK_OK;
}

static int plan_prepare_partition(hark_context *ctx, hark_fgb_plan *pl)
{
    if (pl->pbuf) return HARK_OK;
    // bucket = key >> shift; KPB = 1 << shift keys per bucket, 12 B of LDS each in the consumer.
    // default: ~256 buckets (one consumer workgroup per CU), i.e. shift = ceil(log2 G) - 8 in [4, 13]
    int lg = 0;
    while (((int64_t)1 << lg) < pl->G) lg++;
    int shift = pl->shift ? (int)pl->shift : (lg - 8 < 4 ? 4 : (lg - 8 > 13 ? 13 : lg - 8));
    while ((((pl->G - 1) >> shift) + 1) > kMaxBuckets) shift++;
    if (((int64_t)12 << shift) > kAggTableBudget)
        return hark_fail(ctx, HARK_EUNSUPPORTED, "fgb: G = %lld needs more than %d buckets of <= %lld keys",
                         (long long)pl->G, kMaxBuckets, (long long)(kAggTableBudget / 12));
    pl->shift = shift;
    pl->P = ((pl->G - 1) >> shift) + 1;
    pl->nwg = pl->grid ? pl->grid : (int64_t)ctx->num_cu;       // one 1024-thread producer per CU: it needs ~110 VGPRs, so only one is resident anyway
    int64_t chunk = pl->chunk_rows ? pl->chunk_rows : (int64_t)1 << 30;   // 10.4 GB of slabs per 2^30 rows; fewer, longer slabs
    chunk = (chunk + kTileRows - 1) / kTileRows * kTileRows;
    if (pl->max_rows > 0 && chunk > pl->max_rows) chunk = (pl->max_rows + kTileRows - 1) / kTileRows * kTileRows;
    pl->chunk_rows = chunk;
    // a (bucket, workgroup) slab holds slack x the uniform share of a chunk in which every
    // row survives, plus room for the fluctuation of a few tiles
    const int64_t slack = pl->slack_pct ? pl->slack_pct : 130;
    int64_t cap = chunk / (pl->P * pl->nwg) * slack / 100 + 256;
    cap = (cap + kLine - 1) / kLine * kLine + 2 * kLine;      // whole 128-byte lines; one spare for the final flush
    if (cap > 0x7FFFFFF0ll) return hark_fail(ctx, HARK_EARG, "fgb: chunk too large");
    pl->cap = cap;
    // the second geometry (f32-sum / value-operator passes with a value column, k_fgb_dense_f32): at most 128 buckets whose
    // producer rings hold one 8-byte LDS word per pair (FMT 3) -- when 128 buckets of <= 8192 keys cover G
    size_t bytes = (size_t)pl->P * (size_t)pl->nwg * (size_t)cap * sizeof(uint2);
    pl->shift8 = pl->P8 = pl->cap8 = 0;
    if (pl->pairfmt == 3 || pl->pairfmt == 0) {
        int s8 = shift;
        while ((((pl->G - 1) >> s8) + 1) > kMaxBuckets8e) s8++;
        if (((int64_t)12 << s8) <= kAggTableBudget) {
            const int64_t P8 = ((pl->G - 1) >> s8) + 1;
            int64_t cap8 = chunk / (P8 * pl->nwg) * slack / 100 + 256;
            cap8 = (cap8 + kLine - 1) / kLine * kLine + 2 * kLine;
            if (cap8 <= 0x7FFFFFF0ll) {
                pl->shift8 = s8; pl->P8 = P8; pl->cap8 = cap8;
                const size_t b8 = (size_t)P8 * (size_t)pl->nwg * (size_t)cap8 * sizeof(uint2);
                if (b8 > bytes) bytes = b8;
            }
        }
    }
    HARK_TRY(hark_alloc(ctx, (void **)&pl->pbuf, bytes));
    HARK_TRY(hark_alloc(ctx, (void **)&pl->counts, (size_t)pl->P * (size_t)pl->nwg * sizeof(uint32_t)));
    return HARK_OK;
}

// Does this key column take the window path?  The plan's knob ("window": 1 always, 2 never; HARK_FGB_WINDOW=1 / 0 the same for
// every plan), else the test of fgb_cluster_test_kernel -- one small launch and one synchronisation, once per plan and column:
// the verdict sticks to (column, rows) until a check finds that the window path sent more than a third of the rows to global atomics.
static int fgb_window_wanted(hark_context *ctx, hark_fgb_plan *pl, const int32_t *k, int64_t n, bool *window)
{
    *window = false;
    if (const char *e = getenv("HARK_FGB_WINDOW")) { *window = atoi(e) == 1; pl->win_k = k; pl->win_n = n; pl->win_verdict = atoi(e) == 2 ? 2 : *window ? 1 : 0; return HARK_OK; }   // (2: rotated loads)
    if (pl->window == 1) { *window = true; return HARK_OK; }
    if (pl->window == 3) { pl->win_k = k; pl->win_n = n; pl->win_verdict = 2; return HARK_OK; }
    if (pl->window == 2 || n < ((int64_t)1 << 20)) { if (pl->win_k == k && pl->win_verdict == 2 && pl->window == 2) pl->win_verdict = 0; return HARK_OK; }
    if (pl->win_k != k || pl->win_n != n || pl->win_verdict < 0) {
        unsigned long long *out = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctx->h_pin) + 65536 - 128);   // (the kernel writes pinned host memory)
        fgb_cluster_test_kernel<<<1, 1024, 0, ctx->stream>>>(k, n, out);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        pl->win_k = k; pl->win_n = n; pl->win_verdict = (int)reinterpret_cast<volatile unsigned long long *>(out)[0];     // 0 scattered, 1 window, 2 rotated loads
    }
    *window = pl->win_verdict == 1;
    return HARK_OK;
}

// the producer's rotation stride for a chunk of `rows` rows (0: none): the verdict of fgb_cluster_test_kernel was "neighbouring rows share a
// bucket"; group g of the 64 reads batch + g * rot -- an odd stride (block lengths that divide the table evenly would otherwise put
// all 64 places at the same spot of their blocks), 63 * rot below the number of full batches
static int64_t fgb_rot_of(hark_fgb_plan *pl, const void *k, int64_t rows)
{
    if (pl->win_k != k || pl->win_verdict != 2 || getenv("HARK_FGB_NO_ROTATE")) return 0;
    const int64_t nfull = rows / kBatchRows;
    if (nfull < 128) return 0;
    pl->rot_rows += rows;
    return (nfull / 64 - 1) | 1;
}

int k_fgb_dense_f32(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr,
                    const int32_t *k, const float *v, int64_t n)
{
    if (n == 0) return HARK_OK;
    const int64_t G = pl->G;
    auto misaligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) != 0; };
    if ((p && misaligned(p)) || misaligned(k) || (v && misaligned(v)))
        return hark_fail(ctx, HARK_EARG, "fgb: columns must be 16-byte aligned");
    u64 *gsum = reinterpret_cast<u64 *>(pl->acc_sum);
    const int vop = (int)pl->vop;
    unsigned long long *gcnt = pl->acc_cnt;
    int algo = (int)pl->algo;
    // auto: LDS tables while 12 B x G fits a workgroup (159 KiB); the partition path up to 256 buckets x 8192 keys;
    // beyond that (G > 2^21) one global atomic pair per surviving row (slow, but any G works)
    if (algo == 0) algo = (G * 12 <= kLdsTableBudget) ? 1 : (G <= (int64_t)kMaxBuckets * (kAggTableBudget / 12)) ? 3 : 2;
    if (algo == 1 && G * 12 > 159 * 1024)
        return hark_fail(ctx, HARK_EUNSUPPORTED, "fgb: LDS path needs 12*G <= 159 KiB");
    hipStream_t st = ctx->stream;

    if (algo == 1) {
        // replicate the table per lane while it stays under 24 KiB (tiny G)
        int RL = 0;
        while (RL < 5 && (G << (RL + 1)) * 12 <= 24 * 1024) RL++;
        const size_t lds = (size_t)(G << RL) * 12;
        // one 1024-thread workgroup per CU measured best (0.78 of peak vs 0.76 at two; profiles/r01_notes.md)
        int64_t grid = pl->grid ? pl->grid : (int64_t)ctx->num_cu;
        const int64_t need = (n / kVec + 1023) / 1024;
        if (grid > need) grid = need > 0 ? need : 1;
        auto launch = [&](auto op, auto vm) -> int {
            constexpr int OP = decltype(op)::value;
            constexpr int VM = decltype(vm)::value;
            if (lds > 64 * 1024)
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_lds_kernel<OP, VM>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            TimedLaunch tl(pl, st, 0);
            fgb_lds_kernel<OP, VM><<<dim3((unsigned)grid), dim3(1024), lds, st>>>(p, k, v, n, thr, (int)G, RL, gsum, gcnt, pl->err, (int)pl->xform, vop);
            HIP_TRY(ctx, hipGetLastError());
            return HARK_OK;
        };
        if (!v)                                                          // COUNT only: no value column is read
            return dispatch_op(cmp, p != nullptr, [&](auto op) -> int { return launch(op, std::integral_constant<int, 2>{}); });
        if (vop == VOP_F32SUM && pl->xform == 0)
            return dispatch_op(cmp, p != nullptr, [&](auto op) -> int { return launch(op, std::integral_constant<int, 1>{}); });
        return dispatch_op(cmp, p != nullptr, [&](auto op) -> int { return launch(op, std::integral_constant<int, 0>{}); });
    }
    if (algo == 2) {
        int64_t grid = pl->grid ? pl->grid : (int64_t)ctx->num_cu * 8;
        const int64_t need = (n / kVec + 255) / 256;
        if (grid > need) grid = need > 0 ? need : 1;
        return dispatch_op(cmp, p != nullptr, [&](auto op) -> int {
            constexpr int OP = decltype(op)::value;
            TimedLaunch tl(pl, st, 0);
            fgb_atomic_kernel<OP><<<dim3((unsigned)grid), dim3(256), 0, st>>>(p, k, v, n, thr, G, gsum, gcnt, pl->err, vop, (int)pl->xform);
            HIP_TRY(ctx, hipGetLastError());
            return HARK_OK;
        });
    }
    // algo 3: a key column sorted / clustered by the key takes the window path (one pass, see fgb_window_kernel) ...
    {
        bool window = false;
        HARK_TRY(fgb_window_wanted(ctx, pl, k, n, &window));
        if (window) {
            const size_t lds = (size_t)(kWinKeys << kWinRL) * 12;
            int64_t grid = pl->grid ? pl->grid : (int64_t)ctx->num_cu;
            const int64_t need = (n / kVec + 1023) / 1024;
            if (grid > need) grid = need > 0 ? need : 1;
            auto launch = [&](auto op, auto vm) -> int {
                constexpr int OP = decltype(op)::value;
                constexpr int VM = decltype(vm)::value;
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_window_kernel<OP, VM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                TimedLaunch tl(pl, st, 1);
                fgb_window_kernel<OP, VM><<<dim3((unsigned)grid), dim3(1024), lds, st>>>(p, k, v, n, thr, G, gsum, gcnt, pl->err, (int)pl->xform, vop,
                                                                                        reinterpret_cast<uint32_t *>(pl->err + 4));
                HIP_TRY(ctx, hipGetLastError());
                return HARK_OK;
            };
            pl->win_rows += n;
            if (!v) return dispatch_op(cmp, p != nullptr, [&](auto op) -> int { return launch(op, std::integral_constant<int, 2>{}); });
            if (vop == VOP_F32SUM && pl->xform == 0)
                return dispatch_op(cmp, p != nullptr, [&](auto op) -> int { return launch(op, std::integral_constant<int, 1>{}); });
            return dispatch_op(cmp, p != nullptr, [&](auto op) -> int { return launch(op, std::integral_constant<int, 0>{}); });
        }
    }
    // ... everything else: partition + per-bucket LDS aggregation, chunked
    HARK_TRY(plan_prepare_partition(ctx, pl));
    // geometry: one-word ring entries in <= 128 buckets (pairfmt = 3) are no faster than the split rings in 256 buckets at the
    // headline's selectivity (2.57-2.69 against 2.46-2.63 ms per 1e9 rows, profiles/r03_notes.md) -- but when (nearly) every row
    // survives they are: 32 arrivals per ring and batch instead of 16, rings that hold two units, a sweep of 128 rings every
    // second batch instead of 256 every batch (producer 2.92 -> 2.65 ms per 1e9 rows without a predicate,
    // profiles/r06_nofilter_knobs.txt).  So: no predicate at all (BASELINE configs[2] as written, every reference
    // query_groupby), or a predicate that let >= 75 % of the rows through the last time this plan was checked.
    // COUNT-only keeps its keys-only format
    const bool crowded = p == nullptr || pl->sel_pct >= 75;
    const bool use8 = v != nullptr && pl->P8 > 0 && (pl->pairfmt == 3 || (pl->pairfmt == 0 && crowded && !getenv("HARK_FGB_NO_CROWDED")));
    pl->rows_fed += n;
    const int P = (int)(use8 ? pl->P8 : pl->P), shift = (int)(use8 ? pl->shift8 : pl->shift), nwg = (int)pl->nwg;
    const uint32_t cap = (uint32_t)(use8 ? pl->cap8 : pl->cap);
    const size_t lds_agg = (size_t)12 << shift;
    return dispatch_op(cmp, p != nullptr, [&](auto op) -> int {
        constexpr int OP = decltype(op)::value;
        if (lds_agg > 64 * 1024) {
            int rc = dispatch_vop(vop, [&](auto vopc) -> int {
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg_kernel<decltype(vopc)::value>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_agg));
                return HARK_OK;
            });
            if (rc) return rc;
        }
        const int fmt = !v ? 2 : use8 ? 3 : pl->pairfmt != 1 ? 1 : 0;   // keys only (COUNT) / 6-byte units from one-word ring entries / from split rings / 8-byte pairs on request
        const bool c6 = fmt == 1 || fmt == 3;                        // what the consumer reads
        const int split = (P <= kMaxBuckets / 2 && getenv("HARK_FGB_NOSPLIT") == nullptr) ? 2 : 1;
        const size_t lds_part = part_lds_bytes(P, fmt);
        const bool fast = !v || (vop == VOP_F32SUM && pl->xform == 0);   // the headline operator (and COUNT) is compiled in
        const void *fn = fmt == 2 ? reinterpret_cast<const void *>(&fgb_part_kernel<OP, 0, 2>)
                       : fmt == 3 ? (fast ? reinterpret_cast<const void *>(&fgb_part_kernel<OP, 0, 3>) : reinterpret_cast<const void *>(&fgb_part_kernel<OP, 1, 3>))
                       : fast ? (c6 ? reinterpret_cast<const void *>(&fgb_part_kernel<OP, 0, 1>) : reinterpret_cast<const void *>(&fgb_part_kernel<OP, 0, 0>))
                              : (c6 ? reinterpret_cast<const void *>(&fgb_part_kernel<OP, 1, 1>) : reinterpret_cast<const void *>(&fgb_part_kernel<OP, 1, 0>));
        HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
        if (lds_agg > 64 * 1024) {
            int rc6 = dispatch_vop(vop, [&](auto vopc) -> int {
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg6_kernel<decltype(vopc)::value>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_agg));
                return HARK_OK;
            });
            if (rc6) return rc6;
        }
        for (int64_t r0 = 0; r0 < n; r0 += pl->chunk_rows) {
            const int64_t r1 = r0 + pl->chunk_rows < n ? r0 + pl->chunk_rows : n;
            HIP_TRY(ctx, hipMemsetAsync(pl->err + 1, 0, 4, st));                 // the producers' batch counter
            {
                TimedLaunch tl(pl, st, 1);
                const int period = (int)pl->period;                            // batches between sweeps (0 = default)
                const int64_t rot = fgb_rot_of(pl, k, r1 - r0);
#define HARK_LAUNCH_PART(MODE, FMTV) do { \
                    if (rot > 0) { \
                        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_part_kernel<OP, MODE, FMTV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part)); \
                        fgb_part_kernel<OP, MODE, FMTV, true><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>( \
                            p, k, v, r0, r1, thr, G, shift, P, pl->pbuf, pl->counts, cap, gsum, gcnt, pl->err, period, vop, (int)pl->xform, 0, 0, rot); \
                    } else fgb_part_kernel<OP, MODE, FMTV><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>( \
                            p, k, v, r0, r1, thr, G, shift, P, pl->pbuf, pl->counts, cap, gsum, gcnt, pl->err, period, vop, (int)pl->xform, 0, 0, 0); } while (0)
                if (fmt == 2) HARK_LAUNCH_PART(0, 2);
                else if (fmt == 3) { if (fast) HARK_LAUNCH_PART(0, 3); else HARK_LAUNCH_PART(1, 3); }
                else if (fast) { if (c6) HARK_LAUNCH_PART(0, 1); else HARK_LAUNCH_PART(0, 0); }
                else { if (c6) HARK_LAUNCH_PART(1, 1); else HARK_LAUNCH_PART(1, 0); }
#undef HARK_LAUNCH_PART
            }
            HIP_TRY(ctx, hipGetLastError());
            {
                TimedLaunch tl(pl, st, 2);
                int rc = fmt == 2 ? [&]() -> int {
                    fgb_agg2_kernel<<<dim3((unsigned)P), dim3(1024), (size_t)4 << shift, st>>>(
                        reinterpret_cast<const unsigned char *>(pl->pbuf), pl->counts, cap, nwg, shift, G, gcnt);
                    return HARK_OK;
                }() : dispatch_vop(vop, [&](auto vopc) -> int {
                    if (c6)
                        fgb_agg6_kernel<decltype(vopc)::value><<<dim3((unsigned)(P * split)), dim3(1024), lds_agg, st>>>(
                            reinterpret_cast<const unsigned char *>(pl->pbuf), pl->counts, cap, nwg, shift, G, gsum, gcnt, split);
                    else
                        fgb_agg_kernel<decltype(vopc)::value><<<dim3((unsigned)P), dim3(1024), lds_agg, st>>>(
                            pl->pbuf, pl->counts, cap, nwg, shift, G, gsum, gcnt);
                    return HARK_OK;
                });
                if (rc) return rc;
            }
            HIP_TRY(ctx, hipGetLastError());
        }
        return HARK_OK;
    });
}

static int fgb_check_err(hark_context *ctx, hark_fgb_plan *pl)
{
    int32_t *e = reinterpret_cast<int32_t *>(ctx->h_pin);
    HIP_TRY(ctx, hipMemcpyAsync(e, pl->err, 32, hipMemcpyDeviceToHost, ctx->stream));   // the error word, the pairs partitioned so far, the window path's counters
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    {   // the window path since the last check: rows it could not keep in LDS (never reset on the device)
        const int64_t outside = (int64_t)(uint32_t)e[5] - (int64_t)(uint32_t)pl->win_outside_seen, rows = pl->win_rows - pl->win_rows_seen;
        pl->win_moves = (int64_t)(uint32_t)e[4];
        if (rows > 0 && (uint32_t)outside > (uint64_t)rows / 3) pl->win_verdict = 0;           // not a clustered column after all: the partition path from now on
        // (a third: stray rows cost two global atomics each, but a column that is four fifths in order still takes 46 ms per 1e9 rows through the partition, 18 here)
        pl->win_outside_seen = (uint32_t)e[5]; pl->win_rows_seen = pl->win_rows;
    }
    {   // the predicate's selectivity since the last check (the next run picks its geometry by it)
        const int64_t pairs = (int64_t)(((uint64_t)(uint32_t)e[3] << 32) | (uint32_t)e[2]), rows = pl->rows_fed - pl->rows_seen;
        if (rows > 0) pl->sel_pct = (pairs - pl->pairs_seen) * 100 / rows;
        pl->rows_seen = pl->rows_fed; pl->pairs_seen = pairs;
    }
    if (*e != 0) {
        const int code = *e;
        HIP_TRY(ctx, hipMemsetAsync(pl->err, 0, sizeof(int32_t), ctx->stream));
        return hark_fail(ctx, code, "filter_groupby: a surviving row has a key outside [0, %lld)", (long long)pl->G);
    }
    return HARK_OK;
}

// u32 flavour of the raw operator: opcode of the plan ("vop" knob: 1 sum, 2 max, 3 min, 4 prod;
// wrapping arithmetic as in groupby.fut:35-41), no predicate.
int hark_op_groupby_dense_u32(hark_context *ctx, hark_fgb_plan *pl, const uint32_t *k, const uint32_t *v, int64_t n)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl) return HARK_EARG;
    if (n < 0 || (n && (!k || !v))) return hark_fail(ctx, HARK_EARG, "groupby_dense_u32: null column");
    if (n > 0xFFFFFFFFll || (pl->max_rows && n > pl->max_rows)) return hark_fail(ctx, HARK_EARG, "groupby_dense_u32: too many rows for this plan");
    return k_fgb_dense_f32(ctx, pl, nullptr, 0, 0.0f, reinterpret_cast<const int32_t *>(k), reinterpret_cast<const float *>(v), n);
}

int hark_fgb_finish_u32(hark_context *ctx, hark_fgb_plan *pl, uint32_t *val_out, int64_t *count_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl) return HARK_EARG;
    if (val_out || count_out) {
        int64_t blocks = (pl->G + 255) / 256;
        if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
        fgb_finish_u32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(reinterpret_cast<const u64 *>(pl->acc_sum), pl->acc_cnt, pl->G, val_out, count_out);
        HIP_TRY(ctx, hipGetLastError());
    }
    return fgb_check_err(ctx, pl);
}

// the low words of one of the plan's accumulators after a statistics pass (which: 0 sums, 1 minima, 2 maxima), groups in table order
int hark_fgb_finish_u32_of(hark_context *ctx, hark_fgb_plan *pl, int which, uint32_t *val_out)
{
    hark_device_guard guard__(ctx);
    const u64 *acc = !pl ? nullptr : which == 0 ? reinterpret_cast<const u64 *>(pl->acc_sum) : which == 1 ? pl->acc_min : pl->acc_max;
    if (!ctx || !pl || !val_out || !acc) return HARK_EARG;
    int64_t blocks = (pl->G + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
    fgb_finish_u32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(acc, pl->acc_cnt, pl->G, val_out, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

// the low words of the plan's SECOND accumulator (acc_min: value 2 of a pair pass), groups in table order
int hark_fgb_finish_u32_second(hark_context *ctx, hark_fgb_plan *pl, uint32_t *val_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl || !val_out || !pl->acc_min) return HARK_EARG;
    int64_t blocks = (pl->G + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
    fgb_finish_u32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(pl->acc_min, pl->acc_cnt, pl->G, val_out, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

// Stream-ordered read-out without the host round trip: the sticky error word stays on the device until hark_fgb_check
// (or a later hark_fgb_finish) reads it -- a loop of steps then never waits for the host between steps.
int hark_fgb_finish_async(hark_context *ctx, hark_fgb_plan *pl, float *sum_out, int64_t *count_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl) return HARK_EARG;
    if (pl->vop != VOP_F32SUM) return hark_fail(ctx, HARK_EARG, "fgb_finish: this plan accumulates a u32 operator, use hark_fgb_finish_u32");
    int64_t blocks = (pl->G + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
    if (sum_out || count_out) {
        fgb_finish_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(pl->acc_sum, pl->acc_cnt, pl->G, sum_out, count_out);
        HIP_TRY(ctx, hipGetLastError());
    }
    return HARK_OK;
}

int hark_fgb_finish(hark_context *ctx, hark_fgb_plan *pl, float *sum_out, int64_t *count_out)
{
    HARK_TRY(hark_fgb_finish_async(ctx, pl, sum_out, count_out));
    return hark_fgb_check(ctx, pl);
}

// Sum of the event-timed kernel durations since the last call, by kernel kind
// (0 = single-kernel path, 1 = partition producer, 2 = partition consumer).
int hark_fgb_timing(hark_context *ctx, hark_fgb_plan *pl, double *ms_by_kind, int64_t *launches_by_kind)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl || !ms_by_kind || !launches_by_kind) return HARK_EARG;
    for (int i = 0; i < 3; i++) { ms_by_kind[i] = 0.0; launches_by_kind[i] = 0; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t s = 0; s + 1 < pl->ev_used; s += 2) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pl->ev[s], pl->ev[s + 1]));
        const int kind = pl->ev_kind[s / 2];
        ms_by_kind[kind] += ms; launches_by_kind[kind] += 1;
    }
    pl->ev_used = 0;
    return HARK_OK;
}

int hark_fgb_finish_typed(hark_context *ctx, hark_fgb_plan *pl, int32_t kind, const uint32_t *pos, void *out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl || !out || kind < 0 || kind > 11 || kind == 6) return HARK_EARG;
    int64_t blocks = (pl->G + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
    fgb_decode_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(reinterpret_cast<const u64 *>(pl->acc_sum), pl->acc_cnt, pl->G, kind, pos, out);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;                       // the plan's error word is read ONCE per statement (hark_fgb_check), not after every read-out
}

// reads the plan's sticky device error word (a surviving row's key outside [0, G)): one small copy + stream synchronisation
int hark_fgb_check(hark_context *ctx, hark_fgb_plan *pl)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl) return HARK_EARG;
    return fgb_check_err(ctx, pl);
}

// One pass for SUM / COUNT / AVG + MIN + MAX of ONE 4-byte value column (vk: 0 f32, 1 i32, 2 u32).  Only the partition
// path with <= 4096 keys per bucket qualifies (20 B per key of LDS in the consumer); *ran is false otherwise, and also
// when the data overflowed a slab or a ring (heavy skew: this pass has no single-row fallback) -- the caller then runs
// the separate passes.  Accumulators: acc_sum / acc_cnt as usual, acc_min / acc_max as order words in 64-bit slots.
// the several-aggregates window pass over all n rows (the caller has initialised the accumulators)
static int fgb_run_windowx(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr, const int32_t *k, int64_t n, const WinAgg &A)
{
    const size_t lds = (size_t)(kWinKeys << kWinXRL) * 20;
    int64_t grid = pl->grid ? pl->grid : (int64_t)ctx->num_cu;
    const int64_t need = (n / kVec + 1023) / 1024;
    if (grid > need) grid = need > 0 ? need : 1;
    pl->win_rows += n;
    return dispatch_op(cmp, p != nullptr, [&](auto op) -> int {
        constexpr int OP = decltype(op)::value;
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_windowx_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        TimedLaunch tl(pl, ctx->stream, 1);
        fgb_windowx_kernel<OP><<<dim3((unsigned)grid), dim3(1024), lds, ctx->stream>>>(p, k, n, thr, pl->G, A, pl->acc_cnt, pl->err, reinterpret_cast<uint32_t *>(pl->err + 4));
        HIP_TRY(ctx, hipGetLastError());
        return HARK_OK;
    });
}

// after a window pass of the several-aggregates entries: the sticky error word (a key out of range), as their partition passes read it
static int fgb_window_done(hark_context *ctx, hark_fgb_plan *pl, bool *ran)
{
    int64_t e = 0;
    HARK_TRY(hark_read_words(ctx, pl->err, &e, 1));
    const int32_t code = (int32_t)(e & 0xFFFFFFFFll);
    if (code != 0) {
        HIP_TRY(ctx, hipMemsetAsync(pl->err, 0, sizeof(int32_t), ctx->stream));
        return hark_fail(ctx, code, "filter_groupby: a surviving row has a key outside [0, %lld)", (long long)pl->G);
    }
    *ran = true;
    return HARK_OK;
}

int k_fgb_dense_stats(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr,
                      const int32_t *k, const void *v, int64_t n, int vk, bool *ran)
{
    *ran = false;
    const int64_t G = pl->G;
    if (n <= 0 || pl->algo != 0 || G * 12 <= kLdsTableBudget || G > (int64_t)kMaxBuckets * 4096 || n > pl->max_rows) return HARK_OK;
    auto misaligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) != 0; };
    if ((p && misaligned(p)) || misaligned(k) || misaligned(v)) return HARK_OK;
    bool window = false;                                               // a key column sorted / clustered by the key: the window pass (fgb_windowx_kernel)
    HARK_TRY(fgb_window_wanted(ctx, pl, k, n, &window));
    if (!window) {
        HARK_TRY(plan_prepare_partition(ctx, pl));
        if (pl->shift > 12) return HARK_OK;
    }
    const int P = (int)pl->P, shift = (int)pl->shift, nwg = (int)pl->nwg;
    if (!pl->acc_min) HARK_TRY(hark_alloc(ctx, (void **)&pl->acc_min, (size_t)G * 8));
    if (!pl->acc_max) HARK_TRY(hark_alloc(ctx, (void **)&pl->acc_max, (size_t)G * 8));
    hipStream_t st = ctx->stream;
    const int64_t blocks = (G + 255) / 256 > (int64_t)ctx->num_cu * 4 ? (int64_t)ctx->num_cu * 4 : (G + 255) / 256;
    fgb_init_stats_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(reinterpret_cast<u64 *>(pl->acc_sum), pl->acc_cnt, pl->acc_min, pl->acc_max, G);   // one launch instead of three memsets and a fill
    u64 *gsum = reinterpret_cast<u64 *>(pl->acc_sum);
    if (window) {
        // vk 0: f32 (f64 sum of the values, extremes of their order words), 1: i32 (sum and extremes of the biased words), 2: u32
        const uint32_t *c = static_cast<const uint32_t *>(v);
        const int xo = vk == 0 ? XF_F32_ORDER : vk == 1 ? XF_I32_ORDER : XF_NONE;
        const WinAgg A = {c, c, c, vk == 0 ? (int)VOP_F32SUM : (int)VOP_U32SUM64, vk == 1 ? (int)XF_I32_ORDER : (int)XF_NONE, VOP_U32MIN, xo, VOP_U32MAX, xo, gsum, pl->acc_min, pl->acc_max};
        HARK_TRY(fgb_run_windowx(ctx, pl, p, cmp, thr, k, n, A));
        return fgb_window_done(ctx, pl, ran);
    }
    const size_t lds_agg = (size_t)20 << shift, lds_part = part_lds_bytes(P, 1);
    int rc = dispatch_op(cmp, p != nullptr, [&](auto op) -> int {
        constexpr int OP = decltype(op)::value;
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_part_kernel<OP, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
        for (int64_t r0 = 0; r0 < n; r0 += pl->chunk_rows) {
            const int64_t r1 = r0 + pl->chunk_rows < n ? r0 + pl->chunk_rows : n;
            HIP_TRY(ctx, hipMemsetAsync(pl->err + 1, 0, 4, st));                 // the producers' batch counter
            {
                TimedLaunch tl(pl, st, 1);
                const int64_t rot = fgb_rot_of(pl, k, r1 - r0);
                if (rot > 0) {
                    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_part_kernel<OP, 0, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
                    fgb_part_kernel<OP, 0, 1, true><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>(
                        p, k, static_cast<const float *>(v), r0, r1, thr, G, shift, P, pl->pbuf, pl->counts, (uint32_t)pl->cap, gsum, pl->acc_cnt, pl->err, 0, 0, 0, 0, 1, rot);
                } else
                fgb_part_kernel<OP, 0, 1><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>(
                    p, k, static_cast<const float *>(v), r0, r1, thr, G, shift, P, pl->pbuf, pl->counts, (uint32_t)pl->cap, gsum, pl->acc_cnt, pl->err, 0, 0, 0, 0, 1, 0);
            }
            HIP_TRY(ctx, hipGetLastError());
            TimedLaunch tl(pl, st, 2);
#define HARK_STATS(VK) do { \
                if (lds_agg > 64 * 1024) HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg6_stats_kernel<VK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_agg)); \
                fgb_agg6_stats_kernel<VK><<<dim3((unsigned)P), dim3(1024), lds_agg, st>>>(reinterpret_cast<const unsigned char *>(pl->pbuf), pl->counts, \
                    (uint32_t)pl->cap, nwg, shift, G, gsum, pl->acc_cnt, pl->acc_min, pl->acc_max); } while (0)
            if (vk == 0) HARK_STATS(0); else if (vk == 1) HARK_STATS(1); else HARK_STATS(2);
#undef HARK_STATS
            HIP_TRY(ctx, hipGetLastError());
        }
        return HARK_OK;
    });
    if (rc) return rc;
    int64_t e = 0;
    HARK_TRY(hark_read_words(ctx, pl->err, &e, 1));                // err is an int32 in a >= 8-byte pool block
    const int32_t code = (int32_t)(e & 0xFFFFFFFFll);
    if (code != 0) HIP_TRY(ctx, hipMemsetAsync(pl->err, 0, sizeof(int32_t), st));
    if (code == kErrOverflow) return HARK_OK;                      // *ran stays false: the caller runs the separate passes
    if (code != 0) return hark_fail(ctx, code, "filter_groupby: a surviving row has a key outside [0, %lld)", (long long)G);
    *ran = true;
    return HARK_OK;
}

// Two or three aggregates of different 4-byte columns in ONE pass (see fgb_partv_kernel): value 1 with operator vop1 / xf1
// into acc_sum, value 2 with vop2 in {U32MAX, U32MIN} / xf2 into acc_min (used as the plan's second accumulator), value 3
// (v3 != nullptr: triple pass) likewise into acc_max, row counts into acc_cnt.  Only the partition path with <= 4096 keys
// per bucket qualifies; *ran is false otherwise, and when skew overflowed a ring or a slab (no single-row fallback; a
// triple's 14-byte entries fill the plan's slabs -- 10.4 B per row -- from ~70 % selectivity on): the caller runs the
// smaller passes.
int k_fgb_dense_multi(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr, const int32_t *k,
                      const void *v1, int vop1, int xf1, const void *v2, int vop2, int xf2, const void *v3, int vop3, int xf3, int64_t n, bool *ran)
{
    *ran = false;
    const int64_t G = pl->G;
    const int nv = v3 ? 3 : 2;
    if (getenv("HARK_NO_PAIR_PASS") || (nv == 3 && getenv("HARK_NO_TRIPLE_PASS"))) return HARK_OK;     // A/B knobs
    if (n <= 0 || pl->algo != 0 || G * 12 <= kLdsTableBudget || G > (int64_t)kMaxBuckets * 4096 || n > pl->max_rows) return HARK_OK;
    auto is_ext = [](int vop) { return vop == VOP_U32MAX || vop == VOP_U32MIN; };
    if (!(vop1 == VOP_F32SUM || vop1 == VOP_U32SUM64 || vop1 == VOP_U32SUM || is_ext(vop1)) || !is_ext(vop2) || (v3 && !is_ext(vop3))) return HARK_OK;
    auto misaligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) != 0; };
    if ((p && misaligned(p)) || misaligned(k) || misaligned(v1) || misaligned(v2) || (v3 && misaligned(v3))) return HARK_OK;
    bool window = false;                                               // a key column sorted / clustered by the key: the window pass (fgb_windowx_kernel)
    HARK_TRY(fgb_window_wanted(ctx, pl, k, n, &window));
    if (!window) {
        HARK_TRY(plan_prepare_partition(ctx, pl));
        if (pl->shift > 12) return HARK_OK;
    }
    // half as many buckets as the plan's one-value passes, in the same workspace
    const int shift = (int)pl->shift + 1, P = window ? 1 : (int)(((G - 1) >> shift) + 1), nwg = (int)pl->nwg;
    if (P > kPairBuckets) return HARK_OK;
    const size_t slab_bytes = window ? 0 : ((size_t)pl->P * (size_t)pl->cap * 8 / (size_t)P) & ~(size_t)15;
    if (!window && slab_bytes < (size_t)4 * multi_unit_bytes(nv)) return HARK_OK;
    if (!pl->acc_min) HARK_TRY(hark_alloc(ctx, (void **)&pl->acc_min, (size_t)G * 8));
    if (v3 && !pl->acc_max) HARK_TRY(hark_alloc(ctx, (void **)&pl->acc_max, (size_t)G * 8));
    hipStream_t st = ctx->stream;
    const int64_t blocks = (G + 255) / 256 > (int64_t)ctx->num_cu * 4 ? (int64_t)ctx->num_cu * 4 : (G + 255) / 256;
    u64 *gsum = reinterpret_cast<u64 *>(pl->acc_sum);
    fgb_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(gsum, G, vop_identity(vop1));
    HIP_TRY(ctx, hipMemsetAsync(pl->acc_cnt, 0, (size_t)G * 8, st));
    fgb_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(pl->acc_min, G, vop_identity(vop2));
    if (v3) fgb_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(pl->acc_max, G, vop_identity(vop3));
    if (window) {
        if (vop1 == VOP_U32PROD) return HARK_OK;
        const WinAgg A = {static_cast<const uint32_t *>(v1), static_cast<const uint32_t *>(v2), static_cast<const uint32_t *>(v3), vop1, xf1, vop2, xf2, v3 ? vop3 : (int)VOP_U32MAX, xf3,
                          gsum, pl->acc_min, pl->acc_max};
        HARK_TRY(fgb_run_windowx(ctx, pl, p, cmp, thr, k, n, A));
        return fgb_window_done(ctx, pl, ran);
    }
    const size_t lds_agg = (size_t)(nv == 3 ? 20 : 16) << shift, lds_part = partv_lds_bytes(P, nv);
    const uint32_t inv2 = vop2 == VOP_U32MIN ? 0xFFFFFFFFu : 0u, inv3 = vop3 == VOP_U32MIN ? 0xFFFFFFFFu : 0u;
    unsigned char *pbuf = reinterpret_cast<unsigned char *>(pl->pbuf);
    const uint32_t *c1 = static_cast<const uint32_t *>(v1), *c2 = static_cast<const uint32_t *>(v2), *c3 = static_cast<const uint32_t *>(v3);
    auto run = [&](auto op, auto nvc) -> int {
        constexpr int OP = decltype(op)::value, NV = decltype(nvc)::value;
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_partv_kernel<OP, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
        for (int64_t r0 = 0; r0 < n; r0 += pl->chunk_rows) {
            const int64_t r1 = r0 + pl->chunk_rows < n ? r0 + pl->chunk_rows : n;
            HIP_TRY(ctx, hipMemsetAsync(pl->err + 1, 0, 4, st));                 // the producers' batch counter
            {
                TimedLaunch tl(pl, st, 1);
                const int64_t rot = fgb_rot_of(pl, k, r1 - r0);
                if (rot > 0) {
                    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_partv_kernel<OP, NV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
                    fgb_partv_kernel<OP, NV, true><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>(p, k, c1, c2, c3, r0, r1, thr, G, shift, P, pbuf, pl->counts, slab_bytes, pl->err, rot);
                } else
                fgb_partv_kernel<OP, NV><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>(
                    p, k, c1, c2, c3, r0, r1, thr, G, shift, P, pbuf, pl->counts, slab_bytes, pl->err, 0);
            }
            HIP_TRY(ctx, hipGetLastError());
            TimedLaunch tl(pl, st, 2);
            int rc2 = dispatch_vop(vop1, [&](auto v) -> int {
                constexpr int V1 = decltype(v)::value;
                if constexpr (V1 == VOP_U32PROD) return HARK_EUNSUPPORTED;
                else {
                    if (lds_agg > 64 * 1024) HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_aggv_kernel<V1, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_agg));
                    fgb_aggv_kernel<V1, NV><<<dim3((unsigned)(2 * P)), dim3(1024), lds_agg, st>>>(pbuf, pl->counts, slab_bytes, nwg, shift, G, gsum, pl->acc_cnt,
                                                                                                  pl->acc_min, pl->acc_max, xf1, xf2, xf3, inv2, inv3);
                    return HARK_OK;
                }
            });
            if (rc2) return rc2;
            HIP_TRY(ctx, hipGetLastError());
        }
        return HARK_OK;
    };
    int rc = dispatch_op(cmp, p != nullptr, [&](auto op) -> int {
        return nv == 3 ? run(op, std::integral_constant<int, 3>{}) : run(op, std::integral_constant<int, 2>{});
    });
    if (rc) return rc;
    int64_t e = 0;
    HARK_TRY(hark_read_words(ctx, pl->err, &e, 1));                // err is an int32 in a >= 8-byte pool block
    const int32_t code = (int32_t)(e & 0xFFFFFFFFll);
    if (code != 0) HIP_TRY(ctx, hipMemsetAsync(pl->err, 0, sizeof(int32_t), st));
    if (code == kErrOverflow) return HARK_OK;                      // *ran stays false: the caller runs the smaller passes
    if (code != 0) return hark_fail(ctx, code, "filter_groupby: a surviving row has a key outside [0, %lld)", (long long)G);
    *ran = true;
    return HARK_OK;
}

int k_fgb_dense_pair(hark_context *ctx, hark_fgb_plan *pl, const float *p, int cmp, float thr, const int32_t *k,
                     const void *v1, int vop1, int xf1, const void *v2, int vop2, int xf2, int64_t n, bool *ran)
{
    return k_fgb_dense_multi(ctx, pl, p, cmp, thr, k, v1, vop1, xf1, v2, vop2, xf2, nullptr, VOP_U32MAX, 0, n, ran);
}

// Typed read-out of one of the plan's accumulator arrays: which = 0 acc_sum, 1 acc_min, 2 acc_max.
int hark_fgb_finish_typed_from(hark_context *ctx, hark_fgb_plan *pl, int32_t which, int32_t kind, const uint32_t *pos, void *out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !pl || !out || kind < 0 || kind > 11 || kind == 6 || which < 0 || which > 2) return HARK_EARG;
    const u64 *acc = which == 0 ? reinterpret_cast<const u64 *>(pl->acc_sum) : which == 1 ? pl->acc_min : pl->acc_max;
    if (!acc) return hark_fail(ctx, HARK_EARG, "fgb: this plan has no min/max accumulators");
    int64_t blocks = (pl->G + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
    fgb_decode_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(acc, pl->acc_cnt, pl->G, kind, pos, out);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

// GROUP BY over arbitrary u32 keys with one value operator: hash-partition (producer in hash
// mode) + LDS hash tables (fgb_agg_hash_kernel).  Returns UNORDERED (key, value slot, count)
// arrays on the device (caller frees with hark_free) and the number of groups.  *fits is false
// when the data overflowed the geometry (heavy skew, or more distinct keys than 64 rounds hold):
// the caller then uses the sort-based path.
int k_fgb_hash_u32(hark_context *ctx, const uint32_t *k, const uint32_t *v, int64_t n, int vop, int xf,
                   uint32_t **keys_out, unsigned long long **vals_out, unsigned long long **cnts_out, int64_t *G_out, bool *fits,
                   uint32_t *rounds_hint /* in: 0 or the R a previous pass over the SAME key column needed; out: the R used */,
                   bool compact /* ONE u32 operator without row counts (fgb_agg_hash_ops_kernel<1>); *cnts_out stays null */,
                   hark_hash_part *part /* optional: the partition of (k, v) is kept in it / taken from it (see hark_internal.h) */,
                   int *why_not /* optional: why *fits is false -- HARK_HASH_NOFIT_* (hark_internal.h) */,
                   const hark_row_pred *pred /* optional WHERE, fused into the producer: an f32 column with a comparison, or a
                                                survivor bitmask with HARK_CMP_MASK -- only surviving rows are partitioned */,
                   int stats_vk /* >= 0: ONE statistics pass (fgb_agg_hash_stats_kernel) over raw values of kind 0 f32 / 1 i32 / 2 u32:
                                    *vals_out = 64-bit sums, *mins_out / *maxs_out = order words; vop / xf / compact are ignored */,
                   unsigned long long **mins_out, unsigned long long **maxs_out,
                   uint32_t ref_ops /* != 0: two or three of the reference's u32 operators over v in ONE pass (fgb_agg_hash_ops_kernel): operator of
                                       slot j in byte j; *vals_out = slot 0 | slot 1 << 32, *cnts_out = slot 2; vop / xf / compact are ignored */)
{
    const bool stats = stats_vk >= 0;
    const int nops = ref_ops == 0 ? 0 : (ref_ops >> 16) ? 3 : 2;
    if (stats) { compact = false; *mins_out = nullptr; *maxs_out = nullptr; }
    if (nops) { compact = false; xf = 0; }
    const float *pp = pred ? pred->p : nullptr;
    const int pcmp = pred ? pred->cmp : 0;
    const float pthr = pred ? pred->thr : 0.0f;
    if (why_not) *why_not = HARK_HASH_FITS;
    if (compact && (xf != 0 || !(vop == VOP_U32SUM || vop == VOP_U32MAX || vop == VOP_U32MIN || vop == VOP_U32PROD))) compact = false;
    const int fill = stats ? kHashSFill : nops == 2 ? HashOpsGeo<2>::fill : nops == 3 ? HashOpsGeo<3>::fill : compact ? HashOpsGeo<1>::fill : kHashFill;
    *keys_out = nullptr; *vals_out = nullptr; *cnts_out = nullptr; *G_out = 0; *fits = false;
    if (n <= 0 || n > 0xFFFFFFFFll) { if (why_not) *why_not = HARK_HASH_NOFIT_ROWS; return HARK_OK; }
    // 512 buckets of 32-pair rings (256 x 64 before): half as many distinct keys per bucket -- 2^21 distinct keys fit ONE
    // round of the 8-byte tables instead of two (plus the failed first attempt and the sample round that found that out)
    const int hash_bits = getenv("HARK_HASH_BITS8") ? 8 : 9, P = 1 << hash_bits, nwg = ctx->num_cu;
    // TagGroups (the consumers' key identities) keeps the 32 - hash_bits low bits of a mixed key in 24-bit multiplies and 16-bit
    // tags: fewer than 256 buckets would merge distinct keys into one identity without any error
    static_assert(32 - 8 <= 24, "TagGroups::home multiplies 24-bit words");
    if (P < 256) return hark_fail(ctx, HARK_EARG, "hash group-by: at least 256 buckets (TagGroups holds 24 low key bits)");
    int64_t cap = n / ((int64_t)P * nwg) * 130 / 100 + 256;
    cap = (cap + kLine - 1) / kLine * kLine + 2 * kLine;
    uint2 *pbuf = nullptr; uint32_t *counts = nullptr; int32_t *err = nullptr; unsigned long long *cursor = nullptr;
    uint32_t *okey = nullptr; u64 *oval = nullptr, *ocnt = nullptr, *omin = nullptr, *omax = nullptr;
    hipStream_t st = ctx->stream;
    const bool reuse = part && part->pbuf && part->k == k && part->v == v && part->n == n && part->cap == cap && part->p == pp;
    if (part && part->pbuf && !reuse) k_fgb_hash_part_free(ctx, part);
    int rc = HARK_OK;
    if (reuse) { pbuf = static_cast<uint2 *>(part->pbuf); counts = part->counts; }
    else {
        rc = hark_alloc(ctx, (void **)&pbuf, (size_t)P * nwg * (size_t)cap * sizeof(uint2));
        if (!rc) rc = hark_alloc(ctx, (void **)&counts, (size_t)P * nwg * sizeof(uint32_t));
    }
    // ONE block of status words, read back in ONE round trip per aggregation attempt (each costs ~25 us of idle device):
    // word 0 the producer's error, word 1 the consumers', word 2 the output cursor (= groups emitted)
    int32_t *perr = nullptr;
    if (!rc) rc = hark_alloc(ctx, (void **)&perr, 32);
    if (!rc) { err = perr + 2; cursor = reinterpret_cast<unsigned long long *>(perr) + 2; }
    int64_t emitted = 0;
    int32_t e = 0, e_prod = 0;
    auto read_status = [&]() -> int {
        int64_t w[3] = {0, 0, 0};
        int r2 = hark_read_words(ctx, perr, w, 3);
        e_prod = (int32_t)w[0]; e = e_prod ? e_prod : (int32_t)w[1]; emitted = w[2];
        return r2;
    };
    uint32_t used_R = 1;
    HIP_TRY_RC(ctx, rc, hipMemsetAsync(perr, 0, 32, st));
    auto launch_producer = [&]() -> int {
        const size_t lds_part = part_lds_bytes(P, 0);
        return dispatch_op(pcmp, pp != nullptr, [&](auto op) -> int {
            constexpr int OP = decltype(op)::value;
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_part_kernel<OP, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part));
            fgb_part_kernel<OP, 2, 0><<<dim3((unsigned)nwg), dim3(kPartThreads), lds_part, st>>>(
                pp, reinterpret_cast<const int32_t *>(k), reinterpret_cast<const float *>(v), 0, n, pthr, (int64_t)1 << 32, 0, P,
                pbuf, counts, (uint32_t)cap, nullptr, nullptr, perr, 0, vop, xf, hash_bits, 0);
            HIP_TRY(ctx, hipGetLastError());
            return HARK_OK;
        });
    };
    // (the producer's error word is read together with the first aggregation's: a slab or ring overflow -- skewed keys --
    // costs one wasted consumer pass on the way to the sort-based path instead of a round trip on every call)
    if (!rc && !reuse) rc = launch_producer();
    if (!rc) {
        const size_t lds_hash = (size_t)kHashCap * 14;                 // 8-byte value slot + 4-byte count + 2-byte tag per entry
        // run all rounds of an R-round aggregation; e != 0 afterwards means some table overflowed
        auto run_rounds = [&](uint32_t R, uint32_t r_begin, uint32_t r_end) -> int {
            const unsigned long long out_cap = (unsigned long long)P * fill * (r_end - r_begin);
            hark_free(ctx, okey); hark_free(ctx, oval); hark_free(ctx, ocnt); hark_free(ctx, omin); hark_free(ctx, omax);
            okey = nullptr; oval = nullptr; ocnt = nullptr; omin = nullptr; omax = nullptr;
            int r2 = hark_alloc(ctx, (void **)&okey, (size_t)out_cap * 4);
            if (!r2) r2 = hark_alloc(ctx, (void **)&oval, (size_t)out_cap * 8);
            if (!r2 && !compact) r2 = hark_alloc(ctx, (void **)&ocnt, (size_t)out_cap * 8);
            if (!r2 && stats) r2 = hark_alloc(ctx, (void **)&omin, (size_t)out_cap * 8);
            if (!r2 && stats) r2 = hark_alloc(ctx, (void **)&omax, (size_t)out_cap * 8);
            if (r2) return r2;
            HIP_TRY(ctx, hipMemsetAsync(perr + 2, 0, 24, st));        // the consumers' error word and the cursor; the producer's word stays
            if (stats) {
                const size_t lds_s = (size_t)kHashSCap * 22;          // sum 8 + count, smallest, largest 4 each + a 2-byte tag
                auto go = [&](auto vkc) -> int {
                    constexpr int VK = decltype(vkc)::value;
                    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg_hash_stats_kernel<VK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
                    for (uint32_t r = r_begin; r < r_end; r++)
                        fgb_agg_hash_stats_kernel<VK><<<dim3((unsigned)P), dim3(1024), lds_s, st>>>(pbuf, counts, (uint32_t)cap, nwg, R - 1, r, okey, oval, ocnt, omin, omax, cursor, out_cap, err);
                    HIP_TRY(ctx, hipGetLastError());
                    return HARK_OK;
                };
                r2 = stats_vk == 0 ? go(std::integral_constant<int, 0>{}) : stats_vk == 1 ? go(std::integral_constant<int, 1>{}) : go(std::integral_constant<int, 2>{});
                if (!r2) r2 = read_status();
                return r2;
            }
            if (nops) {
                auto go = [&](auto nc) -> int {
                    constexpr int NOPS = decltype(nc)::value;
                    const size_t lds_o = (size_t)HashOpsGeo<NOPS>::cap * (2 + 4 * NOPS);   // 16-bit tag + one 32-bit slot per operator
                    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg_hash_ops_kernel<NOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_o));
                    for (uint32_t r = r_begin; r < r_end; r++)
                        fgb_agg_hash_ops_kernel<NOPS><<<dim3((unsigned)P), dim3(1024), lds_o, st>>>(pbuf, counts, (uint32_t)cap, nwg, R - 1, r, okey, oval, ocnt, cursor, out_cap, err, ref_ops);
                    HIP_TRY(ctx, hipGetLastError());
                    return HARK_OK;
                };
                r2 = nops == 2 ? go(std::integral_constant<int, 2>{}) : go(std::integral_constant<int, 3>{});
                if (!r2) r2 = read_status();
                return r2;
            }
            r2 = dispatch_vop(vop, [&](auto vopc) -> int {
                constexpr int VOP = decltype(vopc)::value;
                if constexpr (VOP == VOP_U32SUM || VOP == VOP_U32MAX || VOP == VOP_U32MIN || VOP == VOP_U32PROD) {
                    if (compact) {                                            // one operator: the multi-operator consumer with one slot per entry (round 5:
                                                                              // 245 us per 1e8 pairs; the 8-byte-entry consumer it replaces took 435)
                        const size_t lds_o = (size_t)HashOpsGeo<1>::cap * 6;
                        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg_hash_ops_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_o));
                        for (uint32_t r = r_begin; r < r_end; r++)
                            fgb_agg_hash_ops_kernel<1><<<dim3((unsigned)P), dim3(1024), lds_o, st>>>(pbuf, counts, (uint32_t)cap, nwg, R - 1, r, okey, oval, nullptr, cursor, out_cap, err, (uint32_t)VOP);
                        HIP_TRY(ctx, hipGetLastError());
                        return HARK_OK;
                    }
                }
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&fgb_agg_hash_kernel<VOP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hash));
                for (uint32_t r = r_begin; r < r_end; r++)
                    fgb_agg_hash_kernel<VOP><<<dim3((unsigned)P), dim3(1024), lds_hash, st>>>(pbuf, counts, (uint32_t)cap, nwg, R - 1, r, okey, oval, ocnt, cursor, out_cap, err, xf);   // R is a power of two
                HIP_TRY(ctx, hipGetLastError());
                return HARK_OK;
            });
            if (!r2) r2 = read_status();
            return r2;
        };
        const uint32_t hint = rounds_hint ? *rounds_hint : 0u;
        constexpr uint32_t kMaxRoundsWorth = 16;
        used_R = hint ? hint : 1u;
        rc = run_rounds(used_R, 0, used_R);
        if (!rc && e_prod != 0 && !reuse) {
            // A slab overflowed.  Skewed keys do that -- and so does the FIRST pass over freshly allocated slabs: the workgroups
            // that fault the new pages in fall behind, the others draw their batches (first come, first served) and fill
            // slabs sized for 1.3 x an even share of the table's rows.  Found in round 5: the first sparse-key statement of every
            // process ran producer + consumer + the whole sort path (46 ms per 1e9 rows against 4 ms from the second on).
            // The pages are mapped now: once more, and only a second overflow says "skew".
            HIP_TRY_RC(ctx, rc, hipMemsetAsync(perr, 0, 32, st));
            if (!rc) rc = launch_producer();
            if (!rc) rc = run_rounds(used_R, 0, used_R);
        }
        if (!rc && e_prod != 0 && why_not) *why_not = HARK_HASH_NOFIT_SKEW;            // a slab or a ring overflowed (twice): the keys are skewed
        // a hint is the R that ANOTHER pass over this key column needed, and the consumers' tables differ in size (6144 keys per
        // bucket and round with one or two operators, 4608 with three, 3072 typed, 2560 statistics): a hinted R that overflows
        // is doubled until it fits, instead of declaring the column unfit for the hash path for good
        while (!rc && e != 0 && hint && e_prod == 0 && used_R < kMaxRoundsWorth) { used_R *= 2; rc = run_rounds(used_R, 0, used_R); }
        if (!rc && e != 0 && !hint && e_prod == 0) {
            // too many distinct keys for one round: estimate them from ONE round of a 64-round split (a 1/64 sample
            // of the key space), then run exactly the number of rounds that needs -- or give up right away
            rc = run_rounds(64, 0, 1);
            const int64_t sample = emitted;
            if (!rc && e == 0) {
                const double per_bucket = 64.0 * (double)sample / P * 1.3;
                // every round re-reads the bucket's slabs (~0.35 ms per round and 1e8 rows): beyond kMaxRoundsWorth rounds the
                // sort-based path (~7 ms per 1e8 rows) is the faster one -- with 512 buckets 64 rounds would "fit" 2e8 keys
                uint32_t R = 2;
                while (R < kMaxRoundsWorth && per_bucket > (double)fill * R) R *= 2;
                if (per_bucket > (double)fill * kMaxRoundsWorth) e = kErrOverflow;
                else {
                    rc = run_rounds(R, 0, R); used_R = R;
                    if (!rc && e != 0 && R < kMaxRoundsWorth) { rc = run_rounds(R * 2, 0, R * 2); used_R = R * 2; }   // one retry for uneven buckets
                }
            }
        }
    }
    if (!rc && e != 0 && why_not && *why_not == HARK_HASH_FITS) *why_not = HARK_HASH_NOFIT_DISTINCT;   // the tables overflowed
    if (!rc && e == 0) {
        const int64_t G = emitted;
        if (!rc) {
            *keys_out = okey; *vals_out = oval; *cnts_out = ocnt; *G_out = G; *fits = true; okey = nullptr; oval = nullptr; ocnt = nullptr;
            if (stats) { *mins_out = omin; *maxs_out = omax; omin = nullptr; omax = nullptr; }
        }
        if (!rc && rounds_hint) *rounds_hint = used_R;
    }
    if (part && !rc && (reuse || e == 0)) {                        // a good partition stays with the caller
        part->pbuf = pbuf; part->counts = counts; part->cap = cap; part->n = n; part->k = k; part->v = v; part->xf = xf; part->p = pp;
    } else {
        if (part) { part->pbuf = nullptr; part->counts = nullptr; }
        hark_free(ctx, pbuf); hark_free(ctx, counts);
    }
    hark_free(ctx, perr);
    hark_free(ctx, okey); hark_free(ctx, oval); hark_free(ctx, ocnt); hark_free(ctx, omin); hark_free(ctx, omax);
    return rc;
}

void k_fgb_hash_part_free(hark_context *ctx, hark_hash_part *part)
{
    if (!part) return;
    hark_free(ctx, part->pbuf); hark_free(ctx, part->counts);
    part->pbuf = nullptr; part->counts = nullptr;
}

// Typed read-out (fgb_decode_kernel kinds) of plain accumulator arrays, e.g. the sorted output of k_fgb_hash_u32.
int k_fgb_decode(hark_context *ctx, const unsigned long long *acc, const unsigned long long *cnt, int64_t G, int kind, void *out)
{
    if (G <= 0) return HARK_OK;
    int64_t blocks = (G + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 4) blocks = (int64_t)ctx->num_cu * 4;
    fgb_decode_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(acc, cnt, G, kind, nullptr, out);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}
