// k_cjoin.hip -- the join for a probe column that is SORTED or CLUSTERED by the join key (a fact table kept in key order, the
// output of a GROUP BY joined back).  Replaces futhark/join.fut:58-68 (the merge of the two sorted sides) for such inputs.
//
// Why a path of its own: the partitioned join (k_hjoin.hip) routes every probe row through per-bucket LDS rings into
// workgroup-private slabs, both sized for rows that SCATTER over the 512 buckets.  Consecutive rows of a sorted column fall into
// ONE bucket: a batch of 4096 rows then sweeps a ring of 32 entries 128 times and overruns the workgroup's slab of that bucket
// -- 1e8 sorted probe rows took 58 ms (50 of them in the partition, then the sort-merge path) against 1.1 ms for the same rows
// shuffled (tools/join_cluster_probe.py).  What hurts there helps here: consecutive rows need NEIGHBOURING places of the sorted build
// side, so a plain search per row finds its lines in the caches.
//   1. cj_test_kernel (one workgroup, on the second stream under the build side's sort): 1024 pairs of probe rows against 1024
//      quantiles of a sample of the BUILD column -- rows half a batch apart, or four lanes apart, closer in key order than rows
//      anywhere apart: this path; neighbouring rows in one bucket but too far apart for it (a column sorted block by block): the
//      partition with rotated loads (jpart_kernel's ROT variant, k_hjoin.hip); else the plain partition.
//   2. cj_search_kernel: every WAVE takes stretches of 256 consecutive rows (no workgroup barrier once the index -- every
//      (s / 8192)-th build key, 32 KiB of LDS -- is staged): the smallest and largest key of the stretch find their segments of the
//      index with 64 lanes at once, one or two 64-probe steps over the sorted build keys narrow them down, the build entries between
//      the two (a sorted column: ~26) are loaded into the wave's 2 KiB of LDS and every row finds its rank by a search there; a
//      stretch that spans more than 512 build keys searches row by row through the index and memory.  Matching rows leave as
//      (rank, row) in ROW order into the stretch's own piece of a scratch array, with the stretch's count and first / last rank.
//   3. cj_scan1_kernel / cj_scan2_kernel: the stretches' offsets; and whether the ranks ascend over the whole column (a sorted
//      column: the output order (key, left row, right row) is then the row order and nothing needs sorting).
//   4. cj_emit_kernel: the stretches' rows to their places, with the partner counts and the carried / rank-ordered columns.
// A column that is clustered but not ascending (descending, sorted runs in shuffled order) gets the general ordering of the caller
// (two radix sorts over the MATCHING rows).
#include "hark_internal.h"

namespace {

constexpr int kCThreads = 1024, kCVec = 4, kCBatch = kCThreads * kCVec;
template <typename K> struct CIdx { static constexpr int N = 32768 / (int)sizeof(K); };   // entries of the LDS index over the build side (32 KiB)

// The test, against 1024 quantiles of the BUILD side (a sample of its unsorted column, sorted here by the workgroup -- shuffles
// inside the waves, LDS for partners 64 and more lanes away -- or read off the sorted keys when the test runs late):
//   * rows kCtFar places apart (half a batch of the partition) within kCtNear / 1024 of the build side -- a batch then spans at
//     most ~4 of the partition's 512 buckets (measured, 1e8 rows sorted block by block, tools/join_cluster_probe.py: blocks of
//     1e5 rows 3.4 ms partitioned / 10.3 searched, of 1e6 rows 27 / 2.8);
//   * or rows kCtClose places apart (four lanes of the search) within kCtKeys build keys (256 until the partition learnt to read with
//     rotated loads: blocks of 1e6 sorted rows, 160 keys from row to sixteenth row, are faster that way), the cell's width read off the
//     neighbouring quantiles -- a wave's rows then share a few lines of the build side (sorted runs of 256 rows in shuffled
//     order: 1.7 ms searched, while the partition overruns its slabs and the probe side gets sorted: 7.9; runs of 16 rows:
//     2.0 searched, 1.3 partitioned -- rows 16 apart lie in different runs there, and the verdict is "scattered").
// The second measure decides first (this path), then "neighbouring rows within a quarter of a bucket" (rotated loads), then the first.
// All against the same measure for rows ANYWHERE apart (the key of another pair): a column of few distinct keys, or one that
// mostly misses the build side's range, is close to itself everywhere.  One pair per thread: the kernel waits for ~3000
// address translations of rows all over the columns, not for its arithmetic (4096 pairs: 75 us).
constexpr int kCtSample = 1024, kCtPairs = 1024, kCtFar = 2048, kCtClose = 16, kCtNear = 4, kCtKeys = 32;

__device__ __forceinline__ uint32_t cj_mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

template <typename K> __device__ __forceinline__ K cj_shfl_up(K x, int d);
template <> __device__ __forceinline__ uint32_t cj_shfl_up<uint32_t>(uint32_t x, int d) { return (uint32_t)__shfl_up((int)x, d, 64); }
template <> __device__ __forceinline__ uint64_t cj_shfl_up<uint64_t>(uint64_t x, int d)
{
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
template <typename K> __device__ __forceinline__ K cj_shfl_xor(K x, int j);
template <> __device__ __forceinline__ uint32_t cj_shfl_xor<uint32_t>(uint32_t x, int j) { return (uint32_t)__shfl_xor((int)x, j, 64); }
template <> __device__ __forceinline__ uint64_t cj_shfl_xor<uint64_t>(uint64_t x, int j)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, j, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), j, 64);
    return ((uint64_t)hi << 32) | lo;
}

// keys: the probe column; build: the build side's key column as the tables hold it (sorted_build = 0: both compared as raw bit
// patterns -- for i64 keys that is the signed order rotated, which serves a test of closeness as well) or its sorted keys
// (sorted_build = 1: `bias` makes the probe keys comparable).
// verdict[0] = 1: clustered; [1] / [2]: pairs half a batch / anywhere apart in neighbouring cells, [3] / [4]: pairs four lanes /
// anywhere apart within kCtKeys build keys (diagnostics).  `verdict` may be pinned host memory.
template <typename K>
__global__ __launch_bounds__(1024) void cj_test_kernel(const K *__restrict__ keys, int64_t n, K bias, const K *__restrict__ build, int64_t s, int sorted_build,
                                                       unsigned long long *verdict)
{
    __shared__ K s_k[kCtSample], s_a[kCtPairs];
    __shared__ uint32_t s_cnt[6];
    const int tid = threadIdx.x;
    if (tid < 6) s_cnt[tid] = 0u;
    K x = sorted_build ? build[(int64_t)(((uint64_t)tid * (uint64_t)s) / kCtSample)] : build[(int64_t)(((uint64_t)cj_mix(2u * (uint32_t)tid + 1u) * (uint64_t)s) >> 32)];
    const int64_t r = (int64_t)(((uint64_t)cj_mix(0x9E3779B9u + (uint32_t)tid) * (uint64_t)(n - kCtFar)) >> 32);   // n >= 2^18 (the caller's threshold)
    const K ka = keys[r] ^ bias, kb = keys[r + kCtFar] ^ bias, kc = keys[r + kCtClose] ^ bias;   // in flight while the sample is sorted
    if (!sorted_build)
        for (int k = 2; k <= kCtSample; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                K y;
                if (j >= 64) { s_k[tid] = x; __syncthreads(); y = s_k[tid ^ j]; __syncthreads(); }
                else y = cj_shfl_xor<K>(x, j);
                const bool keep_min = ((tid & k) == 0) == ((tid & j) == 0);
                x = keep_min ? (y < x ? y : x) : (y > x ? y : x);
            }
    s_k[tid] = x; s_a[tid] = ka;
    __syncthreads();
    const K kf = s_a[(tid + kCtPairs / 2) & (kCtPairs - 1)];              // a row anywhere else: the first key of another pair
    auto quant = [&](K v) -> int {                                        // sample keys below v
        int pos = 0;
#pragma unroll
        for (int step = kCtSample / 2; step > 0; step >>= 1) if (s_k[pos + step - 1] < v) pos += step;
        return pos + (s_k[pos] < v ? 1 : 0);
    };
    const int qa = quant(ka), qb = quant(kb), qf = quant(kf);
    const int lo = max(qa - 2, 0), hi = min(qa + 2, kCtSample - 1);
    // kCtKeys build keys in key units around ka: a cell (s / 1024 build keys) is (s_k[hi] - s_k[lo]) / (hi - lo) wide; at most kCtNear cells
    const double cells = fmin((double)kCtKeys * (double)kCtSample / (double)s, (double)kCtNear);
    const double reach = (double)(s_k[hi] - s_k[lo]) / (double)(hi - lo) * cells;
    const K dc = ka > kc ? ka - kc : kc - ka, df = ka > kf ? ka - kf : kf - ka;
    const uint32_t near = abs(qa - qb) <= kCtNear, far = abs(qa - qf) <= kCtNear, fine = (double)dc <= reach, ffar = (double)df <= reach;
    // ... and within a quarter of a bucket of the partition (512 buckets: half a cell): neighbouring rows share a bucket
    const double breach = (double)(s_k[hi] - s_k[lo]) / (double)(hi - lo) * 0.5;
    const uint32_t bnear = (double)dc <= breach, bfar = (double)df <= breach;
    const unsigned long long m0 = __ballot(near), m1 = __ballot(far), m2 = __ballot(fine), m3 = __ballot(ffar), m4 = __ballot(bnear), m5 = __ballot(bfar);
    if ((tid & 63) == 0) {
        atomicAdd(&s_cnt[0], __popcll(m0)); atomicAdd(&s_cnt[1], __popcll(m1)); atomicAdd(&s_cnt[2], __popcll(m2)); atomicAdd(&s_cnt[3], __popcll(m3));
        atomicAdd(&s_cnt[4], __popcll(m4)); atomicAdd(&s_cnt[5], __popcll(m5));
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t a = s_cnt[0], b = s_cnt[1], c = s_cnt[2], d = s_cnt[3], e = s_cnt[4], f = s_cnt[5];
        verdict[1] = a; verdict[2] = b; verdict[3] = c; verdict[4] = d; verdict[5] = e; verdict[6] = f;
        // 1: the search path; else 2 when three of four neighbouring pairs share a bucket (and not because everything does): the
        // partition with rotated loads (a probe column sorted block by block: blocks of 2e5 ... 5e5 rows took 4.7-6.9 ms against 0.9)
        // measured (1e8 x 1e7, tools/join_cluster_probe.py with HARK_JOIN_CLUSTERED=1 / 2): sorted 0.88 ms searched / 1.53 rotated, sorted runs
        // of 256 rows 0.98 / 4.2 (the rotated partition overruns its slabs there), blocks of 1e6 rows 2.57 / 1.70, of 3e5 rows 6.80 / 1.65
        const bool strong = a > b && (a - b) * 8u > (uint32_t)kCtPairs, fine_ = c > d && (c - d) * 8u > (uint32_t)kCtPairs;
        const bool shared = e >= 3u * (uint32_t)kCtPairs / 4u && f < (uint32_t)kCtPairs / 4u;
        verdict[0] = fine_ ? 1ull : shared ? 2ull : strong ? 1ull : 0ull;
        __threadfence_system();                                           // (`verdict` may be host memory)
    }
}

// keys: the probe column (raw), x = key ^ bias is compared with the sorted build keys (64-bit keys: biased by 2^63).
// Every WAVE works on its own: 256 consecutive rows (a "stretch"; a workgroup's waves take the sixteen stretches of a batch of 4096),
// no workgroup barrier after the index is staged -- with barriers every batch waited three times for its slowest wave's memory
// round trips (0.63 ms per 1e8 rows; this: see profiles/r06_notes.md).
template <typename K>
__global__ __launch_bounds__(kCThreads) void cj_search_kernel(const K *__restrict__ keys, int64_t n, K bias, const K *__restrict__ rkeys, int64_t s,
                                                              uint2 *__restrict__ tmp, uint32_t *__restrict__ bcount, uint32_t *__restrict__ bfirst,
                                                              uint32_t *__restrict__ blast, int32_t *__restrict__ unsorted)
{
    constexpr int IDX = CIdx<K>::N, WS = 2048 / (int)sizeof(K);           // WS: build keys of a wave's slice (2 KiB: the same room as its 256 staged rows)
    __shared__ K s_idx[IDX];
    __shared__ K s_top[64];                                               // every (IDX / 64)-th index entry (read by lane: no bank conflicts)
    __shared__ uint2 s_stage[kCBatch];                                    // sixteen slices of build keys, 2 KiB per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto idx_pos = [&](uint32_t i) -> uint32_t { return (uint32_t)(((uint64_t)i * (uint64_t)s) / (uint64_t)IDX); };
    for (int i = tid; i < IDX; i += kCThreads) s_idx[i] = rkeys[idx_pos((uint32_t)i)];
    if (tid < 64) s_top[tid] = rkeys[idx_pos((uint32_t)tid * (uint32_t)(IDX / 64))];
    const K kmin = rkeys[0], kmax = rkeys[s - 1];
    // the first build entry >= key (a key inside [kmin, kmax]) lies in [lo, hi]: rkeys[lo - 1] < key (or lo = 0), rkeys[hi] >= key
    auto segment = [&](K key, uint32_t &lo, uint32_t &hi) {
        uint32_t pos = 0;                                                 // index entries below the key
#pragma unroll
        for (int step = IDX / 2; step > 0; step >>= 1) if (s_idx[pos + step - 1] < key) pos += step;
        if (s_idx[pos] < key) pos++;                                      // (the steps count among the first IDX - 1 entries)
        lo = pos == 0u ? 0u : idx_pos(pos - 1u) + 1u;
        hi = pos == (uint32_t)IDX ? (uint32_t)(s - 1) : idx_pos(pos);
        if (lo > hi) lo = hi;                                             // (pos = 0: the key IS the smallest build key)
    };
    auto wave_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };   // a wave's own LDS traffic, in program order
    // the same for a key the whole wave shares: 64 index entries compared at once, twice (13 dependent LDS reads otherwise -- the search
    // kernel is bound by instruction issue, profiles/r06_notes.md 8)
    auto segment_wave = [&](K key, uint32_t &lo, uint32_t &hi) {
        constexpr int B1 = IDX / 64;                                      // entries per lane-block of the first round (128 / 64)
        uint32_t pos = (uint32_t)__popcll(__ballot(s_top[lane] < key));               // blocks whose first entry is below the key: the key's block is pos - 1
        if (pos > 0u) {
            const uint32_t b0 = (pos - 1u) * (uint32_t)B1;                // entries b0 .. b0 + B1 - 1; entry b0 is below the key
            uint32_t below = 0;
#pragma unroll
            for (int t = 0; t < B1 / 64; t++) below += (uint32_t)__popcll(__ballot(s_idx[b0 + (uint32_t)(t * 64 + lane)] < key));
            pos = b0 + below;                                             // index entries below the key
        }
        lo = pos == 0u ? 0u : idx_pos(pos - 1u) + 1u;
        hi = pos == (uint32_t)IDX ? (uint32_t)(s - 1) : idx_pos(pos);
        if (lo > hi) lo = hi;
    };
    __syncthreads();
    K *sl = reinterpret_cast<K *>(s_stage + wave * (kCBatch / 16));     // this wave's slice of build keys
    const int64_t nbatch = (n + kCBatch - 1) / kCBatch;
    auto load = [&](int64_t batch, K (&x)[kCVec]) {                       // the lane's four rows of a batch (raw keys; rows past the end: 0, never used)
        const int64_t r0 = batch * kCBatch + (int64_t)tid * kCVec;
        if (r0 + kCVec <= n) {
            if (sizeof(K) == 4) {
                const hark_u4v q = __builtin_nontemporal_load(reinterpret_cast<const hark_u4v *>(keys + r0));
                x[0] = (K)q.x; x[1] = (K)q.y; x[2] = (K)q.z; x[3] = (K)q.w;
            } else {
                const hark_u4v q0 = __builtin_nontemporal_load(reinterpret_cast<const hark_u4v *>(keys + r0));
                const hark_u4v q1 = __builtin_nontemporal_load(reinterpret_cast<const hark_u4v *>(keys + r0 + 2));
                x[0] = (K)(((uint64_t)q0.y << 32) | q0.x); x[1] = (K)(((uint64_t)q0.w << 32) | q0.z);
                x[2] = (K)(((uint64_t)q1.y << 32) | q1.x); x[3] = (K)(((uint64_t)q1.w << 32) | q1.z);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kCVec; j++) x[j] = r0 + j < n ? keys[r0 + j] : (K)0;
        }
    };
    bool bad = false;
    K xn[kCVec];
    if ((int64_t)blockIdx.x < nbatch) load(blockIdx.x, xn);
    for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const int64_t r0 = batch * kCBatch + (int64_t)tid * kCVec;
        K x[kCVec];
#pragma unroll
        for (int j = 0; j < kCVec; j++) x[j] = xn[j];
        if (batch + gridDim.x < nbatch) load(batch + gridDim.x, xn);      // the next batch's keys travel while this one is searched
        // ---- the wave's 256 consecutive rows together: the build entries between their smallest and largest key.  Few of them
        // (a sorted column: 256 rows of 1e8 over 1e7 build keys span ~26) -> that slice into LDS, every row searched there;
        // many -> row by row through the index and the build keys in memory.
        uint32_t valid = 0;
        K mn = (K)~(K)0, mx = (K)0;
#pragma unroll
        for (int j = 0; j < kCVec; j++) {
            x[j] ^= bias;
            if (r0 + j < n && x[j] >= kmin && x[j] <= kmax) { valid |= 1u << j; mn = x[j] < mn ? x[j] : mn; mx = x[j] > mx ? x[j] : mx; }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const K a = cj_shfl_xor<K>(mn, d), b = cj_shfl_xor<K>(mx, d);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        uint32_t lo[kCVec], hi[kCVec], found = 0, c = 0;
#pragma unroll
        for (int j = 0; j < kCVec; j++) { lo[j] = 1u; hi[j] = 0u; }       // (lo > hi: no such row, or its key lies outside the build side)
        const size_t sub = (size_t)batch * 16 + (size_t)wave;             // the stretch: rows [256 sub, 256 sub + 256)
        if (mn <= mx) {                                                   // (wave-uniform: some row of the wave is inside the build side's range)
            bool sliced = false;
            uint32_t l1, h1, l2, h2;
            segment_wave(mn, l1, h1); segment_wave(mx, l2, h2);
            // 64 probes per step and search (a segment of ~1200 keys: the first step leaves ~19), until the entries between the two fit the slice
            while ((l1 < h1 || l2 < h2) && h2 - l1 >= (uint32_t)WS) {
                const uint32_t n1 = h1 - l1, n2 = h2 - l2;
                const uint32_t p1 = l1 + (uint32_t)(((uint64_t)lane * n1) >> 6), p2 = l2 + (uint32_t)(((uint64_t)lane * n2) >> 6);
                const K v1 = rkeys[p1], v2 = rkeys[p2];
                const int c1 = __popcll(__ballot(v1 < mn)), c2 = __popcll(__ballot(v2 < mx));   // the probes ascend: those below the key come first
                if (l1 < h1) {
                    if (c1 == 0) h1 = l1;
                    else { const uint32_t nh = c1 < 64 ? l1 + (uint32_t)(((uint64_t)c1 * n1) >> 6) : h1; l1 = l1 + (uint32_t)(((uint64_t)(c1 - 1) * n1) >> 6) + 1u; h1 = nh; }
                }
                if (l2 < h2) {
                    if (c2 == 0) h2 = l2;
                    else { const uint32_t nh = c2 < 64 ? l2 + (uint32_t)(((uint64_t)c2 * n2) >> 6) : h2; l2 = l2 + (uint32_t)(((uint64_t)(c2 - 1) * n2) >> 6) + 1u; h2 = nh; }
                }
            }
            const uint32_t first = l1, len = h2 - l1 + 1u;                // every valid row's first build entry >= its key lies in [first, first + len)
            if (len <= (uint32_t)WS) {
                sliced = true;
#pragma unroll
                for (int t = 0; t < WS / 64; t++) { const uint32_t i = (uint32_t)(lane + 64 * t); if (i < len) sl[i] = rkeys[first + i]; }
                wave_sync();
                const uint32_t top = 1u << (31 - __clz((int)len));       // the largest power of two <= len (a sorted column: 16 or 32, not WS / 2)
#pragma unroll
                for (int j = 0; j < kCVec; j++) {
                    if (!(valid & (1u << j))) continue;
                    uint32_t pos = 0;                                     // slice entries below the key
                    for (uint32_t step = top; step > 0u; step >>= 1) if (pos + step - 1u < len && sl[pos + step - 1u] < x[j]) pos += step;   // (top: wave-uniform)
                    if (pos < len && sl[pos] < x[j]) pos++;
                    lo[j] = hi[j] = first + pos;
                    if (pos < len && sl[pos] == x[j]) { found |= 1u << j; c++; }
                }
                wave_sync();                                              // (the slice is read before the next stretch's overwrites it)
            }
            if (!sliced) {
#pragma unroll
                for (int j = 0; j < kCVec; j++)
                    if (valid & (1u << j)) segment(x[j], lo[j], hi[j]);
                for (;;) {
                    K v[kCVec];
                    uint32_t mid[kCVec];
                    bool go = false;
#pragma unroll
                    for (int j = 0; j < kCVec; j++) {
                        mid[j] = lo[j] + ((hi[j] - lo[j]) >> 1);
                        const bool act = lo[j] < hi[j];
                        go = go || act;
                        v[j] = act ? rkeys[mid[j]] : (K)0;
                    }
                    if (!__any(go)) break;
#pragma unroll
                    for (int j = 0; j < kCVec; j++)
                        if (lo[j] < hi[j]) { if (v[j] < x[j]) lo[j] = mid[j] + 1u; else hi[j] = mid[j]; }
                }
#pragma unroll
                for (int j = 0; j < kCVec; j++)
                    if (lo[j] == hi[j] && rkeys[lo[j]] == x[j]) { found |= 1u << j; c++; }
            }
        }
        // ---- the matching rows of the stretch, in row order, straight from the registers (a lane's rows lie side by side, the lanes'
        // one behind the other).  Do the ranks ascend?  They do when the KEYS of the wave's rows inside the build side's range do -- a
        // descent among them sends the join to the general ordering even when no matching row is involved: a column in order has none
        // -- and across stretches when no stretch starts below an earlier one's largest rank (cj_scan1/2_kernel: first / last below).
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const uint32_t cnt = __shfl(incl, 63, 64);
        if (cnt) {
            // order of the valid keys: inside the lane, and against the nearest valid key of the lanes before it
            K lastv = (K)0, firstv = (K)0;
            bool anyv = false, down = false;
#pragma unroll
            for (int j = 0; j < kCVec; j++)
                if (valid & (1u << j)) { if (anyv && x[j] < lastv) down = true; if (!anyv) firstv = x[j]; lastv = x[j]; anyv = true; }
            K run = anyv ? lastv : (K)0;                                  // the largest "last valid key" of this lane and the lanes before it (keys that ascend: the nearest one)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const K t = cj_shfl_up<K>(run, d); if (lane >= d && t > run) run = t; }
            const K before = cj_shfl_up<K>(run, 1);
            if (anyv && lane > 0 && before > firstv) down = true;
            if (down) bad = true;
            uint2 *dst = tmp + sub * 256 + (incl - c);
            uint32_t at = 0, fr = 0xFFFFFFFFu, lr = 0u;
#pragma unroll
            for (int j = 0; j < kCVec; j++)
                if (found & (1u << j)) { dst[at++] = make_uint2(lo[j], (uint32_t)(r0 + j)); fr = min(fr, lo[j]); lr = max(lr, lo[j]); }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { fr = min(fr, (uint32_t)__shfl_xor((int)fr, d, 64)); lr = max(lr, (uint32_t)__shfl_xor((int)lr, d, 64)); }
            if (lane == 0) { bfirst[sub] = fr; blast[sub] = lr; }
        }
        if (lane == 0) bcount[sub] = cnt;
    }
    if (bad) *unsorted = 1;
}

// The stretches' offsets: cj_scan1_kernel scans 4096 stretches per workgroup (a wave 256 of them, 64 at a time: coalesced reads,
// scans by shuffles) -- boff[sub] = matching rows before the stretch within its workgroup's 4096, gsum / gmax / gminf the workgroup's
// total, largest last rank and smallest first rank; cj_scan2_kernel scans those (one workgroup) into goff and the grand total.
// *unsorted when a stretch starts below a rank seen before it (the search kernel has checked the inside of every stretch).
__global__ __launch_bounds__(1024) void cj_scan1_kernel(const uint32_t *__restrict__ bcount, const uint32_t *__restrict__ bfirst, const uint32_t *__restrict__ blast,
                                                        uint32_t nsub, uint32_t *__restrict__ boff, uint32_t *__restrict__ gsum, uint32_t *__restrict__ gmax,
                                                        uint32_t *__restrict__ gminf, int32_t *__restrict__ unsorted)
{
    __shared__ uint32_t s_sum[16], s_max[16], s_minf[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b0 = min(nsub, blockIdx.x * 4096u + wave * 256u), b1 = min(nsub, b0 + 256u);
    uint32_t run = 0, seen = 0, minf = 0xFFFFFFFFu;
    bool bad = false;
    for (uint32_t base = b0; base < b1; base += 64u) {
        const uint32_t b = base + lane;
        const uint32_t c = b < b1 ? bcount[b] : 0u;
        const uint32_t f = c ? bfirst[b] : 0xFFFFFFFFu, l = c ? blast[b] : 0u;
        uint32_t incl = c, pm = l;                                        // inclusive sum / inclusive max over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64), m = __shfl_up(pm, d, 64);
            if (lane >= (uint32_t)d) { incl += t; pm = max(pm, m); }
        }
        uint32_t before = __shfl_up(pm, 1, 64);                           // the largest rank of the lanes before this one ...
        before = max(lane ? before : 0u, seen);                           // ... and of the wave's earlier rounds
        if (c && before > f) bad = true;
        if (b < b1) boff[b] = run + incl - c;
        run += __shfl(incl, 63, 64);
        seen = max(seen, __shfl(pm, 63, 64));
        minf = min(minf, f);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) minf = min(minf, __shfl_down(minf, d, 64));
    if (lane == 0) { s_sum[wave] = run; s_max[wave] = seen; s_minf[wave] = minf; }
    __syncthreads();
    uint32_t woff = 0, wseen = 0, all = 0, allmax = 0, allmin = 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t w = 0; w < 16u; w++) {
        if (w < wave) { woff += s_sum[w]; wseen = max(wseen, s_max[w]); }
        all += s_sum[w]; allmax = max(allmax, s_max[w]); allmin = min(allmin, s_minf[w]);
    }
    if (wseen > s_minf[wave]) bad = true;                                 // (no rows in this wave: the minimum is all ones)
    if (woff) for (uint32_t b = b0 + lane; b < b1; b += 64u) boff[b] += woff;
    if (bad) *unsorted = 1;
    if (tid == 0) { gsum[blockIdx.x] = all; gmax[blockIdx.x] = allmax; gminf[blockIdx.x] = allmin; }
}

__global__ __launch_bounds__(1024) void cj_scan2_kernel(const uint32_t *__restrict__ gsum, const uint32_t *__restrict__ gmax, const uint32_t *__restrict__ gminf,
                                                        uint32_t ngroup, uint32_t *__restrict__ goff, unsigned long long *__restrict__ total, int32_t *__restrict__ unsorted)
{
    __shared__ uint32_t s_sum[1024], s_max[1024];
    const uint32_t tid = threadIdx.x, chunk = (ngroup + 1023u) / 1024u;   // (2^32 rows: 4096 groups, four per thread)
    const uint32_t g0 = min(ngroup, tid * chunk), g1 = min(ngroup, g0 + chunk);
    uint32_t sum = 0, seen = 0, minf = 0xFFFFFFFFu;
    bool bad = false;
    for (uint32_t g = g0; g < g1; g++) {
        if (gsum[g] && seen > gminf[g]) bad = true;
        if (gsum[g]) { seen = max(seen, gmax[g]); minf = min(minf, gminf[g]); }
        sum += gsum[g];
    }
    s_sum[tid] = sum; s_max[tid] = seen;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t a = tid >= d ? s_sum[tid - d] : 0u, m = tid >= d ? s_max[tid - d] : 0u;
        __syncthreads();
        s_sum[tid] += a; s_max[tid] = max(s_max[tid], m);
        __syncthreads();
    }
    const uint32_t before = tid ? s_sum[tid - 1u] : 0u, prev = tid ? s_max[tid - 1u] : 0u;
    if (prev > minf) bad = true;
    uint32_t run = before;
    for (uint32_t g = g0; g < g1; g++) { goff[g] = run; run += gsum[g]; }
    if (bad) *unsorted = 1;
    if (tid == 1023u) *total = (unsigned long long)s_sum[1023];
}

// The stretches' rows to their places.  A wave takes 64 stretches at a time: when none of them holds more than eight rows (one row in a
// hundred matching: 2-3 per stretch) every LANE copies its own stretch -- a wave per stretch spent 0.13 ms per 1e8 probe rows on
// 390 K stretches of 2-3 rows --, otherwise the wave copies them one after the other, 64 rows at a time.
__global__ __launch_bounds__(1024) void cj_emit_kernel(const uint2 *__restrict__ tmp, const uint32_t *__restrict__ bcount, const uint32_t *__restrict__ boff,
                                                       const uint32_t *__restrict__ goff, uint32_t nsub,
                                                       const uint32_t *__restrict__ runlen, const uint32_t *__restrict__ lval, const uint32_t *__restrict__ rranked,
                                                       uint32_t *__restrict__ rank, uint32_t *__restrict__ lrow, uint32_t *__restrict__ cnt,
                                                       uint32_t *__restrict__ lv, uint32_t *__restrict__ rv)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    auto put = [&](size_t at, const uint2 e) {
        if (rank) rank[at] = e.x;
        if (lrow) lrow[at] = e.y;
        if (cnt) cnt[at] = runlen[e.x];
        if (lv) lv[at] = lval[e.y];
        if (rv) rv[at] = rranked[e.x];
    };
    for (uint32_t s0 = (blockIdx.x * 16u + wave) * 64u; s0 < nsub; s0 += gridDim.x * 16u * 64u) {
        const uint32_t sub = s0 + lane;
        const uint32_t c = sub < nsub ? bcount[sub] : 0u;
        const size_t off = c ? (size_t)boff[sub] + goff[sub >> 12] : 0;
        uint32_t most = c;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, d, 64));
        if (most == 0u) continue;
        if (most <= 8u) {
            const uint2 *src = tmp + (size_t)sub * 256;
            for (uint32_t i = 0; i < c; i++) put(off + i, src[i]);
        } else {
            for (unsigned long long todo = __ballot(c != 0u); todo; todo &= todo - 1ull) {
                const int l = __ffsll((long long)todo) - 1;
                const uint32_t cl = (uint32_t)__shfl((int)c, l, 64);
                const size_t ol = (size_t)(((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(off >> 32), l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)off, l, 64));
                const uint2 *src = tmp + (size_t)(s0 + (uint32_t)l) * 256;
                for (uint32_t i = lane; i < cl; i += 64u) put(ol + i, src[i]);
            }
        }
    }
}

template <typename K>
int run_clustered(hark_context *ctx, const K *lcol, K bias, int64_t n, const K *rkeys, int64_t s, const uint32_t *runlen, const int32_t *flags,
                  const uint32_t *lval, const uint32_t *rranked, uint32_t **rank_out, uint32_t **lrow_out, uint32_t **cnt_out, uint32_t **lval_out,
                  uint32_t **rval_out, int64_t *m_out, bool *used, bool *dup_out, bool rows_needed, int64_t *general_out)
{
    *used = false;
    hipStream_t st = ctx->stream;
    const int64_t nbatch = (n + kCBatch - 1) / kCBatch, nsub = nbatch * 16, ngroup = (nsub + 4095) / 4096;   // batches of 4096 rows, stretches of 256, 4096 stretches
    uint2 *tmp = nullptr;
    uint32_t *bcount = nullptr, *bfirst = nullptr, *blast = nullptr, *boff = nullptr, *gsum = nullptr, *gmax = nullptr, *gminf = nullptr, *goff = nullptr;
    int64_t *info = nullptr;                                              // [0] matching rows, [1] low: not ascending, [2] the caller's flags
    uint32_t *rank = nullptr, *lrow = nullptr, *cnt = nullptr, *lv = nullptr, *rv = nullptr;
    int rc = hark_alloc(ctx, (void **)&tmp, 8 * (size_t)nsub * 256);
    if (!rc) rc = hark_alloc(ctx, (void **)&bcount, 4 * ((size_t)nsub * 4 + (size_t)ngroup * 4));
    if (!rc) rc = hark_alloc(ctx, (void **)&info, 64);
    auto cleanup = [&]() { hark_free(ctx, tmp); hark_free(ctx, bcount); hark_free(ctx, info); };
    if (rc == HARK_ENOMEM) { cleanup(); ctx->err.clear(); return HARK_OK; }   // no room: the sort-merge path needs less
    if (rc) { cleanup(); return rc; }
    bfirst = bcount + nsub; blast = bfirst + nsub; boff = blast + nsub; gsum = boff + nsub; gmax = gsum + ngroup; gminf = gmax + ngroup; goff = gminf + ngroup;
    HIP_TRY_RC(ctx, rc, hipMemsetAsync(info, 0, 64, st));
    int grid = (int)std::min<int64_t>(nbatch, (int64_t)ctx->num_cu * 2);
    if (const char *e = getenv("HARK_CJOIN_GRID")) { const int g = atoi(e); if (g >= 1 && g <= 65536) grid = (int)std::min<int64_t>(g, nbatch); }   // tests: several batches per workgroup at any size
    HARK_LAUNCH_RC(ctx, rc, cj_search_kernel<K><<<dim3((unsigned)grid), dim3(kCThreads), 0, st>>>(lcol, n, bias, rkeys, s, tmp, bcount, bfirst, blast, reinterpret_cast<int32_t *>(info + 1)));
    HARK_LAUNCH_RC(ctx, rc, cj_scan1_kernel<<<dim3((unsigned)ngroup), 1024, 0, st>>>(bcount, bfirst, blast, (uint32_t)nsub, boff, gsum, gmax, gminf, reinterpret_cast<int32_t *>(info + 1)));
    HARK_LAUNCH_RC(ctx, rc, cj_scan2_kernel<<<1, 1024, 0, st>>>(gsum, gmax, gminf, (uint32_t)ngroup, goff, reinterpret_cast<unsigned long long *>(info), reinterpret_cast<int32_t *>(info + 1)));
    HIP_TRY_RC(ctx, rc, hipMemcpyAsync(info + 2, flags, 8, hipMemcpyDeviceToDevice, st));
    int64_t words[3] = {0, 0, 0};
    if (!rc) rc = hark_read_words(ctx, info, words, 3);
    const int64_t M = words[0];
    const bool general = (words[1] & 0xFFFFFFFFll) != 0 || getenv("HARK_JOIN_FULLSORT") != nullptr;
    const bool dup = ((words[2] >> 32) & 0xFFFFFFFFll) != 0;
    *dup_out = dup;
    *general_out = general ? 1 : 0;
    if (!rc && M > 0) {
        const bool skip_rows = !rows_needed && !dup && !general && !getenv("HARK_JOIN_ROWS");
        if (!skip_rows) rc = hark_alloc(ctx, (void **)&rank, 4 * (size_t)M);
        if (!rc && !skip_rows) rc = hark_alloc(ctx, (void **)&lrow, 4 * (size_t)M);
        if (!rc && dup) rc = hark_alloc(ctx, (void **)&cnt, 4 * (size_t)M);
        if (!rc && lval && !general) rc = hark_alloc(ctx, (void **)&lv, 4 * (size_t)M);
        if (!rc && rranked && !dup && !general) rc = hark_alloc(ctx, (void **)&rv, 4 * (size_t)M);
        const int g = (int)std::min<int64_t>((nsub + 1023) / 1024, (int64_t)ctx->num_cu * 8);
        HARK_LAUNCH_RC(ctx, rc, cj_emit_kernel<<<dim3((unsigned)g), 1024, 0, st>>>(tmp, bcount, boff, goff, (uint32_t)nsub, runlen, lval, rranked, rank, lrow, cnt, lv, rv));
    }
    cleanup();                                                            // stream-ordered reuse
    if (rc) { hark_free(ctx, rank); hark_free(ctx, lrow); hark_free(ctx, cnt); hark_free(ctx, lv); hark_free(ctx, rv); return rc; }
    *rank_out = rank; *lrow_out = lrow; *cnt_out = cnt; *lval_out = lv; *rval_out = rv; *m_out = M; *used = true;
    return HARK_OK;
}

} // namespace

// the test's three words at the end of the context's pinned scratch (the kernel writes host memory; read after the stream's event)
static unsigned long long *cj_verdict_words(hark_context *ctx) { return reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctx->h_pin) + 65536 - 64); }

// Enqueues the test on `st`; k_cjoin_verdict reads it once the caller has waited for that stream's work.  build: the build side's
// key column (sorted_build = false: before it is sorted, as the table holds it) or its sorted (64-bit: biased) keys.
int k_cjoin_test(hark_context *ctx, hipStream_t st, const void *lcol, bool k64, int64_t n, const void *build, int64_t s, bool sorted_build)
{
    int rc = HARK_OK;
    unsigned long long *v = cj_verdict_words(ctx);
    if (k64) HARK_LAUNCH_RC(ctx, rc, cj_test_kernel<uint64_t><<<1, 1024, 0, st>>>(static_cast<const uint64_t *>(lcol), n, sorted_build ? 0x8000000000000000ull : 0ull,
                                                                                  static_cast<const uint64_t *>(build), s, sorted_build ? 1 : 0, v));
    else HARK_LAUNCH_RC(ctx, rc, cj_test_kernel<uint32_t><<<1, 1024, 0, st>>>(static_cast<const uint32_t *>(lcol), n, 0u, static_cast<const uint32_t *>(build), s, sorted_build ? 1 : 0, v));
    return rc;
}

int k_cjoin_verdict(hark_context *ctx)                                    // 0: rows scatter, 1: clustered (the search path), 2: neighbouring rows share a bucket (rotated loads)
{
    const volatile unsigned long long *v = cj_verdict_words(ctx);
    return (int)v[0];
}

int k_cjoin_run(hark_context *ctx, const void *lcol, bool k64, int64_t n, const void *rkeys, int64_t s, const uint32_t *runlen, const int32_t *flags,
                const uint32_t *lval, const uint32_t *rranked, uint32_t **rank_out, uint32_t **lrow_out, uint32_t **cnt_out, uint32_t **lval_out,
                uint32_t **rval_out, int64_t *m_out, bool *used, bool *dup_out, bool rows_needed, int64_t *general_out)
{
    return k64 ? run_clustered<uint64_t>(ctx, static_cast<const uint64_t *>(lcol), 0x8000000000000000ull, n, static_cast<const uint64_t *>(rkeys), s, runlen, flags, lval, rranked,
                                         rank_out, lrow_out, cnt_out, lval_out, rval_out, m_out, used, dup_out, rows_needed, general_out)
               : run_clustered<uint32_t>(ctx, static_cast<const uint32_t *>(lcol), 0u, n, static_cast<const uint32_t *>(rkeys), s, runlen, flags, lval, rranked,
                                         rank_out, lrow_out, cnt_out, lval_out, rval_out, m_out, used, dup_out, rows_needed, general_out);
}
