// k_msort.hip -- ORDER BY / argsort on 64-bit keys: a most-significant-digit-first bucket sort of 16-byte tuples.
//
// Replaces, for large tables of well-spread keys, the least-significant-digit-first tuple passes of k_sort.hip
// (sort_i64_tuples: four radix passes over 16-byte tuples + a run fix-up = five sweeps that read and write 16 B per row, and
// a histogram pass each: 19.6 GB moved per 1e8 rows, profiles/r04_op_traffic.txt).  The tuples carry the row id, so nothing
// has to be stable on the way -- (key, row id) is a total order -- and three sweeps do:
//
//   msd_sample / msd_setup   bounds of the keys from a sample -> a monotone map key -> position a in [0, 2^24) -> bucket d in [0, D),
//                            D = 256 * nb2 (a = ((key - kmin) >> sh) * mulA >> 32, clamped: whatever the sample missed lands in the
//                            first or last bucket, the order of the buckets still is the order of the keys; d = a >> s)
//   msd_hist                 workgroup w counts the level-1 digits (d / nb2) of ITS rows [w S, (w + 1) S): exact sizes of
//                            the 256 x 256 slabs, so no slab can overflow whatever the order of the input (a sorted column
//                            sends all rows of a workgroup to one bucket) -- and, of one key in eight, the keys per cell of 4096 positions
//   msd_scan1 / msd_eq       a level-1 bucket far above the average = a lumpy distribution (normal, exponential ...): the cells'
//                            counts then EQUALISE the map (d = the cell's share of the buckets, interpolated inside the cell),
//                            msd_hist counts again under the new map, and only a second failure gives up
//   msd_part<true>   sweep 1 the same workgroup forms tuples (key ^ xorm, row id, value) from its rows and routes them to its
//                            slab of their bucket: a tile of 4096 tuples is counting-sorted by digit in LDS, every bucket's
//                            run leaves in whole 128-byte lines (up to seven tuples per bucket wait in LDS for the next tile)
//   msd_part<false>  sweep 2 one workgroup per level-1 bucket reads the bucket (its slabs lie one behind the other, padded
//                            with dead tuples to whole lines) and routes by d % nb2 into regions of fixed capacity, the same way
//   msd_bounds               where every final bucket's keys start and how wide they are: the final digit's constants
//   msd_final        sweep 3 persistent workgroups sort the final buckets (<= 2560 tuples each) in LDS: counting sort by the next
//                            11 bits, ranks inside the runs of equal digits by comparing (key, row id), keys / row ids / values
//                            stored at their ranks; the next bucket's tuples are loaded while a bucket is sorted
//
// Whatever does not fit -- a final bucket over its capacity (many copies of a key, tight clusters), too much ranking work in a
// bucket -- raises a word and the caller takes the tuple passes as before (hark's result does not depend on the path).
#include "hark_internal.h"
#include <cstdio>
#include <vector>

typedef unsigned long long u64;

namespace {

constexpr int kT = 1024, kR = 4, kTile = kT * kR;            // partition workgroup: threads, tuples per thread and tile
constexpr int kB = 256;                                      // buckets per partition level
constexpr uint32_t kDead = 0xFFFFFFFFu;                      // row id of a padding tuple
// The final workgroup: FT threads sort buckets of up to 5 FT tuples with 4 FT final digits -- FT = 512 (2560 tuples, three workgroups
// share a CU's LDS) up to 1.05e8 rows, FT = 1024 (5120 tuples, one workgroup per CU) for tables of up to 2.1e8 rows, whose 65536
// final buckets hold up to 3200 tuples on average.
constexpr int kFR = 5;
constexpr int kFBinsOf(int ft) { return 4 * ft; }
constexpr int kFCapOf(int ft) { return kFR * ft; }
constexpr uint32_t kFWorkMax = 1u << 18;                     // ... and the sum of the squared run lengths of a bucket that is ranked at all (a run of
                                                             // 500 alone; the tail buckets of a normal distribution come to 1e5): buckets of a
                                                             // column with hundreds of copies of every key are over it, and the first one says so
constexpr int kFRunMax = 512;                                // longest run of one final digit that is ranked by comparing (quadratic).  The work bound above is the
                                                             // one that bites: a run of more than 512 is over it by itself (512^2 = kFWorkMax), so this test only
                                                             // backs it up -- columns with such runs always take the caller's tuple passes and are remembered
                                                             // (hark_column::msd_unfit)
constexpr int kSampleWg = 256, kSampleStride = 61;          // one line of sixteen keys in 61 is looked at for the bounds: ~61 keys of ANY distribution lie below the
                                                             // sampled minimum (and above the maximum); with one in 509 it was ~500, all clamped
                                                             // into one final digit of the first bucket -- over the ranking bound for a normal
                                                             // distribution's tails, which the margin of the map does not cover

#ifdef HARK_MSD_CHECK
// bounds checks for experiments (tools/ab_build.sh chk "-DHARK_MSD_CHECK"): a violation is recorded in flag[4..7] and the access skipped
#define MSD_CHK(ok, code, val) ((ok) ? true : (atomicCAS(&flag[4], 0, (int32_t)(code)) == 0 ? (flag[5] = (int32_t)(val), flag[6] = (int32_t)blockIdx.x, flag[7] = (int32_t)threadIdx.x, atomicOr(&flag[0], 64), false) : false))
#else
#define MSD_CHK(ok, code, val) true
#endif
constexpr int kCells = 4096;                                 // cells of the equalisation table: position >> 12
struct MsdMap { u64 kmin; uint32_t sh, mulA, dmax, s24; };   // position a = reduced key * mulA >> 32 < 2^24; bucket = a >> s24 unless equalised

__device__ __forceinline__ uint32_t reduced_key(u64 key, u64 kmin, uint32_t sh)
{
    const u64 rel = key > kmin ? key - kmin : 0ull;          // (below the sampled minimum: the first bucket)
    const u64 h = rel >> sh;
    return h > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)h;    // (above the sampled maximum: the last bucket)
}
// the position of a key: < 2^24 for reduced keys inside the sampled range by the choice of mulA, clamped for those above it
__device__ __forceinline__ uint32_t position_of(u64 key, const MsdMap &m)
{
    const uint32_t a = __umulhi(reduced_key(key, m.kmin, m.sh), m.mulA);
    return a < 0xFFFFFFu ? a : 0xFFFFFFu;
}
// The equalised coordinate of a position, in 1/32768 buckets: the table holds per cell {the coordinate of its first position,
// the coordinate's growth per position x 256} (the factor 256: a cell with a handful of keys still spreads them over its
// positions -- with whole units per position a tail cell's keys all shared one coordinate, and one final digit)
__device__ __forceinline__ uint32_t equalised(uint2 t, uint32_t a) { return t.x + (uint32_t)(((u64)t.y * (a & 4095u)) >> 8); }
// tab (LDS, or null): the equalisation table
__device__ __forceinline__ uint32_t bucket_of(u64 key, const MsdMap &m, const uint2 *tab)
{
    const uint32_t a = position_of(key, m);
    if (!tab) return a >> m.s24;
    const uint32_t d = equalised(tab[a >> 12], a) >> 15;
    return d < m.dmax ? d : m.dmax;
}

// Has an earlier step given up?  ONE lane asks and the workgroup shares the answer: the word may be raised by another workgroup of
// the SAME kernel while this one's waves are still starting, and a workgroup whose waves disagree loses the work of those that
// left (round 5: a final bucket's scan total was never written, the scatter indices behind it went anywhere in LDS and beyond).
__device__ __forceinline__ bool msd_gave_up(const int32_t *flag, bool *equalised = nullptr)
{
    __shared__ int32_t s_gave_up, s_eq;
    if (threadIdx.x == 0) { s_gave_up = *reinterpret_cast<const volatile int32_t *>(flag); s_eq = reinterpret_cast<const volatile int32_t *>(flag)[2]; }
    __syncthreads();
    if (equalised) *equalised = s_eq != 0;
    return s_gave_up != 0;
}
// the equalisation table into LDS (when the map is equalised), else no table
__device__ __forceinline__ const uint2 *msd_table(bool equalised, const uint2 *__restrict__ tab_g, uint2 *tab_lds)
{
    if (!equalised) return nullptr;
    for (int i = threadIdx.x; i < kCells; i += blockDim.x) tab_lds[i] = tab_g[i];
    __syncthreads();
    return tab_lds;
}
// Tuples of a wave are counted / ranked per bucket with LDS atomics; when every live lane of the wave has the SAME bucket (a
// sorted or clustered column: all 64 atomics on one address, one after the other) one lane adds for all.
__device__ __forceinline__ uint32_t bucket_rank(uint32_t *cnt, uint32_t d, bool live)
{
    const unsigned long long lm = __ballot(live);
    if (lm == 0ull) return 0u;
    const int first = __ffsll((long long)lm) - 1;
    const uint32_t d0 = (uint32_t)__shfl((int)d, first, 64);
    if (__ballot(live && d != d0) == 0ull) {
        uint32_t base = 0u;
        if ((int)(threadIdx.x & 63) == first) base = atomicAdd(&cnt[d0], (uint32_t)__popcll(lm));
        base = (uint32_t)__shfl((int)base, first, 64);
        return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
    }
    return live ? atomicAdd(&cnt[d], 1u) : 0u;
}

// ---- bounds from a sample -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void msd_sample_kernel(const u64 *__restrict__ col, int64_t n, u64 xorm, int64_t stride, u64 *__restrict__ mm)
{
    __shared__ u64 s_min[16], s_max[16];
    u64 lo = ~0ull, hi = 0ull;
    // whole 128-byte lines (sixteen keys, eight lanes x two keys), one line in `stride`: single keys 488 bytes apart cost a memory
    // transaction each (1.6 M of them for 1e8 rows: 65 us); the same number of keys from one line in 61 costs a sixteenth
    const int64_t g = (int64_t)blockIdx.x * 1024 + threadIdx.x, lines = (n + 15) / 16, lstep = (int64_t)gridDim.x * 128 * stride;
    for (int64_t line = (g >> 3) * stride; line < lines; line += lstep) {
        const int64_t i = line * 16 + (g & 7) * 2;
        for (int q = 0; q < 2; q++) if (i + q < n) { const u64 k = col[i + q] ^ xorm; lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n > 0) { const u64 k = col[n - 1] ^ xorm; lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    for (int d = 32; d; d >>= 1) {
        const u64 l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = lo; s_max[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) { lo = s_min[w] < lo ? s_min[w] : lo; hi = s_max[w] > hi ? s_max[w] : hi; }
        if (lo <= hi) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }   // (one pair of atomics per workgroup: 1024 workgroups on two addresses took 60 us)
    }
}

__global__ void msd_setup_kernel(const u64 *__restrict__ mm, uint32_t D, MsdMap *__restrict__ map, int32_t *__restrict__ flag)
{
    u64 lo = mm[0], hi = mm[1] >= mm[0] ? mm[1] : mm[0];
    if (((hi - lo) >> 32) == 0ull) flag[0] = 2;              // keys within 2^32 of each other: the caller's 32-bit paths are the better ones
    // The sample's extremes are not the column's: one key in `stride` was looked at, so ~stride keys lie below the sampled minimum
    // (and above the maximum), and clamped to the first bucket's first digit they would be one long run for msd_final to rank.
    // The map covers 1/64 of the range more on either side: uniform keys then all fall inside it (the outermost buckets stay
    // emptier), and only true outliers are clamped.
    const u64 margin = ((hi - lo) >> 6) + 1ull;
    lo = lo > margin ? lo - margin : 0ull;
    hi = hi < ~0ull - margin ? hi + margin : ~0ull;
    const u64 range = hi - lo;                               // the largest relative key
    uint32_t sh = 0;
    while ((range >> sh) > 0xFFFFFFFEull) sh++;
    const u64 r32 = (range >> sh) + 1ull;                    // reduced keys lie in [0, r32), r32 <= 2^32 - 1
    u64 mulA = (1ull << 56) / r32;                           // a = h * mulA >> 32 < 2^24 for h < r32
    if (mulA > 0xFFFFFFFFull) mulA = 0xFFFFFFFFull;          // (fewer than 2^24 distinct reduced keys: a = h, about)
    uint32_t s24 = 24u; for (uint32_t x = D; x > 1u; x >>= 1) s24--;
    map->kmin = lo; map->sh = sh; map->mulA = (uint32_t)mulA; map->dmax = D - 1u; map->s24 = s24;
}

// ---- level-1 histogram: exact slab sizes ------------------------------------------------------------------------------------
// pass 0: under the affine map, and the keys per cell of 4096 positions beside it; pass 1 (only when msd_scan1 found the
// distribution lumpy and msd_eq equalised the map): the slab sizes again, under the new map.
__global__ __launch_bounds__(kT) void msd_hist_kernel(int pass, const u64 *__restrict__ col, int64_t n, int64_t slice, u64 xorm, const MsdMap *__restrict__ mapp,
                                                      const uint2 *__restrict__ tab_g, int nb2log, uint32_t *__restrict__ counts1 /* [nwg][256] */,
                                                      uint32_t *__restrict__ cells /* [4096], pass 0 */, const int32_t *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_raw);               // [kB]
    uint32_t *s_cell = s_cnt + kB;                                         // [kCells] (pass 0)
    uint2 *s_tab = reinterpret_cast<uint2 *>(s_cell);                      // [kCells] (pass 1: the same room)
    bool eq;
    if (msd_gave_up(flag, &eq)) return;                                    // an earlier step gave up: nothing to do
    if (pass == 1 && !eq) return;                                          // the affine map did: nothing to count again
    const MsdMap m = *mapp;
    const uint2 *tab = pass == 1 ? msd_table(true, tab_g, s_tab) : nullptr;
    if (threadIdx.x < kB) s_cnt[threadIdx.x] = 0u;
    if (pass == 0) for (int i = threadIdx.x; i < kCells; i += kT) s_cell[i] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    // eight keys per thread and round in four 16-byte loads, all issued before the first digit is counted (64 KB in flight per CU:
    // with one 8-byte load per lane the pass ran at the latency of a load, 3.4 TB/s)
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    auto count = [&](u64 key, bool live, bool cell_too) {
        const uint32_t a = position_of(key, m);
        uint32_t d = a >> m.s24;
        if (tab) { d = equalised(tab[a >> 12], a) >> 15; d = d < m.dmax ? d : m.dmax; }
        bucket_rank(s_cnt, d >> nb2log, live);
        if (cell_too) bucket_rank(s_cell, a >> 12, live);                  // (one key in eight: the cells' RELATIVE sizes are what the map is made of)
    };
    for (int64_t t0 = lo; t0 < hi; t0 += 8 * kT) {             // (lo and the slices are multiples of the tile: 16-byte aligned pairs)
        u64x2 k[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t i = t0 + 2 * ((int64_t)q * kT + threadIdx.x);
            if (i + 1 < hi) k[q] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(col + i));
            else { k[q].x = i < hi ? col[i] : 0ull; k[q].y = 0ull; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t i = t0 + 2 * ((int64_t)q * kT + threadIdx.x);
            count(k[q].x ^ xorm, i < hi, pass == 0 && q == 0);
            count(k[q].y ^ xorm, i + 1 < hi, false);
        }
    }
    __syncthreads();
    if (threadIdx.x < kB) counts1[(size_t)blockIdx.x * kB + threadIdx.x] = s_cnt[threadIdx.x];
    if (pass == 0) for (int i = threadIdx.x; i < kCells; i += kT) { const uint32_t c = s_cell[i]; if (c) atomicAdd(&cells[i], c); }
}
constexpr size_t msd_hist_lds() { return (size_t)kB * 4 + (size_t)kCells * 8; }

// exclusive scan over the 1024 threads of a workgroup (x = this thread's sum); *total = the sum of all
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t x, uint32_t *s_wave /* [16] */, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = x;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int q = 0; q < 16; q++) { const uint32_t v = s_wave[q]; if (q < wv) before += v; all += v; }
    *total = all;
    return before + incl - x;
}

// Slab offsets, bucket-major: bucket b's slabs (workgroup 0 .. nwg - 1) lie one behind the other, every slab padded to whole
// lines of eight tuples.  off1[w][b] = first tuple of slab (w, b); bstart[b] = first tuple of bucket b, bstart[256] = the end;
// bfirst[b] = the rows of the buckets before b (where bucket b's rows start in the sorted output).
// Also the first place where a lumpy distribution shows: a level-1 bucket with more than 1.5 x the average would overflow its
// final buckets (capacity 1.6-3.2 x their average).  Under the affine map (pass 0) that asks for the equalised map (flag[2]);
// under the equalised map (pass 1) the path gives up -- before the sweeps, not after two of them.
__global__ __launch_bounds__(1024) void msd_scan1_kernel(int pass, const uint32_t *__restrict__ counts1, int nwg, uint32_t *__restrict__ off1, uint32_t *__restrict__ bstart,
                                                         uint32_t *__restrict__ bfirst, int64_t n, int32_t *__restrict__ flag)
{
    __shared__ uint32_t s_wave[16], s_wave2[16], s_bucket[kB];
    bool eq;
    if (msd_gave_up(flag, &eq)) return;
    if (pass == 1 && !eq) return;
    if (threadIdx.x < kB) s_bucket[threadIdx.x] = 0u;
    __syncthreads();
    // thread t owns the entries [t * per, (t + 1) * per) of the bucket-major order (entry e = b * nwg + w)
    const int total = kB * nwg, per = (total + 1023) / 1024;
    uint32_t sum = 0, exact = 0;
    for (int e = threadIdx.x * per; e < (threadIdx.x + 1) * per && e < total; e++) {
        const int b = e / nwg, w = e - b * nwg; const uint32_t c = counts1[(size_t)w * kB + b];
        sum += (c + 7u) & ~7u; exact += c;
        if (c) atomicAdd(&s_bucket[b], c);
    }
    __syncthreads();
    if (threadIdx.x < kB && (int64_t)s_bucket[threadIdx.x] > n / kB + n / (2 * kB) + 4096) { if (pass == 0) flag[2] = 1; else atomicOr(&flag[0], 32); }
    uint32_t all, all2;
    uint32_t at = block_excl_scan_1024(sum, s_wave, &all);
    uint32_t ex = block_excl_scan_1024(exact, s_wave2, &all2);
    if (threadIdx.x == 0) { bstart[kB] = all; bfirst[kB] = all2; }
    for (int e = threadIdx.x * per; e < (threadIdx.x + 1) * per && e < total; e++) {
        const int b = e / nwg, w = e - b * nwg;
        const uint32_t c = counts1[(size_t)w * kB + b];
        off1[(size_t)w * kB + b] = at;
        if (w == 0) { bstart[b] = at; bfirst[b] = ex; }
        at += (c + 7u) & ~7u; ex += c;
    }
}

// The equalised map: the keys before a cell and the keys in it, as shares of the D buckets in 1/32768 buckets -- `equalised()`
// (monotone: a cell's last position stays below the next cell's first coordinate).
__global__ __launch_bounds__(1024) void msd_eq_kernel(const uint32_t *__restrict__ cells, int64_t n, uint32_t D, uint2 *__restrict__ tab, const int32_t *__restrict__ flag)
{
    __shared__ uint32_t s_wave[16];
    bool eq;
    if (msd_gave_up(flag, &eq) || !eq) return;
    const uint32_t c0 = cells[4 * threadIdx.x], c1 = cells[4 * threadIdx.x + 1], c2 = cells[4 * threadIdx.x + 2], c3 = cells[4 * threadIdx.x + 3];
    uint32_t all;
    const uint32_t before = block_excl_scan_1024(c0 + c1 + c2 + c3, s_wave, &all);
    const uint32_t cs[4] = {c0, c1, c2, c3};
    u64 run = before;
    const u64 tot = all ? all : 1u;                                        // (the cells hold one key in eight: shares of THEIR total)
    for (int j = 0; j < 4; j++) {
        const u64 first = (run * D << 15) / tot, per = (((u64)cs[j] * D << 15) / tot) >> 4;      // growth per position x 256 (a cell has 4096 positions)
        tab[4 * threadIdx.x + j] = uint2{(uint32_t)first, (uint32_t)per};
        run += cs[j];
    }
}

// Where a final bucket's reduced keys start (the smallest reduced key the map sends to bucket f or beyond) and the factor that
// spreads the bucket's reduced keys over the `fbins` final digits: msd_final's constants, one thread per bucket.
__global__ __launch_bounds__(256) void msd_bounds_kernel(const MsdMap *__restrict__ mapp, const uint2 *__restrict__ tab, uint32_t D, uint32_t fbins, uint32_t *__restrict__ lo_h /* [D + 1] */,
                                                         uint32_t *__restrict__ mul3 /* [D] */, const int32_t *__restrict__ flag)
{
    bool eq;
    if (msd_gave_up(flag, &eq)) return;
    const MsdMap m = *mapp;
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f > D) return;
    // the smallest position a with bucket(a) >= f
    auto first_position = [&](uint32_t b) -> u64 {
        if (b >= D) return 1ull << 24;
        if (!eq) return (u64)b << m.s24;
        const u64 want = (u64)b << 15;
        int lo = 0, hi = kCells - 1;                                       // the last cell whose first share is <= want
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((u64)tab[mid].x <= want) lo = mid; else hi = mid - 1; }
        const uint2 t = tab[lo];
        const u64 need = want - t.x;
        u64 pos = need == 0 ? 0 : (t.y ? ((need << 8) + t.y - 1) / t.y : 4096ull);
        if (pos > 4096ull) pos = 4096ull;
        return (u64)lo * 4096ull + pos;
    };
    auto first_reduced = [&](uint32_t b) -> u64 {                           // the smallest h with h * mulA >> 32 >= position
        const u64 a = first_position(b);
        const u64 h = ((a << 32) + m.mulA - 1ull) / m.mulA;
        return h > 0xFFFFFFFFull ? 0xFFFFFFFFull : h;
    };
    const u64 h0 = first_reduced(f);
    lo_h[f] = (uint32_t)h0;
    if (f < D) {
        const u64 h1 = first_reduced(f + 1u);
        const u64 w = h1 > h0 ? h1 - h0 : 1ull;
        u64 x = ((u64)fbins << 32) / w; if (x > 0xFFFFFFFFull) x = 0xFFFFFFFFull;
        // A bucket of the equalised map that spans more than a cell is a TAIL bucket: its keys thin out exponentially towards one end,
        // and a digit that divides its reduced keys evenly puts most of them into a few bins.  Factor 0 says: take the final digit
        // from the equalised coordinate itself (its bits below the bucket number follow the cells' counts).
        if (eq && first_position(f + 1u) - first_position(f) > 4096ull) x = 0ull;
        mul3[f] = (uint32_t)x;
    }
}

// ---- the partition sweep ------------------------------------------------------------------------------------------------------
// exclusive scan of 256 counters by one wave (four counters per lane)
__device__ __forceinline__ void scan256_by_wave(const uint32_t *cnt, uint32_t *base)
{
    const int lane = threadIdx.x & 63;
    const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
    uint32_t incl = c0 + c1 + c2 + c3;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    const uint32_t excl = incl - (c0 + c1 + c2 + c3);
    base[4 * lane] = excl; base[4 * lane + 1] = excl + c0; base[4 * lane + 2] = excl + c0 + c1; base[4 * lane + 3] = excl + c0 + c1 + c2;
}

// FIRST: sweep 1 (source = the table's columns, one workgroup per row slice, destination = exact slabs);
// else sweep 2 (source = a level-1 bucket of tuples, one workgroup per bucket, destination = regions of `cap2` tuples).
template <bool FIRST>
__global__ __launch_bounds__(kT) void msd_part_kernel(
    const u64 *__restrict__ col, const uint32_t *__restrict__ valcol, int64_t n, int64_t slice, u64 xorm,      // FIRST
    const uint4 *__restrict__ tin, const uint32_t *__restrict__ bstart,                                          // !FIRST
    const MsdMap *__restrict__ mapp, const uint2 *__restrict__ tab_g, int nb2log,
    uint4 *__restrict__ tout, const uint32_t *__restrict__ off1 /* FIRST: [nwg][256] */, uint32_t cap2 /* !FIRST */,
    uint32_t *__restrict__ counts2 /* !FIRST: [256 << nb2log] */, const uint32_t *__restrict__ bfirst /* !FIRST */, uint32_t *__restrict__ outoff /* !FIRST */,
    int32_t *__restrict__ flag, size_t tin_cap, size_t tout_cap /* tuples in the two buffers (checked builds) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4 *buf = reinterpret_cast<uint4 *>(lds_raw);                       // [kTile] the tile, sorted by digit
    uint4 *carry = buf + kTile;                                            // [kB][8] up to seven tuples per bucket that wait for a whole line
    uint32_t *cnt = reinterpret_cast<uint32_t *>(carry + kB * 8);          // [kB] this tile's tuples per bucket
    uint32_t *base = cnt + kB;                                             // [kB] ... their first slot in buf
    uint32_t *cur = base + kB;                                             // [kB] tuples written to the bucket's destination so far
    uint32_t *ncarry = cur + kB;                                           // [kB]
    uint32_t *big = ncarry + kB, *nbig = big + 8;                          // [8] buckets with a long run in this tile (a tile has eight at most), their number
    uint2 *s_tab = reinterpret_cast<uint2 *>(big + 16);                    // [kCells] the equalisation table (when the map is equalised)
    bool eq;
    if (msd_gave_up(flag, &eq)) return;                                    // an earlier step gave up: nothing to do
    const MsdMap m = *mapp;
    const uint2 *tab = msd_table(eq, tab_g, s_tab);
    const int nb = FIRST ? kB : (1 << nb2log);                             // buckets of this sweep
    const uint32_t b1 = blockIdx.x;
    int64_t lo, hi;
    if (FIRST) { lo = (int64_t)b1 * slice; hi = lo + slice < n ? lo + slice : n; }
    else { lo = bstart[b1]; hi = bstart[b1 + 1]; }
    if (threadIdx.x < kB) { cur[threadIdx.x] = 0u; ncarry[threadIdx.x] = 0u; }
    auto dest_of = [&](uint32_t b) -> uint4 * {
        if (FIRST) return tout + off1[(size_t)b1 * kB + b];
        return tout + ((size_t)(b1 << nb2log) + b) * cap2;
    };
    auto load = [&](int64_t t0, uint4 (&t)[kR]) {
#pragma unroll
        for (int k = 0; k < kR; k++) {
            const int64_t i = t0 + (int64_t)k * kT + threadIdx.x;
            if (FIRST) {
                if (i < hi) {
                    const u64 key = __builtin_nontemporal_load(col + i) ^ xorm;
                    t[k] = uint4{(uint32_t)key, (uint32_t)(key >> 32), (uint32_t)i, valcol ? __builtin_nontemporal_load(valcol + i) : 0u};
                } else t[k] = uint4{0u, 0u, kDead, 0u};
            } else t[k] = i < hi && MSD_CHK((size_t)i < tin_cap, 5, i) ? ld_nt16(tin + i) : uint4{0u, 0u, kDead, 0u};
        }
    };
    bool over = false;
    uint4 nx[kR];
    if (lo < hi) load(lo, nx);
    for (int64_t t0 = lo; t0 < hi; t0 += kTile) {
        uint4 t[kR];
#pragma unroll
        for (int k = 0; k < kR; k++) t[k] = nx[k];
        if (t0 + kTile < hi) load(t0 + kTile, nx);                         // the next tile's loads are in flight while this one is sorted
        if (threadIdx.x < kB) cnt[threadIdx.x] = 0u;
        if (threadIdx.x == 0) *nbig = 0u;
        lds_barrier();
        uint32_t d[kR], r[kR];
#pragma unroll
        for (int k = 0; k < kR; k++) {
            const uint32_t bk = bucket_of(((u64)t[k].y << 32) | t[k].x, m, tab);
            d[k] = FIRST ? (bk >> nb2log) : (bk & (uint32_t)(nb - 1));
            r[k] = bucket_rank(cnt, d[k], t[k].z != kDead);
        }
        lds_barrier();
        if (threadIdx.x < 64) scan256_by_wave(cnt, base);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < kR; k++) if (t[k].z != kDead && MSD_CHK(base[d[k]] + r[k] < (uint32_t)kTile, 6, base[d[k]] + r[k])) buf[base[d[k]] + r[k]] = t[k];
        lds_barrier();
        // every bucket's run leaves in whole lines: sixteen lanes per bucket; what is left (< 8 tuples) waits in `carry`.  A run of
        // kBig tuples and more (a sorted or clustered column puts a whole tile into one bucket: 256 rounds for one group of
        // sixteen lanes while the other 63 wait) is left to the whole workgroup afterwards.
        constexpr uint32_t kBig = 512;
        const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
        auto copy_out = [&](int b, uint32_t lane, uint32_t lanes, bool everyone) {
            const uint32_t cb = ncarry[b], run = cnt[b], total = cb + run, nfull = total & ~7u, at = cur[b];
            if (everyone) lds_barrier();                                   // (every wave has read the bucket's state before the first one moves it on)
            if (!FIRST && at + nfull > cap2) { over = true; return; }
            uint4 *dst = dest_of((uint32_t)b) + at;
            const uint4 *src = buf + base[b];
            for (uint32_t i = lane; i < nfull; i += lanes) if (MSD_CHK((size_t)(dst + i - tout) < tout_cap && (i < cb || base[b] + i - cb < (uint32_t)kTile), FIRST ? 1 : 2, dst + i - tout)) st_nt16(dst + i, i < cb ? carry[b * 8 + i] : src[i - cb]);
            const uint32_t rem = total - nfull, j = nfull + lane;
            uint4 keep = uint4{0u, 0u, kDead, 0u};
            if (lane < rem) keep = j < cb ? carry[b * 8 + j] : src[j - cb];
            if (lane < rem) carry[b * 8 + lane] = keep;                    // (a wave's LDS operations complete in order: the reads above came first;
                                                                           // only the first eight lanes touch `carry`)
            if (lane == 0) { cur[b] = at + nfull; ncarry[b] = rem; }
        };
        for (int b = grp; b < nb; b += kT / 16) {
            if (ncarry[b] + cnt[b] >= kBig) { if (l == 0) big[atomicAdd(nbig, 1u)] = (uint32_t)b; continue; }
            copy_out(b, (uint32_t)l, 16u, false);
        }
        lds_barrier();
        const uint32_t nbg = *nbig;
        for (uint32_t q = 0; q < nbg; q++) copy_out((int)big[q], threadIdx.x, (uint32_t)kT, true);
        lds_barrier();
    }
    // the last partial lines, padded with dead tuples
    {
        const int grp = threadIdx.x >> 3, l = threadIdx.x & 7;
        for (int b = grp; b < nb; b += kT / 8) {
            const uint32_t rem = ncarry[b], at = cur[b];
            if (rem) {
                if (!FIRST && at + 8u > cap2) over = true;
                else if (MSD_CHK((size_t)(dest_of((uint32_t)b) + at + l - tout) < tout_cap, FIRST ? 3 : 4, dest_of((uint32_t)b) + at + l - tout)) st_nt16(dest_of((uint32_t)b) + at + l, (uint32_t)l < rem ? carry[b * 8 + l] : uint4{0u, 0u, kDead, 0u});
            }
            if (!FIRST && l == 0) { counts2[((size_t)b1 << nb2log) + b] = at + rem; cnt[b] = at + rem; }
        }
    }
    if (!FIRST) {
        // where every sub-bucket's rows start in the output: the bucket's first row + the sub-buckets before it
        for (int b = nb + threadIdx.x; b < kB; b += kT) cnt[b] = 0u;
        lds_barrier();
        if (threadIdx.x < 64) scan256_by_wave(cnt, base);
        lds_barrier();
        const uint32_t first = bfirst[b1];
        for (int b = threadIdx.x; b < nb; b += kT) outoff[((size_t)b1 << nb2log) + b] = first + base[b];
    }
    if (over) atomicOr(&flag[0], 4);
}
constexpr size_t msd_part_lds() { return (size_t)kTile * 16 + (size_t)kB * 8 * 16 + (size_t)kB * 4 * 4 + 64 + (size_t)kCells * 8; }
constexpr size_t msd_final_lds(int ft) { return (size_t)kFCapOf(ft) * 16 + (size_t)(kFBinsOf(ft) + 4) * 4 + (size_t)(ft / 64) * 4 * 2 + 4 * 256 * 4; }

// ---- sweep 3: the final buckets are sorted in LDS -------------------------------------------------------------------------------
// Two workgroups per CU walk over the buckets; the NEXT bucket's tuples (and the size of the one after it) are loaded while a
// bucket is sorted -- one workgroup per bucket spent its life waiting for a chain of dependent loads (flag, size, tuples:
// ~10 us per bucket with two workgroups per CU: 1.26 ms per 1e8 rows for 0.8 ms of traffic).
constexpr int kFMine = 256;                                  // buckets per workgroup at most (D <= 65536, >= 256 workgroups)
template <int kFT>
__global__ __launch_bounds__(kFT) void msd_final_kernel(const uint4 *__restrict__ tin, uint32_t cap2, const uint32_t *__restrict__ counts2, const uint32_t *__restrict__ outoff,
                                                        const uint32_t *__restrict__ lo_h, const uint32_t *__restrict__ mul3, const uint2 *__restrict__ tab_g, uint32_t D, const MsdMap *__restrict__ mapp, u64 *__restrict__ keys_out, uint32_t *__restrict__ perm_out,
                                                        uint32_t *__restrict__ val_out, u64 out_xor, int32_t *__restrict__ flag, size_t n_rows)
{
    constexpr int kFCap = kFCapOf(kFT), kFBins = kFBinsOf(kFT), kFDigitShift = kFT == 512 ? 4 : 3;    // (15 bits of coordinate inside a bucket -> log2(kFBins) of them)
    static_assert(kFT == 512 || kFT == 1024, "two geometries");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4 *buf = reinterpret_cast<uint4 *>(lds_raw);                       // [kFCap]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(buf + kFCap);             // [kFBins + 4] counts, then exclusive offsets (cnt[kFBins] = the total)
    uint32_t *s_wave = cnt + kFBins + 4;                                   // [kFT / 64]
    uint32_t *s_c = s_wave + kFT / 64, *s_o = s_c + kFMine;                // [kFMine] each: size and first output row of this workgroup's buckets,
    uint32_t *s_lo = s_o + kFMine, *s_m3 = s_lo + kFMine;                  // ... their first reduced key and the factor of their final digit
    uint32_t *s_sq = s_m3 + kFMine;                                        // [kFT / 64]
    if (msd_gave_up(flag)) return;                                         // an earlier step gave up: nothing to do
    const MsdMap m = *mapp;
    const uint32_t step = gridDim.x;
    uint32_t f = blockIdx.x;
    if (f >= D) return;
    // kFR loads, whatever the bucket's size (a slot beyond it reads the bucket's first tuple again): a number of loads the compiler
    // can count.  With a load per live slot only, every use of a loaded tuple sat behind an s_waitcnt vmcnt(0) -- right behind the
    // loads of the NEXT bucket, which were meant to be in flight while this one is sorted.
    auto load = [&](uint32_t bucket, uint32_t count, uint4 (&t)[kFR]) {
        const uint4 *src = tin + (size_t)bucket * cap2;
#pragma unroll
        for (int k = 0; k < kFR; k++) { const uint32_t i = (uint32_t)k * kFT + threadIdx.x; t[k] = MSD_CHK(bucket < D && count <= cap2, 7, bucket) ? ld_nt16(src + (i < count ? i : 0u)) : uint4{0u, 0u, 0u, 0u}; }
    };
    // sizes and output offsets of all of this workgroup's buckets, once (a load per bucket inside the loop would sit in front of
    // every LDS wait: scalar loads and LDS operations share a counter)
    for (uint32_t j = threadIdx.x; j < (uint32_t)kFMine; j += kFT) {
        const uint32_t b = f + j * step;
        s_c[j] = b < D ? counts2[b] : 0u; s_o[j] = b < D ? outoff[b] : 0u; s_lo[j] = b < D ? lo_h[b] : 0u; s_m3[j] = b < D ? mul3[b] : 0u;
    }
    __syncthreads();
    uint32_t c = s_c[0];
    uint4 t[kFR];
    load(f, c, t);
#pragma unroll
    for (int k = 0; k < kFR; k++) asm volatile("" : "+v"(t[k].x), "+v"(t[k].y), "+v"(t[k].z), "+v"(t[k].w));   // (arrived: inside the loop t is never a pending load)
    bool dup = false;
    for (uint32_t it = 0;; it++) {
        const uint32_t fn = f + step;
        const bool more = fn < D;
        const uint32_t c_next = more ? s_c[it + 1] : 0u;
        uint4 nx[kFR];
        load(more ? fn : f, c_next, nx);                                   // in flight while this bucket is sorted (the last round loads for nobody)
        if (c) {
            // the final digit of a key: its reduced key relative to the bucket's first, spread over kFBins bins
            const uint32_t hlo = s_lo[it], m3 = s_m3[it];
            auto bin_of = [&](u64 key) -> uint32_t {
                if (m3 == 0u) {                                            // a tail bucket of the equalised map (msd_bounds): the coordinate's next bits
                    const uint32_t a = position_of(key, m);
                    uint32_t x = equalised(tab_g[a >> 12], a);
                    const uint32_t xmax = ((m.dmax + 1u) << 15) - 1u;      // (positions behind the last sampled key belong to the last bucket's last digit)
                    x = x < xmax ? x : xmax;
                    return (x >> kFDigitShift) & (uint32_t)(kFBins - 1);
                }
                const uint32_t h = reduced_key(key, m.kmin, m.sh);
                const uint32_t rel = h > hlo ? h - hlo : 0u;
                const uint32_t e = __umulhi(rel, m3);
                return e < (uint32_t)kFBins ? e : (uint32_t)kFBins - 1u;
            };
            MSD_CHK(c <= (uint32_t)kFCap, 10, c);
            for (int i = threadIdx.x; i < kFBins + 4; i += kFT) cnt[i] = 0u;
            lds_barrier();
            uint32_t e[kFR], r[kFR];
#pragma unroll
            for (int k = 0; k < kFR; k++) {
                const uint32_t i = (uint32_t)k * kFT + threadIdx.x;
                if ((uint32_t)k * kFT < c && i < c) { e[k] = bin_of(((u64)t[k].y << 32) | t[k].x); r[k] = atomicAdd(&cnt[e[k]], 1u); }   // (the first test is the wave's: whole slots are skipped)
            }
            lds_barrier();
            {   // exclusive scan of the bins: four per thread, a wave scan, the waves' totals
                const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
                const uint32_t c0 = cnt[4 * threadIdx.x], c1 = cnt[4 * threadIdx.x + 1], c2 = cnt[4 * threadIdx.x + 2], c3 = cnt[4 * threadIdx.x + 3];
                uint32_t incl = c0 + c1 + c2 + c3;
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
                // ... and the work of the ranking below, which is quadratic in the length of a run: the squares of the bins' sizes
                uint32_t sq = c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
                for (int d = 32; d; d >>= 1) sq += __shfl_xor(sq, d, 64);
                if (lane == 63) { s_wave[wv] = incl; s_sq[wv] = sq; }
                lds_barrier();
                uint32_t before = 0, work = 0;
                for (int q = 0; q < kFT / 64; q++) { if (q < wv) before += s_wave[q]; work += s_sq[q]; }
#ifdef HARK_MSD_CHECK
                if (work > kFWorkMax && threadIdx.x == 0 && atomicCAS(&flag[4], 0, 101) == 0) { flag[5] = (int32_t)f; flag[6] = (int32_t)work; flag[7] = (int32_t)c; flag[3] = (int32_t)m3; atomicOr(&flag[0], 64); }
#endif
                if (work > kFWorkMax) { if (threadIdx.x == 0) atomicOr(&flag[0], 16); return; }   // many equal keys: the caller's other path (no other
                                                                                                   // workgroup waits for this one: it may just leave)
                const uint32_t excl = before + incl - (c0 + c1 + c2 + c3);
                cnt[4 * threadIdx.x] = excl; cnt[4 * threadIdx.x + 1] = excl + c0; cnt[4 * threadIdx.x + 2] = excl + c0 + c1; cnt[4 * threadIdx.x + 3] = excl + c0 + c1 + c2;
                if (threadIdx.x == kFT - 1) cnt[kFBins] = excl + c0 + c1 + c2 + c3;
            }
            lds_barrier();
#pragma unroll
            for (int k = 0; k < kFR; k++) { const uint32_t i = (uint32_t)k * kFT + threadIdx.x; if ((uint32_t)k * kFT < c && i < c && MSD_CHK(cnt[kFBins] == c, 11, cnt[kFBins]) && MSD_CHK(cnt[e[k]] + r[k] < (uint32_t)kFCap, 9, cnt[e[k]] + r[k])) buf[cnt[e[k]] + r[k]] = t[k]; }
            lds_barrier();
            // The next bucket's tuples are taken out of the load registers HERE, before this bucket's stores are issued: loads and stores
            // share one in-order counter, so a wait for the loads behind the stores would sit out the stores' completion -- every bucket.
#pragma unroll
            for (int k = 0; k < kFR; k++) t[k] = nx[k];
            // rank inside the run of equal digits by (key, row id) and store at the rank: a tuple moves inside its run only (a few
            // rows), so a wave's stores stay within a few lines of each other
            const size_t o = s_o[it];
#pragma unroll
            for (int k = 0; k < kFR; k++) {
                const uint32_t i = (uint32_t)k * kFT + threadIdx.x;
                if ((uint32_t)k * kFT < c && i < c) {
                    const uint4 me = buf[i];
                    const u64 key = ((u64)me.y << 32) | me.x;
                    const uint32_t b = bin_of(key), a0 = cnt[b], a1 = cnt[b + 1];
                    uint32_t at = a0;
                    if (a1 - a0 > (uint32_t)kFRunMax) {
                        cnt[kFBins + 1] = 1u;                                        // (the scan left this word zero)
#ifdef HARK_MSD_CHECK
                        if (atomicCAS(&flag[4], 0, 100) == 0) { flag[5] = (int32_t)f; flag[6] = (int32_t)(a1 - a0); flag[7] = (int32_t)((b << 16) | (c & 0xFFFF)); flag[3] = (int32_t)hlo; atomicOr(&flag[0], 64); }
#endif
                    }
                    else if (a1 - a0 > 1u) for (uint32_t j = a0; j < a1; j++) {
                        const uint4 q = buf[j];
                        const u64 kj = ((u64)q.y << 32) | q.x;
                        at += (kj < key || (kj == key && q.z < me.z)) ? 1u : 0u;
                        dup |= kj == key && j != i;
                    }
                    if (MSD_CHK(o + at < n_rows && at < c, 8, o + at)) {
                        keys_out[o + at] = key ^ out_xor; perm_out[o + at] = me.z;
                        if (val_out) val_out[o + at] = me.w;
                    }
                }
            }
            lds_barrier();                                                 // (buf and cnt are the next bucket's from here; an LDS-only barrier: the
                                                                           // next bucket's loads stay in flight)
            if (cnt[kFBins + 1]) { if (threadIdx.x == 0) atomicOr(&flag[0], 16); return; }   // a long run: its rows were not ranked -- the caller's other path
        }
        else {
#pragma unroll
            for (int k = 0; k < kFR; k++) t[k] = nx[k];
        }
        if (!more) break;
        f = fn; c = c_next;
    }
    if (dup) flag[1] = 1;
}

} // namespace

// Ascending argsort of col ^ xorm (unsigned order), ties by row id: keys[i] = (col[perm[i]] ^ xorm) ^ out_xor, *val_out = valcol[perm[i]].
// *done = false when the path does not apply or gave up (the caller then takes the tuple passes): nothing is returned.
int k_sort_i64_msd(hark_context *ctx, const void *col, int64_t n, const uint32_t *valcol, uint64_t *keys, uint32_t **perm_out, uint32_t **val_out,
                   bool *done, int *unique_out, uint64_t xorm, uint64_t out_xor, int8_t *unfit /* optional, the column's hark_column::msd_unfit: read and set */)
{
    *done = false;
    if (unfit && *unfit) return HARK_OK;                                   // it gave up on this column before
    if (n < ((int64_t)1 << 20) || n >= 0xFFFFFFFFll || getenv("HARK_SORT_NO_MSD")) return HARK_OK;
    // final buckets of 800-1600 tuples on average (capacity 2560; up to 3200 and 5120 beyond 1.05e8 rows): D = 256 * nb2, nb2 a power of two <= 256
    int nb2log = 0;
    while (nb2log < 8 && n / ((int64_t)kB << nb2log) > 1600) nb2log++;
    const bool wide = n / ((int64_t)kB << nb2log) > 1600;                  // more than ~1.05e8 rows: the final workgroups of 1024 threads
    if (n / ((int64_t)kB << nb2log) > 3200) return HARK_OK;                // more than ~2.1e8 rows: the tuple passes
    const int ft = wide ? 1024 : 512;
    const int D = kB << nb2log;
    hipStream_t st = ctx->stream;
    const int nwg = ctx->num_cu > 0 && ctx->num_cu <= 1024 ? ctx->num_cu : 256;
    const int64_t slice = ((n + nwg - 1) / nwg + kTile - 1) / kTile * kTile;
    const uint32_t cap2 = (uint32_t)kFCapOf(ft);
    u64 *mm = nullptr; MsdMap *map = nullptr; int32_t *flag = nullptr;
    uint32_t *counts1 = nullptr, *off1 = nullptr, *bstart = nullptr, *bfirst = nullptr, *counts2 = nullptr, *outoff = nullptr, *cells = nullptr, *lo_h = nullptr, *mul3 = nullptr;
    uint32_t *perm = nullptr, *val = nullptr;
    uint2 *tab = nullptr;
    uint4 *slabs = nullptr, *regions = nullptr;
    unsigned char *small = nullptr;
    // one block for the small arrays: bounds 64 | map 32 | flags 32 | cells | table | counts1 | off1 | bstart | bfirst | counts2 | outoff | lo_h | mul3
    const size_t o_cells = 128, o_tab = o_cells + (size_t)kCells * 4, o_c1 = o_tab + (size_t)kCells * 8, o_off1 = o_c1 + (size_t)nwg * kB * 4,
                 o_bstart = o_off1 + (size_t)nwg * kB * 4, o_bfirst = o_bstart + (size_t)(kB + 8) * 4, o_c2 = o_bfirst + (size_t)(kB + 8) * 4,
                 o_out = o_c2 + (size_t)D * 4, o_lo = o_out + (size_t)D * 4, o_m3 = o_lo + (size_t)(D + 8) * 4, small_bytes = o_m3 + (size_t)D * 4;
    const size_t slab_tuples = (size_t)n + 8ull * kB * nwg;
    int rc = hark_alloc(ctx, (void **)&small, small_bytes);
    if (!rc) rc = hark_alloc(ctx, (void **)&slabs, slab_tuples * 16);
    if (!rc) rc = hark_alloc(ctx, (void **)&regions, (size_t)D * cap2 * 16);
    if (!rc) rc = hark_alloc(ctx, (void **)&perm, (size_t)n * 4);
    if (!rc && valcol && val_out) rc = hark_alloc(ctx, (void **)&val, (size_t)n * 4);
    auto cleanup = [&](bool keep) {
        hark_free(ctx, small); hark_free(ctx, slabs); hark_free(ctx, regions);
        if (!keep) { hark_free(ctx, perm); hark_free(ctx, val); }
    };
    if (rc == HARK_ENOMEM) { cleanup(false); ctx->err.clear(); return HARK_OK; }
    if (rc) { cleanup(false); return rc; }
    mm = reinterpret_cast<u64 *>(small); map = reinterpret_cast<MsdMap *>(small + 64); flag = reinterpret_cast<int32_t *>(small + 96);
    cells = reinterpret_cast<uint32_t *>(small + o_cells); tab = reinterpret_cast<uint2 *>(small + o_tab);
    counts1 = reinterpret_cast<uint32_t *>(small + o_c1); off1 = reinterpret_cast<uint32_t *>(small + o_off1);
    bstart = reinterpret_cast<uint32_t *>(small + o_bstart); bfirst = reinterpret_cast<uint32_t *>(small + o_bfirst);
    counts2 = reinterpret_cast<uint32_t *>(small + o_c2); outoff = reinterpret_cast<uint32_t *>(small + o_out);
    lo_h = reinterpret_cast<uint32_t *>(small + o_lo); mul3 = reinterpret_cast<uint32_t *>(small + o_m3);
    static const u64 mm_init[2] = {~0ull, 0ull};                          // (static: an asynchronous copy may read it after an early return)
    HIP_TRY_RC(ctx, rc, hipMemsetAsync(small + 64, 0, 64 + (size_t)kCells * 4, st));   // map, flags (flag[0] gave up, [1] equal keys exist, [2] equalised map), cells
    HIP_TRY_RC(ctx, rc, hipMemcpyAsync(mm, mm_init, 16, hipMemcpyHostToDevice, st));
    const u64 *c64 = static_cast<const u64 *>(col);
    const size_t lds = msd_part_lds();
    static bool lds_attributes_of[64] = {};                          // once per device (the sizes are constants)
    bool &lds_attributes_set = lds_attributes_of[ctx->device & 63];
    if (!lds_attributes_set) {
        HIP_TRY_RC(ctx, rc, hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_part_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY_RC(ctx, rc, hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_part_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY_RC(ctx, rc, hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_final_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)msd_final_lds(512)));
        HIP_TRY_RC(ctx, rc, hipFuncSetAttribute(reinterpret_cast<const void *>(&msd_final_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)msd_final_lds(1024)));
        lds_attributes_set = !rc;
    }
    const dim3 g1((unsigned)nwg), b1(kT);
    HARK_LAUNCH_RC(ctx, rc, msd_sample_kernel<<<dim3(kSampleWg), dim3(1024), 0, st>>>(c64, n, xorm, kSampleStride, mm));
    HARK_LAUNCH_RC(ctx, rc, msd_setup_kernel<<<dim3(1), dim3(1), 0, st>>>(mm, (uint32_t)D, map, flag));
    // the slab sizes under the affine map; a lumpy distribution asks for the equalised map (flag[2]) and is counted again: the second
    // round's kernels return at once when nobody asked
    HARK_LAUNCH_RC(ctx, rc, msd_hist_kernel<<<g1, b1, msd_hist_lds(), st>>>(0, c64, n, slice, xorm, map, tab, nb2log, counts1, cells, flag));
    HARK_LAUNCH_RC(ctx, rc, msd_scan1_kernel<<<dim3(1), dim3(1024), 0, st>>>(0, counts1, nwg, off1, bstart, bfirst, n, flag));
    HARK_LAUNCH_RC(ctx, rc, msd_eq_kernel<<<dim3(1), dim3(1024), 0, st>>>(cells, n, (uint32_t)D, tab, flag));
    HARK_LAUNCH_RC(ctx, rc, msd_hist_kernel<<<g1, b1, msd_hist_lds(), st>>>(1, c64, n, slice, xorm, map, tab, nb2log, counts1, cells, flag));
    HARK_LAUNCH_RC(ctx, rc, msd_scan1_kernel<<<dim3(1), dim3(1024), 0, st>>>(1, counts1, nwg, off1, bstart, bfirst, n, flag));
    HARK_LAUNCH_RC(ctx, rc, msd_bounds_kernel<<<dim3((unsigned)(D / 256 + 1)), dim3(256), 0, st>>>(map, tab, (uint32_t)D, (uint32_t)kFBinsOf(ft), lo_h, mul3, flag));
    HARK_LAUNCH_RC(ctx, rc, msd_part_kernel<true><<<g1, b1, lds, st>>>(c64, val ? valcol : nullptr, n, slice, xorm, nullptr, nullptr, map, tab, nb2log, slabs, off1, 0u, nullptr, nullptr, nullptr, flag, 0, slab_tuples));
    HARK_LAUNCH_RC(ctx, rc, msd_part_kernel<false><<<dim3(kB), b1, lds, st>>>(nullptr, nullptr, n, 0, 0ull, slabs, bstart, map, tab, nb2log, regions, nullptr, cap2, counts2, bfirst, outoff, flag, slab_tuples, (size_t)D * cap2));
    // >= 256 workgroups (<= 256 buckets each); three workgroups of 512 threads or one of 1024 per CU
    const int per_cu = wide ? 1 : 3;
    const int fgrid = D < per_cu * 256 ? D : (per_cu * nwg >= 256 && per_cu * nwg <= D ? per_cu * nwg : 256);
    if (wide) HARK_LAUNCH_RC(ctx, rc, msd_final_kernel<1024><<<dim3((unsigned)fgrid), dim3(1024), msd_final_lds(1024), st>>>(regions, cap2, counts2, outoff, lo_h, mul3, tab, (uint32_t)D, map, reinterpret_cast<u64 *>(keys), perm, val, out_xor, flag, (size_t)n));
    else HARK_LAUNCH_RC(ctx, rc, msd_final_kernel<512><<<dim3((unsigned)fgrid), dim3(512), msd_final_lds(512), st>>>(regions, cap2, counts2, outoff, lo_h, mul3, tab, (uint32_t)D, map, reinterpret_cast<u64 *>(keys), perm, val, out_xor, flag, (size_t)n));
    int64_t verdict = 0;
    if (!rc) rc = hark_read_words(ctx, flag, &verdict, 1);
    if (rc) { cleanup(false); return rc; }
    if (getenv("HARK_SORT_MSD_VERBOSE")) {                                           // experiments: what the path decided
        int32_t fl[8]; hark_d2h(ctx, fl, flag, 32);
        fprintf(stderr, "msd sort: n=%lld D=%d gave_up=%x equal_keys=%d equalised=%d\n", (long long)n, D, fl[0], fl[1], fl[2]);
    }
#ifdef HARK_MSD_CHECK
    if (verdict & 64) { int32_t fl[8]; hark_d2h(ctx, fl, flag, 32); fprintf(stderr, "MSD CHECK: violation code %d value %d block %d thread %x (n=%lld D=%d nb2log=%d flags %x equalised %d word3 %u)\n", fl[4], fl[5], fl[6], fl[7], (long long)n, D, nb2log, fl[0], fl[2], (unsigned)fl[3]);
        if (fl[4] == 100 || fl[4] == 101) { std::vector<uint32_t> lo((size_t)D + 1), m3v((size_t)D); hark_d2h(ctx, lo.data(), lo_h, ((size_t)D + 1) * 4); hark_d2h(ctx, m3v.data(), mul3, (size_t)D * 4); int ff = fl[5]; fprintf(stderr, "   bucket %d: lo_h %u next %u mul3 %u (prev lo %u)\n", ff, lo[ff], lo[ff + 1], m3v[ff], ff ? lo[ff - 1] : 0u); } }
#endif
    if ((verdict & 0xFFFFFFFFll) != 0) {                                             // did not fit: the tuple passes
        if (unfit && ((verdict & 0xFFFFFFFFll) & ~2ll) != 0) *unfit = 1;            // (bit 1: keys within 2^32 of each other -- nothing was lost)
        cleanup(false); return HARK_OK;
    }
    if (unique_out) *unique_out = ((verdict >> 32) & 0xFFFFFFFFll) ? 0 : 1;
    cleanup(true);
    *perm_out = perm;
    if (val_out) *val_out = val;
    *done = true;
    return HARK_OK;
}
