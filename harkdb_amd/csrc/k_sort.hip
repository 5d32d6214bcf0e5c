// k_sort.hip -- stable LSD radix sort of (u32 key, u32 payload) pairs, 8-bit
// digits (4 passes), plus the key transforms that reduce i32 / f32 / i64 and
// descending orders to it.
//
// Replaces the reference's `rsort`: 32 passes of a 1-bit stable split, each
// with two scans, a reduce and a scatter of whole rows (groupby.fut:8-22,
// join.fut:9-23).  Same result (stable, ascending UNSIGNED key), 8x fewer
// passes, and only (key, row id) moves -- rows are gathered once at the end.
//
// One pass = three launches over a fixed decomposition of the input into
// `nblk` contiguous slices (one workgroup each):
//   1. digit_hist_kernel   per-slice 256-bin digit histogram -> hist[bin][blk]
//   2. scan_hist_rows_kernel  per-digit exclusive scan over the slices (+ digit totals;
//                          the scatter kernel prefixes the 256 totals itself)
//   3. digit_scatter_kernel walks its slice in tiles (4096 or 12288 keys, see below); inside a tile a
//      wave ranks each key among equal digits with a wave64 ballot match (8
//      ballots + popcount of the lower lanes), waves are chained by a prefix
//      over per-wave digit counts, keys are staged digit-sorted in LDS and
//      written out as contiguous runs.  Positions are assigned strictly in
//      input order, so the pass is stable.
#include "hark_internal.h"
#include "sort_networks.h"
#include <type_traits>

namespace {

// Two geometries of a pass.  A digit run of a tile is tile / 256 keys long; runs shorter than a 128-byte line leave the
// CU as partial lines, and whether those merge in the XCD's L2 before they are evicted depends on how many workgroups
// keep 2 x 256 output lines open at once.  Measured (1e8 keys, three passes + the difference mask, one box):
// 256 threads x 16 keys, 8 slices per CU 2.44 ms; 512 x 16, 2 per CU 2.38; 1024 x 8, 1 per CU 2.26; 1024 x 12, 1 per CU
// 2.13 ms; with the next tile's keys requested one tile ahead 2.07 (profiles/r02_notes.md §9).  Sorts of a few million
// keys are faster with the small tiles (more workgroups in flight), so the large geometry is used from 2^25 keys on.
constexpr int kBins = 256;
struct GeoSmall { static constexpr int T = 256, R = 16, PER_CU = 8; };
struct GeoLarge { static constexpr int T = 1024, R = 12, PER_CU = 1; };
constexpr int64_t kLargeSortFrom = (int64_t)1 << 25;
template <typename GEO> constexpr size_t scatter_lds() { return (size_t)GEO::T * GEO::R * 8 + (size_t)(GEO::T / 64) * kBins * 8 + kBins * 4 + kBins * 8 + 64; }

// HARK_SORT_TILED=1 selects the scatter kernel of rounds 1-2 (digit_scatter_kernel: ballot match) for A/B runs; the default is
// digit_scatter2_kernel (LDS match).  Measured and dropped in round 3 (profiles/r03_notes.md): two 512-thread workgroups per
// CU (either kernel: slower, more output streams), 512 threads x 24 keys, and a kernel that write-combines digit runs across
// tiles in per-digit LDS rings and stores whole 128-byte lines only (exact HBM traffic, no faster).
// The passes that sort keys differing in the bits of `diff`.  Byte plan: one 8-bit pass per key byte with a differing bit
// (any holes in the mask are skipped).  Range plan: the differing bits span [lo, hi] -> ceil(bits / 8) passes over digits
// of EQUAL width (20 bits: 7 + 7 + 6 instead of 8 + 8 + 4; bits 4..19: two passes where the byte plan needs three).  The range
// plan is taken when it needs fewer passes, or as many with narrower digits (fewer, longer digit runs per tile;
// HARK_SORT_PLAN=byte switches it off for A/B runs).
struct SortPass { int shift, width; };
static int plan_passes(uint32_t diff, SortPass *out)
{
    if (!diff) return 0;
    int nb = 0;
    SortPass bytes[4];
    for (int b = 0; b < 4; b++) if ((diff >> (8 * b)) & 0xFFu) bytes[nb++] = SortPass{8 * b, 8};
    const int lo = __builtin_ctz(diff), hi = 31 - __builtin_clz(diff), bits = hi - lo + 1;
    const int P = (bits + 7) / 8, w = (bits + P - 1) / P;
    static const bool byte_only = getenv("HARK_SORT_PLAN") && !strcmp(getenv("HARK_SORT_PLAN"), "byte");
    if (!byte_only && (P < nb || (P == nb && w < 8))) {
        for (int i = 0; i < P; i++) { const int sh = lo + i * w, left = hi + 1 - sh; out[i] = SortPass{sh, left < w ? left : w}; }
        return P;
    }
    for (int i = 0; i < nb; i++) out[i] = bytes[i];
    return nb;
}

static bool old_scatter()
{
    static const bool v = getenv("HARK_SORT_TILED") && atoi(getenv("HARK_SORT_TILED")) == 1;
    return v;
}

// slices (= workgroups) of a pass over n keys and the keys per slice (whole tiles; 12288 is a multiple of every large tile)
static void sort_geometry(int64_t n, int num_cu, bool *large, int64_t *nblk_out, int64_t *slice_out)
{
    const bool lg = n >= kLargeSortFrom;
    const int64_t tile = lg ? GeoLarge::T * GeoLarge::R : GeoSmall::T * GeoSmall::R, per_cu = lg ? GeoLarge::PER_CU : GeoSmall::PER_CU;
    int64_t nblk = (n + tile - 1) / tile;
    if (nblk > (int64_t)num_cu * per_cu) nblk = (int64_t)num_cu * per_cu;
    if (nblk < 1) nblk = 1;
    int64_t slice = (n + nblk - 1) / nblk;
    slice = (slice + tile - 1) / tile * tile;
    nblk = (n + slice - 1) / slice;
    if (nblk < 1) nblk = 1;
    *large = lg; *nblk_out = nblk; *slice_out = slice;
}

// Lanes of the wave whose digit equals this lane's digit (among `valid` lanes).
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t vote = __ballot(bit);
        m &= bit ? vote : ~vote;
    }
    return m;
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

template <typename GEO>
__global__ __launch_bounds__(GEO::T) void digit_hist_kernel(const uint32_t *__restrict__ keys, int64_t n, int64_t slice,
                                                                  int shift, uint32_t xor_mask, uint32_t *__restrict__ hist, int nblk)
{
    const int sh = shift & 255;                                   // `shift` carries the digit's width in bits 8.. (0 = 8 bits)
    const uint32_t dmask = (shift >> 8) ? (1u << (shift >> 8)) - 1u : 255u;
    // plain ds_add_u32 per key (~10 lanes/clk/CU on gfx950); the slice starts on a
    // multiple of 4096 keys, so 16-byte loads are aligned
    __shared__ uint32_t s_hist[kBins];
    if (threadIdx.x < kBins) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice;
    const int64_t hi = lo + slice < n ? lo + slice : n;
    const int64_t nvec = (hi - lo) / 4;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + lo);
    auto count4 = [&](const uint4 q) {
        atomicAdd(&s_hist[((q.x ^ xor_mask) >> sh) & dmask], 1u);
        atomicAdd(&s_hist[((q.y ^ xor_mask) >> sh) & dmask], 1u);
        atomicAdd(&s_hist[((q.z ^ xor_mask) >> sh) & dmask], 1u);
        atomicAdd(&s_hist[((q.w ^ xor_mask) >> sh) & dmask], 1u);
    };
    int64_t i = threadIdx.x;
    for (; i + 3 * (int64_t)blockDim.x < nvec; i += 4 * (int64_t)blockDim.x) {     // four 16-byte loads in flight per lane (read once: non-temporal)
        const uint4 a = ld_nt16(k4 + i), b = ld_nt16(k4 + i + blockDim.x), c = ld_nt16(k4 + i + 2 * blockDim.x), d = ld_nt16(k4 + i + 3 * blockDim.x);
        count4(a); count4(b); count4(c); count4(d);
    }
    for (; i < nvec; i += blockDim.x) count4(k4[i]);
    for (int64_t i = lo + nvec * 4 + threadIdx.x; i < hi; i += blockDim.x)
        atomicAdd(&s_hist[((keys[i] ^ xor_mask) >> sh) & dmask], 1u);
    __syncthreads();
    if (threadIdx.x < kBins) hist[(size_t)threadIdx.x * nblk + blockIdx.x] = s_hist[threadIdx.x];
}

// The histogram of the keys' LOW byte per slice and the bits in which any key differs from the first one (a pass over a
// byte in which all keys agree is skipped) from ONE read of the keys: replaces the difference-mask pass AND the histogram
// launch of the first pass whenever that pass starts at bit 0 (narrower digits fold the rows: fold_hist_rows_kernel).
// hist0[bin][blk].  (A first version counted all four bytes -- only the first pass can use a histogram of the INPUT order,
// and three more LDS atomics per key made this the slowest read of the sort: 139 us per 1e8 keys against 71 for
// digit_hist_kernel.)
// diff[1] != 0: the keys are NOT in ascending order of (key ^ xor_mask) -- an input that is (a table kept in key order, a
// dimension table sorted by its primary key, the result of a GROUP BY) needs no pass at all: the stable sort of a sorted
// sequence is the sequence.  Every thread checks its own 16 bytes and the word behind them (the next lane's: a shuffle; the
// wave's last lane reads it) until it has seen a descent: on shuffled keys that is its first load.
__global__ __launch_bounds__(1024) void multi_hist_kernel(const uint32_t *__restrict__ keys, int64_t n, int64_t slice, uint32_t xor_mask,
                                                          uint32_t *__restrict__ hist0, int nblk, uint32_t *__restrict__ diff)
{
    __shared__ uint32_t s_hist[kBins];
    __shared__ uint32_t s_acc;
    for (int i = threadIdx.x; i < kBins; i += blockDim.x) s_hist[i] = 0u;
    if (threadIdx.x == 0) s_acc = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice;
    const int64_t hi = lo + slice < n ? lo + slice : n;
    const int64_t nvec = (hi - lo) / 4;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + lo);
    const uint32_t w0 = keys[0];
    uint32_t acc = 0u;
    // A low byte in which every key of the wave agrees (keys that are multiples of 256: 64 lanes adding to ONE counter
    // serialise) is counted by one lane.
    auto count1 = [&](uint32_t k) {
        acc |= k ^ w0;
        const uint32_t d = (k ^ xor_mask) & 255u;
        const uint64_t active = __ballot(true);
        if (__ballot(d != (uint32_t)__builtin_amdgcn_readfirstlane((int)d)) == 0ull) { if ((active & lanemask_lt()) == 0ull) atomicAdd(&s_hist[d], (uint32_t)__popcll(active)); }
        else atomicAdd(&s_hist[d], 1u);
    };
    auto count4 = [&](const uint4 q) { count1(q.x); count1(q.y); count1(q.z); count1(q.w); };
    bool descent = false;
    const int lane = threadIdx.x & 63;
    // q = keys[lo + 4 at .. + 4): in order, and not above the word behind them?  `wave`: the lanes of the wave hold consecutive 16
    // bytes (the next lane's first word is the word behind this lane's last)
    auto ordered4 = [&](const uint4 q, int64_t at, bool wave) {
        const uint32_t x = q.x ^ xor_mask, y = q.y ^ xor_mask, z = q.z ^ xor_mask, w = q.w ^ xor_mask;
        const bool all = wave && __ballot(true) == ~0ull;             // (the loop's last round may have left some lanes behind)
        uint32_t nx = all ? (uint32_t)__shfl_down((int)x, 1, 64) : 0u;
        if (descent) return;
        if (!all || lane == 63) { const int64_t g = lo + 4 * (at + 1); nx = g < n ? keys[g] ^ xor_mask : 0xFFFFFFFFu; }
        descent = x > y || y > z || z > w || w > nx;
    };
    int64_t i = threadIdx.x;
    for (; i + 3 * (int64_t)blockDim.x < nvec; i += 4 * (int64_t)blockDim.x) {
        const uint4 a = ld_nt16(k4 + i), b = ld_nt16(k4 + i + blockDim.x), c = ld_nt16(k4 + i + 2 * blockDim.x), d = ld_nt16(k4 + i + 3 * blockDim.x);
        count4(a); count4(b); count4(c); count4(d);
        ordered4(a, i, true); ordered4(b, i + blockDim.x, true); ordered4(c, i + 2 * (int64_t)blockDim.x, true); ordered4(d, i + 3 * (int64_t)blockDim.x, true);
    }
    for (; i < nvec; i += blockDim.x) { const uint4 a = k4[i]; count4(a); ordered4(a, i, false); }
    for (int64_t j = lo + nvec * 4 + threadIdx.x; j < hi; j += blockDim.x) {
        const uint32_t k = keys[j];
        count1(k);
        if (j + 1 < n && (k ^ xor_mask) > (keys[j + 1] ^ xor_mask)) descent = true;
    }
    if (descent && __hip_atomic_load(diff + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(diff + 1, 1u);
    for (int d = 32; d > 0; d >>= 1) acc |= __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr(&s_acc, acc);
    __syncthreads();
    for (int i2 = threadIdx.x; i2 < kBins; i2 += blockDim.x) hist0[(size_t)i2 * nblk + blockIdx.x] = s_hist[i2];
    if (threadIdx.x == 0) {
        const uint32_t mine = s_acc;
        if (mine & ~__hip_atomic_load(diff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(diff, mine);
    }
}

// The low-byte histogram as the histogram of a digit of `width` < 8 bits at bit 0: row d += rows d + j * 2^width, the folded
// rows cleared.  One workgroup per row of the result.
__global__ __launch_bounds__(256) void fold_hist_rows_kernel(uint32_t *__restrict__ hist, int nblk, int width)
{
    const int d = blockIdx.x, step = 1 << width;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        uint32_t sum = hist[(size_t)d * nblk + i];
        for (int r = d + step; r < kBins; r += step) { sum += hist[(size_t)r * nblk + i]; hist[(size_t)r * nblk + i] = 0u; }
        hist[(size_t)d * nblk + i] = sum;
    }
}

// Exclusive scan of each digit's row hist[d][0..nblk) (one workgroup per digit)
// plus the row total; the scatter kernel adds the prefix over digit totals itself.
__global__ __launch_bounds__(256) void scan_hist_rows_kernel(uint32_t *__restrict__ hist, int nblk, uint32_t *__restrict__ row_total)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblk; base += 1024) {               // 4 counters per thread per step
        const int i = base + threadIdx.x * 4;
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = i + j < nblk ? row[i + j] : 0u;
        const uint32_t tsum = x[0] + x[1] + x[2] + x[3];
        uint32_t incl = tsum;
        for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = s_carry;
        for (int w = 0; w < wave; w++) run += s_wave[w];
        run += incl - tsum;
#pragma unroll
        for (int j = 0; j < 4; j++) { if (i + j < nblk) row[i + j] = run; run += x[j]; }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = s_carry;
}

// IOTA (vals_in == nullptr): "payload = input position" (first pass of an argsort) -- compiled in, so that no
// select between a loaded and a computed payload sits in the tile loop (it would drain the loads at the merge).
template <typename GEO, bool IOTA>
__global__ __launch_bounds__(GEO::T) void digit_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
    int64_t n, int64_t slice, int shift, uint32_t xor_mask, const uint32_t *__restrict__ hist, int nblk,
    const uint32_t *__restrict__ row_total)
{
    const int sh = shift & 255;                                   // `shift` carries the digit's width in bits 8.. (0 = 8 bits)
    const uint32_t dmask = (shift >> 8) ? (1u << (shift >> 8)) - 1u : 255u;
    constexpr int kSortThreads = GEO::T, kSortWaves = GEO::T / 64, kRounds = GEO::R, kSortTile = GEO::T * GEO::R;
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    uint32_t *s_key = reinterpret_cast<uint32_t *>(sort_lds);                         // [kSortTile]
    uint32_t *s_val = s_key + kSortTile;                                             // [kSortTile]
    uint32_t (*s_wcnt)[kBins] = reinterpret_cast<uint32_t (*)[kBins]>(s_val + kSortTile);       // [waves][bins] per-wave digit counts of the tile
    uint32_t (*s_wbase)[kBins] = s_wcnt + kSortWaves;                                // [waves][bins] tile-local start of (wave, digit)
    uint32_t *s_tstart = reinterpret_cast<uint32_t *>(s_wbase + kSortWaves);         // [bins] tile-local start of each digit
    int64_t *s_gpos = reinterpret_cast<int64_t *>(s_tstart + kBins);                 // [bins] global position of the next key of each digit
    uint32_t *s_scan = reinterpret_cast<uint32_t *>(s_gpos + kBins);                 // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = (int64_t)blockIdx.x * slice;
    const int64_t hi = lo + slice < n ? lo + slice : n;
    {   // first output position of (digit tid, this slice) = digits before + this digit's earlier slices
        const uint32_t tot = tid < kBins ? row_total[tid] : 0u;
        uint32_t incl = tot;
        for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        if (lane == 63 && tid < kBins) s_scan[wave] = incl;
        __syncthreads();
        if (tid < kBins) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += s_scan[w];
            s_gpos[tid] = (int64_t)(carry + incl - tot) + (int64_t)hist[(size_t)tid * nblk + blockIdx.x];
        }
        __syncthreads();
    }
    const uint64_t lt = lanemask_lt();

    // GeoLarge (one workgroup per CU: nothing else covers the latency of the loads): the NEXT tile's keys are requested
    // while this one is ranked, staged and written, and this tile's payloads -- first needed when the keys are staged --
    // arrive under the ranking.  (Keys AND payloads one tile ahead spill registers at 1024 threads.)
    constexpr bool kAhead = GEO::PER_CU == 1;
    constexpr int kN = kAhead ? kRounds : 1;
    uint32_t nkey[kN];
    auto fetch_keys = [&](int64_t tb) {
        const int64_t wb = tb + (int64_t)wave * (64 * kRounds);
#pragma unroll
        for (int r = 0; r < kN; r++) { const int64_t i = wb + r * 64 + lane; nkey[r] = keys_in[i < hi ? i : hi - 1]; }      // clamped, not predicated: no branch per load
    };
    if (kAhead) fetch_keys(lo);
    for (int64_t tbase = lo; tbase < hi; tbase += kSortTile) {
        for (int i = tid; i < kSortWaves * kBins; i += kSortThreads) (&s_wcnt[0][0])[i] = 0;
        lds_barrier();
        // ---- rank inside the wave's contiguous 1024-key chunk -----------------
        uint32_t key[kRounds], val[kRounds], rank[kRounds];
        const int64_t wbase = tbase + (int64_t)wave * (64 * kRounds);
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const int64_t i = wbase + r * 64 + lane;
            const bool valid = i < hi;
            if (kAhead) key[r] = nkey[kAhead ? r : 0];
            else key[r] = valid ? keys_in[i] : 0u;
            if (!kAhead) val[r] = valid ? (IOTA ? (uint32_t)i : vals_in[i]) : 0u;
            rank[r] = valid ? 0u : 0xFFFFFFFFu;
        }
        if (kAhead) {
#pragma unroll
            for (int r = 0; r < kRounds; r++) { const int64_t i = wbase + r * 64 + lane; val[r] = IOTA ? (uint32_t)i : vals_in[i < hi ? i : hi - 1]; }
        }
        if (kAhead && tbase + kSortTile < hi) fetch_keys(tbase + kSortTile);
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const bool valid = rank[r] != 0xFFFFFFFFu;
            const uint32_t d = ((key[r] ^ xor_mask) >> sh) & dmask;
            const uint64_t peers = match_digit(d, valid);
            if (valid) {
                const uint32_t before = s_wcnt[wave][d];            // count from earlier rounds (wave-private row)
                rank[r] = before + (uint32_t)__popcll(peers & lt);
                if ((peers & lt) == 0) s_wcnt[wave][d] = before + (uint32_t)__popcll(peers);
            }
            // the leader's store above must land before the next round's loads of the same row
            __builtin_amdgcn_wave_barrier();
        }
        lds_barrier();
        // ---- chain the waves, lay the digits out in the tile -------------------
        uint32_t tcnt = 0;
        {
            uint32_t run = 0;
            if (tid < kBins) {
#pragma unroll
                for (int w = 0; w < kSortWaves; w++) { s_wbase[w][tid] = run; run += s_wcnt[w][tid]; }
            }
            tcnt = run;                                             // keys of digit `tid` in this tile
            uint32_t incl = tcnt;
            for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
            if (lane == 63 && tid < kBins) s_scan[wave] = incl;
            lds_barrier();
            if (tid < kBins) {
                uint32_t carry = 0;
                for (int w = 0; w < wave; w++) carry += s_scan[w];
                s_tstart[tid] = carry + incl - tcnt;
            }
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            if (rank[r] != 0xFFFFFFFFu) {
                const uint32_t d = ((key[r] ^ xor_mask) >> sh) & dmask;
                const uint32_t slot = s_tstart[d] + s_wbase[wave][d] + rank[r];
                s_key[slot] = key[r]; s_val[slot] = val[r];
            }
        }
        lds_barrier();
        // ---- write digit runs ---------------------------------------------------
        const int tile_n = (int)((hi - tbase) < kSortTile ? (hi - tbase) : kSortTile);
        for (int slot = tid; slot < tile_n; slot += kSortThreads) {
            const uint32_t kk = s_key[slot];
            const uint32_t d = ((kk ^ xor_mask) >> sh) & dmask;
            const int64_t pos = s_gpos[d] + (slot - (int)s_tstart[d]);
            keys_out[pos] = kk; vals_out[pos] = s_val[slot];
        }
        lds_barrier();
        if (tid < kBins) s_gpos[tid] += tcnt;
        // (the __syncthreads at the top of the next tile orders this update)
    }
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and broadcasts (gfx9 family): six vector adds with
// the data movement folded in -- no cross-lane index registers, which a shuffle-based scan keeps live (and spilled: they
// are loop invariants) across a whole kernel.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2 and 3
    return v;
}

// ---- the pass for large inputs, second edition ----------------------------------------------------------------------
// PMC counters of digit_scatter_kernel<GeoLarge> (profiles/r03_notes.md): 216 M vector instructions per 1e8-key pass
// -- 138 per 64 keys -- keep the four SIMDs of a CU busy for 0.35 of the pass's 0.46 ms: the pass is bound by instruction
// issue, not by its 16 B per key.  Most of them are the ballot match (eight ballots whose per-lane 64-bit selects are
// vector instructions) and 64-bit position arithmetic.  Here
//   * the lanes of a wave that share a digit are found through LDS: every lane ORs its lane bit into a per-wave table of
//     256 64-bit masks (ds_or_b64), reads its digit's mask back and clears it -- three LDS instructions instead of ~56
//     vector ones, and still a pure function of the digits (a set union: no ordering assumption on the atomics);
//   * keys and payloads are staged as one 8-byte word, positions are 32-bit (n < 2^32), and the write-out needs ONE table
//     read per key (delta[d] = first output position of the digit - its start in the tile);
//   * validity tests only run in a slice's last tile.
// Same result as digit_scatter_kernel (positions are assigned in input order: stable).
template <typename GEO> constexpr size_t scatter2_lds()
{
    return (size_t)GEO::T * GEO::R * 8 + (size_t)(GEO::T / 64) * kBins * 8 + (size_t)(GEO::T / 64) * kBins * 2 * 2 + (size_t)kBins * 4 * 3 + 64;
}
template <typename GEO, bool IOTA>
__global__ __launch_bounds__(GEO::T) void digit_scatter2_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
    int64_t n, int64_t slice, int shift, uint32_t xor_mask, const uint32_t *__restrict__ hist, int nblk,
    const uint32_t *__restrict__ row_total)
{
    const int sh = shift & 255;                                   // `shift` carries the digit's width in bits 8.. (0 = 8 bits)
    const uint32_t dmask = (shift >> 8) ? (1u << (shift >> 8)) - 1u : 255u;
    constexpr int T = GEO::T, W = GEO::T / 64, R = GEO::R, TILE = GEO::T * GEO::R;
    typedef unsigned long long u64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    uint2 *s_kv = reinterpret_cast<uint2 *>(sort_lds);                              // [TILE] (key, payload), digit-sorted
    uint32_t *s_mask = reinterpret_cast<uint32_t *>(s_kv + TILE);                   // [W][2][bins] lanes of the wave holding the digit, low / high 32 lanes (all zero between rounds)
    uint16_t (*s_wcnt)[kBins] = reinterpret_cast<uint16_t (*)[kBins]>(s_mask + (size_t)W * 2 * kBins);   // [W][bins] per-wave digit counts of the tile
    uint16_t (*s_wbase)[kBins] = s_wcnt + W;                                        // [W][bins] first tile slot of (wave, digit)
    uint32_t *s_tstart = reinterpret_cast<uint32_t *>(s_wbase + W);                 // [bins] start of the digit in the tile
    uint32_t *s_delta = s_tstart + kBins;                                           // [bins] output position of the digit's next key - s_tstart
    uint32_t *s_gpos = s_delta + kBins;                                             // [bins] output position of the digit's next key
    uint32_t *s_scan = s_gpos + kBins;                                              // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = (int64_t)blockIdx.x * slice;
    const int64_t hi = lo + slice < n ? lo + slice : n;
    for (int i = tid; i < W * 2 * kBins; i += T) s_mask[i] = 0u;
    {   // first output position of (digit tid, this slice) = digits before + this digit's earlier slices
        const uint32_t tot = tid < kBins ? row_total[tid] : 0u;
        const uint32_t incl = wave_incl_scan(tot);
        if (lane == 63 && tid < kBins) s_scan[wave] = incl;
        __syncthreads();
        if (tid < kBins) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += s_scan[w];
            s_gpos[tid] = carry + incl - tot + hist[(size_t)tid * nblk + blockIdx.x];
        }
        __syncthreads();
    }
    // The mask of a digit is kept as two 32-bit words in two tables (lanes 0..31 / 32..63): a lane ORs its bit into ONE
    // 4-byte word, so the 64 lanes of a round spread over all 64 LDS banks -- with one 8-byte word per digit they shared 32
    // bank pairs, two cycles each (SQ_LDS_BANK_CONFLICT was 60 % of the LDS cycles of a pass over a 256-valued digit).
    const uint32_t lanebit = 1u << (lane & 31);
    const u64 lt = lanemask_lt();
    uint32_t *mymask = s_mask + (size_t)wave * 2 * kBins;         // (not volatile: the address-space inference skips volatile accesses -> FLAT instructions)
    uint32_t *myhalf = mymask + (lane >> 5) * kBins;
    uint16_t *mycnt = s_wcnt[wave];
    uint32_t nkey[R];
    // One tile.  FULL (every tile but a slice's last, ragged one) carries no validity tests and a FIXED number of loads and
    // stores per lane, so that the compiler can count them: loads and stores share one in-order counter on gfx950, and with
    // a countable stream the wait for the prefetched keys leaves the previous tile's 2 x R stores in flight (s_waitcnt
    // vmcnt(n) instead of vmcnt(0)) -- they drain under this tile's ranking.  The ragged tile is peeled out of the loop
    // for the same reason (a path with an unknown number of stores into the loop head would force vmcnt(0) there).
    auto tile = [&](int64_t tbase, auto full_tag, auto next_tag) {
        constexpr bool FULL = decltype(full_tag)::value, NEXT_FULL = decltype(next_tag)::value;
        const int64_t wbase = tbase + (int64_t)wave * (64 * R);
        uint32_t key[R], val[R], rank[R];
#pragma unroll
        for (int r = 0; r < R; r++) key[r] = nkey[r];
        // the wave's own row of digit counts (its readers of the previous tile passed that tile's second barrier)
        reinterpret_cast<uint2 *>(mycnt)[lane] = uint2{0u, 0u};
        // ---- rank inside the wave's contiguous chunk: lanes with my digit = the mask the wave ORs together in LDS ------
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool valid = FULL || wbase + r * 64 + lane < hi;
            const uint32_t d = ((key[r] ^ xor_mask) >> sh) & dmask;
            rank[r] = 0xFFFFFFFFu;
            if (valid) {
                // relaxed atomics keep the three accesses in program order for the compiler; the LDS executes a wave's
                // instructions in order, so the read sees the whole wave's ORs and the clear follows every lane's read
                __hip_atomic_fetch_or(myhalf + d, lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);      // ds_or_b32
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // (no instruction: plain loads, so that the two words come with ONE ds_read2st64_b32)
                const uint32_t plo = mymask[d], phi = mymask[kBins + d];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const u64 peers = ((u64)phi << 32) | plo;
                __hip_atomic_store(myhalf + d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);              // clean again for the next round
                const u64 below = peers & lt;
                const uint32_t before = mycnt[d];
                rank[r] = before + (uint32_t)__popcll(below);
                if (below == 0ull) mycnt[d] = (uint16_t)(before + (uint32_t)__popcll(peers));
            }
            __builtin_amdgcn_wave_barrier();
        }
        // this tile's payloads (first needed when the keys are staged) and the next tile's keys travel under the rest of the tile
        if (IOTA) {
#pragma unroll
            for (int r = 0; r < R; r++) val[r] = (uint32_t)(wbase + r * 64 + lane);
        } else if (FULL) {
            const uint32_t *src = vals_in + wbase + lane;
#pragma unroll
            for (int r = 0; r < R; r++) val[r] = __builtin_nontemporal_load(src + r * 64);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) { const int64_t i = wbase + r * 64 + lane; val[r] = vals_in[i < hi ? i : hi - 1]; }
        }
        if (NEXT_FULL) {                                            // the steady state: no clamping, immediate offsets
            const uint32_t *src = keys_in + wbase + TILE + lane;
#pragma unroll
            for (int r = 0; r < R; r++) nkey[r] = __builtin_nontemporal_load(src + r * 64);
        } else if (FULL && tbase + TILE < hi) {                     // a ragged tile follows (once per slice): clamped addresses
            for (int r = 0; r < R; r++) { const int64_t i = wbase + TILE + r * 64 + lane; nkey[r] = keys_in[i < hi ? i : hi - 1]; }
        }
        lds_barrier();
        // ---- chain the waves, lay the digits out in the tile ----------------------------------------------------------
        uint32_t tcnt = 0, incl = 0;
        if (tid < kBins) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { s_wbase[w][tid] = (uint16_t)run; run += s_wcnt[w][tid]; }
            tcnt = run;
            incl = wave_incl_scan(tcnt);                             // (all 64 lanes of the four digit waves are active here)
            if (lane == 63) s_scan[wave] = incl;
        }
        lds_barrier();
        if (tid < kBins) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += s_scan[w];
            const uint32_t ts = carry + incl - tcnt, gp = s_gpos[tid];
            s_delta[tid] = gp - ts;                                  // wraps; slot + delta is the output position
            s_gpos[tid] = gp + tcnt;
#pragma unroll
            for (int w = 0; w < W; w++) s_wbase[w][tid] = (uint16_t)(s_wbase[w][tid] + ts);   // (wave, digit) -> its first slot in the tile: ONE read per key when staging
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (FULL || rank[r] != 0xFFFFFFFFu) {
                const uint32_t d = ((key[r] ^ xor_mask) >> sh) & dmask;
                s_kv[(uint32_t)s_wbase[wave][d] + rank[r]] = uint2{key[r], val[r]};
            }
        }
        lds_barrier();
        // ---- write digit runs --------------------------------------------------------------------------------------------
        if (FULL) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int slot = tid + r * T;
                const uint2 kv = s_kv[slot];
                const uint32_t pos = (uint32_t)slot + s_delta[((kv.x ^ xor_mask) >> sh) & dmask];
                keys_out[pos] = kv.x; vals_out[pos] = kv.y;
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four keys at a time: the scheduler would otherwise keep all R
            }                                                           // iterations' words, positions and addresses live at once (spills)
        } else {
            const int tile_n = (int)(hi - tbase);
            for (int slot = tid; slot < tile_n; slot += T) {
                const uint2 kv = s_kv[slot];
                const uint32_t pos = (uint32_t)slot + s_delta[((kv.x ^ xor_mask) >> sh) & dmask];
                keys_out[pos] = kv.x; vals_out[pos] = kv.y;
            }
        }
        // (the next tile's first barrier comes after its ranking, which touches neither s_kv nor the digit tables)
    };
    {   // the first tile's keys
        const int64_t wb = lo + (int64_t)wave * (64 * R);
#pragma unroll
        for (int r = 0; r < R; r++) { const int64_t i = wb + r * 64 + lane; nkey[r] = keys_in[i < hi ? i : hi - 1]; }
    }
    // the steady state (a full tile followed by a full tile) is a loop of its own; the slice's last full tile and its ragged
    // tail are peeled, so that no clamped-address arithmetic and no uncountable store loop sit on the hot path
    int64_t tbase = lo;
    for (; tbase + 2 * (int64_t)TILE <= hi; tbase += TILE) tile(tbase, std::true_type{}, std::true_type{});
    if (tbase + TILE <= hi) { tile(tbase, std::true_type{}, std::false_type{}); tbase += TILE; }
    if (tbase < hi) tile(tbase, std::false_type{}, std::false_type{});
}

// Order-preserving 32-bit sort word of a 4-byte key: unsigned order of the words = the column's own order
// (u32 unsigned, i32 signed, f32 IEEE with -0.0 == +0.0 and every NaN after +inf, numpy's order).
__device__ __forceinline__ uint32_t sort_word_of(uint32_t w, int dtype)
{
    if (dtype == HARK_I32) w ^= 0x80000000u;
    else if (dtype == HARK_F32) {
        if (w == 0x80000000u) w = 0u;
        if ((w & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;
        w ^= (w & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
    }
    return w;
}

__device__ __forceinline__ uint32_t sort_word(const void *src, int dtype, int part, int64_t i)
{
    // part 0: the (only / low) 32-bit sort word, part 1: the high word of an i64.
    if (dtype == HARK_I64) {
        const uint64_t x = static_cast<const uint64_t *>(src)[i] ^ 0x8000000000000000ull;
        return part ? (uint32_t)(x >> 32) : (uint32_t)x;
    }
    return sort_word_of(static_cast<const uint32_t *>(src)[i], dtype);
}

// dst[i] = sort word of src[i]; *diff |= bits in which any word differs from the first one (a radix pass
// over a byte in which all keys agree is the identity permutation and is skipped).  4-byte keys move as
// 16-byte vectors (pool blocks and table columns are 16-byte aligned).
__global__ __launch_bounds__(256) void transform_keys_kernel(const void *__restrict__ src, int dtype, int part, uint32_t *__restrict__ dst, int64_t n,
                                                             uint32_t *__restrict__ diff)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t w0 = sort_word(src, dtype, part, 0);
    uint32_t acc = 0u;
    if (dtype != HARK_I64 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const uint4 *s4 = static_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        const int64_t nvec = n / 4;
        for (int64_t i = t0; i < nvec; i += stride) {
            uint4 q = s4[i];
            q.x = sort_word_of(q.x, dtype); q.y = sort_word_of(q.y, dtype); q.z = sort_word_of(q.z, dtype); q.w = sort_word_of(q.w, dtype);
            if (dst) d4[i] = q;                                        // dst == null: only the difference mask is wanted
            acc |= (q.x ^ w0) | (q.y ^ w0) | (q.z ^ w0) | (q.w ^ w0);
        }
        for (int64_t i = nvec * 4 + t0; i < n; i += stride) { const uint32_t w = sort_word(src, dtype, part, i); if (dst) dst[i] = w; acc |= w ^ w0; }
    } else if (dtype == HARK_I64 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        // two keys per 16-byte load, two loads in flight per lane (one 8-byte load per lane and trip ran at 4.2 TB/s)
        const uint4 *s4 = static_cast<const uint4 *>(src);
        uint2 *d2 = reinterpret_cast<uint2 *>(dst);
        const int64_t nvec = n / 2;
        const uint32_t bias = part ? 0x80000000u : 0u;
        auto two = [&](int64_t i, const uint4 q) {
            const uint32_t wa = (part ? q.y : q.x) ^ bias, wb = (part ? q.w : q.z) ^ bias;
            if (dst) d2[i] = uint2{wa, wb};
            acc |= (wa ^ w0) | (wb ^ w0);
        };
        int64_t i = t0;
        for (; i + stride < nvec; i += 2 * stride) { const uint4 qa = s4[i], qb = s4[i + stride]; two(i, qa); two(i + stride, qb); }
        for (; i < nvec; i += stride) two(i, s4[i]);
        for (int64_t r = nvec * 2 + t0; r < n; r += stride) { const uint32_t w = sort_word(src, dtype, part, r); if (dst) dst[r] = w; acc |= w ^ w0; }
    } else {
        for (int64_t i = t0; i < n; i += stride) { const uint32_t w = sort_word(src, dtype, part, i); if (dst) dst[i] = w; acc |= w ^ w0; }
    }
    if (diff) {
        // one atomic per WORKGROUP, and only while it still adds a bit: thousands of waves OR-ing into one word
        // serialise at the memory side (measured 0.19 ms for a 1e7-key column, 15x the read itself)
        __shared__ uint32_t s_acc;
        if (threadIdx.x == 0) s_acc = 0u;
        __syncthreads();
        for (int d = 32; d > 0; d >>= 1) acc |= __shfl_xor(acc, d, 64);
        if ((threadIdx.x & 63) == 0 && acc) atomicOr(&s_acc, acc);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t mine = s_acc;
            if (mine & ~__hip_atomic_load(diff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(diff, mine);
        }
    }
}

__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                                         uint32_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

__global__ __launch_bounds__(256) void gather_u64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                                         uint64_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

// ---- i64 keys: high word first, then the short runs of equal high words ------------------------------------------
// An LSD sort of 64-bit keys is eight 8-bit passes.  Keys that are spread over 64 bits (hashes, ids: BASELINE
// configs[3]) are almost ordered by their high words alone: a stable sort by the high word (four passes), then every
// run of equal high words -- a handful of keys -- is sorted by (low word, position in the run) in registers.  A run
// longer than kRunMax raises *too_long and the caller takes the eight-pass path.
constexpr int kRunMax = 16;
struct RunElem { uint64_t key; uint32_t perm, pos; };
__global__ __launch_bounds__(256) void i64_fix_runs_kernel(const uint64_t *__restrict__ col, const uint32_t *__restrict__ hi_sorted,
                                                           uint32_t *__restrict__ perm, uint64_t *__restrict__ keys_out, int64_t n, int32_t *__restrict__ too_long,
                                                           uint32_t prefix_mask /* the bits of the high word the keys were sorted by */)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto lt = [](const RunElem &a, const RunElem &b) { return a.key < b.key || (a.key == b.key && a.pos < b.pos); };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t h = hi_sorted[i] & prefix_mask;
        if (i > 0 && (hi_sorted[i - 1] & prefix_mask) == h) continue;            // not a run head
        int len = 1;
        while (len <= kRunMax && i + len < n && (hi_sorted[i + len] & prefix_mask) == h) len++;
        if (len > kRunMax) { *too_long = 1; continue; }
        if (len == 1) { keys_out[i] = col[perm[i]] ^ 0x8000000000000000ull; continue; }
        RunElem v[kRunMax];
#pragma unroll
        for (int j = 0; j < kRunMax; j++) {
            const bool in = j < len;
            const uint32_t p = in ? perm[i + j] : 0u;
            v[j].key = in ? (col[p] ^ 0x8000000000000000ull) : ~0ull;
            v[j].perm = p; v[j].pos = in ? (uint32_t)j : 0xFFFFu;
        }
        if (len <= 4) { RunElem w[4] = {v[0], v[1], v[2], v[3]}; net_sort4(w, lt); v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3]; }
        else if (len <= 8) { RunElem w[8]; for (int j = 0; j < 8; j++) w[j] = v[j]; net_sort8(w, lt); for (int j = 0; j < 8; j++) v[j] = w[j]; }
        else net_sort16(v, lt);
#pragma unroll
        for (int j = 0; j < kRunMax; j++) if (j < len) { perm[i + j] = v[j].perm; keys_out[i + j] = v[j].key; }
    }
}

// ---- i64 keys as 16-byte tuples (biased key, row id, one 4-byte column) ----------------------------------------------------
// The argsort above leaves a PERMUTATION: the sorted keys, and any column wanted in sorted order, are then s random reads
// each (12.5 M of them cost 0.24-0.28 ms: BASELINE configs[3]'s build side paid that twice).  Here the key, its row id and
// one column travel through the passes as one 16-byte word, so the run fix-up and the outputs are sequential.  The digit
// is taken from the tuple's second word (the high word of the biased key); the kernel is digit_scatter2_kernel with a
// four-word payload (see there for the ranking and the counted loads / stores).
struct GeoTuple { static constexpr int T = 256, R = 12, PER_CU = 8; };
constexpr size_t tuple_scatter_lds() { return (size_t)GeoTuple::T * GeoTuple::R * 16 + (size_t)(GeoTuple::T / 64) * kBins * (8 + 2 + 2) + kBins * 8 + 64; }

template <bool FIRST>
__global__ __launch_bounds__(GeoTuple::T) void tuple_hist_kernel(const uint64_t *__restrict__ col, const uint4 *__restrict__ tin, int64_t n, int64_t slice,
                                                                 int shift, uint32_t *__restrict__ hist, int nblk, uint64_t xorm)
{
    const int sh = shift & 255;
    const uint32_t dmask = (shift >> 8) ? (1u << (shift >> 8)) - 1u : 255u;
    __shared__ uint32_t s_hist[kBins];
    if (threadIdx.x < kBins) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice;
    const int64_t hi = lo + slice < n ? lo + slice : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t w = FIRST ? (uint32_t)((col[i] ^ xorm) >> 32) : tin[i].y;
        atomicAdd(&s_hist[(w >> sh) & dmask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kBins) hist[(size_t)threadIdx.x * nblk + blockIdx.x] = s_hist[threadIdx.x];
}

// The histogram of a pass over tuples reads 16 bytes per tuple for one digit.  The pass before it knows the digit of
// every tuple it places: it writes that digit as ONE BYTE beside the tuple (tuple_scatter_kernel<.., true>), and the
// histogram reads n bytes instead of 16 n (10^8 tuples: 0.27 ms -> 0.03 ms per pass).
__global__ __launch_bounds__(GeoTuple::T) void digit_byte_hist_kernel(const uint8_t *__restrict__ dig, int64_t n, int64_t slice, uint32_t *__restrict__ hist, int nblk)
{
    __shared__ uint32_t s_hist[kBins];
    if (threadIdx.x < kBins) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice;                 // (a multiple of the tile: 16-byte loads are aligned)
    const int64_t hi = lo + slice < n ? lo + slice : n;
    const int64_t nvec = (hi - lo) / 16;
    const uint4 *d4 = reinterpret_cast<const uint4 *>(dig + lo);
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) {
        const uint4 q = ld_nt16(d4 + i);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            atomicAdd(&s_hist[w[j] & 255u], 1u); atomicAdd(&s_hist[(w[j] >> 8) & 255u], 1u);
            atomicAdd(&s_hist[(w[j] >> 16) & 255u], 1u); atomicAdd(&s_hist[w[j] >> 24], 1u);
        }
    }
    for (int64_t i = lo + nvec * 16 + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&s_hist[dig[i]], 1u);
    __syncthreads();
    if (threadIdx.x < kBins) hist[(size_t)threadIdx.x * nblk + blockIdx.x] = s_hist[threadIdx.x];
}

// FIRST: the tuples are built from the key column (biased by 2^63), the row id and valcol (may be null: zero).
// DIG: the NEXT pass's digit of every tuple goes out as a byte beside it (dig_out[pos], shift / width in next_shift).
template <bool FIRST, bool DIG>
__global__ __launch_bounds__(GeoTuple::T) void tuple_scatter_kernel(
    const uint64_t *__restrict__ col, const uint32_t *__restrict__ valcol, const uint4 *__restrict__ tin, uint4 *__restrict__ tout,
    int64_t n, int64_t slice, int shift, const uint32_t *__restrict__ hist, int nblk, const uint32_t *__restrict__ row_total,
    uint64_t xorm /* FIRST: the keys enter as key ^ xorm (2^63: ascending signed order; its complement: descending) */,
    uint8_t *__restrict__ dig_out, int next_shift)
{
    const int sh = shift & 255;
    const uint32_t dmask = (shift >> 8) ? (1u << (shift >> 8)) - 1u : 255u;
    const int nsh = next_shift & 255;
    const uint32_t nmask = (next_shift >> 8) ? (1u << (next_shift >> 8)) - 1u : 255u;
    constexpr int T = GeoTuple::T, W = T / 64, R = GeoTuple::R, TILE = T * R;
    typedef unsigned long long u64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    uint4 *s_t = reinterpret_cast<uint4 *>(sort_lds);                                // [TILE] tuples, digit-sorted
    uint32_t *s_mask = reinterpret_cast<uint32_t *>(s_t + TILE);                     // [W][2][bins]
    uint16_t (*s_wcnt)[kBins] = reinterpret_cast<uint16_t (*)[kBins]>(s_mask + (size_t)W * 2 * kBins);
    uint16_t (*s_wbase)[kBins] = s_wcnt + W;
    uint32_t *s_delta = reinterpret_cast<uint32_t *>(s_wbase + W);                   // [bins] output position of the digit's first key of the tile - its tile slot
    uint32_t *s_gpos = s_delta + kBins;                                              // [bins] output position of the digit's next key
    uint32_t *s_scan = s_gpos + kBins;                                               // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = (int64_t)blockIdx.x * slice;
    const int64_t hi = lo + slice < n ? lo + slice : n;
    for (int i = tid; i < W * 2 * kBins; i += T) s_mask[i] = 0u;
    {
        const uint32_t tot = tid < kBins ? row_total[tid] : 0u;
        const uint32_t incl = wave_incl_scan(tot);
        if (lane == 63 && tid < kBins) s_scan[wave] = incl;
        __syncthreads();
        if (tid < kBins) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += s_scan[w];
            s_gpos[tid] = carry + incl - tot + hist[(size_t)tid * nblk + blockIdx.x];
        }
        __syncthreads();
    }
    const uint32_t lanebit = 1u << (lane & 31);
    const u64 lt = lanemask_lt();
    uint32_t *mymask = s_mask + (size_t)wave * 2 * kBins;
    uint32_t *myhalf = mymask + (lane >> 5) * kBins;
    uint16_t *mycnt = s_wcnt[wave];
    uint4 nt[R];                                                  // the next tile's tuples (FIRST: keys only, the rest is filled in after the ranking)
    auto load_next = [&](int64_t wbase, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int r = 0; r < R; r++) {
            int64_t i = wbase + r * 64 + lane;
            if (!FULL) i = i < hi ? i : hi - 1;
            if (FIRST) { const u64 k = __builtin_nontemporal_load(col + i) ^ xorm; nt[r].x = (uint32_t)k; nt[r].y = (uint32_t)(k >> 32); }
            else nt[r] = ld_nt16(tin + i);
        }
    };
    auto tile = [&](int64_t tbase, auto full_tag, auto next_tag) {
        constexpr bool FULL = decltype(full_tag)::value, NEXT_FULL = decltype(next_tag)::value;
        const int64_t wbase = tbase + (int64_t)wave * (64 * R);
        uint4 cur[R];
        uint32_t rank[R];
#pragma unroll
        for (int r = 0; r < R; r++) cur[r] = nt[r];
        reinterpret_cast<uint2 *>(mycnt)[lane] = uint2{0u, 0u};
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool valid = FULL || wbase + r * 64 + lane < hi;
            const uint32_t d = (cur[r].y >> sh) & dmask;
            rank[r] = 0xFFFFFFFFu;
            if (valid) {
                __hip_atomic_fetch_or(myhalf + d, lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint32_t plo = mymask[d], phi = mymask[kBins + d];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const u64 peers = ((u64)phi << 32) | plo;
                __hip_atomic_store(myhalf + d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const u64 below = peers & lt;
                const uint32_t before = mycnt[d];
                rank[r] = before + (uint32_t)__popcll(below);
                if (below == 0ull) mycnt[d] = (uint16_t)(before + (uint32_t)__popcll(peers));
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (FIRST) {                                                // row id and the column's value; they and the next tile's keys travel under the rest of the tile
#pragma unroll
            for (int r = 0; r < R; r++) {
                int64_t i = wbase + r * 64 + lane;
                if (!FULL) i = i < hi ? i : hi - 1;
                cur[r].z = (uint32_t)i;
                cur[r].w = valcol ? __builtin_nontemporal_load(valcol + i) : 0u;
            }
        }
        if (NEXT_FULL) load_next(wbase + TILE, std::true_type{});
        else if (FULL && tbase + TILE < hi) load_next(wbase + TILE, std::false_type{});
        lds_barrier();
        uint32_t tcnt = 0, incl = 0;
        if (tid < kBins) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { s_wbase[w][tid] = (uint16_t)run; run += s_wcnt[w][tid]; }
            tcnt = run;
            incl = wave_incl_scan(tcnt);
            if (lane == 63) s_scan[wave] = incl;
        }
        lds_barrier();
        if (tid < kBins) {
            uint32_t carry = 0;
            for (int w = 0; w < wave; w++) carry += s_scan[w];
            const uint32_t ts = carry + incl - tcnt, gp = s_gpos[tid];
            s_delta[tid] = gp - ts;
            s_gpos[tid] = gp + tcnt;
#pragma unroll
            for (int w = 0; w < W; w++) s_wbase[w][tid] = (uint16_t)(s_wbase[w][tid] + ts);
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (FULL || rank[r] != 0xFFFFFFFFu) {
                const uint32_t d = (cur[r].y >> sh) & dmask;
                s_t[(uint32_t)s_wbase[wave][d] + rank[r]] = cur[r];
            }
        }
        lds_barrier();
        if (FULL) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int slot = tid + r * T;
                const uint4 t = s_t[slot];
                const uint32_t pos = (uint32_t)slot + s_delta[(t.y >> sh) & dmask];
                tout[pos] = t;
                if (DIG) dig_out[pos] = (uint8_t)((t.y >> nsh) & nmask);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const int tile_n = (int)(hi - tbase);
            for (int slot = tid; slot < tile_n; slot += T) {
                const uint4 t = s_t[slot];
                const uint32_t pos = (uint32_t)slot + s_delta[(t.y >> sh) & dmask];
                tout[pos] = t;
                if (DIG) dig_out[pos] = (uint8_t)((t.y >> nsh) & nmask);
            }
        }
    };
    load_next(lo + (int64_t)wave * (64 * R), std::false_type{});
    int64_t tbase = lo;
    for (; tbase + 2 * (int64_t)TILE <= hi; tbase += TILE) tile(tbase, std::true_type{}, std::true_type{});
    if (tbase + TILE <= hi) { tile(tbase, std::true_type{}, std::false_type{}); tbase += TILE; }
    if (tbase < hi) tile(tbase, std::false_type{}, std::false_type{});
}

// Tuples sorted by (the prefix bits of) their high words -> sorted keys, row ids and column values.  Every tuple finds the
// run of equal prefixes it sits in (a handful of neighbours, cached lines) and its place in it: the number of tuples of
// the run that are smaller by (key, position).  No lane waits for another one's sorting network; sequential reads, writes
// within a run's span.  A run longer than kRunMax raises *too_long (the caller takes the general path).
__global__ __launch_bounds__(256) void tuple_fix_runs_kernel(const uint4 *__restrict__ tin, uint64_t *__restrict__ keys_out, uint32_t *__restrict__ perm_out,
                                                             uint32_t *__restrict__ val_out, int64_t n, int32_t *__restrict__ too_long, uint32_t prefix_mask,
                                                             uint64_t out_xor /* the keys are written as key ^ out_xor (the ORDER BY entry takes its bias off here) */)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto hi_of = [&](int64_t j) { return reinterpret_cast<const uint32_t *>(tin + j)[1] & prefix_mask; };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        // the tuple and its two neighbours' prefixes in one round trip; more than half of the tuples are alone in their run
        const uint4 t = ld_nt16(tin + i);
        const uint32_t hp = i > 0 ? hi_of(i - 1) : 0u, hn = i + 1 < n ? hi_of(i + 1) : 0u;
        const uint32_t h = t.y & prefix_mask;
        const uint64_t key = ((uint64_t)t.y << 32) | t.x;
        int64_t a = i, b = i + 1;
        if (i > 0 && hp == h) { a--; while (a > 0 && i - a <= kRunMax && hi_of(a - 1) == h) a--; }
        if (i + 1 < n && hn == h) { b++; while (b < n && b - a <= kRunMax && hi_of(b) == h) b++; }
        if (b - a > kRunMax) { *too_long = 1; continue; }
        int64_t at = a;
        for (int64_t j = a; j < b; j++) {
            if (j == i) continue;
            const uint2 q = *reinterpret_cast<const uint2 *>(tin + j);
            const uint64_t kj = ((uint64_t)q.y << 32) | q.x;
            at += (kj < key || (kj == key && j < i)) ? 1 : 0;
            if (kj == key) too_long[1] = 1;                          // equal keys exist (benign race: every writer stores 1): the join wants to know
        }
        keys_out[at] = key ^ out_xor; perm_out[at] = t.z;
        if (val_out) val_out[at] = t.w;
    }
}

__global__ __launch_bounds__(256) void gather_biased_u64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm, uint64_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[perm[i]] ^ 0x8000000000000000ull;
}

} // namespace

// Workspace bytes for sorting n pairs.
size_t k_sort_workspace_bytes(int64_t n, int num_cu)
{
    bool large; int64_t nblk, slice;
    sort_geometry(n, num_cu, &large, &nblk, &slice);
    return (size_t)kBins * ((size_t)nblk + 1) * sizeof(uint32_t);      // histograms + the 256 digit totals
}

static int64_t grid256(hark_context *ctx, int64_t n)
{
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
    return blocks < 1 ? 1 : blocks;
}

// Stable sort of n (key, val) pairs ascending by (key ^ xor_mask) as unsigned.  The input keys are `keys_first`
// (device, read only, may be a table column) or, when that is null, the contents of keys_a; keys_a, keys_b,
// vals_a, vals_b are scratch of the same size.  The payload of the input is `vals_first` (device, read
// only, may be a table column) or, when null, the input position.  `pass_mask` is the DIFFERENCE MASK of the keys (the
// bits in which any two of them differ): plan_passes turns it into at most four stable passes over digits of at most
// eight bits that cover every differing bit.  On return *keys_out / *vals_out point at the buffers (among the
// four) that hold the result.  n < 2^32.
int k_sort_pairs_u32(hark_context *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
                     const uint32_t *vals_first, int64_t n, uint32_t xor_mask, uint32_t *hist_ws, uint32_t pass_mask,
                     uint32_t **keys_out, uint32_t **vals_out, const uint32_t *keys_first = nullptr, uint32_t *hist4 = nullptr)
{
    // hist4 (optional): the per-slice histogram of the LOW byte of the INPUT keys (k_multi_hist, same geometry and xor_mask):
    // a first pass that starts at bit 0 takes its histogram from there instead of reading the keys again
    *keys_out = keys_a; *vals_out = vals_a;
    if (n <= 0) return HARK_OK;
    if (n > 0xFFFFFFFFll) return hark_fail(ctx, HARK_EARG, "sort: at most 2^32-1 rows");
    bool large; int64_t nblk, slice;
    sort_geometry(n, ctx->num_cu, &large, &nblk, &slice);
    hipStream_t st = ctx->stream;
    const uint32_t *kin = keys_first ? keys_first : keys_a, *vin = vals_first;
    uint32_t *kout = keys_first ? keys_a : keys_b, *vout = vals_b;   // a read-only input leaves both scratch buffers free
    bool first = true;
    SortPass plan[4];
    const int npass = plan_passes(pass_mask, plan);
    for (int pi = 0; pi < npass; pi++) {
        const int shift = plan[pi].shift | (plan[pi].width << 8);
        uint32_t *hist_in = hist_ws;                                  // [bins][nblk], scanned in place; the digit totals go to hist_ws + bins * nblk
        if (first && hist4 && plan[pi].shift == 0) {                  // the low byte's histogram of the input order (k_multi_hist)
            hist_in = hist4;
            if (plan[pi].width < 8) fold_hist_rows_kernel<<<dim3(1u << plan[pi].width), dim3(256), 0, st>>>(hist4, (int)nblk, plan[pi].width);
        } else
        if (large) digit_hist_kernel<GeoLarge><<<dim3((unsigned)nblk), dim3(GeoLarge::T / (nblk > ctx->num_cu ? 2 : 1)), 0, st>>>(kin, n, slice, shift, xor_mask, hist_ws, (int)nblk);
        else digit_hist_kernel<GeoSmall><<<dim3((unsigned)nblk), dim3(GeoSmall::T), 0, st>>>(kin, n, slice, shift, xor_mask, hist_ws, (int)nblk);
        scan_hist_rows_kernel<<<kBins, 256, 0, st>>>(hist_in, (int)nblk, hist_ws + (size_t)kBins * nblk);
        const uint32_t *tot = hist_ws + (size_t)kBins * nblk;
        // one launcher for every scatter kernel: KERNEL<IOTA> with `threads` threads and `lds` bytes of dynamic LDS
#define HARK_SCATTER(KERNEL_T, KERNEL_F, threads, lds)                                                                              \
        do {                                                                                                                        \
            if ((lds) > 64 * 1024) {                                                                                                \
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL_F), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds))); \
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL_T), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds))); \
            }                                                                                                                       \
            if (vin) KERNEL_F<<<dim3((unsigned)nblk), dim3(threads), (lds), st>>>(kin, vin, kout, vout, n, slice, shift, xor_mask, hist_in, (int)nblk, tot); \
            else KERNEL_T<<<dim3((unsigned)nblk), dim3(threads), (lds), st>>>(kin, vin, kout, vout, n, slice, shift, xor_mask, hist_in, (int)nblk, tot);     \
        } while (0)
        if (old_scatter()) {
            if (large) HARK_SCATTER((digit_scatter_kernel<GeoLarge, true>), (digit_scatter_kernel<GeoLarge, false>), GeoLarge::T, scatter_lds<GeoLarge>());
            else HARK_SCATTER((digit_scatter_kernel<GeoSmall, true>), (digit_scatter_kernel<GeoSmall, false>), GeoSmall::T, scatter_lds<GeoSmall>());
        } else if (large) HARK_SCATTER((digit_scatter2_kernel<GeoLarge, true>), (digit_scatter2_kernel<GeoLarge, false>), GeoLarge::T, scatter2_lds<GeoLarge>());
        else HARK_SCATTER((digit_scatter2_kernel<GeoSmall, true>), (digit_scatter2_kernel<GeoSmall, false>), GeoSmall::T, scatter2_lds<GeoSmall>());
#undef HARK_SCATTER
        HIP_TRY(ctx, hipGetLastError());
        *keys_out = kout; *vals_out = vout;
        kin = kout; vin = vout;
        kout = (kout == keys_b) ? keys_a : keys_b;
        vout = (vout == vals_b) ? vals_a : vals_b;
        first = false;
    }
    if (first) {                                                   // no pass ran: the order is the input order
        if (keys_first) HIP_TRY(ctx, hipMemcpyAsync(keys_a, keys_first, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        if (vals_first) HIP_TRY(ctx, hipMemcpyAsync(vals_a, vals_first, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        else { iota_u32_kernel<<<dim3((unsigned)grid256(ctx, n)), dim3(256), 0, st>>>(vals_a, n); HIP_TRY(ctx, hipGetLastError()); }
    }
    return HARK_OK;
}

// dst = sort words of a column; *diff_host (optional) receives the bits in which the words differ.
int k_transform_keys(hark_context *ctx, const void *src, int dtype, int part, uint32_t *dst, int64_t n, uint32_t *diff_host)
{
    if (diff_host) *diff_host = 0u;
    if (n <= 0) return HARK_OK;
    uint32_t *diff = nullptr;
    if (diff_host) {
        HARK_TRY(hark_alloc(ctx, (void **)&diff, 16));
        if (hipMemsetAsync(diff, 0, 16, ctx->stream) != hipSuccess) { hark_free(ctx, diff); return hark_fail(ctx, HARK_EHIP, "sort: memset failed"); }
    }
    transform_keys_kernel<<<dim3((unsigned)grid256(ctx, n)), dim3(256), 0, ctx->stream>>>(src, dtype, part, dst, n, diff);
    int rc = hipGetLastError() == hipSuccess ? HARK_OK : hark_fail(ctx, HARK_EHIP, "sort: transform failed");
    if (!rc && diff_host) { int64_t w = 0; rc = hark_read_words(ctx, diff, &w, 1); *diff_host = (uint32_t)w; }
    if (diff) hark_free(ctx, diff);
    return rc;
}

// hist4 (pool block, caller frees) = per-slice histograms of all four digits of `keys` ^ xor_mask; *diff_host = the bits in
// which the keys differ.  One read of the keys, one host synchronisation.
static int k_multi_hist(hark_context *ctx, const uint32_t *keys, int64_t n, uint32_t xor_mask, uint32_t **hist4_out, uint32_t *diff_host, bool *sorted = nullptr)
{
    *hist4_out = nullptr; *diff_host = 0u;
    if (sorted) *sorted = false;
    bool large; int64_t nblk, slice;
    sort_geometry(n, ctx->num_cu, &large, &nblk, &slice);
    uint32_t *h4 = nullptr, *diff = nullptr;
    HARK_TRY(hark_alloc(ctx, (void **)&h4, (size_t)kBins * nblk * sizeof(uint32_t)));
    int rc = hark_alloc(ctx, (void **)&diff, 16);
    if (!rc && hipMemsetAsync(diff, 0, 16, ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: memset failed");
    if (!rc) {
        multi_hist_kernel<<<dim3((unsigned)nblk), dim3(large ? 1024 : 256), 0, ctx->stream>>>(keys, n, slice, xor_mask, h4, (int)nblk, diff);
        if (hipGetLastError() != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: histogram launch failed");
    }
    int64_t w = 0;
    if (!rc) rc = hark_read_words(ctx, diff, &w, 1);
    hark_free(ctx, diff);
    if (rc) { hark_free(ctx, h4); return rc; }
    *hist4_out = h4; *diff_host = (uint32_t)w;
    if (sorted) *sorted = (uint32_t)((uint64_t)w >> 32) == 0u && !getenv("HARK_SORT_NO_PRESORTED");
    return HARK_OK;
}

static uint32_t passes_of(uint32_t diff)
{
    uint32_t m = 0;
    for (int b = 0; b < 4; b++) if ((diff >> (8 * b)) & 0xFFu) m |= 1u << b;
    return m;
}

int k_gather(hark_context *ctx, const void *src, int esz, const uint32_t *idx, void *dst, int64_t n)
{
    if (n <= 0) return HARK_OK;
    const int64_t blocks = grid256(ctx, n);
    if (esz == 4) gather_u32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(static_cast<const uint32_t *>(src), idx, static_cast<uint32_t *>(dst), n);
    else gather_u64_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(static_cast<const uint64_t *>(src), idx, static_cast<uint64_t *>(dst), n);
    HIP_TRY(ctx, hipGetLastError());
    return HARK_OK;
}

// Stable sort of a column of any supported dtype together with a 32-bit payload.  payload == nullptr: the
// payload is the row id (an argsort); otherwise a device column of n 4-byte values (only for 4-byte keys), which
// travels with the keys so that no gather is needed afterwards.  On return *vals_out (pool block, n x u32, caller
// frees) holds the payload in sorted order; if words_out is non-null it receives the sorted keys as 32-bit words:
// the column VALUES themselves for u32 / i32 keys, the order-preserving sort words for f32 keys (equality tests
// only); not available for i64.  Radix passes over key bytes in which all keys agree are skipped.
static int k_sort_column_lsd(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending, const uint32_t *payload,
                             uint32_t **vals_out, uint32_t **words_out)
{
    *vals_out = nullptr;
    if (words_out) *words_out = nullptr;
    if (n <= 0) return HARK_OK;
    if (payload && dtype == HARK_I64) return hark_fail(ctx, HARK_EARG, "sort: payload-carrying sort needs a 4-byte key");
    uint32_t *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr, *ws = nullptr;
    const size_t b = (size_t)n * 4;
    int rc = hark_alloc(ctx, (void **)&k0, b);
    if (!rc) rc = hark_alloc(ctx, (void **)&k1, b);
    if (!rc) rc = hark_alloc(ctx, (void **)&v0, b);
    if (!rc) rc = hark_alloc(ctx, (void **)&v1, b);
    if (!rc) rc = hark_alloc(ctx, (void **)&ws, k_sort_workspace_bytes(n, ctx->num_cu));
    const uint32_t xm = descending ? 0xFFFFFFFFu : 0u;
    uint32_t diff = 0u, *ko = k0, *vo = v0;
    if (dtype == HARK_U32 || dtype == HARK_I32) {
        // integer keys are sorted as they are: the sign flip of i32 is part of the digit mask, the first pass reads the
        // column itself, and the sorted "words" are the sorted column values (no transform pass, no inverse)
        const uint32_t xk = xm ^ (dtype == HARK_I32 ? 0x80000000u : 0u);
        uint32_t *h4 = nullptr;
        bool sorted = false;
        if (!rc) rc = k_multi_hist(ctx, static_cast<const uint32_t *>(col), n, xk, &h4, &diff, &sorted);   // difference mask + every digit's histograms: one read
        if (!rc && sorted) {
            // the column is in order already: the stable sort is the identity (keys and payload as they are, row ids 0 .. n - 1)
            HIP_TRY_RC(ctx, rc, hipMemcpyAsync(k0, col, b, hipMemcpyDeviceToDevice, ctx->stream));
            if (payload) HIP_TRY_RC(ctx, rc, hipMemcpyAsync(v0, payload, b, hipMemcpyDeviceToDevice, ctx->stream));
            else HARK_LAUNCH_RC(ctx, rc, iota_u32_kernel<<<dim3((unsigned)grid256(ctx, n)), dim3(256), 0, ctx->stream>>>(v0, n));
            ko = k0; vo = v0;
        } else
        if (!rc) rc = k_sort_pairs_u32(ctx, k0, k1, v0, v1, payload, n, xk, ws, diff, &ko, &vo, static_cast<const uint32_t *>(col), h4);
        hark_free(ctx, h4);                                          // (stream-ordered reuse: the passes are enqueued)
    } else {
        if (!rc) rc = k_transform_keys(ctx, col, dtype, 0, k0, n, &diff);
        if (!rc) rc = k_sort_pairs_u32(ctx, k0, k1, v0, v1, payload, n, xm, ws, diff, &ko, &vo);
    }
    if (!rc && dtype == HARK_I64) {
        // LSD over 64 bits: after the low word, sort (stably) by the high word gathered through the current
        // permutation; the permutation travels as the payload.
        uint32_t *kf = (ko == k0) ? k1 : k0, *vf = (vo == v0) ? v1 : v0;           // the free buffer of each pair
        rc = k_transform_keys(ctx, col, dtype, 1, nullptr, n, &diff);              // do the high words differ at all?
        if (!rc && passes_of(diff) != 0u) {                                        // (counts and sums below 2^32 do not: done)
            rc = k_transform_keys(ctx, col, dtype, 1, ko, n, nullptr);             // ko's sorted low words are no longer needed
            if (!rc) rc = k_gather(ctx, ko, 4, vo, kf, n);
            uint32_t *ko2 = kf, *vo2 = vo;
            // buffers: keys in kf (scratch ko), payload = vo (read only in the first pass, then ping-pong vf <-> vo)
            if (!rc) rc = k_sort_pairs_u32(ctx, kf, ko, vo, vf, vo, n, xm, ws, diff, &ko2, &vo2);
            ko = ko2; vo = vo2;
        }
    }
    if (rc == HARK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort kernels failed");
    uint32_t *bufs[4] = {k0, k1, v0, v1};
    for (uint32_t *q : bufs) {
        if (!q) continue;
        if (!rc && q == vo) continue;
        if (!rc && q == ko && words_out && dtype != HARK_I64) continue;
        hark_free(ctx, q);
    }
    hark_free(ctx, ws);
    if (rc) return rc;
    *vals_out = vo;
    if (words_out && dtype != HARK_I64) *words_out = ko;
    return HARK_OK;
}

int k_argsort_i64_keys(hark_context *ctx, const void *col, int64_t n, uint32_t **perm_out, uint64_t **keys_out, const uint32_t *valcol, uint32_t **val_out, int *unique_out, bool *plain_out = nullptr, int8_t *msd_unfit = nullptr);

// Stable sort of a column with a 32-bit payload (see k_sort_column_lsd).  An ascending argsort of an i64 column takes
// the high-word-first path of k_argsort_i64_keys (four passes + a run fix-up instead of eight passes).
int k_sort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending, const uint32_t *payload,
                  uint32_t **vals_out, uint32_t **words_out)
{
    if (dtype == HARK_I64 && !descending && !payload && n >= 4096) {
        if (words_out) *words_out = nullptr;
        uint64_t *keys = nullptr;
        const int rc = k_argsort_i64_keys(ctx, col, n, vals_out, &keys, nullptr, nullptr, nullptr);
        hark_free(ctx, keys);
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) return hark_fail(ctx, HARK_EHIP, "sort kernels failed");
        return rc;
    }
    return k_sort_column_lsd(ctx, col, dtype, n, descending, payload, vals_out, words_out);
}

int k_argsort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending,
                     uint32_t **perm_out, uint32_t **sorted_words_out)
{
    return k_sort_column(ctx, col, dtype, n, descending, nullptr, perm_out, sorted_words_out);
}

// k_msort.hip: three sweeps of 16-byte tuples, most significant digit first, for large tables of well-spread keys (same outputs
// as sort_i64_tuples; *done = false when it does not apply or gave up)
int k_sort_i64_msd(hark_context *ctx, const void *col, int64_t n, const uint32_t *valcol, uint64_t *keys, uint32_t **perm_out, uint32_t **val_out,
                   bool *done, int *unique_out, uint64_t xorm, uint64_t out_xor, int8_t *unfit);

// Stable ascending argsort of an i64 column together with the SORTED keys (biased by 2^63: unsigned order = signed order):
// *perm_out (n x u32) and *keys_out (n x u64) are pool blocks the caller frees.  High word first + run fix-up (above)
// when the high words differ, the plain low-word sort when they do not, the eight-pass path as the fallback.
// valcol / val_out (optional): a 4-byte column of the same table; *val_out then holds it in sorted order (valcol[perm[i]]).
// When the high words differ, key, row id and the column's value travel through the passes as 16-byte tuples
// (sort_i64_tuples): no random read anywhere.  On the other paths *val_out is a gather through the permutation.
static int sort_i64_tuples(hark_context *ctx, const void *col, int64_t n, uint32_t diff_hi, const uint32_t *valcol,
                           uint64_t *keys, uint32_t **perm_out, uint32_t **val_out, bool *done, int *unique_out,
                           uint64_t xorm = 0x8000000000000000ull /* keys_out holds key ^ xorm, ascending */, uint64_t out_xor = 0 /* ... ^ out_xor */)
{
    *done = false;
    if (n > 0xFFFFFFFFll) return HARK_OK;                          // 32-bit positions and row ids (the general path reports the limit)
    hipStream_t st = ctx->stream;
    // Up to 2^24 keys are only sorted by the top 24 bits of their high words (three passes instead of four): keys spread
    // over 64 bits then share a prefix with 0.75 others on average, and the run fix-up orders whole keys anyway.
    const uint32_t prefix_mask = n <= ((int64_t)1 << 24) && (diff_hi & 0xFFFFFF00u) ? 0xFFFFFF00u : 0xFFFFFFFFu;
    SortPass plan[4];
    const int np = plan_passes(diff_hi & prefix_mask, plan);
    if (np == 0) return HARK_OK;
    constexpr int64_t tile = (int64_t)GeoTuple::T * GeoTuple::R;
    int64_t nblk = (n + tile - 1) / tile;
    if (nblk > (int64_t)ctx->num_cu * GeoTuple::PER_CU) nblk = (int64_t)ctx->num_cu * GeoTuple::PER_CU;
    int64_t slice = ((n + nblk - 1) / nblk + tile - 1) / tile * tile;
    nblk = (n + slice - 1) / slice;
    uint4 *ta = nullptr, *tb = nullptr;
    uint32_t *ws = nullptr, *perm = nullptr, *val = nullptr;
    int32_t *flag = nullptr;
    int rc = hark_alloc(ctx, (void **)&ta, (size_t)n * 16);
    if (!rc) rc = hark_alloc(ctx, (void **)&tb, (size_t)n * 16);
    if (!rc) rc = hark_alloc(ctx, (void **)&ws, (size_t)kBins * ((size_t)nblk + 1) * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&flag, 16);
    if (!rc) rc = hark_alloc(ctx, (void **)&perm, (size_t)n * 4);
    if (!rc && valcol && val_out) rc = hark_alloc(ctx, (void **)&val, (size_t)n * 4);
    uint8_t *dig[2] = {nullptr, nullptr};                          // the next pass's digits, one byte per tuple (two buffers in turn)
    if (!rc && np > 1 && !getenv("HARK_SORT_NO_DIGIT_BYTES")) {
        rc = hark_alloc(ctx, (void **)&dig[0], (size_t)n + 16);
        if (!rc && np > 2) rc = hark_alloc(ctx, (void **)&dig[1], (size_t)n + 16);
        if (!rc && np <= 2) dig[1] = nullptr;
    }
    auto cleanup = [&](bool keep) {
        hark_free(ctx, ta); hark_free(ctx, tb); hark_free(ctx, ws); hark_free(ctx, flag); hark_free(ctx, dig[0]); hark_free(ctx, dig[1]);
        if (!keep) { hark_free(ctx, perm); hark_free(ctx, val); }
    };
    if (rc == HARK_ENOMEM) { cleanup(false); ctx->err.clear(); return HARK_OK; }      // the permutation path needs half the room
    if (rc) { cleanup(false); return rc; }
    const size_t lds = tuple_scatter_lds();
    hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void *>(&tuple_scatter_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (he == hipSuccess) he = hipFuncSetAttribute(reinterpret_cast<const void *>(&tuple_scatter_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (he == hipSuccess) he = hipFuncSetAttribute(reinterpret_cast<const void *>(&tuple_scatter_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (he == hipSuccess) he = hipFuncSetAttribute(reinterpret_cast<const void *>(&tuple_scatter_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t *row_total = ws + (size_t)kBins * nblk;
    const uint64_t *c64 = static_cast<const uint64_t *>(col);
    const uint4 *tin = nullptr;
    uint4 *tout = ta;
    for (int pi = 0; pi < np && he == hipSuccess; pi++) {
        const int shift = plan[pi].shift | (plan[pi].width << 8);
        const dim3 grid((unsigned)nblk), block(GeoTuple::T);
        // this pass's histogram: from the key column (first pass), or from the digit bytes the pass before wrote
        if (pi == 0) tuple_hist_kernel<true><<<grid, block, 0, st>>>(c64, nullptr, n, slice, shift, ws, (int)nblk, xorm);
        else if (dig[(pi - 1) & 1]) digit_byte_hist_kernel<<<grid, block, 0, st>>>(dig[(pi - 1) & 1], n, slice, ws, (int)nblk);
        else tuple_hist_kernel<false><<<grid, block, 0, st>>>(nullptr, tin, n, slice, shift, ws, (int)nblk, xorm);
        scan_hist_rows_kernel<<<dim3(kBins), dim3(256), 0, st>>>(ws, (int)nblk, row_total);
        const bool more = pi + 1 < np && dig[pi & 1] != nullptr;
        const int next_shift = pi + 1 < np ? (plan[pi + 1].shift | (plan[pi + 1].width << 8)) : 0;
        uint8_t *dout = more ? dig[pi & 1] : nullptr;
        if (pi == 0 && more) tuple_scatter_kernel<true, true><<<grid, block, lds, st>>>(c64, val ? valcol : nullptr, nullptr, tout, n, slice, shift, ws, (int)nblk, row_total, xorm, dout, next_shift);
        else if (pi == 0) tuple_scatter_kernel<true, false><<<grid, block, lds, st>>>(c64, val ? valcol : nullptr, nullptr, tout, n, slice, shift, ws, (int)nblk, row_total, xorm, nullptr, 0);
        else if (more) tuple_scatter_kernel<false, true><<<grid, block, lds, st>>>(nullptr, nullptr, tin, tout, n, slice, shift, ws, (int)nblk, row_total, xorm, dout, next_shift);
        else tuple_scatter_kernel<false, false><<<grid, block, lds, st>>>(nullptr, nullptr, tin, tout, n, slice, shift, ws, (int)nblk, row_total, xorm, nullptr, 0);
        he = hipGetLastError();
        tin = tout; tout = tout == ta ? tb : ta;
    }
    int64_t general = 0;
    if (he == hipSuccess) he = hipMemsetAsync(flag, 0, 16, st);
    if (he == hipSuccess) {
        tuple_fix_runs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(tin, keys, perm, val, n, flag, prefix_mask, out_xor);
        he = hipGetLastError();
    }
    if (he != hipSuccess) { cleanup(false); return hark_fail(ctx, HARK_EHIP, "sort: tuple pass failed: %s", hipGetErrorString(he)); }
    rc = hark_read_words(ctx, flag, &general, 1);
    if (rc) { cleanup(false); return rc; }
    if ((general & 0xFFFFFFFFll) != 0) { cleanup(false); return HARK_OK; }          // a long run of equal prefixes: the general path
    if (unique_out) *unique_out = ((general >> 32) & 0xFFFFFFFFll) ? 0 : 1;         // (the fix-up compared every pair of keys that could be equal)
    cleanup(true);
    *perm_out = perm;
    if (val_out) *val_out = val;
    *done = true;
    return HARK_OK;
}

// Stable DESCENDING argsort of an i64 column by the tuple passes only: *keys_out holds ~(key ^ 2^63) ascending (the caller
// undoes it), *done = false when the tuple path does not apply (equal high words, a long run of equal prefixes, no room) --
// nothing is returned then and the caller takes its general path.
int k_argsort_i64_desc_tuples(hark_context *ctx, const void *col, int64_t n, uint32_t **perm_out, uint64_t **keys_out, const uint32_t *valcol, uint32_t **val_out, bool *done,
                              uint64_t out_xor /* *keys_out is written ^ out_xor: 0x7FFF... gives the plain keys back */, int8_t *msd_unfit /* optional: hark_column::msd_unfit of a table column */)
{
    *perm_out = nullptr; *keys_out = nullptr; *done = false;
    if (val_out) *val_out = nullptr;
    if (n <= 0) return HARK_OK;
    uint64_t *keys = nullptr;                                       // ONE block for whichever path delivers (the three sweeps first, the tuple passes after them)
    if (!getenv("HARK_SORT_NO_TUPLES")) {
        HARK_TRY(hark_alloc(ctx, (void **)&keys, (size_t)n * 8));
        const int rc0 = k_sort_i64_msd(ctx, col, n, valcol, keys, perm_out, val_out, done, nullptr, 0x7FFFFFFFFFFFFFFFull, out_xor, msd_unfit);
        if (rc0 || *done) { if (rc0) hark_free(ctx, keys); else *keys_out = keys; return rc0; }
    }
    uint32_t diff_hi = 0u;
    { const int rcd = k_transform_keys(ctx, col, HARK_I64, 1, nullptr, n, &diff_hi); if (rcd) { hark_free(ctx, keys); return rcd; } }   // (the bits in which the high words differ: the same either way)
    if (passes_of(diff_hi) == 0u || getenv("HARK_SORT_NO_TUPLES")) { hark_free(ctx, keys); return HARK_OK; }
    if (!keys) HARK_TRY(hark_alloc(ctx, (void **)&keys, (size_t)n * 8));
    const int rc = sort_i64_tuples(ctx, col, n, diff_hi, valcol, keys, perm_out, val_out, done, nullptr, 0x7FFFFFFFFFFFFFFFull, out_xor);
    if (rc || !*done) { hark_free(ctx, keys); return rc; }
    *keys_out = keys;
    return HARK_OK;
}

int k_argsort_i64_keys(hark_context *ctx, const void *col, int64_t n, uint32_t **perm_out, uint64_t **keys_out, const uint32_t *valcol, uint32_t **val_out,
                       int *unique_out /* optional: 1 all keys distinct, 0 equal keys exist, -1 not determined (the permutation paths) */,
                       bool *plain_out /* optional: set when *keys_out holds the PLAIN keys (the tuple path took the bias off at its last write), else biased */,
                       int8_t *msd_unfit /* optional: hark_column::msd_unfit of a table column */)
{
    if (plain_out) *plain_out = false;
    *perm_out = nullptr; *keys_out = nullptr;
    if (val_out) *val_out = nullptr;
    if (unique_out) *unique_out = -1;
    if (n <= 0) return HARK_OK;
    hipStream_t st = ctx->stream;
    uint64_t *keys = nullptr;
    int rc = hark_alloc(ctx, (void **)&keys, (size_t)n * 8);
    if (rc) return rc;
    bool done = false;
    if (!getenv("HARK_SORT_I64_LSD") && !getenv("HARK_SORT_NO_TUPLES")) {
        rc = k_sort_i64_msd(ctx, col, n, valcol, keys, perm_out, val_out, &done, unique_out, 0x8000000000000000ull, plain_out ? 0x8000000000000000ull : 0ull, msd_unfit);
        if (rc) { hark_free(ctx, keys); return rc; }
        if (done) { if (plain_out) *plain_out = true; *keys_out = keys; return HARK_OK; }
    }
    uint32_t diff_hi = 0u;
    rc = k_transform_keys(ctx, col, HARK_I64, 1, nullptr, n, &diff_hi);
    if (!rc && passes_of(diff_hi) != 0u && !getenv("HARK_SORT_I64_LSD") && !getenv("HARK_SORT_NO_TUPLES"))
        rc = sort_i64_tuples(ctx, col, n, diff_hi, valcol, keys, perm_out, val_out, &done, unique_out, 0x8000000000000000ull, plain_out ? 0x8000000000000000ull : 0ull);
    if (!rc && done && plain_out) *plain_out = true;
    if (!rc && !done && passes_of(diff_hi) != 0u && !getenv("HARK_SORT_I64_LSD")) {
        uint32_t *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr, *ws = nullptr; int32_t *flag = nullptr;
        const size_t b = (size_t)n * 4;
        rc = hark_alloc(ctx, (void **)&k0, b);
        if (!rc) rc = hark_alloc(ctx, (void **)&k1, b);
        if (!rc) rc = hark_alloc(ctx, (void **)&v0, b);
        if (!rc) rc = hark_alloc(ctx, (void **)&v1, b);
        if (!rc) rc = hark_alloc(ctx, (void **)&ws, k_sort_workspace_bytes(n, ctx->num_cu));
        if (!rc) rc = hark_alloc(ctx, (void **)&flag, 16);
        uint32_t *ko = k0, *vo = v0;
        if (!rc) rc = k_transform_keys(ctx, col, HARK_I64, 1, k0, n, nullptr);
        // Up to 2^24 keys are only sorted by the top 24 bits of their high words (three passes instead of four): keys spread
        // over 64 bits then share a prefix with 0.75 others on average, and the run fix-up below orders whole keys anyway.
        const uint32_t prefix_mask = n <= ((int64_t)1 << 24) && (diff_hi & 0xFFFFFF00u) ? 0xFFFFFF00u : 0xFFFFFFFFu;
        if (!rc) rc = k_sort_pairs_u32(ctx, k0, k1, v0, v1, nullptr, n, 0u, ws, diff_hi & prefix_mask, &ko, &vo);
        int64_t general = 0;
        if (!rc) {
            HIP_TRY_RC(ctx, rc, hipMemsetAsync(flag, 0, 16, st));
            i64_fix_runs_kernel<<<dim3((unsigned)grid256(ctx, n)), dim3(256), 0, st>>>(static_cast<const uint64_t *>(col), ko, vo, keys, n, flag, prefix_mask);
            if (hipGetLastError() != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: run fix-up launch failed");
            if (!rc) rc = hark_read_words(ctx, flag, &general, 1);
        }
        if (!rc && (general & 0xFFFFFFFFll) == 0) { *perm_out = vo; done = true; }
        uint32_t *bufs[4] = {k0, k1, v0, v1};
        for (uint32_t *q : bufs) if (q && !(done && q == vo)) hark_free(ctx, q);
        hark_free(ctx, ws); hark_free(ctx, flag);
    }
    if (!rc && !done) {                                              // equal high words, or a long run of them: the general path
        uint32_t *perm = nullptr;
        rc = k_sort_column_lsd(ctx, col, HARK_I64, n, false, nullptr, &perm, nullptr);
        if (!rc) {
            gather_biased_u64_kernel<<<dim3((unsigned)grid256(ctx, n)), dim3(256), 0, st>>>(static_cast<const uint64_t *>(col), perm, keys, n);
            if (hipGetLastError() != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "sort: gather launch failed");
        }
        if (rc) hark_free(ctx, perm); else *perm_out = perm;
    }
    if (!rc && valcol && val_out && !*val_out) {                     // the permutation paths: one gather
        rc = hark_alloc(ctx, (void **)val_out, (size_t)n * 4);
        if (!rc) rc = k_gather(ctx, valcol, 4, *perm_out, *val_out, n);
    }
    if (rc) {
        hark_free(ctx, keys);
        if (*perm_out) { hark_free(ctx, *perm_out); *perm_out = nullptr; }
        if (val_out && *val_out) { hark_free(ctx, *val_out); *val_out = nullptr; }
        return rc;
    }
    *keys_out = keys;
    return HARK_OK;
}



// ---------------------------------------------------------------------------
// Hash partitioning of rows for the multi-GPU repartition (all-to-all) and gathers
// ---------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void hash_dest_kernel(const void *__restrict__ col, int dtype, int64_t n, uint32_t nparts, uint32_t *__restrict__ dest)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h;
        // f32 keys hash by their sort word: -0.0 and +0.0 (one group) and all NaNs must reach the same owner rank
        if (dtype != HARK_I64) h = mix32(dtype == HARK_F32 ? sort_word_of(static_cast<const uint32_t *>(col)[i], HARK_F32) : static_cast<const uint32_t *>(col)[i]);
        else { const uint64_t x = static_cast<const uint64_t *>(col)[i]; h = mix32((uint32_t)x ^ mix32((uint32_t)(x >> 32))); }
        dest[i] = (uint32_t)(((uint64_t)h * nparts) >> 32);          // uniform in [0, nparts)
    }
}

// Order-preserving 64-bit sort word of a key (the sort's own words, so range parts and the local sort agree).
__device__ __forceinline__ uint64_t order_word(const void *col, int dtype, int64_t i)
{
    if (dtype == HARK_I64) return static_cast<const uint64_t *>(col)[i] ^ 0x8000000000000000ull;
    return sort_word_of(static_cast<const uint32_t *>(col)[i], dtype);
}

// dest = number of splitters <= key (ascending) -- equal keys share a part; mirrored for descending.
__global__ __launch_bounds__(256) void range_dest_kernel(const void *__restrict__ col, int dtype, int64_t n, const void *__restrict__ splitters,
                                                         int nsplit, int descending, uint32_t *__restrict__ dest)
{
    __shared__ uint64_t s_split[256];
    if ((int)threadIdx.x < nsplit) s_split[threadIdx.x] = order_word(splitters, dtype, threadIdx.x);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t w = order_word(col, dtype, i);
        int lo = 0, hi = nsplit;                                     // first splitter > w
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_split[mid] <= w) lo = mid + 1; else hi = mid; }
        dest[i] = (uint32_t)(descending ? nsplit - lo : lo);
    }
}

// One stable 8-bit pass on the part ids in `dest` -> perm_out grouped by part, part sizes to the host.
int partition_by_dest(hark_context *ctx, uint32_t *dest, int64_t n, int nparts, uint32_t *perm_out, int64_t *counts_host, const char *who)
{
    uint32_t *dtmp = nullptr, *vtmp = nullptr, *ws = nullptr;
    int rc = hark_alloc(ctx, (void **)&dtmp, (size_t)n * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&vtmp, (size_t)n * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&ws, k_sort_workspace_bytes(n, ctx->num_cu));
    // the permutation lands in perm_out (the "tmp" side of a one-pass sort)
    uint32_t *ko = nullptr, *vo = nullptr;
    if (!rc) rc = k_sort_pairs_u32(ctx, dest, dtmp, vtmp, perm_out, nullptr, n, 0u, ws, 0xFFu, &ko, &vo);   // one 8-bit pass over the part ids: the payload lands in vals_b = perm_out
    if (!rc) {
        // the digit totals of the pass are the part sizes (ws: 256*nblk histogram, then 256 totals)
        bool large; int64_t nblk, slice;
        sort_geometry(n, ctx->num_cu, &large, &nblk, &slice);
        std::vector<uint32_t> tot(256);
        if (hipMemcpyAsync(tot.data(), ws + (size_t)kBins * nblk, 256 * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, who);
        else for (int i = 0; i < nparts; i++) counts_host[i] = tot[i];
    }
    hark_free(ctx, dtmp); hark_free(ctx, vtmp); hark_free(ctx, ws);
    return rc;
}

} // namespace

extern "C" {

// Rows of a key column -> destination part = hash(key) * nparts >> 32 (equal keys, equal
// part, on every rank).  perm_out (device, n x u32) lists the row ids grouped by part,
// rows of one part in table order; counts_host[nparts] receives the part sizes.
int hark_op_partition_by_hash(hark_context *ctx, const void *key_col, int32_t dtype, int64_t n, int32_t nparts,
                              uint32_t *perm_out, int64_t *counts_host)
{
    hark_device_guard guard__(ctx);
    if (!ctx || n < 0 || nparts < 1 || nparts > 256 || !counts_host || (n && (!key_col || !perm_out))) return HARK_EARG;
    for (int i = 0; i < nparts; i++) counts_host[i] = 0;
    if (n == 0) return HARK_OK;
    if (n > 0xFFFFFFFFll) return hark_fail(ctx, HARK_EARG, "partition_by_hash: at most 2^32-1 rows");
    uint32_t *dest = nullptr;
    int rc = hark_alloc(ctx, (void **)&dest, (size_t)n * 4);
    if (!rc) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
        hash_dest_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(key_col, (int)dtype, n, (uint32_t)nparts, dest);
        rc = partition_by_dest(ctx, dest, n, nparts, perm_out, counts_host, "partition_by_hash: reading part sizes failed");
    }
    hark_free(ctx, dest);
    return rc;
}

// Range partition for the distributed ORDER BY (sample sort): part = number of splitters <= key
// in the column's own order (u32 unsigned, i32/i64 signed, f32 IEEE with -0 == +0), so equal keys
// share a part and parts are ordered; `descending` mirrors the part ids.  `splitters_host` holds
// nparts-1 ascending values of the column's dtype.  Same outputs as hark_op_partition_by_hash.
int hark_op_partition_by_range(hark_context *ctx, const void *key_col, int32_t dtype, int64_t n, int32_t nparts,
                               const void *splitters_host, int32_t descending, uint32_t *perm_out, int64_t *counts_host)
{
    hark_device_guard guard__(ctx);
    if (!ctx || n < 0 || nparts < 1 || nparts > 256 || !counts_host || (nparts > 1 && !splitters_host) ||
        (n && (!key_col || !perm_out)) || hark_dtype_size(dtype) == 0) return HARK_EARG;
    for (int i = 0; i < nparts; i++) counts_host[i] = 0;
    if (n == 0) return HARK_OK;
    if (n > 0xFFFFFFFFll) return hark_fail(ctx, HARK_EARG, "partition_by_range: at most 2^32-1 rows");
    uint32_t *dest = nullptr;
    void *split_dev = nullptr;
    const size_t sb = (size_t)(nparts - 1) * hark_dtype_size(dtype);
    int rc = hark_alloc(ctx, (void **)&dest, (size_t)n * 4);
    if (!rc && sb) rc = hark_alloc(ctx, &split_dev, sb);
    if (!rc && sb && hipMemcpyAsync(split_dev, splitters_host, sb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = hark_fail(ctx, HARK_EHIP, "partition_by_range: splitter upload failed");
    if (!rc) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
        range_dest_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(key_col, dtype, n, split_dev, nparts - 1, descending ? 1 : 0, dest);
        rc = partition_by_dest(ctx, dest, n, nparts, perm_out, counts_host, "partition_by_range: reading part sizes failed");
    }
    hark_free(ctx, dest); if (split_dev) hark_free(ctx, split_dev);
    return rc;
}

// dst[i] = src[idx[i]] for 4- or 8-byte elements (device pointers).
int hark_op_gather(hark_context *ctx, const void *src, int32_t dtype, const uint32_t *idx, void *dst, int64_t n)
{
    hark_device_guard guard__(ctx);
    if (!ctx || n < 0 || (n && (!src || !idx || !dst))) return HARK_EARG;
    return k_gather(ctx, src, (int)hark_dtype_size(dtype), idx, dst, n);
}

} // extern "C"
