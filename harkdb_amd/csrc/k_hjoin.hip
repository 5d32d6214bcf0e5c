// k_hjoin.hip -- partitioned join with the build side staged in LDS, bucket by bucket.
//
// Replaces the 32-pass sort of (n + s) tagged triples and the sequential per-key loop of
// futhark/join.fut:58-68 for large probe sides (the sort-merge path of k_join.hip stays for small
// inputs and as the fallback).  Output order is the reference's: ascending key, then left row id,
// then right row id (join.fut:55-75).
//
//   1. the BUILD side (db2, s rows) is argsorted by key (stable: row ids ascend inside a key);
//   2. P - 1 splitters are read off the sorted build keys at equal distances, so every bucket owns
//      a contiguous slice of ~s / P sorted build entries WHATEVER the key distribution is, and
//      buckets are ordered by key (a hash would balance as well but lose the reference's order);
//      when a sample of the probe keys shows them crowding a stretch of the build side, the cut is
//      made by the sampled rows' weight instead (jhot_select_kernel);
//   3. jpart_kernel streams the PROBE keys once (non-temporal, 16 B per lane), drops keys outside
//      [min, max] of the build side, finds each row's bucket (an interpolation guess checked against
//      four splitters in LDS, binary search when the guess is off by more than one) and routes
//      (key, row id) through per-bucket LDS rings into workgroup-private slabs, whole 128-byte
//      lines only -- the write-combining scheme of k_fgb.hip's producer;
//   4. jbucket_kernel: one workgroup per bucket stages its slice of sorted build keys in LDS
//      (96 KiB; longer slices in rounds) with a bitmap of hashed keys in front, streams the bucket's
//      probe pairs, queues the candidates per wave and binary-searches them 64 at a time: a hit
//      yields the GLOBAL rank of the first equal build entry, so a matching probe row becomes
//      (rank, left row id), appended wave by wave to the bucket's survivor slab;
//   5. jorder_kernel: the survivors of a bucket (only the matching rows) are brought into
//      (rank, left row) order inside the CU.  The bucket kernel has already dealt them into BINS --
//      fixed, equal ranges of the bucket's ranks, sized so that a bin's survivors fit the LDS stage --
//      and counted them per group of ranks, so the order kernel reads each bin twice (one LDS
//      counter per rank, then placement into the stage), places every row inside its key's run by
//      counting the smaller row ids, and writes (rank, left row, partner count) / the carried
//      columns; buckets are rank ranges in ascending order, so the concatenation is sorted (a rank
//      with > 64 probe rows: two radix sorts instead);
//   6. unique build keys: survivor i IS output row i (the right row id is one gather away);
//      otherwise the partner counts are scanned and one lane per matching row writes its
//      (left row id, right row id) pairs -- join_expand_kernel of k_join.hip.
// Skewed probe keys can overflow a slab: the kernel reports it and the caller falls back.
#include "hark_internal.h"
#include "sort_networks.h"
#include <type_traits>

int k_sort_column(hark_context *ctx, const void *col, int dtype, int64_t n, bool descending, const uint32_t *payload,
                  uint32_t **vals_out, uint32_t **words_out);

namespace {

__device__ __forceinline__ void st_hidden_b128(void *p, uint4 v)          // (see st_hidden_nt_b128 in hark_internal.h: the s_nop is part of it)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
}

constexpr int kJThreads = 1024;
constexpr int kJErrOverflow = 100;
constexpr int kJIdx = 2048;                               // entries of the bucket kernel's radix index over a round's keys
constexpr int kMaxBins = 64;                             // survivor bins per bucket (fixed rank ranges, written by the bucket kernel, read by the order kernel)
constexpr int kBinSlots = kMaxBins + 1;                  // ... their fill counts per bucket, and the overflow area's behind them
constexpr size_t kJLdsBudget = 160 * 1024 - 512;          // one workgroup per CU owns (almost) all of its LDS

struct JPair32 { uint32_t key, row; };                                    // 8 bytes: 16 per 128-byte line
struct __attribute__((aligned(16))) JPair64 { uint64_t key; uint32_t row, pad; };   // 16 bytes: 8 per line

// the order kernel's geometry (jorder_kernel): survivors staged per sub-round and per-rank counters -- with a third word per
// survivor 10240 x 12 B + 6144 counters, without 12288 x 8 B + 10240 counters (~150 KiB of LDS either way)
constexpr int kStage = 12288, kStageCarry = 10240, kFine = 10240, kFineCarry = 6144, kCoarse = 2048, kTieMax = 64;
constexpr int kLongMax = kStage / (kTieMax + 1) + 1;      // runs of more than kTieMax rows that fit the stage
constexpr int kWaveSortMax = 2048;                        // ... sorted by one wave up to this length (sixteen runs at a time), by the whole workgroup beyond (one after the other:
                                                          // a hundred neighbouring keys with ~1000 rows each took 1.3 ms that way while this was 1024)
// the stage by words per survivor: (rank, left row) + the carried column + the probe key's low word (64-bit keys: confirmed at write-out)
constexpr int stage_of(int extra_words) { return extra_words == 0 ? kStage : extra_words == 1 ? kStageCarry : 8192; }
constexpr int fine_of(int extra_words) { return extra_words == 0 ? kFine : extra_words == 1 ? kFineCarry : 4096; }

// Bins of a bucket's survivors: the bucket's len ranks are cut into nb <= kMaxBins ranges of gw GROUPS of the coarse histogram
// (2^gs ranks each), as few as keep a bin's EXPECTED survivors a twelfth under the order kernel's stage when every second
// probe pair of the bucket finds a partner and the partners are spread evenly; a fuller bin simply takes several sub-rounds
// in the order kernel -- each of which sweeps the WHOLE bin, which is why the width is any number of groups and not a power
// of two: with 2^bs ranks per bin BASELINE configs[3] landed on 12 bins of 10.2 K survivors for a stage of 8192, every bin
// took two sub-rounds and the order kernel read its survivors four times instead of twice (2.66 GB against 1.4 GB by the
// counters).  Both kernels derive the geometry from the same inputs.
struct JBins {
    int gs, nb; uint32_t gw, magic, cap, ov_at, ov_cap;                  // ov_*: the bucket's overflow area (entries that found their bin full)
    __device__ __forceinline__ uint32_t bin_of_group(uint32_t g) const { return gw == 1u ? g : __umulhi(g, magic); }   // g / gw, exact for g, gw < 2^16
};
__device__ __forceinline__ JBins jbins_of(uint32_t len, uint32_t nprobe, int stage_cap, size_t region)
{
    JBins g;
    g.gs = 0;
    while (((len + (1u << g.gs) - 1u) >> g.gs) > (uint32_t)kCoarse) g.gs++;
    const uint32_t ngroups = max(1u, (len + (1u << g.gs) - 1u) >> g.gs), room = (uint32_t)(stage_cap - stage_cap / 12);
    uint32_t want = (nprobe / 2u + room - 1u) / room;
    want = want < 1u ? 1u : want > (uint32_t)kMaxBins ? (uint32_t)kMaxBins : want;
    g.gw = (ngroups + want - 1u) / want;
    g.nb = (int)((ngroups + g.gw - 1u) / g.gw);
    g.magic = g.gw == 1u ? 0u : 0xFFFFFFFFu / g.gw + 1u;
    // Half of the region is dealt out to the bins, the other half takes the survivors that find their bin full, in the order
    // they come (the order kernel reads them in every sub-round, filtered by rank).  A bucket's probe pairs fill at most half
    // of the region (that is how the caller sizes it), so NO distribution of the hits over the ranks overflows anything: a bin
    // crowded by a few frequent keys only costs the order kernel extra sweeps of that bucket's overflow.
    g.ov_at = (uint32_t)((region / 2) & ~(size_t)15);
    g.ov_cap = (uint32_t)(region - g.ov_at);
    g.cap = (uint32_t)(((size_t)g.ov_at / (size_t)g.nb) & ~(size_t)15);
    return g;
}

template <typename K> struct JTraits;
// P buckets, rings of Q entries, VEC rows per lane and batch; the bucket kernel stages CHUNK sorted build keys per
// round plus a bitmap of 2^BM_BITS bits over them (measured: a 16-step binary search in LDS for EVERY probe pair
// cost 1.36 ms per 1e8 pairs -- instruction issue, not the loads; most pairs have no partner and now leave after
// one bit test).  u32: 88 KiB of keys + 32 KiB of bitmap (+ 24 KiB of candidate queues + 8 KiB of rank-group counters), one
// round up to 1.15e7 build rows; u64: 96 + 16 (+ 32 + 8) KiB.
// (512 x 32 / 512 x 16 measured against 256 x 64 / 1024 x 8: profiles/r02_notes.md, r03_notes.md 5.1)
template <> struct JTraits<uint32_t> { typedef JPair32 E; static constexpr int P = 512, Q = 32, VEC = 4, CHUNK = 22528, BM_BITS = 18; };
template <> struct JTraits<uint64_t> { typedef JPair64 E; static constexpr int P = 512, Q = 16, VEC = 2, CHUNK = 12288, BM_BITS = 17; };

__device__ __forceinline__ uint32_t jhash(uint32_t k) { return k * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t jhash(uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 32); }

__device__ __forceinline__ bool jwg_or(bool pred, uint32_t *flags, int &phase)
{
    const int s = phase;
    phase = s == 2 ? 0 : s + 1;
    if (__ballot(pred) != 0ull && (threadIdx.x & 63) == 0) flags[s] = 1u;
    if (threadIdx.x == 0) flags[phase] = 0u;
    lds_barrier();
    return flags[s] != 0u;
}

// runlen[i] = number of build entries equal to rkeys[i] when i starts a run, else 0.
template <typename K>
__global__ __launch_bounds__(256) void jrunlen_kernel(const K *__restrict__ rkeys, int64_t s, uint32_t *__restrict__ runlen, int32_t *__restrict__ dup)
{
    bool any = false;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s; i += stride) {
        const K key = rkeys[i];
        uint32_t len = 0;
        if (i == 0 || rkeys[i - 1] != key) {
            int64_t ub = i + 1;
            int steps = 0;
            while (ub < s && rkeys[ub] == key && steps < 8) { ub++; steps++; }
            if (ub < s && rkeys[ub] == key) {
                int64_t a = ub, b = s;
                while (a < b) { const int64_t mid = (a + b) >> 1; if (rkeys[mid] <= key) a = mid + 1; else b = mid; }
                ub = a;
            }
            len = (uint32_t)(ub - i);
            any = any || len > 1u;
        }
        runlen[i] = len;
    }
    if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) *dup = 1;       // benign race: every writer stores 1
}

__global__ __launch_bounds__(256) void jcnt_kernel(const uint32_t *__restrict__ rank, int64_t m, const uint32_t *__restrict__ runlen, uint32_t *__restrict__ cnt)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) cnt[i] = runlen[rank[i]];
}

// ---- heavy hitters of the probe side -----------------------------------------------------------------------------
// A foreign-key column is skewed: a few keys own a large share of the probe rows.  Sent through the partition they would
// flood ONE bucket (its slabs and survivor bins are sized for an even share) and one rank of the order kernel.  They
// never get there:
//   1. jhot_sample_kernel reads one probe key in n / S (S = n / 256, at most 131072) and counts equal samples in a hash table;
//      jhot_select_kernel keeps the (at most kHotMax) keys sampled cmin times or more, looks up their ranks in the sorted
//      build side and lays out an open-addressing set of them that fits the partition kernel's LDS;
//   2. jpart_kernel drops the rows of a hot key (one LDS probe per row, nothing at all when the set is empty) and counts,
//      batch by batch, those whose key has partners;
//   3. the rows of a hot key are a contiguous block of the output at the key's rank, IN ROW ORDER -- which a stable
//      partition of the probe column delivers without any sorting: the counts are scanned (jsum_kernel), the hot rows
//      are written out in row order as (key index, row) by a second pass over the column (jhot_compact_kernel, a wave per
//      batch), ONE stable pass of the radix sort groups them by key, the order kernel leaves the blocks free (every later
//      row of the bucket moves back by the blocks before it) and reports where they start, and jhot_place_kernel copies
//      the groups there.
// One more pass over the probe keys, paid only when a hot key exists (the kernels leave at once when the set is empty).
constexpr int kHotMax = 256, kHotSlots = 1024, kHotCand = 4096, kHotSampleMax = 131072, kQuant = 4096;
constexpr uint32_t kNoRank = 0xFFFFFFFFu;
constexpr uint16_t kNoHot = 0xFFFFu;

struct JHotHead {
    uint32_t H, ncand, Hp, pad;                // hot keys, candidates of the sample, hot keys with partners
    unsigned long long mhot;                   // matching probe rows of all hot keys
    // the keys with partners, ascending (dense): what the order kernel and the placement need
    uint32_t prank[kHotMax], pbucket[kHotMax]; // rank of the key's first sorted build entry; the bucket of that rank
    unsigned long long prows[kHotMax];         // probe rows of the key
    unsigned long long pbefore[kHotMax];       // rows of the hot keys before it
    unsigned long long pdst[kHotMax];          // first output row of its block (written by the order kernel)
    uint16_t slot_p[kHotSlots];                // by slot of the set: the key's index among those with partners, kNoHot: it has none (its rows are dropped)
    uint32_t cand[kHotCand];                   // slots of the sample table whose count reached cmin
};
template <typename K> struct JHotSet : JHotHead { K empty; K slots[kHotSlots]; };   // empty: a value that is no hot key marks the free slots

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // a wave's own LDS traffic, in program order

template <typename K> __device__ __forceinline__ uint32_t jhot_find(const K *slots, K empty, K key)       // the key's slot, ~0u: not a hot key
{
    uint32_t slot = jhash(key) >> 22;
    K x = slots[slot];
    if (x == empty) return ~0u;                                           // (first: a probe key may equal the marker) -- where almost every row leaves
    while (x != key) {
        slot = (slot + 1u) & (uint32_t)(kHotSlots - 1);
        x = slots[slot];
        if (x == empty) return ~0u;
    }
    return slot;
}

// table: tkey[mask + 1] (key + 1; 0 = free: one memset clears everything the hot path owns), tcnt[mask + 1]; mask + 1 >= 4 S, so a
// free slot is always found.  A workgroup
// counts its 1024 samples in LDS first and adds every distinct key once: the samples of a frequent key (one row in twenty is
// 13000 of them) would otherwise queue up behind ONE word of the table.
template <typename K>
__global__ __launch_bounds__(1024) void jhot_sample_kernel(const K *__restrict__ keys, int64_t n, K bias, uint32_t S, unsigned long long *__restrict__ tkey,
                                                           uint32_t *__restrict__ tcnt, uint32_t mask, uint32_t cmin, JHotHead *__restrict__ hot)
{
    constexpr int LS = 2048;                                              // local slots: twice the samples
    __shared__ unsigned long long s_key[LS];
    __shared__ uint32_t s_cnt[LS];
    const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
    for (int i = threadIdx.x; i < LS; i += 1024) { s_key[i] = ~0ull; s_cnt[i] = 0u; }
    __syncthreads();
    if (t < S) {
        const uint64_t stride = (uint64_t)n / S;                          // >= 1: S <= n
        const uint64_t row = (uint64_t)t * stride + (uint64_t)((t * 0x9E3779B1u) >> 7) % stride;   // a fixed stride would lock onto periodic data
        const unsigned long long k = (unsigned long long)(K)(keys[row] ^ bias);
        if (k != ~0ull) {                                                 // (the free marker: such a key is never hot)
            uint32_t slot = jhash((uint64_t)k) >> 21;
            for (;;) {
                const unsigned long long old = atomicCAS(&s_key[slot], ~0ull, k);
                if (old == ~0ull || old == k) break;
                slot = (slot + 1u) & (uint32_t)(LS - 1);
            }
            atomicAdd(&s_cnt[slot], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LS; i += 1024) {
        const uint32_t c = s_cnt[i];
        if (c == 0u) continue;
        const unsigned long long k = s_key[i], k1 = k + 1ull;             // (k is not all ones)
        uint32_t slot = (jhash((uint64_t)k) ^ (uint32_t)k) & mask;
        for (;;) {
            const unsigned long long old = atomicCAS(&tkey[slot], 0ull, k1);
            if (old == 0ull || old == k1) break;
            slot = (slot + 1u) & mask;
        }
        const uint32_t before = atomicAdd(&tcnt[slot], c);
        if (before < cmin && before + c >= cmin) { const uint32_t at = atomicAdd(&hot->ncand, 1u); if (at < (uint32_t)kHotCand) hot->cand[at] = slot; }
    }
}

// Where the sampled rows fall in the sorted build side, per bucket of the even cut (to 1 / kQuant of the build side: quantile keys
// in LDS): jhot_select_kernel sees from it whether the probe rows crowd a stretch of the build side.  Keys outside the build
// side's range are dropped by the partition and weigh nothing.  (Its own kernel since the sample runs BEFORE the build side is
// sorted, on the context's second stream, hidden behind the sort: this one needs the sorted keys.)
template <typename K>
__global__ __launch_bounds__(1024) void jhot_coarse_kernel(const unsigned long long *__restrict__ tkey, const uint32_t *__restrict__ tcnt, uint32_t tslots,
                                                           const K *__restrict__ rkeys, int64_t s, uint32_t *__restrict__ hcoarse, int P)
{
    __shared__ K s_q[kQuant];                                             // every (s / kQuant)-th sorted build key
    __shared__ uint32_t s_coarse[1024];                                   // this workgroup's samples per even bucket (P <= 1024): added to hcoarse once
    s_coarse[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < kQuant; i += 1024) s_q[i] = rkeys[(int64_t)(((uint64_t)i * (uint64_t)s) / kQuant)];
    __syncthreads();
    const K kmax = rkeys[s - 1];
    const uint32_t stride = gridDim.x * 1024u;
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < tslots; i += stride) {
        const unsigned long long k1 = tkey[i];
        if (k1 == 0ull) continue;
        const K key = (K)(k1 - 1ull);
        if (key >= s_q[0] && key <= kmax) {
            int lo = 0, hi = kQuant;                                       // the number of quantile keys <= key, less one: its stretch
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_q[mid] <= key) lo = mid; else hi = mid; }
            atomicAdd(&s_coarse[(uint32_t)(((uint64_t)lo * (uint64_t)P) / kQuant)], tcnt[i]);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < P && s_coarse[threadIdx.x]) atomicAdd(&hcoarse[threadIdx.x], s_coarse[threadIdx.x]);
}

// The sampled keys' places in the sorted build side, cell by cell (cells of s / cells ranks) -- only when the buckets are to be cut
// by weight (info[4], decided by jhot_select_kernel): one binary search per distinct sampled key that is no hot key.
template <typename K>
__global__ __launch_bounds__(256) void jhot_fine_kernel(const unsigned long long *__restrict__ tkey, const uint32_t *__restrict__ tcnt, uint32_t tslots,
                                                        const K *__restrict__ rkeys, int64_t s, const JHotSet<K> *__restrict__ hot, const int64_t *__restrict__ info,
                                                        uint32_t *__restrict__ hfine, uint32_t cells)
{
    if (info[4] == 0) return;
    __shared__ K s_slots[kHotSlots];
    const bool any_hot = hot->H != 0u;
    const K empty = any_hot ? hot->empty : (K)0;
    if (any_hot) for (int i = threadIdx.x; i < kHotSlots; i += 256) s_slots[i] = hot->slots[i];
    __syncthreads();
    const K kmin = rkeys[0], kmax = rkeys[s - 1];
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < tslots; i += stride) {
        const unsigned long long k1 = tkey[i];
        if (k1 == 0ull) continue;
        const K key = (K)(k1 - 1ull);
        if (key < kmin || key > kmax || (any_hot && jhot_find<K>(s_slots, empty, key) != ~0u)) continue;   // (a hot key's rows do not go through the partition)
        int64_t lo = 0, hi = s;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rkeys[mid] < key) lo = mid + 1; else hi = mid; }
        atomicAdd(&hfine[(uint32_t)(((uint64_t)lo * cells) / (uint64_t)s)], tcnt[i]);
    }
}

// one launch clears everything the join's kernels expect zeroed (a hipMemsetAsync of a few megabytes of odd size becomes several
// fill kernels ~10 us apart: 80 us of a 2.6 ms join went there)
__global__ __launch_bounds__(256) void jclear_kernel(uint4 *__restrict__ p, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = uint4{0u, 0u, 0u, 0u};
}

template <typename K>
__device__ __forceinline__ void jhot_select_body(const unsigned long long *__restrict__ tkey, const uint32_t *__restrict__ tcnt, uint32_t cmin,
                                                 const K *__restrict__ rkeys, int64_t s, JHotSet<K> *__restrict__ hot, uint32_t *__restrict__ hcoarse, int P)
{
    __shared__ K s_key[kHotMax];
    __shared__ uint32_t s_cs[kHotMax];                                    // the sample-table slot of the key (its number of samples)
    __shared__ uint32_t s_rank[kHotMax];                                  // by place in ascending key order
    __shared__ K s_slots[kHotSlots];
    __shared__ uint16_t s_p[kHotSlots];
    __shared__ uint32_t s_n, s_free;
    const int tid = threadIdx.x;
    const uint32_t nc = min(hot->ncand, (uint32_t)kHotCand);
    if (nc == 0) return;                                                  // H = 0 (the caller cleared the block)
    uint32_t thr = cmin;                                                  // the smallest cmin * 2^j that leaves at most kHotMax keys
    for (;;) {
        if (tid == 0) s_n = 0u;
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t i = tid; i < nc; i += 1024) mine += tcnt[hot->cand[i]] >= thr ? 1u : 0u;
        if (mine) atomicAdd(&s_n, mine);
        __syncthreads();
        const uint32_t left = s_n;
        __syncthreads();
        if (left <= (uint32_t)kHotMax) break;
        thr *= 2u;
    }
    if (tid == 0) { s_n = 0u; s_free = (uint32_t)kHotMax + 1u; }
    __syncthreads();
    for (uint32_t i = tid; i < nc; i += 1024) {
        const uint32_t slot = hot->cand[i];
        if (tcnt[slot] >= thr) { const uint32_t at = atomicAdd(&s_n, 1u); s_key[at] = (K)(tkey[slot] - 1ull); s_cs[at] = slot; }
    }
    __syncthreads();
    const int H = (int)s_n;
    if (H == 0) return;
    K mykey = (K)0;
    int pos = 0;
    if (tid < H) {
        mykey = s_key[tid];
        for (int j = 0; j < H; j++) pos += s_key[j] < mykey ? 1 : 0;      // the keys are distinct: its place in ascending order
        int64_t lo = 0, hi = s;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rkeys[mid] < mykey) lo = mid + 1; else hi = mid; }
        s_rank[pos] = (lo < s && rkeys[lo] == mykey) ? (uint32_t)lo : kNoRank;
        if (mykey >= rkeys[0] && mykey <= rkeys[s - 1]) {                 // its rows do not go through the partition: they weigh nothing in the cut
            // (the stretch the sample kernel counted it in: the last quantile key <= mykey)
            int qa = 0, qz = kQuant;
            while (qz - qa > 1) { const int mid = (qa + qz) >> 1; if (rkeys[(int64_t)(((uint64_t)mid * (uint64_t)s) / kQuant)] <= mykey) qa = mid; else qz = mid; }
            atomicSub(&hcoarse[(uint32_t)(((uint64_t)qa * (uint64_t)P) / kQuant)], tcnt[s_cs[tid]]);
        }
    }
    if (tid <= H) {                                                       // the smallest of 0 .. H that is no hot key marks the free slots
        bool used = false;
        for (int j = 0; j < H; j++) used = used || s_key[j] == (K)tid;
        if (!used) atomicMin(&s_free, (uint32_t)tid);
    }
    __syncthreads();
    const K empty = (K)s_free;
    s_slots[tid] = empty; s_p[tid] = kNoHot;                              // (kHotSlots threads)
    int p = 0, hp = 0;                                                    // its place among the keys with partners (ascending keys = ascending ranks)
    for (int j = 0; j < H; j++) { const int has = s_rank[j] != kNoRank ? 1 : 0; if (j < pos) p += has; hp += has; }
    __syncthreads();
    if (tid < H) {
        uint32_t slot = jhash(mykey) >> 22;
        for (;;) {
            typedef typename std::conditional<sizeof(K) == 8, unsigned long long, unsigned int>::type A;
            const A old = atomicCAS(reinterpret_cast<A *>(&s_slots[slot]), (A)empty, (A)mykey);
            if (old == (A)empty) break;
            slot = (slot + 1u) & (uint32_t)(kHotSlots - 1);
        }
        if (s_rank[pos] != kNoRank) { s_p[slot] = (uint16_t)p; hot->prank[p] = s_rank[pos]; }
    }
    __syncthreads();
    hot->slots[tid] = s_slots[tid]; hot->slot_p[tid] = s_p[tid];
    if (tid == 0) { hot->empty = empty; hot->H = (uint32_t)H; hot->Hp = (uint32_t)hp; }
}

// The hot keys (above), then the BUCKETS: bstart[b] = first sorted build position of bucket b (b = 0..P), splitters[b - 1] = first
// key of bucket b.  Evenly spread probe keys: P equal slices of the sorted build side, whatever its key distribution
// (rank b s / P).  But probe rows may crowd a stretch of the build side without any single key being frequent enough
// for the sample to call it hot -- ten thousand neighbouring keys with three thousand rows each: a third of the probe side
// in ONE even bucket, whose slabs overflow (11.9 ms through the sort-merge path, profiles/r05_join_skew.txt).  The sample says
// where the rows fall (hfine: samples per cell of s / cells ranks; hcoarse: per even bucket); when an even bucket holds more
// than 1.5 x its share (+ noise), the cut is made by WEIGHT instead: a bucket ends where 0.2 x (share of the build entries)
// + 0.8 x (share of the sampled rows) reaches b / P -- at most 1.25 x the even share of the probe rows (+ a cell) and at most
// 5 x the even share of the build entries (more rounds for a bucket with few probe rows).  One workgroup; the scan of the
// cells runs only when the cut is by weight.
template <typename K>
__global__ __launch_bounds__(1024) void jhot_select_kernel(const unsigned long long *__restrict__ tkey, const uint32_t *__restrict__ tcnt, uint32_t cmin,
                                                           const K *__restrict__ rkeys, int64_t s, JHotSet<K> *__restrict__ hot, uint32_t *__restrict__ hcoarse, int P,
                                                           K *__restrict__ splitters, uint32_t *__restrict__ bstart, int allow_weighted, int64_t *__restrict__ info)
{
    __shared__ unsigned long long s_T;
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x, lane = tid & 63;
    jhot_select_body<K>(tkey, tcnt, cmin, rkeys, s, hot, hcoarse, P);
    if (tid == 0) { s_T = 0ull; s_max = 0u; }
    __syncthreads();
    {   // how crowded is the most crowded even bucket?
        uint32_t v = tid < P ? hcoarse[tid] : 0u, mx = v;
        unsigned long long sum = v;
        for (int d = 32; d > 0; d >>= 1) { sum += __shfl_down(sum, d, 64); mx = max(mx, (uint32_t)__shfl_down((int)mx, d, 64)); }
        if (lane == 0 && sum) { atomicAdd(&s_T, sum); atomicMax(&s_max, mx); }
    }
    __syncthreads();
    const double mean = (double)s_T / (double)P;
    const bool weighted = allow_weighted && s_T >= 2048ull && (double)s_max > 1.5 * mean + 5.0 * sqrt(mean) + 8.0 && s >= 4 * (int64_t)P;
    if (tid <= P) {                                                       // the even cut (jhot_cut_kernel overrides it when the cut is by weight)
        const int b = tid;
        if (b == 0) bstart[0] = 0u;
        else if (b == P) bstart[P] = (uint32_t)s;
        else {
            const K key = rkeys[(int64_t)b * s / P];
            int64_t lo = 0, hi = s;                                    // a run of equal keys is never cut
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rkeys[mid] < key) lo = mid + 1; else hi = mid; }
            bstart[b] = (uint32_t)lo;
            splitters[b - 1] = key;
        }
    }
    if (tid == 0) info[4] = weighted ? 1 : 0;                            // (jhot_fine_kernel / jhot_cut_kernel run on it; the host reads it with the totals)
}

// The cut by weight (see above; both kernels leave at once otherwise).  First the cells' counts -> a running sum inside chunks of
// kCutChunk cells (one workgroup each: a single workgroup took 0.35 ms over 2^20 cells), the chunks' totals aside; then bucket b
// ends behind the first cell c with
//     2 (c + 1) T + 8 C rows(c) >= 10 C T b / P      (0.2 x the build entries' share + 0.8 x the sampled rows' share reaches b / P).
constexpr int kCutChunk = 16384;
__global__ __launch_bounds__(1024) void jhot_scan_cells_kernel(uint32_t *__restrict__ hfine, uint32_t cells, uint32_t *__restrict__ ctot, const int64_t *__restrict__ info)
{
    if (info[4] == 0) return;
    __shared__ uint32_t s_w2[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)kCutChunk + (uint32_t)tid * 16u;
    uint4 q[4];
    uint4 *v = reinterpret_cast<uint4 *>(hfine + base);
    const bool mine = base < cells;                                       // (cells is a multiple of 16)
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = mine ? v[k] : uint4{0u, 0u, 0u, 0u};
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { q[k].x += run; q[k].y += q[k].x; q[k].z += q[k].y; q[k].w += q[k].z; run = q[k].w; }
    uint32_t incl = run;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) s_w2[wave] = incl;
    __syncthreads();
    uint32_t before = incl - run, total = 0;
    for (int w = 0; w < 16; w++) { const uint32_t x = s_w2[w]; if (w < wave) before += x; total += x; }
    if (mine) {
#pragma unroll
        for (int k = 0; k < 4; k++) { q[k].x += before; q[k].y += before; q[k].z += before; q[k].w += before; v[k] = q[k]; }
    }
    if (tid == 0) ctot[blockIdx.x] = total;
}

template <typename K>
__global__ __launch_bounds__(1024) void jhot_cut_kernel(const K *__restrict__ rkeys, int64_t s, const uint32_t *__restrict__ hfine, uint32_t cells, const uint32_t *__restrict__ ctot,
                                                        int P, K *__restrict__ splitters, uint32_t *__restrict__ bstart, const int64_t *__restrict__ info)
{
    if (info[4] == 0) return;
    __shared__ uint32_t s_cpre[64 + 1];                                   // rows before a chunk (2^20 cells: 64 chunks)
    const int tid = threadIdx.x;
    const uint32_t nchunks = (cells + (uint32_t)kCutChunk - 1u) / (uint32_t)kCutChunk;
    if (tid == 0) { uint32_t run = 0; for (uint32_t c = 0; c < nchunks; c++) { s_cpre[c] = run; run += ctot[c]; } s_cpre[nchunks] = run; }
    __syncthreads();
    const unsigned long long Tf = s_cpre[nchunks];
    if (tid >= 1 && tid < P && Tf) {
        const int b = tid;
        const unsigned long long C = cells, want = (10ull * C * Tf * (unsigned long long)b) / (unsigned long long)P;
        uint32_t lo = 0, hi = cells - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const unsigned long long rows = (unsigned long long)s_cpre[mid / (uint32_t)kCutChunk] + hfine[mid];
            if (2ull * (mid + 1ull) * Tf + 8ull * C * rows >= want) hi = mid; else lo = mid + 1u;
        }
        int64_t pos = (int64_t)(((unsigned long long)(lo + 1u) * (unsigned long long)s) / C);
        if (pos > s - 1) pos = s - 1;
        const K key = rkeys[pos];
        int64_t a = 0, z = s;                                            // a run of equal keys is never cut
        while (a < z) { const int64_t mid = (a + z) >> 1; if (rkeys[mid] < key) a = mid + 1; else z = mid; }
        bstart[b] = (uint32_t)a;
        splitters[b - 1] = key;
    }
}

// the rows of the hot keys with partners, in row order: (index among those keys, row); a wave per batch of the partition
// kernel (BATCH rows).  One stable pass of the radix sort over the index then groups them by key with the row order kept --
// the reference's order inside a key (join.fut:66).  The keys' rows are counted on the way (prows).
template <typename K>
__global__ __launch_bounds__(1024) void jhot_compact_kernel(const K *__restrict__ keys, int64_t n, K bias, JHotSet<K> *__restrict__ hot,
                                                            const uint32_t *__restrict__ bprefix, uint32_t nbatch,
                                                            uint32_t *__restrict__ hkey, uint32_t *__restrict__ hrow)
{
    constexpr int VEC = JTraits<K>::VEC, BATCH = kJThreads * VEC;
    __shared__ K s_slots[kHotSlots];
    __shared__ uint16_t s_p[kHotSlots];
    __shared__ uint32_t s_cnt[kHotMax];
    const uint32_t Hp = hot->Hp;
    if (Hp == 0u || hot->mhot == 0ull) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_slots[tid] = hot->slots[tid]; s_p[tid] = hot->slot_p[tid];
    if (tid < kHotMax) s_cnt[tid] = 0u;
    const K empty = hot->empty;
    __syncthreads();
    const uint32_t batch = blockIdx.x * 16u + (uint32_t)wave;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t r0 = (int64_t)batch * BATCH, r1 = batch < nbatch ? min(n, r0 + (int64_t)BATCH) : r0;
    uint32_t at = batch < nbatch ? bprefix[batch] : 0u;                   // wave-uniform: the next free place of the stream
    constexpr int U = 4;                                                  // loads in flight per lane
    for (int64_t base = r0; base < r1; base += (int64_t)U * 64 * VEC) {
        K kk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t r = base + ((int64_t)u * 64 + lane) * VEC;
            if (r + VEC <= n) {
                const hark_u4v q = __builtin_nontemporal_load(reinterpret_cast<const hark_u4v *>(keys + r));
                if (sizeof(K) == 4) { kk[u][0] = (K)q.x; kk[u][1 % VEC] = (K)q.y; kk[u][2 % VEC] = (K)q.z; kk[u][3 % VEC] = (K)q.w; }
                else { kk[u][0] = (K)(((uint64_t)q.y << 32) | q.x); kk[u][1 % VEC] = (K)(((uint64_t)q.w << 32) | q.z); }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; j++) kk[u][j] = r + j < n ? keys[r + j] : (K)0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t r = base + ((int64_t)u * 64 + lane) * VEC;
            uint32_t pj[VEC], mine = 0, lower = 0, total = 0;
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                pj[j] = kNoHot;
                if (r + j < r1) { const uint32_t slot = jhot_find<K>(s_slots, empty, kk[u][j] ^ bias); if (slot != ~0u) pj[j] = s_p[slot]; }
                const bool is = pj[j] != kNoHot;
                const unsigned long long m = __ballot(is);
                total += (uint32_t)__popcll(m); lower += (uint32_t)__popcll(m & below);      // rows of lower lanes come first, then this lane's in turn
                if (is) mine |= 1u << j;
            }
            if (total) {
                uint32_t o = at + lower;
#pragma unroll
                for (int j = 0; j < VEC; j++) if (mine & (1u << j)) { hkey[o] = pj[j]; hrow[o] = (uint32_t)(r + j); atomicAdd(&s_cnt[pj[j]], 1u); o++; }
                at += total;
            }
        }
    }
    __syncthreads();
    if ((uint32_t)tid < Hp && s_cnt[tid]) atomicAdd(&hot->prows[tid], (unsigned long long)s_cnt[tid]);
}

// the hot keys' buckets and the rows before each (prows is complete: the compaction has run)
__global__ __launch_bounds__(kHotMax) void jhot_finish_kernel(JHotHead *__restrict__ hot, const uint32_t *__restrict__ bstart, int P)
{
    __shared__ unsigned long long s_rows[kHotMax];
    const int p = threadIdx.x, Hp = (int)hot->Hp;
    s_rows[p] = p < Hp ? hot->prows[p] : 0ull;
    __syncthreads();
    if (p >= Hp) return;
    unsigned long long before = 0;
    for (int j = 0; j < p; j++) before += s_rows[j];
    const uint32_t r = hot->prank[p];
    int lo = 0, hi = P;                                                   // the b with bstart[b] <= r < bstart[b + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bstart[mid] <= r) lo = mid; else hi = mid; }
    hot->pbucket[p] = (uint32_t)lo; hot->pbefore[p] = before; hot->pdst[p] = 0ull;
}

// the stream of hot rows, grouped by key and in row order inside a key, to the blocks the order kernel left free
__global__ __launch_bounds__(256) void jhot_place_kernel(const JHotHead *__restrict__ hot, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ srow, int64_t mhot,
                                                         const uint32_t *__restrict__ runlen, const uint32_t *__restrict__ lval, const uint32_t *__restrict__ rranked,
                                                         uint32_t *__restrict__ rank_out, uint32_t *__restrict__ lrow_out, uint32_t *__restrict__ cnt_out,
                                                         uint32_t *__restrict__ lval_out, uint32_t *__restrict__ rval_out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < mhot; i += stride) {
        const uint32_t p = skey[i], row = srow[i], rk = hot->prank[p];
        const unsigned long long o = hot->pdst[p] + ((unsigned long long)i - hot->pbefore[p]);
        if (rank_out) { rank_out[o] = rk; lrow_out[o] = row; }
        if (cnt_out) cnt_out[o] = runlen[rk];
        if (lval_out) lval_out[o] = lval[row];
        if (rval_out) rval_out[o] = rranked[rk];
    }
}

// ---- probe side: range partition of (key, row id) ---------------------------------------------------------------
// LDS: E ring[P][Q]; K ext[P + 2]; u32 s_w[P] (head << 16 | count); int s_lcur[P]; u32 flags[4]; K s_hot[kHotSlots] + u16 s_hp[kHotSlots]
// (the set of hot keys, whose rows take the stable partition above instead: read only when there is one).
// HIDDEN: the batches' loads are issued from inline assembly and waited for by hand (below); false: plain non-temporal loads
// whose waits the compiler places (HARK_JOIN_PLAIN_LOADS=1: the cross-check of tests/test_gpu_hjoin.py -- the hand-placed
// waits depend on the compiler never touching a destination register between a load and its wait, which nothing checks at
// build time, so the tests run every join shape through both kernels).
template <typename K, bool HIDDEN, bool ROT = false /* rotated loads compiled in: see `rot` */>
__global__ __launch_bounds__(kJThreads) void jpart_kernel(const K *__restrict__ keys, int64_t n, K bias, const K *__restrict__ splitters,
                                                          const K *__restrict__ rkeys, int64_t s,
                                                          typename JTraits<K>::E *__restrict__ slabs, uint32_t *__restrict__ counts, uint32_t cap,
                                                          int period, int32_t *__restrict__ err,
                                                          const uint32_t *__restrict__ lval /* 16-byte entries only, may be null: a probe-side column that
                                                                                               travels in the entries' fourth word */,
                                                          const JHotSet<K> *__restrict__ hot, uint32_t *__restrict__ btotal /* [batches] rows of hot keys with partners */,
                                                          int64_t rot = 0 /* ROT: the 64 sixteen-lane groups of the workgroup read their rows of 64 DIFFERENT batches, group g those
                                                                             of batch + g * rot (mod the full batches) -- a probe column sorted block by block (sorted files one
                                                                             behind the other) then reaches 64 buckets per batch instead of one or two; for a fixed group the
                                                                             map is a rotation of the batches: every row is read once (as fgb_part_kernel's, k_fgb.hip) */)
{
    typedef typename JTraits<K>::E E;
    constexpr int P = JTraits<K>::P, Q = JTraits<K>::Q, VEC = JTraits<K>::VEC;
    constexpr int LINE = 128 / (int)sizeof(E), PER_LANE = 16 / (int)sizeof(E);          // entries per line / per 16-byte lane piece
    constexpr int BATCH = kJThreads * VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    E *ring = reinterpret_cast<E *>(lds_raw);
    K *ext = reinterpret_cast<K *>(ring + (size_t)P * Q);             // [P + 2] bucket b holds the keys with ext[b] <= key < ext[b + 1]
    uint32_t *s_w = reinterpret_cast<uint32_t *>(ext + P + 2);
    int *s_lcur = reinterpret_cast<int *>(s_w + P);
    uint32_t *flags = reinterpret_cast<uint32_t *>(s_lcur + P);
    K *s_hot = reinterpret_cast<K *>(flags + 4);                       // (16 bytes on from s_lcur's end: aligned for 8-byte keys)
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const bool any_hot = hot->H != 0u;
    const K hot_empty = any_hot ? hot->empty : (K)0;
    uint16_t *s_hp = reinterpret_cast<uint16_t *>(s_hot + kHotSlots);
    if (any_hot) for (int i = tid; i < kHotSlots; i += kJThreads) { s_hot[i] = hot->slots[i]; s_hp[i] = hot->slot_p[i]; }
    const int cap_lines = (int)(cap / LINE) - 1;                                  // the last line takes the final partial flush
    for (int b = tid; b < P; b += kJThreads) { s_w[b] = 0u; s_lcur[b] = 0; }
    for (int b = tid; b < P + 2; b += kJThreads) ext[b] = b == 0 ? (K)0 : b < P ? splitters[b - 1] : (K)~(K)0;
    if (tid < 4) flags[tid] = 0u;
    const K kmin = rkeys[0], kmax = rkeys[s - 1];
    // The splitters are equidistant quantiles of the sorted build keys, so for evenly spread keys the bucket of a key is
    // close to (key - min) * P / (max - min + 1): the guess is checked against ext[guess - 1 .. guess + 2] (four
    // independent LDS reads, one round trip) and lands on guess - 1, guess or guess + 1; anything else -- clustered
    // keys -- takes the binary search (nine DEPENDENT reads; with the search for every row the kernel took 0.60 ms per
    // 1e8 rows, without any search 0.37 ms).
    const K range = kmax - kmin;
    int gshift = 0;
    while (sizeof(K) == 8 && ((uint64_t)range >> gshift) > 0xFFFFFFFEull) gshift++;
    const uint32_t r32 = (uint32_t)((uint64_t)range >> gshift);
    const bool use_guess = r32 >= (uint32_t)P;
    const uint32_t gmul = use_guess ? (uint32_t)((((uint64_t)P) << 32) / ((uint64_t)r32 + 1ull)) : 0u;
    __syncthreads();
    int phase = 0, since = 0;
    bool overflow = false, aborted = false;
    const int64_t nbatch = (n + BATCH - 1) / BATCH;
    E *myslab = slabs + (size_t)wg * cap;                                       // + b * nwg * cap

    // The batches' loads are issued and waited for by hand (ld_hidden_* / wait_vm): three register sets in turn, a batch is
    // used when the two loaded after it may still be in flight.  Left to the compiler every batch waited with vmcnt(0) --
    // for the prefetch issued just before it: its count of the memory instructions between a load and its use breaks down
    // at the sweep's stores (conditional, in loops) and again at the loop header that merges the paths.  Every batch issues
    // its loads whether or not its rows exist (a lane whose VEC rows do not all lie below n reads rows 0.. and drops them;
    // the last n % VEC rows go through a batch of their own at the end); the sweep's stores are hidden too.
    const bool has_val = sizeof(E) == 16 && lval != nullptr;
    const uint32_t *vsrc = has_val ? lval : reinterpret_cast<const uint32_t *>(keys);     // (any readable address: the words are not used)
    const int64_t nfullb = n / BATCH;                                         // (the ragged last batch stays where it is)
    auto place = [&](int64_t batch) -> int64_t {                              // the batch whose rows this lane reads when the workgroup works on `batch`
        if constexpr (ROT) { if (batch < nfullb) { batch += (int64_t)(tid >> 4) * rot; if (batch >= nfullb) batch -= nfullb; } }
        return batch;
    };
    auto load = [&](int64_t batch, hark_u4v &kq, hark_u2v &vq) {
        int64_t r = place(batch) * BATCH + (int64_t)tid * VEC;
        if (r + VEC > n) r = 0;
        if (HIDDEN) {
            ld_hidden_nt_b128(kq, keys + r);                         // VEC * sizeof(K) = 16 bytes
            if (sizeof(E) == 16) ld_hidden_nt_b64(vq, vsrc + r);     // VEC = 2 rows: one 8-byte load (r is even)
        } else {
            kq = __builtin_nontemporal_load(reinterpret_cast<const hark_u4v *>(keys + r));
            if (sizeof(E) == 16) vq = __builtin_nontemporal_load(reinterpret_cast<const hark_u2v *>(vsrc + r));
        }
    };
    auto arrived = [&](hark_u4v &kq, hark_u2v &vq) {                  // the batch in (kq, vq) is about to be used
        if (!HIDDEN) return;
        constexpr int kAheadLoads = sizeof(E) == 16 ? 4 : 2;           // loads of the two batches behind the one being used
        if (sizeof(E) == 16) wait_vm<kAheadLoads>(kq, vq); else wait_vm<kAheadLoads>(kq);
    };
    auto unpack = [&](const hark_u4v t, const hark_u2v q, K (&kk)[VEC], uint32_t (&vv)[VEC]) {
        if (sizeof(K) == 4) { kk[0] = (K)t.x; kk[1 % VEC] = (K)t.y; kk[2 % VEC] = (K)t.z; kk[3 % VEC] = (K)t.w; }
        else { kk[0] = (K)(((uint64_t)t.y << 32) | t.x); kk[1 % VEC] = (K)(((uint64_t)t.w << 32) | t.z); }
        vv[0] = q.x; vv[1 % VEC] = q.y;
    };

    // rows r .. r + VEC - 1 of this lane (nrows of them exist)
    // (hot_tag: compiled twice -- with the set's probe, and exactly as it was without it: with one body and a run-time test the
    // kernel without hot keys lost 30-60 us per 1e8 rows against round 5's, interleaved on one box)
    auto process = [&](auto hot_tag, int64_t r, int nrows, const K (&kraw)[VEC], const uint32_t (&vraw)[VEC], bool flush_now) {
        constexpr bool HOT = decltype(hot_tag)::value;
        uint32_t pending = 0, hot_rows = 0;
        K kk[VEC];
        uint32_t bk[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            kk[j] = kraw[j] ^ bias;
            if (j < nrows && kk[j] >= kmin && kk[j] <= kmax) {
                if (!HOT) pending |= 1u << j;
                else {
                    const uint32_t slot = jhot_find<K>(s_hot, hot_empty, kk[j]);
                    if (slot == ~0u) pending |= 1u << j;
                    else if (s_hp[slot] != kNoHot) hot_rows++;             // a hot key: its rows take the other road (counted here when it has partners)
                }
            }
            // bucket = number of splitters <= key = the b with ext[b] <= key < ext[b + 1]
            const K key = kk[j];
            uint32_t g = __umulhi((uint32_t)((uint64_t)(key - kmin) >> gshift), gmul);
            g = g < (uint32_t)P ? g : (uint32_t)(P - 1);
            const K em = ext[g > 0u ? g - 1u : 0u], e0 = ext[g], e1 = ext[g + 1u], e2 = ext[g + 2u];
            uint32_t pos;
            if (use_guess && e0 <= key && key < e1) pos = g;
            else if (use_guess && e1 <= key && key < e2) pos = g + 1u;
            else if (use_guess && g > 0u && em <= key && key < e0) pos = g - 1u;
            else {
                pos = 0;
#pragma unroll
                for (int step = P / 2; step > 0; step >>= 1) if (ext[pos + step] <= key) pos += step;      // ext[i + 1] = splitter i
            }
            bk[j] = pos < (uint32_t)P ? pos : (uint32_t)(P - 1);
        }
        if (HOT) {
            if constexpr (ROT) {                                       // (the rows of sixteen lanes lie in one batch)
                for (int d = 8; d > 0; d >>= 1) hot_rows += __shfl_down(hot_rows, d, 16);
                if ((tid & 15) == 0 && hot_rows) atomicAdd(&btotal[r / BATCH], hot_rows);
            } else {
                for (int d = 32; d > 0; d >>= 1) hot_rows += __shfl_down(hot_rows, d, 64);
                if ((tid & 63) == 0 && hot_rows) atomicAdd(&btotal[r / BATCH], hot_rows);  // (the rows of a wave lie in one batch)
            }
        }
        bool again;
        do {
            uint32_t olds[VEC];                                        // the returning atomics of a lane are issued back to back: one LDS round trip
#pragma unroll
            for (int j = 0; j < VEC; j++) { olds[j] = 0u; if (pending & (1u << j)) olds[j] = atomicAdd(&s_w[bk[j]], 1u); }
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                if (pending & (1u << j)) {
                    const uint32_t b = bk[j];
                    const uint32_t old = olds[j], pos = old & 0xFFFFu;
                    if (pos < (uint32_t)Q) {
                        E e; e.key = kk[j]; e.row = (uint32_t)(r + j);
                        if (sizeof(E) == 16) reinterpret_cast<uint32_t *>(&e)[3] = has_val ? vraw[j] : 0u;
                        ring[b * Q + (((old >> 16) + pos) & (Q - 1))] = e;
                        pending &= ~(1u << j);
                    } else atomicSub(&s_w[b], 1u);                     // ring full: retry after the sweep
                }
            }
            const bool full = jwg_or(pending != 0, flags, phase);
            if (!(full || flush_now)) break;
            // ---- sweep: 8 lanes per bucket store its complete 128-byte lines, 16 bytes per lane
            for (int b = tid >> 3; b < P; b += kJThreads / 8) {
                const uint32_t w = s_w[b];
                const int cnt = (int)(w & 0xFFFFu), lines = cnt / LINE;
                if (lines) {
                    const int i = tid & 7, head = (int)(w >> 16), lc = s_lcur[b];
                    for (int q = 0; q < lines; q++) {
                        const uint4 piece = *reinterpret_cast<const uint4 *>(&ring[b * Q + ((head + q * LINE + PER_LANE * i) & (Q - 1))]);
                        if (lc + q < cap_lines) {
                            E *dst = myslab + (size_t)b * nwg * cap + (size_t)(lc + q) * LINE + PER_LANE * i;
                            st_hidden_nt_b128(dst, piece);
                        } else { overflow = true; flags[3] = 1u; }
                    }
                    if (i == 0) {
                        s_w[b] = ((uint32_t)((head + lines * LINE) & (Q - 1)) << 16) | (uint32_t)(cnt - lines * LINE);
                        s_lcur[b] = min(lc + lines, cap_lines);
                    }
                }
            }
            since = 0;
            again = jwg_or(pending != 0, flags, phase);
            // a full slab: the join will not use this partition (the caller falls back), and rows that crowd one bucket -- a
            // probe column sorted by the key -- would sweep its ring 128 times per batch: everybody leaves
            if (flags[3] != 0u) { aborted = true; break; }
        } while (again);
    };

    auto run = [&](auto hot_tag) {
    hark_u4v qA = {0u, 0u, 0u, 0u}, qB = qA, qC = qA;
    hark_u2v wA = {0u, 0u}, wB = wA, wC = wA;
    K kk_[VEC];
    uint32_t vv_[VEC];
    const int64_t nfull = n / VEC * VEC;                              // rows that full lanes cover
    auto rows_of = [&](int64_t batch, int64_t &r) -> int { r = place(batch) * BATCH + (int64_t)tid * VEC; return r + VEC <= nfull ? VEC : 0; };
    load(wg, qA, wA); load((int64_t)wg + nwg, qB, wB); load((int64_t)wg + 2 * (int64_t)nwg, qC, wC);
    const bool has_tail = wg == 0 && nfull < n;                        // the last n % VEC rows: a batch of their own (workgroup 0)
    for (int64_t batch = wg;;) {
        int64_t r;
        int nr;
        if (batch >= nbatch) break;
        arrived(qA, wA);
        unpack(qA, wA, kk_, vv_);
        nr = rows_of(batch, r);
        process(hot_tag, r, nr, kk_, vv_, ++since >= period || (batch + nwg >= nbatch && !has_tail));
        if (aborted) break;
        load(batch + 3 * (int64_t)nwg, qA, wA); batch += nwg;
        if (batch >= nbatch) break;
        arrived(qB, wB);
        unpack(qB, wB, kk_, vv_);
        nr = rows_of(batch, r);
        process(hot_tag, r, nr, kk_, vv_, ++since >= period || (batch + nwg >= nbatch && !has_tail));
        if (aborted) break;
        load(batch + 3 * (int64_t)nwg, qB, wB); batch += nwg;
        if (batch >= nbatch) break;
        arrived(qC, wC);
        unpack(qC, wC, kk_, vv_);
        nr = rows_of(batch, r);
        process(hot_tag, r, nr, kk_, vv_, ++since >= period || (batch + nwg >= nbatch && !has_tail));
        if (aborted) break;
        load(batch + 3 * (int64_t)nwg, qC, wC); batch += nwg;
    }
    if (HIDDEN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the loads still in flight write registers: let them land
    if (has_tail) {
        const int nr = tid == 0 ? (int)(n - nfull) : 0;
        for (int j = 0; j < VEC; j++) { kk_[j] = (tid == 0 && j < nr) ? keys[nfull + j] : (K)0; vv_[j] = (tid == 0 && j < nr && has_val) ? lval[nfull + j] : 0u; }
        process(hot_tag, nfull, nr, kk_, vv_, true);
    }
    };
    if (any_hot) run(std::true_type{}); else run(std::false_type{});
    // ---- what is left (< LINE entries per bucket) goes out as one partial line
    for (int b = tid; b < P; b += kJThreads) {
        const uint32_t w = s_w[b];
        const int l = (int)(w & 0xFFFFu), head = (int)(w >> 16);
        E *dst = myslab + (size_t)b * nwg * cap + (size_t)s_lcur[b] * LINE;
        for (int j = 0; j < l; j++) dst[j] = ring[b * Q + ((head + j) & (Q - 1))];
        counts[(size_t)b * nwg + wg] = (uint32_t)(s_lcur[b] * LINE + l);
    }
    if (overflow) *err = kJErrOverflow;
}

// ---- one workgroup per bucket: build slice in LDS, probe pairs streamed past it --------------------------------
// Most probe pairs have no partner and leave after a range test and ONE bit test; the candidates of a wave's 128-pair
// step would keep a search busy with a part of the lanes, so they are queued (per wave, in LDS) and searched 64 at a time
// with every lane busy (a radix index over the staged keys narrows each search to a dozen keys).
template <typename K>
__global__ __launch_bounds__(kJThreads) void jbucket_kernel(const typename JTraits<K>::E *__restrict__ slabs, const uint32_t *__restrict__ counts,
                                                            uint32_t cap, int nwg, const K *__restrict__ rkeys, const uint32_t *__restrict__ bstart,
                                                            int chunk_cap, uint2 *__restrict__ surv, size_t region /* survivor entries per bucket */,
                                                            uint32_t *__restrict__ scount /* [P] survivors of the bucket */,
                                                            uint32_t *__restrict__ sbins /* [P][kBinSlots] survivors in every bin, in the overflow area */,
                                                            uint4 *__restrict__ srec /* 64-bit keys (16-byte entries): every survivor as ONE record (rank, left row, the entry's
                                                                                        fourth word, low word of the PROBE key -- a hit on a truncated build key is confirmed by the
                                                                                        order kernel, where the build keys are read in order) instead of `surv` */,
                                                            uint32_t *__restrict__ scoarse /* [P][kCoarse] survivors per group of ranks: the order kernel's first histogram */,
                                                            int stage_cap /* the order kernel's stage (sizes the bins) */, int32_t *__restrict__ err,
                                                            int allow_trunc)
{
    typedef typename JTraits<K>::E E;
    // candidate queues: 8-byte pairs are tested two per lane and step (64 + 128 queued at most); 16-byte entries one per lane
    // and half-step (63 + 64), which leaves room for their fourth word in the queue
    constexpr int BM_BITS = JTraits<K>::BM_BITS, BM_WORDS = 1 << (BM_BITS - 5), QCAP = sizeof(E) == 16 ? 128 : 192;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(lds_raw);                 // [BM_WORDS] one bit per hashed chunk key
    // The sorted keys are stored SKEWED, key i at i + i / 32 (64 for 4-byte words): the first steps of a binary search read
    // keys a power of two apart -- up to 64 lanes on 64 different keys of ONE bank.  (Measured on BASELINE configs[3]'s
    // share: the LDS busy for two thirds of the kernel, 82 % of that bank conflicts.)
    constexpr int SKK = sizeof(K) == 8 ? 5 : 6, CHUNK_PAD = JTraits<K>::CHUNK >> SKK;
    K *chunk = reinterpret_cast<K *>(bitmap + BM_WORDS);                       // [CHUNK + CHUNK_PAD] sorted build keys of this round
    K *qkey_all = chunk + JTraits<K>::CHUNK + CHUNK_PAD;                       // [16 waves][QCAP] candidate keys
    uint32_t *qrow_all = reinterpret_cast<uint32_t *>(qkey_all + (kJThreads / 64) * QCAP);   // [16 waves][QCAP] their row ids
    uint32_t *qval_all = qrow_all + (kJThreads / 64) * QCAP;                   // [16 waves][QCAP] their fourth words (16-byte entries)
    uint32_t *s_coarse = qval_all + (sizeof(E) == 16 ? (kJThreads / 64) * QCAP : 0);   // [kCoarse] survivors per group of 2^gs ranks (for the order kernel)
    // a radix index over the round's keys: s_idx[k] = first key whose (key - first key) >> S is >= k.  A search then starts
    // in a span of m / 2048 keys (a dozen when the keys are spread evenly) instead of all m: ~5 dependent LDS reads per batch
    // of 64 candidates instead of 16.  (Neither this nor the skew above moved the kernel's time on BASELINE configs[3]'s
    // share -- the lookup is not what it waits for, profiles/r03_notes.md 5.1 -- but both take load off the LDS.)
    uint16_t *s_idx = reinterpret_cast<uint16_t *>(s_coarse + kCoarse);                 // [kJIdx + 1]
    __shared__ uint32_t s_bincur[kBinSlots];                              // survivors in every bin so far; [kMaxBins]: in the overflow area
    __shared__ uint32_t s_np, s_gone;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = kJThreads >> 6;
    K *qkey = qkey_all + wave * QCAP;
    uint32_t *qrow = qrow_all + wave * QCAP;
    uint32_t *qval = qval_all + wave * QCAP;
    const uint32_t lo = bstart[b], hi = bstart[b + 1];
    for (int i = tid; i < kCoarse; i += kJThreads) s_coarse[i] = 0u;
    if (tid < kBinSlots) s_bincur[tid] = 0u;
    if (tid == 0) { s_np = 0u; s_gone = (uint32_t)*err; }
    __syncthreads();
    if (s_gone != 0u) return;                                            // the partition gave up (a full slab): nobody reads this bucket's survivors
    {   // the bucket's probe pairs (sizes the bins)
        uint32_t np = 0;
        for (int i = tid; i < nwg; i += kJThreads) np += min(counts[(size_t)b * nwg + i], cap);
        for (int d = 32; d > 0; d >>= 1) np += __shfl_down(np, d, 64);
        if (lane == 0 && np) atomicAdd(&s_np, np);
    }
    __syncthreads();
    const JBins bins = jbins_of(hi - lo, s_np, stage_cap, region);       // the order kernel's grouping of this bucket's ranks and its bins
    const int gs = bins.gs;
    const uint32_t bincap = bins.cap;
    uint2 *out = surv ? surv + (size_t)b * region : nullptr;            // 8-byte entries (32-bit keys)
    uint4 *rout = srec ? srec + (size_t)b * region : nullptr;
    bool bin_full = false;
    const unsigned long long below = (1ull << lane) - 1ull;
    // 64-bit keys: a bucket of up to two chunks of build keys (BASELINE configs[3]: 24.4 K against 12 K per 96-KiB chunk)
    // would stream its probe entries twice.  Instead the chunk area holds the keys TRUNCATED to 32 bits,
    // (key - first key) >> ts, twice as many, and the search runs on those.  Exact among the build keys as long as equal
    // truncated keys are equal keys -- checked when they are loaded; a bucket that fails the check (keys clustered below bit
    // ts) takes the rounds as before.  A PROBE key can still share its truncation with a build key it differs from in the
    // ts dropped bits: every survivor therefore carries its probe key's low word (srec), and the order kernel compares it
    // with the build key of the survivor's rank when it writes the rows out in rank order -- reads that walk the sorted
    // keys front to back.  (Round 3 confirmed every hit here, against the key in memory: 7e7 random 4-byte reads, 0.31 ms of
    // the kernel's 0.99.)  A mismatch anywhere makes the host run the join again without truncated rounds.
    uint32_t *c32 = reinterpret_cast<uint32_t *>(chunk);
    auto ck = [&](int i) -> K & { return chunk[i + (i >> SKK)]; };
    auto c3 = [&](int i) -> uint32_t & { return c32[i + (i >> 6)]; };
    __shared__ int s_tbad;
    bool tmode = sizeof(K) == 8 && allow_trunc && hi - lo > (uint32_t)chunk_cap && hi - lo <= 2u * (uint32_t)chunk_cap;
    K kb = (K)0;
    int ts = 0;
    if (tmode) {
        if (tid == 0) s_tbad = 0;
        kb = rkeys[lo];
        const uint64_t range = (uint64_t)(rkeys[hi - 1] - kb);
        while ((range >> ts) > 0xFFFFFFFFull) ts++;
        __syncthreads();
        const int m = (int)(hi - lo);
        for (int i = tid; i < m; i += kJThreads) c3(i) = (uint32_t)((uint64_t)(rkeys[lo + i] - kb) >> ts);
        __syncthreads();
        bool bad = false;
        for (int i = tid + 1; i < m; i += kJThreads) bad = bad || (c3(i) == c3(i - 1) && rkeys[lo + i] != rkeys[lo + i - 1]);
        if (bad) s_tbad = 1;
        __syncthreads();
        tmode = s_tbad == 0;
    }
    const uint32_t round_keys = tmode ? hi - lo : (uint32_t)chunk_cap;
    for (uint32_t base = lo; base < hi; base += round_keys) {
        const int m = (int)min(round_keys, hi - base);
        __syncthreads();                                             // the previous round's readers are done
        for (int i = tid; i < BM_WORDS; i += kJThreads) bitmap[i] = 0u;
        if (!tmode) for (int i = tid; i < m; i += kJThreads) ck(i) = rkeys[base + i];
        __syncthreads();
        for (int i = tid; i < m; i += kJThreads) { const uint32_t h = jhash(tmode ? rkeys[base + i] : ck(i)) >> (32 - BM_BITS); atomicOr(&bitmap[h >> 5], 1u << (h & 31u)); }
        __syncthreads();
        const K first = tmode ? kb : ck(0), last = tmode ? rkeys[hi - 1] : ck(m - 1);
        const bool has_prev = base > lo;
        const K prev_last = has_prev ? rkeys[base - 1] : (K)0;         // a run continued from the previous round was matched there
        // the index: keys as offsets from the round's first key, cut down to kJIdx classes
        const uint64_t span = tmode ? (uint64_t)c3(m - 1) : (uint64_t)(last - first);
        int S = 0;
        while ((span >> S) >= (uint64_t)kJIdx) S++;
        auto klass = [&](int i) -> int { return tmode ? (int)(c3(i) >> S) : (int)((uint64_t)(ck(i) - first) >> S); };
        for (int i = tid; i < m; i += kJThreads) {
            const int kc = klass(i), kp = i ? klass(i - 1) : -1;
            for (int k = kp + 1; k <= kc; k++) s_idx[k] = (uint16_t)i;
            if (i == m - 1) for (int k = kc + 1; k <= kJIdx; k++) s_idx[k] = (uint16_t)m;
        }
        __syncthreads();
        auto candidate = [&](K key) -> bool {                          // range test, then one bit
            if (!(key >= first && key <= last) || (has_prev && key == prev_last)) return false;
            const uint32_t h = jhash(key) >> (32 - BM_BITS);
            return (bitmap[h >> 5] >> (h & 31u)) & 1u;
        };
        int qn = 0;                                                    // wave-uniform: candidates queued
        auto enqueue = [&](bool c, K key, uint32_t row, uint32_t val) {
            const unsigned long long mask = __ballot(c);
            if (c) { const int at = qn + __popcll(mask & below); qkey[at] = key; qrow[at] = row; if (sizeof(E) == 16) qval[at] = val; }
            qn += __popcll(mask);
        };
        // a hit goes to the bin of its rank (a fixed range of the bucket's ranks: what the order kernel stages together): one
        // returning LDS atomic on the bin's cursor is its place (stores the compiler does not count: see st_hidden_b32)
        auto commit = [&](bool match, uint32_t pos, uint32_t row, uint32_t val, uint32_t klow) {
            if (match) {
                const uint32_t r = base + pos - lo;
                const uint32_t bin = bins.bin_of_group(r >> gs);
                uint32_t at = atomicAdd(&s_bincur[bin], 1u);
                size_t o = (size_t)bin * bincap + at;
                if (at >= bincap) { at = atomicAdd(&s_bincur[kMaxBins], 1u); o = (size_t)bins.ov_at + at; if (at < bins.ov_cap) at = 0u; else at = bincap; }   // the bin is full: the overflow area
                if (at < bincap) {
                    // one store per hit: a 16-byte record (three arrays of 8 + 4 + 4 bytes at first: a third of the kernel was its ~36
                    // short store segments per wave step, profiles/r04_notes.md 9; then a record + the rank alone for the order
                    // kernel's histogram sweep, which now keeps a sub-round's records in registers instead)
                    if (sizeof(E) == 16) st_hidden_b128(rout + o, uint4{base + pos, row, val, klow});
                    else st_hidden_b64(out + o, uint2{base + pos, row});
                    atomicAdd(&s_coarse[r >> gs], 1u);
                } else bin_full = true;                                // (cannot happen: the overflow area holds all of a bucket's probe pairs)
            }
        };
        // search `cnt` queued candidates (the last ones), one per lane: the lower bound in the index class's span
        auto search = [&](int cnt, K &key, uint32_t &row, uint32_t &val, int &pos) -> bool {
            qn -= cnt;
            const bool act = lane < cnt;
            key = act ? qkey[qn + lane] : (K)0;
            row = act ? qrow[qn + lane] : 0u;
            val = (sizeof(E) == 16 && act) ? qval[qn + lane] : 0u;
            if (tmode) {
                const uint32_t t = act ? (uint32_t)((uint64_t)(key - kb) >> ts) : 0u;
                const uint32_t kq = t >> S;                          // < kJIdx: the candidate passed the range test
                int n = act ? (int)s_idx[kq + 1] - (int)s_idx[kq] : 0;
                pos = (int)s_idx[kq];
                while (__ballot(n > 0) != 0ull) {
                    const int half = n >> 1;
                    const bool go = n > 0 && c3(pos + half) < t;
                    pos = go ? pos + half + 1 : pos;
                    n = go ? n - half - 1 : half;
                }
                return act && pos < m && c3(pos) == t;               // equal truncated keys are equal keys: the first of them is the lower bound
            }
            const uint32_t kq = act ? (uint32_t)((uint64_t)(key - first) >> S) : 0u;
            int n = act ? (int)s_idx[kq + 1] - (int)s_idx[kq] : 0;
            pos = (int)s_idx[kq];
            while (__ballot(n > 0) != 0ull) {
                const int half = n >> 1;
                const bool go = n > 0 && ck(pos + half) < key;
                pos = go ? pos + half + 1 : pos;
                n = go ? n - half - 1 : half;
            }
            return act && pos < m && ck(pos) == key;
        };
        auto drain = [&](int cnt) {                                    // search the last `cnt` queued candidates, commit the hits
            K key; uint32_t row, val; int pos = 0;
            const bool match = search(cnt, key, row, val, pos);
            commit(match, (uint32_t)pos, row, val, (uint32_t)key);
        };
        // A wave walks its slabs (every 16th of the bucket's) as ONE stream of 128-entry steps -- two entries per lane: one
        // 16-byte load for 8-byte pairs, two for 16-byte entries -- with the loads of the next two steps in flight across
        // slab boundaries (a slab of BASELINE configs[3] is eight steps long).  Every step issues its loads whether or not
        // there is anything left to read (a clamped address): see above.
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        constexpr bool WIDE = sizeof(E) == 16;
        const int nslab = wave < nwg ? (nwg - wave + nwaves - 1) / nwaves : 0;          // <= 64 (at most 1024 partition workgroups)
        const uint32_t mycount = lane < nslab ? min(counts[(size_t)b * nwg + wave + lane * nwaves], cap) : 0u;
        struct Cursor { int j; uint32_t step, count; };
        auto seek = [&](Cursor &c) {                                   // the first step at or after c that has entries
            while (c.j < nslab && c.step * 128u >= c.count) { c.j++; c.step = 0u; c.count = c.j < nslab ? (uint32_t)__shfl((int)mycount, c.j, 64) : 0u; }
        };
        auto fetch = [&](const Cursor &c, u4v &r0, u4v &r1) {
            const bool live = c.j < nslab;
            const u4v *src = reinterpret_cast<const u4v *>(slabs + ((size_t)b * nwg + wave % nwg + (size_t)(live ? c.j : 0) * nwaves) * cap);
            if (WIDE) {
                const uint32_t e0 = c.step * 128u + lane, e1 = e0 + 64u;
                r0 = __builtin_nontemporal_load(src + (live && e0 < c.count ? e0 : 0u));
                r1 = __builtin_nontemporal_load(src + (live && e1 < c.count ? e1 : 0u));
            } else {
                const uint32_t e0 = c.step * 128u + 2u * lane;
                r0 = __builtin_nontemporal_load(src + (live && e0 < c.count ? (e0 >> 1) : 0u));
            }
        };
        auto process = [&](const u4v x0, const u4v x1, const Cursor cur) {
            K key0, key1; uint32_t row0, row1, val0 = 0u, val1 = 0u; bool v0, v1;
            if (WIDE) {
                const uint32_t e0 = cur.step * 128u + lane;
                v0 = e0 < cur.count; v1 = e0 + 64u < cur.count;
                key0 = (K)(((uint64_t)x0.y << 32) | x0.x); row0 = x0.z; val0 = x0.w;
                key1 = (K)(((uint64_t)x1.y << 32) | x1.x); row1 = x1.z; val1 = x1.w;
            } else {
                const uint32_t e0 = cur.step * 128u + 2u * lane;
                v0 = e0 < cur.count; v1 = e0 + 1u < cur.count;
                key0 = (K)x0.x; row0 = x0.y; key1 = (K)x0.z; row1 = x0.w;
            }
            enqueue(v0 && candidate(key0), key0, row0, val0);        // qn < 64 before
            if (WIDE && qn >= 64) drain(64);                         // 16-byte entries: never more than 63 + 64 queued
            enqueue(v1 && candidate(key1), key1, row1, val1);        // 8-byte pairs: at most 64 + 128 = QCAP queued
            while (qn >= 64) drain(64);
        };
        // three register sets in turn, no copies between them: a move out of a register whose load is still in flight
        // would be a wait for it
        Cursor cA{0, 0u, nslab > 0 ? (uint32_t)__shfl((int)mycount, 0, 64) : 0u};
        seek(cA);
        Cursor cB = cA; cB.step++; seek(cB);
        Cursor cD = cB; cD.step++; seek(cD);
        Cursor nx = cD;
        u4v a0 = {0u, 0u, 0u, 0u}, a1 = a0, b0 = a0, b1 = a0, d0 = a0, d1 = a0;
        fetch(cA, a0, a1); fetch(cB, b0, b1); fetch(cD, d0, d1);
        for (;;) {
            if (cA.j >= nslab) break;
            process(a0, a1, cA); nx.step++; seek(nx); cA = nx; fetch(cA, a0, a1);
            if (cB.j >= nslab) break;
            process(b0, b1, cB); nx.step++; seek(nx); cB = nx; fetch(cB, b0, b1);
            if (cD.j >= nslab) break;
            process(d0, d1, cD); nx.step++; seek(nx); cD = nx; fetch(cD, d0, d1);
        }
        if (qn > 0) drain(qn);                                        // the round's leftovers (fewer than 64)
    }
    if (bin_full) *err = kJErrOverflow;
    __syncthreads();
    if (tid < kBinSlots) sbins[(size_t)b * kBinSlots + tid] = tid < kMaxBins ? min(s_bincur[tid], bincap) : min(s_bincur[kMaxBins], bins.ov_cap);
    if (tid == 0) { uint32_t t = min(s_bincur[kMaxBins], bins.ov_cap); for (int j = 0; j < bins.nb; j++) t += min(s_bincur[j], bincap); scount[b] = t; }
    for (int i = tid; i < kCoarse; i += kJThreads) scoarse[(size_t)b * kCoarse + i] = s_coarse[i];
}

__global__ __launch_bounds__(1024) void jsum_kernel(const uint32_t *__restrict__ scount, int P, unsigned long long *__restrict__ total,
                                                    JHotHead *__restrict__ hot, uint32_t *__restrict__ btotal, uint32_t nbatch, unsigned long long *__restrict__ mhot_out)
{
    __shared__ unsigned long long s_t;
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_t = 0ull;
    __syncthreads();
    unsigned long long x = tid < P ? scount[tid] : 0u;
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    if (lane == 0 && x) atomicAdd(&s_t, x);
    __syncthreads();
    if (tid == 0) *total = s_t;
    // ... and the hot rows (in the same launch: a kernel of its own cost 5 us of every join): btotal[batch] (rows of the batch that carry
    // a hot key with partners, counted by the partition kernel) -> exclusive prefix, the first place of every batch in the stream
    // of hot rows; the total -> *mhot_out (for the host)
    if (hot->Hp == 0u) { if (tid == 0) *mhot_out = 0ull; return; }
    const uint32_t per = (nbatch + 1023u) / 1024u, s0 = min(nbatch, (uint32_t)tid * per), s1 = min(nbatch, s0 + per);
    uint32_t sum = 0;
    for (uint32_t i = s0; i < s1; i++) sum += btotal[i];
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum, all = 0;
    for (int w = 0; w < 16; w++) { const uint32_t y = s_wave[w]; if (w < wave) run += y; all += y; }
    for (uint32_t i = s0; i < s1; i++) { const uint32_t c = btotal[i]; btotal[i] = run; run += c; }
    if (tid == 0) { hot->mhot = all; *mhot_out = all; }
}

// Survivor bins -> contiguous arrays (rank, left row, partner count) ORDERED BY (rank, left row): buckets are rank
// ranges in ascending order, so a workgroup orders its own bucket and writes behind the earlier buckets.
//   1. the bucket kernel's histogram of the survivors over groups of 2^gs ranks gives every group's place in the output;
//   2. sub-rounds: as many consecutive groups as fit the LDS stage (stage_cap survivors) and the fine counters (FINE
//      ranks), cut at bin boundaries where possible -- a sub-round reads the bins its rank range overlaps: whole bins, or
//      (a crowded bin) one bin several times;
//   3. inside a sub-round: one LDS counter per rank (histogram, exclusive scan, placement into the stage), then every
//      survivor finds its place inside its rank's run by counting the smaller row ids of the run, and the rows leave the
//      CU in output order.
// (Round 3 read a bucket's survivors as ONE unordered slab and binned them itself -- read, write, read twice: a third of
// the kernel's time and 35 % of its traffic; the bins now arrive from the bucket kernel.)
// A group of 2^gs ranks with more survivors than the stage holds, or a rank with more than kTieMax rows, raises *general:
// the bucket is copied as it is and the caller sorts all survivors with radix passes instead.
// CARRY: every survivor has a third word (a probe-side output column, sval / lval_out) that is ordered along with it; the
// stage then holds kStageCarry survivors (12 bytes each).
// VERIFY (64-bit keys): every survivor also has its probe key's low word (skey); at write-out -- ranks ascending: the sorted
// build keys are read front to back -- it is compared with the low word of the build key of the survivor's rank, which
// confirms the bucket kernel's hits on TRUNCATED keys (the truncation covers every bit above the dropped ones, so the two
// low words decide equality).  A mismatch sets general[2]: the host runs the join again without truncated rounds.
template <bool CARRY, bool VERIFY>
__global__ __launch_bounds__(kJThreads) void jorder_kernel(const uint2 *__restrict__ surv, size_t region, const uint32_t *__restrict__ scount,
                                                           const uint32_t *__restrict__ sbins, const uint32_t *__restrict__ counts, uint32_t cap, int nwg,
                                                           const uint32_t *__restrict__ bstart, int P, const uint32_t *__restrict__ runlen,
                                                           uint32_t *__restrict__ rank, uint32_t *__restrict__ lrow, uint32_t *__restrict__ cnt_out /* may be null */,
                                                           int stage_cap /* <= kStage / kStageCarry (tests: smaller) */, int32_t *__restrict__ general,
                                                           uint32_t *__restrict__ lval_out,
                                                           const uint32_t *__restrict__ scoarse,
                                                           const uint32_t *__restrict__ rranked /* may be null: a build-side column in rank order ... */,
                                                           uint32_t *__restrict__ rval_out /* ... read off for every survivor (unique build keys: survivor = output row) */,
                                                           const uint4 *__restrict__ srec /* VERIFY (64-bit keys): the survivors as records (rank, left row, carried word, probe-key low word); `surv` is not used then */,
                                                           const uint64_t *__restrict__ rkeys64 /* VERIFY: the sorted build keys */,
                                                           JHotHead *__restrict__ hot /* the hot keys' blocks: left free here, their first rows reported (pdst) */,
                                                           void *__restrict__ scratch_all, size_t scratch_stride /* per bucket: room for all of its survivors (the partition's slabs,
                                                                                                                  read for the last time by the bucket kernel) */,
                                                           int pieces_ok /* 0: a group of ranks with more survivors than the stage holds sends the bucket to the caller's radix
                                                                            sorts (cheap while the join has few matching rows); 1: it is taken rank by rank here */)
{
    constexpr int XW = (CARRY ? 1 : 0) + (VERIFY ? 1 : 0), STAGE = stage_of(XW), FINE = fine_of(XW), RPT = STAGE / kJThreads;
    static_assert(RPT * kJThreads == STAGE, "a sub-round's survivors are dealt RPT to a lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint2 *stage = reinterpret_cast<uint2 *>(lds_raw);                         // [STAGE]
    uint32_t *stv = reinterpret_cast<uint32_t *>(stage + STAGE);               // [STAGE] third words (CARRY)
    uint32_t *stk = stv + (CARRY ? STAGE : 0);                                 // [STAGE] probe-key low words (VERIFY)
    uint32_t *fine = stk + (VERIFY ? STAGE : 0);                               // [FINE + 1]
    uint32_t *coarse = fine + FINE + 1;                                        // [kCoarse + 1] counts, then exclusive prefix
    __shared__ unsigned long long s_dst;
    __shared__ uint32_t s_wave[kJThreads / 64];
    __shared__ uint32_t s_bincnt[kBinSlots];
    __shared__ uint32_t s_np;
    __shared__ int s_bad, s_big;
    __shared__ uint32_t s_h0, s_nh, s_nlong, s_anylong;
    __shared__ uint32_t s_hrank[kHotMax], s_hcum[kHotMax + 1];               // the bucket's hot ranks (ascending), the hot rows before each (inside the bucket)
    __shared__ uint2 s_long[kLongMax];                                       // runs of more than kTieMax rows in the stage (first, length)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_dst = 0ull; s_bad = 0; s_big = 0; s_np = 0u; s_h0 = 0u; s_nh = 0u; s_nlong = 0u; s_anylong = 0u; }
    if (tid < kBinSlots) s_bincnt[tid] = sbins[(size_t)b * kBinSlots + tid];
    __syncthreads();
    const uint32_t Hp = hot->Hp;                                           // hot keys with partners: dense and ascending, so a bucket's are a range [h0, h0 + nh)
    if ((uint32_t)tid < Hp) {
        const uint32_t hb = hot->pbucket[tid];
        if (hb < (uint32_t)b) atomicAdd(&s_h0, 1u); else if (hb == (uint32_t)b) atomicAdd(&s_nh, 1u);
    }
    unsigned long long part = 0;
    for (int qq = tid; qq < b; qq += kJThreads) part += scount[qq];
    for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
    if (lane == 0 && part) atomicAdd(&s_dst, part);
    {   // the bucket's probe pairs: the bucket kernel sized the bins from them
        uint32_t np = 0;
        for (int i = tid; i < nwg; i += kJThreads) np += min(counts[(size_t)b * nwg + i], cap);
        for (int d = 32; d > 0; d >>= 1) np += __shfl_down(np, d, 64);
        if (lane == 0 && np) atomicAdd(&s_np, np);
    }
    const uint32_t lo = bstart[b], len = bstart[b + 1] - lo, nb = scount[b];
    __syncthreads();
    const uint32_t h0 = s_h0, nh = s_nh;
    if ((uint32_t)tid < nh) s_hrank[tid] = hot->prank[h0 + tid];
    if (tid == 0) { uint32_t run = 0; for (uint32_t i = 0; i < nh; i++) { s_hcum[i] = run; run += (uint32_t)hot->prows[h0 + i]; } s_hcum[nh] = run; }
    const unsigned long long hot_before = Hp == 0u ? 0ull : h0 < Hp ? hot->pbefore[h0] : hot->mhot;
    // rows of the bucket's hot keys that come before rank r (their blocks lie between the rows written here)
    auto hot_shift = [&](uint32_t r) -> uint32_t {
        if (nh == 0u) return 0u;
        uint32_t a = 0, z = nh;                                            // the number of hot ranks < r
        while (a < z) { const uint32_t mid = (a + z) >> 1; if (s_hrank[mid] < r) a = mid + 1u; else z = mid; }
        return s_hcum[a];
    };
    const JBins bins = jbins_of(len, s_np, stage_cap, region);
    const int gs = bins.gs;                                                    // a bin = bins.gw consecutive groups
    const uint2 *src = VERIFY ? nullptr : surv + (size_t)b * region;
    const uint4 *srcr = VERIFY ? srec + (size_t)b * region : nullptr;
    bool contig = false;                                                   // the bucket's survivors were regrouped (below): src / srcr then hold them group after group
    static_assert(VERIFY || !CARRY, "a carried column travels in the 16-byte records of the 64-bit path");
    bool mismatch = false;
    const uint32_t ngroups = (len + (1u << gs) - 1u) >> gs;
    for (uint32_t i = tid; i <= ngroups; i += kJThreads) coarse[i] = i < ngroups ? scoarse[(size_t)b * kCoarse + i] : 0u;   // counted by the bucket kernel
    __syncthreads();
    const unsigned long long dst = s_dst + hot_before;
    if (nb == 0) {                                                     // nothing but (perhaps) hot blocks
        if ((uint32_t)tid < nh) hot->pdst[h0 + tid] = dst + s_hcum[tid];
        return;
    }
    // a pass over a bin's survivors (the pieces of a crowded bin: the usual sub-round keeps its survivors in registers, below)
    // keeps 8 loads per lane in flight: one workgroup owns the CU, and with a single load per lane the passes ran at the
    // latency of a load, not at the CU's share of the bandwidth.  f(entry (rank, left row), third word, probe-key low word)
    auto sweep = [&](size_t off, uint32_t i1, auto &&f) {
        // every lane issues its eight loads of a step before it uses any (clamped addresses past the end); with the full steps
        // peeled and a one-load-at-a-time tail (the first version) a bin of ~7 K survivors went through the tail, seven dependent
        // round trips per lane and sweep
        for (uint32_t i = tid; i < i1; i += 8u * kJThreads) {
            uint2 e[8]; uint32_t v[8], kl[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const size_t at = off + min(i + (uint32_t)k * kJThreads, i1 - 1u);
                if (!VERIFY) { e[k] = src[at]; v[k] = 0u; kl[k] = 0u; }
                else { const uint4 q = srcr[at]; e[k] = uint2{q.x, q.y}; v[k] = q.z; kl[k] = q.w; }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) if (i + (uint32_t)k * kJThreads < i1) f(e[k], v[k], kl[k]);
        }
    };
    // the bins that hold the ranks of groups [g0, g1)
    auto sweep_bins = [&](uint32_t g0, uint32_t g1, auto &&f) {
        if (contig) { sweep((size_t)coarse[g0], coarse[g1] - coarse[g0], f); return; }      // (coarse: the scanned counts, see below)
        for (uint32_t j = bins.bin_of_group(g0); j <= bins.bin_of_group(g1 - 1u); j++) sweep((size_t)j * bins.cap, s_bincnt[j], f);
        if (s_bincnt[kMaxBins]) sweep((size_t)bins.ov_at, s_bincnt[kMaxBins], f);      // the overflow area holds rows of any rank (f looks at the rank)
    };
    // block-wide exclusive scan helper over a[0, m): a contiguous segment per thread, waves chained through LDS;
    // returns the total
    auto scan_excl = [&](uint32_t *a, uint32_t m) -> uint32_t {
        const uint32_t per = (m + kJThreads - 1) / kJThreads, s0 = min(m, tid * per), s1 = min(m, s0 + per);
        uint32_t sum = 0;
        for (uint32_t i = s0; i < s1; i++) sum += a[i];
        uint32_t incl = sum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        __syncthreads();                                               // s_wave's previous readers are done
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum, total = 0;
        for (int w = 0; w < kJThreads / 64; w++) { const uint32_t x = s_wave[w]; if (w < wave) run += x; total += x; }
        for (uint32_t i = s0; i < s1; i++) { const uint32_t c = a[i]; a[i] = run; run += c; }
        __syncthreads();
        return total;
    };
    {   // a bucket of more than kCoarse * kFine sorted build entries (the cut never splits a run of equal keys, so one build
        // key repeated ~1.7e7 times gives one): a single group of 2^gs ranks would not fit the fine counters.  (A group with more
        // survivors than the stage holds is NOT a reason any more: it is taken rank by rank, below.)
        if ((1u << gs) > (uint32_t)FINE && tid == 0) s_bad = 1;
        bool big = false;
        for (uint32_t i = tid; i < ngroups; i += kJThreads) big = big || coarse[i] > (uint32_t)stage_cap;
        if (big) { s_big = 1; if (!pieces_ok) s_bad = 1; }
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    scan_excl(coarse, ngroups + 1);                                    // coarse[g] = survivors before group g; coarse[ngroups] = nb
    if (bad) {                                                         // bin after bin, as they are; the hot blocks behind them (the caller sorts everything)
        if ((uint32_t)tid < nh) hot->pdst[h0 + tid] = dst + nb + s_hcum[tid];
        uint32_t before = 0;                                          // the survivors of the bins before bin j (the overflow area last)
        for (uint32_t j = 0; j <= (uint32_t)bins.nb; j++) {
            const uint32_t cnt_j = j < (uint32_t)bins.nb ? s_bincnt[j] : s_bincnt[kMaxBins];
            const unsigned long long o = dst + before;
            before += cnt_j;
            for (uint32_t i = tid; i < cnt_j; i += kJThreads) {
                const size_t at = (j < (uint32_t)bins.nb ? (size_t)j * bins.cap : (size_t)bins.ov_at) + i;
                const uint4 q = VERIFY ? srcr[at] : uint4{src[at].x, src[at].y, 0u, 0u};
                const uint2 e = uint2{q.x, q.y};
                if (rank) { rank[o + i] = e.x; lrow[o + i] = e.y; }    // (the general sort needs them: the host runs the kernel again with the arrays if it left them out)
                if (cnt_out) cnt_out[o + i] = runlen[e.x];
                if (CARRY) lval_out[o + i] = q.z;                      // (the caller drops the carried words on the general path)
                if (rranked) rval_out[o + i] = rranked[e.x];
                if (VERIFY && (uint32_t)rkeys64[e.x] != q.w) mismatch = true;
            }
        }
        if (tid == 0) *general = 1;
        if (mismatch) general[2] = 1;
        return;
    }
    // ---- a bucket whose probe rows crowd a few ranks (survivors in the overflow area, which every sub-round would sweep, or a
    // group of ranks with more survivors than the stage holds, taken in pieces that each sweep its whole bin): ONE pass
    // brings the bucket's survivors into group order first -- the bucket kernel's counts per group, scanned, are the places --
    // in the bucket's slabs of the partition (read for the last time by the bucket kernel).  Every sub-round then reads its
    // groups' survivors as one contiguous range, a piece of a crowded group sweeps that group alone.
    if (s_big != 0 || s_bincnt[kMaxBins] != 0u) {
        for (uint32_t g = tid; g < ngroups; g += kJThreads) fine[g] = coarse[g];     // cursors (FINE >= kCoarse)
        __syncthreads();
        uint2 *sc2 = static_cast<uint2 *>(scratch_all) + (size_t)b * scratch_stride;
        uint4 *sc4 = static_cast<uint4 *>(scratch_all) + (size_t)b * scratch_stride;
        auto regroup_one = [&](uint2 e, uint32_t v, uint32_t kl) {
            const uint32_t at = atomicAdd(&fine[(e.x - lo) >> gs], 1u);
            if (!VERIFY) sc2[at] = e; else sc4[at] = uint4{e.x, e.y, v, kl};
        };
        for (uint32_t j = 0; j < (uint32_t)bins.nb; j++) sweep((size_t)j * bins.cap, s_bincnt[j], regroup_one);
        if (s_bincnt[kMaxBins]) sweep((size_t)bins.ov_at, s_bincnt[kMaxBins], regroup_one);
        __threadfence();
        __syncthreads();
        if (VERIFY) srcr = sc4; else src = sc2;
        contig = true;
    }
    const uint32_t max_groups = max(1u, (uint32_t)FINE >> gs);
    bool too_long = false;
    // the end of the sub-round that starts at group g0: the largest g1 with coarse[g1] - coarse[g0] <= stage_cap and
    // g1 - g0 <= max_groups (a single group always fits: checked above), moved back to a bin boundary when one lies
    // inside (a sub-round is then a run of whole bins, or a piece of one crowded bin)
    auto sub_end = [&](uint32_t g0) -> uint32_t {
        uint32_t a = g0 + 1, z = min(ngroups, g0 + max_groups);
        const uint32_t base_cnt = coarse[g0];
        while (a < z) { const uint32_t mid = (a + z + 1) >> 1; if (coarse[mid] - base_cnt <= (uint32_t)stage_cap) a = mid; else z = mid - 1; }
        if (a < ngroups) { const uint32_t snap = bins.bin_of_group(a) * bins.gw; if (snap > g0) a = snap; }
        return a;
    };
    uint32_t g0 = 0;
    while (g0 < ngroups) {
        const uint32_t base_cnt = coarse[g0];
        const uint32_t g1 = sub_end(g0), r0 = g0 << gs, r1 = min(len, g1 << gs), nr = r1 - r0, nsub = coarse[g1] - base_cnt;
        if (!nsub && (uint32_t)tid < nh) {                              // no rows here: a hot block of these ranks starts where the sub-round does
            const uint32_t rr = s_hrank[tid] - lo;
            if (rr >= r0 && rr < r1) hot->pdst[h0 + tid] = dst + base_cnt + s_hcum[tid];
        }
        if (nsub) {
            // the window of ranks being staged: [w_r0, w_r0 + w_nr) of the bucket, w_base survivors of the bucket before it, w_nsub in it
            uint32_t w_r0 = r0, w_nr = nr, w_base = base_cnt, w_nsub = nsub;
            // (the 65th row of a rank raises s_anylong: only then are the stage's runs looked over for long ones, below)
            auto count_one = [&](uint2 e, uint32_t, uint32_t) { const uint32_t r = e.x - lo - w_r0; if (r < w_nr && atomicAdd(&fine[r], 1u) == (uint32_t)kTieMax) s_anylong = 1u; };
            auto place_one = [&](uint2 e, uint32_t v, uint32_t kl) {
                const uint32_t r = e.x - lo - w_r0;
                if (r < w_nr) { const uint32_t at = atomicAdd(&fine[r], 1u); stage[at] = e; if (CARRY) stv[at] = v; if (VERIFY) stk[at] = kl; }   // afterwards fine[r] = end of rank r's rows
            };
            auto finish_window = [&]() {
            __syncthreads();
            if ((uint32_t)tid < nh) {                                   // a hot key's block starts behind the rows of the smaller ranks (it has none of its own here)
                const uint32_t rr = s_hrank[tid] - lo;
                if (rr >= w_r0 && rr < w_r0 + w_nr) hot->pdst[h0 + tid] = dst + w_base + (rr > w_r0 ? fine[rr - w_r0 - 1u] : 0u) + s_hcum[tid];
            }
            // ---- a rank with more than kTieMax rows (a key too rare for the sample, too frequent for the counting below): its
            // run of the stage is sorted by left row in place, one wave per run -- a bitonic network over the run padded to a power
            // of two (the padding is never touched: every exchange moves the smaller row to the lower index)
            const bool any_long = s_anylong != 0u;                      // (written during the count, two barriers ago)
            if (any_long) {
                for (uint32_t r = tid; r < w_nr; r += kJThreads) {
                    const uint32_t a0 = r ? fine[r - 1u] : 0u, a1 = fine[r];
                    if (a1 - a0 > (uint32_t)kTieMax) s_long[atomicAdd(&s_nlong, 1u)] = uint2{a0, a1 - a0};   // (at most STAGE / (kTieMax + 1) of them)
                }
                __syncthreads();
            }
            const uint32_t nlong = any_long ? s_nlong : 0u;
            auto exchange = [&](uint32_t a0, uint32_t L, uint32_t i, uint32_t j) {          // i < j
                if (j < L) {
                    const uint32_t x = stage[a0 + i].y, y = stage[a0 + j].y;
                    if (x > y) {
                        stage[a0 + i].y = y; stage[a0 + j].y = x;
                        if (CARRY) { const uint32_t t = stv[a0 + i]; stv[a0 + i] = stv[a0 + j]; stv[a0 + j] = t; }
                        if (VERIFY) { const uint32_t t = stk[a0 + i]; stk[a0 + i] = stk[a0 + j]; stk[a0 + j] = t; }
                    }
                }
            };
            // `width` lanes (a wave, or the whole workgroup for the runs a wave would take too long over) walk the network; `sync`
            // separates its steps
            auto bitonic = [&](uint32_t a0, uint32_t L, uint32_t me, uint32_t width, auto &&sync) {
                uint32_t p2 = 128u;
                while (p2 < L) p2 <<= 1;
                // (every stride is a power of two: shifts and masks -- with x / hk and x % hk on run-time values the network spent most of
                // its instructions dividing: 14 ns per element and run)
                for (uint32_t lk = 1u; (1u << lk) <= p2; lk++) {
                    const uint32_t lhk = lk - 1u, hmask = (1u << lhk) - 1u, kk = 1u << lk;
                    for (uint32_t x = me; x < (p2 >> 1); x += width) { const uint32_t blk = x >> lhk, off = x & hmask, base = blk << lk; exchange(a0, L, base + off, base + kk - 1u - off); }
                    sync();
                    for (uint32_t lj = lhk; lj-- > 0u;) {
                        const uint32_t jj = 1u << lj;
                        for (uint32_t x = me; x < (p2 >> 1); x += width) { const uint32_t i = ((x >> lj) << (lj + 1u)) | (x & (jj - 1u)); exchange(a0, L, i, i + jj); }
                        sync();
                    }
                }
            };
            for (uint32_t q = 0; q < nlong; q++)                        // (at most STAGE / kWaveSortMax of these)
                if (s_long[q].y > (uint32_t)kWaveSortMax) bitonic(s_long[q].x, s_long[q].y, (uint32_t)tid, (uint32_t)kJThreads, [] { __syncthreads(); });
            for (uint32_t q = wave; q < nlong; q += kJThreads / 64)
                if (s_long[q].y <= (uint32_t)kWaveSortMax) bitonic(s_long[q].x, s_long[q].y, (uint32_t)lane, 64u, [] { wave_lds_sync(); });
            if (any_long) { __syncthreads(); if (tid == 0) { s_nlong = 0u; s_anylong = 0u; } }
            // ---- rows of one rank into left-row order, and out: every survivor counts the rows of ITS rank (the stage's run
            // [fine[r-1], fine[r]), five or so) that are smaller than its own -- row ids are distinct, so that is its place in
            // the run -- and stores itself there.  (A first version sorted each run in registers with sorting networks, one
            // lane per rank: lanes of a wave held runs of 2..16 rows, so every wave ran the 4-, 8- and 16-input networks on
            // 64-bit words one after the other -- ~1400 vector instructions per wave and sub-round against ~250 here -- and
            // then wrote the stage out in a separate pass.)
            const unsigned long long o = dst + w_base;
            // four survivors per lane at a time: their LDS round trips and the reads through the rank overlap
            constexpr int kB = 4;
            for (uint32_t i0 = tid; i0 < w_nsub; i0 += kB * kJThreads) {
                uint2 e[kB]; uint32_t s0[kB], s1[kB], at[kB], v3[kB], kl[kB], rv[kB], rl[kB], rk[kB];
#pragma unroll
                for (int q = 0; q < kB; q++) {
                    const uint32_t i = min(i0 + (uint32_t)q * kJThreads, w_nsub - 1u);
                    e[q] = stage[i]; v3[q] = CARRY ? stv[i] : 0u; kl[q] = VERIFY ? stk[i] : 0u; at[q] = i;
                }
#pragma unroll
                for (int q = 0; q < kB; q++) {
                    const uint32_t r = e[q].x - lo - w_r0;
                    s0[q] = r ? fine[r - 1] : 0u; s1[q] = fine[r];
                    rv[q] = rranked ? rranked[e[q].x] : 0u;               // ranks ascend along the stage: an (almost) sequential read
                    rl[q] = cnt_out ? runlen[e[q].x] : 0u;
                    rk[q] = VERIFY ? reinterpret_cast<const uint32_t *>(rkeys64)[2u * (size_t)e[q].x] : 0u;   // the build key's low word (little endian), read in rank order
                }
#pragma unroll
                for (int q = 0; q < kB; q++) {
                    if (s1[q] - s0[q] > (uint32_t)kTieMax) {}                    // a long run: sorted above, the row is in its place
                    else if (s1[q] - s0[q] > 1u) {
                        uint32_t c = 0;
                        for (uint32_t j = s0[q]; j < s1[q]; j++) c += stage[j].y < e[q].y ? 1u : 0u;
                        at[q] = s0[q] + c;
                    }
                }
#pragma unroll
                for (int q = 0; q < kB; q++) {
                    if (i0 + (uint32_t)q * kJThreads < w_nsub) {
                        const unsigned long long w = o + at[q] + hot_shift(e[q].x);
                        if (VERIFY && rk[q] != kl[q]) mismatch = true;             // a hit on a truncated key that is none
                        if (rank) { rank[w] = e[q].x; lrow[w] = e[q].y; }           // null: every result column arrives through lval_out / rval_out
                        if (cnt_out) cnt_out[w] = rl[q];
                        if (CARRY) lval_out[w] = v3[q];
                        if (rranked) rval_out[w] = rv[q];
                    }
                }
            }
            __syncthreads();
            };
            if (nsub > (uint32_t)stage_cap) {
                // ---- ONE group of 2^gs ranks with more survivors than the stage holds (probe rows crowding a few neighbouring keys,
                // none of them frequent enough for the sample): taken in pieces of whole ranks.  Every piece counts the survivors
                // of the group's remaining ranks (a sweep of the group's bin and the overflow area), takes as many ranks as fit
                // the stage and places those (a second sweep).  A single rank that does not fit goes out as it is, and the caller's
                // radix sorts order everything (*general).
                uint32_t ra = 0, done = 0;
                while (ra < nr) {
                    w_r0 = r0 + ra; w_nr = nr - ra;
                    for (uint32_t i = tid; i <= w_nr; i += kJThreads) fine[i] = 0u;
                    __syncthreads();
                    sweep_bins(g0, g1, count_one);
                    __syncthreads();
                    scan_excl(fine, w_nr + 1u);                            // fine[r] = survivors of the ranks [w_r0, w_r0 + r)
                    uint32_t a = 0, z = w_nr;                              // the most ranks whose survivors fit
                    while (a < z) { const uint32_t mid = (a + z + 1u) >> 1; if (fine[mid] <= (uint32_t)stage_cap) a = mid; else z = mid - 1u; }
                    if (a == 0u) {                                         // rank w_r0 alone is too much: its rows as they come
                        const uint32_t cnt1 = fine[1];
                        __syncthreads();
                        if (tid == 0) fine[0] = 0u;
                        __syncthreads();
                        auto dump_one = [&](uint2 e, uint32_t v, uint32_t kl) {
                            if (e.x - lo != w_r0) return;
                            const unsigned long long w = dst + base_cnt + done + atomicAdd(&fine[0], 1u) + hot_shift(e.x);
                            if (rank) { rank[w] = e.x; lrow[w] = e.y; }
                            if (cnt_out) cnt_out[w] = runlen[e.x];
                            if (CARRY) lval_out[w] = v;
                            if (rranked) rval_out[w] = rranked[e.x];
                            if (VERIFY && (uint32_t)rkeys64[e.x] != kl) mismatch = true;
                        };
                        sweep_bins(g0, g1, dump_one);
                        too_long = true;
                        __syncthreads();
                        ra += 1u; done += cnt1;
                        continue;
                    }
                    w_nr = a; w_nsub = fine[a]; w_base = base_cnt + done;
                    __syncthreads();                                       // (every thread has read fine[a] before the placement moves the cursors)
                    if (w_nsub) { sweep_bins(g0, g1, place_one); finish_window(); }
                    else if ((uint32_t)tid < nh) { const uint32_t rr = s_hrank[tid] - lo; if (rr >= w_r0 && rr < w_r0 + w_nr) hot->pdst[h0 + tid] = dst + w_base + s_hcum[tid]; }
                    ra += a; done += w_nsub;
                }
                g0 = g1;
                continue;
            }
            // A sub-round of WHOLE bins (the usual case: bins are sized for the stage) holds at most STAGE = RPT x 1024 survivors:
            // every lane reads its RPT of them ONCE, keeps them in registers over the histogram, the scan and the placement.  (Two
            // sweeps over memory before -- the first over a separate array of the ranks alone, which the bucket kernel no longer
            // writes: 4 of 20 bytes per survivor.)  A piece of a crowded bin sweeps the whole bin twice and filters by rank.
            const uint32_t jA = bins.bin_of_group(g0), jB = bins.bin_of_group(g1 - 1u);
            uint32_t in_bins = 0;
            if (contig) in_bins = nsub;                                  // (regrouped: the sub-round's survivors are one contiguous range)
            else for (uint32_t j = jA; j <= jB; j++) in_bins += s_bincnt[j];
            const bool whole = contig ? true
                                      : (g0 == jA * bins.gw && (g1 == ngroups || g1 == (jB + 1u) * bins.gw) && in_bins <= (uint32_t)(RPT * kJThreads) && s_bincnt[kMaxBins] == 0u);
            if (whole) {
                uint2 e[RPT]; uint32_t v[RPT], kl[RPT];
#pragma unroll
                for (int k = 0; k < RPT; k++) {
                    uint32_t y = min(tid + (uint32_t)k * kJThreads, in_bins - 1u), j = jA;
                    if (!contig) while (y >= s_bincnt[j]) { y -= s_bincnt[j]; j++; }   // (in_bins >= nsub > 0: the walk ends inside [jA, jB])
                    const size_t at = contig ? (size_t)base_cnt + y : (size_t)j * bins.cap + y;
                    if (!VERIFY) { e[k] = src[at]; v[k] = 0u; kl[k] = 0u; }
                    else { const uint4 q = srcr[at]; e[k] = uint2{q.x, q.y}; v[k] = q.z; kl[k] = q.w; }
                }
                for (uint32_t i = tid; i <= nr; i += kJThreads) fine[i] = 0u;       // (under the loads)
                __syncthreads();
#pragma unroll
                for (int k = 0; k < RPT; k++) if (tid + (uint32_t)k * kJThreads < in_bins) count_one(e[k], 0u, 0u);
                __syncthreads();
                scan_excl(fine, nr);
#pragma unroll
                for (int k = 0; k < RPT; k++) if (tid + (uint32_t)k * kJThreads < in_bins) place_one(e[k], v[k], kl[k]);
            } else {
                for (uint32_t i = tid; i <= nr; i += kJThreads) fine[i] = 0u;
                __syncthreads();
                sweep_bins(g0, g1, count_one);
                __syncthreads();
                scan_excl(fine, nr);
                sweep_bins(g0, g1, place_one);
            }
            finish_window();
        }
        g0 = g1;
    }
    if (__ballot(too_long) != 0ull && lane == 0) *general = 1;
    if (VERIFY && __ballot(mismatch) != 0ull && lane == 0) general[2] = 1;
}

// survivor bins -> two contiguous arrays (rank, left row), unordered inside a bucket (the FULLSORT knob: radix sorts by the
// caller); dst offsets = exclusive scan of scount, done by every workgroup for itself (P <= 1024 values)
__global__ __launch_bounds__(256) void jcompact_kernel(const uint2 *__restrict__ surv, size_t region, const uint32_t *__restrict__ scount,
                                                       const uint32_t *__restrict__ sbins, const uint32_t *__restrict__ counts, uint32_t cap, int nwg,
                                                       const uint32_t *__restrict__ bstart, int stage_cap, uint32_t *__restrict__ rank, uint32_t *__restrict__ lrow,
                                                       const uint4 *__restrict__ srec /* 64-bit keys: the survivors as records (then `surv` is null) */, const uint64_t *__restrict__ rkeys64, int32_t *__restrict__ flags)
{
    __shared__ unsigned long long s_dst;
    __shared__ uint32_t s_np, s_off[kMaxBins + 2];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { s_dst = 0ull; s_np = 0u; }
    __syncthreads();
    unsigned long long part = 0;
    for (int q = threadIdx.x; q < b; q += blockDim.x) part += scount[q];
    for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
    if ((threadIdx.x & 63) == 0 && part) atomicAdd(&s_dst, part);
    uint32_t np = 0;
    for (int i = threadIdx.x; i < nwg; i += blockDim.x) np += min(counts[(size_t)b * nwg + i], cap);
    for (int d = 32; d > 0; d >>= 1) np += __shfl_down(np, d, 64);
    if ((threadIdx.x & 63) == 0 && np) atomicAdd(&s_np, np);
    __syncthreads();
    const JBins bins = jbins_of(bstart[b + 1] - bstart[b], s_np, stage_cap, region);
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int j = 0; j < bins.nb; j++) { s_off[j] = run; run += sbins[(size_t)b * kBinSlots + j]; }
        s_off[bins.nb] = run; s_off[bins.nb + 1] = run + sbins[(size_t)b * kBinSlots + kMaxBins];      // the overflow area behind the bins
    }
    __syncthreads();
    const unsigned long long dst = s_dst;
    for (int j = 0; j <= bins.nb; j++) {
        const size_t at0 = (size_t)b * region + (j < bins.nb ? (size_t)j * bins.cap : (size_t)bins.ov_at);
        const uint32_t cnt = s_off[j + 1] - s_off[j];
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const uint4 q = srec ? srec[at0 + i] : uint4{surv[at0 + i].x, surv[at0 + i].y, 0u, 0u};
            rank[dst + s_off[j] + i] = q.x; lrow[dst + s_off[j] + i] = q.y;
            if (srec && (uint32_t)rkeys64[q.x] != q.w) flags[2] = 1;       // a hit on a truncated key that is not one
        }
    }
}

// ---- the sample runs EARLY, on the context's second stream, while the build side is being sorted on the first -----------------
// One block, cleared by one launch: the survivor total and the error word (info, 64 B), the sample table (keys, counts), the batches'
// counts, the cells and the even buckets' counts, the set's head; the rest of the set behind.
struct JHotLayout {
    uint32_t S, cmin, tslots, nbatch, cells;
    size_t o_tkey, o_tcnt, o_btotal, o_hfine, o_hcoarse, clear_bytes;       // (the set lies at clear_bytes)
};
template <typename K> JHotLayout jhot_layout(int64_t n, int64_t s)
{
    JHotLayout L;
    L.S = (uint32_t)std::min<int64_t>(kHotSampleMax, std::max<int64_t>(4096, n / 256));
    L.cmin = 5;
    if (const char *e = getenv("HARK_JOIN_HOTMIN")) { const int c = atoi(e); if (c >= 2) L.cmin = (uint32_t)c; }   // tests: hot keys in small tables
    L.tslots = 1;
    while (L.tslots < 2u * L.S) L.tslots <<= 1;                               // (a workgroup's samples are counted in LDS first: at most S distinct keys arrive)
    L.nbatch = (uint32_t)((n + (int64_t)kJThreads * JTraits<K>::VEC - 1) / ((int64_t)kJThreads * JTraits<K>::VEC));      // the partition kernel's batches
    L.cells = 4096;                                                          // cells of the sorted build side the sampled rows are counted in (about one rank each, 2^20 at most)
    while (L.cells < (1u << 20) && (int64_t)L.cells < s) L.cells <<= 1;
    L.o_tkey = 64; L.o_tcnt = L.o_tkey + 8 * (size_t)L.tslots; L.o_btotal = L.o_tcnt + 4 * (size_t)L.tslots;
    L.o_hfine = L.o_btotal + (4 * (size_t)L.nbatch + 15) / 16 * 16; L.o_hcoarse = L.o_hfine + 4 * (size_t)L.cells;
    L.clear_bytes = L.o_hcoarse + 4 * (size_t)JTraits<K>::P + 4 * 64;
    return L;
}
static bool jhot_off() { return getenv("HARK_JOIN_NOHOT") != nullptr || getenv("HARK_JOIN_FULLSORT") != nullptr; }   // A/B + tests

template <typename K>
static int jhot_start(hark_context *ctx, hipStream_t st, unsigned char *block, const JHotLayout &L, const K *lcol, K bias, int64_t n)
{
    int rc = HARK_OK;
    const size_t n16 = (L.clear_bytes + sizeof(JHotHead) + 15) / 16;             // (the block is longer: the rest of the set lies behind the head)
    HARK_LAUNCH_RC(ctx, rc, jclear_kernel<<<dim3((unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)ctx->num_cu * 8)), 256, 0, st>>>(reinterpret_cast<uint4 *>(block), n16));
    if (!jhot_off())
        HARK_LAUNCH_RC(ctx, rc, jhot_sample_kernel<K><<<(L.S + 1023) / 1024, 1024, 0, st>>>(lcol, n, bias, L.S, reinterpret_cast<unsigned long long *>(block + L.o_tkey), reinterpret_cast<uint32_t *>(block + L.o_tcnt),
                                                                                          L.tslots - 1u, L.cmin, reinterpret_cast<JHotHead *>(block + L.clear_bytes)));
    return rc;
}

template <typename K>
int run_partitioned(hark_context *ctx, const K *lcol, K bias, int64_t n, const K *rkeys, int64_t s, const uint32_t *runlen,
                    int32_t *flags /* device: [0] general sort needed, [1] duplicate build keys */, const uint32_t *lval, const uint32_t *rranked,
                    uint32_t **rank_out, uint32_t **lrow_out, uint32_t **cnt_out, uint32_t **lval_out, uint32_t **rval_out, int64_t *m_out, bool *used, bool *dup_out,
                    bool rows_needed /* false: the caller takes its result columns from lval_out / rval_out alone (unique build keys) */, int64_t *general_out,
                    bool allow_trunc = true /* 64-bit keys: buckets of up to two chunks run as one round over truncated keys (hits confirmed by the order kernel) */)
{
    typedef typename JTraits<K>::E E;
    constexpr int P = JTraits<K>::P, Q = JTraits<K>::Q, VEC = JTraits<K>::VEC, LINE = 128 / (int)sizeof(E);
    *used = false;
    hipStream_t st = ctx->stream;
    const int nwg = ctx->num_cu;
    // slab capacity: twice the uniform share + 5 lines, a multiple of the line (1.5 x until round 6: a bucket that drew half as many
    // rows again as the others -- a hundred neighbouring keys with a thousand rows each -- sent the whole join to the sort-merge path)
    const int64_t avg = (n + (int64_t)P * nwg - 1) / ((int64_t)P * nwg);
    int64_t slabx = 200;                                                         // per cent of the uniform share
    // rotated loads (a probe column sorted block by block, see jpart_kernel's `rot`): a group's 64 (32) rows land in ONE bucket, so a
    // workgroup's slab of a bucket fills in a dozen chunks of 64 rows instead of row by row -- 3.8 sigma above the mean at twice the even
    // share, and the groups' places are a lattice, not chance: 1.25e8 i64 rows in blocks of 1e5 overran a slab there.  Four times it is.
    const int64_t nfullb_rot = n / ((int64_t)kJThreads * VEC);
    const int64_t rot = ctx->join_rotate && nfullb_rot >= 128 && !getenv("HARK_JOIN_NO_ROTATE") ? ((nfullb_rot / 64 - 1) | 1) : 0;
    if (rot > 0) slabx = 400;
    if (const char *e = getenv("HARK_JOIN_SLABX")) { const int c = atoi(e); if (c >= 100 && c <= 1000) slabx = c; }   // A/B
    int64_t cap64 = (avg * slabx / 100 + 5 * LINE + LINE - 1) / LINE * LINE;
    if (cap64 > 0x7FFFFFF0ll) return HARK_OK;
    const uint32_t cap = (uint32_t)cap64;
    K *splitters = nullptr; uint32_t *bstart = nullptr, *counts = nullptr, *scount = nullptr, *sbins = nullptr;
    int64_t *info = nullptr;                                                     // [0] survivors (u64), [1] error word of the partition
    E *slabs = nullptr; uint2 *surv = nullptr;
    uint32_t *rank = nullptr, *lrow = nullptr, *cnt = nullptr, *lv = nullptr, *rv = nullptr;
    uint4 *srec = nullptr;
    const bool carry = lval != nullptr && sizeof(E) == 16;                      // the fourth word of the 16-byte entries
    const bool verify = sizeof(K) == 8;                                          // survivors carry their probe key's low word to the order kernel
    if (getenv("HARK_JOIN_NOTRUNC")) allow_trunc = false;
    *lval_out = nullptr; *rval_out = nullptr;
    const size_t sstride = (size_t)nwg * cap;                                    // probe slabs of a bucket
    // survivors of a bucket: twice the room of all of its probe pairs -- one half dealt out equally to its bins, the other half
    // for the survivors that find their bin full (jbins_of)
    const size_t region = 2 * sstride;
    int rc = hark_alloc(ctx, (void **)&splitters, sizeof(K) * P);
    if (!rc) rc = hark_alloc(ctx, (void **)&bstart, 4 * (size_t)(P + 1));
    if (!rc) rc = hark_alloc(ctx, (void **)&counts, 4 * (size_t)P * nwg);
    if (!rc) rc = hark_alloc(ctx, (void **)&scount, 4 * (size_t)P);
    if (!rc) rc = hark_alloc(ctx, (void **)&sbins, 4 * (size_t)P * kBinSlots);
    if (!rc) rc = hark_alloc(ctx, (void **)&slabs, sizeof(E) * (size_t)P * sstride);
    if (!rc && !verify) rc = hark_alloc(ctx, (void **)&surv, 8 * (size_t)P * region);          // 32-bit keys: (rank, left row)
    if (!rc && verify) rc = hark_alloc(ctx, (void **)&srec, 16 * (size_t)P * region);          // 64-bit keys: records + their ranks alone
    uint32_t *scoarse = nullptr;
    if (!rc) rc = hark_alloc(ctx, (void **)&scoarse, 4 * (size_t)P * kCoarse);
    // the heavy hitters and the buckets' cut (see jhot_sample_kernel): the block may have been prepared -- cleared and sampled on the
    // context's second stream while the build side was sorted (k_join_hot_prepare) -- or is set up here
    JHotSet<K> *hot = nullptr;
    unsigned long long *tkey = nullptr;
    uint32_t *tcnt = nullptr, *btotal = nullptr, *hfine = nullptr, *hcoarse = nullptr;
    const bool no_hot = jhot_off();
    const JHotLayout L = jhot_layout<K>(n, s);
    const uint32_t S = L.S, cmin = L.cmin, tslots = L.tslots, nbatch = L.nbatch, cells = L.cells;
    (void)S;
    const size_t hot_clear = L.clear_bytes;
    unsigned char *hot_block = nullptr;
    bool prepared = false;
    if (ctx->join_prep && ctx->join_prep_col == static_cast<const void *>(lcol) && ctx->join_prep_n == n && ctx->join_prep_s == s) {
        hot_block = static_cast<unsigned char *>(ctx->join_prep); ctx->join_prep = nullptr; prepared = true;
    } else if (!rc) rc = hark_alloc(ctx, (void **)&hot_block, hot_clear + sizeof(JHotSet<K>));
    if (!rc) {
        info = reinterpret_cast<int64_t *>(hot_block);
        tkey = reinterpret_cast<unsigned long long *>(hot_block + L.o_tkey);
        tcnt = reinterpret_cast<uint32_t *>(hot_block + L.o_tcnt);
        btotal = reinterpret_cast<uint32_t *>(hot_block + L.o_btotal);
        hfine = reinterpret_cast<uint32_t *>(hot_block + L.o_hfine);
        hcoarse = reinterpret_cast<uint32_t *>(hot_block + L.o_hcoarse);
        hot = reinterpret_cast<JHotSet<K> *>(hot_block + hot_clear);
    }
    auto cleanup = [&]() {
        hark_free(ctx, splitters); hark_free(ctx, bstart); hark_free(ctx, counts); hark_free(ctx, scount); hark_free(ctx, sbins);
        hark_free(ctx, slabs); hark_free(ctx, surv); hark_free(ctx, srec); hark_free(ctx, scoarse);
        if (prepared && ctx->aux_event) (void)hipStreamWaitEvent(st, ctx->aux_event, 0);      // (its sample may still be running on the second stream: the block is reused in THIS stream's order)
        hark_free(ctx, hot_block);
    };
    if (rc == HARK_ENOMEM) {                                                     // no room for the partition workspace: the sort-merge path
        cleanup();                                                               // needs far less (used stays false)
        ctx->err.clear();
        return HARK_OK;
    }
    if (rc) { cleanup(); return rc; }
    int32_t *err = reinterpret_cast<int32_t *>(info + 1);
    unsigned long long *total = reinterpret_cast<unsigned long long *>(info);
    if (prepared) { if (hipStreamWaitEvent(st, ctx->aux_event, 0) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: waiting for the sample failed"); }
    else rc = jhot_start<K>(ctx, st, hot_block, L, lcol, bias, n);
    if (!rc && !no_hot)
        HARK_LAUNCH_RC(ctx, rc, jhot_coarse_kernel<K><<<dim3((unsigned)std::min<uint32_t>((tslots + 1023u) / 1024u, (uint32_t)ctx->num_cu)), 1024, 0, st>>>(tkey, tcnt, tslots, rkeys, s, hcoarse, P));
    // the hot keys of the sample and the buckets' cuts (even, or by the sampled rows' weight when they crowd a stretch of the build side)
    HARK_LAUNCH_RC(ctx, rc, jhot_select_kernel<K><<<1, 1024, 0, st>>>(tkey, tcnt, cmin, rkeys, s, hot, hcoarse, P, splitters, bstart, getenv("HARK_JOIN_EVEN_CUTS") ? 0 : 1, info));
    if (!no_hot) {                                                               // (both leave at once unless the cut is by weight)
        HARK_LAUNCH_RC(ctx, rc, jhot_fine_kernel<K><<<dim3((unsigned)std::min<uint32_t>((tslots + 255u) / 256u, (uint32_t)ctx->num_cu * 4u)), 256, 0, st>>>(tkey, tcnt, tslots, rkeys, s, hot, info, hfine, cells));
        HARK_LAUNCH_RC(ctx, rc, jhot_scan_cells_kernel<<<(cells + kCutChunk - 1) / kCutChunk, 1024, 0, st>>>(hfine, cells, hcoarse + P, info));   // (the chunks' totals behind the coarse counts)
        HARK_LAUNCH_RC(ctx, rc, jhot_cut_kernel<K><<<1, 1024, 0, st>>>(rkeys, s, hfine, cells, hcoarse + P, P, splitters, bstart, info));
    }
    if (rc) { cleanup(); return rc; }
    const size_t lds_part = sizeof(E) * (size_t)P * Q + sizeof(K) * (P + 2) + 8 * (size_t)P + 16 + (sizeof(K) + 2) * (size_t)kHotSlots;
    const bool plain_loads = getenv("HARK_JOIN_PLAIN_LOADS") != nullptr;        // tests: the compiler-counted twin of the partition kernel
    hipError_t he = plain_loads ? hipFuncSetAttribute(reinterpret_cast<const void *>(&jpart_kernel<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part)
                                : hipFuncSetAttribute(reinterpret_cast<const void *>(&jpart_kernel<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part);
    // rows per bucket and batch = BATCH / P; sweep before a ring of Q entries (less one line of carry) can fill up
    const int per_batch = kJThreads * VEC / P;
    int period = (Q - LINE) / (per_batch + per_batch / 2);
    if (period < 1) period = 1;
    int chunk_cap = JTraits<K>::CHUNK;
    if (const char *e = getenv("HARK_JOIN_CHUNK")) { const int c = atoi(e); if (c >= 1 && c < chunk_cap) chunk_cap = c; }   // tests: force several rounds per bucket
    constexpr size_t lds_bucket = sizeof(K) * (size_t)(JTraits<K>::CHUNK + (JTraits<K>::CHUNK >> (sizeof(K) == 8 ? 5 : 6))) + ((size_t)1 << (JTraits<K>::BM_BITS - 3))
                                + (sizeof(E) == 16 ? (size_t)(kJThreads / 64) * 128 * 16 : (size_t)(kJThreads / 64) * 192 * 8) + (size_t)kCoarse * 4
                                + (size_t)(kJIdx + 8) * 2;
    static_assert(lds_bucket <= kJLdsBudget, "bucket kernel LDS");
    if (he == hipSuccess) he = hipFuncSetAttribute(reinterpret_cast<const void *>(&jbucket_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bucket);
    const int xw = (carry ? 1 : 0) + (verify ? 1 : 0);                          // words a survivor carries beside (rank, left row)
    int stage_cap = stage_of(xw);
    if (const char *e = getenv("HARK_JOIN_STAGE")) { const int c = atoi(e); if (c >= 1 && c < stage_cap) stage_cap = c; }   // tests: many sub-rounds per bucket
    if (he == hipSuccess) {
        // one stream-ordered chain, one host read at the end: partition -> bucket probe -> survivor total
        // a probe column whose neighbouring rows share a bucket (the clustering test's third verdict): rotated loads
        ctx->last_join_rotated = rot > 0;
        if (rot > 0 && !plain_loads) {
            he = hipFuncSetAttribute(reinterpret_cast<const void *>(&jpart_kernel<K, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part);
            if (he == hipSuccess) HARK_LAUNCH_RC(ctx, rc, jpart_kernel<K, true, true><<<dim3((unsigned)nwg), dim3(kJThreads), lds_part, st>>>(lcol, n, bias, splitters, rkeys, s, slabs, counts, cap, period, err, carry ? lval : nullptr, hot, btotal, rot));
        } else if (rot > 0) {
            he = hipFuncSetAttribute(reinterpret_cast<const void *>(&jpart_kernel<K, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part);
            if (he == hipSuccess) HARK_LAUNCH_RC(ctx, rc, jpart_kernel<K, false, true><<<dim3((unsigned)nwg), dim3(kJThreads), lds_part, st>>>(lcol, n, bias, splitters, rkeys, s, slabs, counts, cap, period, err, carry ? lval : nullptr, hot, btotal, rot));
        } else
        if (plain_loads) HARK_LAUNCH_RC(ctx, rc, jpart_kernel<K, false><<<dim3((unsigned)nwg), dim3(kJThreads), lds_part, st>>>(lcol, n, bias, splitters, rkeys, s, slabs, counts, cap, period, err, carry ? lval : nullptr, hot, btotal));
        else HARK_LAUNCH_RC(ctx, rc, jpart_kernel<K, true><<<dim3((unsigned)nwg), dim3(kJThreads), lds_part, st>>>(lcol, n, bias, splitters, rkeys, s, slabs, counts, cap, period, err, carry ? lval : nullptr, hot, btotal));
        HARK_LAUNCH_RC(ctx, rc, jbucket_kernel<K><<<dim3((unsigned)P), dim3(kJThreads), lds_bucket, st>>>(slabs, counts, cap, nwg, rkeys, bstart, chunk_cap, surv, region, scount, sbins, srec, scoarse, stage_cap, err, allow_trunc ? 1 : 0));
        HARK_LAUNCH_RC(ctx, rc, jsum_kernel<<<1, 1024, 0, st>>>(scount, P, total, hot, btotal, nbatch, reinterpret_cast<unsigned long long *>(info + 3)));
        HIP_TRY_RC(ctx, rc, hipMemcpyAsync(info + 2, flags, 8, hipMemcpyDeviceToDevice, st));   // the duplicate-keys flag rides along with the same host read
    }
    if (he != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: setting the dynamic LDS size of the partition / bucket kernel failed: %s", hipGetErrorString(he));
    int64_t words[5] = {0, 0, 0, 0, 0}, M = 0;
    if (!rc) rc = hark_read_words(ctx, info, words, 5);
    ctx->last_join_weighted = words[4] != 0;
    const int64_t mhot = words[3];                                              // matching rows of the hot keys: blocks between the others' rows
    M = words[0] + mhot;
    if (!rc && (int32_t)(words[1] & 0xFFFFFFFFll) != 0) { cleanup(); ctx->last_join_overflow = true; return HARK_OK; }   // a slab or a survivor bin overflowed (skew): caller falls back
    const bool dup = ((words[2] >> 32) & 0xFFFFFFFFll) != 0;
    *dup_out = dup;
    *general_out = 0;
    if (!rc && M > 0) {
        if (dup || getenv("HARK_JOIN_FULLSORT")) rranked = nullptr;              // a survivor is an output row only when the build keys are unique
        // (rank, left row) of the output rows are 8 of the 16 bytes the order kernel writes per row: left out when nobody
        // reads them -- unique build keys and every result column carried (lval) or read off in rank order (rranked)
        bool skip_rows = !rows_needed && !dup && !getenv("HARK_JOIN_FULLSORT") && (lval == nullptr || carry) && !getenv("HARK_JOIN_ROWS");
        if (!rc && !skip_rows) rc = hark_alloc(ctx, (void **)&rank, 4 * (size_t)M);
        if (!rc && !skip_rows) rc = hark_alloc(ctx, (void **)&lrow, 4 * (size_t)M);
        if (!rc && dup) rc = hark_alloc(ctx, (void **)&cnt, 4 * (size_t)M);     // unique build keys: every survivor has exactly one partner
        if (!rc && carry) rc = hark_alloc(ctx, (void **)&lv, 4 * (size_t)M);
        if (!rc && rranked) rc = hark_alloc(ctx, (void **)&rv, 4 * (size_t)M);
        bool mismatch = false;
        // a crowded group of ranks: ordered in pieces by the order kernel, or -- while the whole join has few matching rows, whose two
        // radix sorts cost less than the pieces' sweeps (a hundred neighbouring keys with a thousand rows each: 1.5 against 2.7 ms) --
        // by the general path
        int pieces_ok = M > 8000000 ? 1 : 0;
        if (const char *e = getenv("HARK_JOIN_PIECES")) pieces_ok = atoi(e) != 0;     // tests: either way at any size
        uint32_t *hkey = nullptr, *hrow = nullptr;
        if (!rc && mhot > 0) {                                                  // the hot rows in row order (and every key's number of them: the order kernel needs it)
            rc = hark_alloc(ctx, (void **)&hkey, 4 * (size_t)mhot);
            if (!rc) rc = hark_alloc(ctx, (void **)&hrow, 4 * (size_t)mhot);
            HARK_LAUNCH_RC(ctx, rc, jhot_compact_kernel<K><<<(nbatch + 15) / 16, 1024, 0, st>>>(lcol, n, bias, hot, btotal, nbatch, hkey, hrow));
            HARK_LAUNCH_RC(ctx, rc, jhot_finish_kernel<<<1, kHotMax, 0, st>>>(hot, bstart, P));
        }
        for (int attempt = 0; attempt < 2 && !rc; attempt++) {
            const uint64_t *rk64 = reinterpret_cast<const uint64_t *>(rkeys);
            if (getenv("HARK_JOIN_FULLSORT")) {                                  // A/B + tests: plain compaction, radix sorts by the caller
                HARK_LAUNCH_RC(ctx, rc, jcompact_kernel<<<dim3((unsigned)P), 256, 0, st>>>(surv, region, scount, sbins, counts, cap, nwg, bstart, stage_cap, rank, lrow, srec, rk64, flags));
                HIP_TRY_RC(ctx, rc, hipMemsetAsync(flags, 1, 1, st));
            } else {
                const size_t lds_order = (size_t)stage_of(xw) * (8 + 4 * (size_t)xw) + (size_t)(fine_of(xw) + 1) * 4 + (size_t)(kCoarse + 1) * 4;
                auto launch = [&](auto carry_tag, auto verify_tag) -> hipError_t {
                    constexpr bool C = decltype(carry_tag)::value, V = decltype(verify_tag)::value;
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&jorder_kernel<C, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_order);
                    if (e != hipSuccess) return e;
                    jorder_kernel<C, V><<<dim3((unsigned)P), dim3(kJThreads), lds_order, st>>>(surv, region, scount, sbins, counts, cap, nwg, bstart, P, runlen,
                                                                                             rank, lrow, cnt, stage_cap, flags, C ? lv : nullptr, scoarse, rranked, rv,
                                                                                             V ? srec : nullptr, V ? rk64 : nullptr, hot, slabs, sstride, pieces_ok);
                    return hipGetLastError();
                };
                he = verify ? (carry ? launch(std::true_type{}, std::true_type{}) : launch(std::false_type{}, std::true_type{}))
                            : launch(std::false_type{}, std::false_type{});             // (a carried column exists on the 64-bit path only)
            }
            if (!rc && he != hipSuccess) rc = hark_launch_failed(ctx, he, "jorder_kernel<<<", __FILE__, __LINE__);
            // does the order stand, or do the survivors need the general sort (skew)?  Read here, while the survivors are alive:
            // an order kernel that left the rows out runs once more to deliver them.  [2]: a hit on a truncated key was none.
            int64_t fw[2] = {0, 0};
            if (!rc) rc = hark_read_words(ctx, flags, fw, 2);
            *general_out = fw[0] & 0xFFFFFFFFll;
            mismatch = (fw[1] & 0xFFFFFFFFll) != 0;
            if (rc || mismatch || !(*general_out && skip_rows)) break;
            skip_rows = false;
            rc = hark_alloc(ctx, (void **)&rank, 4 * (size_t)M);
            if (!rc) rc = hark_alloc(ctx, (void **)&lrow, 4 * (size_t)M);
        }
        if (!rc && !mismatch && mhot > 0) {                                     // the hot keys' rows into the blocks the order kernel left free
            uint32_t *skey = nullptr, *srow = nullptr;
            if (!rc) rc = k_sort_column(ctx, hkey, HARK_U32, mhot, false, hrow, &srow, &skey);       // by key, stable: one pass over a byte
            int64_t g = (mhot + 255) / 256;
            if (g > (int64_t)ctx->num_cu * 16) g = (int64_t)ctx->num_cu * 16;
            HARK_LAUNCH_RC(ctx, rc, jhot_place_kernel<<<dim3((unsigned)g), 256, 0, st>>>(hot, skey, srow, mhot, runlen, carry ? lval : nullptr, rranked, rank, lrow, cnt, lv, rv));
            hark_free(ctx, skey); hark_free(ctx, srow);                          // stream-ordered reuse
        }
        hark_free(ctx, hkey); hark_free(ctx, hrow);
        if (!rc && mismatch) {
            // a probe key shared its truncation with a build key it differs from (keys that cluster below the dropped bits): the
            // survivors hold rows that do not join.  Once more from the partition, with full keys in every round.
            cleanup();
            hark_free(ctx, rank); hark_free(ctx, lrow); hark_free(ctx, cnt); hark_free(ctx, lv); hark_free(ctx, rv);
            if (!allow_trunc) return hark_fail(ctx, HARK_EHIP, "join: key mismatch without truncated rounds");
            if (hipMemsetAsync(flags, 0, 4, st) != hipSuccess || hipMemsetAsync(flags + 2, 0, 4, st) != hipSuccess) return hark_fail(ctx, HARK_EHIP, "join: flag reset failed");
            return run_partitioned<K>(ctx, lcol, bias, n, rkeys, s, runlen, flags, lval, rranked, rank_out, lrow_out, cnt_out, lval_out, rval_out, m_out, used, dup_out,
                                      rows_needed, general_out, false);
        }
    }
    cleanup();                                                              // stream-ordered reuse: the ordering above is enqueued first
    if (rc) { hark_free(ctx, rank); hark_free(ctx, lrow); hark_free(ctx, cnt); hark_free(ctx, lv); hark_free(ctx, rv); return rc; }
    *rank_out = rank; *lrow_out = lrow; *cnt_out = cnt; *lval_out = lv; *rval_out = rv; *m_out = M; *used = true;
    return HARK_OK;
}

} // namespace

// Called by hark_entry_join BEFORE it sorts the build side: the probe keys' sample (and the clearing of the partition's small state)
// starts on the context's second stream and runs under the sort's kernels (35 + 6 us of every join otherwise).  Nothing happens
// when the partitioned path will not run.  k_join_hot_release drops a block nobody took.
int k_join_hot_prepare(hark_context *ctx, const void *lcol, bool k64, int64_t n, const void *rcol, int64_t s)
{
    if (ctx->join_prep) k_join_hot_release(ctx);
    if (n < ((int64_t)1 << 18) || s < 4096 || n + s > 0xFFFFFFFFll || getenv("HARK_JOIN_SORTMERGE") || getenv("HARK_JOIN_NO_EARLY_SAMPLE")) return HARK_OK;
    if (!ctx->aux_stream) {
        if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->main_event, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->aux_stream = nullptr; return HARK_OK; }   // (no second stream: the sample runs in line)
    }
    const JHotLayout L = k64 ? jhot_layout<uint64_t>(n, s) : jhot_layout<uint32_t>(n, s);
    unsigned char *block = nullptr;
    const size_t bytes = L.clear_bytes + (k64 ? sizeof(JHotSet<uint64_t>) : sizeof(JHotSet<uint32_t>));
    if (hark_alloc(ctx, (void **)&block, bytes) != HARK_OK) { ctx->err.clear(); return HARK_OK; }
    // the block comes from the pool in the FIRST stream's order (a block freed there may still be read by kernels queued there)
    int rc = HARK_OK;
    HIP_TRY_RC(ctx, rc, hipEventRecord(ctx->main_event, ctx->stream));
    HIP_TRY_RC(ctx, rc, hipStreamWaitEvent(ctx->aux_stream, ctx->main_event, 0));
    if (!rc) rc = k64 ? jhot_start<uint64_t>(ctx, ctx->aux_stream, block, L, static_cast<const uint64_t *>(lcol), 0x8000000000000000ull, n)
                      : jhot_start<uint32_t>(ctx, ctx->aux_stream, block, L, static_cast<const uint32_t *>(lcol), 0u, n);
    ctx->join_prep_tested = false;
    if (!rc && !getenv("HARK_JOIN_CLUSTERED")) { rc = k_cjoin_test(ctx, ctx->aux_stream, lcol, k64, n, rcol, s, false); ctx->join_prep_tested = rc == HARK_OK; }   // is the column clustered by key? (k_cjoin.hip)
    HIP_TRY_RC(ctx, rc, hipEventRecord(ctx->aux_event, ctx->aux_stream));
    ctx->join_prep = block; ctx->join_prep_col = lcol; ctx->join_prep_n = n; ctx->join_prep_s = s;
    if (rc) { k_join_hot_release(ctx); return rc; }
    return HARK_OK;
}

void k_join_hot_release(hark_context *ctx)
{
    if (!ctx->join_prep) return;
    if (ctx->aux_event) (void)hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0);
    hark_free(ctx, ctx->join_prep);
    ctx->join_prep = nullptr;
}

// Matching probe rows of the join as (global rank of the first equal sorted build entry, probe row id), sorted by
// (rank, probe row), and the partner count of each (*cnt_out == nullptr with *unique: the build keys are all distinct,
// every survivor has exactly one partner).  lcol: the probe key column (u32 bit patterns, or i64 when k64);
// rkeys: the SORTED build keys (u32, or u64 biased by 2^63 when k64).  *used = false: nothing was produced (tiny input,
// or skew overflowed a slab) and the caller takes the sort-merge path.  Outputs are pool blocks the caller frees.
// lval (optional, i64 keys only): a 4-byte probe-side column that travels with the probe rows; *lval_out then holds its
// value for every matching probe row, in the order of rank_out / lrow_out -- or stays null (32-bit keys, the FULLSORT knob,
// or the general ordering path was needed: the caller gathers the column through lrow_out instead).
// rranked (optional): a 4-byte build-side column in RANK order (column[perm[rank]]); with unique build keys *rval_out then
// holds its value for every matching probe row (= output row), read off by the order kernel; null otherwise.
// rows_needed = false: the caller reads neither *rank_out nor *lrow_out when the build keys turn out unique and *lval_out /
// *rval_out deliver its columns; both then stay null (the order kernel does not write them).
int k_join_partitioned(hark_context *ctx, const void *lcol, bool k64, int64_t n, const void *rkeys, int64_t s, const uint32_t *lval, const uint32_t *rranked,
                       uint32_t **rank_out, uint32_t **lrow_out, uint32_t **cnt_out, uint32_t **lval_out, uint32_t **rval_out, int64_t *m_out, bool *used, bool *unique,
                       bool rows_needed, int build_unique /* 1: the caller knows that all build keys are distinct (no run lengths are computed) */)
{
    *rank_out = nullptr; *lrow_out = nullptr; *cnt_out = nullptr; *lval_out = nullptr; *rval_out = nullptr; *m_out = 0; *used = false; *unique = false;
    if (getenv("HARK_JOIN_FULLSORT")) lval = nullptr;
    if (n < ((int64_t)1 << 18) || s < 4096 || n + s > 0xFFFFFFFFll) return HARK_OK;
    if (getenv("HARK_JOIN_SORTMERGE")) return HARK_OK;                          // A/B knob
    uint32_t *rank = nullptr, *lrow = nullptr, *cnt = nullptr, *runlen = nullptr, *lv = nullptr, *rv = nullptr;
    int64_t M = 0;
    int32_t *flag = nullptr;                                   // [0] the survivors need the general sort (skew), [1] duplicate build keys
    int rc = hark_alloc(ctx, (void **)&flag, 16);
    if (!rc && build_unique != 1) rc = hark_alloc(ctx, (void **)&runlen, 4 * (size_t)s);
    if (rc) { hark_free(ctx, flag); return rc; }
    HIP_TRY_RC(ctx, rc, hipMemsetAsync(flag, 0, 16, ctx->stream));
    if (build_unique != 1) {   // run lengths of the sorted build keys: the partner count of a survivor is runlen[rank]
        int64_t g1 = (s + 255) / 256;
        const int64_t gcap = (int64_t)ctx->num_cu * 16;
        if (g1 > gcap) g1 = gcap;
        if (k64) HARK_LAUNCH_RC(ctx, rc, jrunlen_kernel<uint64_t><<<dim3((unsigned)g1), 256, 0, ctx->stream>>>(static_cast<const uint64_t *>(rkeys), s, runlen, flag + 1));
        else HARK_LAUNCH_RC(ctx, rc, jrunlen_kernel<uint32_t><<<dim3((unsigned)g1), 256, 0, ctx->stream>>>(static_cast<const uint32_t *>(rkeys), s, runlen, flag + 1));
    }
    if (rc) { hark_free(ctx, flag); hark_free(ctx, runlen); return rc; }
    bool dup = true;
    int64_t general = 0;                                       // the order kernel's verdict (read in run_partitioned, while it can still run again)
    // A probe column sorted or clustered by the key would send every batch of the partition into one ring (k_cjoin.hip): it is
    // searched in row order instead.  The test ran behind the early sample on the second stream (long done: the build side
    // has been sorted since), or runs here.
    int verdict = 0;                                           // 1: the search path, 2: the partition with rotated loads (k_cjoin.hip, cj_test_kernel)
    if (const char *e = getenv("HARK_JOIN_CLUSTERED")) verdict = atoi(e);                        // tests, A/B: 0 / 1 / 2 whatever the rows look like
    else if (ctx->join_prep && ctx->join_prep_tested && ctx->join_prep_col == lcol && ctx->join_prep_n == n) {
        if (hipEventSynchronize(ctx->aux_event) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: waiting for the clustering test failed");
        else verdict = k_cjoin_verdict(ctx);
    } else {
        rc = k_cjoin_test(ctx, ctx->stream, lcol, k64, n, rkeys, s, true);
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "join: the clustering test failed");
        if (!rc) verdict = k_cjoin_verdict(ctx);
    }
    if (rc) { hark_free(ctx, flag); hark_free(ctx, runlen); return rc; }
    const bool clustered = verdict == 1;
    ctx->join_rotate = verdict == 2;
    ctx->last_join_rotated = false;
    ctx->last_join_overflow = false;
    ctx->last_join_clustered = clustered;
    if (clustered) rc = k_cjoin_run(ctx, lcol, k64, n, rkeys, s, runlen, flag, k64 ? lval : nullptr, rranked, &rank, &lrow, &cnt, &lv, &rv, &M, used, &dup, rows_needed, &general);
    else
        rc = k64 ? run_partitioned<uint64_t>(ctx, static_cast<const uint64_t *>(lcol), 0x8000000000000000ull, n, static_cast<const uint64_t *>(rkeys), s, runlen, flag, lval, rranked, &rank, &lrow, &cnt, &lv, &rv, &M, used, &dup, rows_needed, &general)
                 : run_partitioned<uint32_t>(ctx, static_cast<const uint32_t *>(lcol), 0u, n, static_cast<const uint32_t *>(rkeys), s, runlen, flag, nullptr, rranked, &rank, &lrow, &cnt, &lv, &rv, &M, used, &dup, rows_needed, &general);
    // rotated loads on a column of narrow sorted runs overrun the slabs (a group's 64 rows land in ONE bucket, and the groups fall on the
    // buckets by chance): the search path is the better way out than sorting the probe side (2.6 against 4.2 ms per 1e8 rows)
    if (!rc && !*used && verdict == 2 && ctx->last_join_overflow) {
        ctx->last_join_clustered = true; ctx->last_join_rotated = false; ctx->last_join_weighted = false;
        HIP_TRY_RC(ctx, rc, hipMemsetAsync(flag, 0, 4, ctx->stream));
        if (!rc) rc = k_cjoin_run(ctx, lcol, k64, n, rkeys, s, runlen, flag, k64 ? lval : nullptr, rranked, &rank, &lrow, &cnt, &lv, &rv, &M, used, &dup, rows_needed, &general);
    }
    if (rc || !*used) { hark_free(ctx, flag); hark_free(ctx, runlen); return rc; }
    if (M == 0) { hark_free(ctx, rank); hark_free(ctx, lrow); hark_free(ctx, cnt); hark_free(ctx, lv); hark_free(ctx, rv); hark_free(ctx, flag); hark_free(ctx, runlen); return HARK_OK; }
    // (rank, left row) order.  Fast path: jorder_kernel delivered it.  A rank with more than kTieMax probe rows, or a
    // group of ranks too crowded for the LDS stage, takes the general path: stable radix sort by left row, then by rank
    // (each skips the passes no byte needs), partner counts looked up again.
    uint32_t *rank1 = nullptr, *lrow1 = nullptr, *lrow2 = nullptr, *rank2 = nullptr;
    if (!rc && general) {
        hark_free(ctx, lv); lv = nullptr;                       // the radix sorts carry one payload: the columns are gathered by the caller instead
        hark_free(ctx, rv); rv = nullptr;
        rc = k_sort_column(ctx, lrow, HARK_U32, M, false, rank, &rank1, &lrow1);
        if (!rc) rc = k_sort_column(ctx, rank1, HARK_U32, M, false, lrow1, &lrow2, &rank2);
        hark_free(ctx, rank1); hark_free(ctx, lrow1);
        if (!rc && cnt) {
            int64_t g2 = (M + 255) / 256;
            const int64_t gcap = (int64_t)ctx->num_cu * 16;
            if (g2 > gcap) g2 = gcap;
            HARK_LAUNCH_RC(ctx, rc, jcnt_kernel<<<dim3((unsigned)g2), 256, 0, ctx->stream>>>(rank2, M, runlen, cnt));
        }
    } else if (!rc) {
        rank2 = rank; lrow2 = lrow; rank = lrow = nullptr;     // in place
    }
    hark_free(ctx, flag);
    hark_free(ctx, rank); hark_free(ctx, lrow);
    hark_free(ctx, runlen);
    if (rc) { hark_free(ctx, lrow2); hark_free(ctx, rank2); hark_free(ctx, cnt); hark_free(ctx, lv); hark_free(ctx, rv); *used = false; return rc; }
    *rank_out = rank2; *lrow_out = lrow2; *cnt_out = cnt; *lval_out = lv; *rval_out = rv; *m_out = M; *unique = !dup;
    return HARK_OK;
}
