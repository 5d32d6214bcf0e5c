// widedigit.hip -- what would a radix pass over digits WIDER than 8 bits cost on gfx950?  
//
// Every scatter of k_sort.hip runs at the plain-stream floor of the bytes it moves (profiles/r04_floor_table.md), so only
// FEWER passes can make ORDER BY faster: 20-bit keys in 2 passes of 10 bits instead of 3 of 7, 31/32-bit keys in 3 passes of
// 11 instead of 4 of 8.  A wider digit means more, shorter digit runs per tile -- and profiles/r04_notes.md 5 found that
// the scattered runs, not the ranking, are what a pass pays for (305 us with sequential stores, 479 us with 192-byte runs).
// This probe measures exactly that trade without building the stable ranking of a wide digit (which does not fit the LDS:
// the per-wave digit tables of digit_scatter2_kernel are 16 waves x bins x 12 B = 384 KiB at 2048 bins):
//
//   one pass = the product's three launches (per-slice histogram, scan, scatter); the scatter kernel ranks a key inside its
//   tile with ONE returning LDS atomic on its digit's counter (unstable across waves -- the memory pattern, the LDS
//   footprint and the instruction count are those of a pass or better), stages the tile digit-sorted in LDS and writes the
//   digit runs to their exact output positions, with (key, payload) either as two arrays (what the product moves) or as ONE
//   array of 8-byte pairs (runs twice as long in bytes: DESIGN.md 7.4).
//
// Knobs swept: digit width 6..11 bits, tile 12288 or 16384 keys per 1024-thread workgroup, split / pair output, and a run
// with SEQUENTIAL stores (the floor).  Output: ms per pass, and the ORDER BY totals they imply.
// Build: make -C tools widedigit.  Run on the GPU box: tools/widedigit [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int T = 1024;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ void gen_kernel(uint32_t *keys, uint32_t *vals, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = (uint64_t)i + 0x9E3779B97F4A7C15ull;
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull; h = (h ^ (h >> 27)) * 0x94D049BB133111EBull; h ^= h >> 31;
        keys[i] = (uint32_t)h; vals[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(T) void hist_kernel(const uint32_t *__restrict__ keys, int64_t n, int64_t slice, int shift, uint32_t bins, uint32_t *__restrict__ hist, int nblk)
{
    extern __shared__ uint32_t s_hist[];
    for (uint32_t i = threadIdx.x; i < bins; i += T) s_hist[i] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += T) atomicAdd(&s_hist[(keys[i] >> shift) & (bins - 1u)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += T) hist[(size_t)i * nblk + blockIdx.x] = s_hist[i];
}

// exclusive scan of hist in (bin, blk) order: one workgroup, sequential carries (the probe does not time it)
__global__ __launch_bounds__(T) void scan_kernel(uint32_t *__restrict__ hist, int64_t total)
{
    __shared__ uint32_t s_part[T];
    const int64_t per = (total + T - 1) / T;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < total ? lo + per : total;
    uint32_t sum = 0;
    for (int64_t i = lo; i < hi; i++) sum += hist[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < T; i++) { const uint32_t x = s_part[i]; s_part[i] = run; run += x; } }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (int64_t i = lo; i < hi; i++) { const uint32_t x = hist[i]; hist[i] = run; run += x; }
}

// MODE 0: two output arrays; 1: one array of (key, payload) pairs; 2: sequential stores (floor: no scatter at all)
template <int R, int MODE>
__global__ __launch_bounds__(T) void scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint2 *__restrict__ pairs_out,
                                                    int64_t n, int64_t slice, int shift, uint32_t bins, const uint32_t *__restrict__ hist, int nblk)
{
    constexpr int TILE = T * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint2 *s_kv = reinterpret_cast<uint2 *>(lds);                       // [TILE]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_kv + TILE);        // [bins] keys of the digit in this tile
    uint32_t *s_start = s_cnt + bins;                                   // [bins] first slot of the digit in the tile
    uint32_t *s_gpos = s_start + bins;                                  // [bins] output position of the digit's next key
    __shared__ uint32_t s_wsum[T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t dmask = bins - 1u;
    const int per = (int)((bins + T - 1) / T);                          // digits per thread in the tile scan (1 or 2)
    for (uint32_t d = tid; d < bins; d += T) { s_cnt[d] = 0u; s_gpos[d] = hist[(size_t)d * nblk + blockIdx.x]; }
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    uint32_t nkey[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int64_t i = lo + (int64_t)wave * (64 * R) + r * 64 + lane; nkey[r] = keys_in[i < hi ? i : hi - 1]; }
    for (int64_t tbase = lo; tbase < hi; tbase += TILE) {
        const int64_t wbase = tbase + (int64_t)wave * (64 * R);
        uint32_t key[R], val[R], rank[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            key[r] = nkey[r];
            const bool valid = wbase + r * 64 + lane < hi;
            rank[r] = valid ? atomicAdd(&s_cnt[(key[r] >> shift) & dmask], 1u) : 0xFFFFFFFFu;        // ds_add_rtn_u32
        }
#pragma unroll
        for (int r = 0; r < R; r++) { const int64_t i = wbase + r * 64 + lane; val[r] = __builtin_nontemporal_load(vals_in + (i < hi ? i : hi - 1)); }
#pragma unroll
        for (int r = 0; r < R; r++) { const int64_t i = wbase + TILE + r * 64 + lane; nkey[r] = __builtin_nontemporal_load(keys_in + (i < hi ? i : hi - 1)); }
        lds_barrier();
        // exclusive scan of the tile's digit counts (bins <= 2 * T)
        uint32_t c[2] = {0u, 0u}, tsum = 0;
        for (int j = 0; j < per; j++) { const uint32_t d = (uint32_t)tid * per + j; c[j] = d < bins ? s_cnt[d] : 0u; tsum += c[j]; }
        uint32_t incl = tsum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        if (lane == 63) s_wsum[wave] = incl;
        lds_barrier();
        uint32_t run = incl - tsum;
        for (int w = 0; w < wave; w++) run += s_wsum[w];
        for (int j = 0; j < per; j++) {
            const uint32_t d = (uint32_t)tid * per + j;
            if (d < bins) { s_start[d] = run; const uint32_t gp = s_gpos[d]; s_cnt[d] = gp - run; s_gpos[d] = gp + c[j]; run += c[j]; }      // s_cnt now holds delta = output position - slot
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R; r++) if (rank[r] != 0xFFFFFFFFu) s_kv[s_start[(key[r] >> shift) & dmask] + rank[r]] = uint2{key[r], val[r]};
        lds_barrier();
        const int tile_n = (int)((hi - tbase) < TILE ? (hi - tbase) : TILE);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int slot = tid + r * T;
            if (slot < tile_n) {
                const uint2 kv = s_kv[slot];
                const uint32_t pos = MODE == 2 ? (uint32_t)(tbase + slot) : (uint32_t)slot + s_cnt[(kv.x >> shift) & dmask];
                if (MODE == 1) __builtin_nontemporal_store(*reinterpret_cast<const unsigned long long *>(&kv), reinterpret_cast<unsigned long long *>(pairs_out + pos));
                else { keys_out[pos] = kv.x; vals_out[pos] = kv.y; }
            }
        }
        lds_barrier();
        for (uint32_t d = tid; d < bins; d += T) s_cnt[d] = 0u;
        lds_barrier();
    }
}

struct Bufs { uint32_t *k_in, *v_in, *k_out, *v_out, *hist; uint2 *p_out; };

template <int R, int MODE>
static float one_pass(const Bufs &b, int64_t n, int shift, int bits, int ncu, bool verify)
{
    const uint32_t bins = 1u << bits;
    const int64_t tile = (int64_t)T * R;
    int64_t nblk = ncu, slice = ((n + nblk - 1) / nblk + tile - 1) / tile * tile;
    nblk = (n + slice - 1) / slice;
    const size_t lds = (size_t)tile * 8 + (size_t)bins * 12;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&scatter_kernel<R, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hist_kernel<<<(int)nblk, T, bins * 4>>>(b.k_in, n, slice, shift, bins, b.hist, (int)nblk);
    scan_kernel<<<1, T>>>(b.hist, (int64_t)bins * nblk);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0));
        scatter_kernel<R, MODE><<<(int)nblk, T, lds>>>(b.k_in, b.v_in, b.k_out, b.v_out, b.p_out, n, slice, shift, bins, b.hist, (int)nblk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    if (verify && MODE != 2) {                                   // digit-sorted, and a permutation of the input (payloads = row ids, each once)
        std::vector<uint32_t> hk(n), hv(n);
        if (MODE == 1) {
            std::vector<uint2> hp(n);
            CK(hipMemcpy(hp.data(), b.p_out, n * 8, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; i++) { hk[i] = hp[i].x; hv[i] = hp[i].y; }
        } else { CK(hipMemcpy(hk.data(), b.k_out, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv.data(), b.v_out, n * 4, hipMemcpyDeviceToHost)); }
        std::vector<uint32_t> hin(n);
        CK(hipMemcpy(hin.data(), b.k_in, n * 4, hipMemcpyDeviceToHost));
        bool ok = true; std::vector<uint8_t> seen(n, 0);
        for (int64_t i = 0; i < n && ok; i++) {
            if (i && ((hk[i - 1] >> shift) & (bins - 1)) > ((hk[i] >> shift) & (bins - 1))) ok = false;
            if (hv[i] >= n || seen[hv[i]] || hin[hv[i]] != hk[i]) ok = false; else seen[hv[i]] = 1;
        }
        printf("    verify bits=%d R=%d mode=%d: %s\n", bits, R, MODE, ok ? "digit-sorted permutation" : "WRONG");
        if (!ok) exit(1);
    }
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 100000000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    Bufs b;
    CK(hipMalloc(&b.k_in, n * 4)); CK(hipMalloc(&b.v_in, n * 4)); CK(hipMalloc(&b.k_out, n * 4)); CK(hipMalloc(&b.v_out, n * 4));
    CK(hipMalloc(&b.p_out, n * 8)); CK(hipMalloc(&b.hist, (size_t)2048 * 1024 * 4));
    gen_kernel<<<2048, 256>>>(b.k_in, b.v_in, n);
    CK(hipDeviceSynchronize());
    printf("widedigit: %lld (key, payload) pairs, %d CUs, one 1024-thread workgroup per CU; ms per scatter pass (median of 5)\n", (long long)n, ncu);
    printf("%5s %9s | %10s %10s | %10s %10s | %10s\n", "bits", "runs/tile", "split 12K", "pairs 12K", "split 16K", "pairs 16K", "seq 12K");
    float t12s[12] = {0}, t12p[12] = {0}, t16s[12] = {0}, t16p[12] = {0};
    bool first = true;
    for (int bits = 6; bits <= 11; bits++) {
        const int shift = 4;
        t12s[bits] = one_pass<12, 0>(b, n, shift, bits, ncu, first || bits == 11);
        t12p[bits] = one_pass<12, 1>(b, n, shift, bits, ncu, first || bits == 11);
        t16s[bits] = one_pass<16, 0>(b, n, shift, bits, ncu, false);
        t16p[bits] = one_pass<16, 1>(b, n, shift, bits, ncu, false);
        const float seq = one_pass<12, 2>(b, n, shift, bits, ncu, false);
        first = false;
        printf("%5d %9d | %10.3f %10.3f | %10.3f %10.3f | %10.3f\n", bits, 1 << bits, t12s[bits], t12p[bits], t16s[bits], t16p[bits], seq);
    }
    auto best = [&](int bits) { return std::min(std::min(t12s[bits], t12p[bits]), std::min(t16s[bits], t16p[bits])); };
    printf("ORDER BY totals implied (scatter passes only; every pass also pays a histogram read of 0.07-0.09 ms):\n");
    printf("  20-bit keys: 3 x 7 bits  %.3f ms (split 12K: the product's geometry) | 2 x 10 bits %.3f ms (best format)\n", 3 * t12s[7], 2 * best(10));
    printf("  31-bit keys: 4 x 8 bits  %.3f ms | 3 x 11/10/10 bits %.3f ms (best format)\n", 4 * t12s[8], best(11) + 2 * best(10));
    printf("  i64 high word (32 bits of 16-byte tuples not modelled): same ratio as 31-bit keys\n");
    return 0;
}
