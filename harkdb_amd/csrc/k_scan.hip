// k_scan.hip -- device-wide exclusive prefix sum of u32 counts (to u32 or i64).
//
// The reference leans on Futhark's `scan (+)` everywhere (groupby.fut:12-13,
// segmented.fut:28, join.fut:61); here one three-launch scan serves head-flag
// numbering, join output offsets and compaction offsets:
//   tile sums (4096 elements per workgroup) -> single-workgroup scan of the
//   sums -> per-tile rescan that adds the tile's offset.
#include "hark_internal.h"

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanPer = 16;
constexpr int kScanTile = kScanThreads * kScanPer;

__device__ __forceinline__ unsigned long long block_exclusive(unsigned long long x, unsigned long long *s_wave, unsigned long long *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long incl = x;
    for (int d = 1; d < 64; d <<= 1) { unsigned long long y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long carry = 0, tot = 0;
    for (int w = 0; w < nw; w++) { if (w < wave) carry += s_wave[w]; tot += s_wave[w]; }
    if (total) *total = tot;
    __syncthreads();
    return carry + incl - x;
}

// a thread's kScanPer consecutive counts: four 16-byte loads where the rows exist and the column is aligned (one 4-byte load
// per count touched 64 lines per wave instruction: the join's 1.09e7-row scan ran at 1.5-2 TB/s), zeros past the end
__device__ __forceinline__ void load16(const uint32_t *__restrict__ in, int64_t base, int64_t n, uint32_t (&v)[kScanPer])
{
    if (base + kScanPer <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
#pragma unroll
        for (int j = 0; j < kScanPer; j += 4) { const uint4 q = *reinterpret_cast<const uint4 *>(in + base + j); v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w; }
    } else {
#pragma unroll
        for (int j = 0; j < kScanPer; j++) v[j] = base + j < n ? in[base + j] : 0u;
    }
}

__global__ __launch_bounds__(kScanThreads) void tile_sums_kernel(const uint32_t *__restrict__ in, int64_t n, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long s_wave[kScanThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    load16(in, base, n, v);
    unsigned long long x = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j++) x += v[j];
    unsigned long long tot;
    block_exclusive(x, s_wave, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(unsigned long long *__restrict__ sums, int64_t m, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < m; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const unsigned long long x = i < m ? sums[i] : 0;
        unsigned long long tot;
        const unsigned long long excl = block_exclusive(x, s_wave, &tot);
        const unsigned long long carry = s_carry;
        if (i < m) sums[i] = carry + excl;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const uint32_t *__restrict__ in, int64_t n, const unsigned long long *__restrict__ offs,
                                                                  uint32_t *__restrict__ out32, int64_t *__restrict__ out64)
{
    __shared__ unsigned long long s_wave[kScanThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    load16(in, base, n, v);
    unsigned long long x = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j++) x += v[j];
    unsigned long long run = offs[blockIdx.x] + block_exclusive(x, s_wave, nullptr);
    const bool whole = base + kScanPer <= n;
    if (whole && out32 && (reinterpret_cast<uintptr_t>(out32) & 15u) == 0) {          // 16-byte stores where the rows exist
        uint32_t o[kScanPer];
#pragma unroll
        for (int j = 0; j < kScanPer; j++) { o[j] = (uint32_t)run; run += v[j]; }
#pragma unroll
        for (int j = 0; j < kScanPer; j += 4) *reinterpret_cast<uint4 *>(out32 + base + j) = uint4{o[j], o[j + 1], o[j + 2], o[j + 3]};
        run -= x;
    } else if (out32) {
        unsigned long long r2 = run;
#pragma unroll
        for (int j = 0; j < kScanPer; j++) { if (base + j < n) out32[base + j] = (uint32_t)r2; r2 += v[j]; }
    }
    if (whole && out64 && (reinterpret_cast<uintptr_t>(out64) & 15u) == 0) {
#pragma unroll
        for (int j = 0; j < kScanPer; j += 2) {
            const unsigned long long a = run, b = run + v[j];
            *reinterpret_cast<ulonglong2 *>(out64 + base + j) = ulonglong2{a, b};
            run = b + v[j + 1];
        }
    } else if (out64) {
#pragma unroll
        for (int j = 0; j < kScanPer; j++) { if (base + j < n) out64[base + j] = (int64_t)run; run += v[j]; }
    }
}

// Small inputs (<= kScanSmall elements: the tile counts of a WHERE, the survivor counts of a join): ONE workgroup does
// the whole scan in a single launch instead of three (each launch is ~5 us of a 0.4 ms statement).
constexpr int64_t kScanSmall = 64 * 1024;
__global__ __launch_bounds__(1024) void scan_small_kernel(const uint32_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out32, int64_t *__restrict__ out64,
                                                          unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    constexpr int PER = 32;                            // 32768 counts per round: the 24 414 tile counts of a 1e8-row WHERE take one
    for (int64_t base = 0; base < n; base += 1024 * PER) {
        const int64_t i0 = base + (int64_t)threadIdx.x * PER;
        uint32_t v[PER];
        unsigned long long x = 0;
        if (i0 + PER <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {   // eight 16-byte loads in flight
#pragma unroll
            for (int j = 0; j < PER; j += 4) { const uint4 q = *reinterpret_cast<const uint4 *>(in + i0 + j); v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w; }
#pragma unroll
            for (int j = 0; j < PER; j++) x += v[j];
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++) { v[j] = i0 + j < n ? in[i0 + j] : 0u; x += v[j]; }
        }
        unsigned long long tot;
        unsigned long long run = s_carry + block_exclusive(x, s_wave, &tot);       // (block_exclusive ends with a barrier: s_carry is read before the update below)
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (i0 + j < n) {
                if (out32) out32[i0 + j] = (uint32_t)run;
                if (out64) out64[i0 + j] = (int64_t)run;
            }
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

} // namespace

// The same scan without the host round trip: the total goes to *total_dev (device, 8 bytes), nothing is synchronised.
// sums_ws: nt + 1 words of scratch from k_scan_workspace_words(n) when n > kScanSmall, else unused.
size_t k_scan_workspace_words(int64_t n) { return n <= kScanSmall ? 0 : (size_t)((n + kScanTile - 1) / kScanTile + 1); }
int k_exclusive_scan_u32_dev(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, unsigned long long *total_dev,
                             unsigned long long *sums_ws)
{
    if (n <= 0) return HARK_OK;
    hipStream_t st = ctx->stream;
    if (n <= kScanSmall) HARK_LAUNCH(ctx, scan_small_kernel<<<1, 1024, 0, st>>>(in, n, out32, out64, total_dev));
    else {
        const int64_t nt = (n + kScanTile - 1) / kScanTile;
        HARK_LAUNCH(ctx, tile_sums_kernel<<<dim3((unsigned)nt), dim3(kScanThreads), 0, st>>>(in, n, sums_ws));
        HARK_LAUNCH(ctx, scan_sums_kernel<<<1, 1024, 0, st>>>(sums_ws, nt, total_dev));
        HARK_LAUNCH(ctx, scan_apply_kernel<<<dim3((unsigned)nt), dim3(kScanThreads), 0, st>>>(in, n, sums_ws, out32, out64));
    }
    return HARK_OK;
}

// Exclusive scan of in[0..n).  Either output may be null.  *total_host receives
// the sum (this call synchronises the stream).  in/out may alias.
int k_exclusive_scan_u32(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, int64_t *total_host)
{
    if (total_host) *total_host = 0;
    if (n <= 0) return HARK_OK;
    const int64_t nt = (n + kScanTile - 1) / kScanTile;
    unsigned long long *sums = nullptr;
    HARK_TRY(hark_alloc(ctx, (void **)&sums, (size_t)(nt + 1) * sizeof(unsigned long long)));
    hipStream_t st = ctx->stream;
    int rc = HARK_OK;
    if (n <= kScanSmall) HARK_LAUNCH_RC(ctx, rc, scan_small_kernel<<<1, 1024, 0, st>>>(in, n, out32, out64, sums + nt));
    else {
        HARK_LAUNCH_RC(ctx, rc, tile_sums_kernel<<<dim3((unsigned)nt), dim3(kScanThreads), 0, st>>>(in, n, sums));
        HARK_LAUNCH_RC(ctx, rc, scan_sums_kernel<<<1, 1024, 0, st>>>(sums, nt, sums + nt));
        HARK_LAUNCH_RC(ctx, rc, scan_apply_kernel<<<dim3((unsigned)nt), dim3(kScanThreads), 0, st>>>(in, n, sums, out32, out64));
    }
    int64_t tot = 0;
    if (!rc) rc = hark_read_words(ctx, sums + nt, &tot, 1);
    hark_free(ctx, sums);
    if (total_host) *total_host = tot;
    return rc;
}
