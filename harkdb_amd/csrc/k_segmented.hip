// k_segmented.hip -- the vendored diku-dk/segmented primitives HarkDB uses, as
// device operators (futhark/lib/github.com/diku-dk/segmented/segmented.fut):
//   segmented_scan   (:7-13)    hark_op_segmented_scan_add_i32
//   segmented_reduce (:20-37)   hark_op_segmented_reduce_add_i32
//   replicated_iota  (:44-50)   hark_op_replicated_iota
//   segmented_iota   (:58-60)   hark_op_segmented_iota
//   expand           (:70-74)   hark_op_expand_indices (the two index vectors `get` is applied to)
// The operator entries (k_groupby.hip, k_join.hip) do not call these -- they
// fuse the same computations into other kernels -- but exposing them lets the
// reference's own 18 known-answer vectors (segmented_tests.fut:5-72) run
// against HIP code as well.
//
// segmented_scan: the pair operator (f1,x1)+(f2,x2) = (f1|f2, f2 ? x2 : x1+x2) is
// associative, so the scan is tiled: per-tile summary (any flag, sum after the
// last flag) -> sequential-in-one-workgroup scan of the summaries -> per-tile
// rescan seeded with the carry.
#include "hark_internal.h"

int k_exclusive_scan_u32(hark_context *ctx, const uint32_t *in, int64_t n, uint32_t *out32, int64_t *out64, int64_t *total_host);

namespace {

constexpr int kT = 256, kPer = 16, kTile = kT * kPer;

struct Pair { int32_t sum; uint32_t flag; };
__device__ __forceinline__ Pair pair_op(Pair a, Pair b) { return Pair{b.flag ? b.sum : a.sum + b.sum, a.flag | b.flag}; }

__device__ __forceinline__ Pair block_scan_pairs(Pair x, Pair *s_wave, Pair *block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    Pair incl = x;
    for (int d = 1; d < 64; d <<= 1) {
        Pair y{__shfl_up(incl.sum, d, 64), __shfl_up(incl.flag, d, 64)};
        if (lane >= d) incl = pair_op(y, incl);
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    Pair carry{0, 0}, tot{0, 0};
    for (int w = 0; w < nw; w++) { if (w < wave) carry = pair_op(carry, s_wave[w]); tot = pair_op(tot, s_wave[w]); }
    if (block_total) *block_total = tot;
    __syncthreads();
    return pair_op(carry, incl);          // inclusive
}

__global__ __launch_bounds__(kT) void seg_tile_summary_kernel(const uint8_t *__restrict__ flags, const int32_t *__restrict__ vals, int64_t n,
                                                              Pair *__restrict__ summary)
{
    __shared__ Pair s_wave[kT / 64];
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPer;
    Pair acc{0, 0};
#pragma unroll
    for (int j = 0; j < kPer; j++) if (base + j < n) acc = pair_op(acc, Pair{vals ? vals[base + j] : 1, (uint32_t)(flags[base + j] != 0)});
    Pair tot;
    block_scan_pairs(acc, s_wave, &tot);
    if (threadIdx.x == 0) summary[blockIdx.x] = tot;
}

// exclusive scan of the tile summaries by one workgroup
__global__ __launch_bounds__(1024) void seg_scan_summaries_kernel(Pair *__restrict__ summary, int64_t m)
{
    __shared__ Pair s_wave[16];
    __shared__ Pair s_carry;
    if (threadIdx.x == 0) s_carry = Pair{0, 0};
    __syncthreads();
    for (int64_t base = 0; base < m; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const Pair x = i < m ? summary[i] : Pair{0, 0};
        Pair tot;
        const Pair incl = block_scan_pairs(x, s_wave, &tot);
        const Pair carry = s_carry;
        // exclusive = carry (+) everything before me = carry (+) incl of the previous thread
        Pair prev{__shfl_up(incl.sum, 1, 64), __shfl_up(incl.flag, 1, 64)};
        __shared__ Pair s_last[16];
        if ((threadIdx.x & 63) == 63) s_last[threadIdx.x >> 6] = incl;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) prev = threadIdx.x == 0 ? Pair{0, 0} : s_last[(threadIdx.x >> 6) - 1];
        if (i < m) summary[i] = pair_op(carry, prev);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = pair_op(carry, tot);
        __syncthreads();
    }
}

__global__ __launch_bounds__(kT) void seg_tile_apply_kernel(const uint8_t *__restrict__ flags, const int32_t *__restrict__ vals, int64_t n,
                                                            const Pair *__restrict__ carry_in, int32_t *__restrict__ out, int32_t minus)
{
    __shared__ Pair s_wave[kT / 64];
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPer;
    Pair e[kPer];
    Pair acc{0, 0};
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        e[j] = base + j < n ? Pair{vals ? vals[base + j] : 1, (uint32_t)(flags[base + j] != 0)} : Pair{0, 0};
        acc = pair_op(acc, e[j]);
    }
    const Pair incl = block_scan_pairs(acc, s_wave, nullptr);
    // exclusive prefix for this thread = carry_in (+) incl of the previous thread
    __shared__ Pair s_incl[kT];
    s_incl[threadIdx.x] = incl;
    __syncthreads();
    Pair run = carry_in[blockIdx.x];
    if (threadIdx.x > 0) run = pair_op(run, s_incl[threadIdx.x - 1]);
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        run = pair_op(run, e[j]);
        if (base + j < n) out[base + j] = run.sum - minus;
    }
}

__global__ __launch_bounds__(256) void seg_ends_kernel(const uint8_t *__restrict__ flags, int64_t n, uint32_t *__restrict__ ends)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ends[i] = flags[(i + 1) % n] ? 1u : 0u;   // rotate 1
}

__global__ __launch_bounds__(256) void seg_pick_kernel(const uint32_t *__restrict__ ends, const uint32_t *__restrict__ pos, const int32_t *__restrict__ scanned,
                                                       int64_t n, int32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) if (ends[i]) out[pos[i]] = scanned[i];
}

__global__ __launch_bounds__(256) void repl_mark_kernel(const uint32_t *__restrict__ s2, int64_t n, int64_t total, int32_t *__restrict__ tmp)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if ((int64_t)s2[i] < total) atomicMax(&tmp[s2[i]], (int32_t)i);      // reduce_by_index ... i32.max 0
}

__global__ __launch_bounds__(256) void gt0_flags_kernel(const int32_t *__restrict__ tmp, int64_t n, uint8_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) flags[i] = tmp[i] > 0;
}

__global__ __launch_bounds__(256) void neq_prev_flags_kernel(const int32_t *__restrict__ idxs, int64_t n, uint8_t *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) flags[i] = idxs[i] != idxs[(i - 1 + n) % n];   // rotate (-1)
}

int grid_for(hark_context *ctx, int64_t n)
{
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// vals == nullptr scans ones (segmented_iota); `minus` is subtracted from every output
int seg_scan(hark_context *ctx, const uint8_t *flags, const int32_t *vals, int64_t n, int32_t *out, int32_t minus)
{
    if (n <= 0) return HARK_OK;
    const int64_t nt = (n + kTile - 1) / kTile;
    Pair *summary = nullptr;
    HARK_TRY(hark_alloc(ctx, (void **)&summary, (size_t)nt * sizeof(Pair)));
    hipStream_t st = ctx->stream;
    int rc = HARK_OK;
    HARK_LAUNCH_RC(ctx, rc, seg_tile_summary_kernel<<<dim3((unsigned)nt), dim3(kT), 0, st>>>(flags, vals, n, summary));   // vals == nullptr: ones
    HARK_LAUNCH_RC(ctx, rc, seg_scan_summaries_kernel<<<1, 1024, 0, st>>>(summary, nt));
    HARK_LAUNCH_RC(ctx, rc, seg_tile_apply_kernel<<<dim3((unsigned)nt), dim3(kT), 0, st>>>(flags, vals, n, summary, out, minus));
    hark_free(ctx, summary);
    return rc;
}

} // namespace

extern "C" {

int hark_op_segmented_scan_add_i32(hark_context *ctx, const uint8_t *flags, const int32_t *vals, int64_t n, int32_t *out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || n < 0 || (n && (!flags || !vals || !out))) return HARK_EARG;
    return seg_scan(ctx, flags, vals, n, out, 0);
}

int hark_op_segmented_iota(hark_context *ctx, const uint8_t *flags, int64_t n, int32_t *out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || n < 0 || (n && (!flags || !out))) return HARK_EARG;
    return seg_scan(ctx, flags, nullptr, n, out, 1);          // segmented.fut:59-60
}

int hark_op_segmented_reduce_add_i32(hark_context *ctx, const uint8_t *flags, const int32_t *vals, int64_t n,
                                     int32_t *out, int64_t *n_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !n_out || n < 0 || (n && (!flags || !vals || !out))) return HARK_EARG;
    *n_out = 0;
    if (n == 0) return HARK_OK;                               // segmented.fut:29
    int32_t *scanned = nullptr; uint32_t *ends = nullptr, *pos = nullptr;
    int rc = hark_alloc(ctx, (void **)&scanned, (size_t)n * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&ends, (size_t)n * 4);
    if (!rc) rc = hark_alloc(ctx, (void **)&pos, (size_t)n * 4);
    if (!rc) rc = seg_scan(ctx, flags, vals, n, scanned, 0);                                       // :24
    int64_t nseg = 0;
    if (!rc) {
        HARK_LAUNCH_RC(ctx, rc, seg_ends_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(flags, n, ends));   // :26
        if (!rc) rc = k_exclusive_scan_u32(ctx, ends, n, pos, nullptr, &nseg);                     // :28-29 (offset-1 = exclusive)
    }
    if (!rc && nseg > 0) {
        HARK_LAUNCH_RC(ctx, rc, seg_pick_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(ends, pos, scanned, n, out));   // :36-37
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "segmented_reduce failed");
    }
    hark_free(ctx, scanned); hark_free(ctx, ends); hark_free(ctx, pos);
    *n_out = nseg;
    return rc;
}

// reps: n non-negative counts.  out must hold sum(reps) entries (query it with
// out == NULL first: *n_out is always set).
int hark_op_replicated_iota(hark_context *ctx, const int32_t *reps, int64_t n, int32_t *out, int64_t *n_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !n_out || n < 0 || (n && !reps)) return HARK_EARG;
    *n_out = 0;
    if (n == 0) return HARK_OK;
    uint32_t *s2 = nullptr; int32_t *tmp = nullptr; uint8_t *fl = nullptr;
    int64_t total = 0;
    int rc = hark_alloc(ctx, (void **)&s2, (size_t)n * 4);
    if (!rc) rc = k_exclusive_scan_u32(ctx, reinterpret_cast<const uint32_t *>(reps), n, s2, nullptr, &total);   // :45-47
    *n_out = total;
    if (!rc && total > 0 && out) {
        rc = hark_alloc(ctx, (void **)&tmp, (size_t)total * 4);
        if (!rc) rc = hark_alloc(ctx, (void **)&fl, (size_t)total);
        if (!rc) {
            HIP_TRY_RC(ctx, rc, hipMemsetAsync(tmp, 0, (size_t)total * 4, ctx->stream));                         // replicate .. 0
            HARK_LAUNCH_RC(ctx, rc, repl_mark_kernel<<<grid_for(ctx, n), 256, 0, ctx->stream>>>(s2, n, total, tmp));     // :48
            HARK_LAUNCH_RC(ctx, rc, gt0_flags_kernel<<<grid_for(ctx, total), 256, 0, ctx->stream>>>(tmp, total, fl));    // :49
            if (!rc) rc = seg_scan(ctx, fl, tmp, total, out, 0);                                                  // :50
            if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "replicated_iota failed");
        }
    }
    hark_free(ctx, s2); hark_free(ctx, tmp); hark_free(ctx, fl);
    return rc;
}

// expand (segmented.fut:70-74): given szs[i] = sz(arr[i]) returns idxs (:72) and
// iotas (:73); the caller evaluates `get arr[idxs[j]] iotas[j]`.
int hark_op_expand_indices(hark_context *ctx, const int32_t *szs, int64_t n, int32_t *idxs, int32_t *iotas, int64_t *n_out)
{
    hark_device_guard guard__(ctx);
    if (!ctx || !n_out) return HARK_EARG;
    HARK_TRY(hark_op_replicated_iota(ctx, szs, n, idxs, n_out));
    const int64_t t = *n_out;
    if (t == 0 || !idxs || !iotas) return HARK_OK;
    uint8_t *fl = nullptr;
    HARK_TRY(hark_alloc(ctx, (void **)&fl, (size_t)t));
    int rc = HARK_OK;
    HARK_LAUNCH_RC(ctx, rc, neq_prev_flags_kernel<<<grid_for(ctx, t), 256, 0, ctx->stream>>>(idxs, t, fl));
    if (!rc) rc = seg_scan(ctx, fl, nullptr, t, iotas, 1);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = hark_fail(ctx, HARK_EHIP, "expand failed");
    hark_free(ctx, fl);
    return rc;
}

} // extern "C"
