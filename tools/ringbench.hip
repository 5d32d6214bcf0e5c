// ringbench.hip -- does a tight write->read hand-off ride on the Infinity Cache / fabric
// instead of HBM?  (round 2 design probe for the fused producer+consumer of k_fgb.hip;
// stand-alone, not part of libhark.so.)  Build: make -C tools.  Run on the GPU box.
//
// Every workgroup (one per CU, 1024 threads) streams its share of three 4-byte columns
// (12 B/row, non-temporal, like fgb_part_kernel) and per 4096-row batch
//   * writes 12 KiB (the 3 B/row of partition pairs at 50 % selectivity) at the head of its own
//     ring, and
//   * reads 12 KiB that a PARTNER workgroup (another XCD, or the same one) wrote `lag` batches ago.
// The ring per workgroup is R batches long; R*12 KiB*256 is the footprint the hand-off traffic
// lives in.  A small footprint that is re-written before it is evicted never has to reach HBM;
// "advancing" (R = all batches) is today's slab layout (every byte written to and read from HBM).
// No synchronisation: this measures bandwidth only (the partner may not have written yet).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

// access flavours (buffer instructions so that the compiler tracks vmcnt): aux 0 plain, 2 nt, 16 sc1, 17 sc0 sc1
#define RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, 0x7fffffff, 0x00020000)

// mode bits: 1 = write the ring, 2 = read the partner's ring
template <int WF, int RF>
__global__ __launch_bounds__(1024) void ring_kernel(const u4v *__restrict__ a, const u4v *__restrict__ b, const u4v *__restrict__ c,
                                                    int64_t batches_per_wg, u4v *ring, int64_t ring_batches, int lag, int partner_step, int mode,
                                                    unsigned *__restrict__ out)
{
    const int w = blockIdx.x, t = threadIdx.x;
    const int pw = (w + partner_step) % gridDim.x;
    u4v acc = {0u, 0u, 0u, 0u};
    const u4v *pa = a + (int64_t)w * batches_per_wg * 1024 + t, *pb = b + (int64_t)w * batches_per_wg * 1024 + t, *pc = c + (int64_t)w * batches_per_wg * 1024 + t;
    const __amdgpu_buffer_rsrc_t mine = RSRC(ring + (int64_t)w * ring_batches * 768);
    const __amdgpu_buffer_rsrc_t theirs = RSRC(ring + (int64_t)pw * ring_batches * 768);
    u4v x0 = __builtin_nontemporal_load(pa), y0 = __builtin_nontemporal_load(pb), z0 = __builtin_nontemporal_load(pc);
    for (int64_t i = 0; i < batches_per_wg; i++) {
        u4v x1 = x0, y1 = y0, z1 = z0;
        if (i + 1 < batches_per_wg) {                                  // next batch in flight while this one is "processed"
            x1 = __builtin_nontemporal_load(pa + (i + 1) * 1024); y1 = __builtin_nontemporal_load(pb + (i + 1) * 1024); z1 = __builtin_nontemporal_load(pc + (i + 1) * 1024);
        }
        const u4v o = x0 ^ y0 ^ z0;
        if (t < 768) {
            if (mode & 1) __builtin_amdgcn_raw_buffer_store_b128(o, mine, (int)(((i % ring_batches) * 768 + t) * 16), 0, WF);
            if (mode & 2) {
                const int64_t j = i - lag;
                if (j >= 0) acc ^= __builtin_amdgcn_raw_buffer_load_b128(theirs, (int)(((j % ring_batches) * 768 + t) * 16), 0, RF);
            }
        }
        acc ^= o;
        x0 = x1; y0 = y1; z0 = z1;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;
}

__global__ void fill_kernel(uint32_t *a, int64_t n, uint64_t seed)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = (uint32_t)((seed + i) * 0x9E3779B97F4A7C15ull >> 29);
}

template <typename F>
static double time_ms(F &&launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

template <int WF, int RF>
static void run(const char *name, const u4v *a, const u4v *b, const u4v *c, int64_t bpw, u4v *ring, int64_t max_ring_batches, unsigned *out, double t_stream)
{
    for (int partner : {1, 8}) {
        printf("  [%s] partner = w+%d (%s XCD)\n", name, partner, partner % 8 ? "another" : "the same");
        for (int64_t R : {(int64_t)4, (int64_t)16, (int64_t)64, (int64_t)256, max_ring_batches}) {
            if (R > max_ring_batches) continue;
            const int lag = (int)std::min<int64_t>(R / 2, 32);
            double tw = time_ms([&] { ring_kernel<WF, RF><<<256, 1024>>>(a, b, c, bpw, ring, R, lag, partner, 1, out); });
            double twr = time_ms([&] { ring_kernel<WF, RF><<<256, 1024>>>(a, b, c, bpw, ring, R, lag, partner, 3, out); });
            printf("    ring %5lld batches/WG = %8.1f MiB total, lag %2d: stream+write %.3f ms (+%.3f) | stream+write+read %.3f ms (+%.3f)\n", (long long)R,
                   R * 12288.0 * 256 / 1048576.0, lag, tw, tw - t_stream, twr, twr - t_stream);
        }
    }
}

int main(int argc, char **argv)
{
    const int64_t bpw = argc > 1 ? atoll(argv[1]) : 960;            // batches of 4096 rows per workgroup: 960 -> 1.0e9 rows
    const int64_t N = bpw * 4096 * 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d; %lld rows (%.2f GB streamed, %.2f GB hand-off written, same read)\n", prop.name, prop.multiProcessorCount, (long long)N, N * 12 / 1e9, N * 3 / 1e9);
    uint32_t *a, *b, *c; unsigned *out; u4v *ring;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&c, N * 4)); CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&ring, (size_t)bpw * 12288 * 256));
    fill_kernel<<<4096, 256>>>(a, N, 1); fill_kernel<<<4096, 256>>>(b, N, 2); fill_kernel<<<4096, 256>>>(c, N, 3);
    CK(hipMemset(ring, 0, (size_t)bpw * 12288 * 256));
    CK(hipDeviceSynchronize());
    double ts = time_ms([&] { ring_kernel<0, 0><<<256, 1024>>>((u4v *)a, (u4v *)b, (u4v *)c, bpw, ring, 4, 2, 1, 0, out); });
    printf("stream only: %.3f ms = %.2f TB/s\n", ts, N * 12 / ts / 1e9);
    run<0, 0>("plain st / plain ld", (u4v *)a, (u4v *)b, (u4v *)c, bpw, ring, bpw, out, ts);
    run<2, 2>("nt st / nt ld", (u4v *)a, (u4v *)b, (u4v *)c, bpw, ring, bpw, out, ts);
    run<16, 16>("sc1 st / sc1 ld", (u4v *)a, (u4v *)b, (u4v *)c, bpw, ring, bpw, out, ts);
    run<17, 17>("sc0 sc1 st / sc0 sc1 ld", (u4v *)a, (u4v *)b, (u4v *)c, bpw, ring, bpw, out, ts);
    run<2, 16>("nt st / sc1 ld", (u4v *)a, (u4v *)b, (u4v *)c, bpw, ring, bpw, out, ts);
    printf("done\n");
    return 0;
}
