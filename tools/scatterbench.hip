// scatterbench.hip -- the producer's memory traffic without its LDS work: 256 persistent workgroups stream three
// 4-byte columns (non-temporal, two batches of 4096 rows in flight) and write 64 units of 384 B per 8192 rows
//   mode 0: nothing written (the read stream alone)
//   mode 1: the product's layout -- slab (bucket, workgroup), units appended: 65536 write streams
//   mode 2: one contiguous stream per workgroup (same bytes)
//   mode 3: [workgroup][unit number][bucket] -- a sweep over the buckets writes 96 KiB contiguously
// Round 2 probe: the producer runs at 2.58 / 2.83 / 2.97 ms on different GPUs of the pool with identical clocks --
// does a bare write pattern show the same classes?   Build: make -C tools.  Usage: tools/scatterbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
constexpr int kP = 256, kPieces = 24;

__global__ __launch_bounds__(1024) void scatter_kernel(const u4v *__restrict__ a, const u4v *__restrict__ b, const u4v *__restrict__ c, int iters, u4v *out, int mode, unsigned *sink)
{
    const int w = blockIdx.x, t = threadIdx.x;
    const u4v *pa = a + (int64_t)w * iters * 2048 + t, *pb = b + (int64_t)w * iters * 2048 + t, *pc = c + (int64_t)w * iters * 2048 + t;
    u4v x0 = __builtin_nontemporal_load(pa), x1 = __builtin_nontemporal_load(pa + 1024);
    u4v y0 = __builtin_nontemporal_load(pb), y1 = __builtin_nontemporal_load(pb + 1024);
    u4v z0 = __builtin_nontemporal_load(pc), z1 = __builtin_nontemporal_load(pc + 1024);
    u4v acc = {0u, 0u, 0u, 0u};
    const int64_t units_per_slab = ((int64_t)iters * 64 + kP - 1) / kP + 1;      // every bucket gets a unit every 4th iteration
    for (int it = 0; it < iters; it++) {
        u4v nx0 = x0, nx1 = x1, ny0 = y0, ny1 = y1, nz0 = z0, nz1 = z1;
        const u4v o = x0 ^ y0 ^ z0 ^ x1 ^ y1 ^ z1;
        if (it + 1 < iters) {
            const int64_t q = (int64_t)(it + 1) * 2048;
            nx0 = __builtin_nontemporal_load(pa + q); nx1 = __builtin_nontemporal_load(pa + q + 1024);
            ny0 = __builtin_nontemporal_load(pb + q); ny1 = __builtin_nontemporal_load(pb + q + 1024);
            nz0 = __builtin_nontemporal_load(pc + q); nz1 = __builtin_nontemporal_load(pc + q + 1024);
        }
        if (mode && t < (1024 / kPieces) * kPieces) {
            for (int u = t / kPieces; u < 64; u += 1024 / kPieces) {
                const int bk = (it * 64 + u) & (kP - 1), piece = t % kPieces;
                const int64_t un = (int64_t)(it * 64 + u) / kP;                 // this bucket's unit number in this workgroup
                int64_t at;
                if (mode == 1) at = (((int64_t)bk * kP + w) * units_per_slab + un) * kPieces + piece;
                else if (mode == 2) at = ((int64_t)w * iters * 64 + (int64_t)it * 64 + u) * kPieces + piece;
                else at = (((int64_t)w * units_per_slab + un) * kP + bk) * kPieces + piece;
                __builtin_nontemporal_store(o, out + at);
            }
        }
        acc ^= o;
        x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1; z0 = nz0; z1 = nz1;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

__global__ void fill_kernel(uint32_t *a, int64_t n, uint64_t seed)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = (uint32_t)((seed + i) * 0x9E3779B97F4A7C15ull >> 29);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 477;
    const int64_t N = (int64_t)iters * 8192 * kP;
    char bus[64] = {0}; CK(hipDeviceGetPCIBusId(bus, 63, 0));
    printf("device %s; %lld rows (%.2f GB read, %.2f GB written)\n", bus, (long long)N, N * 12 / 1e9, (double)iters * kP * 64 * 384 / 1e9);
    uint32_t *a, *b, *c; unsigned *sink; u4v *out;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&c, N * 4)); CK(hipMalloc(&sink, 4096));
    const int64_t units_per_slab = ((int64_t)iters * 64 + kP - 1) / kP + 1;
    CK(hipMalloc(&out, (size_t)kP * kP * units_per_slab * 384 + (size_t)kP * iters * 64 * 384));
    fill_kernel<<<4096, 256>>>(a, N, 1); fill_kernel<<<4096, 256>>>(b, N, 2); fill_kernel<<<4096, 256>>>(c, N, 3);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"read stream alone", "+ units into 65536 slabs (product layout)", "+ units, one contiguous stream per workgroup", "+ units, [workgroup][unit][bucket]"};
    for (int round = 0; round < 2; round++)
        for (int mode = 0; mode < 4; mode++) {
            std::vector<float> ts;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                scatter_kernel<<<kP, 1024>>>((u4v *)a, (u4v *)b, (u4v *)c, iters, out, mode, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("mode %d %-48s %.3f ms\n", mode, names[mode], ts[ts.size() / 2]);
        }
    return 0;
}
