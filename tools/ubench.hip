// ubench.hip -- stand-alone gfx950 micro-benchmarks that size the design of
// k_fgb.hip (not part of libhark.so).  Build: make -C tools.  Run on the GPU box.
//   1. streaming read of three 4-byte columns (the 12 B/row of the headline query)
//   2. scattered global atomics (f32 / u32 / u64) into a 2^20-entry table
//   3. scattered 8-byte stores (register-direct partitioning)
//   4. write-then-read of a W-MiB buffer (does the 256 MiB Infinity Cache absorb
//      the partition buffer of a chunk?)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void fill_kernel(uint32_t *a, int64_t n, uint64_t seed, uint32_t mask)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        a[i] = (uint32_t)splitmix64(seed + i) & mask;
}

template <int UNROLL>
__global__ void read3_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, const uint4 *__restrict__ c,
                             int64_t nvec, uint32_t *__restrict__ out)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
        uint4 x[UNROLL], y[UNROLL], z[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; z[u] = c[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += x[u].x ^ x[u].w ^ y[u].y ^ y[u].z ^ z[u].x ^ z[u].w;
    }
    for (; i < nvec; i += stride) { uint4 x = a[i], y = b[i], z = c[i]; acc += x.x ^ y.y ^ z.z; }
    if (acc == 0x12345678u) out[0] = acc;   // keep the loads alive
}

template <typename T>
__global__ void scatter_atomic_kernel(const uint32_t *__restrict__ keys, int64_t n, T *__restrict__ table)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (sizeof(T) == 4 && !std::is_integral<T>::value) unsafeAtomicAdd(&table[keys[i]], (T)1);
        else atomicAdd(&table[keys[i]], (T)1);
    }
}

__global__ void scatter_store_kernel(const uint32_t *__restrict__ keys, int64_t n, uint2 *__restrict__ buf, uint32_t cap_mask,
                                     uint32_t *__restrict__ cursors, int shift)
{
    // emulates register-direct partitioning: slot = bucket*cap + (cursor[bucket]++ & cap_mask)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t key = keys[i], b = key >> shift;
        uint32_t pos = atomicAdd(&cursors[b], 1u) & cap_mask;
        buf[(size_t)b * (cap_mask + 1) + pos] = uint2{key, (uint32_t)i};
    }
}

__global__ void write_kernel(uint4 *__restrict__ buf, int64_t nvec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride)
        buf[i] = uint4{(uint32_t)i, 1u, 2u, 3u};
}

__global__ void read1_kernel(const uint4 *__restrict__ buf, int64_t nvec, uint32_t *__restrict__ out)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < nvec; i += 2 * stride) { uint4 x = buf[i], y = buf[i + stride]; acc += x.x ^ x.w ^ y.y ^ y.z; }
    for (; i < nvec; i += stride) { uint4 x = buf[i]; acc += x.x ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void reread_kernel(const uint4 *__restrict__ buf, int64_t nvec, int passes, uint32_t *__restrict__ out)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; p++) {
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + stride < nvec; i += 2 * stride) { uint4 x = buf[i], y = buf[i + stride]; acc += x.x ^ x.w ^ y.y ^ y.z; }
        for (; i < nvec; i += stride) { uint4 x = buf[i]; acc += x.x ^ x.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// streams three columns and, per 3 vectors read, writes one vector into / reads one vector from a small ring buffer
__global__ void mixed_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, const uint4 *__restrict__ c, int64_t nvec,
                             uint4 *__restrict__ ring, int64_t ringvec, uint32_t *__restrict__ out)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 x = a[i], y = b[i], z = c[i];
        const int64_t r = (i / 3) % ringvec;
        if (i % 3 == 0) ring[r] = uint4{x.x, y.y, z.z, x.w};
        else if (i % 3 == 1) { uint4 q = ring[(r + ringvec / 2) % ringvec]; acc += q.x ^ q.w; }
        acc += x.x ^ y.y ^ z.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// reads three columns and writes one third of the volume back as a plain stream (the producer's 3:1 mix, no scatter)
template <bool NT>
__global__ void read3_write1_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, const uint4 *__restrict__ c, int64_t nvec,
                                    uint4 *__restrict__ dst)
{
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        uint4 x, y, z;
        if (NT) {
            u4v t0 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(a + i)), t1 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(b + i)),
                t2 = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(c + i));
            x = uint4{t0.x, t0.y, t0.z, t0.w}; y = uint4{t1.x, t1.y, t1.z, t1.w}; z = uint4{t2.x, t2.y, t2.z, t2.w};
        } else { x = a[i]; y = b[i]; z = c[i]; }
        const uint4 o{x.x ^ y.y, y.x ^ z.z, z.x ^ x.w, x.y ^ y.w};
        if (NT) __builtin_nontemporal_store(u4v{o.x, o.y, o.z, o.w}, reinterpret_cast<u4v *>(dst + i)); else dst[i] = o;
    }
}

template <typename F>
static double time_ms(F &&launch, int reps = 7)
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

int main(int argc, char **argv)
{
    int64_t N = argc > 1 ? atoll(argv[1]) : (int64_t)256 << 20;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  L2=%d KiB  mem=%.1f GiB\n", prop.name, prop.multiProcessorCount, prop.l2CacheSize / 1024,
           prop.totalGlobalMem / 1073741824.0);
    uint32_t *a, *b, *c, *out;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&c, N * 4)); CK(hipMalloc(&out, 4096));
    fill_kernel<<<2048, 256>>>(a, N, 1, 0xFFFFFFFFu); fill_kernel<<<2048, 256>>>(b, N, 2, (1u << 20) - 1);
    fill_kernel<<<2048, 256>>>(c, N, 3, 0xFFFFFFFFu);
    CK(hipDeviceSynchronize());

    printf("\n[1] streaming read of 3 columns, N=%lld rows (%.2f GB)\n", (long long)N, N * 12 / 1e9);
    for (int threads : {256, 512, 1024}) for (int mult : {1, 2, 4, 8, 16}) {
        int grid = 256 * mult * 256 / threads; if (grid < 256) grid = 256;
        double ms1 = time_ms([&] { read3_kernel<1><<<grid, threads>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, out); });
        double ms2 = time_ms([&] { read3_kernel<2><<<grid, threads>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, out); });
        double ms4 = time_ms([&] { read3_kernel<4><<<grid, threads>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, out); });
        printf("  threads=%4d grid=%5d  unroll1 %.3f ms %.2f TB/s | unroll2 %.3f ms %.2f TB/s | unroll4 %.3f ms %.2f TB/s\n", threads, grid,
               ms1, N * 12 / ms1 / 1e9, ms2, N * 12 / ms2 / 1e9, ms4, N * 12 / ms4 / 1e9);
    }

    int64_t M = std::min<int64_t>(N, (int64_t)64 << 20);
    printf("\n[2] scattered global atomics, %lld uniform keys into 2^20 entries\n", (long long)M);
    {
        void *table; CK(hipMalloc(&table, 8 << 20)); CK(hipMemset(table, 0, 8 << 20));
        double t1 = time_ms([&] { scatter_atomic_kernel<float><<<2048, 256>>>(b, M, (float *)table); }, 3);
        double t2 = time_ms([&] { scatter_atomic_kernel<uint32_t><<<2048, 256>>>(b, M, (uint32_t *)table); }, 3);
        double t3 = time_ms([&] { scatter_atomic_kernel<unsigned long long><<<2048, 256>>>(b, M, (unsigned long long *)table); }, 3);
        printf("  f32 add: %.3f ms = %.2f G atomics/s | u32 add: %.3f ms = %.2f G/s | u64 add: %.3f ms = %.2f G/s\n",
               t1, M / t1 / 1e6, t2, M / t2 / 1e6, t3, M / t3 / 1e6);
        CK(hipFree(table));
    }

    printf("\n[3] scattered 8-byte stores through per-bucket cursors (register-direct partition), %lld rows\n", (long long)M);
    for (int shift : {12, 10}) {
        int P = 1 << (20 - shift);
        uint32_t cap = (uint32_t)(((M / P) * 2)); uint32_t capp2 = 1; while (capp2 < cap) capp2 <<= 1;
        uint2 *buf; uint32_t *cur; CK(hipMalloc(&buf, (size_t)P * capp2 * 8)); CK(hipMalloc(&cur, P * 4)); CK(hipMemset(cur, 0, P * 4));
        double t = time_ms([&] { scatter_store_kernel<<<2048, 256>>>(b, M, buf, capp2 - 1, cur, shift); }, 3);
        printf("  P=%4d buckets: %.3f ms = %.2f G rows/s (%.2f TB/s of pairs)\n", P, t, M / t / 1e6, M * 8 / t / 1e9);
        CK(hipFree(buf)); CK(hipFree(cur));
    }

    printf("\n[4] write W MiB then read it back (separate launches)\n");
    for (int W : {16, 32, 64, 128, 192, 256, 512, 2048}) {
        int64_t nvec = (int64_t)W * 1048576 / 16;
        uint4 *buf; CK(hipMalloc(&buf, nvec * 16));
        double tw = time_ms([&] { write_kernel<<<2048, 256>>>(buf, nvec); }, 5);
        double tr = time_ms([&] { write_kernel<<<2048, 256>>>(buf, nvec); read1_kernel<<<2048, 256>>>(buf, nvec, out); }, 5);
        double tro = time_ms([&] { read1_kernel<<<2048, 256>>>(buf, nvec, out); }, 5);
        printf("  W=%5d MiB: write %.3f ms (%.2f TB/s) | write+read %.3f ms | read-only (warm) %.3f ms (%.2f TB/s)\n", W, tw,
               nvec * 16 / tw / 1e9, tr, tro, nvec * 16 / tro / 1e9);
        CK(hipFree(buf));
    }
    printf("\n[5] in-kernel repeated read of a W-MiB buffer (Infinity Cache bandwidth), 16 passes\n");
    for (int W : {32, 64, 128, 192, 256, 384, 1024}) {
        int64_t nvec = (int64_t)W * 1048576 / 16;
        uint4 *buf; CK(hipMalloc(&buf, nvec * 16));
        write_kernel<<<2048, 256>>>(buf, nvec);
        for (int grid : {512, 2048}) {
            double t = time_ms([&] { reread_kernel<<<grid, 256>>>(buf, nvec, 16, out); }, 3);
            printf("  W=%5d MiB grid=%4d: %.3f ms for 16 passes -> %.2f TB/s\n", W, grid, t, 16.0 * nvec * 16 / t / 1e9);
        }
        CK(hipFree(buf));
    }
    printf("\n[6] read 3 GB stream + concurrently re-read/overwrite a 128 MiB buffer (does the stream evict it?)\n");
    {
        int64_t nvec = (int64_t)128 * 1048576 / 16;
        uint4 *buf; CK(hipMalloc(&buf, nvec * 16));
        write_kernel<<<2048, 256>>>(buf, nvec);
        double t0 = time_ms([&] { read3_kernel<2><<<512, 256>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, out); });
        double t1 = time_ms([&] { mixed_kernel<<<512, 256>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, buf, nvec, out); });
        printf("  stream only %.3f ms; stream + write&read of 128 MiB x%d rounds: %.3f ms (extra bytes %.2f GB)\n", t0, (int)(N / 4 / (nvec)), t1, 2.0 * (N / 4) * 16 / 3 / 1e9);
        CK(hipFree(buf));
    }
    printf("\n[7] streaming mix of the partition producer: read 3 columns, write 1 column-equivalent (contiguous)\n");
    {
        uint4 *dst; CK(hipMalloc(&dst, N * 4));
        for (int grid : {256, 512, 2048}) {
            double t0 = time_ms([&] { read3_write1_kernel<false><<<grid, 1024>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, dst); });
            double t1 = time_ms([&] { read3_write1_kernel<true><<<grid, 1024>>>((uint4 *)a, (uint4 *)b, (uint4 *)c, N / 4, dst); });
            printf("  grid=%4d x1024: plain %.3f ms (%.2f TB/s total) | non-temporal %.3f ms (%.2f TB/s total)\n", grid, t0, N * 16 / t0 / 1e9, t1, N * 16 / t1 / 1e9);
        }
        CK(hipFree(dst));
    }
    printf("done\n");
    return 0;
}
