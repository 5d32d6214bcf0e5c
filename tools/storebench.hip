// storebench.hip -- does the cache policy of a STORE move a streaming kernel on gfx950?  The write path is what every
// partition-style kernel of libhark pays for (profiles/r04_notes.md 5, 6): 16-byte stores issued from inline assembly with
// each combination of the gfx942+ policy bits (sc0, sc1, nt), in (a) a plain copy (16 B read + 16 B written per lane step)
// and (b) the headline producer's mix (read three words, write 3 B/row = a quarter of one column, contiguous).
// Build: make -C tools storebench.  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

template <int POL> __device__ __forceinline__ void st16(void *p, u4v v)
{
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

template <int POL>
__global__ __launch_bounds__(256) void copy_kernel(const u4v *__restrict__ src, u4v *__restrict__ dst, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < nvec; i += 2 * stride) {
        const u4v a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        st16<POL>(dst + i, a); st16<POL>(dst + i + stride, b);
    }
    for (; i < nvec; i += stride) st16<POL>(dst + i, __builtin_nontemporal_load(src + i));
}

// three columns read, a quarter of one column's volume written (every fourth lane step stores)
template <int POL>
__global__ __launch_bounds__(256) void mix_kernel(const u4v *__restrict__ a, const u4v *__restrict__ b, const u4v *__restrict__ c, u4v *__restrict__ dst, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u4v acc = {0u, 0u, 0u, 0u};
    size_t step = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride, step++) {
        const u4v x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i), z = __builtin_nontemporal_load(c + i);
        acc ^= x ^ y ^ z;
        if ((step & 3) == 3) { st16<POL>(dst + (i >> 2), acc); }          // (i >> 2: a contiguous quarter-size stream per step group; exact placement is irrelevant here)
    }
    if (acc.x == 0x12345u) dst[0] = acc;
}

template <int POL> void run(const char *name, u4v *a, u4v *b, u4v *c, u4v *d, size_t nvec)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> tc, tm;
    for (int r = 0; r < 7; r++) {
        float ms;
        CK(hipEventRecord(e0)); copy_kernel<POL><<<2048, 256>>>(a, d, nvec); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tc.push_back(ms);
        CK(hipEventRecord(e0)); mix_kernel<POL><<<2048, 256>>>(a, b, c, d, nvec); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tm.push_back(ms);
    }
    std::sort(tc.begin(), tc.end()); std::sort(tm.begin(), tm.end());
    const double bytes = (double)nvec * 16;
    printf("%-14s copy %.4f ms = %.2f TB/s (r+w)   mix 3r + 1/4 w %.4f ms = %.2f TB/s\n", name, tc[3], 2 * bytes / tc[3] / 1e9, tm[3], 3.25 * bytes / tm[3] / 1e9);
}

int main()
{
    const size_t nvec = (size_t)1 << 26;                       // 1 GiB per buffer
    u4v *a, *b, *c, *d;
    CK(hipMalloc(&a, nvec * 16)); CK(hipMalloc(&b, nvec * 16)); CK(hipMalloc(&c, nvec * 16)); CK(hipMalloc(&d, nvec * 16));
    CK(hipMemset(a, 1, nvec * 16)); CK(hipMemset(b, 2, nvec * 16)); CK(hipMemset(c, 3, nvec * 16)); CK(hipMemset(d, 0, nvec * 16));
    run<0>("plain", a, b, c, d, nvec); run<1>("nt", a, b, c, d, nvec); run<2>("sc0", a, b, c, d, nvec); run<3>("sc1", a, b, c, d, nvec);
    run<4>("sc0 sc1", a, b, c, d, nvec); run<5>("sc0 nt", a, b, c, d, nvec); run<6>("sc1 nt", a, b, c, d, nvec); run<7>("sc0 sc1 nt", a, b, c, d, nvec);
    return 0;
}
