// libsort_yardstick.hip -- a yardstick, not product code: how long does the vendor library's radix sort (rocPRIM, the
// decoupled-look-back "onesweep" of /opt/rocm/include/rocprim/device/device_radix_sort.hpp) take for the ORDER BY workloads
// of bench.py on this card?  hark's own sort (k_sort.hip) is hand-written and does not link this; the numbers say whether a
// different pass structure (one histogram pass up front + look-back inside every scatter pass, instead of a histogram pass
// per scatter pass) would be worth building.  Build: make -C tools libsort_yardstick.  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/rocprim_version.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <class K> __global__ void fill_kernel(K *k, uint32_t *v, int64_t n, int bits)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        k[i] = (K)(bits >= 64 ? x : (x & ((1ull << bits) - 1)));
        v[i] = (uint32_t)i;
    }
}

template <class K> static void run(const char *name, int64_t n, int key_bits, int sort_bits)
{
    K *k0, *k1; uint32_t *v0, *v1;
    CK(hipMalloc(&k0, n * sizeof(K))); CK(hipMalloc(&k1, n * sizeof(K))); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    fill_kernel<K><<<2048, 256>>>(k0, v0, n, key_bits);
    size_t tmp_bytes = 0; void *tmp = nullptr;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, sort_bits));
    CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0));
        CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, sort_bits));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s n = %lld: median %.3f ms, best %.3f ms  (temporary storage %.1f MB)\n", name, (long long)n, ts[ts.size() / 2], ts[0], tmp_bytes / 1e6);
    CK(hipFree(k0)); CK(hipFree(k1)); CK(hipFree(v0)); CK(hipFree(v1)); CK(hipFree(tmp));
}

int main()
{
    const int64_t n = 100000000;
    printf("rocPRIM %d radix_sort_pairs, keys + a 32-bit payload (the row index), not in place\n", ROCPRIM_VERSION);
    run<uint32_t>("u32 keys below 2^20, bits [0, 20)   (ORDER_BY)", n, 20, 20);
    run<uint32_t>("u32 keys below 2^20, bits [0, 32)   (what a caller without key statistics does)", n, 20, 32);
    run<uint32_t>("u32 keys below 2^31, bits [0, 31)   (ORDER_BY_32bit)", n, 31, 31);
    run<uint32_t>("u32 keys, bits [0, 32)", n, 32, 32);
    run<uint64_t>("u64 keys, bits [0, 64)              (ORDER_BY_i64)", n, 64, 64);
    run<uint64_t>("u64 keys below 2^40, bits [0, 40)", n, 40, 40);
    return 0;
}
