// What does FETCH_SIZE count for plain and for non-temporal loads?  Two kernels read the same 1 GiB once with 16-byte loads;
// run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the counter per kernel (profiles/r04_notes.md).
// Build: make -C tools fetchcount_test     Run: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- tools/fetchcount_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_plain(const u4v *__restrict__ src, size_t n16, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u4v q = src[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
    if (acc == 0x12345678u) *out = acc;
}
__global__ __launch_bounds__(256) void read_nt(const u4v *__restrict__ src, size_t n16, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u4v q = __builtin_nontemporal_load(src + i); acc += q.x ^ q.y ^ q.z ^ q.w; }
    if (acc == 0x12345678u) *out = acc;
}
// the same bytes as 4-byte loads with a stride of 8 bytes (the low words of 64-bit keys): every line is touched, half of it used
__global__ __launch_bounds__(256) void read_low_words(const unsigned *__restrict__ src, size_t n8, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) acc += src[2 * i];
    if (acc == 0x12345678u) *out = acc;
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *buf; unsigned *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    if (hipMemset(buf, 1, bytes) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; rep++) {
        read_plain<<<4096, 256>>>(static_cast<const u4v *>(buf), bytes / 16, out);
        read_nt<<<4096, 256>>>(static_cast<const u4v *>(buf), bytes / 16, out);
        read_low_words<<<4096, 256>>>(static_cast<const unsigned *>(buf), bytes / 8, out);
    }
    const hipError_t e = hipDeviceSynchronize();
    printf("read 1 GiB three ways, twice: %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
