// Does gfx950 execute scalar atomics (s_atomic_add, returning, tracked by lgkmcnt -- not by the vmcnt queue of the vector
// loads and stores)?  Every wave of 256 x 16 draws 1000 numbers; all 4 096 000 must be distinct and dense.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(1024) void draw(unsigned *ctr, unsigned *out, int per)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < per; i++) {
        unsigned v = 1u;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if ((threadIdx.x & 63) == 0) out[(size_t)wave * per + i] = v;
    }
}
int main()
{
    const int per = 1000, waves = 256 * 16;
    unsigned *ctr, *out; CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&out, (size_t)waves * per * 4)); CK(hipMemset(ctr, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    draw<<<256, 1024>>>(ctr, out, per);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h((size_t)waves * per); unsigned total;
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&total, ctr, 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    bool ok = total == h.size();
    for (size_t i = 0; i < h.size() && ok; i++) ok = h[i] == i;
    printf("scalar atomics: counter %u of %zu, numbers %s; %.3f ms = %.1f M draws/s\n", total, h.size(), ok ? "distinct and dense" : "WRONG", ms, h.size() / ms / 1e3);
    return ok ? 0 : 1;
}
