// ldsbench.hip -- LDS atomic / RMW throughput on gfx950 (sizes the aggregation kernels).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t x) { return x * 1664525u + 1013904223u; }

// MODE 0: ds_add_f32 lane-private slot; 1: ds_add_f32 all lanes same slot; 2: ds_add_f32 random in 4096;
// 3: ds_add_u32 random in 4096; 4: ds_add_rtn_u32 random in 256; 5: plain RMW lane-private (64 replicas, 16 keys);
// 6: ds_add_f32 random in 4096, half the lanes masked off; 7: ds_add_u32 lane-private; 8: ds_add_f32 random in 16*32 replicated (key*32+lane%32)
// 9: ds_add_u64 random in 4096
template <int MODE>
__global__ void lds_kernel(int iters, uint32_t *out)
{
    __shared__ uint32_t s[16384];
    float *sf = reinterpret_cast<float *>(s);
    unsigned long long *s64 = reinterpret_cast<unsigned long long *>(s);
    double *sd = reinterpret_cast<double *>(s);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = 0;
    __syncthreads();
    uint32_t r = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        r = lcg(r);
        const uint32_t rnd = r >> 8;
        if constexpr (MODE == 0) unsafeAtomicAdd(&sf[threadIdx.x], 1.0f);
        else if constexpr (MODE == 1) unsafeAtomicAdd(&sf[7], 1.0f);
        else if constexpr (MODE == 2) unsafeAtomicAdd(&sf[rnd & 4095], 1.0f);
        else if constexpr (MODE == 3) atomicAdd(&s[rnd & 4095], 1u);
        else if constexpr (MODE == 4) acc += atomicAdd(&s[rnd & 255], 1u);
        else if constexpr (MODE == 5) { uint32_t a = ((rnd & 15) << 6) | lane; sf[a] = sf[a] + 1.0f; }
        else if constexpr (MODE == 6) { if (rnd & 0x10000) unsafeAtomicAdd(&sf[rnd & 4095], 1.0f); }
        else if constexpr (MODE == 7) atomicAdd(&s[threadIdx.x], 1u);
        else if constexpr (MODE == 8) unsafeAtomicAdd(&sf[((rnd & 15) << 5) | (lane & 31)], 1.0f);
        else if constexpr (MODE == 9) atomicAdd(&s64[rnd & 4095], 1ull);
        else if constexpr (MODE == 10) {                    // CAS-loop float add, random in 4096
            uint32_t *a = &s[rnd & 4095]; uint32_t old = *a, assumed;
            do { assumed = old; old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
        }
        else if constexpr (MODE == 11) unsafeAtomicAdd(&sd[rnd & 4095], 1.0);
        else if constexpr (MODE == 12) atomicAdd(&s[7], 1u);
        else if constexpr (MODE == 13) atomicAdd(&s[rnd & 15], 1u);
        else if constexpr (MODE == 14) atomicAdd(&s64[rnd & 15], 1ull);
        else if constexpr (MODE == 15) atomicAdd(&s64[((rnd & 15) << 5) | (lane & 31)], 1ull);
        else if constexpr (MODE == 16) atomicMax(&s[rnd & 4095], rnd);
        else if constexpr (MODE == 17) { atomicAdd(&s64[rnd & 4095], 1ull); atomicAdd(&s[8192 + (rnd & 4095)], 1u); }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[7] + acc;
}

template <int MODE>
static void run(const char *name, int threads, uint32_t *out)
{
    const int iters = 8192, grid = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    lds_kernel<MODE><<<grid, threads>>>(iters, out); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0)); lds_kernel<MODE><<<grid, threads>>>(iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    double ms = t[2];
    double laneops = (double)iters * threads;           // per CU (1 WG per CU)
    printf("  %-58s threads=%4d: %7.3f ms  %6.2f lane-ops/ns/CU  (~%5.2f per clk @2.1GHz)  chip %7.1f G/s\n", name, threads, ms,
           laneops / (ms * 1e6), laneops / (ms * 1e6) / 2.1, laneops * 256 / (ms * 1e6));
}

int main()
{
    uint32_t *out; CK(hipMalloc(&out, 4096));
    for (int threads : {256, 1024}) {
        run<0>("ds_add_f32 lane-private slot", threads, out);
        run<7>("ds_add_u32 lane-private slot", threads, out);
        run<1>("ds_add_f32 all lanes one slot", threads, out);
        run<2>("ds_add_f32 random in 4096", threads, out);
        run<6>("ds_add_f32 random in 4096, ~half lanes active", threads, out);
        run<3>("ds_add_u32 random in 4096", threads, out);
        run<9>("ds_add_u64 random in 4096", threads, out);
        run<4>("ds_add_rtn_u32 random in 256", threads, out);
        run<8>("ds_add_f32 16 keys x 32 lane replicas", threads, out);
        run<5>("plain read+add+write, 16 keys x 64 lane-private", threads, out);
        run<10>("CAS-loop f32 add random in 4096", threads, out);
        run<11>("ds_add_f64 random in 4096", threads, out);
        run<12>("ds_add_u32 all lanes one slot", threads, out);
        run<13>("ds_add_u32 random in 16 (no replicas)", threads, out);
        run<14>("ds_add_u64 random in 16 (no replicas)", threads, out);
        run<15>("ds_add_u64 16 keys x 32 lane replicas", threads, out);
        run<16>("ds_max_u32 random in 4096", threads, out);
        run<17>("ds_add_u64 + ds_add_u32 random in 4096 (pair)", threads, out);
    }
    return 0;
}
