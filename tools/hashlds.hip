// hashlds.hip -- what does the LDS side of a hash GROUP BY probe cost on gfx950, layout by layout?  
//
// fgb_agg_hash_kernel is LDS-bound (profiles/r05_hash_pmc.txt: the LDS arrays are busy 72 % of the kernel's cycles, 59 % of
// those cycles are bank-conflict cycles).  Per pair it issues two ds_read_b128 (the eight 32-bit tags of the key's home
// group), one ds_add_f64 (SUM) and one ds_add_u32 (COUNT), all at random addresses.  This probe runs ONLY that LDS work --
// steady state, every pair a hit, slots drawn uniformly, eight pairs per lane in flight as in the kernel, one 1024-thread
// workgroup per CU, no global traffic -- for the layouts one could give the table, and prints the time for 2.5e8 pairs
// chip-wide, to be read against the stream floor of those pairs (0.34 ms) and the kernel's 0.68-0.73 ms.
// Build: make -C tools hashlds.  Run on the GPU box: tools/hashlds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kSlots = 8192;
__device__ __forceinline__ uint32_t lcg(uint32_t x) { return x * 1664525u + 1013904223u; }
__device__ __forceinline__ uint32_t mixw(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// the slot-in-group a probe's tag read leads to: three well-mixed bits of what was read (a first version took the low bits of
// an XOR of two multiplicative tags: only two of the eight values ever came out, the atomics of a step piled onto a quarter of
// the banks, and the probe reported reads and atomics as costing twice as much together as apart)
__device__ __forceinline__ uint32_t low3(uint32_t a, uint32_t b) { return ((a ^ b) * 0x9E3779B1u) >> 29; }

// what one pair does:
//  bit 0: two ds_read_b128 of a 32-byte group of 32-bit tags      bit 1: one ds_read_b128 of a 16-byte group (16-bit tags x 8 .. or 16)
//  bit 2: one ds_read_b64 of an 8-byte group                       bit 3: ds_add_f64 at the slot
//  bit 4: ds_add_u32 at the slot                                   bit 5: ds_add_u64 at the slot (integer sums / a packed word)
//  bit 6: ds_add_f64 of a SECOND value array (count kept as a double)
//  bit 7: two neighbouring 8-byte groups at an 8-byte-aligned address (ds_read2_b64: a four-tag home group and the next one)
template <int WHAT, bool PIPE = false, bool PHASED = false>
__global__ __launch_bounds__(1024) void probe_kernel(int steps, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double *t_val = reinterpret_cast<double *>(lds);                              // [kSlots]
    uint32_t *t_tag = reinterpret_cast<uint32_t *>(t_val + kSlots);               // [kSlots]
    uint32_t *t_cnt = t_tag + kSlots;                                             // [kSlots]
    double *t_val2 = reinterpret_cast<double *>(t_cnt + kSlots);                  // [kSlots / 2] (only when bit 6)
    for (int i = threadIdx.x; i < kSlots; i += blockDim.x) { t_val[i] = 0.0; t_tag[i] = mixw((uint32_t)i); t_cnt[i] = 0u; }
    if (WHAT & 64) for (int i = threadIdx.x; i < kSlots / 2; i += blockDim.x) t_val2[i] = 0.0;
    __syncthreads();
    uint32_t r = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    uint32_t acc = 0;
    const uint4 *tag4 = reinterpret_cast<const uint4 *>(t_tag);
    const uint2 *tag2 = reinterpret_cast<const uint2 *>(t_tag);
    uint32_t pat[8];                                                              // PIPE: the slots the previous step's tag reads led to
#pragma unroll
    for (int j = 0; j < 8; j++) pat[j] = (threadIdx.x * 8u + j) & (kSlots - 1);
    for (int s = 0; s < steps; s++) {
        uint32_t slot[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { r = lcg(r); slot[j] = (r >> 9) & (kSlots - 1); }
        uint4 qa[8], qb[8]; uint2 qc[8], qd[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (WHAT & 1) { qa[j] = tag4[2u * (slot[j] >> 3)]; qb[j] = tag4[2u * (slot[j] >> 3) + 1u]; }
            if (WHAT & 2) qa[j] = tag4[slot[j] >> 2];                            // 16-byte groups anywhere in the tag array
            if (WHAT & 4) qc[j] = tag2[slot[j] >> 1];
            if (WHAT & 128) { const uint32_t a8 = (slot[j] >> 1) & (kSlots / 2 - 1); qc[j] = tag2[a8]; qd[j] = tag2[a8 + 1u]; }
        }
        if (PHASED) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's reads are done before any wave's atomics start
        if (PIPE) {
            // software pipeline: this step's tag reads are in flight while the PREVIOUS step's atomics are issued (the LDS returns
            // in order, so the wait for the reads below is a counted lgkmcnt that leaves these atomics outstanding)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t at = pat[j];
                if (WHAT & 8) unsafeAtomicAdd(&t_val[at], 1.0);
                if (WHAT & 16) atomicAdd(&t_cnt[at], 1u);
                if (WHAT & 32) atomicAdd(reinterpret_cast<unsigned long long *>(&t_val[at]), 1ull);
                if (WHAT & 64) unsafeAtomicAdd(&t_val2[at >> 1], 1.0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t at = slot[j];
            if (WHAT & 1) { acc += qa[j].x ^ qa[j].w ^ qb[j].y; at = (at & ~7u) | low3(qa[j].x, qb[j].z); }       // the slot depends on the tags read (as a real probe's does)
            if (WHAT & 2) { acc += qa[j].x ^ qa[j].w; at = (at & ~7u) | low3(qa[j].x, qa[j].z); }
            if (WHAT & 4) { acc += qc[j].x; at = (at & ~7u) | low3(qc[j].x, qc[j].y); }
            if (WHAT & 128) { acc += qc[j].x ^ qd[j].y; at = (at & ~7u) | low3(qc[j].x, qd[j].y); }
            if (PIPE) { pat[j] = at; continue; }
            if (WHAT & 8) unsafeAtomicAdd(&t_val[at], 1.0);
            if (WHAT & 16) atomicAdd(&t_cnt[at], 1u);
            if (WHAT & 32) atomicAdd(reinterpret_cast<unsigned long long *>(&t_val[at]), 1ull);
            if (WHAT & 64) unsafeAtomicAdd(&t_val2[at >> 1], 1.0);
            if (WHAT & 256) atomicAdd(reinterpret_cast<uint32_t *>(t_val) + at, 1u);
        }
        if (PHASED) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and every wave's atomics before the next step's reads
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + t_cnt[7] + (uint32_t)t_val[9];
}

template <int WHAT, bool PIPE = false, bool PHASED = false> static void run(const char *name, uint32_t *out, int ncu)
{
    const double pairs = 2.5e8;
    const int steps = (int)(pairs / ((double)ncu * 1024 * 8) + 0.5);
    const size_t lds = (size_t)kSlots * 16 + ((WHAT & 64) ? (size_t)kSlots * 4 : 0);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_kernel<WHAT, PIPE, PHASED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < 6; rep++) {
        CK(hipEventRecord(e0));
        probe_kernel<WHAT, PIPE, PHASED><<<ncu, 1024, lds>>>(steps, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const double done = (double)steps * ncu * 1024 * 8;
    printf("%-86s %7.3f ms per 2.5e8 pairs  (%.1f LDS-busy-equivalent cycles per 64 pairs at 2.4 GHz)\n", name, ts[2] * pairs / done,
           ts[2] * 1e-3 * 2.4e9 / (done / ncu / 64));
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint32_t *out; CK(hipMalloc(&out, 4096 * 4));
    printf("hashlds: LDS work of a hash-table probe alone, %d CUs x 1024 threads, 8 pairs per lane and step, random slots in %d\n", ncu, kSlots);
    run<0>("nothing (address generation only)", out, ncu);
    run<1>("2 x ds_read_b128 (eight 32-bit tags: today's probe)", out, ncu);
    run<2>("1 x ds_read_b128 (sixteen / eight 16-bit tags)", out, ncu);
    run<4>("1 x ds_read_b64  (four 16-bit tags)", out, ncu);
    run<8>("ds_add_f64", out, ncu);
    run<16>("ds_add_u32", out, ncu);
    run<32>("ds_add_u64", out, ncu);
    run<8 | 16>("ds_add_f64 + ds_add_u32 (the hit of SUM + COUNT)", out, ncu);
    run<1 | 8 | 16>("2 x b128 + f64 + u32  = fgb_agg_hash_kernel today", out, ncu);
    run<2 | 8 | 16>("1 x b128 + f64 + u32  = 16-bit tags", out, ncu);
    run<4 | 8 | 16>("1 x b64  + f64 + u32  = four 16-bit tags per group", out, ncu);
    run<2 | 8>("1 x b128 + f64        (no count atomic)", out, ncu);
    run<2 | 32>("1 x b128 + u64        (one integer atomic: u32 sums, or a packed word)", out, ncu);
    run<2 | 8 | 64>("1 x b128 + f64 + f64  (count as a double beside the sum)", out, ncu);
    run<128>("ds_read2_b64 (two neighbouring four-tag groups, 8-byte aligned)", out, ncu);
    run<128 | 8 | 16>("read2_b64 + f64 + u32", out, ncu);
    run<4 | 16 | 32>("1 x b64 + u64 + u32", out, ncu);
    run<4 | 16>("1 x b64 + u32", out, ncu);
    run<4 | 16 | 256>("1 x b64 + u32 + u32 (two 32-bit operators: the reference entry)", out, ncu);
    run<2 | 16 | 256>("1 x b128 + u32 + u32", out, ncu);
    printf("-- software-pipelined: a step's tag reads are issued BEFORE the previous step's atomics, and waited for after them\n");
    run<1 | 8 | 16, true>("2 x b128 + f64 + u32, pipelined", out, ncu);
    run<2 | 8 | 16, true>("1 x b128 + f64 + u32, pipelined", out, ncu);
    run<4 | 8 | 16, true>("1 x b64  + f64 + u32, pipelined", out, ncu);
    run<2 | 32, true>("1 x b128 + u64, pipelined", out, ncu);
    run<2 | 8, true>("1 x b128 + f64, pipelined", out, ncu);
    printf("-- phased: a workgroup barrier between the reads and the atomics of a step, and after the atomics (reads and atomics of different waves never meet in the LDS queue)\n");
    run<1 | 8 | 16, false, true>("2 x b128 + f64 + u32, phased", out, ncu);
    run<2 | 8 | 16, false, true>("1 x b128 + f64 + u32, phased", out, ncu);
    run<4 | 8 | 16, false, true>("1 x b64  + f64 + u32, phased", out, ncu);
    run<2 | 32, false, true>("1 x b128 + u64, phased", out, ncu);
    printf("-- more combinations\n");
    run<4 | 32>("1 x b64  + u64", out, ncu);
    run<4 | 8>("1 x b64  + f64", out, ncu);
    run<1 | 32>("2 x b128 + u64", out, ncu);
    run<1 | 16>("2 x b128 + u32", out, ncu);
    run<2 | 16>("1 x b128 + u32", out, ncu);
    run<2 | 16 | 32>("1 x b128 + u64 + u32", out, ncu);
    return 0;
}
