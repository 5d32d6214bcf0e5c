// Are the returning LDS atomics of ONE wave instruction handed out in LANE ORDER among the lanes that hit the same address?
// (A stable radix rank could then be ONE ds_add_rtn_u32 per key -- rank = old value -- instead of the lane-mask exchange of
// k_sort.hip's digit_scatter2_kernel: five LDS instructions per key.)  Nothing in the ISA documents promises it, so this
// probe checks it on the device: every wave of 256 workgroups x 16 waves draws digits from distributions with few and many
// distinct values, issues the atomics, and compares what it got with the lane-ordered rank computed from ballots.
// Build: make -C tools ldsorder_test.  Run on the GPU box: prints the number of violations (0 = lane order held everywhere).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(1024) void probe(unsigned long long *bad, unsigned long long *total, int rounds)
{
    __shared__ uint32_t cnt[16][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0u;
    __syncthreads();
    uint32_t mine[256 / 64 * 0 + 1];
    (void)mine;
    uint32_t seen[4] = {0u, 0u, 0u, 0u};            // this lane's view is not enough: expected values come from ballots + a shadow table in registers of lane 0.. -> use LDS shadow
    (void)seen;
    __shared__ uint32_t shadow[16][256];
    for (int i = lane; i < 256; i += 64) shadow[wave][i] = 0u;
    __syncthreads();
    unsigned long long nbad = 0, ntot = 0;
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < rounds; r++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const int mode = r & 3;                     // 0: 256 values, 1: 16 values, 2: 2 values, 3: all lanes one value
        const uint32_t d = mode == 0 ? (x >> 9) & 255u : mode == 1 ? (x >> 9) & 15u : mode == 2 ? (x >> 9) & 1u : 7u;
        const uint32_t got = atomicAdd(&cnt[wave][d], 1u);                 // ds_add_rtn_u32, all 64 lanes in one instruction
        // expected: earlier rounds' count of d + the lanes below me with the same d
        unsigned long long peers = ~0ull;
        for (int b = 0; b < 8; b++) { const unsigned long long m = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? m : ~m; }
        const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t before = shadow[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (below == 0) shadow[wave][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (got != before + below) nbad++;
        ntot++;
    }
    atomicAdd(bad, nbad); atomicAdd(total, ntot);
}

int main()
{
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, 16); hipMemset(d, 0, 16);
    probe<<<256, 1024>>>(d, d + 1, 4000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("ds_add_rtn_u32 lane-order probe: %llu violations in %llu atomics\n", h[0], h[1]);
    return h[0] ? 1 : 0;
}
