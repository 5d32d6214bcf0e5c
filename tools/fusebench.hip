// fusebench.hip -- what would a FUSED producer+consumer of the large-G filter->group-by cost in memory traffic and
// hand-off protocol alone?  (round 2 design probe; stand-alone, not part of libhark.so.  Build: make -C tools.)
//
// 256 persistent workgroups (one per CU, 1024 threads).  Every workgroup
//   * streams its share of three 4-byte columns (12 B/row, non-temporal, two batches of 4096 rows per iteration, the
//     next iteration's loads in flight) -- the producer's read stream;
//   * per iteration hands 64 units of 384 B (64 pairs of 6 B: what 8192 rows leave at 50 % selectivity) to 64 of the
//     256 consumers in rotation: sc1 (write-through) 16-byte stores into a mailbox ring of S slots per (consumer,
//     producer), published ONE ITERATION LATER by a 4-byte sc1 store of the unit count into pub[consumer][producer]
//     (the loop's own wait for the next rows has drained the stores by then: no extra fence in the loop);
//   * as consumer b polls pub[b][0..255] (1 KiB, sc1 loads), fetches the new units (sc1 loads, 24 lanes x 16 B per
//     unit) and acknowledges them in ack[producer][b]; a producer skips a hand-off whose ring is full (counted).
// Nothing waits for anything (no spin loops): a probe of bandwidth and protocol overhead, not a correct hand-off.
// No LDS work at all: the real kernel's ring scatter and table atomics would have to hide under this.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#define RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, 0x7fffffff, 0x00020000)
constexpr int kP = 256, kPieces = 24;                    // workgroups; 16-byte pieces per 384-byte unit
constexpr int AUX_SC1 = 16;

// mode bits: 1 = write units, 2 = publish + poll + fetch + ack
__global__ __launch_bounds__(1024) void fuse_kernel(const u4v *__restrict__ a, const u4v *__restrict__ b, const u4v *__restrict__ c, int iters,
                                                    u4v *mail, int S, uint32_t *pub, uint32_t *ack, int mode, unsigned long long *stats, unsigned *sink)
{
    __shared__ uint32_t s_ack[kP], s_list[kP * 2], s_nlist, s_sent[kP];
    const int w = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < kP; i += 1024) { s_ack[i] = 0u; s_sent[i] = 0u; }
    if (t == 0) s_nlist = 0u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_mail = RSRC(mail), r_pub = RSRC(pub), r_ack = RSRC(ack);
    const u4v *pa = a + (int64_t)w * iters * 2048 + t, *pb = b + (int64_t)w * iters * 2048 + t, *pc = c + (int64_t)w * iters * 2048 + t;
    u4v x0 = __builtin_nontemporal_load(pa), x1 = __builtin_nontemporal_load(pa + 1024);
    u4v y0 = __builtin_nontemporal_load(pb), y1 = __builtin_nontemporal_load(pb + 1024);
    u4v z0 = __builtin_nontemporal_load(pc), z1 = __builtin_nontemporal_load(pc + 1024);
    u4v acc = {0u, 0u, 0u, 0u};
    uint32_t seen = 0u;                                  // thread p < 256 as consumer: units of producer p taken so far
    unsigned long long skipped = 0, fetched = 0;
    for (int it = 0; it < iters; it++) {
        u4v nx0 = x0, nx1 = x1, ny0 = y0, ny1 = y1, nz0 = z0, nz1 = z1;
        // (the compiler's wait for x0.. below drains everything issued in the previous iteration, stores included)
        const u4v o = x0 ^ y0 ^ z0 ^ x1 ^ y1 ^ z1;
        // ---- publish what the previous iteration stored
        if ((mode & 2) && it > 0 && t < 64) {
            const int bk = ((it - 1) * 64 + t + w) & (kP - 1);
            __builtin_amdgcn_raw_buffer_store_b32(s_sent[bk], r_pub, (bk * kP + w) * 4, 0, AUX_SC1);
        }
        if (it + 1 < iters) {
            const int64_t q = (int64_t)(it + 1) * 2048;
            nx0 = __builtin_nontemporal_load(pa + q); nx1 = __builtin_nontemporal_load(pa + q + 1024);
            ny0 = __builtin_nontemporal_load(pb + q); ny1 = __builtin_nontemporal_load(pb + q + 1024);
            nz0 = __builtin_nontemporal_load(pc + q); nz1 = __builtin_nontemporal_load(pc + q + 1024);
        }
        // ---- flow control: the consumers' acknowledgements of MY units (1 KiB, every 4th iteration)
        if ((mode & 2) && (it & 3) == 0 && t < 64) {
            const u4v k = __builtin_amdgcn_raw_buffer_load_b128(r_ack, (w * kP + 4 * t) * 4, 0, AUX_SC1);
            s_ack[4 * t] = k.x; s_ack[4 * t + 1] = k.y; s_ack[4 * t + 2] = k.z; s_ack[4 * t + 3] = k.w;
        }
        __syncthreads();
        // ---- hand 64 units to 64 consumers: 24 lanes x 16 B each
        if (mode & 1) {
            for (int u = t / kPieces; u < 64; u += 1024 / kPieces) {
                if (t >= (1024 / kPieces) * kPieces) break;
                const int bk = (it * 64 + u + w) & (kP - 1), piece = t % kPieces;
                const uint32_t sent = s_sent[bk];
                if ((mode & 2) && sent - s_ack[bk] >= (uint32_t)S) { if (piece == 0) skipped++; continue; }
                const int slot = (int)(sent % (uint32_t)S);
                __builtin_amdgcn_raw_buffer_store_b128(o, r_mail, (int)(((((int64_t)bk * kP + w) * S + slot) * kPieces + piece) * 16), 0, AUX_SC1);
            }
            __syncthreads();
            if (t < 64) {
                const int bk = (it * 64 + t + w) & (kP - 1);
                if (!(mode & 2) || s_sent[bk] - s_ack[bk] < (uint32_t)S) s_sent[bk]++;
            }
        }
        // ---- consumer: poll the 256 producers' counts for MY bucket, fetch what is new, acknowledge
        if (mode & 2) {
            uint32_t have = 0;
            if (t < kP) have = __builtin_amdgcn_raw_buffer_load_b32(r_pub, (w * kP + t) * 4, 0, AUX_SC1);
            if (t < kP && have != seen) {
                const uint32_t n_new = have - seen < (uint32_t)S ? have - seen : (uint32_t)S;
                for (uint32_t q = 0; q < n_new; q++) { const uint32_t at = atomicAdd(&s_nlist, 1u); if (at < kP * 2) s_list[at] = ((uint32_t)t << 8) | ((seen + q) % (uint32_t)S); }
                seen = have;
                __builtin_amdgcn_raw_buffer_store_b32(seen, r_ack, (t * kP + w) * 4, 0, AUX_SC1);
            }
            __syncthreads();
            const uint32_t nl = s_nlist < kP * 2 ? s_nlist : kP * 2;
            for (uint32_t u = t / kPieces; u < nl; u += 1024 / kPieces) {
                if (t >= (1024 / kPieces) * kPieces) break;
                const uint32_t e = s_list[u], prod = e >> 8, slot = e & 255u;
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(r_mail, (int)(((((int64_t)w * kP + prod) * S + slot) * kPieces + t % kPieces) * 16), 0, AUX_SC1);
                if (t % kPieces == 0) fetched++;
            }
            __syncthreads();
            if (t == 0) s_nlist = 0u;
        }
        acc ^= o;
        x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1; z0 = nz0; z1 = nz1;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
    if (skipped) atomicAdd(&stats[0], skipped);
    if (fetched) atomicAdd(&stats[1], fetched);
}

__global__ void fill_kernel(uint32_t *a, int64_t n, uint64_t seed)
{
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = (uint32_t)((seed + i) * 0x9E3779B97F4A7C15ull >> 29);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 477;               // x 256 workgroups x 8192 rows = 1.0e9 rows
    const int64_t N = (int64_t)iters * 8192 * kP;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d; %lld rows (%.2f GB streamed, %.2f GB of units handed over)\n", prop.name, prop.multiProcessorCount, (long long)N, N * 12 / 1e9, (double)iters * kP * 64 * 384 / 1e9);
    uint32_t *a, *b, *c, *pub, *ack; unsigned *sink; unsigned long long *stats; u4v *mail;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&c, N * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&stats, 64));
    CK(hipMalloc(&pub, kP * kP * 4)); CK(hipMalloc(&ack, kP * kP * 4));
    const int Smax = 16;
    CK(hipMalloc(&mail, (size_t)kP * kP * Smax * 384));
    fill_kernel<<<4096, 256>>>(a, N, 1); fill_kernel<<<4096, 256>>>(b, N, 2); fill_kernel<<<4096, 256>>>(c, N, 3);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int S : {2, 4, 8, 16}) {
        for (int mode : {0, 1, 3}) {
            std::vector<float> ts;
            unsigned long long st[2] = {0, 0};
            for (int rep = 0; rep < 4; rep++) {
                CK(hipMemset(pub, 0, kP * kP * 4)); CK(hipMemset(ack, 0, kP * kP * 4)); CK(hipMemset(stats, 0, 64));
                CK(hipEventRecord(e0));
                fuse_kernel<<<kP, 1024>>>((u4v *)a, (u4v *)b, (u4v *)c, iters, mail, S, pub, ack, mode, stats, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ts.push_back(ms);
                CK(hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost));
            }
            std::sort(ts.begin(), ts.end());
            const double units = (double)iters * kP * 64;
            printf("S=%2d (%6.1f MiB of mailboxes) mode=%d (%s): %.3f ms   skipped %.1f %%  fetched %.1f %% of the units\n", S, (double)kP * kP * S * 384 / 1048576.0, mode,
                   mode == 0 ? "stream only" : mode == 1 ? "stream + unit stores" : "stream + stores + publish/poll/fetch/ack", ts[ts.size() / 2], 100.0 * st[0] / units, 100.0 * st[1] / units);
            if (mode == 0 && S != 2) {}
        }
    }
    printf("done\n");
    return 0;
}
