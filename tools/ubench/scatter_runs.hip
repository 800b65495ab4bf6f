// Microbenchmark behind DESIGN.md 4.2: what does a radix-scatter pass cost as a function of the number of buckets and
// the run length a tile contributes to each bucket?  Pure data movement, no ranking: tile t (T records of 16 bytes,
// read coalesced) writes a run of T/NB records to every one of NB buckets, at the place a stable scatter would
// (bucket base + t * run).  `staged` = consecutive threads write consecutive slots of the tile's bucket order (what a
// kernel that stages the tile in LDS does); otherwise every lane writes its record where it falls (direct).
// `seq` consecutive tiles are taken by one workgroup one after the other.  Tiles are dealt to XCDs in contiguous ranges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_scatter(const u32x4 *__restrict__ in, u32x4 *__restrict__ out,
                                                     int64_t count, int nb, int T, int seq, uint32_t ngroups,
                                                     int staged) {
    const uint32_t q = ngroups >> 3, rem = ngroups & 7u, x = blockIdx.x & 7u, kk = blockIdx.x >> 3;
    const uint32_t group = x * q + (x < rem ? x : rem) + kk;
    const int run = T / nb;
    const int64_t bucket_len = count / nb;
    for (int sub = 0; sub < seq; sub++) {
        const int64_t t = (int64_t)group * seq + sub;
        for (int s = threadIdx.x; s < T; s += THREADS) {
            const u32x4 r = in[t * T + s];
            const int slot = staged ? s : (int)(((uint32_t)s * 1031u) & (uint32_t)(T - 1));
            const int b = slot / run, off = slot - b * run;
            out[(int64_t)b * bucket_len + t * run + off] = r;
        }
    }
}

int main() {
    const int64_t count = (int64_t)320 << 20;   // 3.4e8 records of 16 bytes
    u32x4 *in, *out;
    CK(hipMalloc(&in, count * 16));
    CK(hipMalloc(&out, count * 16));
    CK(hipMemset(in, 1, count * 16));
    CK(hipMemset(out, 0, count * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Cfg { int threads, nb, T, seq, staged; };
    const Cfg cfgs[] = {{256, 256, 4096, 1, 1},   {256, 256, 4096, 1, 0},   {1024, 256, 8192, 1, 1},
                        {1024, 1024, 8192, 1, 1}, {1024, 2048, 8192, 1, 1}, {1024, 4096, 8192, 1, 1},
                        {1024, 4096, 8192, 1, 0}, {1024, 4096, 8192, 4, 1}, {1024, 4096, 8192, 4, 0},
                        {1024, 2048, 8192, 4, 1}, {256, 4096, 4096, 1, 0},  {256, 4096, 4096, 8, 0},
                        {1024, 4096, 8192, 16, 1}, {1024, 1, 8192, 1, 1}};
    for (const Cfg &c : cfgs) {
        const uint32_t ngroups = (uint32_t)(count / ((int64_t)c.T * c.seq));
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            if (c.threads == 256)
                hipLaunchKernelGGL(k_scatter<256>, dim3(ngroups), dim3(256), 0, 0, in, out, count, c.nb, c.T, c.seq, ngroups, c.staged);
            else
                hipLaunchKernelGGL(k_scatter<1024>, dim3(ngroups), dim3(1024), 0, 0, in, out, count, c.nb, c.T, c.seq, ngroups, c.staged);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("threads %4d buckets %4d tile %5d x %2d %-7s run %3d B : %.3f ms  %.2f TB/s (read + write)\n", c.threads, c.nb, c.T,
               c.seq, c.staged ? "staged" : "direct", c.T / c.nb * 16, best, 2.0 * count * 16 / best * 1e-9);
    }
    return 0;
}
