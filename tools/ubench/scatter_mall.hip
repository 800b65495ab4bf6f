// Microbenchmark: does the Infinity Cache (256 MiB) carry a radix pass?  Two scatter passes over 3.4e8 records of 16 bytes
// (A -> B by 256 buckets, then B -> A), either each pass over the WHOLE array (what the LSD sort of cs_transpose does:
// every pass reads and writes 5.4 GB from / to HBM), or segment by segment -- both passes of one segment of S records
// back to back, buckets inside the segment, before the next segment is touched -- so that the second pass reads what the
// first just wrote while it may still sit in the Infinity Cache.  Pure data movement (tools/ubench/scatter_runs.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// tile t of the segment (T records, read coalesced) writes a run of T / nb records to each of nb buckets of the segment
__global__ __launch_bounds__(256) void k_scatter(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, int64_t seg_len,
                                                 int nb, int T, uint32_t ntiles) {
    const uint32_t q = ntiles >> 3, rem = ntiles & 7u, x = blockIdx.x & 7u, kk = blockIdx.x >> 3;
    const uint32_t t = x * q + (x < rem ? x : rem) + kk;
    const int run = T / nb;
    const int64_t bucket_len = seg_len / nb;
    for (int s = threadIdx.x; s < T; s += 256) {
        const u32x4 r = in[(int64_t)t * T + s];
        const int b = s / run, off = s - b * run;
        out[(int64_t)b * bucket_len + (int64_t)t * run + off] = r;
    }
}

int main() {
    const int64_t count = (int64_t)320 << 20;
    u32x4 *a, *b;
    CK(hipMalloc(&a, count * 16));
    CK(hipMalloc(&b, count * 16));
    CK(hipMemset(a, 1, count * 16));
    CK(hipMemset(b, 0, count * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int T = 2048, nb = 256;
    const int64_t segs[] = {count, (int64_t)64 << 20, (int64_t)16 << 20, (int64_t)8 << 20, (int64_t)4 << 20, (int64_t)2 << 20, (int64_t)1 << 20};
    for (int64_t seg : segs) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            for (int64_t s0 = 0; s0 < count; s0 += seg) {
                const uint32_t ntiles = (uint32_t)(seg / T);
                hipLaunchKernelGGL(k_scatter, dim3(ntiles), dim3(256), 0, 0, a + s0, b + s0, seg, nb, T, ntiles);
                hipLaunchKernelGGL(k_scatter, dim3(ntiles), dim3(256), 0, 0, b + s0, a + s0, seg, nb, T, ntiles);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("segment %10lld records (%7.1f MB in + out): two passes %.3f ms = %.3f ms per pass, %.2f TB/s (read + write)\n",
               (long long)seg, 2.0 * seg * 16 / 1e6, best, best / 2, 2.0 * 2.0 * count * 16 / best * 1e-9);
    }
    return 0;
}
