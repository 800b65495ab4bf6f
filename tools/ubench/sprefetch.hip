// Microbenchmark (not product code): can the SCALAR memory path pull HBM lines into an XCD's L2 at a useful
// rate, beside / instead of the vector path?  Each wave issues s_load_dword at a 128-byte stride over its own
// region (one request per L2 line), up to 15 outstanding (lgkmcnt is 4 bits), results discarded.
//   mode 0: scalar prefetch only           -> lines/s through the scalar cache path
//   mode 1: vector streaming read only     -> baseline (16 B per lane, nt)
//   mode 2: both: waves 0..V-1 stream region A with vector loads, waves V.. prefetch region B with s_loads
//   mode 3: vector streaming of a region a scalar-prefetch kernel has just walked (run after mode 0 on the same
//           buffer, sized to fit L2+MALL) -- not used here
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// region = bytes per workgroup; every WG walks [wg*region, (wg+1)*region)
template <int BATCH>
__device__ __forceinline__ void sprefetch_lines(const char *base, long nlines, int w, int nw) {
    // wave w of nw takes lines w, w+nw, ...; BATCH s_loads in flight, then wait for all (SMEM returns out of order)
    for (long l = w; l < nlines; l += (long)nw * BATCH) {
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            long ll = l + (long)b * nw;
            if (ll < nlines) {
                const unsigned long long pv = (unsigned long long)(base + ll * 128);
                const unsigned long long p = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pv >> 32)) << 32) |
                                             (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pv);
                // destination: a fixed high SGPR the compiler does not otherwise use (checked in the .s): the load
                // lands asynchronously, so it must not be a register the compiler may re-use while it is in flight
                asm volatile("s_load_dword s96, %0, 0x0" : : "s"(p) : "s96", "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

__global__ __launch_bounds__(1024) void k_mix(const char *A, const char *B, long region, int vwaves, int mode,
                                              unsigned int *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const char *a = A + (long)blockIdx.x * region;
    const char *b = B + (long)blockIdx.x * region;
    unsigned int acc = 0;
    if (mode == 0) {
        sprefetch_lines<12>(b, region / 128, wave, nw);
    } else if (mode == 1 || (mode == 2 && wave < vwaves)) {
        const int v = mode == 1 ? nw : vwaves;
        // 4 x 1 KiB loads in flight per wave
        const long chunks = region / 1024;
        for (long c = wave; c < chunks; c += (long)v * 4) {
            u32x4 r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                long cc = c + (long)u * v;
                if (cc >= chunks) cc = chunks - 1;
                r[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a + cc * 1024) + lane);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
        }
    } else {
        sprefetch_lines<12>(b, region / 128, wave - vwaves, nw - vwaves);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const long total = 4l << 30;  // 4 GiB per buffer: beyond the 256 MiB Infinity Cache
    char *A, *B;
    unsigned int *sink;
    CK(hipMalloc(&A, total));
    CK(hipMalloc(&B, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(A, 1, total));
    CK(hipMemset(B, 2, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256;
    const long region = total / grid;
    struct Cfg { int mode, threads, vwaves; const char *name; };
    std::vector<Cfg> cfgs = {
        {1, 1024, 16, "vector stream, 16 waves"},   {1, 512, 8, "vector stream, 8 waves"},
        {1, 256, 4, "vector stream, 4 waves"},      {0, 1024, 0, "scalar prefetch, 16 waves"},
        {0, 512, 0, "scalar prefetch, 8 waves"},    {0, 256, 0, "scalar prefetch, 4 waves"},
        {0, 64, 0, "scalar prefetch, 1 wave"},      {2, 1024, 12, "12 vector + 4 scalar waves"},
        {2, 1024, 8, "8 vector + 8 scalar waves"},  {2, 768, 8, "8 vector + 4 scalar waves"},
    };
    for (auto &c : cfgs) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mix, dim3(grid), dim3(c.threads), 0, 0, A, B, region, c.vwaves, c.mode, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) {
                double gb_v = (c.mode == 0 ? 0 : (double)total) / 1e9, gb_s = (c.mode == 1 ? 0 : (double)total) / 1e9;
                printf("%-32s %8.3f ms  vector %.2f TB/s  scalar-touched %.2f TB/s (lines x 128 B)\n", c.name, ms,
                       gb_v / ms, gb_s / ms);
            }
        }
    }
    return 0;
}
