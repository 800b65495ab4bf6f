// Microbenchmark: what clock does the chip hold for a LIGHT, latency-bound kernel (few active lanes, one short
// wave per CU, launched back to back) compared with a kernel that fills it?  Reads s_memtime (shader clock) and
// s_memrealtime (100 MHz) around a dependent fp64 chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_chain(double *out, unsigned long long *stamps, int iters, int active_lanes) {
    const int lane = threadIdx.x & 63;
    if (lane >= active_lanes) return;
    double a = 1.0 + lane, b = 1.0000001;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) a = a / b - 1e-9;          // dependent fp64 divide + subtract
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && (threadIdx.x >> 6) == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    double *out;
    unsigned long long *st;
    const int maxb = 4096;
    CK(hipMalloc(&out, sizeof(double) * maxb * 1024));
    CK(hipMalloc(&st, sizeof(unsigned long long) * 2 * maxb));
    std::vector<unsigned long long> h(2 * maxb);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Cfg { int blocks, threads, lanes, iters, reps; const char *name; };
    Cfg cfgs[] = {{374, 64, 4, 400, 300, "374 waves x 4 lanes, 400 steps, 300 launches"},
                  {374, 64, 64, 400, 300, "374 waves x 64 lanes, 400 steps, 300 launches"},
                  {2048, 1024, 64, 400, 300, "2048 x 1024 threads, 400 steps, 300 launches"},
                  {374, 64, 4, 400, 300, "374 waves x 4 lanes again"}};
    for (auto &c : cfgs) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < c.reps; r++) hipLaunchKernelGGL(k_chain, dim3(c.blocks), dim3(c.threads), 0, 0, out, st, c.iters, c.lanes);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * c.blocks, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (int b = 0; b < c.blocks; b++) mhz.push_back(100.0 * (double)h[2 * b] / (double)h[2 * b + 1]);
        std::sort(mhz.begin(), mhz.end());
        printf("%-48s  %.1f us per launch, in-kernel clock median %.0f MHz (min %.0f), %.0f shader cycles per step\n", c.name,
               ms * 1e3 / c.reps, mhz[mhz.size() / 2], mhz[0], (double)h[0] / c.iters);
    }
    return 0;
}
