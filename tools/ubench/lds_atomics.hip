// Microbenchmark behind DESIGN.md 4.4: what do LDS atomics cost on gfx950?  Every lane hits a pseudo-random slot of a
// 2048-entry table; reported: clock cycles of the CU per wave instruction (all resident waves issuing), for plain
// reads / writes and for the three atomics cs_multiply's hash insert uses (compare-and-swap with return, unsigned min,
// fp64 add).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int SLOTS = 2048;

template <int OP>
__global__ __launch_bounds__(256) void k_lds(int iters, unsigned *sink) {
    __shared__ unsigned tab[SLOTS];
    __shared__ double dtab[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += 256) {
        tab[i] = 0xffffffffu;
        dtab[i] = 0.0;
    }
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            const unsigned s = (x >> 8) & (SLOTS - 1);
            if (OP == 0) acc += tab[s];
            if (OP == 1) tab[s] = x;
            if (OP == 2) atomicAdd(&tab[s], 1u);                       // no return value used
            if (OP == 3) acc += atomicCAS(&tab[s], 0xffffffffu, x);   // returning
            if (OP == 4) atomicMin(&tab[s], x);
            if (OP == 5) unsafeAtomicAdd(&dtab[s], 1.0);
            if (OP == 6) { acc += atomicCAS(&tab[s], 0xffffffffu, x); atomicMin(&tab[s ^ 1], x); unsafeAtomicAdd(&dtab[s], 1.0); }
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int OP>
static int run(const char *name, int wgs_per_cu, unsigned *sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000, cus = 256;
    hipLaunchKernelGGL(k_lds<OP>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 10, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_lds<OP>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr_per_cu = (double)wgs_per_cu * 4 * iters * 8 * (OP == 6 ? 3 : 1);
    printf("%-44s %d workgroups/CU: %.3f ms, %.1f ns per wave instruction per CU (~%.0f cycles at 2.4 GHz)\n", name, wgs_per_cu, ms,
           ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.4);
    return 0;
}

int main() {
    unsigned *sink;
    CK(hipMalloc(&sink, 64));
    for (int w : {1, 4}) {
        run<0>("ds_read_b32, random slot", w, sink);
        run<1>("ds_write_b32, random slot", w, sink);
        run<2>("atomic add u32 (no return), random slot", w, sink);
        run<3>("compare-and-swap (returning), random slot", w, sink);
        run<4>("atomic min u32, random slot", w, sink);
        run<5>("atomic add f64, random slot", w, sink);
        run<6>("cas + min + add f64 (the hash insert)", w, sink);
    }
    return 0;
}
