// Microbenchmark: issue cost of the instructions k_chol_clique is made of, on a full chip (every SIMD busy, W waves per
// SIMD): v_mul_f64, v_add_f64, v_fma_f64, v_readlane_b32, v_mov_b32, v_mov_b64 dpp, ds_read_b64 broadcast.  Reported: shader clocks
// (s_memtime) per wave instruction per SIMD, and the clock the chip held (s_memrealtime is 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(int iters, double *out, unsigned long long *stamps) {
    __shared__ double lds[256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    int s0 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (OP == 0) {
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (OP == 1) {
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if (OP == 2) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 3) {
                int t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_;
                asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9\n"
                             "v_readlane_b32 %4, %12, 11\n v_readlane_b32 %5, %13, 13\n v_readlane_b32 %6, %14, 15\n v_readlane_b32 %7, %15, 17\n"
                             : "=s"(t0_), "=s"(t1_), "=s"(t2_), "=s"(t3_), "=s"(t4_), "=s"(t5_), "=s"(t6_), "=s"(t7_)
                             : "v"(__double2loint(a0)), "v"(__double2loint(a1)), "v"(__double2loint(a2)), "v"(__double2loint(a3)),
                               "v"(__double2loint(a4)), "v"(__double2loint(a5)), "v"(__double2loint(a6)), "v"(__double2loint(a7)));
                s0 += t0_ ^ t1_ ^ t2_ ^ t3_ ^ t4_ ^ t5_ ^ t6_ ^ t7_;
            } else if (OP == 4) {
                asm volatile("v_mul_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_mul_f64 %2, %2, %8\n v_add_f64 %3, %3, %9\n"
                             "v_mul_f64 %4, %4, %8\n v_add_f64 %5, %5, %9\n v_mul_f64 %6, %6, %8\n v_add_f64 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 5) {   // fp32 multiply for comparison
                float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
                asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                             "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f));
                a0 = f0; a1 = f1; a2 = f2; a3 = f3;
            } else if (OP == 6) {   // broadcast LDS read, 8 bytes
                const double *p = lds + (i & 63);
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                             "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                             : "v"((unsigned)(size_t)p));
            } else if (OP == 7) {   // broadcast LDS read, 16 bytes
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 q0, q1, q2, q3;
                const double *p = lds + (i & 63);
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"((unsigned)(size_t)p));
                a0 += q0.x + q1.y; a1 += q2.x + q3.y;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0;
}

template <int OP>
static int run(const char *name, int wgs_per_cu, int per_iter, double *out, unsigned long long *st) {
    const int cus = 256, iters = 4000;
    const int grid = cus * wgs_per_cu;
    std::vector<unsigned long long> h(2 * grid * 4);
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(256), 0, 0, 10, out, st);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(256), 0, 0, iters, out, st);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    double clk = 0, real = 0;
    for (int i = 0; i < grid * 4; i++) { clk += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    clk /= grid * 4; real /= grid * 4;
    const double instr_per_wave = (double)iters * 4 * per_iter;
    // a SIMD hosts wgs_per_cu waves (one wave of each workgroup): they share its issue
    printf("%-22s %d waves/SIMD: %6.2f shader clocks per wave instruction per SIMD, shader clock %.0f MHz (s_memtime), kernel %.3f ms -> %.2f ns per instruction per SIMD\n",
           name, wgs_per_cu, clk / (instr_per_wave * wgs_per_cu), clk / real * 100.0, ms, ms * 1e6 / (instr_per_wave * wgs_per_cu));
    return 0;
}

int main() {
    double *out;
    unsigned long long *st;
    CK(hipMalloc(&out, sizeof(double) * 256 * 256 * 8));
    CK(hipMalloc(&st, 16 * 256 * 8 * 4));
    for (int w : {1, 3}) {
        run<0>("v_mul_f64", w, 8, out, st);
        run<1>("v_add_f64", w, 8, out, st);
        run<2>("v_fma_f64", w, 8, out, st);
        run<4>("v_mul_f64 + v_add_f64", w, 8, out, st);
        run<3>("v_readlane_b32", w, 8, out, st);
        run<5>("v_mul_f32", w, 8, out, st);
        run<6>("ds_read_b64 broadcast", w, 8, out, st);
        run<7>("ds_read_b128 broadcast", w, 4, out, st);
    }
    return 0;
}
