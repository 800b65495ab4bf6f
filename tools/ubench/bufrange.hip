// What does the range check of a raw buffer resource (stride 0) do on gfx950?  Loads and stores of 8 / 16 bytes at offsets around
// num_records, with and without a scalar offset.  Build: hipcc --offload-arch=gfx950 -O2 -o bufrange bufrange.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(double *buf, int records, int soff, double *out, double *st) {
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, records, 0x00020000);
    // lane l reads 16 bytes at byte offset 16 l (+ soff)
    u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, soff, 0);
    double2 v = __builtin_bit_cast(double2, u);
    out[2 * lane] = v.x;
    out[2 * lane + 1] = v.y;
    u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, soff, 0);
    out[128 + lane] = __builtin_bit_cast(double, w);
    // negative-looking offset
    u32x2 z = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(-(lane + 1)) * 8u, 0, 0);
    out[192 + lane] = __builtin_bit_cast(double, z);
    // stores through a second resource over st
    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(st, 0, records, 0x00020000);
    double2 s2 = make_double2(1000.0 + lane, 2000.0 + lane);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s2), rt, lane * 16, soff, 0);
}
int main() {
    const int N = 256;
    std::vector<double> h(N), o(256), s(N, -1.0);
    for (int i = 0; i < N; i++) h[i] = i + 1;
    double *d, *dout, *dst;
    hipMalloc(&d, N * 8); hipMalloc(&dout, 256 * 8); hipMalloc(&dst, N * 8);
    hipMemcpy(d, h.data(), N * 8, hipMemcpyHostToDevice);
    for (int soff : {0, 64}) {
        const int records = 40 * 8;   // 40 doubles = 320 bytes
        hipMemcpy(dst, s.data(), N * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, records, soff, dout, dst);
        hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
        std::vector<double> sb(N);
        hipMemcpy(sb.data(), dst, N * 8, hipMemcpyDeviceToHost);
        printf("records %d bytes, soffset %d\n b128 loads (lane: x y):", records, soff);
        for (int l = 14; l < 26; l++) printf(" %d:%g,%g", l, o[2 * l], o[2 * l + 1]);
        printf("\n b64 loads:");
        for (int l = 28; l < 44; l++) printf(" %d:%g", l, o[128 + l]);
        printf("\n b64 loads at negative offsets:");
        for (int l = 0; l < 4; l++) printf(" %d:%g", l, o[192 + l]);
        printf("\n stores (first untouched index after each written run):");
        int last = -1;
        for (int i = 0; i < N; i++) if (sb[i] != -1.0) last = i;
        printf(" last written double index %d (value %g)\n", last, last >= 0 ? sb[last] : 0.0);
    }
    return 0;
}
