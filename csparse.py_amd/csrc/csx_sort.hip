// Device primitives: exclusive scan, column expansion, stable LSD radix sort,
// segment boundaries.  These carry the integer side of cs_cumsum
// (csparse.py:767-784) and of the stable counting sort inside cs_transpose
// (csparse.py:2305-2314) on the device.
//
// Stability matters: cs_transpose emits each output column in ascending
// (source column, source position) order, and p[]/i[] must be bit-exact.  The
// sort is therefore a least-significant-digit radix sort whose scatter ranks
// equal digits by (workgroup, wave, round, lane), i.e. by source position.
#include "csx_internal.h"

namespace csx {

// ---------------------------------------------------------------- scan ----
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns the
// exclusive prefix and the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
    __shared__ int wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_inclusive_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; k++) {
        int s = wsum[k];
        if (k < w) off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int32_t *in, int64_t n, int32_t *sums) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int64_t idx = base + (int64_t)k * SCAN_THREADS + threadIdx.x;
        if (idx < n) acc += in[idx];
    }
    int tot;
    block_exclusive_scan(acc, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int32_t *in, int32_t *out, int64_t n,
                                                             const int32_t *block_off) {
    // thread t owns SCAN_ITEMS consecutive elements so that the scan order is the array order
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        acc += v[k];
    }
    int tot;
    int pre = block_exclusive_scan(acc, &tot) + (block_off ? block_off[blockIdx.x] : 0);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = pre;
        pre += v[k];
        if (base + k == n - 1) out[n] = pre;
    }
}

__global__ void k_set_i32(int32_t *p, int32_t v) { *p = v; }

int scan_exclusive_i32(const int32_t *in, int32_t *out, int64_t n, int64_t *total_host) {
    hipStream_t s = ctx().stream;
    if (n <= 0) {
        hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, s, out, 0);
        CSX_LAUNCH_CHECK();
        if (total_host) *total_host = 0;
        return CSX_OK;
    }
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    int32_t *sums = nullptr;
    if (nb > 1) {
        CSX_TRY(dalloc(&sums, (size_t)nb + 1));
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, n, sums);
        CSX_LAUNCH_CHECK();
        int st = scan_exclusive_i32(sums, sums, nb, nullptr);
        if (st != CSX_OK) {
            dfree(sums);
            return st;
        }
    }
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, n, sums);
    CSX_LAUNCH_CHECK();
    if (total_host) {
        int32_t t = 0;
        CSX_HIP(hipMemcpyAsync(&t, out + n, sizeof t, hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        *total_host = t;
    } else if (sums) {
        CSX_HIP(hipStreamSynchronize(s));  // sums is freed below
    }
    dfree(sums);
    return CSX_OK;
}

// ------------------------------------------------------ column expansion ----
__global__ __launch_bounds__(256) void k_expand_columns(const int32_t *Ap, int32_t n, int32_t *col) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t j = wave; j < n; j += nwaves) {
        int32_t b = Ap[j], e = Ap[j + 1];
        for (int32_t p = b + lane; p < e; p += 64) col[p] = (int32_t)j;
    }
}

int expand_columns(const int32_t *Ap, int32_t n, int32_t nnz, int32_t *col) {
    if (n == 0 || nnz == 0) return CSX_OK;
    int64_t blocks = ((int64_t)n + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_expand_columns, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, Ap, n, col);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

// ---------------------------------------------------------- radix sort ----
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ROUNDS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;  // records per workgroup
constexpr int RS_BINS = 256;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t *key, int64_t count, int shift,
                                                        uint32_t nblocks, int32_t *hist) {
    __shared__ int h[RS_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    if (base + RS_TILE <= count && (reinterpret_cast<uintptr_t>(key) & 15) == 0) {
        // full tile: 16-byte loads, a wave instruction covers 1 KiB (the order of the keys does not matter here)
        typedef uint32_t u32x4h __attribute__((ext_vector_type(4)));
        const u32x4h *k4 = reinterpret_cast<const u32x4h *>(key + base);
#pragma unroll
        for (int r = 0; r < RS_ROUNDS / 4; r++) {
            const u32x4h v = k4[r * RS_THREADS + threadIdx.x];
            atomicAdd(&h[(v.x >> shift) & (RS_BINS - 1)], 1);
            atomicAdd(&h[(v.y >> shift) & (RS_BINS - 1)], 1);
            atomicAdd(&h[(v.z >> shift) & (RS_BINS - 1)], 1);
            atomicAdd(&h[(v.w >> shift) & (RS_BINS - 1)], 1);
        }
    } else {
#pragma unroll 4
        for (int r = 0; r < RS_ROUNDS; r++) {
            int64_t idx = base + (int64_t)r * RS_THREADS + threadIdx.x;
            if (idx < count) atomicAdd(&h[(key[idx] >> shift) & (RS_BINS - 1)], 1);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one workgroup tile (4096 records).  Ranks follow (wave, round, lane) = source
// order.  Records are first placed in LDS in their sorted order inside the tile, then written out
// position by position: consecutive threads write consecutive slots of a bucket, so the stores are
// runs of whole 64/128-byte pieces instead of 4/8-byte singles.
template <bool HAS_A, bool HAS_V, bool WRITE_KEY>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t *key, const uint32_t *a, const double *v,
                                                           int64_t count, int shift, uint32_t nblocks,
                                                           const int32_t *goff, uint32_t *okey, uint32_t *oa,
                                                           double *ov, int getenv_flat) {
    __shared__ int wh[RS_WAVES][RS_BINS];
    __shared__ int gbase[RS_BINS];  // global slot of a bucket's first record minus its local start
    __shared__ int wsum[RS_WAVES];
    __shared__ uint32_t s_key[RS_TILE];
    __shared__ uint32_t s_a[HAS_A ? RS_TILE : 1];
    __shared__ double s_v[HAS_V ? RS_TILE : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) wh[k][threadIdx.x] = 0;
    __syncthreads();
    // Workgroups are dealt to the 8 XCDs round-robin; give each XCD a contiguous range of tiles, so that
    // neighbouring tiles -- whose runs in a bucket are adjacent in memory -- meet in the same L2 and the
    // partial lines at the run boundaries are merged there.
    uint32_t tile = blockIdx.x;
    if (!getenv_flat) {
        const uint32_t q = nblocks >> 3, rem = nblocks & 7u, x = blockIdx.x & 7u, kk = blockIdx.x >> 3;
        tile = x * q + (x < rem ? x : rem) + kk;
    }
    // wave w owns the contiguous sub-tile [w*64*ROUNDS, (w+1)*64*ROUNDS) of this workgroup's tile
    const int64_t tbase = (int64_t)tile * RS_TILE;
    const int64_t wbase = tbase + (int64_t)w * 64 * RS_ROUNDS;
    // the whole sub-tile goes to registers first: every load of the tile is in flight before the
    // (latency-bound) ranking rounds start, instead of one exposed round trip per round
    uint32_t kreg[RS_ROUNDS], areg[HAS_A ? RS_ROUNDS : 1];
    double vreg[HAS_V ? RS_ROUNDS : 1];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t idx = wbase + r * 64 + lane;
        const int64_t cl = idx < count ? idx : count - 1;   // clamped: count > 0
        kreg[r] = key[cl];
        if (HAS_A) areg[r] = a[cl];
        if (HAS_V) vreg[r] = v[cl];
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t idx = wbase + r * 64 + lane;
        if (idx < count) atomicAdd(&wh[w][(kreg[r] >> shift) & (RS_BINS - 1)], 1);
    }
    __syncthreads();
    {
        // digit d = threadIdx.x: local start of the bucket inside the tile = exclusive scan over digits
        const int d = threadIdx.x;
        int tot = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) tot += wh[k][d];
        int inc = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            int t = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += t;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int off = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++)
            if (k < w) off += wsum[k];
        int run = off + inc - tot;  // local start of bucket d
        gbase[d] = goff[(size_t)d * nblocks + tile] - run;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) {
            int c = wh[k][d];
            wh[k][d] = run;
            run += c;
        }
    }
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        int64_t idx = wbase + r * 64 + lane;
        bool valid = idx < count;
        uint32_t k = kreg[r];
        uint32_t d = (k >> shift) & (RS_BINS - 1);
        // lanes holding the same digit (multi-split by ballots, one per digit bit)
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        int rank = __popcll(peers & lt);
        int pos = 0;
        if (valid) pos = wh[w][d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][d] = pos + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            s_key[pos] = k;
            if (HAS_A) s_a[pos] = areg[r];
            if (HAS_V) s_v[pos] = vreg[r];
        }
    }
    __syncthreads();
    const int tcount = (int)((count - tbase) < RS_TILE ? (count - tbase) : RS_TILE);
    for (int i = threadIdx.x; i < tcount; i += RS_THREADS) {
        const uint32_t k = s_key[i];
        const int64_t g = (int64_t)gbase[(k >> shift) & (RS_BINS - 1)] + i;
        if (WRITE_KEY) okey[g] = k;
        if (HAS_A) oa[g] = s_a[i];
        if (HAS_V) ov[g] = s_v[i];
    }
}

template <bool HAS_A, bool HAS_V>
static int launch_scatter(bool write_key, dim3 grid, hipStream_t s, const uint32_t *key, const uint32_t *a,
                          const double *v, int64_t count, int shift, uint32_t nblocks, const int32_t *goff,
                          uint32_t *okey, uint32_t *oa, double *ov) {
    const int flat = ablation_env("CSX_SORT_FLAT") ? 1 : 0;
    if (write_key)
        hipLaunchKernelGGL((k_rs_scatter<HAS_A, HAS_V, true>), grid, dim3(RS_THREADS), 0, s, key, a, v, count, shift,
                           nblocks, goff, okey, oa, ov, flat);
    else
        hipLaunchKernelGGL((k_rs_scatter<HAS_A, HAS_V, false>), grid, dim3(RS_THREADS), 0, s, key, a, v, count, shift,
                           nblocks, goff, okey, oa, ov, flat);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

int stable_sort_by_key(const uint32_t *key, const uint32_t *a, const double *v, int64_t count, uint32_t key_limit,
                       uint32_t *out_key, uint32_t *out_a, double *out_v) {
    if (count <= 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (unsigned long long)key_limit) bits++;
    const int passes = (bits + 7) / 8;
    const uint32_t nblocks = (uint32_t)((count + RS_TILE - 1) / RS_TILE);
    const bool has_a = a != nullptr, has_v = v != nullptr;

    int32_t *hist = nullptr;
    uint32_t *tk[2] = {nullptr, nullptr}, *ta[2] = {nullptr, nullptr};
    double *tv[2] = {nullptr, nullptr};
    int st = dalloc(&hist, (size_t)RS_BINS * nblocks + 1);
    const int ntmp = passes > 2 ? 2 : passes - 1;
    for (int t = 0; t < ntmp && st == CSX_OK; t++) {
        st = dalloc(&tk[t], (size_t)count);
        if (st == CSX_OK && has_a) st = dalloc(&ta[t], (size_t)count);
        if (st == CSX_OK && has_v) st = dalloc(&tv[t], (size_t)count);
    }
    const uint32_t *ik = key, *ia = a;
    const double *iv = v;
    for (int ps = 0; ps < passes && st == CSX_OK; ps++) {
        const bool last = ps == passes - 1;
        uint32_t *ok = last ? out_key : tk[ps & 1];
        uint32_t *oa = last ? out_a : ta[ps & 1];
        double *ov = last ? out_v : tv[ps & 1];
        const int shift = ps * 8;
        hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(RS_THREADS), 0, s, ik, count, shift, nblocks, hist);
        st = hipGetLastError() == hipSuccess ? CSX_OK : CSX_ERUNTIME;
        if (st == CSX_OK) st = scan_exclusive_i32(hist, hist, (int64_t)RS_BINS * nblocks, nullptr);
        if (st != CSX_OK) break;
        const bool wk = ok != nullptr;
        if (has_a && has_v)
            st = launch_scatter<true, true>(wk, dim3(nblocks), s, ik, ia, iv, count, shift, nblocks, hist, ok, oa, ov);
        else if (has_a)
            st = launch_scatter<true, false>(wk, dim3(nblocks), s, ik, ia, iv, count, shift, nblocks, hist, ok, oa, ov);
        else if (has_v)
            st = launch_scatter<false, true>(wk, dim3(nblocks), s, ik, ia, iv, count, shift, nblocks, hist, ok, oa, ov);
        else
            st = launch_scatter<false, false>(wk, dim3(nblocks), s, ik, ia, iv, count, shift, nblocks, hist, ok, oa, ov);
        ik = ok;
        ia = oa;
        iv = ov;
    }
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    if (st == CSX_ERUNTIME) set_error("stable_sort_by_key: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    dfree(hist);
    for (int t = 0; t < 2; t++) {
        dfree(tk[t]);
        dfree(ta[t]);
        dfree(tv[t]);
    }
    return st;
}

// ----------------------------------------------------------- boundaries ----
// position q (0..count) closes the keys (sorted_key[q-1], sorted_key[q]]: ptr[r] = q for those r.
// Four positions per thread through one 16-byte load (boundaries are rare, the key stream is the cost).
__global__ __launch_bounds__(256) void k_boundaries(const uint32_t *skey, int64_t count, int32_t nkeys, int32_t *ptr) {
    typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
    const int64_t q0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (q0 > count) return;
    int64_t prev = (q0 == 0) ? -1 : (int64_t)skey[q0 - 1];
    if (q0 + 4 <= count && (reinterpret_cast<uintptr_t>(skey) & 15) == 0) {
        const u32x4b v = *reinterpret_cast<const u32x4b *>(skey + q0);
        const uint32_t k4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t cur = (int64_t)k4[k];
            for (int64_t r = prev + 1; r <= cur; r++) ptr[r] = (int32_t)(q0 + k);
            prev = cur;
        }
        return;
    }
    const int64_t qend = q0 + 4 <= count ? q0 + 4 : count + 1;   // the last, partial group closes with q = count
    for (int64_t q = q0; q < qend; q++) {
        const int64_t cur = (q == count) ? (int64_t)nkeys : (int64_t)skey[q];
        for (int64_t r = prev + 1; r <= cur; r++) ptr[r] = (int32_t)q;
        prev = cur;
    }
}

int boundaries_from_sorted(const uint32_t *sorted_key, int64_t count, int32_t nkeys, int32_t *ptr) {
    const int64_t threads = count / 4 + 1;
    const int64_t blocks = (threads + 255) / 256;
    hipLaunchKernelGGL(k_boundaries, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, sorted_key, count, nkeys, ptr);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_cumsum(csx_handle_t hp, csx_handle_t hc, int64_t n, int64_t *total) {
    CSX_TRY(require_ready());
    Vec *p = ivec(hp), *c = ivec(hc);
    if (!p || !c || n < 0 || p->len < n + 1 || c->len < n) return CSX_EINVAL;
    CSX_TRY(scan_exclusive_i32((const int32_t *)c->d, (int32_t *)p->d, n, total));
    if (n)
        CSX_HIP(hipMemcpyAsync(c->d, p->d, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx().stream));
    return CSX_OK;
}
