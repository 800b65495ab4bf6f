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
    }
    dfree(sums);   // (no wait: the pool's blocks are used on the context's stream only, whoever gets this one next queues behind the scan)
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
constexpr int RS_ROUNDS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;  // records per workgroup
constexpr int RS_BINS = 256;

// Between two passes a record with both payloads travels as its key (4 bytes, an array of its own: the next pass's
// histogram reads nothing else) and (a, v) packed into 12 bytes: a tile's run in a bucket is then 16 x 12 = 192
// contiguous bytes instead of 64 + 128 in two places, and runs below 128 bytes are what slows a scatter down
// (tools/ubench/scatter_runs.hip: 64-byte runs 81 %, 32-byte runs 49 % of the rate of 128-byte runs).
struct __attribute__((packed, aligned(4))) Pay {
    uint32_t a, vlo, vhi;
};

// hist[tile][digit]: one coalesced 1 KiB row per tile
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t *key, int64_t count, int shift, uint32_t mask,
                                                        int32_t *hist) {
    __shared__ int h[RS_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    if (base + RS_TILE <= count && (reinterpret_cast<uintptr_t>(key) & 15) == 0) {
        // full tile: 16-byte loads, a wave instruction covers 1 KiB (the order of the keys does not matter here)
        typedef uint32_t u32x4h __attribute__((ext_vector_type(4)));
        const u32x4h *k4 = reinterpret_cast<const u32x4h *>(key + base);
        u32x4h v[RS_ROUNDS / 4];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS / 4; r++) v[r] = k4[r * RS_THREADS + threadIdx.x];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS / 4; r++) {
            atomicAdd(&h[(v[r].x >> shift) & mask], 1);
            atomicAdd(&h[(v[r].y >> shift) & mask], 1);
            atomicAdd(&h[(v[r].z >> shift) & mask], 1);
            atomicAdd(&h[(v[r].w >> shift) & mask], 1);
        }
    } else {
#pragma unroll 4
        for (int r = 0; r < RS_ROUNDS; r++) {
            int64_t idx = base + (int64_t)r * RS_THREADS + threadIdx.x;
            if (idx < count) atomicAdd(&h[(key[idx] >> shift) & mask], 1);
        }
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * RS_BINS + threadIdx.x] = h[threadIdx.x];
}

// the same for 16-bit keys (the short-key mode of a transpose, see stable_sort_by_key_ex)
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist16(const uint16_t *key, int64_t count, int shift, uint32_t mask,
                                                          int32_t *hist) {
    __shared__ int h[RS_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    if (base + RS_TILE <= count && (reinterpret_cast<uintptr_t>(key) & 15) == 0) {
        typedef uint32_t u32x2h __attribute__((ext_vector_type(2)));
        const u32x2h *k2 = reinterpret_cast<const u32x2h *>(key + base);      // four keys per load
        u32x2h v[RS_ROUNDS / 4];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS / 4; r++) v[r] = k2[r * RS_THREADS + threadIdx.x];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS / 4; r++) {
            const uint32_t w2[2] = {v[r].x, v[r].y};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                atomicAdd(&h[((w2[q] & 0xffffu) >> shift) & mask], 1);
                atomicAdd(&h[((w2[q] >> 16) >> shift) & mask], 1);
            }
        }
    } else {
#pragma unroll 4
        for (int r = 0; r < RS_ROUNDS; r++) {
            int64_t idx = base + (int64_t)r * RS_THREADS + threadIdx.x;
            if (idx < count) atomicAdd(&h[((uint32_t)key[idx] >> shift) & mask], 1);
        }
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * RS_BINS + threadIdx.x] = h[threadIdx.x];
}

// hist[tile][digit] -> goff[tile][digit] = records with a smaller digit + records with this digit in earlier tiles,
// i.e. the exclusive scan in (digit, tile) order, done on the tile-major matrix in three small steps: sums over
// chunks of tiles, one workgroup that scans the chunk sums in (digit, chunk) order, running sums inside a chunk.
__global__ __launch_bounds__(RS_BINS) void k_rs_colsum(const int32_t *hist, uint32_t nblocks, uint32_t chunk,
                                                       int32_t *part) {
    const uint32_t t0 = blockIdx.x * chunk, t1 = min(t0 + chunk, nblocks);
    int s = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; t++) s += hist[(size_t)t * RS_BINS + threadIdx.x];
    part[(size_t)blockIdx.x * RS_BINS + threadIdx.x] = s;
}

// one workgroup, thread = (segment of chunks, digit): running sums over the chunks per digit, and dbase[d] = records
// with a smaller digit
constexpr int RS_SEGS = 4;
__global__ __launch_bounds__(RS_BINS * RS_SEGS) void k_rs_chunkscan(int32_t *part, uint32_t nchunks, int32_t *dbase) {
    __shared__ int segsum[RS_SEGS][RS_BINS];
    const int d = threadIdx.x & (RS_BINS - 1), sg = threadIdx.x / RS_BINS;
    const uint32_t per = (nchunks + RS_SEGS - 1) / RS_SEGS;
    const uint32_t c0 = min(sg * per, nchunks), c1 = min(c0 + per, nchunks);
    int run = 0;
#pragma unroll 8
    for (uint32_t c = c0; c < c1; c++) run += part[(size_t)c * RS_BINS + d];
    segsum[sg][d] = run;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < RS_SEGS; k++) {
        const int v = segsum[k][d];
        if (k < sg) before += v;
        total += v;
    }
    run = before;
#pragma unroll 8
    for (uint32_t c = c0; c < c1; c++) {
        const int v = part[(size_t)c * RS_BINS + d];
        part[(size_t)c * RS_BINS + d] = run;
        run += v;
    }
    // exclusive scan of the digit totals by the first segment's threads (wave scans + 4 wave sums)
    __shared__ int wtot[RS_BINS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = 0;
    if (sg == 0) {
        inc = wave_inclusive_scan(total, lane);
        if (lane == 63) wtot[w] = inc;
    }
    __syncthreads();
    if (sg == 0) {
        int off = 0;
#pragma unroll
        for (int k = 0; k < RS_BINS / 64; k++)
            if (k < w) off += wtot[k];
        dbase[d] = off + inc - total;
    }
}

__global__ __launch_bounds__(RS_BINS) void k_rs_tileprefix(int32_t *hist, const int32_t *part, const int32_t *dbase,
                                                           uint32_t nblocks, uint32_t chunk) {
    const uint32_t t0 = blockIdx.x * chunk, t1 = min(t0 + chunk, nblocks);
    int run = part[(size_t)blockIdx.x * RS_BINS + threadIdx.x] + dbase[threadIdx.x];
#pragma unroll 8
    for (uint32_t t = t0; t < t1; t++) {
        const int v = hist[(size_t)t * RS_BINS + threadIdx.x];
        hist[(size_t)t * RS_BINS + threadIdx.x] = run;
        run += v;
    }
}

// cs_transpose's first pass needs the column of every record.  One 512-byte descriptor per tile, written before the
// pass and read by the tile with a single load that depends on nothing: word 0 = j0, the column that holds the tile's
// first position (largest j with Ap[j] <= tile base); word 1 = K, the number of columns that start strictly inside the
// tile; then the first RS_DESC_STARTS of those starts, relative to the tile base, 16 bits each.
constexpr int RS_DESC_WORDS = 128;
constexpr int RS_DESC_STARTS = (RS_DESC_WORDS - 2) * 2;

__global__ void k_rs_tiledesc_head(const int32_t *Ap, int32_t n, int64_t count, uint32_t nblocks, uint32_t *desc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks) return;
    const int64_t tb = (int64_t)t * RS_TILE;
    int64_t te = tb + RS_TILE - 1;
    if (te > count - 1) te = count - 1;
    int32_t lo = 0, hi = n - 1;   // largest j with Ap[j] <= tb
    while (lo < hi) {
        const int32_t mid = lo + (hi - lo + 1) / 2;
        if ((int64_t)Ap[mid] <= tb) lo = mid;
        else hi = mid - 1;
    }
    const int32_t j0 = lo;
    hi = n - 1;                   // largest j with Ap[j] <= te (>= j0)
    while (lo < hi) {
        const int32_t mid = lo + (hi - lo + 1) / 2;
        if ((int64_t)Ap[mid] <= te) lo = mid;
        else hi = mid - 1;
    }
    desc[(size_t)t * RS_DESC_WORDS] = (uint32_t)j0;
    desc[(size_t)t * RS_DESC_WORDS + 1] = (uint32_t)(lo - j0);
}

__global__ void k_rs_tiledesc_starts(const int32_t *Ap, int32_t n, int64_t count, uint32_t *desc) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t p = Ap[j];
    if (p >= count) return;                       // empty columns at the end start in no tile
    const int64_t t = p / RS_TILE;
    const uint32_t rel = (uint32_t)(p - t * RS_TILE);
    if (rel == 0) return;                         // at the tile base: that is j0 or an empty column before it
    const int64_t k = j - (int64_t)desc[(size_t)t * RS_DESC_WORDS] - 1;
    if (k < RS_DESC_STARTS) reinterpret_cast<uint16_t *>(desc + (size_t)t * RS_DESC_WORDS + 2)[k] = (uint16_t)rel;
}

struct RsArgs {
    const uint32_t *key, *a;   // inputs: keys; 32-bit payload as an array (or none / expanded / inside pay)
    const double *v;           //         64-bit payload as an array
    const Pay *pay;            //         (a, v) records of the previous pass
    uint32_t *okey, *oa;       // outputs (okey may be null)
    double *ov;
    Pay *opay;
    const int32_t *goff;       // [tile][digit] global slot of the tile's first record with that digit
    const int32_t *xp;         // not null: a = column of the record's position under the column pointers xp
    const uint32_t *desc;      //           and the per-tile descriptors for it
    int32_t *optr;             // not null (last pass): optr[key] = min(position of a record with that key)
    int64_t count;
    int shift;
    uint32_t mask, nblocks;
    int flat;
    // short-key mode (KM != 0): 16-bit keys between the passes, the first digit riding in the top bits of `a`
    const uint16_t *key16;
    uint16_t *okey16;
    int k16_w0;                // bits of the first digit
};

// Stable scatter of one workgroup tile (4096 records).  Ranks follow (wave, round, lane) = source
// order.  Records are first placed in LDS in their sorted order inside the tile, then written out
// position by position: consecutive threads write consecutive slots of a bucket, so the stores are
// runs of whole 64/128-byte pieces instead of 4/8-byte singles.
// KM (short-key mode of a three-pass sort with derived columns -- a transpose): 0 = off; 1 = first pass: 32-bit keys in,
// key >> w0 out as 16 bits, the first digit packed into the top w0 bits of the column; 2 = middle pass: 16-bit keys in and
// out; 3 = last pass: 16-bit keys in, the column unpacked, the full key rebuilt for the pointer array.  Saves 2 bytes of
// every key read and written after the first digit is spent: 12 bytes per record over the three passes and their histograms.
template <bool HAS_A, bool HAS_V, bool IN_AOS, bool OUT_AOS, int KM = 0>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const RsArgs g) {
    static_assert(!(IN_AOS || OUT_AOS) || (HAS_A && HAS_V), "packed records carry both payloads");
    __shared__ int wh[RS_WAVES][RS_BINS];
    __shared__ int gbase[RS_BINS];  // global slot of a bucket's first record minus its local start
    __shared__ uint32_t s_key[RS_TILE];
    int *const wsum = reinterpret_cast<int *>(s_key);   // (the scan's wave totals: s_key is idle between the counting and the placement)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[(HAS_A || HAS_V) ? (HAS_A ? 4 : 0) * RS_TILE + (HAS_V ? 8 : 0) * RS_TILE : 16];
    uint32_t *const s_a = reinterpret_cast<uint32_t *>(s_raw);
    double *const s_v = reinterpret_cast<double *>(s_raw + (HAS_A ? 4 * RS_TILE : 0));
    Pay *const s_pay = reinterpret_cast<Pay *>(s_raw);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t count = g.count;
    const int shift = g.shift;
    const uint32_t mask = g.mask, nblocks = g.nblocks;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) wh[k][threadIdx.x] = 0;
    // Workgroups are dealt to the 8 XCDs round-robin; give each XCD a contiguous range of tiles, so that
    // neighbouring tiles -- whose runs in a bucket are adjacent in memory -- meet in the same L2 and the
    // partial lines at the run boundaries are merged there.
    uint32_t tile = blockIdx.x;
    if (!g.flat) {
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
    // cs_transpose's first pass: the tile's descriptor is requested with the records
    const bool expand = HAS_A && !IN_AOS && g.xp != nullptr;
    uint32_t d0 = 0, d1 = 0;
    if (expand) {
        d0 = g.desc[(size_t)tile * RS_DESC_WORDS + lane];
        d1 = g.desc[(size_t)tile * RS_DESC_WORDS + 64 + lane];
    }
    // keys first, payloads after: loads return in order, so the counting and ranking below wait for the keys only and
    // the payloads keep arriving meanwhile (the barriers of this kernel order LDS, not global memory: lds_barrier)
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t idx = wbase + r * 64 + lane;
        kreg[r] = KM >= 2 ? (uint32_t)g.key16[idx < count ? idx : count - 1] : g.key[idx < count ? idx : count - 1];   // clamped: count > 0
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t idx = wbase + r * 64 + lane;
        const int64_t cl = idx < count ? idx : count - 1;
        if (IN_AOS) {
            const Pay pr = g.pay[cl];
            areg[r] = pr.a;
            vreg[r] = __hiloint2double((int)pr.vhi, (int)pr.vlo);
        } else {
            if (HAS_A && !g.xp) areg[r] = g.a[cl];
            if (HAS_V) vreg[r] = g.v[cl];
        }
    }
    if (expand) {
        // the 32-bit payload is the record's column
        const int32_t j0 = __builtin_amdgcn_readlane((int)d0, 0), K = __builtin_amdgcn_readlane((int)d0, 1);
        const int32_t j1 = j0 + K;
        if (K <= RS_DESC_STARTS) {
            // every start is in the descriptor: wave w paints the positions of the columns w, w + 4, ... (and of j0,
            // which holds the positions before the first start) with their column, then each record reads its own
            uint32_t *const s_col = reinterpret_cast<uint32_t *>(s_raw);   // free until the ranking rounds
            const int wu = __builtin_amdgcn_readfirstlane(w);
            for (int c = wu - 1; c < K; c += RS_WAVES) {
                int b = 0, e = RS_TILE;
                if (c >= 0) {
                    const int h = c + 4;
                    const uint32_t wd = (h >> 1) < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)d0, (h >> 1) & 63)
                                                      : (uint32_t)__builtin_amdgcn_readlane((int)d1, (h >> 1) & 63);
                    b = (int)((wd >> (16 * (h & 1))) & 0xffffu);
                }
                if (c + 1 < K) {
                    const int h = c + 5;
                    const uint32_t wd = (h >> 1) < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)d0, (h >> 1) & 63)
                                                      : (uint32_t)__builtin_amdgcn_readlane((int)d1, (h >> 1) & 63);
                    e = (int)((wd >> (16 * (h & 1))) & 0xffffu);
                }
                for (int q = b + lane; q < e; q += 64) s_col[q] = (uint32_t)(j0 + 1 + c);
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < RS_ROUNDS; r++) areg[r] = s_col[w * 64 * RS_ROUNDS + r * 64 + lane];
        } else if (K <= RS_TILE) {
            // many short columns: their starts are listed in LDS (s_key is free until the ranking rounds) and every
            // record counts the starts at or before its position, 16 searches side by side
            for (int k = threadIdx.x; k < K; k += RS_THREADS) s_key[k] = (uint32_t)((int64_t)g.xp[j0 + 1 + k] - tbase);
            lds_barrier();
            int lo[RS_ROUNDS];
#pragma unroll
            for (int r = 0; r < RS_ROUNDS; r++) lo[r] = 0;
            int top = 1;
            while (top * 2 <= K) top *= 2;
            for (int half = top; half > 0; half >>= 1) {
#pragma unroll
                for (int r = 0; r < RS_ROUNDS; r++) {
                    const uint32_t li = (uint32_t)(w * 64 * RS_ROUNDS + r * 64 + lane);
                    const int mid = lo[r] + half;
                    if (mid <= K && s_key[mid - 1] <= li) lo[r] = mid;
                }
            }
#pragma unroll
            for (int r = 0; r < RS_ROUNDS; r++) areg[r] = (uint32_t)(j0 + lo[r]);
        } else {  // a stretch of empty columns: search the pointer array itself
#pragma unroll 1
            for (int r = 0; r < RS_ROUNDS; r++) {
                const int64_t pos = wbase + r * 64 + lane;
                int32_t lo = j0, hi = j1;
                while (lo < hi) {
                    const int32_t mid = lo + (hi - lo + 1) / 2;
                    if ((int64_t)g.xp[mid] <= pos) lo = mid;
                    else hi = mid - 1;
                }
                areg[r] = (uint32_t)lo;
            }
        }
    }
    if (KM == 1) {
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) areg[r] |= (kreg[r] & mask) << (32 - g.k16_w0);   // the digit this pass spends
    }
    lds_barrier();
    // Counting and ranking share one multi-split per round: the lanes of a round that hold the same digit are found by
    // ballots (one per digit bit); the first of them adds the whole group to the wave's histogram, every lane keeps its
    // rank in the group for the placement below.  (One LDS atomic per lane instead -- the earlier form -- serialises when a
    // round's keys are clustered: the last pass of a transpose sees runs of 64 equal rows, 64 lanes on one counter.)
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rk[RS_ROUNDS];   // rank in the group | group size << 8
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const int64_t idx = wbase + r * 64 + lane;
        const bool valid = idx < count;
        const uint32_t d = (kreg[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt), cnt = __popcll(peers);
        rk[r] = rank | (cnt << 8);
        if (valid && rank == 0) atomicAdd(&wh[w][d], cnt);
    }
    lds_barrier();
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
        lds_barrier();
        int off = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++)
            if (k < w) off += wsum[k];
        int run = off + inc - tot;  // local start of bucket d
        gbase[d] = g.goff[(size_t)tile * RS_BINS + d] - run;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) {
            int c = wh[k][d];
            wh[k][d] = run;
            run += c;
        }
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        int64_t idx = wbase + r * 64 + lane;
        bool valid = idx < count;
        uint32_t k = kreg[r];
        uint32_t d = (k >> shift) & mask;
        const int rank = rk[r] & 255;
        int pos = 0;
        if (valid) pos = wh[w][d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wh[w][d] = pos + (rk[r] >> 8);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            s_key[pos] = k;
            if (OUT_AOS) {
                Pay pr;
                pr.a = areg[r];
                pr.vlo = (uint32_t)__double2loint(vreg[r]);
                pr.vhi = (uint32_t)__double2hiint(vreg[r]);
                s_pay[pos] = pr;
            } else {
                if (HAS_A) s_a[pos] = areg[r];
                if (HAS_V) s_v[pos] = vreg[r];
            }
        }
    }
    lds_barrier();
    const int tcount = (int)((count - tbase) < RS_TILE ? (count - tbase) : RS_TILE);
    for (int i = threadIdx.x; i < tcount; i += RS_THREADS) {
        const uint32_t k = s_key[i];
        const int64_t gp = (int64_t)gbase[(k >> shift) & mask] + i;
        if (KM == 1) g.okey16[gp] = (uint16_t)(k >> g.k16_w0);
        else if (KM == 2) g.okey16[gp] = (uint16_t)k;
        else if (KM == 0 && g.okey) g.okey[gp] = k;
        if (OUT_AOS) {
            g.opay[gp] = s_pay[i];
        } else if (KM == 3) {
            // the column without the packed digit; the full key = the 16 bits that travelled and that digit
            const int ps = 32 - g.k16_w0;
            const uint32_t a = s_a[i], full = (k << g.k16_w0) | (a >> ps);
            g.oa[gp] = a & ((1u << ps) - 1u);
            if (HAS_V) g.ov[gp] = s_v[i];
            if (g.optr && (i == 0 || s_key[i - 1] != k || (s_a[i - 1] >> ps) != (a >> ps))) atomicMin(&g.optr[full], (int32_t)gp);
        } else {
            if (HAS_A) g.oa[gp] = s_a[i];
            if (HAS_V) g.ov[gp] = s_v[i];
        }
        // last pass: equal keys are neighbours in a bucket; the first of each group in this tile proposes its slot
        if (KM == 0 && g.optr && (i == 0 || s_key[i - 1] != k)) atomicMin(&g.optr[k], (int32_t)gp);
    }
}

template <bool HAS_A, bool HAS_V>
static void launch_scatter(bool in_aos, bool out_aos, dim3 grid, hipStream_t s, const RsArgs &g) {
    if constexpr (HAS_A && HAS_V) {
        if (in_aos && out_aos) hipLaunchKernelGGL((k_rs_scatter<true, true, true, true>), grid, dim3(RS_THREADS), 0, s, g);
        else if (in_aos) hipLaunchKernelGGL((k_rs_scatter<true, true, true, false>), grid, dim3(RS_THREADS), 0, s, g);
        else if (out_aos) hipLaunchKernelGGL((k_rs_scatter<true, true, false, true>), grid, dim3(RS_THREADS), 0, s, g);
        else hipLaunchKernelGGL((k_rs_scatter<true, true, false, false>), grid, dim3(RS_THREADS), 0, s, g);
    } else {
        hipLaunchKernelGGL((k_rs_scatter<HAS_A, HAS_V, false, false>), grid, dim3(RS_THREADS), 0, s, g);
    }
}

// ---- reverse running minimum: p[r] = min(p[r], ..., p[n-1]) (turns "first slot of each key" into column pointers) ----
constexpr int SM_ITEMS = 8;
constexpr int SM_TILE = SCAN_THREADS * SM_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void k_sufmin_block(const int32_t *p, int64_t n, int32_t *bm) {
    __shared__ int wmin[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SM_TILE;
    int m = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < SM_ITEMS; k++) {
        const int64_t idx = base + (int64_t)k * SCAN_THREADS + threadIdx.x;
        if (idx < n) m = min(m, p[idx]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = min(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < SCAN_THREADS / 64; k++) m = min(m, wmin[k]);
        bm[blockIdx.x] = m;
    }
}

// bm_suffix: already the reverse running minimum of the block minima (or null for a single block)
__global__ __launch_bounds__(SCAN_THREADS) void k_sufmin_apply(int32_t *p, int64_t n, const int32_t *bm_suffix,
                                                               int64_t nb) {
    __shared__ int tmin[SCAN_THREADS];
    const int64_t base = (int64_t)blockIdx.x * SM_TILE + (int64_t)threadIdx.x * SM_ITEMS;
    int v[SM_ITEMS];
    int m = 0x7fffffff;
#pragma unroll
    for (int k = SM_ITEMS - 1; k >= 0; k--) {
        v[k] = (base + k < n) ? p[base + k] : 0x7fffffff;
        m = min(m, v[k]);
    }
    tmin[threadIdx.x] = m;
    __syncthreads();
    // minimum over the threads after this one: suffix scan over 256 values (Hillis-Steele in LDS)
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {
        const int o = (threadIdx.x + d < SCAN_THREADS) ? tmin[threadIdx.x + d] : 0x7fffffff;
        __syncthreads();
        tmin[threadIdx.x] = min(tmin[threadIdx.x], o);
        __syncthreads();
    }
    int carry = (bm_suffix && (int64_t)blockIdx.x + 1 < nb) ? bm_suffix[blockIdx.x + 1] : 0x7fffffff;
    if (threadIdx.x + 1 < SCAN_THREADS) carry = min(carry, tmin[threadIdx.x + 1]);
#pragma unroll
    for (int k = SM_ITEMS - 1; k >= 0; k--) {
        carry = min(carry, v[k]);
        if (base + k < n) p[base + k] = carry;
    }
}

int suffix_min_i32(int32_t *p, int64_t n) {
    if (n <= 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    const int64_t nb = (n + SM_TILE - 1) / SM_TILE;
    int32_t *bm = nullptr;
    if (nb > 1) {
        CSX_TRY(dalloc(&bm, (size_t)nb));
        hipLaunchKernelGGL(k_sufmin_block, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, p, n, bm);
        const int st = suffix_min_i32(bm, nb);
        if (st != CSX_OK) {
            dfree(bm);
            return st;
        }
    }
    hipLaunchKernelGGL(k_sufmin_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, p, n, bm, nb);
    int st = hipGetLastError() == hipSuccess ? CSX_OK : CSX_ERUNTIME;
    if (bm) {
        if (hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;  // bm is freed below
        dfree(bm);
    }
    return st;
}

__global__ void k_ptr_init(int32_t *ptr, int32_t nkeys, int32_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nkeys) ptr[i] = 0x7fffffff;
    else if (i == nkeys) ptr[i] = count;
}

int stable_sort_by_key_ex(const uint32_t *key, const uint32_t *a, const double *v, int64_t count, uint32_t key_limit,
                          uint32_t *out_key, uint32_t *out_a, double *out_v, const SortExtra *ex) {
    hipStream_t s = ctx().stream;
    const int32_t *xp = ex ? ex->expand_ptr : nullptr;
    int32_t *optr = ex ? ex->out_ptr : nullptr;
    if (optr) {
        const int64_t nk = (int64_t)ex->nkeys + 1;
        hipLaunchKernelGGL(k_ptr_init, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, optr, ex->nkeys,
                           (int32_t)(count > 0 ? count : 0));
        CSX_LAUNCH_CHECK();
    }
    if (count <= 0) return optr ? suffix_min_i32(optr, (int64_t)ex->nkeys + 1) : CSX_OK;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (unsigned long long)key_limit) bits++;
    const int passes = (bits + 7) / 8;
    const uint32_t nblocks = (uint32_t)((count + RS_TILE - 1) / RS_TILE);
    uint32_t chunk = (nblocks + 1023) / 1024;
    if (chunk < 64) chunk = 64;
    const uint32_t nchunks = (nblocks + chunk - 1) / chunk;
    const bool has_a = a != nullptr || xp != nullptr, has_v = v != nullptr;
    const bool packed = has_a && has_v;   // (a, v) travel as 12-byte records between passes
    // Short keys (a transpose with values of a matrix with fewer than 2^(32 - w0) columns and 17 .. 24 key bits): after the
    // first pass has spent its digit the keys travel as 16 bits, that digit in the top bits of the column word.
    const int w0 = bits / passes + (0 < bits % passes ? 1 : 0);
    const bool k16 = xp != nullptr && packed && passes == 3 && optr != nullptr && out_key == nullptr && bits - w0 <= 16 &&
                     (uint64_t)ex->expand_n <= (1ull << (32 - w0)) && ctx().opt.sort_short_keys;

    DevScope scope;
    int32_t *hist = nullptr, *part = nullptr;
    uint32_t *desc = nullptr;
    uint32_t *tk[2] = {nullptr, nullptr}, *ta[2] = {nullptr, nullptr};
    double *tv[2] = {nullptr, nullptr};
    Pay *tp[2] = {nullptr, nullptr};
    int st = scope.alloc(&hist, (size_t)RS_BINS * nblocks);
    if (st == CSX_OK) st = scope.alloc(&part, (size_t)RS_BINS * (nchunks + 1));
    int32_t *const dbase = part + (size_t)RS_BINS * nchunks;
    if (st == CSX_OK && xp) st = scope.alloc(&desc, (size_t)nblocks * RS_DESC_WORDS);
    const int ntmp = passes > 2 ? 2 : passes - 1;
    uint16_t *tk16[2] = {nullptr, nullptr};
    for (int t = 0; t < ntmp && st == CSX_OK; t++) {
        if (k16) st = scope.alloc(&tk16[t], (size_t)count + 8);
        else st = scope.alloc(&tk[t], (size_t)count);
        if (packed) {
            if (st == CSX_OK) st = scope.alloc(&tp[t], (size_t)count);
        } else {
            if (st == CSX_OK && has_a) st = scope.alloc(&ta[t], (size_t)count);
            if (st == CSX_OK && has_v) st = scope.alloc(&tv[t], (size_t)count);
        }
    }
    if (st == CSX_OK && xp) {
        hipLaunchKernelGGL(k_rs_tiledesc_head, dim3((nblocks + 255) / 256), dim3(256), 0, s, xp, ex->expand_n, count, nblocks,
                           desc);
        hipLaunchKernelGGL(k_rs_tiledesc_starts, dim3((unsigned)(((int64_t)ex->expand_n + 255) / 256)), dim3(256), 0, s, xp,
                           ex->expand_n, count, desc);
        if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
    }
    RsArgs g{};
    g.key = key;
    g.a = a;
    g.v = v;
    g.xp = xp;
    g.desc = desc;
    g.count = count;
    g.nblocks = nblocks;
    g.flat = ablation_env("CSX_SORT_FLAT") ? 1 : 0;
    bool in_aos = false;
    int shift = 0;
    for (int ps = 0; ps < passes && st == CSX_OK; ps++) {
        const bool last = ps == passes - 1;
        const int width = bits / passes + (ps < bits % passes ? 1 : 0);   // digits as even as the key allows
        const bool out_aos = packed && !last;
        g.shift = shift;
        g.mask = (1u << width) - 1u;
        g.okey = last ? out_key : tk[ps & 1];
        g.oa = last ? out_a : ta[ps & 1];
        g.ov = last ? out_v : tv[ps & 1];
        g.opay = last ? nullptr : tp[ps & 1];
        g.optr = last ? optr : nullptr;
        g.goff = hist;
        if (k16) {
            g.k16_w0 = w0;
            g.okey16 = last ? nullptr : tk16[ps & 1];
            g.okey = nullptr;
            if (ps > 0) g.shift = shift - w0;            // in the 16-bit key the first digit is gone
        }
        if (k16 && ps > 0) hipLaunchKernelGGL(k_rs_hist16, dim3(nblocks), dim3(RS_THREADS), 0, s, g.key16, count, g.shift, g.mask, hist);
        else hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(RS_THREADS), 0, s, g.key, count, g.shift, g.mask, hist);
        hipLaunchKernelGGL(k_rs_colsum, dim3(nchunks), dim3(RS_BINS), 0, s, hist, nblocks, chunk, part);
        hipLaunchKernelGGL(k_rs_chunkscan, dim3(1), dim3(RS_BINS * RS_SEGS), 0, s, part, nchunks, dbase);
        hipLaunchKernelGGL(k_rs_tileprefix, dim3(nchunks), dim3(RS_BINS), 0, s, hist, part, dbase, nblocks, chunk);
        if (k16 && ps == 0) hipLaunchKernelGGL((k_rs_scatter<true, true, false, true, 1>), dim3(nblocks), dim3(RS_THREADS), 0, s, g);
        else if (k16 && !last) hipLaunchKernelGGL((k_rs_scatter<true, true, true, true, 2>), dim3(nblocks), dim3(RS_THREADS), 0, s, g);
        else if (k16) hipLaunchKernelGGL((k_rs_scatter<true, true, true, false, 3>), dim3(nblocks), dim3(RS_THREADS), 0, s, g);
        else if (has_a && has_v) launch_scatter<true, true>(in_aos, out_aos, dim3(nblocks), s, g);
        else if (has_a) launch_scatter<true, false>(false, false, dim3(nblocks), s, g);
        else if (has_v) launch_scatter<false, true>(false, false, dim3(nblocks), s, g);
        else launch_scatter<false, false>(false, false, dim3(nblocks), s, g);
        if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
        g.key = g.okey;
        g.key16 = g.okey16;
        g.a = g.oa;
        g.v = g.ov;
        g.pay = g.opay;
        g.xp = nullptr;
        in_aos = out_aos;
        shift += width;
    }
    if (st == CSX_OK && optr) st = suffix_min_i32(optr, (int64_t)ex->nkeys + 1);
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;   // the temporaries go back now
    if (st == CSX_ERUNTIME) set_error("stable_sort_by_key: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    return st;
}

int stable_sort_by_key(const uint32_t *key, const uint32_t *a, const double *v, int64_t count, uint32_t key_limit,
                       uint32_t *out_key, uint32_t *out_a, double *out_v) {
    return stable_sort_by_key_ex(key, a, v, count, key_limit, out_key, out_a, out_v, nullptr);
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
