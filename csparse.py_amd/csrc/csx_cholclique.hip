// Forests of cliques: cs_schol + cs_chol (csparse.py:2051-2072, 561-619) for matrices whose elimination forest is a set of
// CLIQUES ON CONSECUTIVE COLUMNS -- block-diagonal SPD matrices with dense blocks, batches of small dense systems (G-spd,
// BASELINE config 5).  For such a matrix nothing of the general pattern machine (csx_cholsym.hip: postorder, start pairs,
// two radix sorts, row-subtree walks, two more sorts) is needed:
//
//   u[k] = the smallest row of column k's upper part (rows <= k; k itself when there is none).
//   The forest is a set of cliques on consecutive columns  <=>  u[0] = 0 and, for k > 0, u[k] = k (k starts a block)
//   or u[k] = u[k-1] (k continues the block of k-1).
//   (=>: column a of L receives no fill, so it is full iff every later column of the block holds A(a, k).  <=: every column
//   of the block reaches the block's first column, no column reaches further back, so etree = a chain per block and every
//   column of L is full below the diagonal inside its block.)
//
// Then parent[k] = k + 1 inside a block, -1 at its end; column k of L holds rows k .. end of the block (cs_chol appends
// rows in ascending order, the diagonal first: L.i follows from cp alone); and the values are one read of A's upper part
// and one write of L: k_chol_clique keeps a whole block (<= 64 columns) in the registers of one wave.
//
// Second form (round 4, "chol.forest"): where that rule fails the same u[] gives BLOCKS of consecutive columns closed under
// their upper entries (k starts one iff min_{j >= k} u[j] = k); for blocks of <= 64 columns tree, counts and the pattern of L
// come from a symbolic elimination on 64-bit row masks in the registers of one wave (k_forest_symbolic, further down) and
// the same block kernel factors them, storing only the pattern's rows -- forests of small SPARSE trees.  Both rules are
// restated on integers and checked against the plain-C port without a GPU in tests/test_forest_masks_model.py.
#include <algorithm>
#include <cstdio>

#include "csx_internal.h"
#include "csx_sweep.h"
#include "csx_cholclique.h"

namespace csx {

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// Raise a flag that millions of threads may want to raise: look first.  A store per wave to ONE address is served one after
// the other by the memory side (~10 ns each: 12 ms for the 1.2 M waves of a 4.8 M-column matrix that is no clique forest -- more
// than everything else in the recognition together); the look is a cached read.
__device__ __forceinline__ void cq_raise(int *flag) {
    if (*(volatile int *)flag == 0) *flag = 1;
}

// ---- recognition ---------------------------------------------------------------------------------------------------
// one wave per column: u[k]; flags[0] |= the upper part of some column is not strictly ascending (duplicates, unsorted,
// a negative row) -- the block kernel scatters a column's entries in parallel and needs them distinct
// 16 lanes per column, four entries per lane and step through one aligned 16-byte load (a wave instruction moves 1 KB:
// the load pipeline tracks instructions, not bytes -- with a wave per column and 4 bytes per lane this pass ran at 1.1 TB/s).
// (Arrays of a wrapped matrix end where they end: the last quad of the array is read element by element.)
__device__ __forceinline__ int4 cq_load4(const int32_t *__restrict__ idx, int32_t p, int32_t nnz) {
    if (p + 4 <= nnz) return *(const int4 *)(idx + p);
    int4 v = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);   // the array's last, partial quad: element by element
    if (p < nnz) v.x = idx[p];
    if (p + 1 < nnz) v.y = idx[p + 1];
    if (p + 2 < nnz) v.z = idx[p + 2];
    return v;
}

// (round 5: CQ_MIN_COLS consecutive columns to a group of 16 lanes, the first 64 entries of all of them requested before any is looked
// at: with one column per group a wave had ONE 1 KB load in flight in front of ~60 instructions of shuffles -- 2.9 TB/s, 0.44 ms of the
// 2.5 ms factor call at 5M columns)
constexpr int CQ_MIN_COLS = 4;
__global__ __launch_bounds__(256) void k_clique_min(int32_t n, int32_t nnz, const int32_t *__restrict__ Ap,
                                                    const int32_t *__restrict__ Ai, int32_t *__restrict__ u, int *flags) {
    const int t = threadIdx.x & 15;
    const int64_t g64 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    int32_t kk[CQ_MIN_COLS], bb[CQ_MIN_COLS], ee[CQ_MIN_COLS];
    int4 first4[CQ_MIN_COLS];
#pragma unroll
    for (int q = 0; q < CQ_MIN_COLS; q++) {
        const int64_t k64 = g64 * CQ_MIN_COLS + q;
        kk[q] = (int32_t)(k64 < n ? k64 : n - 1);                // the spare groups / columns of the last wave repeat column n - 1
        bb[q] = Ap[kk[q]];
        ee[q] = Ap[kk[q] + 1];
    }
#pragma unroll
    for (int q = 0; q < CQ_MIN_COLS; q++) {
        const int32_t p = (bb[q] & ~3) + 4 * t;
        first4[q] = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
        if (p < ee[q]) first4[q] = cq_load4(Ai, p, nnz);
    }
#pragma unroll
    for (int q = 0; q < CQ_MIN_COLS; q++) {
    const int64_t k64 = g64 * CQ_MIN_COLS + q;
    const int32_t k = kk[q];
    const int32_t b = bb[q], e = ee[q];
    // The column every forest this path was made for consists of: at most 64 entries (with the alignment slack: one step), the
    // first one row r0 <= k, entry t = row r0 + t up to row k, everything after it below the diagonal.  Sixteen lanes agree on that
    // with four comparisons each, one broadcast and four shuffles; the general rule below (prefix maxima over the lanes, counts,
    // positions: ~60 instructions a step) is for the columns that are anything else.  Same u[k], no flag raised: what the rule
    // below finds for such a column.
    if (e - (b & ~3) <= 64 && e > b) {
        const int4 v = first4[q];
        const int32_t r[4] = {v.x, v.y, v.z, v.w};
        const int sl = b & 3;                                   // lane 0's element holding the column's first entry
        const int32_t rs = sl == 0 ? r[0] : sl == 1 ? r[1] : sl == 2 ? r[2] : r[3];
        const int32_t r0 = __shfl(rs, 0, 16);
        bool fits = r0 >= 0 && r0 <= k;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int32_t pos = 4 * t + c - sl;                 // position in the column
            if (pos >= 0 && pos < e - b) fits = fits && (pos <= k - r0 ? r[c] == r0 + pos : r[c] > k);
        }
        int ok = fits ? 1 : 0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ok &= __shfl_xor(ok, o, 16);
        if (ok && e - b >= k - r0 + 1) {                         // (uniform over the 16 lanes; the upper part is all there)
            if (t == 0 && k64 < n) u[k] = r0;
            continue;
        }
    }
    int32_t mn = k, run = -1;          // smallest upper row; largest upper row of the steps before
    int32_t nup = 0, lastpos = -1;     // upper entries of this lane; position (in the column) of its last one
    bool bad = false;
    for (int32_t p0 = b & ~3; p0 < e; p0 += 64) {
        const int32_t p = p0 + 4 * t;
        int4 v = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
        if (p0 == (b & ~3)) v = first4[q];
        else if (p < e) v = cq_load4(Ai, p, nnz);
        const int32_t r[4] = {v.x, v.y, v.z, v.w};
        int32_t lmax = -1, first = 0x7fffffff;   // of this lane's upper entries
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const bool up = p + c >= b && p + c < e && r[c] <= k;
            if (up) {
                if (r[c] <= lmax) bad = true;
                if (first == 0x7fffffff) first = r[c];
                lmax = max(lmax, r[c]);
                mn = min(mn, r[c]);
                nup++;
                lastpos = p + c - b;
            }
        }
        // largest upper row of the lanes before this one (exclusive prefix maximum over the 16 lanes), and of earlier steps
        int32_t pre = lmax;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int32_t other = __shfl_up(pre, o, 16);
            if (t >= o) pre = max(pre, other);
        }
        int32_t excl = __shfl_up(pre, 1, 16);
        if (t == 0) excl = -1;
        excl = max(excl, run);
        if (first != 0x7fffffff && first <= excl) bad = true;
        run = max(run, __shfl(pre, 15, 16));
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 16));
        nup += __shfl_xor(nup, o, 16);
        lastpos = max(lastpos, __shfl_xor(lastpos, o, 16));
    }
    if (bad) cq_raise(&flags[0]);
    // flags[3] |= not "dense and in front": the upper part of a column is then rows u[k] .. k, one each, stored first -- entry t of
    // the column IS row u[k] + t, and the block kernel need not read the row indices at all
    if (nup != k - mn + 1 || lastpos != nup - 1) cq_raise(&flags[3]);
    if (t == 0 && k64 < n) u[k] = mn;
    }
}

// thread per column: the rule above (flags[1] |= broken), block starts, parent, the last column of every block
__global__ __launch_bounds__(256) void k_clique_mark(int32_t n, const int32_t *__restrict__ u, int32_t *__restrict__ is_start,
                                                     int32_t *__restrict__ parent, int32_t *__restrict__ end_of, int *flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t uk = u[k];
    const bool ok = k == 0 ? uk == 0 : (uk == (int32_t)k || uk == u[k - 1]);
    if (!ok || uk < 0) {
        cq_raise(&flags[1]);
        is_start[k] = 0;
        parent[k] = -1;
        return;
    }
    is_start[k] = uk == (int32_t)k ? 1 : 0;
    const bool last = k == n - 1 || u[k + 1] == (int32_t)k + 1;
    parent[k] = last ? -1 : (int32_t)k + 1;
    if (last) end_of[uk] = (int32_t)k;
}

// thread per column: column counts of L; block list; size statistics (stats[0] = widest block, lnz in *lnz) reduced per
// workgroup first (an atomic per block of the matrix on ONE address took 1.8 ms at 78 125 blocks)
__global__ __launch_bounds__(256) void k_clique_counts(int32_t n, const int32_t *__restrict__ u, const int32_t *__restrict__ is_start,
                                                       const int32_t *__restrict__ block_id, const int32_t *__restrict__ end_of,
                                                       int32_t *__restrict__ count, int32_t *__restrict__ start, int *stats,
                                                       unsigned long long *lnz, int *min_bs, int *nblocks) {
    __shared__ int s_max[4], s_min[4];
    __shared__ unsigned long long s_sum[4];
    int mx = 0, mn = 0x7fffffff;
    unsigned long long sum = 0ull;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t a = u[k], e = a >= 0 ? end_of[a] : (int32_t)k;   // (a matrix that failed the rule: anything, nothing is used)
        count[k] = e - (int32_t)k + 1;
        if (is_start[k]) {
            const int32_t bs = e - a + 1;
            start[block_id[k]] = (int32_t)k;
            mx = max(mx, bs);
            mn = min(mn, bs);
            sum += (unsigned long long)bs * (unsigned long long)(bs + 1) / 2ull;
        }
        if (k == n - 1) {
            start[block_id[k] + is_start[k]] = n;
            *nblocks = block_id[k] + is_start[k];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, __shfl_xor(mx, o));
        mn = min(mn, __shfl_xor(mn, o));
        sum += __shfl_xor(sum, o);
    }
    if ((threadIdx.x & 63) == 0) {
        s_max[threadIdx.x >> 6] = mx;
        s_min[threadIdx.x >> 6] = mn;
        s_sum[threadIdx.x >> 6] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&stats[0], max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
        atomicMin(min_bs, min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3])));
        atomicAdd(lnz, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    }
}

// ---- forests of small SPARSE trees (round 4, second step) -------------------------------------------------------------
// When the rule for cliques fails the matrix may still fall into BLOCKS of consecutive columns closed under their upper
// entries: k starts a block iff no column j >= k reaches above k, i.e. min_{j >= k} u[j] = k (a reverse running minimum of
// u).  Every elimination tree then lies inside one block.  For blocks of at most 64 columns the whole symbolic analysis of
// a block runs in the registers of one wave on 64-bit ROW MASKS: lane r holds the pattern of row r of L as bits; column j
// of L is the ballot of bit j over the lanes; eliminating column j adds to every row that has it the column's rows
// between j and the row (cs_ereach's row subtrees without walking a tree); the count of column j is the popcount of its
// ballot, its parent the first row below the diagonal (csparse.py:1136-1169, :703-764 for what the general path computes
// with a tree, a postorder and skeleton counts).
__global__ __launch_bounds__(256) void k_forest_mark(int32_t n, const int32_t *__restrict__ u, const int32_t *__restrict__ smin,
                                                     int32_t *__restrict__ is_start, int *flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (u[k] < 0) cq_raise(&flags[1]);
    is_start[k] = smin[k] == (int32_t)k ? 1 : 0;
}

// block list from the starts; stats[0] = widest block
__global__ __launch_bounds__(256) void k_forest_starts(int32_t n, const int32_t *__restrict__ is_start,
                                                       const int32_t *__restrict__ block_id, int32_t *__restrict__ start) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (is_start[k]) start[block_id[k]] = (int32_t)k;
    if (k == n - 1) start[block_id[k] + is_start[k]] = n;
}

__global__ __launch_bounds__(256) void k_forest_widest(int32_t nblocks, const int32_t *__restrict__ start, int *stats) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int w = b < nblocks ? start[b + 1] - start[b] : 0;
    for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o));
    if ((threadIdx.x & 63) == 0 && w > *(volatile int *)&stats[0]) atomicMax(&stats[0], w);
}

// the row masks of one block after symbolic elimination (lane = row; wave-uniform c0, bs <= 64); *mycol = the column of L
// this lane's index names (rows as bits, the diagonal included)
__device__ __forceinline__ unsigned long long cq_block_masks(int32_t c0, int32_t bs, int lane, const int32_t *__restrict__ Ap,
                                                             const int32_t *__restrict__ Ai, unsigned long long *mycol) {
    unsigned long long mask = 0ull;
    if (lane < bs) {
        const int32_t col = c0 + lane;
        mask = 1ull << lane;
        for (int32_t p = Ap[col]; p < Ap[col + 1]; p++) {       // the upper part of column r of A is row r of L's pattern
            const int32_t i = Ai[p];
            if (i <= col && i >= c0) mask |= 1ull << (i - c0);
        }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long col_of_lane = 0ull;
    for (int j = 0; j < bs; j++) {
        const unsigned long long cj = __ballot((mask >> j) & 1ull);       // rows of column j (final: columns < j are done)
        if (lane == j) col_of_lane = cj;
        if (((mask >> j) & 1ull) && lane > j) mask |= cj & below & ~((2ull << j) - 1ull);
    }
    *mycol = col_of_lane;
    return mask;
}

__global__ __launch_bounds__(256) void k_forest_symbolic(const int32_t *__restrict__ start, int32_t nblocks,
                                                         const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                         int32_t *__restrict__ parent, int32_t *__restrict__ count,
                                                         unsigned long long *__restrict__ colmask, unsigned long long *lnz) {
    const int lane = threadIdx.x & 63;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long mine = 0ull;
    if (t < nblocks) {
        const int32_t c0 = __builtin_amdgcn_readfirstlane(start[t]);
        const int32_t bs = __builtin_amdgcn_readfirstlane(start[t + 1]) - c0;
        unsigned long long cj = 0ull;
        (void)cq_block_masks(c0, bs, lane, Ap, Ai, &cj);
        if (lane < bs) {
            const unsigned long long under = cj & ~((2ull << lane) - 1ull);     // rows below the diagonal
            count[c0 + lane] = 1 + __popcll(under);
            colmask[c0 + lane] = cj;
            parent[c0 + lane] = under ? c0 + (__ffsll((long long)under) - 1) : -1;
            mine = 1ull + (unsigned long long)__popcll(under);
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    __shared__ unsigned long long s_sum[4];
    if (lane == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(lnz, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

void free_clique(CliqueForest *F) {
    dfree(F->parent);
    dfree(F->cp);
    dfree(F->start);
    dfree(F->colmask);
    dfree(F->order);
    F->parent = F->cp = F->start = F->order = nullptr;
    F->colmask = nullptr;
}

void free_clique_cache(CliqueForest *F) {
    if (!F) return;
    free_clique(F);
    delete F;
}

__global__ void k_cq_init(int *flags) {
    if (threadIdx.x < 8) flags[threadIdx.x] = threadIdx.x == 6 ? 0x7fffffff : 0;
}

__global__ __launch_bounds__(256) void k_cq_size_key(const int32_t *__restrict__ start, int32_t nblocks, int32_t max_bs,
                                                     uint32_t *__restrict__ key, uint32_t *__restrict__ id) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks) return;
    key[t] = (uint32_t)min(max(max_bs - (start[t + 1] - start[t]), 0), max_bs);     // biggest first
    id[t] = (uint32_t)t;
}
// F->order for a forest of blocks of unequal sizes: in matrix order a workgroup of the block kernel (four waves, a block each) keeps
// its LDS until its biggest block is done -- a block of 64 columns is 500 times the arithmetic of one of 8 -- and the waves of its
// small blocks sit idle; blocks of like size side by side leave together
static int clique_order(CliqueForest *F) {
    if (F->order || F->nblocks <= 0 || (F->min_bs == F->max_bs && !F->sparse)) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    uint32_t *key = nullptr, *id = nullptr;
    CSX_TRY(tmp.alloc(&key, (size_t)F->nblocks));
    CSX_TRY(tmp.alloc(&id, (size_t)F->nblocks));
    CSX_TRY(dalloc(&F->order, (size_t)F->nblocks));
    hipLaunchKernelGGL(k_cq_size_key, dim3((unsigned)((F->nblocks + 255) / 256)), dim3(256), 0, s, F->start, F->nblocks, F->max_bs, key, id);
    CSX_LAUNCH_CHECK();
    CSX_TRY(stable_sort_by_key(key, id, nullptr, F->nblocks, (uint32_t)F->max_bs + 1, nullptr, (uint32_t *)F->order, nullptr));
    CSX_HIP(hipStreamSynchronize(s));       // (key / id are temporaries)
    return CSX_OK;
}

int clique_forest(const Csc *A, CliqueForest *F, bool *ok) {
    *ok = false;
    const int32_t n = A->n;
    if (n <= 0 || A->m != A->n) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *u = nullptr, *is_start = nullptr, *block_id = nullptr, *end_of = nullptr, *count = nullptr, *start_all = nullptr;
    int *flags = nullptr;   // [0] not ascending, [1] not a clique forest, [2] widest block, [3] not "dense and in front"; [4..5] lnz (64 bits),
                            // [6] narrowest block, [7] number of blocks
    CSX_TRY(tmp.alloc(&u, (size_t)n));
    CSX_TRY(tmp.alloc(&is_start, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&block_id, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&end_of, (size_t)n));
    CSX_TRY(tmp.alloc(&count, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&start_all, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&flags, 8));
    CSX_TRY(dalloc(&F->parent, (size_t)n));
    CSX_TRY(dalloc(&F->cp, (size_t)n + 1));
    // Everything a forest of cliques needs is queued BEFORE the host looks at a flag -- one wait for the whole analysis (round 4
    // waited after the rule, after the block count and after lnz).  A matrix that fails the rule has paid for two scans and the counts
    // in vain (0.2 ms at 5M columns) and goes on to the second rule below.
    hipLaunchKernelGGL(k_cq_init, dim3(1), dim3(64), 0, s, flags);
    hipLaunchKernelGGL(k_clique_min, dim3(blocks_for(((int64_t)n + CQ_MIN_COLS - 1) / CQ_MIN_COLS * 16)), dim3(256), 0, s, n, A->nnz, A->p, A->i, u, flags);
    hipLaunchKernelGGL(k_clique_mark, dim3(blocks_for(n)), dim3(256), 0, s, n, u, is_start, F->parent, end_of, flags);
    CSX_TRY(scan_exclusive_i32(is_start, block_id, n, nullptr));
    hipLaunchKernelGGL(k_clique_counts, dim3(std::min(blocks_for(n), 1024u)), dim3(256), 0, s, n, u, is_start, block_id, end_of, count,
                       start_all, flags + 2, (unsigned long long *)(flags + 4), flags + 6, flags + 7);
    CSX_TRY(scan_exclusive_i32(count, F->cp, n, nullptr));
    int h[8] = {0};
    CSX_HIP(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (h[1]) {
        // not a forest of cliques: blocks of consecutive columns closed under their upper entries?  (needs sorted upper parts,
        // like the block kernel; blocks of at most 64 columns)
        if (!ctx().opt.chol_forest || h[0]) {
            free_clique(F);
            return CSX_OK;
        }
        int32_t *smin = nullptr;
        CSX_TRY(tmp.alloc(&smin, (size_t)n));
        CSX_HIP(hipMemcpyAsync(smin, u, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        CSX_TRY(suffix_min_i32(smin, n));
        hipLaunchKernelGGL(k_cq_init, dim3(1), dim3(64), 0, s, flags);
        hipLaunchKernelGGL(k_forest_mark, dim3(blocks_for(n)), dim3(256), 0, s, n, u, smin, is_start, flags);
        int64_t nb = 0;
        CSX_TRY(scan_exclusive_i32(is_start, block_id, n, &nb));
        CSX_TRY(dalloc(&F->start, (size_t)nb + 1));
        hipLaunchKernelGGL(k_forest_starts, dim3(blocks_for(n)), dim3(256), 0, s, n, is_start, block_id, F->start);
        hipLaunchKernelGGL(k_forest_widest, dim3(blocks_for(nb)), dim3(256), 0, s, (int32_t)nb, F->start, flags + 2);
        CSX_HIP(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        if (h[1] || h[2] > CLIQUE_MAX_BLOCK) {      // a negative index, or a block the wave kernel cannot hold
            free_clique(F);
            return CSX_OK;
        }
        CSX_TRY(dalloc(&F->colmask, (size_t)n));
        hipLaunchKernelGGL(k_forest_symbolic, dim3(blocks_for(nb * 64)), dim3(256), 0, s, F->start, (int32_t)nb, A->p, A->i, F->parent,
                           count, F->colmask, (unsigned long long *)(flags + 4));
        CSX_TRY(scan_exclusive_i32(count, F->cp, n, nullptr));
        CSX_HIP(hipMemcpyAsync(h, flags, sizeof h, hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        unsigned long long lz = 0;
        std::memcpy(&lz, h + 4, sizeof lz);
        if (lz > 0x7fffffffull) {
            free_clique(F);
            return CSX_OK;
        }
        F->n = n;
        F->nblocks = (int32_t)nb;
        F->max_bs = h[2];
        F->min_bs = 0;
        F->lnz = (int64_t)lz;
        F->ascending = true;
        F->dense_in_front = false;
        F->sparse = true;
        CSX_TRY(clique_order(F));
        *ok = true;
        return CSX_OK;
    }
    unsigned long long lnz = 0;
    std::memcpy(&lnz, h + 4, sizeof lnz);
    if (lnz > 0x7fffffffull) {   // L does not fit int32 indices: the general path reports it
        free_clique(F);
        return CSX_OK;
    }
    const int32_t nblocks = h[7];
    CSX_TRY(dalloc(&F->start, (size_t)nblocks + 1));
    CSX_HIP(hipMemcpyAsync(F->start, start_all, ((size_t)nblocks + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    F->n = n;
    F->nblocks = nblocks;
    F->max_bs = h[2];
    F->min_bs = h[6];
    F->lnz = (int64_t)lnz;
    F->ascending = h[0] == 0;
    F->dense_in_front = h[0] == 0 && h[3] == 0;
    CSX_TRY(clique_order(F));
    *ok = true;
    return CSX_OK;
}

__global__ __launch_bounds__(256) void k_clique_compare(int32_t n, const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                        const int32_t *__restrict__ ca, const int32_t *__restrict__ cb, int *bad) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n) return;
    if (ca[q] != cb[q] || (q < n && pa[q] != pb[q])) cq_raise(bad);
}

// cs_chol(A, S): the caller's S.parent / S.cp (host arrays) must be this forest's; uploaded and compared on the device
// The upload of S (40 MB at 5M columns, from pageable memory: 0.8 ms of host time) and the comparison run on a stream of their
// own (the context's side stream), so that csx_chol can start the block kernel first and pay only for the longer of the two.
int clique_matches_begin(const CliqueForest &F, const int32_t *parent, const int32_t *cp, CliqueCompare *c) {
    const int32_t n = F.n;
    CSX_TRY(dalloc(&c->dp, (size_t)n));
    CSX_TRY(dalloc(&c->dc, (size_t)n + 1));
    CSX_TRY(dalloc(&c->bad, 1));
    c->parent = parent;
    c->cp = cp;
    c->F = &F;
    // The pool's blocks are ordered on the context's stream only: whoever held these three before may still be using them in work
    // queued there (a solve or a product the caller did not wait for).  The side stream therefore starts behind everything the
    // context's stream holds NOW -- before the caller's block kernel goes in, so the upload still runs beside that kernel.
    CSX_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    CSX_HIP(hipEventRecord(c->ev, ctx().stream));
    return CSX_OK;
}

// after the caller has put its own work on the context's stream
int clique_matches_run(CliqueCompare *c) {
    hipStream_t s = ctx().side;
    const int32_t n = c->F->n;
    CSX_HIP(hipStreamWaitEvent(s, c->ev, 0));
    CSX_HIP(hipMemsetAsync(c->bad, 0, sizeof(int), s));
    CSX_HIP(hipMemcpyAsync(c->dp, c->parent, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(c->dc, c->cp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_clique_compare, dim3(blocks_for((int64_t)n + 1)), dim3(256), 0, s, n, c->dp, c->F->parent, c->dc, c->F->cp, c->bad);
    CSX_HIP(hipMemcpyAsync(&c->h, c->bad, sizeof(int), hipMemcpyDeviceToHost, s));
    return CSX_OK;
}

int clique_matches_end(CliqueCompare *c, bool *same) {
    *same = false;
    int st = CSX_OK;
    if (hipStreamSynchronize(ctx().side) != hipSuccess) st = CSX_ERUNTIME;
    if (c->ev) (void)hipEventDestroy(c->ev);
    c->ev = nullptr;
    dfree(c->dp);
    dfree(c->dc);
    dfree(c->bad);
    c->dp = c->dc = nullptr;
    c->bad = nullptr;
    if (st == CSX_OK) *same = c->h == 0;
    return st;
}

// ---- values --------------------------------------------------------------------------------------------------------
// One wave per block of bs <= 64 columns; lane r owns ROW r of the block, a[c] = element (r, c) of the lower triangle.
//
// Loading: cs_chol reads the UPPER part of C (csparse.py:593-595): entry C(i, k), i <= k, is element (k, i) of the lower
// triangle, i.e. lane k's register i -- a coalesced read of column k of A delivers one lane's registers spread over the
// wave, so 16 columns at a time pass through an LDS tile (column-major with an odd stride) and the 16 lanes that own them
// read their rows back.  Entries the pattern lacks are zeros (fill).
//
// Factoring: right-looking, in panels of 8 columns.  Every element receives its updates - L(r, j) L(c, j) for j = 0, 1, ...
// in ascending order, multiply then subtract, then one division by the pivot: the operation sequence of cs_chol's up-looking
// row solve on a chain (csparse.py:598-612; cs_ereach hands the columns over in ascending order), so L.x is bit-identical to
// it.  L(c, j) reaches the lanes through LDS (the panel's finished columns are written there once, an update reads its factor
// with a broadcast read; see "factor" below for why not v_readlane).  The register file has no dynamic index, and the whole triangle unrolled would be 90 KB of code against 64 KB of instruction cache: the loop body is
// written for a WINDOW whose first eight registers are the current panel -- factor those, update the rest of the window by
// them (groups of eight columns, skipped when they lie beyond the block), store the panel, slide the window by eight
// registers -- 18 KB of code, executed bs / 8 times, no update outside the triangle's rectangle of live groups.
//
// Storing: column g of the block goes to L.x / L.i at Lp[c0] + g bs - g (g - 1) / 2, rows ascending: runs of up to 512 / 256 bytes.
//
// SMALL: fewer than 2^29 entries in A -- one buffer resource per array and the column's start as the scalar offset of the load;
// DENSE: every upper part is rows c0 .. c0 + k stored in front (k_clique_min) -- the row indices are not read.
constexpr int CQ_WAVES = 4;
constexpr int CQ_CH = 16;     // columns staged at a time
constexpr int CQ_LD = 65;     // doubles per staged column
constexpr int CQ_G = 8;       // columns whose loads are in flight together

// Lanes of ONE wave talk through LDS here: a store by one lane, a load of that slot by the others.  The hardware keeps a
// wave's LDS operations in order, but the language is per thread: without a fence the compiler may let the other lanes'
// load run before the storing lane's branch (it did, in an experiment that sent the pivot this way: the readers got the
// previous column's value).  A wavefront-scope
// release / acquire pair around a wave barrier tells it; it costs no instruction.
__device__ __forceinline__ void cq_wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A column of A (or of L) starts at a wave-uniform place: it is addressed through a BUFFER RESOURCE -- scalar base, size in
// bytes -- and a lane's 32-bit byte offset, computed once per kernel (lane * 4, lane * 8).  No 64-bit address per lane and
// load (that arithmetic was a quarter of the load phase's instructions, and the kernel is bound by instruction issue), and the
// hardware's range check returns zero for the lanes past the column's end instead of clamps and exec masks around the load.
typedef unsigned int cq_u32x2 __attribute__((ext_vector_type(2)));
constexpr int CQ_RSRC_FLAGS = 0x00020000;   // raw buffer of 32-bit data, gfx9 family

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cq_rsrc(const void *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, CQ_RSRC_FLAGS);
}

__device__ __forceinline__ double cq_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// PARTS: 1 load, 2 factor, 4 store -- 7 is the kernel; the others exist in the ablation build only (what each phase costs)
// SPARSE: the blocks are small trees (CliqueForest::sparse).  The arithmetic is the dense block's -- an entry outside L's
// pattern is a zero that stays zero -- and only the store differs: column g keeps the rows of its mask, compacted.
// EMIT (csx_cholsol_factor): the kernel also writes what the matrix-core solve reads beside L.x -- W_ii = inv(L_ii) (the diagonal
// tiles are kept in LDS as their panels finish; after the last panel lane (tile, column) inverts its column, the arithmetic of
// k_mfma_frags), the guard's measure and the plan's block list -- and does NOT write L.i (rows j, j + 1, ... in every column:
// csc_fill_rows makes them from L.p when somebody asks).  L.x is written as always; for EQUAL blocks of 16 / 32 / 64 columns that is
// all (k_cholsol_mfma reads the off-diagonal tiles where they lie); blocks of unequal sizes (em.frag_off) also get their padded tiles
// -L_ij as fragments (as a panel of eight columns is finished: the lanes of tile row i hold them), csx_trimfma.h's layout.
constexpr int CQ_DIAG = 16 * 17;      // doubles per kept diagonal tile (odd stride)
constexpr int CQ_TILE = CQ_CH * CQ_LD + 2;                 // the staging tile + the spare slot rejected entries go to
constexpr int CQ_TILE_EMIT = 8 * 64 + 4 * CQ_DIAG;         // colbuf + four diagonal tiles (>= CQ_TILE)
static_assert(CQ_TILE_EMIT >= CQ_TILE, "the staging tile must fit");

// RELAXED ("chol.exact" = 0, opt-in): the same elimination with what bit-identity forbids -- every update one fused multiply-add
// instead of a multiply and a subtraction (4 032 -> 2 016 vector instructions per 64-column block), the pivot column scaled by a
// reciprocal square root refined to full precision (v_rsq_f64 + two Newton steps, ~12 instructions; L(j,j) = d * rsqrt(d) corrected
// once) instead of an IEEE square root and an IEEE division per column (~35 instructions each, 64 of each per block).  L.x then
// equals the exact kernel's to rounding (1e-13 normwise tested), which is what BASELINE.json's north_star asks of x[].
__device__ __forceinline__ double cq_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);                 // ~2^-26 relative
#pragma unroll
    for (int it = 0; it < 2; it++) {                    // y <- y + y (1 - d y^2) / 2
        const double e = __builtin_fma(-d * y, y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}

template <int PARTS, bool SMALL, bool DENSE, bool SPARSE, bool EMIT, bool RELAXED = false>
__global__ __launch_bounds__(64 * CQ_WAVES, 3) void k_chol_clique(const int32_t *__restrict__ start, int32_t nblocks, int32_t n, int32_t nnz,
                                                                 const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                                 const double *__restrict__ Ax, const int32_t *__restrict__ Lp,
                                                                 int32_t *__restrict__ Li, double *__restrict__ Lx, int *notspd,
                                                                 const unsigned long long *__restrict__ colmask, CliqueEmit em) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) double s_tile[CQ_WAVES][EMIT ? CQ_TILE_EMIT : CQ_TILE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t q = (int64_t)blockIdx.x * CQ_WAVES + w;
    if (q >= nblocks) return;   // no workgroup barrier below
    const int64_t t = em.order ? (int64_t)__builtin_amdgcn_readfirstlane(em.order[q]) : q;     // (unequal blocks: biggest first, CliqueForest::order)
    const int32_t c0 = __builtin_amdgcn_readfirstlane(start[t]);
    const int32_t bs = __builtin_amdgcn_readfirstlane(start[t + 1]) - c0;
    double *tile = s_tile[w];
    double a[64];
#pragma unroll
    for (int c = 0; c < 64; c++) a[c] = 0.0;
    if (!(PARTS & 1)) {
#pragma unroll
        for (int c = 0; c < 64; c++) a[c] = c == lane ? 64.0 : 0.0078125;
    }
    // ---- load ----
    // (what is NOT done here matters as much as what is: the kernel is bound by instruction issue -- 7 000 vector
    // instructions per block at 2.2 ns each -- so the loads are unconditional (clamped addresses, no exec juggling), the
    // scatter writes rejected entries to a spare slot instead of branching, and the read-back runs under one exec mask)
    const int lane4 = lane * 4, lane8 = lane * 8;
    const __amdgpu_buffer_rsrc_t ri_all = cq_rsrc(Ai, SMALL ? nnz * 4 : 0), rx_all = cq_rsrc(Ax, SMALL ? nnz * 8 : 0);
    if (PARTS & 1) {
        const int32_t apb = Ap[min(c0 + lane, n)], ape = Ap[min(c0 + lane + 1, n)];      // lane l: column c0 + l
        const bool lng = lane < bs && ape - apb > 64;
        const bool any_long = __ballot(lng) != 0ull;
#pragma unroll
        for (int q = 0; q < 64 / CQ_CH; q++) {
            if (CQ_CH * q < bs) {
                {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    d2 *t2 = (d2 *)tile;
#pragma unroll
                    for (int e = 0; e < (CQ_CH * CQ_LD / 2 + 63) / 64; e++)
                        if (e * 64 + lane < CQ_CH * CQ_LD / 2) t2[e * 64 + lane] = d2{0.0, 0.0};
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int kk0 = 0; kk0 < CQ_CH; kk0 += CQ_G) {
                    // CQ_G columns at a time, index and value of every column requested before anything is used
                    int32_t ii[CQ_G], slot[CQ_G];
                    double vv[CQ_G];
                    bool in[CQ_G];
#pragma unroll
                    for (int uu = 0; uu < CQ_G; uu++) {
                        const int k = CQ_CH * q + kk0 + uu;
                        const int32_t pb = __builtin_amdgcn_readlane(apb, k);
                        const int32_t pe = k < bs ? __builtin_amdgcn_readlane(ape, k) : pb;
                        const int32_t len = min(pe - pb, 64);                              // uniform; longer columns: below
                        in[uu] = lane < len;
                        // the value is wanted for upper entries only; in an ascending column those are the first k + 1: the
                        // lanes past them get zero without a request (an upper entry further back -- lower entries stored in
                        // front of it -- is fetched below): half of A.x is never read
                        if (SMALL) {   // ONE resource per array, the column's start as the instruction's scalar offset
                            if (!DENSE) ii[uu] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(ri_all, lane4, pb * 4, 0);
                            vv[uu] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx_all, min(lane8, 8 * k), pb * 8, 0));   // (the lanes past entry k ask for it again: no new line)
                        } else {       // 2^29 entries or more: byte offsets past 32 bits -- a resource per column
                            if (!DENSE) ii[uu] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(cq_rsrc(Ai + pb, len * 4), lane4, 0, 0);
                            vv[uu] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cq_rsrc(Ax + pb, min(len, k + 1) * 8), lane8, 0, 0));
                        }
                        // DENSE (found by k_clique_min for the whole matrix): entry t of the column is row c0 + t, t <= k -- the
                        // row indices are not read: a third of the kernel's fetches
                        if (DENSE) ii[uu] = c0 + lane;
                    }
                    bool late = false;
#pragma unroll
                    for (int uu = 0; uu < CQ_G; uu++) {
                        const int k = CQ_CH * q + kk0 + uu;
                        const uint32_t rel = (uint32_t)(ii[uu] - c0);          // rows c0 .. c0 + k are the upper part
                        const bool okk = in[uu] && rel <= (uint32_t)k;
                        slot[uu] = okk ? (kk0 + uu) * CQ_LD + (int)rel : CQ_CH * CQ_LD;
                        late |= okk && lane > k;
                    }
                    if (!DENSE && __ballot(late) != 0ull) {
#pragma unroll
                        for (int uu = 0; uu < CQ_G; uu++) {
                            const int k = CQ_CH * q + kk0 + uu;
                            if (slot[uu] != CQ_CH * CQ_LD && lane > k) vv[uu] = Ax[__builtin_amdgcn_readlane(apb, k) + lane];
                        }
                    }
#pragma unroll
                    for (int uu = 0; uu < CQ_G; uu++) tile[slot[uu]] = vv[uu];
                }
                if (any_long) {   // columns of more than 64 entries (a lower part with duplicates, say)
                    for (int kk = 0; kk < CQ_CH && CQ_CH * q + kk < bs; kk++) {
                        const int k = CQ_CH * q + kk;
                        const int32_t pb = __builtin_amdgcn_readlane(apb, k), pe = __builtin_amdgcn_readlane(ape, k);
                        for (int32_t p0 = pb + 64; p0 < pe; p0 += 64) {
                            const int32_t p = p0 + lane;
                            const int32_t i = p < pe ? Ai[p] : 0x7fffffff;
                            if (i <= c0 + k && i >= c0) tile[kk * CQ_LD + i - c0] = Ax[p];
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if ((lane >> 4) == q) {   // the 16 lanes whose rows these columns are; what lies right of the diagonal stays 0
                    const double *row = tile + (lane & 15) * CQ_LD;
#pragma unroll
                    for (int c = 0; c < CQ_CH * (q + 1); c++) a[c] = row[c];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // ---- factor ----
    // L(c, j) reaches the lanes through LDS: the panel's finished columns are written there (one ds_write per lane and
    // column) and an update reads its factor with a BROADCAST read -- measured on this chip (tools/ubench/valu_rate.hip, every
    // SIMD busy): v_readlane_b32 3.6 ns per wave instruction and SIMD, two of them per fp64 value; a broadcast ds_read_b64
    // 4.0 ns, in the LDS pipe beside the vector ALU; v_mul_f64 / v_add_f64 2.2 ns each.  With v_readlane the factor phase
    // took 2.2 ms at 5M rows, 62 % of it the readlanes.
    const int64_t base = Lp[c0];
    double *em_diag = tile + 8 * 64;                                   // EMIT: the diagonal tiles, beside colbuf
    double *em_frag = nullptr;
    double em_lmax = 0.0;
    const int em_rows = EMIT ? 16 * ((bs + 15) >> 4) : 0;               // EMIT: the block's rows padded to whole tiles
    if (EMIT) {
        // equal blocks (no frag_off): the W tiles only, NB of them per block -- k_cholsol_mfma reads the off-diagonal tiles in L.x;
        // unequal blocks: csx_trimfma.h's layout, padded off-diagonal tiles included
        em_frag = em.frag + (em.frag_off ? (size_t)em.frag_off[t] : (size_t)t * (size_t)((bs >> 4) * 256));
        if (em.trees) {
            if (lane < bs) em.tree_nodes[c0 + lane] = c0 + lane;
            if (lane == 0) em.trees[t] = Tree{c0, bs};
        }
        if (bs & 15) {      // a block that does not fill its last tile: the diagonal tiles start as the identity (the padding's part stays)
#pragma unroll
            for (int e = 0; e < (4 * CQ_DIAG + 63) / 64; e++) {
                const int q = e * 64 + lane;
                if (q < 4 * CQ_DIAG) em_diag[q] = ((q % CQ_DIAG) / 17 == (q % CQ_DIAG) % 17) ? 1.0 : 0.0;
            }
            cq_wave_sync_lds();
        }
    }
    double *colbuf = tile;   // [8][64]   (s_setprio around the phases -- loading waves first, or factoring waves first -- 1.61-1.65 ms: no gain)
#pragma unroll 1
    for (int J = 0; J < bs; J += 8) {
#pragma unroll
        for (int jw = 0; (PARTS & 2) && jw < 8; jw++) {
            const int g = J + jw;
            if (g < bs) {
                const double d = cq_bcast(a[jw], g);    // (the pivot through LDS instead -- a store by its lane, a broadcast read --
                                                        // saves 7 ns of issue and costs an LDS round trip on the chain: no gain, 1.72 ms either way)
                if (d <= 0.0 && lane == 0) atomicMin(notspd, c0 + g);   // csparse.py:612
                double ljj, l;
                if (RELAXED) {
                    const double y = cq_rsqrt(d);
                    double sq = d * y;                                   // sqrt(d), corrected once: s + (d - s^2) y / 2
                    sq = __builtin_fma(__builtin_fma(-sq, sq, d), 0.5 * y, sq);
                    ljj = sq;
                    l = a[jw] * y;
                } else {
                    ljj = sqrt(d);
                    l = a[jw] / ljj;
                }
                a[jw] = lane == g ? ljj : l;
                colbuf[jw * 64 + lane] = a[jw];
                cq_wave_sync_lds();
                if (jw < 7) {
                    double sc[8];
#pragma unroll
                    for (int cw = jw + 1; cw < 8; cw++) sc[cw] = colbuf[jw * 64 + J + cw];
#pragma unroll
                    for (int cw = jw + 1; cw < 8; cw++) {
                        if (RELAXED) {
                            a[cw] = __builtin_fma(-a[jw], sc[cw], a[cw]);
                        } else {
                            const double pr = a[jw] * sc[cw];
                            a[cw] = a[cw] - pr;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int gq = 1; (PARTS & 2) && gq < 8; gq++) {
            if (J + 8 * gq < bs) {
                // a group of eight columns takes the panel's columns one after the other; the eight updates by one panel
                // column are independent of each other
#pragma unroll
                for (int jw = 0; jw < 8; jw++) {
                    double sc[8], pr[8];
                    const double *src = colbuf + jw * 64 + J + 8 * gq;
#pragma unroll
                    for (int cc = 0; cc < 8; cc++) sc[cc] = (PARTS & 8) ? a[(jw + cc) & 63] : src[cc];
                    if (RELAXED) {
#pragma unroll
                        for (int cc = 0; cc < 8; cc++) a[8 * gq + cc] = __builtin_fma(-a[jw], sc[cc], a[8 * gq + cc]);
                    } else {
#pragma unroll
                        for (int cc = 0; cc < 8; cc++) pr[cc] = a[jw] * sc[cc];
#pragma unroll
                        for (int cc = 0; cc < 8; cc++) a[8 * gq + cc] = a[8 * gq + cc] - pr[cc];
                    }
                }
            }
        }
#pragma unroll
        for (int jw = 0; jw < 8; jw++) {
            const int g = J + jw;
            if (SPARSE && g < bs) {
                // (wave-uniform addresses: scalar loads, no register held across the factor phase)
                const unsigned long long cg = colmask[c0 + g];
                const int32_t lb = Lp[c0 + g];
                if (((cg >> lane) & 1ull) && ((PARTS & 4) || a[jw] == 12345.678)) {
                    const int32_t pos = lb + __popcll(cg & ((1ull << lane) - 1ull));
                    Lx[pos] = a[jw];
                    Li[pos] = c0 + lane;
                }
            } else if (g < bs) {   // uniform; the column's place in L is a scalar base, a lane adds its row
                const int64_t colbase = base + (int64_t)g * bs - (int64_t)g * (g - 1) / 2 - g;
                if (lane >= g && lane < bs && ((PARTS & 4) || a[jw] == 12345.678)) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cq_u32x2, a[jw]), cq_rsrc(Lx + colbase, bs * 8), lane8, 0, 0);
                    if (!EMIT) __builtin_amdgcn_raw_buffer_store_b32((unsigned int)(c0 + lane), cq_rsrc(Li + colbase, bs * 4), lane4, 0, 0);
                }
            }
        }
        if (EMIT && J < bs) {
            // columns J .. J + 7 lie in tile column tj, k-steps sx0 and sx0 + 1; lane = (tile row ti, row m of the tile).  Fragment
            // (tile (ti, tj), k-step sx) holds element (m, 4 sx + kq) at position 16 kq + m: for one column 16 lanes write 128
            // contiguous bytes.  (Above the diagonal a[] is +0.0: nothing of it is read.)
            const int tj = J >> 4, ti = lane >> 4, m = lane & 15;
            if (ti > tj && lane < em_rows && em.frag_off) {   // (rows past the block, in its last tile row: a[] is zero there -- the padding)
                double *dst = em_frag + (size_t)(((ti * (ti + 1) / 2 + tj) * 4 + ((J & 8) >> 2)) * 64 + m);
#pragma unroll
                for (int jw = 0; jw < 8; jw++) dst[(jw >> 2) * 64 + (jw & 3) * 16] = J + jw < bs ? -a[jw] : 0.0;
            } else if (ti == tj && lane < bs) {
                double *dd = em_diag + ti * CQ_DIAG + m * 17 + (J & 8);
#pragma unroll
                for (int jw = 0; jw < 8; jw++)
                    if (J + jw < bs) dd[jw] = a[jw];
            }
#pragma unroll
            for (int jw = 0; jw < 8; jw++) em_lmax = fmax(em_lmax, fabs(a[jw]));
        }
#pragma unroll
        for (int c = 0; c < 56; c++) a[c] = a[c + 8];
    }
    if (EMIT) {
        cq_wave_sync_lds();
        const int blk = lane >> 4, col = lane & 15;
        double wmax = 0.0;
        if (16 * blk < bs) {
            double wcol[16];
            tile_inverse_column(em_diag + blk * CQ_DIAG, 17, col, wcol);
            // fragment (tile (blk, blk), k-step col >> 2), positions 16 (col & 3) + r: this lane's sixteen values are contiguous
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *dst = (d2 *)(em_frag + (size_t)(((em.frag_off ? blk * (blk + 1) / 2 + blk : blk) * 4 + (col >> 2)) * 64 + (col & 3) * 16));
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                dst[r >> 1] = d2{wcol[r], wcol[r + 1]};
                wmax = fmax(wmax, fmax(fabs(wcol[r]), fabs(wcol[r + 1])));
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            em_lmax = fmax(em_lmax, __shfl_xor(em_lmax, d, 64));
            wmax = fmax(wmax, __shfl_xor(wmax, d, 64));
        }
        if (lane == 0) {   // (look first: after the first blocks hardly any raises the maximum, and an atomic per block on one address is served one after the other)
            const unsigned long long bits = (unsigned long long)__double_as_longlong(em_lmax * wmax);
            if (!(bits <= *(volatile unsigned long long *)em.cond_bits)) atomicMax(em.cond_bits, bits);
        }
    }
}

// ---- values, rounding-equal, on the matrix cores ("chol.exact" = 0; equal dense blocks of 16 / 32 / 48 / 64 columns) --------------
// A blocked right-looking Cholesky of one block per wave on 16 x 16 tiles, written for the UPPER factor U = L' so that every tile
// lives in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (c = l & 15, rq = l >> 4), register r <-> element (rq + 4 r, c))
// from the load to the store and never changes it: for 16 x 16 matrices P, Q the product P Q is four matrix instructions whose
// A operand is the accumulator form of P' and whose B operand is the accumulator form of Q, register by register.  Step k:
//   diagonal tile  T_kk = U_kk' U_kk and V_kk = inv(U_kk): forward elimination of [T_kk | I] by rows, lane = COLUMN (16 lanes the
//                  tile's, 16 the identity's), registers = rows: row j is scaled by a refined reciprocal square root of its pivot,
//                  written to LDS -- it is a finished row of [U_kk | inv(U_kk')] -- and every later row takes its multiple off, the
//                  multiplier read back from that LDS row (the tile is symmetric: the multiplier of row i is the row's own entry i);
//   panel          U_ki = inv(U_kk') T_ki   (i > k):  A operand V_kk, B operand T_ki;
//   trailing part  T_ij -= U_ki' U_kj   (k < i <= j): A operand -U_ki, B operand U_kj.
// What the kernel writes: L.x (column j of L is row j of U: for one register four columns of L, 16 consecutive rows each); with
// EMIT what the matrix-core solve reads beside L.x -- the V tiles (the inverses of the diagonal tiles) register by register, lane by lane;
// for blocks of unequal sizes (RAGGED) also their negated panel tiles, the fragments of csx_trimfma.h --
// 512 contiguous bytes per store; without EMIT the row indices.  64 matrix instructions and about 2 400 vector instructions per
// 64-column block where the bit-identical kernel issues about 11 000 (csparse.py:598-617 is the arithmetic being regrouped).
constexpr int CM_WAVES = 4;
constexpr int CM_RS = 34;           // doubles per LDS row of [U | inv(U')] (32 used)
constexpr int CM_TS = 17;           // doubles per LDS row of the transposed diagonal tile

// RAGGED (blocks of UNEQUAL sizes, launched per size class with `list`): a block of bs columns, 16 (NB - 1) < bs <= 16 NB, is factored
// as a block of 16 NB columns whose last rows / columns are the identity: loads past the block return zero (the range check), the
// padded diagonal is set to one, the stores and the pivot test stop at bs, the fragments go to frag_off[t] (csx_trimfma.h's layout).
template <int NB, bool EMIT, bool RAGGED = false>
__global__ __launch_bounds__(64 * CM_WAVES, 2) void k_chol_block_mfma(const int32_t *__restrict__ start, int32_t nblocks,
                                                                     const int32_t *__restrict__ Ap, const double *__restrict__ Ax,
                                                                     const int32_t *__restrict__ Lp, int32_t *__restrict__ Li,
                                                                     double *__restrict__ Lx, int *notspd, CliqueEmit em,
                                                                     const int32_t *__restrict__ list) {
    typedef double f4 __attribute__((ext_vector_type(4)));
    constexpr int BS = 16 * NB;
    __shared__ __attribute__((aligned(16))) double s_rows[CM_WAVES][16 * CM_RS];
    __shared__ double s_tile[CM_WAVES][16 * CM_TS];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t q = (int64_t)blockIdx.x * CM_WAVES + w;
    if (q >= nblocks) return;      // no workgroup barrier below
    const int64_t t = RAGGED ? (int64_t)__builtin_amdgcn_readfirstlane(list[q]) : q;
    const int32_t c0 = __builtin_amdgcn_readfirstlane(start[t]);
    const int32_t bs = RAGGED ? __builtin_amdgcn_readfirstlane(start[t + 1]) - c0 : BS;
    const int c = lane & 15, rq = lane >> 4;
    double *rows = s_rows[w], *tt = s_tile[w];
    // ---- load: tile (k, i), k <= i: element (16 k + rq + 4 r, 16 i + c) of the upper triangle = entry (that row) of column 16 i + c
    // (every upper part is rows c0 .. column, one each, stored in front: k_clique_min's "dense in front") ----
    f4 T[NB][NB];
    {
        const int64_t a0 = Ap[c0];
        const __amdgpu_buffer_rsrc_t rx = cq_rsrc(Ax + a0, (int)((int64_t)Ap[c0 + bs] - a0) * 8);
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const bool col_in = !RAGGED || 16 * i + c < bs;                  // (a column of the padding has no entries)
            const uint32_t colo = col_in ? (uint32_t)(Ap[c0 + 16 * i + c] - a0) : 0u;
#pragma unroll
            for (int k = 0; k <= i; k++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * k + rq + 4 * r;
                    // (below the diagonal of a diagonal tile: past the upper part of the column -- not read, zero)
                    const bool up = (k < i || rq + 4 * r <= c) && col_in;
                    T[k][i][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, up ? (colo + row) * 8u : 0xfffffff8u, 0, 0));
                    if (RAGGED && k == i && k == NB - 1 && rq + 4 * r == c && 16 * i + c >= bs) T[k][i][r] = 1.0;   // the padding's diagonal
                }
        }
    }
    const int64_t lbase = Lp[c0];
    const __amdgpu_buffer_rsrc_t rl = cq_rsrc(Lx + lbase, (bs * (bs + 1) / 2) * 8);
    const __amdgpu_buffer_rsrc_t ri = cq_rsrc(Li + (EMIT ? 0 : lbase), EMIT ? 0 : (bs * (bs + 1) / 2) * 4);
    double *frag = nullptr;
    if (EMIT) {
        frag = em.frag + (em.frag_off ? (size_t)em.frag_off[t] : (size_t)t * (size_t)(NB * 256)) + lane;   // (equal blocks: the V tiles only)
        if (em.trees) {
            if (lane < bs) em.tree_nodes[c0 + lane] = c0 + lane;
            if (lane == 0) em.trees[t] = Tree{c0, bs};
        }
    }
    double lmax = 0.0, wmax = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        // ---- the diagonal tile: to lane = column, registers = rows (its upper part is what T holds: the rest by symmetry) ----
        __builtin_amdgcn_sched_barrier(0);      // (the phases of a step are kept apart: interleaved across steps by the scheduler, the
                                                // tile factor's sixteen rows sat on top of every other live tile and the kernel spilled)
#pragma unroll
        for (int r = 0; r < 4; r++) tt[(rq + 4 * r) * CM_TS + c] = T[k][k][r];
        cq_wave_sync_lds();
        double a[16];
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = tt[min(i, c) * CM_TS + max(i, c)];
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = (lane - 16) == i ? 1.0 : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double d = cq_bcast(a[j], j);
            if (d <= 0.0 && lane == 0 && 16 * k + j < bs) atomicMin(notspd, c0 + 16 * k + j);   // csparse.py:612
            const double y = cq_rsqrt(d);
            double tv = a[j] * y;                                    // row j of [U | inv(U')], finished
            if (lane < j) tv = 0.0;                                  // (left of the diagonal: eliminated)
            if (lane < 32) rows[j * CM_RS + lane] = tv;
            cq_wave_sync_lds();
#pragma unroll
            for (int i = j + 1; i < 16; i++) a[i] = __builtin_fma(-rows[j * CM_RS + i], tv, a[i]);
        }
        // back to the accumulator layout: U_kk (zero below its diagonal) and V = inv(U_kk) = the transpose of the right half
        __builtin_amdgcn_sched_barrier(0);
        f4 U, V;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rq + 4 * r;
            U[r] = row <= c ? rows[row * CM_RS + c] : 0.0;
            V[r] = rows[c * CM_RS + 16 + row];
            lmax = fmax(lmax, fabs(U[r]));
            wmax = fmax(wmax, fabs(V[r]));
        }
        cq_wave_sync_lds();      // (the next tile's rows go to the same place)
        T[k][k] = U;
        if (EMIT) {
#pragma unroll
            for (int sx = 0; sx < 4; sx++) frag[(size_t)(((RAGGED ? k * (k + 1) / 2 + k : k) * 4 + sx) * 64)] = V[sx];
        }
        // ---- panel: U_ki = inv(U_kk') T_ki; its negative is the A operand of the updates and, with EMIT, the solve's fragment ----
        f4 NU[NB];
#pragma unroll
        for (int i = k + 1; i < NB; i++) {
            f4 D = f4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sx = 0; sx < 4; sx++) D = __builtin_amdgcn_mfma_f64_16x16x4f64(V[sx], T[k][i][sx], D, 0, 0, 0);
            T[k][i] = D;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                NU[i][r] = -D[r];
                lmax = fmax(lmax, fabs(D[r]));
            }
            if (EMIT && RAGGED) {
#pragma unroll
                for (int sx = 0; sx < 4; sx++) frag[(size_t)(((i * (i + 1) / 2 + k) * 4 + sx) * 64)] = NU[i][sx];
            }
        }
        // ---- trailing part ----
#pragma unroll
        for (int i = k + 1; i < NB; i++)
#pragma unroll
            for (int j = i; j < NB; j++)
#pragma unroll
                for (int sx = 0; sx < 4; sx++) T[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(NU[i][sx], T[k][j][sx], T[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- store row block k of U = column block k of L: register r of tile (k, i) is L(16 i + c, J), J = 16 k + rq + 4 r ----
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int J = 16 * k + rq + 4 * r;
            const int colbase = J * bs - J * (J - 1) / 2 - J;          // entry (I, J) of the packed block: colbase + I
#pragma unroll
            for (int i = k; i < NB; i++) {
                const int I = 16 * i + c;
                const bool in = I >= J && (!RAGGED || I < bs);         // (J <= I < bs)
                const double xv = T[k][i][r];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cq_u32x2, xv), rl, in ? (uint32_t)(colbase + I) * 8u : 0xfffffff8u, 0, 0);
                if (!EMIT) __builtin_amdgcn_raw_buffer_store_b32((unsigned int)(c0 + I), ri, in ? (uint32_t)(colbase + I) * 4u : 0xfffffffcu, 0, 0);
            }
        }
    }
    if (EMIT) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lmax = fmax(lmax, __shfl_xor(lmax, d, 64));
            wmax = fmax(wmax, __shfl_xor(wmax, d, 64));
        }
        if (lane == 0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(lmax * wmax);
            if (!(bits <= *(volatile unsigned long long *)em.cond_bits)) atomicMax(em.cond_bits, bits);
        }
    }
}

int chol_clique_numeric(const Csc *A, const CliqueForest &F, Csc *L, int *d_notspd, const CliqueEmit *emit, bool relaxed) {
    hipStream_t s = ctx().stream;
    if (F.nblocks == 0 || A->nnz == 0) return CSX_OK;
    const dim3 grid((unsigned)((F.nblocks + CQ_WAVES - 1) / CQ_WAVES));
    if (emit) {
        // what k_cholsol_mfma can take: equal dense blocks of 16 / 32 / 64 columns; with frag_off (csx_trimfma.hip's size classes):
        // dense blocks of any sizes up to 64
        if (F.sparse) return CSX_EINVAL;
        if (!emit->frag_off && (F.min_bs != F.max_bs || (F.max_bs != 16 && F.max_bs != 32 && F.max_bs != 64))) return CSX_EINVAL;
    }
    CliqueEmit em = emit ? *emit : CliqueEmit();
    em.order = F.order;
#define CSX_CQ_GO(PARTS, SMALL, DENSE, SPARSE, EMIT)                                                                                  \
    hipLaunchKernelGGL((k_chol_clique<PARTS, SMALL, DENSE, SPARSE, EMIT>), grid, dim3(64 * CQ_WAVES), 0, s, F.start, F.nblocks, A->n, \
                       A->nnz, A->p, A->i, A->x, L->p, L->i, L->x, d_notspd, F.colmask, em)
#define CSX_CQ(PARTS)                                                                                                       \
    if (emit && A->nnz < (1 << 29) && F.dense_in_front)                                                                         \
        CSX_CQ_GO(PARTS, true, true, false, true);                                                                              \
    else if (emit && A->nnz < (1 << 29))                                                                                        \
        CSX_CQ_GO(PARTS, true, false, false, true);                                                                             \
    else if (emit)                                                                                                              \
        CSX_CQ_GO(PARTS, false, false, false, true);                                                                            \
    else if (F.sparse && A->nnz < (1 << 29))                                                                                    \
        CSX_CQ_GO(PARTS, true, false, true, false);                                                                             \
    else if (F.sparse)                                                                                                          \
        CSX_CQ_GO(PARTS, false, false, true, false);                                                                            \
    else if (A->nnz < (1 << 29) && F.dense_in_front)                                                                            \
        CSX_CQ_GO(PARTS, true, true, false, false);                                                                             \
    else if (A->nnz < (1 << 29))                                                                                                \
        CSX_CQ_GO(PARTS, true, false, false, false);                                                                            \
    else                                                                                                                        \
        CSX_CQ_GO(PARTS, false, false, false, false)
    if (relaxed && !F.sparse && F.dense_in_front && F.min_bs == F.max_bs && F.max_bs % 16 == 0 && ctx().opt.chol_exact == 0 &&
        !(emit && emit->frag_off)) {
        // equal dense blocks of 16 / 32 / 48 / 64 columns: the blocked factorisation on the matrix cores
        const dim3 g2((unsigned)((F.nblocks + CM_WAVES - 1) / CM_WAVES));
#define CSX_CM(NB)                                                                                                                        \
    if (emit)                                                                                                                             \
        hipLaunchKernelGGL((k_chol_block_mfma<NB, true>), g2, dim3(64 * CM_WAVES), 0, s, F.start, F.nblocks, A->p, A->x, L->p, L->i, L->x, \
                           d_notspd, em, nullptr);                                                                                        \
    else                                                                                                                                  \
        hipLaunchKernelGGL((k_chol_block_mfma<NB, false>), g2, dim3(64 * CM_WAVES), 0, s, F.start, F.nblocks, A->p, A->x, L->p, L->i, L->x, \
                           d_notspd, em, nullptr)
        switch (F.max_bs / 16) {
            case 1: CSX_CM(1); break;
            case 2: CSX_CM(2); break;
            case 3: CSX_CM(3); break;
            default: CSX_CM(4); break;
        }
#undef CSX_CM
        CSX_LAUNCH_CHECK();
        return CSX_OK;
    }
    if (relaxed && !F.sparse && F.dense_in_front && ctx().opt.chol_exact == 0 && emit && emit->frag_off && emit->list) {
        // dense blocks of UNEQUAL sizes (csx_cholsol_factor's size classes): the same kernel per class, padded with the identity
        for (int c = 0; c < 4; c++) {
            const int32_t cnt = emit->cls_start[c + 1] - emit->cls_start[c];
            if (cnt <= 0) continue;
            const dim3 g3((unsigned)((cnt + CM_WAVES - 1) / CM_WAVES));
            const int32_t *lst = emit->list + emit->cls_start[c];
#define CSX_CMR(NB)                                                                                                                    \
    hipLaunchKernelGGL((k_chol_block_mfma<NB, true, true>), g3, dim3(64 * CM_WAVES), 0, s, F.start, cnt, A->p, A->x, L->p, L->i, L->x, \
                       d_notspd, em, lst)
            switch (c) {
                case 0: CSX_CMR(1); break;
                case 1: CSX_CMR(2); break;
                case 2: CSX_CMR(3); break;
                default: CSX_CMR(4); break;
            }
#undef CSX_CMR
            CSX_LAUNCH_CHECK();
        }
        return CSX_OK;
    }
    if (relaxed && A->nnz < (1 << 29)) {      // "chol.exact" = 0: the rounding-equal arithmetic (the common shapes; others stay exact)
        if (emit && F.dense_in_front) {
            hipLaunchKernelGGL((k_chol_clique<7, true, true, false, true, true>), grid, dim3(64 * CQ_WAVES), 0, s, F.start, F.nblocks, A->n,
                               A->nnz, A->p, A->i, A->x, L->p, L->i, L->x, d_notspd, F.colmask, em);
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
        if (emit) {
            hipLaunchKernelGGL((k_chol_clique<7, true, false, false, true, true>), grid, dim3(64 * CQ_WAVES), 0, s, F.start, F.nblocks, A->n,
                               A->nnz, A->p, A->i, A->x, L->p, L->i, L->x, d_notspd, F.colmask, em);
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
        if (!F.sparse) {
            if (F.dense_in_front)
                hipLaunchKernelGGL((k_chol_clique<7, true, true, false, false, true>), grid, dim3(64 * CQ_WAVES), 0, s, F.start, F.nblocks,
                                   A->n, A->nnz, A->p, A->i, A->x, L->p, L->i, L->x, d_notspd, F.colmask, em);
            else
                hipLaunchKernelGGL((k_chol_clique<7, true, false, false, false, true>), grid, dim3(64 * CQ_WAVES), 0, s, F.start, F.nblocks,
                                   A->n, A->nnz, A->p, A->i, A->x, L->p, L->i, L->x, d_notspd, F.colmask, em);
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
    }
    int parts = 7;
#ifdef CSX_ABLATION
    if (const char *e = ablation_env("CSX_CQ_PARTS")) parts = atoi(e);
    switch (parts) {
        case 1: CSX_CQ(1); break;
        case 2: CSX_CQ(2); break;
        case 3: CSX_CQ(3); break;
        case 5: CSX_CQ(5); break;
        case 6: CSX_CQ(6); break;
        case 10: CSX_CQ(10); break;
        default: CSX_CQ(7); break;
    }
#else
    (void)parts;
    CSX_CQ(7);
#endif
#undef CSX_CQ
#undef CSX_CQ_GO
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

// ---- row indices on demand (Csc::rows_pending) -------------------------------------------------------------------------
// every column j holds the rows j, j + 1, ..., j + count - 1: 16 lanes per column
__global__ __launch_bounds__(256) void k_fill_rows(int32_t n, const int32_t *__restrict__ Lp, int32_t *__restrict__ Li) {
    const int t = threadIdx.x & 15;
    const int64_t j64 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (j64 >= n) return;
    const int32_t j = (int32_t)j64, b = Lp[j], e = Lp[j + 1];
    for (int32_t p = b + t; p < e; p += 16) Li[p] = j + (p - b);
}

int csc_fill_rows(Csc *A) {
    if (!A->rows_pending) return CSX_OK;
    if (!A->i) CSX_TRY(dalloc(&A->i, (size_t)A->nnz));
    if (A->n > 0) {
        hipLaunchKernelGGL(k_fill_rows, dim3(blocks_for((int64_t)A->n * 16)), dim3(256), 0, ctx().stream, A->n, A->p, A->i);
        CSX_LAUNCH_CHECK();
    }
    A->rows_pending = false;
    return CSX_OK;
}

// ---- is this L the factor of a forest of equal dense blocks?  (cholsol plan) ------------------------------------------
// 16 lanes per column (aligned 16-byte loads, as k_clique_min): rows j, j + 1, ... contiguous; the count falls by one from
// column to column inside a block and every block starts with the count of column 0;  stats[0] |= no, stats[1] = that count
__global__ __launch_bounds__(256) void k_clique_factor_shape(int32_t n, int32_t nnz, const int32_t *__restrict__ Lp,
                                                             const int32_t *__restrict__ Li, int *stats) {
    const int t = threadIdx.x & 15;
    const int64_t j64 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (j64 >= n) return;
    const int32_t j = (int32_t)j64;
    const int32_t b = Lp[j], cnt = Lp[j + 1] - b;
    bool bad = cnt < 1 || cnt > n - j;
    if (!bad) {
        const int32_t e = b + cnt;
        for (int32_t p0 = b & ~3; p0 < e; p0 += 64) {
            const int32_t p = p0 + 4 * t;
            if (p >= e) continue;
            const int4 v = cq_load4(Li, p, nnz);
            const int32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (p + c >= b && p + c < e && r[c] != j + (p + c - b)) bad = true;
        }
        if (t == 0) {
            const bool first = j == 0 || Lp[j] - Lp[j - 1] == 1;     // the column before ended its block
            if (!first && Lp[j] - Lp[j - 1] != cnt + 1) bad = true;
            if (first && cnt != Lp[1] - Lp[0]) bad = true;
            if (j == 0) stats[1] = cnt;
        }
    }
    if (bad) cq_raise(&stats[0]);
}

int clique_factor_block_size(const Csc *L, int32_t *bs) {
    *bs = 0;
    const int32_t n = L->n;
    if (n <= 0 || L->m != n || !L->x) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int *stats = nullptr;
    CSX_TRY(tmp.alloc(&stats, 4));
    int h[4] = {0, 0, 0, 0};
    CSX_HIP(hipMemsetAsync(stats, 0, sizeof h, s));
    hipLaunchKernelGGL(k_clique_factor_shape, dim3(blocks_for((int64_t)n * 16)), dim3(256), 0, s, n, L->nnz, L->p, L->i, stats);
    CSX_HIP(hipMemcpyAsync(h, stats, sizeof h, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (h[0] == 0 && h[1] >= 1 && n % h[1] == 0) *bs = h[1];
    return CSX_OK;
}

}  // namespace csx
