// cs_chol numeric (csparse.py:561-619) and the solve phase of cs_cholsol
// (csparse.py:640-643) on the device.
//
// The reference factors row by row (up-looking): row k of L is a sparse triangular
// solve against the rows above it, one interpreted loop nest.  The factor itself
// is unique, so the device computes the SAME L column by column (left-looking):
//     L(j:n, j) = ( C(j:n, j) - sum_{k<j, L(j,k)!=0} L(j,k) * L(j:n, k) ) / sqrt(...)
// where the pattern of every column (diagonal first, rows ascending: the order
// cs_chol emits, :606-617) and, per row j, the list of columns k with L(j,k) != 0
// are produced by the host symbolic phase (csx_host.cpp, from cs_ereach walks).
// L.p / L.i are therefore bit-identical to the reference restatement; L.x differs
// by summation order only (<= 1e-10 relative).
//
// Scheduling follows the elimination tree (S.parent):
//   * a forest of SMALL trees (block-diagonal matrices, batches of independent
//     matrices): one wavefront per tree walks its columns in ascending order, the
//     tree's part of L stays in L1/L2;
//   * big trees: columns grouped by height in the tree (all columns of a level are
//     independent), one wavefront per column, one launch per level; runs of
//     narrow levels are walked by ONE workgroup with a barrier per level.
// A column's running sums live in LDS (wave-private), the finished columns it
// reads are streamed from L2/HBM with coalesced loads.
//
// Solve phase for many right-hand sides (B is n-by-k, row-major, overwritten):
//   x = P b; L y = x; L' z = y; b = P' z        (csparse.py:640-643)
// generic path: device permutation + the level-scheduled triangular solves of
// csx_trisolve.hip.  Forest-of-small-trees path: ONE fused kernel; a workgroup
// takes a tree and 64 right-hand sides per wave, keeps those unknowns in LDS,
// runs forward and backward substitution there (one lane per right-hand side, the
// reference's operation order, so every solution is bit-identical to the
// reference's cs_lsolve + cs_ltsolve on the same L) and writes the result back:
// B is read once and written once, L is read once.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "csx_internal.h"
#include "csx_sweep.h"
#include "csx_cholclique.h"
#include "csx_trimfma.h"

namespace csx {

// csx_cholsym.hip
int chol_symbolic_device(const Csc *A, const int32_t *parent, const int32_t *cp, const int32_t *pinv, int32_t **Lp_out,
                         int32_t **Li_out, int32_t **row_ptr_out, int32_t **row_col_out, int32_t **row_pos_out,
                         int32_t *cp_host_out);
// csx_trisolve.hip
struct TriPlan;
int tri_solve_raw(TriPlan *P, double *X, int32_t nrhs, bool relaxed);
int tri_solve_host_raw(TriPlan *P, double *x, bool *taken);
int tri_analyse_raw(const Csc *T, int kind, TriPlan **out);
void tri_set_mate(TriPlan *P, TriPlan *mate);
void tri_set_level_hint(TriPlan *P, std::vector<int32_t> &&level);
void tri_gather_arrays(const TriPlan *P, const int32_t **ptr, const int32_t **idx, const double **val,
                       const double **diag);

constexpr int CH_ACC = 1024;        // column entries kept in LDS per wave
constexpr int CH_WAVES = 4;         // waves per workgroup in the column kernels
constexpr int CH_SMALL_TREE = 512;  // trees up to this many columns go to the tree kernel
constexpr int CH_NARROW = 64;       // levels with <= this many columns: one 16-wave workgroup per column (k_chol_coop)


// ---- scatter of C = upper(P A P') into the pattern of L ------------------------------------
__device__ __forceinline__ int32_t find_row(const int32_t *rows, int32_t len, int32_t r) {
    int32_t lo = 0, hi = len - 1;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (rows[mid] < r) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// the reference scatters x[Ci[p]] = Cx[p] in storage order: of duplicate entries the last wins
__global__ __launch_bounds__(256) void k_chol_winner(int32_t n, const int32_t *__restrict__ Ap,
                                                     const int32_t *__restrict__ Ai, const int32_t *__restrict__ pinv,
                                                     const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                                     int32_t *win, int *bad) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int32_t j2 = pinv ? pinv[j] : (int32_t)j;
    for (int32_t p = Ap[j] + lane; p < Ap[j + 1]; p += 64) {
        const int32_t i = Ai[p];
        if (i > j) continue;
        const int32_t i2 = pinv ? pinv[i] : i;
        const int32_t c = min(i2, j2), r = max(i2, j2);
        const int32_t b = Lp[c], len = Lp[c + 1] - b;
        const int32_t t = find_row(Li + b, len, r);
        if (Li[b + t] != r) {
            *bad = 1;  // the symbolic pattern does not contain this entry
            continue;
        }
        atomicMax(&win[b + t], p);
    }
}

__global__ __launch_bounds__(256) void k_chol_init(int64_t lnz, const int32_t *__restrict__ win,
                                                   const double *__restrict__ Ax, double *__restrict__ Lx) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < lnz) Lx[q] = win[q] >= 0 ? Ax[win[q]] : 0.0;
}

// ---- one column ---------------------------------------------------------------------------------
// acc_v / acc_r: wave-private LDS (CH_ACC doubles / ints).  Columns longer than CH_ACC keep their
// sums in Lx itself (global), the row search then runs on Li in global memory.
__device__ __forceinline__ void chol_column(int32_t j, const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                            double *Lx, const int32_t *__restrict__ row_ptr,
                                            const int32_t *__restrict__ row_col, const int32_t *__restrict__ row_pos,
                                            double *acc_v, int32_t *acc_r, int lane, int *notspd) {
    const int32_t base = Lp[j], len = Lp[j + 1] - base;
    const bool in_lds = len <= CH_ACC;
    if (in_lds) {
        for (int32_t t = lane; t < len; t += 64) {
            acc_v[t] = Lx[base + t];
            acc_r[t] = Li[base + t];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t qe = row_ptr[j + 1] - 1;   // the row view ends with the diagonal L(j,j)
    // Updates are applied in order, but fetched eight at a time: lane u reads the descriptor of update u
    // (position of L(j,k), end of column k, L(j,k)) -- one dependent round trip for eight updates instead of
    // one each -- and the heads of the eight columns are requested together before any is consumed.
    constexpr int UQ = 8;
    for (int32_t q0 = row_ptr[j]; q0 < qe; q0 += UQ) {
        int32_t posq = 0, kendq = 0;
        double ljkq = 0.0;
        if (lane < UQ && q0 + lane < qe) {
            const int32_t kq = row_col[q0 + lane];
            posq = row_pos[q0 + lane];
            kendq = Lp[kq + 1];
            ljkq = Lx[posq];
        }
        int32_t pos_[UQ], kend_[UQ], r_[UQ];
        double ljk_[UQ], v_[UQ];
#pragma unroll
        for (int u = 0; u < UQ; u++) {
            pos_[u] = __builtin_amdgcn_readlane(posq, u);
            kend_[u] = __builtin_amdgcn_readlane(kendq, u);    // 0 for an absent update: nothing below
            ljk_[u] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ljkq), u),
                                       __builtin_amdgcn_readlane(__double2loint(ljkq), u));
        }
#pragma unroll
        for (int u = 0; u < UQ; u++) {
            const int32_t p = pos_[u] + lane;
            const int32_t pp = p < kend_[u] ? p : pos_[u];      // a valid address either way
            r_[u] = Li[pp];
            v_[u] = Lx[pp];
        }
#pragma unroll
        for (int u = 0; u < UQ; u++) {
            if (pos_[u] + lane < kend_[u]) {
                const double v = v_[u] * ljk_[u];
                if (in_lds) acc_v[find_row(acc_r, len, r_[u])] -= v;
                else Lx[base + find_row(Li + base, len, r_[u])] -= v;
            }
            for (int32_t p = pos_[u] + 64 + lane; p < kend_[u]; p += 64) {   // columns longer than one wave
                const double v = Lx[p] * ljk_[u];
                if (in_lds) acc_v[find_row(acc_r, len, Li[p])] -= v;
                else Lx[base + find_row(Li + base, len, Li[p])] -= v;
            }
            __builtin_amdgcn_wave_barrier();   // one update after the other: they may hit the same rows
        }
    }
    const double d = in_lds ? acc_v[0] : Lx[base];
    if (d <= 0.0 && lane == 0) atomicMin(notspd, j);  // csparse.py:612: not positive definite
    const double ljj = sqrt(d);
    __builtin_amdgcn_wave_barrier();
    for (int32_t t = lane; t < len; t += 64) {
        const double v = in_lds ? acc_v[t] : Lx[base + t];
        Lx[base + t] = t == 0 ? ljj : v / ljj;
    }
    __builtin_amdgcn_wave_barrier();
}

#define CH_SHARED                                              \
    __shared__ double s_acc_v[CH_WAVES][CH_ACC];               \
    __shared__ int32_t s_acc_r[CH_WAVES][CH_ACC];

// one wave per column of a level
__global__ __launch_bounds__(64 * CH_WAVES) void k_chol_level(const int32_t *__restrict__ cols, int32_t count,
                                                             const int32_t *__restrict__ Lp,
                                                             const int32_t *__restrict__ Li, double *Lx,
                                                             const int32_t *__restrict__ row_ptr,
                                                             const int32_t *__restrict__ row_col,
                                                             const int32_t *__restrict__ row_pos, int *notspd) {
    CH_SHARED
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * CH_WAVES + w;
    if (c >= count) return;
    chol_column(cols[c], Lp, Li, Lx, row_ptr, row_col, row_pos, s_acc_v[w], s_acc_r[w], lane, notspd);
}

// One workgroup walks the narrow levels [l0, l1) of a big tree, ALL of its 16 waves working on one
// column at a time.  A chain-like elimination tree (bcsstk16 in natural order: 4810 levels for 4884
// columns) leaves no parallelism between columns, and one wave per column spends its time in dependent
// round trips: row view -> L(j,k) -> column k, ~1.2 us per update, 125 updates per column.  Here the
// updates of a column are dealt to W waves (wave w takes updates w, w+W, ... in order), each wave fetches
// the descriptors of up to CC_Q updates in one round trip and the heads of their columns in another, and
// sums its updates into its OWN partial column in LDS (no atomics: the lanes of one update hit distinct
// rows); the partials are then subtracted from the column in wave order.  So the result is deterministic
// (same bits every run) and agrees with the reference to rounding.  A row -> position map in LDS
// (n <= CC_MAP) replaces the binary search.  W = min(16, CC_ACC / column length); columns longer than
// CC_ACC are updated in place by one wave.
constexpr int CC_WAVES = 16, CC_ACC = 8192, CC_MAP = 12288, CC_Q = 8;
constexpr int CC_RUN_MIN = 16, CC_RUN_MAX = 64;   // narrow levels are factored in two phases, 16 .. 64 levels at a time (see k_chol_coop)

__device__ __forceinline__ int32_t cc_lookup(bool use_map, const int32_t *map, const int32_t *rows, int32_t len, int32_t r) {
    return use_map ? map[r] : find_row(rows, len, r);
}

__global__ __launch_bounds__(64 * CC_WAVES) void k_chol_coop(const int32_t *__restrict__ cols,
                                                            const int32_t *__restrict__ level_ptr, int32_t l0, int32_t l1,
                                                            const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                                            double *Lx, const int32_t *__restrict__ row_ptr,
                                                            const int32_t *__restrict__ row_col,
                                                            const int32_t *__restrict__ row_pos, int32_t n, int *notspd,
                                                            const int32_t *__restrict__ col_level, int mode,
                                                            int32_t lf, int32_t *__restrict__ split) {
    // mode 0: every update of a column, then pivot and scaling.  A long RUN of narrow levels lf .. (the separators at
    // the top of a nested-dissection tree: thousands of columns, a few per level) is done in two phases instead.
    // mode 1, one launch, one workgroup per column of the whole run (levels l0 .. l1-1): the updates that come from
    // columns BELOW the run (level < lf) -- all final before the run starts, so every column of the run takes them at
    // the same time -- and the partly updated column is stored.  mode 2, level after level as in mode 0: the updates
    // from INSIDE the run (level >= lf), then pivot and scaling.  col_level[k] = level of column k, -1 off the list.
    // A row's updates are listed in ascending column order and a nested-dissection ordering numbers the separators
    // last, so the inside updates are the tail of the list: mode 1 leaves the position of the first one in split[j]
    // (atomicMin) and mode 2 starts there instead of reading thousands of descriptors to find a few dozen.
    extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
    double *part = reinterpret_cast<double *>(cc_smem);             // W partial columns of `len` doubles
    int32_t *acc_r = reinterpret_cast<int32_t *>(part + CC_ACC);    // the column's row indices (<= CC_ACC kept)
    int32_t *map = acc_r + CC_ACC;
    const bool use_map = n <= CC_MAP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int32_t l = l0; l < (mode == 1 ? l0 + 1 : l1); l++) {
        // mode 1: the run's columns are one contiguous piece of the level list, a workgroup each
        const int32_t lfirst = level_ptr[l], lcount = mode == 1 ? level_ptr[l1] - lfirst : level_ptr[l + 1] - lfirst;
        // the columns of a level do not depend on each other: launched over ONE level with several workgroups, each
        // takes its share of them; launched with one workgroup over a run of levels, it walks them all
        for (int32_t c = blockIdx.x; c < lcount; c += gridDim.x) {
            const int32_t j = cols[lfirst + c];
            const int32_t base = Lp[j], len = Lp[j + 1] - base;
            const int W = len <= CC_ACC ? min(CC_WAVES, CC_ACC / max(len, 1)) : 0;   // 0: in place, wave 0 alone
            for (int32_t t = tid; t < len; t += 64 * CC_WAVES) {
                const int32_t r = Li[base + t];
                if (W) acc_r[t] = r;
                if (use_map) map[r] = t;
            }
            for (int32_t t = tid; t < W * len; t += 64 * CC_WAVES) part[t] = 0.0;
            __syncthreads();
            const int32_t *rows = W ? acc_r : Li + base;
            const int32_t qe = row_ptr[j + 1] - 1;                     // the row view ends with the diagonal
            const int32_t qb = mode == 2 ? max(row_ptr[j], min(split[j], qe)) : row_ptr[j];
            const int Wq = W ? W : 1;
            if (w < Wq) {
                double *mine = part + (size_t)w * len;
                for (int32_t q0 = qb + w; q0 < qe; q0 += Wq * CC_Q) {
                    // lane u < CC_Q fetches the descriptor of this wave's u-th update: position of L(j,k), end of column k
                    int32_t posq = 0, kendq = 0;
                    double ljkq = 0.0;
                    {
                        const int32_t qu = q0 + Wq * lane;
                        if (lane < CC_Q && qu < qe) {
                            const int32_t kq = row_col[qu];
                            const bool inside = mode != 0 && col_level[kq] >= lf;
                            if (mode == 1 && inside) atomicMin(&split[j], qu);
                            if (mode == 0 || (mode == 1) != inside) {   // mode 1 takes the outside, mode 2 the inside
                                posq = row_pos[qu];
                                kendq = Lp[kq + 1];
                                ljkq = Lx[posq];
                            }
                        }
                    }
                    int32_t pos_[CC_Q], kend_[CC_Q], r_[CC_Q];
                    double ljk_[CC_Q], v_[CC_Q];
#pragma unroll
                    for (int u = 0; u < CC_Q; u++) {
                        pos_[u] = __builtin_amdgcn_readlane(posq, u);
                        kend_[u] = __builtin_amdgcn_readlane(kendq, u);    // 0 for an absent update: nothing below
                        ljk_[u] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ljkq), u),
                                                   __builtin_amdgcn_readlane(__double2loint(ljkq), u));
                    }
#pragma unroll
                    for (int u = 0; u < CC_Q; u++) {
                        const int32_t p = pos_[u] + lane;
                        const int32_t pp = p < kend_[u] ? p : pos_[u];      // a valid address either way
                        r_[u] = Li[pp];
                        v_[u] = Lx[pp];
                    }
#pragma unroll
                    for (int u = 0; u < CC_Q; u++) {
                        if (pos_[u] + lane < kend_[u]) {
                            const int32_t t = cc_lookup(use_map, map, rows, len, r_[u]);
                            const double dv = v_[u] * ljk_[u];
                            if (W) mine[t] += dv;
                            else Lx[base + t] -= dv;
                        }
                        for (int32_t p = pos_[u] + 64 + lane; p < kend_[u]; p += 64) {   // columns longer than one wave
                            const int32_t t = cc_lookup(use_map, map, rows, len, Li[p]);
                            const double dv = Lx[p] * ljk_[u];
                            if (W) mine[t] += dv;
                            else Lx[base + t] -= dv;
                        }
                        __builtin_amdgcn_wave_barrier();   // updates of one wave are applied one after another
                    }
                }
            }
            __syncthreads();
            // column = initial values - partials, in wave order; then the pivot and the scaling
            for (int32_t t = tid; t < len; t += 64 * CC_WAVES) {
                double v = Lx[base + t];
                for (int ww = 0; ww < W; ww++) v -= part[(size_t)ww * len + t];
                if (W) part[t] = v;                       // partial 0 is dead: reuse it for the finished sums
                else if (t == 0) part[0] = v;
            }
            __syncthreads();
            if (mode == 1) {                                  // partly updated column: the inside pass finishes it
                if (W)
                    for (int32_t t = tid; t < len; t += 64 * CC_WAVES) Lx[base + t] = part[t];
                __syncthreads();
                continue;
            }
            const double d = part[0];
            if (d <= 0.0 && tid == 0) atomicMin(notspd, j);  // csparse.py:612: not positive definite
            const double ljj = sqrt(d);
            for (int32_t t = tid; t < len; t += 64 * CC_WAVES) {
                const double v = W ? part[t] : Lx[base + t];
                Lx[base + t] = t == 0 ? ljj : v / ljj;
            }
            __syncthreads();   // the next column reads these values and reuses the LDS arrays
        }
    }
}

// one wave per small tree: its columns in ascending order (children before parents)
__global__ __launch_bounds__(64 * CH_WAVES) void k_chol_trees(const Tree *__restrict__ trees, int32_t ntrees,
                                                             const int32_t *__restrict__ tree_cols,
                                                             const int32_t *__restrict__ Lp,
                                                             const int32_t *__restrict__ Li, double *Lx,
                                                             const int32_t *__restrict__ row_ptr,
                                                             const int32_t *__restrict__ row_col,
                                                             const int32_t *__restrict__ row_pos, int *notspd) {
    CH_SHARED
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * CH_WAVES + w;
    if (t >= ntrees) return;
    const Tree tr = trees[t];
    for (int32_t c = 0; c < tr.count; c++)
        chol_column(tree_cols[tr.first + c], Lp, Li, Lx, row_ptr, row_col, row_pos, s_acc_v[w], s_acc_r[w], lane,
                    notspd);
}

// one wave per DENSE small tree (a dense lower-triangular block on consecutive columns, <= 64 of them:
// the G-spd benchmark's blocks).  The block's columns are contiguous in L.x, so the whole block is copied
// to LDS, factored there, and copied back: L.x is read once and written once.  Right-looking, lane = row;
// every entry L(i,c) receives its updates - L(i,j) L(c,j) for j = 0, 1, ... in ascending order, multiply
// then subtract, then one division by the pivot -- the operation sequence of cs_chol's up-looking row
// solve on a chain elimination tree (csparse.py:598-612), so the factor is bit-identical to it.
constexpr int CD_MAX = 64;
__global__ __launch_bounds__(64 * CH_WAVES) void k_chol_dense_trees(const Tree *__restrict__ trees, int32_t ntrees,
                                                                   const int32_t *__restrict__ tree_cols,
                                                                   const int32_t *__restrict__ Lp, double *Lx,
                                                                   int *notspd) {
#pragma clang fp contract(off)
    __shared__ double sa[CH_WAVES][CD_MAX * (CD_MAX + 1) / 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * CH_WAVES + w;
    if (t >= ntrees) return;   // no workgroup barrier below
    const Tree tr = trees[t];
    const int bs = tr.count;
    const int32_t c0 = tree_cols[tr.first];
    const int64_t base = Lp[c0];
    const int nent = bs * (bs + 1) / 2;
    double *a = sa[w];
    for (int e = lane; e < nent; e += 64) a[e] = Lx[base + e];
    __builtin_amdgcn_wave_barrier();
    const bool row = lane < bs;
    for (int j = 0; j < bs; j++) {
        const int offj = j * bs - j * (j - 1) / 2;      // start of column j (its diagonal)
        const double d = a[offj];
        if (d <= 0.0 && lane == 0) atomicMin(notspd, c0 + j);   // csparse.py:612
        const double ljj = sqrt(d);
        double lij = 0.0;
        if (row && lane > j) {
            lij = a[offj + lane - j] / ljj;
            a[offj + lane - j] = lij;
        }
        if (lane == j) a[offj] = ljj;
        int offc = offj;
#pragma unroll 4
        for (int c = j + 1; c < bs; c++) {
            offc += bs - (c - 1);                        // start of column c
            // L(c,j) from lane c: read with every lane enabled, then predicate only the update
            const int lo = __builtin_amdgcn_readlane(__double2loint(lij), c);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(lij), c);
            const double lcj = __hiloint2double(hi, lo);
            if (row && lane >= c) a[offc + lane - c] -= lij * lcj;
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int e = lane; e < nent; e += 64) Lx[base + e] = a[e];
}

template <class T>
static int upload(T **d, const std::vector<T> &h) {
    CSX_TRY(dalloc(d, h.size()));
    if (!h.empty())
        CSX_HIP(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
    return CSX_OK;
}


// ---- banded factors: the whole window in registers --------------------------------------------------------
// bcsstk16 in natural order: n = 4 884, L is 98.8 % dense inside a band of half-width 140 and its elimination
// tree is a chain (4 810 levels).  Column after column with 125 sparse updates each took 32 ms here, 20 ms on one
// host core.  For such a factor the live part of a RIGHT-looking factorisation is small: element (r, c) is touched
// by the columns r - b .. c - 1 only, so at column j the live elements are j <= c <= r <= j + b -- a triangle of
// side b + 1 (10 K doubles for bcsstk16).  One workgroup keeps that window in REGISTERS (a BW x BW circular
// array spread over 1024 threads, thread = (row slot, a few column slots)), and per column: the entering row
// is fetched through the row view into a staging row, the column's elements are gathered into LDS, scaled
// (sqrt, divisions) and written to L.x, and every thread subtracts l_r l_c from its live elements.  Two
// workgroup barriers per column, no memory traffic but the entering row and the leaving column.  Each element
// receives its updates in ascending column order, multiply and subtract rounded separately: on a chain tree that
// is the reference's operation sequence, so L.x comes out bit-identical (tests/test_gpu_cholesky.py).
#pragma clang fp contract(off)
template <int BW, int THREADS>
__global__ __launch_bounds__(THREADS) void k_chol_band(int32_t n, const int32_t *__restrict__ Lp,
                                                       const int32_t *__restrict__ Li, double *Lx,
                                                       const int32_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ row_col,
                                                       const int32_t *__restrict__ row_pos, int *notspd) {
    constexpr int TPR = THREADS / BW;                // threads per window row
    constexpr int NS = (BW + TPR - 1) / TPR;         // elements of the row per thread
    __shared__ double colbuf[BW], lcol[BW], stage[2][BW];
    const int tid = threadIdx.x;
    const int rr = tid / TPR, tc = tid % TPR;
    const bool owner = rr < BW;
    // Row slot rr holds the window row r = j + dr (r = rr mod BW); of that row this thread keeps the elements on the
    // diagonals dd = tc, tc + TPR, ... (dd = r - c never changes during an element's life).  Element (r, c) is live
    // while c >= j, i.e. dd <= dr: a row enters with all BW diagonals live and loses one per column, so the loops
    // below stop at dd > dr -- and the threads of a wave belong to neighbouring rows, whose dr differ by a few.
    double W[NS];
#pragma unroll
    for (int sl = 0; sl < NS; sl++) W[sl] = 0.0;
    int dr = owner ? (rr + BW - 1) % BW : -1;        // at j = -(BW - 1); -1: not an owner, every loop below is empty
    // The entering row is fetched through three dependent loads (row extent -> column / position -> value) and then
    // staged in LDS by diagonal.  The four stages are spread over four columns: each step issues one load of each
    // kind, for the rows entering 4, 3 and 2 steps later, stages the row entering at the NEXT step from what was
    // loaded a whole step earlier, and takes this step's row out of the other half of the staging buffer -- no step
    // waits for a round trip, and the staging needs no barrier of its own: a column costs two barriers.
    // (A row has at most BW entries: thread t < BW handles entry t.)
    const int32_t j0 = -(BW - 1);
    auto row_extent = [&](int32_t r, int32_t &qb, int32_t &qe) {
        qb = qe = 0;
        if (r >= 0 && r < n) {
            qb = row_ptr[r];
            qe = row_ptr[r + 1];
        }
    };
    auto row_entry = [&](int32_t qb, int32_t qe, int32_t &c, int32_t &p) {
        c = 0;
        p = -1;
        if (qb + tid < qe) {
            c = row_col[qb + tid];
            p = row_pos[qb + tid];
        }
    };
    // at the top of step j: row j + BW - 1 is staged; (c1, p1, v1) = row j + BW; (c2, p2) = row j + BW + 1; extent of
    // row j + BW + 2
    int32_t c1, p1, c2, p2, qb3, qe3;
    double v1;
    if (tid < BW) {
        stage[0][tid] = 0.0;
        stage[1][tid] = 0.0;
    }
    __syncthreads();
    {   // prologue: the state at the top of step j0
        int32_t qb, qe, c0, p0;
        row_extent(j0 + BW - 1, qb, qe);
        row_entry(qb, qe, c0, p0);
        if (p0 >= 0) stage[(j0 & 1)][j0 + BW - 1 - c0] = Lx[p0];
        row_extent(j0 + BW, qb, qe);
        row_entry(qb, qe, c1, p1);
        v1 = p1 >= 0 ? Lx[p1] : 0.0;
        row_extent(j0 + BW + 1, qb, qe);
        row_entry(qb, qe, c2, p2);
        row_extent(j0 + BW + 2, qb3, qe3);
    }
    int32_t ncb = 0, nce = 0, nrow = 0;              // next column's extent and this thread's row index in it
    if (n > 0) {
        ncb = Lp[0];
        nce = Lp[1];
        nrow = ncb + tid < nce ? Li[ncb + tid] : 0;
    }
    __syncthreads();
    for (int32_t j = j0; j < n; j++) {
        const int par = j & 1;
        // ---- pipeline: this step's loads, each independent of the others ----
        int32_t qb4, qe4, c3, p3;
        row_extent(j + BW + 3, qb4, qe4);            // extent of the row entering four steps from now
        row_entry(qb3, qe3, c3, p3);                 // entries of the row entering three steps from now
        const double v2 = p2 >= 0 ? Lx[p2] : 0.0;    // values of the row entering two steps from now
        // ---- 1. the row that enters the window now, r_in = j + BW - 1, was staged during the last step ----
        if (dr == BW - 1) {
#pragma unroll
            for (int sl = 0; sl < NS; sl++)
                if (tc + TPR * sl < BW) W[sl] = stage[par][tc + TPR * sl];
        }
        // ---- and the one entering at the next step is staged now: element (r, c) sits on diagonal r - c ----
        if (p1 >= 0) stage[par ^ 1][j + BW - c1] = v1;    // that half was zeroed during the last step
        // this step's column descriptor was fetched one step ago; fetch the next one now
        const int32_t cb_ = ncb, ce_ = nce, myrow = nrow;
        if (j + 1 >= 0 && j + 1 < n) {
            ncb = Lp[j + 1];
            nce = Lp[j + 2];
            nrow = ncb + tid < nce ? Li[ncb + tid] : 0;
        }
        if (j >= 0) {
            // ---- 2. column j is complete: element (r, j) is the one on diagonal dd == dr of row r ----
            if (dr >= 0 && dr % TPR == tc) {
#pragma unroll
                for (int sl = 0; sl < NS; sl++)
                    if (tc + TPR * sl == dr) colbuf[dr] = W[sl];
            }
            lds_barrier();
            if (tid < BW) stage[par][tid] = 0.0;                // taken out above by every owner: free for step j + 1's staging
            const double d = colbuf[0];
            if (!(d > 0.0)) {                                   // not positive definite (uniform)
                if (tid == 0) atomicMin(notspd, j);
                break;
            }
            const double ljj = sqrt(d);
            if (tid < BW) lcol[tid] = tid == 0 ? ljj : colbuf[tid] / ljj;
            if (cb_ + tid < ce_) Lx[cb_ + tid] = myrow == j ? ljj : colbuf[myrow - j] / ljj;   // <= BW entries per column
            lds_barrier();
            // ---- 3. rank-1 update of the live elements: c > j  <=>  dd < dr ----
            if (dr >= 1) {
                const double lr = lcol[dr];
#pragma unroll
                for (int sl = 0; sl < NS; sl++) {
                    const int dd = tc + TPR * sl;
                    if (dd >= dr) break;
                    const double t = lr * lcol[dr - dd];
                    W[sl] = W[sl] - t;
                }
            }
        } else {
            lds_barrier();
            if (tid < BW) stage[par][tid] = 0.0;
            lds_barrier();
        }
        c1 = c2;
        p1 = p2;
        v1 = v2;
        c2 = c3;
        p2 = p3;
        qb3 = qb4;
        qe3 = qe4;
        if (owner) dr = dr == 0 ? BW - 1 : dr - 1;
    }
}
#pragma clang fp contract(fast)

__global__ __launch_bounds__(256) void k_band_width(int32_t n, const int32_t *__restrict__ Lp,
                                                    const int32_t *__restrict__ Li, int *bw) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) atomicMax(bw, Li[Lp[j + 1] - 1] - (int32_t)j);    // rows ascending: the last entry is the lowest row
}

// csx_cholband.hip: blocked factorisation in a dense band array, for chain-like factors of any band width
size_t chol_wide_band_bytes(int32_t n, int32_t bw);
int chol_wide_band(int32_t n, int32_t bw, const int32_t *Lp, const int32_t *Li, double *Lx, int *notspd, int nb);

// csx_cholband.hip: fundamental supernodes (a, w, r) factored in place as dense trapezoids
struct SnDesc {
    int32_t a, w, r;
};
int chol_supernodes(const void *d_sns, int32_t nsn, int32_t max_w, int32_t max_rows, const int32_t *Lp, double *Lx,
                    int *notspd);
constexpr int32_t SN_MIN_WIDTH = 8;    // narrower chains stay with the column kernels (700 x 700 grid, order 1, numeric part: 32 -> 44 ms, 16 -> 36, 8 -> 31, 4 -> 29)

struct Forest {
    std::vector<Tree> small;             // trees handled by the tree kernel
    std::vector<int32_t> small_cols;     // their columns, ascending inside a tree
    std::vector<int32_t> level_cols;     // columns of big trees sorted by level
    std::vector<int32_t> level_ptr;
    int32_t max_tree = 0;
};

// Partition the elimination forest (parent[]) into small trees and level sets of the big ones.
static void partition_forest(int32_t n, const int32_t *parent, Forest &F) {
    std::vector<int32_t> root((size_t)n), size((size_t)n, 0);
    for (int32_t j = n - 1; j >= 0; j--) root[(size_t)j] = parent[j] < 0 ? j : root[(size_t)parent[j]];
    for (int32_t j = 0; j < n; j++) size[(size_t)root[(size_t)j]]++;
    std::vector<int32_t> slot((size_t)n, -1);
    for (int32_t j = 0; j < n; j++) {
        if (parent[j] >= 0) continue;
        F.max_tree = std::max(F.max_tree, size[(size_t)j]);
        if (size[(size_t)j] <= CH_SMALL_TREE) {
            slot[(size_t)j] = (int32_t)F.small.size();
            F.small.push_back({0, size[(size_t)j]});
        }
    }
    int32_t run = 0;
    for (auto &t : F.small) {
        t.first = run;
        run += t.count;
        t.count = 0;
    }
    F.small_cols.assign((size_t)run, 0);
    std::vector<int32_t> level((size_t)n, 0);
    int32_t nlev = 0;
    for (int32_t j = 0; j < n; j++) {
        const int32_t s = slot[(size_t)root[(size_t)j]];
        if (s >= 0) {
            Tree &t = F.small[(size_t)s];
            F.small_cols[(size_t)(t.first + t.count++)] = j;
        } else {
            nlev = std::max(nlev, level[(size_t)j] + 1);
            if (parent[j] >= 0) level[(size_t)parent[j]] = std::max(level[(size_t)parent[j]], level[(size_t)j] + 1);
        }
    }
    F.level_ptr.assign((size_t)nlev + 1, 0);
    for (int32_t j = 0; j < n; j++)
        if (slot[(size_t)root[(size_t)j]] < 0) F.level_ptr[(size_t)level[(size_t)j] + 1]++;
    for (int32_t l = 0; l < nlev; l++) F.level_ptr[(size_t)l + 1] += F.level_ptr[(size_t)l];
    F.level_cols.assign((size_t)F.level_ptr[(size_t)nlev], 0);
    std::vector<int32_t> fill(F.level_ptr.begin(), F.level_ptr.end() - 1);
    for (int32_t j = 0; j < n; j++)
        if (slot[(size_t)root[(size_t)j]] < 0) F.level_cols[(size_t)fill[(size_t)level[(size_t)j]]++] = j;
}

// what the last csx_chol did (csx_chol_info): 1 = forest of cliques, 2 = forest of small sparse trees (csx_cholclique.hip), 0 = the general path; HIP-event
// time of its numeric part (k_chol_clique alone / everything after the pattern of L on the general path)
static int g_chol_path = -1;
static double g_chol_numeric_ms = 0.0;

static int chol_device(const Csc *A, const int32_t *parent, const int32_t *cp, const int32_t *pinv, Csc *L) {
    hipStream_t s = ctx().stream;
    g_chol_path = -1;
    g_chol_numeric_ms = 0.0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    struct EvGuard {
        hipEvent_t &a, &b;
        ~EvGuard() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev_guard{ev_a, ev_b};
    if (hipEventCreate(&ev_a) != hipSuccess || hipEventCreate(&ev_b) != hipSuccess) return CSX_ERUNTIME;
    auto numeric_ms = [&]() {   // after the stream has been synchronised
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev_a, ev_b) == hipSuccess) g_chol_numeric_ms = ms;
    };
    const int32_t n = A->n;
    L->m = L->n = n;
    L->nnz = cp[n];
    L->owns = true;
    if (n == 0) {
        CSX_TRY(dalloc(&L->p, 1));
        CSX_HIP(hipMemsetAsync(L->p, 0, sizeof(int32_t), s));
        CSX_TRY(dalloc(&L->i, 0));
        CSX_TRY(dalloc(&L->x, 0));
        return CSX_OK;
    }
    // pattern of L and its row view, on the device (csx_cholsym.hip); the host only orders the tree
    const bool timing = getenv("CSX_CHOL_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[cs_chol] %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    if (ctx().opt.chol_clique && ctx().opt.chol_dense_trees && !pinv && A->nnz > 0) {
        // A forest of cliques on consecutive columns (csx_cholclique.hip): L.p / L.i follow from the counts, the values are
        // one read of A's upper part and one write of L, a block to a wave.  The caller's S must be that forest's.
        CliqueForest F;
        bool ok = false, same = false;
        int st = CSX_OK;
        const bool cached = A->clique != nullptr && (!A->clique->sparse || ctx().opt.chol_forest);
        if (cached) {
            // csx_schol's finding for this matrix (dropped by csx_csc_invalidate when the arrays change): a shallow copy; L.p is
            // copied out of it below instead of taken
            F = *A->clique;
            ok = true;
        } else {
            st = clique_forest(A, &F, &ok);
        }
        if (st == CSX_OK && ok && F.ascending && F.max_bs <= CLIQUE_MAX_BLOCK) {
            lap("clique forest");
            // The block kernel is started FIRST; the caller's S (host arrays) is uploaded and compared with the forest's beside it
            // on a stream of its own.  An S that is not A's costs a factor that is thrown away.
            // (the kernel writes L through the forest's own column pointers: an S with another lnz is refused before anything runs)
            if ((int64_t)L->nnz != F.lnz) {
                if (!cached) free_clique(&F);
                return CSX_EINVAL;
            }
            CliqueCompare cmp;
            st = clique_matches_begin(F, parent, cp, &cmp);
            int *d_notspd = nullptr;
            constexpr int NOTSPD_NONE = 0x7f7f7f7f;     // the flag's idle value: set by a byte fill on the device, no host slot in flight
            int hflag = NOTSPD_NONE;
            bool queued = false;                        // device work of this call may be in flight
            if (st == CSX_OK) st = dalloc(&L->i, (size_t)L->nnz);
            if (st == CSX_OK) st = dalloc(&L->x, (size_t)L->nnz);
            if (st == CSX_OK) st = dalloc(&d_notspd, 1);
            if (st == CSX_OK && cached) {
                st = dalloc(&L->p, (size_t)n + 1);
                if (st == CSX_OK && hipMemcpyAsync(L->p, F.cp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
                    st = CSX_ERUNTIME;
            }
            int32_t *own_cp = nullptr;                  // not cached: L takes the forest's column pointers
            if (st == CSX_OK) {
                if (!cached) {
                    L->p = F.cp;
                    own_cp = F.cp;
                }
                if (hipMemsetAsync(d_notspd, 0x7f, sizeof(int), s) != hipSuccess) st = CSX_ERUNTIME;
                queued = true;
            }
            lap("allocations");
            if (st == CSX_OK) (void)hipEventRecord(ev_a, s);
            if (st == CSX_OK) st = chol_clique_numeric(A, F, L, d_notspd, nullptr, !ctx().opt.chol_exact);
            if (st == CSX_OK) (void)hipEventRecord(ev_b, s);
            lap("block kernel");
            if (st == CSX_OK) st = clique_matches_run(&cmp);
            lap("S uploaded");
            if (st == CSX_OK && (hipMemcpyAsync(&hflag, d_notspd, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
                                 hipStreamSynchronize(s) != hipSuccess)) {
                set_error("cs_chol: %s", hipGetErrorString(hipGetLastError()));
                st = CSX_ERUNTIME;
            }
            {
                const int st2 = clique_matches_end(&cmp, &same);    // (always: it waits for the side stream and frees)
                if (st == CSX_OK) st = st2;
            }
            if (own_cp) F.cp = nullptr;                 // L owns it now (the caller frees L on any error)
            // on a failure after work was queued the block kernel may still be writing L.i / L.x, which the caller hands back to
            // the pool: nothing is released before the stream has drained
            if (st != CSX_OK && queued) (void)hipStreamSynchronize(s);
            lap("numeric (blocks) + S compared");
            if (st == CSX_OK && !same) st = CSX_EINVAL;        // S.cp / S.parent do not belong to A
            if (st == CSX_OK) {
                numeric_ms();
                g_chol_path = F.sparse ? 2 : 1;
            }
            dfree(d_notspd);
            if (!cached) free_clique(&F);
            if (st != CSX_OK) return st;
            return hflag != NOTSPD_NONE ? CSX_ENOTSPD : CSX_OK;
        }
        if (!cached) free_clique(&F);
        if (st != CSX_OK) return st;
        lap("no clique forest");
    }
    int32_t *d_rp = nullptr, *d_rc = nullptr, *d_rpos = nullptr, *d_pinv = nullptr, *d_win = nullptr;
    int32_t *d_small_cols = nullptr, *d_level_cols = nullptr, *d_level_ptr = nullptr;
    Tree *d_trees = nullptr, *d_dense = nullptr;
    int *d_flags = nullptr;
    CSX_TRY(chol_symbolic_device(A, parent, cp, pinv, &L->p, &L->i, &d_rp, &d_rc, &d_rpos, nullptr));
    lap("pattern (device)");
    Forest F;
    partition_forest(n, parent, F);
    // small trees that are dense blocks on consecutive columns take the LDS block kernel
    std::vector<Tree> dense_trees, other_trees;
    for (const Tree &tr : F.small) {
        bool dense = tr.count <= CD_MAX;
        const int32_t c0 = F.small_cols[(size_t)tr.first];
        for (int32_t a = 0; dense && a < tr.count; a++) {
            const int32_t c = F.small_cols[(size_t)tr.first + a];
            dense = c == c0 + a && cp[c + 1] - cp[c] == tr.count - a;
        }
        (dense ? dense_trees : other_trees).push_back(tr);
    }
    if (!ctx().opt.chol_dense_trees) {
        other_trees = F.small;
        dense_trees.clear();
    }
    // ---- fundamental supernodes of the big trees: w >= 8 consecutive columns, each the ONLY child of the next, column
    // counts falling by one (the separators of a nested-dissection ordering).  They leave the level lists: when the
    // walk below reaches the level of a supernode's first column, every update from outside it is available (all of
    // them come from below that first column), so its columns take them in one launch and the trapezoid is then
    // factored densely in place (chol_supernodes).
    const int64_t big_cols_all = (int64_t)F.level_cols.size();   // columns of trees too big for the tree kernels
    std::vector<int32_t> col_level_h;                      // level of every column of a big tree, -1 elsewhere
    std::vector<SnDesc> sns;                               // grouped by the level of their first column
    std::vector<int32_t> sn_cols, sn_group_ptr{0}, sn_group_first{0}, sn_group_maxw, sn_group_maxrows;
    std::vector<int32_t> sn_group_at;                      // level -> group index, -1 none
    if (!F.level_cols.empty()) {
        const int32_t nlev0 = (int32_t)F.level_ptr.size() - 1;
        col_level_h.assign((size_t)n, -1);
        for (int32_t lv = 0; lv < nlev0; lv++)
            for (int32_t q = F.level_ptr[(size_t)lv]; q < F.level_ptr[(size_t)lv + 1]; q++) col_level_h[(size_t)F.level_cols[(size_t)q]] = lv;
        sn_group_at.assign((size_t)nlev0 + 1, -1);
        if (ctx().opt.chol_supernodes) {
            std::vector<int32_t> nchild((size_t)n, 0);
            for (int32_t j = 0; j < n; j++)
                if (parent[j] >= 0) nchild[(size_t)parent[j]]++;
            std::vector<std::pair<int32_t, SnDesc>> found;     // (start level, supernode)
            std::vector<char> member((size_t)n, 0);
            for (int32_t j = 0; j < n;) {
                if (col_level_h[(size_t)j] < 0) {
                    j++;
                    continue;
                }
                const int32_t a = j;
                while (j + 1 < n && parent[j] == j + 1 && nchild[(size_t)j + 1] == 1 &&
                       cp[j + 2] - cp[j + 1] == cp[j + 1] - cp[j] - 1)
                    j++;
                const int32_t w = j - a + 1;
                if (w >= SN_MIN_WIDTH) {
                    found.push_back({col_level_h[(size_t)a], SnDesc{a, w, cp[a + 1] - cp[a] - w}});
                    for (int32_t c = a; c <= j; c++) member[(size_t)c] = 1;
                }
                j++;
            }
            if (!found.empty()) {
                std::stable_sort(found.begin(), found.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
                for (size_t f = 0; f < found.size(); f++) {
                    const int32_t lv = found[f].first;
                    if (f == 0 || lv != found[f - 1].first) {
                        if (f) {
                            sn_group_ptr.push_back((int32_t)sn_cols.size());
                            sn_group_first.push_back((int32_t)sns.size());
                        }
                        sn_group_at[(size_t)lv] = (int32_t)sn_group_maxw.size();
                        sn_group_maxw.push_back(0);
                        sn_group_maxrows.push_back(0);
                    }
                    const SnDesc &d = found[f].second;
                    sns.push_back(d);
                    for (int32_t c = d.a; c < d.a + d.w; c++) sn_cols.push_back(c);
                    sn_group_maxw.back() = std::max(sn_group_maxw.back(), d.w);
                    sn_group_maxrows.back() = std::max(sn_group_maxrows.back(), d.w + d.r);
                }
                sn_group_ptr.push_back((int32_t)sn_cols.size());
                sn_group_first.push_back((int32_t)sns.size());
                // the level lists without the supernodes' columns
                std::vector<int32_t> cols2, ptr2{0};
                for (int32_t lv = 0; lv < nlev0; lv++) {
                    for (int32_t q = F.level_ptr[(size_t)lv]; q < F.level_ptr[(size_t)lv + 1]; q++)
                        if (!member[(size_t)F.level_cols[(size_t)q]]) cols2.push_back(F.level_cols[(size_t)q]);
                    ptr2.push_back((int32_t)cols2.size());
                }
                F.level_cols.swap(cols2);
                F.level_ptr.swap(ptr2);
            }
        }
    }
    lap("partition_forest");
    std::vector<int32_t> hpinv;
    if (pinv) hpinv.assign(pinv, pinv + n);
    int st = dalloc(&L->x, (size_t)L->nnz);
    if (st == CSX_OK && pinv) st = upload(&d_pinv, hpinv);
    if (st == CSX_OK) st = dalloc(&d_win, (size_t)L->nnz);
    if (st == CSX_OK) st = dalloc(&d_flags, 3);   // [0] foreign symbolic data, [1] first non-positive pivot, [2] band width
    if (st == CSX_OK) st = upload(&d_trees, other_trees);
    if (st == CSX_OK) st = upload(&d_dense, dense_trees);
    if (st == CSX_OK) st = upload(&d_small_cols, F.small_cols);
    if (st == CSX_OK) st = upload(&d_level_cols, F.level_cols);
    if (st == CSX_OK) st = upload(&d_level_ptr, F.level_ptr);
    int32_t *d_col_level = nullptr, *d_split = nullptr;
    if (st == CSX_OK && !col_level_h.empty()) {
        st = dalloc(&d_split, (size_t)n);
        if (st == CSX_OK) (void)hipMemsetAsync(d_split, 0x7f, (size_t)n * sizeof(int32_t), s);   // "no inside update seen"
    }
    SnDesc *d_sns = nullptr;
    int32_t *d_sn_cols = nullptr, *d_sn_ptr = nullptr;
    if (st == CSX_OK && !col_level_h.empty()) st = upload(&d_col_level, col_level_h);
    if (st == CSX_OK && !sns.empty()) {
        st = upload(&d_sns, sns);
        if (st == CSX_OK) st = upload(&d_sn_cols, sn_cols);
        if (st == CSX_OK) st = upload(&d_sn_ptr, sn_group_ptr);
    }
    int hflags[2] = {0, 0x7fffffff};
    if (st == CSX_OK) {
        (void)hipEventRecord(ev_a, s);
        (void)hipMemsetAsync(d_win, 0xff, (size_t)L->nnz * sizeof(int32_t), s);
        (void)hipMemcpyAsync(d_flags, hflags, sizeof hflags, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_chol_winner, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, A->p, A->i, d_pinv,
                           L->p, L->i, d_win, d_flags);
        hipLaunchKernelGGL(k_chol_init, dim3((unsigned)(((int64_t)L->nnz + 255) / 256)), dim3(256), 0, s,
                           (int64_t)L->nnz, d_win, A->x, L->x);
        // a chain-like big tree whose factor is a narrow band: the register-window kernel does the whole matrix
        bool banded = false;
        const int32_t nlev_all = (int32_t)F.level_ptr.size() - 1;
        const int64_t big_cols = big_cols_all;
        if (ctx().opt.chol_band && big_cols * 2 > n && (int64_t)nlev_all * 4 > big_cols) {
            int hb = 0;
            (void)hipMemsetAsync(d_flags + 2, 0, sizeof(int), s);
            hipLaunchKernelGGL(k_band_width, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, n, L->p, L->i,
                               d_flags + 2);
            (void)hipMemcpyAsync(&hb, d_flags + 2, sizeof(int), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            const int need = hb + 1;
            // wide bands (and, by option, every band): blocked factorisation in a dense band array, if that array fits
            const int wb = ctx().opt.chol_wband;
            // ... and only when the band is mostly FULL (bcsstk16: 89 %, a grid in natural order: 100 %): the dense band
            // array does n x band^2 work whatever the factor holds (an arrow matrix has band n and a sparse factor)
            const bool full_band = (double)L->nnz >= 0.5 * (double)n * ((double)hb + 1.0);
            if ((wb == 2 || (wb == 1 && need > 80)) && full_band) {   // narrower: the register window costs about the same per column
                size_t free_b = 0, total_b = 0, idle_b = 0, live_b = 0;
                pool_stats(&idle_b, &live_b);
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                    chol_wide_band_bytes(n, hb) < (free_b + idle_b) / 2) {
                    st = chol_wide_band(n, hb, L->p, L->i, L->x, d_flags + 1, ctx().opt.chol_wband_nb);
                    banded = true;
                }
            }
#define CSX_BAND(BWV)                                                                                                \
    hipLaunchKernelGGL((k_chol_band<BWV, 1024>), dim3(1), dim3(1024), 0, s, n, L->p, L->i, L->x, d_rp, d_rc, d_rpos, \
                       d_flags + 1)
            if (banded) {}
            else if (need <= 48) { CSX_BAND(48); banded = true; }
            else if (need <= 80) { CSX_BAND(80); banded = true; }
            else if (need <= 112) { CSX_BAND(112); banded = true; }
            else if (need <= 144) { CSX_BAND(144); banded = true; }
            else if (need <= 176) { CSX_BAND(176); banded = true; }
#undef CSX_BAND
        }
        const int32_t nd = banded ? 0 : (int32_t)dense_trees.size();
        if (nd > 0)
            hipLaunchKernelGGL(k_chol_dense_trees, dim3((unsigned)((nd + CH_WAVES - 1) / CH_WAVES)), dim3(64 * CH_WAVES), 0,
                               s, d_dense, nd, d_small_cols, L->p, L->x, d_flags + 1);
        const int32_t nt = banded ? 0 : (int32_t)other_trees.size();
        if (nt > 0)
            hipLaunchKernelGGL(k_chol_trees, dim3((unsigned)((nt + CH_WAVES - 1) / CH_WAVES)), dim3(64 * CH_WAVES), 0, s,
                               d_trees, nt, d_small_cols, L->p, L->i, L->x, d_rp, d_rc, d_rpos, d_flags + 1);
        const int32_t nlev = banded ? 0 : (int32_t)F.level_ptr.size() - 1;
        int32_t l = 0;
        const size_t cc_lds = (size_t)CC_ACC * 12 + (n <= CC_MAP ? (size_t)n * 4 : 0) + 64;
        if (nlev > 0)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_coop), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 256);
        while (l < nlev) {
            if (!banded && !sn_group_at.empty() && sn_group_at[(size_t)l] >= 0) {
                // supernodes whose first column sits at this level: outside updates for all their columns at once
                // (k_chol_coop, first phase), then the dense trapezoids in place
                const int32_t g = sn_group_at[(size_t)l];
                sn_group_at[(size_t)l] = -1;
                hipLaunchKernelGGL(k_chol_coop, dim3((unsigned)(sn_group_ptr[(size_t)g + 1] - sn_group_ptr[(size_t)g])),
                                   dim3(64 * CC_WAVES), cc_lds, s, d_sn_cols, d_sn_ptr, g, g + 1, L->p, L->i, L->x, d_rp, d_rc,
                                   d_rpos, n, d_flags + 1, d_col_level, 1, l, d_split);
                if (chol_supernodes(d_sns + sn_group_first[(size_t)g], sn_group_first[(size_t)g + 1] - sn_group_first[(size_t)g],
                                    sn_group_maxw[(size_t)g], sn_group_maxrows[(size_t)g], L->p, L->x, d_flags + 1) != CSX_OK)
                    break;
            }
            const int32_t cnt = F.level_ptr[(size_t)l + 1] - F.level_ptr[(size_t)l];
            if (cnt == 0) {                          // all of this level's columns belong to supernodes
                l++;
                continue;
            }
            if (cnt > CH_NARROW) {
                hipLaunchKernelGGL(k_chol_level, dim3((unsigned)((cnt + CH_WAVES - 1) / CH_WAVES)), dim3(64 * CH_WAVES), 0,
                                   s, d_level_cols + F.level_ptr[(size_t)l], cnt, L->p, L->i, L->x, d_rp, d_rc, d_rpos,
                                   d_flags + 1);
                l++;
                continue;
            }
            // a narrow level with more than one column: its columns go to as many workgroups in one launch; a run of
            // single-column levels (a chain) is walked by one workgroup without coming back to the host
            auto width = [&](int32_t lv) { return F.level_ptr[(size_t)lv + 1] - F.level_ptr[(size_t)lv]; };
            int32_t e = l + 1;                       // the run of narrow levels, at most CC_RUN_MAX of them at a time:
            while (e < nlev && e - l < CC_RUN_MAX && width(e) <= CH_NARROW && sn_group_at[(size_t)e] < 0) e++;   // the shorter, the less is left inside
            const bool two_phase = e - l >= CC_RUN_MIN && F.level_ptr[(size_t)e] > F.level_ptr[(size_t)l];
            if (two_phase)                           // updates from below the run, for all of its columns at once
                hipLaunchKernelGGL(k_chol_coop, dim3((unsigned)(F.level_ptr[(size_t)e] - F.level_ptr[(size_t)l])),
                                   dim3(64 * CC_WAVES), cc_lds, s, d_level_cols, d_level_ptr, l, e, L->p, L->i, L->x, d_rp, d_rc,
                                   d_rpos, n, d_flags + 1, d_col_level, 1, l, d_split);
            for (int32_t a = l; a < e;) {
                // a level with several columns: a workgroup each in one launch; single-column levels in a row (a chain):
                // one workgroup walks them without coming back to the host
                if (width(a) == 0) {                 // emptied by the supernodes
                    a++;
                    continue;
                }
                int32_t b = a + 1;
                if (width(a) == 1)
                    while (b < e && width(b) == 1) b++;
                hipLaunchKernelGGL(k_chol_coop, dim3((unsigned)width(a)), dim3(64 * CC_WAVES), cc_lds, s, d_level_cols, d_level_ptr,
                                   a, b, L->p, L->i, L->x, d_rp, d_rc, d_rpos, n, d_flags + 1, d_col_level, two_phase ? 2 : 0, l,
                                   d_split);
                a = b;
            }
            l = e;
        }
        (void)hipEventRecord(ev_b, s);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(hflags, d_flags, sizeof hflags, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            set_error("cs_chol: %s", hipGetErrorString(hipGetLastError()));
            st = CSX_ERUNTIME;
        } else {
            numeric_ms();
            g_chol_path = 0;
        }
    }
    lap("numeric (device)");
    dfree(d_rp);
    dfree(d_rc);
    dfree(d_rpos);
    dfree(d_pinv);
    dfree(d_win);
    dfree(d_flags);
    dfree(d_trees);
    dfree(d_dense);
    dfree(d_small_cols);
    dfree(d_level_cols);
    dfree(d_level_ptr);
    dfree(d_col_level);
    dfree(d_split);
    dfree(d_sns);
    dfree(d_sn_cols);
    dfree(d_sn_ptr);
    if (st != CSX_OK) return st;
    if (hflags[0]) return CSX_EINVAL;                 // S.cp / S.parent do not belong to A
    if (hflags[1] != 0x7fffffff) return CSX_ENOTSPD;  // some pivot d <= 0
    return CSX_OK;
}

// ---- solve phase ----------------------------------------------------------------------------------
struct CholPlan {
    hipGraphExec_t g_exec = nullptr;   // "tri.graph": the supernodal solve's launches, captured for the block g_X / g_nrhs
    double *g_X = nullptr;
    int32_t g_nrhs = 0;
    int64_t g_gen = -1;                // the work-space generation of the supernodal plan the capture saw
    int g_opt = -1;                    // "tri.supernodes" at capture time: sn_solve picks its kernels by it
    double *last_X = nullptr;          // the block of the solve before this one, and how many consecutive solves it has been the
    int32_t last_nrhs = 0;             // block of: "tri.graph" = 2 captures on the THIRD consecutive solve of one block -- a capture
    int32_t same_block_runs = 0;       // costs several solves' worth of host time, a block seen once or twice never repays it
    int32_t g_captures = 0;            // captures made so far, host ms of the last one (csx_cholsol_graph_info)
    double g_capture_ms = 0.0;
    int32_t n = 0;
    const Csc *L = nullptr;  // not owned; must outlive the plan
    TriPlan *fwd = nullptr, *bwd = nullptr;
    int32_t *perm = nullptr;    // device: x[j] = b[perm[j]]  (perm = inverse of pinv), nullptr = identity
    double *scratch = nullptr;  // n * scratch_rhs doubles for the permuted block (generic path)
    int64_t scratch_len = 0;
    // forest-of-small-trees fast path
    bool local = false;
    int32_t ntrees = 0, max_nodes = 0;
    Tree *trees = nullptr;
    Tree *trees_by_size = nullptr;   // the same list biggest first: what k_cholsol_local launches by (trees_biggest_first), made at its first launch
    int32_t *tree_nodes = nullptr, *local_id = nullptr;
    // per-tree solve programs, indexed by position k in tree_nodes (row a of tree t: k = first + a):
    // forward terms [f_ptr[k], f_ptr[k+1]) and backward terms [b_ptr[k], b_ptr[k+1]) as
    // (local row id, value) in the reference's update order, plus the diagonal of that row
    int32_t *f_ptr = nullptr, *f_idx = nullptr, *b_ptr = nullptr, *b_idx = nullptr;
    double *f_val = nullptr, *b_val = nullptr, *diagk = nullptr, *diagb = nullptr;
    int32_t *rev_pos = nullptr;
    int dense_bs = 0;  // > 0: every tree is a dense lower-triangular block of this size on contiguous rows
    double *dense_b = nullptr;  // dense only: backward program with every row reversed (sweep-position order)
    // dense, block size 16/32/64, the matrix cores' operands: the inverses W_ii of the diagonal tiles as A fragments (NB tiles of 256
    // doubles per block; the backward sweep reads them transposed) -- and the off-diagonal tiles, which are L's own numbers and are
    // read where they lie: in L.x (a plan on consecutive columns of L: `clique`), or in `lcopy`, the blocks' packed columns copied
    // out of the programs (a plan the general analysis made: it never assumed an order of the rows inside a column of L)
    double *frag_f = nullptr;
    double *lcopy = nullptr;
    // exact (default): every right-hand side is solved in the reference's operation order -- substitution
    // kernels, level walker in source order: bit-identical to cs_lsolve + cs_ltsolve.  !exact
    // (csx_cholsol_set_order(plan, 0)): results equal to rounding; dense 16/32/64 blocks go to the matrix cores
    // (explicit block inverses, built then), the chain walker may take out-of-block terms first.
    bool relaxed = false;
    // forest of equal dense blocks recognised from L itself (cholsol_plan_clique): no triangular-solve plans, the dense
    // programs cut straight out of L.x; f_idx / b_idx / b_val (the fused per-tree kernel's) are made when first needed
    bool clique = false, clique_zero_pivot = false;
    // forests of small trees that are NOT equal dense blocks (cliques of unequal sizes, small sparse trees), rounding-equal order:
    // the trees made dense and bucketed by size class, solved on the matrix cores (csx_trimfma.hip); null: not built / refused
    RaggedMfma *rag = nullptr;
    bool rag_tried = false;
    // csx_cholsol_factor on such a forest in the rounding-equal order: the plan holds the block list and `rag` (built straight from
    // L's columns) and nothing else; a solve the matrix cores do not take (the exact order, the guard's refusal) goes to `full`,
    // the general plan of the same factor, made the first time it is needed
    bool lite = false;
    bool lite_cliques = false;   // ... and every block is a CLIQUE (dense): the exact order can take the padded size classes below
    CholPlan *full = nullptr;
    // exact order on a forest of cliques of UNEQUAL sizes (round 5): the blocks bucketed by size class 8 / 16 / 32 / 64, every block
    // padded at its end with the identity up to its class, the register-resident exact kernel (k_cholsol_dense_exact_dpp) run
    // per class on programs cut out of L.x -- the padding changes no bit (see the kernel)
    struct ExactClass {
        int32_t count = 0;
        Tree *trees = nullptr;
        int32_t *nodes = nullptr, *f_ptr = nullptr, *b_ptr = nullptr;
        double *f_val = nullptr, *dense_b = nullptr, *diagk = nullptr, *diagb = nullptr;
    } xc[5];
    bool xc_built = false;
    bool mfma_tried = false;  // fragments were built, or refused by the growth guard
    double mfma_growth = 0.0; // max|inv(L_ii)| max|L| over the forest (the guard's measure)
    // big trees, rounding-equal order: supernodal schedule (csx_snsolve.hip), built the first time the plan is relaxed
    std::vector<int32_t> parent_h;   // elimination tree of a Cholesky-shaped L (empty: not one)
    int32_t col_levels = 0;          // its height in columns
    bool sn_tried = false;
    SnPlan *sn = nullptr;
};

void free_cholplan(CholPlan *P) {
    if (!P) return;
    if (P->g_exec) (void)hipGraphExecDestroy(P->g_exec);
    free_snplan(P->sn);
    ragged_free(P->rag);
    free_cholplan(P->full);
    for (auto &c : P->xc) {
        dfree(c.trees);
        dfree(c.nodes);
        dfree(c.f_ptr);
        dfree(c.b_ptr);
        dfree(c.f_val);
        dfree(c.dense_b);
        dfree(c.diagk);
        dfree(c.diagb);
    }
    free_triplan(P->fwd);
    free_triplan(P->bwd);
    dfree(P->perm);
    dfree(P->scratch);
    dfree(P->trees);
    dfree(P->trees_by_size);
    dfree(P->tree_nodes);
    dfree(P->local_id);
    dfree(P->f_ptr);
    dfree(P->f_idx);
    dfree(P->f_val);
    dfree(P->b_ptr);
    dfree(P->b_idx);
    dfree(P->b_val);
    dfree(P->diagk);
    dfree(P->diagb);
    dfree(P->rev_pos);
    dfree(P->dense_b);
    dfree(P->frag_f);
    dfree(P->lcopy);
    delete P;
}

// G lanes to a column (4 where the columns are short)
template <int G>
__global__ __launch_bounds__(256) void k_parent_of_sorted_L(int32_t n, const int32_t *__restrict__ Lp,
                                                            const int32_t *__restrict__ Li, int32_t *parent,
                                                            int *unsorted) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (j >= n) return;
    const int32_t b = Lp[j], e = Lp[j + 1];
    if (lane == 0) parent[j] = e - b > 1 ? Li[b + 1] : -1;
    for (int32_t p = b + lane; p < e; p += G) {
        const int32_t r = Li[p];
        if ((p == b && r != j) || (p > b && r <= Li[p - 1])) *unsorted = 1;
    }
}

// ---- the plan's partition on the device, for forests whose trees sit on consecutive columns -----------------------------
// k starts a BLOCK when no column before it reaches row k or below: max_{j < k} (last row of column j) < k -- a prefix
// maximum, taken as a reverse running minimum of the negated, reversed array (suffix_min_i32).  A block is closed under
// L's pattern; with as many roots as blocks every block is exactly one tree, and the host's partition (trees by ascending
// root, a tree's columns ascending: partition_forest) is the identity node list cut at the block starts.
__global__ __launch_bounds__(256) void k_plan_reach(int32_t n, const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                                    int32_t *__restrict__ neg_rev) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t b = Lp[j], e = Lp[j + 1];
    neg_rev[n - 1 - j] = e > b ? -Li[e - 1] : -(int32_t)j;   // rows ascending (k_parent_of_sorted_L checks): the last is the largest
}

// stats[1] += roots
__global__ __launch_bounds__(256) void k_plan_starts(int32_t n, const int32_t *__restrict__ neg_rev_min,
                                                     const int32_t *__restrict__ parent, int32_t *__restrict__ is_start, int *stats) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int root = 0;
    if (k < n) {
        is_start[k] = (k == 0 || -neg_rev_min[n - k] < (int32_t)k) ? 1 : 0;
        root = parent[k] < 0 ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) root += __shfl_xor(root, o);
    __shared__ int s_roots[4];
    if ((threadIdx.x & 63) == 0) s_roots[threadIdx.x >> 6] = root;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&stats[1], s_roots[0] + s_roots[1] + s_roots[2] + s_roots[3]);
}

__global__ __launch_bounds__(256) void k_plan_block_first(int32_t n, const int32_t *__restrict__ is_start,
                                                          const int32_t *__restrict__ block_id, int32_t *__restrict__ start) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (is_start[k]) start[block_id[k]] = (int32_t)k;
    if (k == n - 1) start[block_id[k] + is_start[k]] = n;
}

// stats[2] = widest block
__global__ __launch_bounds__(256) void k_plan_trees(int32_t nblocks, const int32_t *__restrict__ start, Tree *__restrict__ trees,
                                                    int *stats) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int w = 0;
    if (b < nblocks) {
        w = start[b + 1] - start[b];
        trees[b] = Tree{start[b], w};
    }
    for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o));
    if ((threadIdx.x & 63) == 0 && w > *(volatile int *)&stats[2]) atomicMax(&stats[2], w);
}

__global__ __launch_bounds__(256) void k_plan_nodes(int32_t n, const int32_t *__restrict__ is_start,
                                                    const int32_t *__restrict__ block_id, const int32_t *__restrict__ start,
                                                    int32_t *__restrict__ nodes, int32_t *__restrict__ local_id,
                                                    int32_t *__restrict__ rev_pos) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t b = block_id[k] + is_start[k] - 1;
    const int32_t first = start[b], next = start[b + 1];
    nodes[k] = (int32_t)k;
    local_id[k] = (int32_t)k - first;
    rev_pos[k] = first + next - 1 - (int32_t)k;
}

// ---- packing the per-tree programs (plan time) ----
__global__ __launch_bounds__(256) void k_pack_len(int32_t n, const int32_t *__restrict__ nodes,
                                                  const int32_t *__restrict__ rev_pos,
                                                  const int32_t *__restrict__ Gp, const int32_t *__restrict__ Lp,
                                                  int32_t *flen, int32_t *blen) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t j = nodes[k];
    flen[k] = Gp[j + 1] - Gp[j];
    blen[rev_pos[k]] = Lp[j + 1] - Lp[j] - 1;  // the backward sweep visits a tree's rows in reverse
}

template <int G>
__global__ __launch_bounds__(256) void k_pack_fill(int32_t n, const int32_t *__restrict__ nodes,
                                                   const int32_t *__restrict__ rev_pos,
                                                   const int32_t *__restrict__ local_id,
                                                   const int32_t *__restrict__ Gp, const int32_t *__restrict__ Gi,
                                                   const double *__restrict__ Gx, const int32_t *__restrict__ Lp,
                                                   const int32_t *__restrict__ Li, const double *__restrict__ Lx,
                                                   const int32_t *__restrict__ f_ptr, int32_t *f_idx, double *f_val,
                                                   const int32_t *__restrict__ b_ptr, int32_t *b_idx, double *b_val,
                                                   double *diagf, double *diagb) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (k >= n) return;
    const int32_t j = nodes[k], kb = rev_pos[k];
    const int32_t gb = Gp[j], fl = Gp[j + 1] - gb, fo = f_ptr[k];
    for (int32_t q = lane; q < fl; q += G) {
        f_idx[fo + q] = local_id[Gi[gb + q]] * 64;  // premultiplied: X tile is [node][64 lanes]
        f_val[fo + q] = Gx[gb + q];
    }
    const int32_t lb = Lp[j] + 1, bl = Lp[j + 1] - lb, bo = b_ptr[kb];
    for (int32_t q = lane; q < bl; q += G) {
        b_idx[bo + q] = local_id[Li[lb + q]] * 64;
        b_val[bo + q] = Lx[lb + q];
    }
    if (lane == 0) {
        diagf[k] = Lx[Lp[j]];
        diagb[kb] = Lx[Lp[j]];
    }
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(64 * CH_WAVES) void k_cholsol_local(
    const Tree *__restrict__ trees, int32_t ntrees, const int32_t *__restrict__ nodes,
    const int32_t *__restrict__ perm, const int32_t *__restrict__ f_ptr, const int32_t *__restrict__ f_idx,
    const double *__restrict__ f_val, const int32_t *__restrict__ b_ptr, const int32_t *__restrict__ b_idx,
    const double *__restrict__ b_val, const double *__restrict__ diagf, const double *__restrict__ diagb, double *B,
    int32_t nrhs, int32_t chunks, int32_t max_nodes, int32_t waves_per_wg) {
    extern __shared__ __attribute__((aligned(16))) double xt[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= waves_per_wg) return;
    const int64_t task = (int64_t)blockIdx.x * waves_per_wg + w;
    if (task >= (int64_t)ntrees * chunks) return;
    const int32_t t = (int32_t)(task / chunks), h = (int32_t)(task % chunks);
    const Tree tr = trees[t];
    const int32_t rhs = h * 64 + lane;
    const bool live = rhs < nrhs;
    double *X = xt + (size_t)w * max_nodes * 64;
    // rows of B owned by this tree: ids fetched coalesced (64 at a time), handed out by v_readlane;
    // sixteen row loads are kept in flight
    for (int32_t c0 = 0; c0 < tr.count; c0 += 64) {
        const int32_t crow = min(64, tr.count - c0);
        int32_t jrow = 0;
        if (lane < crow) {
            jrow = nodes[tr.first + c0 + lane];
            if (perm) jrow = perm[jrow];
        }
        for (int32_t r0 = 0; r0 < crow; r0 += 16) {
            double tmp[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int32_t row = __builtin_amdgcn_readlane(jrow, min(r0 + u, crow - 1));
                tmp[u] = B[(int64_t)row * nrhs + (live ? rhs : nrhs - 1)];  // clamped: safe unpredicated
            }
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (r0 + u < crow) X[(c0 + r0 + u) * 64 + lane] = tmp[u];
        }
    }
    // forward: L y = x, rows ascending, terms in ascending column order (the reference's push order)
    sweep<true>(tr, f_ptr, f_idx, f_val, diagf, X, lane);
    // backward: L' z = y, rows descending, the column of L in storage order
    sweep<false>(tr, b_ptr, b_idx, b_val, diagb, X, lane);
    for (int32_t c0 = 0; c0 < tr.count; c0 += 64) {
        const int32_t crow = min(64, tr.count - c0);
        int32_t jrow = 0;
        if (lane < crow) {
            jrow = nodes[tr.first + c0 + lane];
            if (perm) jrow = perm[jrow];
        }
        for (int32_t r = 0; r < crow; r++) {
            const int32_t row = __builtin_amdgcn_readlane(jrow, r);
            if (live) B[(int64_t)row * nrhs + rhs] = X[(c0 + r) * 64 + lane];
        }
    }
}
#pragma clang fp contract(fast)

// ---- dense-block specialisation -------------------------------------------------------------------
// Every tree is a dense BS x BS lower-triangular block on contiguous rows (block-diagonal SPD
// matrices, batches of small dense systems).  One wave = one block x 64 right-hand sides; the
// unknowns of a lane's right-hand side live in BS registers, the packed block (row-major in
// sweep order) and the reciprocal diagonal sit in LDS and are broadcast with immediate offsets.
// The same unrolled sweep serves forward and backward substitution: the backward system in sweep
// order (rows descending) is again "unit-ordered lower triangular" once the unknowns are
// reversed in registers.  Uses FMA and a reciprocal diagonal, so x agrees with the
// reference-order kernels to rounding (~1e-16 relative), not bit for bit.
__global__ __launch_bounds__(256) void k_dense_reverse_rows(int32_t n, const int32_t *__restrict__ b_ptr,
                                                           const double *__restrict__ b_val, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= n) return;
    const int32_t b = b_ptr[k], len = b_ptr[k + 1] - b;
    for (int32_t q = lane; q < len; q += 64) out[b + q] = b_val[b + len - 1 - q];
}

typedef __attribute__((address_space(1))) const void *csx_gptr;
typedef __attribute__((address_space(3))) void *csx_lptr;

template <int BS>
__global__ __launch_bounds__(256) void k_cholsol_dense(const Tree *__restrict__ trees, int32_t ntrees,
                                                       const int32_t *__restrict__ nodes,
                                                       const int32_t *__restrict__ perm,
                                                       const int32_t *__restrict__ f_ptr, const double *__restrict__ f_val,
                                                       const int32_t *__restrict__ b_ptr, const double *__restrict__ b_val,
                                                       const double *__restrict__ diagf, const double *__restrict__ diagb,
                                                       double *B, int32_t nrhs, int32_t chunks) {
    constexpr int NT = BS * (BS - 1) / 2;  // strictly-lower entries
    constexpr int MSZ = ((NT + 127) / 128 * 128 > NT + BS) ? (NT + 127) / 128 * 128 : NT + BS;
    __shared__ __attribute__((aligned(16))) double s_m[4][MSZ];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t task = (int64_t)blockIdx.x * 4 + w;
    if (task >= (int64_t)ntrees * chunks) return;
    const int32_t t = (int32_t)(task / chunks), h = (int32_t)(task % chunks);
    const int32_t first = trees[t].first;
    const int32_t rhs = h * 64 + lane;
    const bool live = rhs < nrhs;
    double *M = s_m[w], *RD = s_m[w] + NT;  // RD overlaps the DMA overrun and is written after it
    // rows of B owned by this block (through the fill-reducing permutation, if any)
    int32_t jrow = 0;
    if (lane < BS) {
        jrow = nodes[first + lane];
        if (perm) jrow = perm[jrow];
    }
    // lanes past the last right-hand side load a valid element (their column index is clamped)
    // and never store: the compiler is free to issue these loads unpredicated
    const int32_t rhs_ld = live ? rhs : nrhs - 1;
    double x[BS];
#pragma unroll
    for (int a = 0; a < BS; a++) {
        const int32_t row = __builtin_amdgcn_readlane(jrow, a);
        x[a] = B[(int64_t)row * nrhs + rhs_ld];
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const int32_t *ptr = pass ? b_ptr : f_ptr;
        const double *val = pass ? b_val : f_val;  // b_val here is the row-reversed dense copy
        const double *dg = pass ? diagb : diagf;
        const int32_t base = ptr[first];
        // stage the packed block (row sp of the sweep holds M[sp][t], t = 0..sp-1) with
        // asynchronous global->LDS DMA: 1 KiB per instruction, no registers, all in flight at once.
        // The last piece runs past the block into the RD area, which is written afterwards.
        const double rd = lane < BS ? 1.0 / dg[first + lane] : 0.0;
#pragma unroll
        for (int k = 0; k < (NT + 127) / 128; k++)
            __builtin_amdgcn_global_load_lds((csx_gptr)(val + base + k * 128 + 2 * lane), (csx_lptr)(M + k * 128), 16, 0,
                                             0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < BS) RD[lane] = rd;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sp = 0; sp < BS; sp++) {
            double acc = x[sp];
#pragma unroll
            for (int tt = 0; tt < sp; tt++) acc = fma(-M[sp * (sp - 1) / 2 + tt], x[tt], acc);
            x[sp] = acc * RD[sp];
        }
        __builtin_amdgcn_wave_barrier();
        // reverse the unknowns: after the forward pass this puts them in backward sweep order,
        // after the backward pass it restores row order
#pragma unroll
        for (int a = 0; a < BS / 2; a++) {
            const double tmp = x[a];
            x[a] = x[BS - 1 - a];
            x[BS - 1 - a] = tmp;
        }
    }
    // v_readlane must run with every lane active (it reads lanes that may be past the last
    // right-hand side), so only the store itself is predicated
#pragma unroll
    for (int a = 0; a < BS; a++) {
        const int32_t row = __builtin_amdgcn_readlane(jrow, a);
        if (live) B[(int64_t)row * nrhs + rhs] = x[a];
    }
}

// ---- dense blocks, the reference's order (the DEFAULT solve of a forest of dense blocks) ------------------
// Same organisation as k_cholsol_dense (one wave = one block x 64 right-hand sides, a lane's BS unknowns in
// registers, the packed block staged in LDS by global->LDS DMA and broadcast), but every operation is the
// reference's: multiply and subtract rounded separately, a true division by the diagonal, and the terms of an
// unknown taken in the reference's order -- forward: ascending column (the sweep order); backward: ascending row,
// which in the reversed sweep numbering is DEscending t, so the two passes are two bodies.  x is bit-identical
// to cs_lsolve + cs_ltsolve for every right-hand side; the fused per-tree kernel it replaces on dense forests took
// 24.8 ms per 128 right-hand sides on the 5M-row G-spd.
#pragma clang fp contract(off)
// R right-hand sides per lane: the R subtraction chains of a row are independent, so they fill each other's latency, and one
// broadcast of an L value serves R products (the broadcast -- LDS return path -- and the two separately rounded fp64
// operations per term are what bound this kernel: 4 + 4 cycles of issue per term and right-hand side).
template <int BS, bool BACKWARD, int R>
__device__ __forceinline__ void dense_exact_pass_rows(double (&x)[R][BS], const double *M, const double *D) {
#pragma unroll
    for (int sp = 0; sp < BS; sp++) {
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = x[r][sp];
        if (!BACKWARD) {
#pragma unroll
            for (int tt = 0; tt < sp; tt++) {
                const double mv = M[sp * (sp - 1) / 2 + tt];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double t = mv * x[r][tt];
                    acc[r] = acc[r] - t;
                }
            }
        } else {
#pragma unroll
            for (int tt = sp - 1; tt >= 0; tt--) {
                const double mv = M[sp * (sp - 1) / 2 + tt];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double t = mv * x[r][tt];
                    acc[r] = acc[r] - t;
                }
            }
        }
        const double dv = D[sp];
#pragma unroll
        for (int r = 0; r < R; r++) x[r][sp] = acc[r] / dv;
        // without a fence the optimiser hoists every row's LDS reads to the top of the pass (they depend on nothing):
        // 512 VGPRs and spills at BS = 32, 292 VGPRs at BS = 64
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}


// The same pass with the L values read through a ring (see below); x indexed in reverse for the backward sweep.
// R right-hand sides per lane: the R subtraction chains of a row are independent, so they fill each other's latency, and one
// broadcast of an L value serves R products (the broadcast -- LDS return path -- and the two separately rounded fp64
// operations per term are what bound this kernel: 4 + 4 cycles of issue per term and right-hand side).
//
// The L values come through a RING of K registers filled K terms ahead, in program order and fenced term by term
// (sched_barrier): left to itself the compiler either hoists a whole row's LDS reads (hundreds of VGPRs, spills) or
// reads each pair of values just before its use -- ds_read_b128, s_waitcnt lgkmcnt(0), two products -- and then every
// second term pays the LDS latency (the round-2 kernel: 55 cycles per term and wave where 8 are issued).
template <int BS, bool BACKWARD, int R>
__device__ __forceinline__ void dense_exact_pass(double (&x)[R][BS], const double *M, const double *D) {
    constexpr int K = 12;                               // at most 15 LDS reads can be outstanding (lgkmcnt)
    // The backward sweep runs in sweep positions (position s = row BS - 1 - s): x is indexed through XI instead of being
    // reversed in place -- a reversal between the passes makes the compiler keep both copies of x alive (390 registers
    // at BS = 64 instead of 262).
#define CSX_XI(i) (BACKWARD ? BS - 1 - (i) : (i))
    double ring[K];
    int lsp = 1, le = 0;                                // the next term to request: row lsp, position le in its order
#pragma unroll
    for (int u = 0; u < K; u++) {
        ring[u] = 0.0;
        if (lsp < BS) {
            ring[u] = M[lsp * (lsp - 1) / 2 + (BACKWARD ? lsp - 1 - le : le)];
            if (++le == lsp) {
                lsp++;
                le = 0;
            }
        }
    }
    double dnext = D[0];
    int f = 0;
#pragma unroll
    for (int sp = 0; sp < BS; sp++) {
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = x[r][CSX_XI(sp)];
        const double dv = dnext;
        if (sp + 1 < BS) dnext = D[sp + 1];
#pragma unroll
        for (int e = 0; e < sp; e++) {
            const int tt = BACKWARD ? sp - 1 - e : e;   // the reference's order: ascending columns forward, descending backward
            const double mv = ring[f % K];
            if (lsp < BS) {
                ring[f % K] = M[lsp * (lsp - 1) / 2 + (BACKWARD ? lsp - 1 - le : le)];
                if (++le == lsp) {
                    lsp++;
                    le = 0;
                }
            }
            f++;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double t = mv * x[r][CSX_XI(tt)];
                acc[r] = acc[r] - t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < R; r++) x[r][CSX_XI(sp)] = acc[r] / dv;
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CSX_XI
}

// The same pass with the L values delivered by DPP instead of by broadcast reads.  What bounds the passes above is the LDS
// return path: a broadcast read hands 64 x 16 bytes to the wave to deliver TWO doubles (8 clk per read on a path the four
// SIMDs of a CU share: 4.2 ms per 128 right-hand sides at blocks of 64, twice the 2.1 ms the fp64 operations need).  The
// 64-bit move of this chip takes one DPP control, row_newbcast:k -- every lane receives the value lane k of its ROW of 16
// lanes holds.  So the 16 lanes of a row read 16 consecutive L values of the matrix row being solved (one ds_read_b64 per
// 16 terms, the four rows reading the same 128 bytes), and term k is
//     v_mov_b64_dpp b, Lv row_newbcast:k ;  v_mul_f64 t, b, x_k ;  v_add_f64 acc, acc, -t
// -- the reference's two separately rounded operations plus one register move; no v_readlane, no scalar load, and a
// broadcast read for one term in four only (see dense_exact_term_dpp).  (v_mul_f64 / v_add_f64 are VOP3 and have no DPP form on this ISA; v_fmac_f64 has one, but it fuses.)  A row's L
// values and diagonal are requested while the previous row is being solved.
template <int K>
__device__ __forceinline__ double dpp_row_bcast(const double v) {
    return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + K, 0xf, 0xf, true));
}

// Where the pass finds L.  PACKED = false: the plan's PROGRAMS -- the strictly lower triangle row-major in sweep order (term E of
// sweep row SP at SP (SP - 1) / 2 + E; the backward pass reads the row-reversed copy with the same indices), the diagonals in an
// array of their own.  PACKED = true (round 5): the block's packed COLUMNS as they lie in L.x -- entry (R, C), R >= C, at
// C BS - C (C - 1) / 2 + R - C -- one copy for both passes and no programs at all: forward, sweep row SP is matrix row SP and term
// E its column E; backward, sweep row SP is COLUMN BS - 1 - SP and term E its row BS - 1 - E.
template <int BS>
constexpr int xl_col(int C) { return C * BS - C * (C - 1) / 2 - C; }                 // entry (R, C) at xl_col(C) + R
template <int BS, bool BACKWARD, bool PACKED>
constexpr int xl_term(int SP, int E) {                                               // wave-uniform index of term E of sweep row SP
    return !PACKED ? SP * (SP - 1) / 2 + E : (!BACKWARD ? xl_col<BS>(E) + SP : xl_col<BS>(BS - 1 - SP) + BS - 1 - E);
}
template <int BS, bool BACKWARD, bool PACKED>
constexpr int xl_diag(int SP) {                                                      // index of sweep row SP's diagonal (PACKED: in the copy)
    return !PACKED ? SP : (!BACKWARD ? xl_col<BS>(SP) + SP : xl_col<BS>(BS - 1 - SP) + BS - 1 - SP);
}
// the 16 lanes of a DPP row read terms 16 G .. 16 G + 15 of sweep row NS: lane k at lb[G] + xl_group(NS, G), lb = the lane's part
// (programs: + k; packed forward: the start of column 16 G + k; packed backward: - k, the rows of one column descending)
template <int BS, bool BACKWARD, bool PACKED>
constexpr int xl_group(int NS, int G) {
    return !PACKED ? NS * (NS - 1) / 2 + 16 * G : (!BACKWARD ? NS : xl_col<BS>(BS - 1 - NS) + BS - 1 - 16 * G);
}

// term E of sweep row SP; the L values of a group of 16 terms die with the group's last term, and the same group of the
// NEXT row is requested there: four values live instead of eight (168 VGPRs is three waves per SIMD at blocks of 64)
template <int BS, bool BACKWARD, int MIX, bool PACKED, int SP, int E>
__device__ __forceinline__ void dense_exact_term_dpp(const double (&x)[BS], const double *M, const int (&lb)[4], const double (&lv)[4],
                                                     double (&nxt)[4], double &acc) {
    constexpr int G = E >> 4, NS = SP + 1;
    // One term in four takes its L value by a broadcast LDS read instead: two VALU instructions instead of three, on the LDS
    // path the DPP form left idle.  Measured at blocks of 64 (ms per 128 right-hand sides): none 4.81, one in four 4.59, two
    // in four 4.93 -- at half the terms the return path (4 clk per broadcast value and CU) is 80 % busy and its latency shows.
    constexpr bool via_lds = (E & 3) < MIX;
    const double lbv = via_lds ? M[xl_term<BS, BACKWARD, PACKED>(SP, E)] : dpp_row_bcast<(E & 15)>(lv[G]);
    const double t = lbv * x[BACKWARD ? BS - 1 - E : E];
    acc = acc - t;
    constexpr bool last_of_group = BACKWARD ? (E & 15) == 0 : ((E & 15) == 15 || E == SP - 1);
    if constexpr (last_of_group && NS < BS) nxt[G] = M[lb[G] + xl_group<BS, BACKWARD, PACKED>(NS, G)];
    // a row's products are independent of each other: left alone the scheduler forms dozens of them ahead of the
    // subtraction chain and spills; four at a time is ahead enough
    if constexpr ((E & 3) == (BACKWARD ? 0 : 3) || last_of_group) __builtin_amdgcn_sched_barrier(0);
}

template <int BS, bool BACKWARD, int MIX, bool PACKED, int SP, int... I>
__device__ __forceinline__ void dense_exact_terms_dpp(const double (&x)[BS], const double *M, const int (&lb)[4], const double (&lv)[4],
                                                      double (&nxt)[4], double &acc, std::integer_sequence<int, I...>) {
    // the reference's order: ascending columns forward, descending (in sweep numbering) backward
    (dense_exact_term_dpp<BS, BACKWARD, MIX, PACKED, SP, (BACKWARD ? SP - 1 - I : I)>(x, M, lb, lv, nxt, acc), ...);
}

// sweep row SP: cur = its L values (lane l: entry 16 g + (l & 15) of the row's terms), dv its diagonal; requests row SP + 1
template <int BS, bool BACKWARD, int MIX, bool PACKED, int SP>
__device__ __forceinline__ void dense_exact_row_dpp(double (&x)[BS], const double *M, const int (&lb)[4], const double *D, const double (&cur)[4],
                                                    double (&nxt)[4], const double dv, double &dnext) {
    constexpr int NS = SP + 1;
    if constexpr (NS < BS) {
        // a group the next row has and this one has not (its first term is this row's unknown)
        if constexpr ((NS + 15) / 16 > (SP + 15) / 16) nxt[(SP + 15) / 16] = M[lb[(SP + 15) / 16] + xl_group<BS, BACKWARD, PACKED>(NS, (SP + 15) / 16)];
        dnext = D[xl_diag<BS, BACKWARD, PACKED>(NS)];
    }
    double acc = x[BACKWARD ? BS - 1 - SP : SP];
    dense_exact_terms_dpp<BS, BACKWARD, MIX, PACKED, SP>(x, M, lb, cur, nxt, acc, std::make_integer_sequence<int, SP>{});
    double xr = acc / dv;
    // the row's arithmetic is pure: nothing but its data dependences holds it in place, and the instruction selector sank
    // whole rows of it behind the moves of later rows (256 VGPRs and spills at blocks of 16).  An empty volatile asm that
    // "rewrites" the row's result ties the arithmetic to this point; the memory clobber does the same for the LDS reads.
    asm volatile("" : "+v"(xr) : : "memory");
    x[BACKWARD ? BS - 1 - SP : SP] = xr;
    __builtin_amdgcn_sched_barrier(0);
}

template <int BS, bool BACKWARD, int MIX, bool PACKED, int... SP>
__device__ __forceinline__ void dense_exact_rows_dpp(double (&x)[BS], const double *M, const int (&lb)[4], const double *D,
                                                     std::integer_sequence<int, SP...>) {
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};   // L values of the even / odd sweep rows
    double da = D[xl_diag<BS, BACKWARD, PACKED>(0)], db = 1.0;           // and their diagonals
    (dense_exact_row_dpp<BS, BACKWARD, MIX, PACKED, SP>(x, M, lb, D, (SP & 1) ? b : a, (SP & 1) ? a : b, (SP & 1) ? db : da, (SP & 1) ? da : db), ...);
}

template <int BS, bool BACKWARD, int MIX, bool PACKED>
__device__ __forceinline__ void dense_exact_pass_dpp(double (&x)[BS], const double *M, const double *D, const int lane) {
    const int k = lane & 15;
    int lb[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int c = 16 * g + k < BS ? 16 * g + k : BS - 1;          // (lanes past the block's columns: any address inside the copy)
        lb[g] = !PACKED ? k : (!BACKWARD ? c * BS - c * (c - 1) / 2 - c : -k);
    }
    dense_exact_rows_dpp<BS, BACKWARD, MIX, PACKED>(x, M, lb, D, std::make_integer_sequence<int, BS>{});
}

// second launch-bound argument = waves per SIMD the register allocation must leave room for: without it the
// scheduler spends 284-512 VGPRs on hoisted loads (one wave per SIMD, spills at BS = 32).  R = 2 (two right-hand sides
// per lane, a task = a block x 128 right-hand sides): one wave per SIMD, 2 x BS unknowns in registers.
template <int BS, int R, bool RING>
__global__ __launch_bounds__(256, (RING || R == 2 ? 1 : 2)) void k_cholsol_dense_exact(const Tree *__restrict__ trees, int32_t ntrees,
                                                             const int32_t *__restrict__ nodes,
                                                             const int32_t *__restrict__ perm,
                                                             const int32_t *__restrict__ f_ptr,
                                                             const double *__restrict__ f_val,
                                                             const int32_t *__restrict__ b_ptr,
                                                             const double *__restrict__ b_val,
                                                             const double *__restrict__ diagf,
                                                             const double *__restrict__ diagb, double *B, int32_t nrhs,
                                                             int32_t chunks) {
    constexpr int NT = BS * (BS - 1) / 2;
    constexpr int MSZ = ((NT + 127) / 128 * 128 > NT + BS) ? (NT + 127) / 128 * 128 : NT + BS;
    __shared__ __attribute__((aligned(16))) double s_m[4][MSZ];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t task = (int64_t)blockIdx.x * 4 + w;
    if (task >= (int64_t)ntrees * chunks) return;          // chunks: groups of 64 R right-hand sides
    const int32_t t = (int32_t)(task / chunks), h = (int32_t)(task % chunks);
    const int32_t first = trees[t].first;
    int32_t rhs[R], rhs_ld[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        rhs[r] = (h * R + r) * 64 + lane;
        live[r] = rhs[r] < nrhs;
        rhs_ld[r] = live[r] ? rhs[r] : nrhs - 1;
    }
    double *M = s_m[w], *DG = s_m[w] + NT;   // DG overlaps the DMA overrun and is written after it
    int32_t jrow = 0;
    if (lane < BS) {
        jrow = nodes[first + lane];
        if (perm) jrow = perm[jrow];
    }
    double x[R][BS];
#pragma unroll
    for (int a = 0; a < BS; a++) {
        const int32_t row = __builtin_amdgcn_readlane(jrow, a);
#pragma unroll
        for (int r = 0; r < R; r++) x[r][a] = B[(int64_t)row * nrhs + rhs_ld[r]];
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int32_t *ptr = pass ? b_ptr : f_ptr;
        const double *val = pass ? b_val : f_val;    // b_val here is the row-reversed dense copy
        const double *dg = pass ? diagb : diagf;
        const int32_t base = ptr[first];
        const double dv = lane < BS ? dg[first + lane] : 1.0;
#pragma unroll
        for (int k = 0; k < (NT + 127) / 128; k++)
            __builtin_amdgcn_global_load_lds((csx_gptr)(val + base + k * 128 + 2 * lane), (csx_lptr)(M + k * 128), 16, 0,
                                             0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < BS) DG[lane] = dv;
        __builtin_amdgcn_wave_barrier();
        if (RING) {
            if (pass == 0) dense_exact_pass<BS, false, R>(x, M, DG);
            else dense_exact_pass<BS, true, R>(x, M, DG);
        } else {
            if (pass == 0) dense_exact_pass_rows<BS, false, R>(x, M, DG);
            else dense_exact_pass_rows<BS, true, R>(x, M, DG);
        }
        __builtin_amdgcn_wave_barrier();
        if (!RING) {
#pragma unroll
            for (int a = 0; a < BS / 2; a++) {       // reverse: sweep order of the other pass / back to row order
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double tmp = x[r][a];
                    x[r][a] = x[r][BS - 1 - a];
                    x[r][BS - 1 - a] = tmp;
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < BS; a++) {
        const int32_t row = __builtin_amdgcn_readlane(jrow, a);
#pragma unroll
        for (int r = 0; r < R; r++)
            if (live[r]) B[(int64_t)row * nrhs + rhs[r]] = x[r][a];
    }
}
// The DPP form as a kernel of its own.  Its 158 VGPRs allow three waves per SIMD at blocks of 64, the LDS copy of the block
// (16.6 KB) does not if every wave stages its own (4 x 16.6 KB per workgroup: two workgroups per CU).  Waves of a
// workgroup that solve the SAME block for different right-hand sides share one copy: SHARE = blocks per workgroup
// (4, 2 or 1), 4 / SHARE waves per block, each staging its share of the DMA; two workgroup barriers per pass.
// PACKED (round 5; a plan on all the columns of L, equal blocks): f_val = L.x itself, block t at t BS (BS + 1) / 2 -- the block's
// packed columns are copied ONCE and serve both passes (see xl_term); no programs, no diagonal arrays: the other pointers are unused.
template <int BS, int SHARE, int MIX, bool PACKED>
__global__ __launch_bounds__(256, 3) void k_cholsol_dense_exact_dpp(const Tree *__restrict__ trees, int32_t ntrees,
                                                                    const int32_t *__restrict__ nodes, const int32_t *__restrict__ perm,
                                                                    const int32_t *__restrict__ f_ptr, const double *__restrict__ f_val,
                                                                    const int32_t *__restrict__ b_ptr, const double *__restrict__ b_val,
                                                                    const double *__restrict__ diagf, const double *__restrict__ diagb,
                                                                    double *B, int32_t nrhs, int32_t chunks) {
    constexpr int NT = BS * (BS - 1) / 2, NENT = BS * (BS + 1) / 2;
    constexpr int MSZ = PACKED ? (NENT + 127) / 128 * 128 : (((NT + 127) / 128 * 128 > NT + BS) ? (NT + 127) / 128 * 128 : NT + BS);
    constexpr int WPT = 4 / SHARE;                          // waves per block
    __shared__ __attribute__((aligned(16))) double s_m[SHARE][MSZ];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = w / WPT, sub = w % WPT;
    const int32_t cgroups = chunks / WPT;                   // the host launches this SHARE only when WPT divides chunks
    const int32_t tg = (int32_t)(blockIdx.x / cgroups), cg = (int32_t)(blockIdx.x % cgroups);
    const int32_t t_raw = tg * SHARE + slot, h = cg * WPT + sub;
    const bool valid = t_raw < ntrees;                      // a workgroup past the last block still meets the barriers
    const int32_t t = valid ? t_raw : ntrees - 1;
    const int32_t first = trees[t].first;
    const int32_t rhs = h * 64 + lane;
    const bool live = valid && rhs < nrhs;
    const int32_t rhs_ld = rhs < nrhs ? rhs : nrhs - 1;
    double *M = s_m[slot], *DG = PACKED ? s_m[slot] : s_m[slot] + NT;   // (programs: DG overlaps the DMA overrun and is written after it)
    // (a node of -1 is PADDING: a block of fewer than BS columns padded at its end with the identity -- cholsol_exact_classes.
    // Its unknown starts as +0.0, stays +0.0 through both sweeps and is never stored; the zero coefficients that link it to the
    // real rows come last in every real row's backward sum and subtract (+0.0) (+0.0) = +0.0: no bit of a real unknown changes --
    // for FINITE data: an infinity among a block's unknowns turns the padding into NaN (0 x inf) and the block's other non-finite
    // values with it, where the reference keeps some of them as infinities; a clique's unknowns are all non-finite then either way.)
    int32_t jrow = -1;
    if (lane < BS) {
        jrow = nodes[first + lane];
        if (perm && jrow >= 0) jrow = perm[jrow];
    }
    double x[BS];
#pragma unroll
    for (int a = 0; a < BS; a++) {
        const int32_t row = __builtin_amdgcn_readlane(jrow, a);
        x[a] = row >= 0 ? __builtin_nontemporal_load(B + (int64_t)row * nrhs + rhs_ld) : 0.0;   // (B goes through once per batch)
    }
    auto run_pass = [&](auto pass_tag) {
        constexpr int pass = decltype(pass_tag)::value;
        if (!PACKED || pass == 0) {
            const int32_t *ptr = pass ? b_ptr : f_ptr;
            const double *val = pass ? b_val : f_val;           // b_val here is the row-reversed dense copy
            const double *dg = pass ? diagb : diagf;
            const int64_t base = PACKED ? (int64_t)t * NENT : (int64_t)ptr[first];
            const double dv = (!PACKED && lane < BS) ? dg[first + lane] : 1.0;
            if (pass) __syncthreads();                          // every wave of the block is done with the forward copy
#pragma unroll
            for (int k = 0; k < (PACKED ? MSZ / 128 : (NT + 127) / 128); k++)
                if (k % WPT == sub) {
                    const int e = k * 128 + 2 * lane;
                    // (programs: the overrun of the last chunk reads the next block's -- the arrays are padded by 128; packed: a lane
                    // past the block's end repeats its start)
                    __builtin_amdgcn_global_load_lds((csx_gptr)(val + base + (PACKED && e >= NENT ? 0 : e)), (csx_lptr)(M + k * 128), 16, 0, 0);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // all shares of the copy have landed
            if (!PACKED) {
                if (sub == 0 && lane < BS) DG[lane] = dv;       // behind the DMA overrun of the last chunk
                __syncthreads();
            }
        }
        dense_exact_pass_dpp<BS, pass == 1, MIX, PACKED>(x, M, DG, lane);
    };
    run_pass(std::integral_constant<int, 0>{});
    run_pass(std::integral_constant<int, 1>{});
#pragma unroll
    for (int a = 0; a < BS; a++) {
        const int32_t row = __builtin_amdgcn_readlane(jrow, a);
        if (live && row >= 0) __builtin_nontemporal_store(x[a], B + (int64_t)row * nrhs + rhs);
    }
}
// (Round 4 tried the L values as SCALAR operands: a lane is a right-hand side, so an L value is the same for the whole wave,
// v_mul_f64 takes a scalar register, and a term is then the reference's two operations and nothing else.  The packed rows read
// with s_load_dwordx16 off a wave-uniform base, every line of the block touched by one vector load first so that the scalar
// loads find it in L2; bit-identical, 8.3 ms at blocks of 64 against 4.7 for the DPP form, 3.0 against 2.9 at blocks of 32: a
// scalar load that misses the 16 KB scalar cache returns after ~650 cycles, all of a wave's scalar loads share ONE counter that
// can only be waited down to zero, and ~100 scalar registers hold 16 values to wait behind -- 250 exposed round trips per
// wave.  The division by the diagonal as a multiplication by its correctly rounded reciprocal with two exact corrections
// (Markstein) was built with it, bit-identical, and saves nothing: the compiler's IEEE division is 13 instructions here, not
// 35.  Removed; profiles/r04_ablation.md section 7.)
#pragma clang fp contract(fast)

// ---- dense blocks on the matrix cores -------------------------------------------------------------------
// For a dense 16 NB x 16 NB block the two substitutions are a blocked TRSM on 16 x 16 tiles:
//     forward :  X_i <- W_ii (X_i - sum_{j<i} L_ij X_j)        backward:  X_i <- W_ii' (X_i - sum_{j>i} L_ji' X_j)
// with W_ii = inv(L_ii) formed once at plan time.  Every product is a v_mfma_f64_16x16x4_f64 chain.  Why
// this is the right unit here although fp64 MFMA has no rate advantage on this chip: the L operand
// reaches the pipe as ONE double per lane (A fragment: lane l holds A[l & 15][4 s + (l >> 4)]), read once,
// coalesced, straight from memory -- the lane-per-right-hand-side kernels above need every L value in all
// 64 lanes, and that broadcast (LDS return path, 8 clk per 16 bytes) is what bounds them.  And the f64
// accumulator layout (lane l, register r: row (l >> 4) + 4 r, column l & 15) IS the B-fragment layout of
// k-step r, so a finished tile X_j feeds the next product from its registers: no LDS, no lane movement.
// One wave: one block x 64 right-hand sides = NB x 4 tiles = the same 128 VGPRs of unknowns as before.
// Results equal the substitution kernels to rounding (different association; explicit block inverses), so
// the plan refuses this path when a block inverse is large (max|W| max|L| > 1e4: the error of a product with an explicit inverse grows with it).
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2w __attribute__((ext_vector_type(2)));

// one wave per tree: the A fragments of W_ii = inv(L_ii) (tile i of the block at Wt + (t NB + i) 256: k-step sx, lane (m, kq) holds
// W_ii[m][4 sx + kq]) and, when `lcopy` is given, the block's packed columns (diagonal first, rows ascending: L.x's own layout, block
// t at t BS (BS + 1) / 2), both from the plan's forward programs
template <int NB>
__global__ __launch_bounds__(64) void k_mfma_frags(const Tree *__restrict__ trees, const int32_t *__restrict__ f_ptr,
                                                   const double *__restrict__ f_val, const double *__restrict__ diagk,
                                                   double *__restrict__ Wt, double *__restrict__ lcopy, unsigned long long *cond_bits) {
    constexpr int BS = 16 * NB;
    __shared__ double Ls[BS][BS + 1];
    __shared__ double W[NB][16][17];
    const int lane = threadIdx.x;
    const int32_t t = blockIdx.x, first = trees[t].first, base = f_ptr[first];
    double lmax = 0.0;
    // row a of the packed triangle: lanes c < a (one coalesced request per row, 16 rows' requests in flight; the loop over
    // all BS x BS positions it replaces waited for every load: 5.3 ms of the 5M-row plan)
    const double dgl = lane < BS ? diagk[first + lane] : 0.0;
#pragma unroll
    for (int a0 = 0; a0 < BS; a0 += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int a = a0 + k;
            v[k] = lane < a ? f_val[base + a * (a - 1) / 2 + lane] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int a = a0 + k;
            if (lane < BS) Ls[a][lane] = lane == a ? dgl : v[k];
            lmax = fmax(lmax, fabs(v[k]));
        }
    }
    lmax = fmax(lmax, fabs(dgl));
    __syncthreads();
    double wmax = 0.0;
    {
        const int blk = lane >> 4, col = lane & 15;
        if (blk < NB) {
            double wcol[16];
            tile_inverse_column(&Ls[16 * blk][16 * blk], BS + 1, col, wcol);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                W[blk][r][col] = wcol[r];
                wmax = fmax(wmax, fabs(wcol[r]));
            }
        }
    }
    __syncthreads();
    const int m = lane & 15, kq = lane >> 4;
    double *F = Wt + (size_t)t * (NB * 256) + lane;
    for (int i = 0; i < NB; i++)
        for (int sx = 0; sx < 4; sx++) F[(i * 4 + sx) * 64] = W[i][m][4 * sx + kq];
    if (lcopy) {
        double *C = lcopy + (size_t)t * (BS * (BS + 1) / 2);
        for (int c = 0; c < BS; c++)                       // column c: rows c .. BS - 1, contiguous
            if (lane >= c && lane < BS) C[c * BS - c * (c - 1) / 2 + lane - c] = Ls[lane][c];
    }
    // largest |W| |L| over the forest, as ordered bits of a non-negative double
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        lmax = fmax(lmax, __shfl_xor(lmax, d, 64));
        wmax = fmax(wmax, __shfl_xor(wmax, d, 64));
    }
    if (lane == 0) atomicMax(cond_bits, (unsigned long long)__double_as_longlong(lmax * wmax));
}

// The solve.  Off-diagonal operands are read where the factorisation left them (round 5; until then a second, re-arranged and negated
// copy of them was written beside L.x -- 1 GB per factorisation at 5M rows, a fifth of csx_cholsol_factor's traffic): the A fragment
// of -L_ij for lane (m, kq), k-step sx, is -L(16 i + m, 16 j + 4 sx + kq), the backward sweep's -L_ji' takes L(16 j + 4 sx + kq,
// 16 i + m); the sign is flipped in the register.  In the packed columns neither is a run of 512 aligned bytes (straight from memory
// the forward operand touches eight lines an instruction instead of four and the solve lost 5 %), so the block's packed columns --
// one contiguous, 16-byte aligned piece of L.x: block t at t BS (BS + 1) / 2 -- are copied to LDS once by the waves that solve it
// (global_load_lds, 1 KB an instruction, every line of the block requested once per workgroup instead of twice per wave), and both
// sweeps take their operands there.  SHARE blocks per workgroup, 4 / SHARE waves (chunks of 64 right-hand sides) per block -- the
// host picks the SHARE whose waves-per-block divides the number of chunks.  Lv: L.x itself (a plan on all the columns of L, equal
// blocks) or the plan's packed copy.
template <int NB, int SHARE>
__global__ __launch_bounds__(256, 2) void k_cholsol_mfma(const Tree *__restrict__ trees, int32_t ntrees,
                                                      const int32_t *__restrict__ nodes, const int32_t *__restrict__ perm,
                                                      const double *__restrict__ Lv, const double *__restrict__ Wt, double *B,
                                                      int32_t nrhs, int32_t chunks) {
    constexpr int BS = 16 * NB, NENT = BS * (BS + 1) / 2, WPT = 4 / SHARE;
    constexpr int LSZ = (NENT + 127) / 128 * 128;            // whole copy instructions (128 doubles each)
    constexpr bool WLDS = SHARE <= 2;                        // the W tiles too (2 KB a tile), where two workgroups still fit a CU
    __shared__ __attribute__((aligned(16))) double s_l[SHARE][LSZ];
    __shared__ __attribute__((aligned(16))) double s_w[WLDS ? SHARE : 1][WLDS ? NB * 256 : 2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = w / WPT, sub = w % WPT;
    const int32_t cgroups = chunks / WPT;
    const int32_t tg = (int32_t)(blockIdx.x / cgroups), cg = (int32_t)(blockIdx.x % cgroups);
    const int32_t t_raw = tg * SHARE + slot, h = cg * WPT + sub;
    const bool valid = t_raw < ntrees;                      // a wave past the last block still copies and meets the barrier
    const int32_t t = valid ? t_raw : ntrees - 1;
    const int32_t first = trees[t].first;
    const int col = lane & 15, rq = lane >> 4;
    // rows of B this lane touches: local row 16 i + rq + 4 r
    int64_t roff[NB][4];
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int32_t jr = nodes[first + 16 * i + rq + 4 * r];
            if (perm) jr = perm[jr];
            roff[i][r] = (int64_t)jr * nrhs;
        }
    f64x4 X[NB][4];
    bool live[4];
    int32_t cidx[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int32_t rhs = h * 64 + 16 * c + col;
        live[c] = valid && rhs < nrhs;
        cidx[c] = rhs < nrhs ? rhs : nrhs - 1;   // clamped: loaded, never stored
    }
    // A chunk that is wholly inside the block (and an even nrhs: 16-byte alignment) is moved 16 bytes per lane: the
    // right-hand sides are independent, so which one a (c, col) pair stands for is free -- lane (rq, col) takes the
    // two neighbours 32 c' + 2 col, + 1 of a row with one load and gives them to column chunks 2 c' and 2 c' + 1.
    // A wave instruction then moves 4 rows x 256 contiguous bytes instead of 4 x 128: half the memory instructions.
    const bool wide = (nrhs & 1) == 0 && h * 64 + 64 <= nrhs && (reinterpret_cast<uintptr_t>(B) & 15) == 0;   // uniform
    if (wide) {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int cp = 0; cp < 2; cp++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // (B goes through once per batch, 10 GB of it at 5M rows x 128: the caches are told not to keep it)
                    const f64x2w v = __builtin_nontemporal_load(reinterpret_cast<const f64x2w *>(B + roff[i][r] + h * 64 + 32 * cp + 2 * col));
                    X[i][2 * cp][r] = v.x;
                    X[i][2 * cp + 1][r] = v.y;
                }
    } else {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) X[i][c][r] = B[roff[i][r] + cidx[c]];
    }
    // the block's packed columns to LDS, this wave's share of the copy instructions (a lane past the block's end repeats its start)
    {
        const double *Lb = Lv + (size_t)t * NENT;
#pragma unroll
        for (int k = 0; k < LSZ / 128; k++)
            if (k % WPT == sub) {
                const int e = k * 128 + 2 * lane;
                __builtin_amdgcn_global_load_lds((csx_gptr)(Lb + (e < NENT ? e : 0)), (csx_lptr)(s_l[slot] + k * 128), 16, 0, 0);
            }
        if (WLDS) {
            const double *Wb = Wt + (size_t)t * (NB * 256);
#pragma unroll
            for (int k = 0; k < NB * 2; k++)
                if ((k + LSZ / 128) % WPT == sub)
                    __builtin_amdgcn_global_load_lds((csx_gptr)(Wb + k * 128 + 2 * lane), (csx_lptr)(s_w[WLDS ? slot : 0] + k * 128), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // entry (R, C), R >= C, of the packed columns: C BS - C (C - 1) / 2 + R - C
    const double *Ls = s_l[slot];
    auto col_at = [](int C) { return C * BS - C * (C - 1) / 2 - C; };
    const double *Wb = WLDS ? s_w[WLDS ? slot : 0] : Wt + (size_t)t * (NB * 256);
    const double *F = Wb + lane;
    // Transposed read of a stored tile: the A fragment of tile' for (lane = (m, kq), k-step sx) is element (4 sx + kq, m)
    // of the tile, which the forward layout keeps in k-step m >> 2 at lane (m & 3) * 16 + 4 sx + kq.
    const double *Ft = Wb + (col >> 2) * 64 + (col & 3) * 16 + rq;
#pragma unroll
    for (int i = 0; i < NB; i++) {
#pragma unroll
        for (int j = 0; j < i; j++)
#pragma unroll
            for (int sx = 0; sx < 4; sx++) {
                const double a = -Ls[col_at(16 * j + 4 * sx + rq) + 16 * i + col];
#pragma unroll
                for (int c = 0; c < 4; c++) X[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[j][c][sx], X[i][c], 0, 0, 0);
            }
        f64x4 Y[4];
#pragma unroll
        for (int c = 0; c < 4; c++) Y[c] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            const double a = F[(i * 4 + sx) * 64];
#pragma unroll
            for (int c = 0; c < 4; c++) Y[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[i][c][sx], Y[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) X[i][c] = Y[c];
    }
#pragma unroll
    for (int i = NB - 1; i >= 0; i--) {
#pragma unroll
        for (int j = i + 1; j < NB; j++)
#pragma unroll
            for (int sx = 0; sx < 4; sx++) {
                const double a = -Ls[col_at(16 * i + col) + 16 * j + 4 * sx + rq];   // -L_ji'
#pragma unroll
                for (int c = 0; c < 4; c++) X[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[j][c][sx], X[i][c], 0, 0, 0);
            }
        f64x4 Y[4];
#pragma unroll
        for (int c = 0; c < 4; c++) Y[c] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            const double a = Ft[(size_t)(i * 4) * 64 + 4 * sx];                      // W_ii' from the stored W_ii
#pragma unroll
            for (int c = 0; c < 4; c++) Y[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[i][c][sx], Y[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) X[i][c] = Y[c];
    }
    if (!valid) return;
    if (wide) {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int cp = 0; cp < 2; cp++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    f64x2w v;
                    v.x = X[i][2 * cp][r];
                    v.y = X[i][2 * cp + 1][r];
                    __builtin_nontemporal_store(v, reinterpret_cast<f64x2w *>(B + roff[i][r] + h * 64 + 32 * cp + 2 * col));
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (live[c]) B[roff[i][r] + cidx[c]] = X[i][c][r];
}

__global__ __launch_bounds__(256) void k_perm_rows(const int32_t *__restrict__ perm, const double *__restrict__ src,
                                                   double *__restrict__ dst, int32_t n, int32_t nrhs, int to_x) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * nrhs) return;
    const int64_t j = t / nrhs, r = t % nrhs;
    const int64_t k = perm[j];
    if (to_x) dst[t] = src[k * nrhs + r];        // x[j] = b[perm[j]]
    else dst[k * nrhs + r] = src[t];             // b[perm[j]] = x[j]
}

// ---- plan of a forest of EQUAL DENSE BLOCKS, straight from L ------------------------------------------------------
// What the dense-block kernels read (k_cholsol_dense_exact_dpp, k_cholsol_dense, k_mfma_frags) is a re-arrangement of
// the block's packed triangle: forward program = rows of the strictly lower triangle, row-major (f_val); backward program
// = for sweep position sp (row bs - 1 - sp) the column below the diagonal REVERSED (dense_b); the diagonals in both
// orders.  The general plan gets there through two triangular-solve analyses (a stable transpose of L each), the forest
// partition on the host and two packing kernels: 61 ms at 5M rows.  Here a workgroup stages its block's triangle in LDS
// (one coalesced read of L.x) and writes the four arrays (coalesced): L.x read once, 2 x lnz values written.
template <bool LOCAL>
__global__ __launch_bounds__(256) void k_clique_plan(int32_t ntrees, int32_t bs, const int32_t *__restrict__ Lp,
                                                     const double *__restrict__ Lx, Tree *trees, int32_t *tree_nodes,
                                                     int32_t *f_ptr, int32_t *b_ptr, double *f_val, double *dense_b,
                                                     double *diagk, double *diagb, int32_t *f_idx, int32_t *b_idx,
                                                     double *b_val, int *zero) {
    __shared__ double tri[64 * 65 / 2];
    const int32_t t = blockIdx.x;
    const int32_t first = t * bs, NT = bs * (bs - 1) / 2, nent = bs * (bs + 1) / 2;
    const int64_t base = Lp[first];
    for (int e = threadIdx.x; e < nent; e += 256) tri[e] = Lx[base + e];
    __syncthreads();
    const int64_t po = (int64_t)t * NT;
    // element (a, c), a >= c, of the block: tri[c bs - c (c - 1) / 2 + a - c]
    for (int idx = threadIdx.x; idx < bs * bs; idx += 256) {
        const int a = idx / bs, c = idx % bs;
        if (c < a) {
            const int64_t fo = po + a * (a - 1) / 2 + c;
            const double v = tri[c * bs - c * (c - 1) / 2 + a - c];
            if (!LOCAL) f_val[fo] = v;
            else f_idx[fo] = c * 64;
            // backward: sweep position sp = a, term q = c: L(bs - 1 - q, bs - 1 - sp); the fused kernel's order: L(col + 1 + q, col)
            const int col = bs - 1 - a;
            if (!LOCAL) {
                const int row = bs - 1 - c;
                dense_b[fo] = tri[col * bs - col * (col - 1) / 2 + row - col];
            } else {
                const int row = col + 1 + c;
                b_idx[fo] = row * 64;
                b_val[fo] = tri[col * bs - col * (col - 1) / 2 + row - col];
            }
        }
    }
    if (LOCAL) return;
    if (threadIdx.x < bs) {
        const int a = threadIdx.x;
        const double d = tri[a * bs - a * (a - 1) / 2];
        if (d == 0.0) *zero = 1;
        diagk[first + a] = d;
        diagb[first + bs - 1 - a] = d;
        f_ptr[first + a] = (int32_t)(po + a * (a - 1) / 2);
        b_ptr[first + a] = (int32_t)(po + a * (a - 1) / 2);
        tree_nodes[first + a] = first + a;
    }
    if (threadIdx.x == 0) {
        trees[t] = Tree{first, bs};
        if (t == ntrees - 1) f_ptr[first + bs] = b_ptr[first + bs] = (int32_t)(po + NT);
    }
}

static int cholsol_plan_clique(CholPlan *P, int32_t bs) {
    hipStream_t s = ctx().stream;
    const Csc *L = P->L;
    const int32_t n = P->n, ntrees = n / bs;
    const int64_t tot = (int64_t)ntrees * (bs * (bs - 1) / 2);
    DevScope tmp;
    int *zero = nullptr;
    CSX_TRY(tmp.alloc(&zero, 1));
    CSX_HIP(hipMemsetAsync(zero, 0, sizeof(int), s));
    if (!P->trees) CSX_TRY(dalloc(&P->trees, (size_t)ntrees));          // (a plan csx_cholsol_factor made has its block list already)
    if (!P->tree_nodes) CSX_TRY(dalloc(&P->tree_nodes, (size_t)n));
    CSX_TRY(dalloc(&P->f_ptr, (size_t)n + 1));
    CSX_TRY(dalloc(&P->b_ptr, (size_t)n + 1));
    CSX_TRY(dalloc(&P->diagk, (size_t)n));
    CSX_TRY(dalloc(&P->diagb, (size_t)n));
    CSX_TRY(dalloc(&P->f_val, (size_t)tot + 128));
    CSX_TRY(dalloc(&P->dense_b, (size_t)tot + 128));
    hipLaunchKernelGGL(k_clique_plan<false>, dim3((unsigned)ntrees), dim3(256), 0, s, ntrees, bs, L->p, L->x, P->trees,
                       P->tree_nodes, P->f_ptr, P->b_ptr, P->f_val, P->dense_b, P->diagk, P->diagb, nullptr, nullptr, nullptr,
                       zero);
    CSX_LAUNCH_CHECK();
    int hz = 0;
    CSX_HIP(hipMemcpyAsync(&hz, zero, sizeof hz, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    P->clique = true;
    P->clique_zero_pivot = hz != 0;
    P->ntrees = ntrees;
    P->max_nodes = bs;
    P->local = true;
    P->dense_bs = bs;
    return CSX_OK;
}

// the fused per-tree kernel's programs for such a plan ("cholsol.dense_blocks" = 0 at solve time)
static int cholsol_clique_local(CholPlan *P) {
    if (P->f_idx) return CSX_OK;
    hipStream_t s = ctx().stream;
    const int32_t bs = P->dense_bs;
    const int64_t tot = (int64_t)P->ntrees * (bs * (bs - 1) / 2);
    CSX_TRY(dalloc(&P->f_idx, (size_t)tot + 8));
    CSX_TRY(dalloc(&P->b_idx, (size_t)tot + 8));
    CSX_TRY(dalloc(&P->b_val, (size_t)tot + 128));
    hipLaunchKernelGGL(k_clique_plan<true>, dim3((unsigned)P->ntrees), dim3(256), 0, s, P->ntrees, bs, P->L->p, P->L->x, P->trees,
                       P->tree_nodes, P->f_ptr, P->b_ptr, P->f_val, P->dense_b, P->diagk, P->diagb, P->f_idx, P->b_idx, P->b_val,
                       nullptr);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

static int cholsol_plan(const Csc *L, const int32_t *pinv, CholPlan **out) {
    hipStream_t s = ctx().stream;
    CholPlan *P = new CholPlan();
    *out = P;
    const int32_t n = L->n;
    P->n = n;
    P->L = L;
    const bool timing = getenv("CSX_CHOL_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[cholsol_plan] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    if (pinv) {
        std::vector<int32_t> perm((size_t)n);
        for (int32_t k = 0; k < n; k++) {
            if (pinv[k] < 0 || pinv[k] >= n) return CSX_EINVAL;
            perm[(size_t)pinv[k]] = k;
        }
        CSX_TRY(upload(&P->perm, perm));
    }
    if (n > 0 && ctx().opt.chol_clique && ctx().opt.cholsol_dense_blocks) {
        int32_t bs = 0;
        CSX_TRY(clique_factor_block_size(L, &bs));
        if (bs == 8 || bs == 16 || bs == 32 || bs == 64) return cholsol_plan_clique(P, bs);
    }
    CSX_TRY(tri_analyse_raw(L, CSX_TRI_L, &P->fwd));
    lap("analysis of L");
    CSX_TRY(tri_analyse_raw(L, CSX_TRI_LT, &P->bwd));
    lap("analysis of L'");
    tri_set_mate(P->bwd, P->fwd);   // the rounding-equal order may run L' in push form on the rows of L
    if (n == 0) return CSX_OK;
    // forest of small trees?  (needs a Cholesky-shaped L: diagonal first, rows ascending)
    DevScope tmp;   // d_parent, d_flag, flen, blen: released on every exit
    int32_t *d_parent = nullptr;
    int *d_flag = nullptr;
    CSX_TRY(tmp.alloc(&d_parent, (size_t)n));
    CSX_TRY(tmp.alloc(&d_flag, 1));
    CSX_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), s));
    const bool short_cols = (int64_t)L->nnz < 8 * (int64_t)n;     // a wave per column of three entries is 61 idle lanes
    if (short_cols)
        hipLaunchKernelGGL(k_parent_of_sorted_L<4>, dim3((unsigned)(((int64_t)n + 63) / 64)), dim3(256), 0, s, n, L->p, L->i, d_parent,
                           d_flag);
    else
        hipLaunchKernelGGL(k_parent_of_sorted_L<64>, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, L->p, L->i, d_parent,
                           d_flag);
    // trees on consecutive columns (block-diagonal factors in their natural order): the partition without the host
    int32_t ntrees = 0, max_tree = 0;
    bool on_device = false;
    Forest F;
    if (ctx().opt.chol_clique && ctx().opt.chol_forest) {
        int32_t *nrm = nullptr, *is_start = nullptr, *block_id = nullptr, *start = nullptr;
        int *stats = nullptr;   // [0] spare, [1] roots, [2] widest block
        CSX_TRY(tmp.alloc(&nrm, (size_t)n));
        CSX_TRY(tmp.alloc(&is_start, (size_t)n + 1));
        CSX_TRY(tmp.alloc(&block_id, (size_t)n + 1));
        CSX_TRY(tmp.alloc(&stats, 4));
        CSX_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(int), s));
        const unsigned nbk = (unsigned)(((int64_t)n + 255) / 256);
        hipLaunchKernelGGL(k_plan_reach, dim3(nbk), dim3(256), 0, s, n, L->p, L->i, nrm);
        CSX_TRY(suffix_min_i32(nrm, n));
        hipLaunchKernelGGL(k_plan_starts, dim3(nbk), dim3(256), 0, s, n, nrm, d_parent, is_start, stats);
        int64_t nblocks = 0;
        CSX_TRY(scan_exclusive_i32(is_start, block_id, n, &nblocks));
        int hs[4] = {0, 0, 0, 0}, unsorted = 0;
        CSX_HIP(hipMemcpyAsync(hs, stats, sizeof hs, hipMemcpyDeviceToHost, s));
        CSX_HIP(hipMemcpyAsync(&unsorted, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        if (unsorted) return CSX_OK;
        if (hs[1] == nblocks) {                  // one root to a block: the blocks ARE the trees
            CSX_TRY(tmp.alloc(&start, (size_t)nblocks + 1));
            CSX_TRY(dalloc(&P->trees, (size_t)nblocks));
            hipLaunchKernelGGL(k_plan_block_first, dim3(nbk), dim3(256), 0, s, n, is_start, block_id, start);
            hipLaunchKernelGGL(k_plan_trees, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, s, (int32_t)nblocks, start, P->trees,
                               stats);
            CSX_HIP(hipMemcpyAsync(hs, stats, sizeof hs, hipMemcpyDeviceToHost, s));
            CSX_HIP(hipStreamSynchronize(s));
            if (hs[2] <= 256) {
                CSX_TRY(dalloc(&P->tree_nodes, (size_t)n));
                CSX_TRY(dalloc(&P->local_id, (size_t)n));
                CSX_TRY(dalloc(&P->rev_pos, (size_t)n));
                hipLaunchKernelGGL(k_plan_nodes, dim3(nbk), dim3(256), 0, s, n, is_start, block_id, start, P->tree_nodes, P->local_id,
                                   P->rev_pos);
                CSX_LAUNCH_CHECK();
                ntrees = (int32_t)nblocks;
                max_tree = hs[2];
                on_device = true;
                lap("partition (device)");
            } else {
                dfree(P->trees);
                P->trees = nullptr;
            }
        }
    }
    if (!on_device) {
        std::vector<int32_t> parent((size_t)n);
        int unsorted = 0;
        CSX_HIP(hipMemcpyAsync(parent.data(), d_parent, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipMemcpyAsync(&unsorted, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        if (unsorted) return CSX_OK;
        lap("parent of L");
        partition_forest(n, parent.data(), F);
        lap("partition (host)");
        if (!F.level_cols.empty() || F.max_tree > 256) {   // some tree is too big for LDS: the level-scheduled solves
        {
                // The level sets of the two solves with a Cholesky factor are heights (L x = b) and depths (L' x = b) in its
                // elimination tree: proposed to the plans, which verify them in one pass over the pattern instead of finding
                // them level by level (a nested-dissection factor of a 700 x 700 grid has ~3 000 levels).
                bool tree = true;
                for (int32_t j = 0; j < n && tree; j++) tree = parent[(size_t)j] < 0 || (parent[(size_t)j] > j && parent[(size_t)j] < n);
                if (tree) {
                    std::vector<int32_t> height((size_t)n, 0), depth((size_t)n, 0);
                    for (int32_t j = 0; j < n; j++)
                        if (parent[(size_t)j] >= 0)
                            height[(size_t)parent[(size_t)j]] = std::max(height[(size_t)parent[(size_t)j]], height[(size_t)j] + 1);
                    for (int32_t j = n - 1; j >= 0; j--)
                        if (parent[(size_t)j] >= 0) depth[(size_t)j] = depth[(size_t)parent[(size_t)j]] + 1;
                    for (int32_t j = 0; j < n; j++) P->col_levels = std::max(P->col_levels, height[(size_t)j] + 1);
                    P->parent_h = parent;
                    tri_set_level_hint(P->fwd, std::move(height));
                    tri_set_level_hint(P->bwd, std::move(depth));
                }
            }
            return CSX_OK;
        }
        std::vector<int32_t> local((size_t)n, 0);
        for (const Tree &t : F.small)
            for (int32_t a = 0; a < t.count; a++) local[(size_t)F.small_cols[(size_t)(t.first + a)]] = a;
        CSX_TRY(upload(&P->trees, F.small));
        CSX_TRY(upload(&P->tree_nodes, F.small_cols));
        CSX_TRY(upload(&P->local_id, local));
        std::vector<int32_t> rev((size_t)n, 0);
        for (const Tree &t : F.small)
            for (int32_t a = 0; a < t.count; a++) rev[(size_t)(t.first + a)] = t.first + t.count - 1 - a;
        CSX_TRY(upload(&P->rev_pos, rev));
        lap("node lists (host)");
        ntrees = (int32_t)F.small.size();
        max_tree = F.max_tree;
    }
    const int32_t *Gp, *Gi;
    const double *Gx, *Gd;
    tri_gather_arrays(P->fwd, &Gp, &Gi, &Gx, &Gd);
    int32_t *flen = nullptr, *blen = nullptr;
    CSX_TRY(tmp.alloc(&flen, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&blen, (size_t)n + 1));
    CSX_TRY(dalloc(&P->f_ptr, (size_t)n + 1));
    CSX_TRY(dalloc(&P->b_ptr, (size_t)n + 1));
    CSX_TRY(dalloc(&P->diagk, (size_t)n));
    CSX_TRY(dalloc(&P->diagb, (size_t)n));
    const unsigned nb = (unsigned)(((int64_t)n + 255) / 256);
    hipLaunchKernelGGL(k_pack_len, dim3(nb), dim3(256), 0, s, n, P->tree_nodes, P->rev_pos, Gp, L->p, flen, blen);
    int64_t ftot = 0, btot = 0;
    int st = scan_exclusive_i32(flen, P->f_ptr, n, &ftot);
    if (st == CSX_OK) st = scan_exclusive_i32(blen, P->b_ptr, n, &btot);
    CSX_TRY(st);
    CSX_TRY(dalloc(&P->f_idx, (size_t)ftot + 8));
    CSX_TRY(dalloc(&P->f_val, (size_t)ftot + 128));
    CSX_TRY(dalloc(&P->b_idx, (size_t)btot + 8));
    CSX_TRY(dalloc(&P->b_val, (size_t)btot + 128));
    if (short_cols)
        hipLaunchKernelGGL(k_pack_fill<4>, dim3((unsigned)(((int64_t)n + 63) / 64)), dim3(256), 0, s, n, P->tree_nodes, P->rev_pos,
                           P->local_id, Gp, Gi, Gx, L->p, L->i, L->x, P->f_ptr, P->f_idx, P->f_val, P->b_ptr, P->b_idx, P->b_val,
                           P->diagk, P->diagb);
    else
        hipLaunchKernelGGL(k_pack_fill<64>, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, P->tree_nodes, P->rev_pos,
                           P->local_id, Gp, Gi, Gx, L->p, L->i, L->x, P->f_ptr, P->f_idx, P->f_val, P->b_ptr, P->b_idx, P->b_val,
                           P->diagk, P->diagb);
    CSX_LAUNCH_CHECK();
    CSX_HIP(hipStreamSynchronize(s));
    P->ntrees = ntrees;
    P->max_nodes = max_tree;
    P->local = true;
    lap("programs packed");
    // dense blocks? same size, contiguous rows, column c of a block holding exactly bs - c entries
    // (a forest partitioned on the device is no forest of equal dense blocks: cholsol_plan_clique would have taken it)
    const int32_t bs = on_device ? 0 : F.max_tree;
    if (bs == 8 || bs == 16 || bs == 32 || bs == 64) {
        std::vector<int32_t> hLp((size_t)n + 1);
        CSX_HIP(hipMemcpyAsync(hLp.data(), L->p, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        bool dense = true;
        for (const Tree &t : F.small) {
            if (t.count != bs) {
                dense = false;
                break;
            }
            const int32_t j0 = F.small_cols[(size_t)t.first];
            for (int32_t c = 0; c < bs && dense; c++) {
                const int32_t j = F.small_cols[(size_t)(t.first + c)];
                if (j != j0 + c || hLp[(size_t)j + 1] - hLp[(size_t)j] != bs - c) dense = false;
            }
            if (!dense) break;
        }
        if (dense) {
            int32_t btot_h = 0;
            CSX_HIP(hipMemcpyAsync(&btot_h, P->b_ptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            CSX_HIP(hipStreamSynchronize(s));
            CSX_TRY(dalloc(&P->dense_b, (size_t)btot_h + 128));
            hipLaunchKernelGGL(k_dense_reverse_rows, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, P->b_ptr,
                               P->b_val, P->dense_b);
            CSX_LAUNCH_CHECK();
            CSX_HIP(hipStreamSynchronize(s));
            P->dense_bs = bs;
        }
    }
    return CSX_OK;
}

// Blocked-TRSM operands for the matrix cores (rounding-equal path, built the first time a plan is switched to
// exact = 0).  The path is refused when a block inverse is large: a product with an explicit inverse carries an
// error of about growth x eps x block size, and 1e-10 is the budget (tests/test_gpu_cholesky.py sweeps the
// region below the guard).
constexpr double MFMA_GROWTH_LIMIT = 1e3;

// forests of small trees of any shape (max_nodes <= 80): the trees' forward programs made dense, bucketed by size, on the matrix cores
static int cholsol_build_ragged(CholPlan *P) {
    if (P->rag_tried || !P->local || P->dense_bs || (!P->f_idx && !P->lite) || P->max_nodes > RAG_MAX_ROWS) return CSX_OK;
    P->rag_tried = true;
    RaggedMfma *R = nullptr;
    if (P->lite)      // (csx_cholsol_factor's plan of a forest on consecutive columns: straight from L's columns)
        CSX_TRY(ragged_build(P->trees, P->ntrees, P->max_nodes, P->tree_nodes, nullptr, nullptr, nullptr, nullptr, false, &R, P->L));
    else
        CSX_TRY(ragged_build(P->trees, P->ntrees, P->max_nodes, P->tree_nodes, P->f_ptr, P->f_idx, P->f_val, P->diagk, false, &R));
    if (!R) return CSX_OK;
    P->mfma_growth = R->growth;
    if (R->growth <= RAG_GROWTH_LIMIT) P->rag = R;     // (a NaN fails the comparison: the fused per-tree kernel stays)
    else ragged_free(R);
    return CSX_OK;
}

static int cholsol_build_mfma(CholPlan *P) {
    if (!P->dense_bs) return cholsol_build_ragged(P);
    if (P->mfma_tried || P->dense_bs < 16) return CSX_OK;
    if (P->clique && !P->f_val) CSX_TRY(cholsol_plan_clique(P, P->dense_bs));   // (a plan that has been the block list so far: k_mfma_frags reads programs)
    P->mfma_tried = true;
    hipStream_t s = ctx().stream;
    const int nb16 = P->dense_bs / 16;
    const int32_t bs = P->dense_bs;
    unsigned long long *cond = nullptr, hcond = 0;
    double *ff = nullptr, *lc = nullptr;
    int st = dalloc(&cond, 1);
    if (st == CSX_OK) st = dalloc(&ff, (size_t)P->ntrees * nb16 * 256);
    // (a plan on consecutive columns of L reads the off-diagonal tiles in L.x; the general analysis' plan gets a packed copy)
    // (... or whose L.x is not 16-byte aligned -- a wrapped pointer: the copies to LDS move 16 bytes a lane)
    if (st == CSX_OK && (!P->clique || (reinterpret_cast<uintptr_t>(P->L->x) & 15) != 0)) st = dalloc(&lc, (size_t)P->ntrees * (bs * (bs + 1) / 2));
    if (st == CSX_OK && hipMemsetAsync(cond, 0, sizeof(unsigned long long), s) != hipSuccess) st = CSX_ERUNTIME;
    if (st == CSX_OK) {
        const dim3 g((unsigned)P->ntrees);
        if (nb16 == 1)
            hipLaunchKernelGGL(k_mfma_frags<1>, g, dim3(64), 0, s, P->trees, P->f_ptr, P->f_val, P->diagk, ff, lc, cond);
        else if (nb16 == 2)
            hipLaunchKernelGGL(k_mfma_frags<2>, g, dim3(64), 0, s, P->trees, P->f_ptr, P->f_val, P->diagk, ff, lc, cond);
        else
            hipLaunchKernelGGL(k_mfma_frags<4>, g, dim3(64), 0, s, P->trees, P->f_ptr, P->f_val, P->diagk, ff, lc, cond);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(&hcond, cond, sizeof hcond, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            st = CSX_ERUNTIME;
    }
    dfree(cond);
    double growth = 0.0;
    std::memcpy(&growth, &hcond, sizeof growth);
    P->mfma_growth = growth;
    if (st == CSX_OK && growth <= MFMA_GROWTH_LIMIT) {   // (a NaN fails the comparison: substitution stays)
        P->frag_f = ff;
        P->lcopy = lc;
    } else {
        dfree(ff);
        dfree(lc);
    }
    return st;
}

// Supernodal schedule for the rounding-equal order of a big-tree plan (nullptr when the factor gains nothing from it).
static int cholsol_build_sn(CholPlan *P) {
    const bool say = std::getenv("CSX_CHOL_TIMING") != nullptr;
    if (say && !P->sn_tried)
        std::fprintf(stderr, "cholsol_build_sn: local %d tree %d col_levels %d\n", (int)P->local, (int)!P->parent_h.empty(), P->col_levels);
    if (P->sn_tried || P->local || P->parent_h.empty() || !ctx().opt.tri_supernodes) return CSX_OK;
    P->sn_tried = true;
    const int32_t *Gp, *Gi;
    const double *Gx, *Gd;
    tri_gather_arrays(P->fwd, &Gp, &Gi, &Gx, &Gd);
    if (!Gp || !Gd) return CSX_OK;
    hipStream_t s = ctx().stream;
    const int32_t n = P->n;
    std::vector<int32_t> hLp((size_t)n + 1), hGp((size_t)n + 1);
    CSX_HIP(hipMemcpyAsync(hLp.data(), P->L->p, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipMemcpyAsync(hGp.data(), Gp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    const int st = sn_build(P->L, P->parent_h.data(), hLp.data(), hGp.data(), Gp, Gi, Gx, P->col_levels, &P->sn);
    if (say) {
        int32_t a = 0, b = 0, c = 0;
        if (P->sn) sn_info(P->sn, &a, &b, &c);
        std::fprintf(stderr, "cholsol_build_sn: status %d plan %d supernodes %d levels %d max width %d\n", st, P->sn != nullptr, a, b, c);
    }
    return st;
}

// ---- exact order on forests of cliques of unequal sizes: padded size classes ------------------------------------------------
__global__ __launch_bounds__(256) void k_xc_class(const Tree *__restrict__ trees, int32_t ntrees, uint32_t *__restrict__ key,
                                                  uint32_t *__restrict__ id) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntrees) return;
    const int32_t c = trees[t].count;
    key[t] = c <= 8 ? 0u : c <= 16 ? 1u : c <= 32 ? 2u : c <= 48 ? 3u : 4u;
    id[t] = (uint32_t)t;
}

// k_clique_plan for a block of bs <= BS columns padded at its END with the identity up to BS: slot q of the class holds block
// list[q]; f_val = rows of the strictly lower triangle, row-major; dense_b = for sweep position sp (row BS - 1 - sp) the column
// below the diagonal reversed; the diagonals in both orders; the node list with -1 for the padding
template <int BS>
__global__ __launch_bounds__(256) void k_xc_plan(const uint32_t *__restrict__ list, const Tree *__restrict__ blocks,
                                                 const int32_t *__restrict__ Lp, const double *__restrict__ Lx, Tree *trees,
                                                 int32_t *nodes, int32_t *f_ptr, int32_t *b_ptr, double *f_val, double *dense_b,
                                                 double *diagk, double *diagb, int32_t count) {
    constexpr int NT = BS * (BS - 1) / 2;
    __shared__ double tri[BS * (BS + 1) / 2];
    const int32_t q = blockIdx.x;
    const Tree blk = blocks[list[q]];
    const int32_t c0 = blk.first, bs = blk.count;
    const int64_t base = Lp[c0];
    const int nent = bs * (bs + 1) / 2;
    for (int e = threadIdx.x; e < nent; e += 256) tri[e] = Lx[base + e];
    __syncthreads();
    // element (r, c), r >= c, of the padded block
    auto Lpad = [&](int r, int c) -> double {
        if (r < bs) return tri[c * bs - c * (c - 1) / 2 + r - c];       // (then c < bs too)
        return r == c ? 1.0 : 0.0;
    };
    const int64_t po = (int64_t)q * NT;
    for (int idx = threadIdx.x; idx < BS * BS; idx += 256) {
        const int a = idx / BS, c = idx % BS;
        if (c < a) {
            f_val[po + a * (a - 1) / 2 + c] = Lpad(a, c);
            dense_b[po + a * (a - 1) / 2 + c] = Lpad(BS - 1 - c, BS - 1 - a);
        }
    }
    if (threadIdx.x < BS) {
        const int a = threadIdx.x;
        const double d = Lpad(a, a);
        diagk[q * BS + a] = d;
        diagb[q * BS + BS - 1 - a] = d;
        f_ptr[q * BS + a] = (int32_t)(po + a * (a - 1) / 2);
        b_ptr[q * BS + a] = (int32_t)(po + a * (a - 1) / 2);
        nodes[q * BS + a] = a < bs ? c0 + a : -1;
    }
    if (threadIdx.x == 0) {
        trees[q] = Tree{q * BS, BS};
        if (q == count - 1) f_ptr[count * BS] = b_ptr[count * BS] = (int32_t)(po + NT);
    }
}

static int cholsol_exact_classes_build(CholPlan *P) {
    if (P->xc_built) return CSX_OK;
    hipStream_t s = ctx().stream;
    const int32_t nt = P->ntrees;
    DevScope tmp;
    uint32_t *key = nullptr, *id = nullptr, *skey = nullptr, *list = nullptr;
    int32_t *bounds = nullptr;
    CSX_TRY(tmp.alloc(&key, (size_t)nt));
    CSX_TRY(tmp.alloc(&id, (size_t)nt));
    CSX_TRY(tmp.alloc(&skey, (size_t)nt));
    CSX_TRY(tmp.alloc(&list, (size_t)nt));
    CSX_TRY(tmp.alloc(&bounds, 6));
    hipLaunchKernelGGL(k_xc_class, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, P->trees, nt, key, id);
    CSX_LAUNCH_CHECK();
    CSX_TRY(stable_sort_by_key(key, id, nullptr, nt, 5, skey, list, nullptr));
    CSX_TRY(boundaries_from_sorted(skey, nt, 5, bounds));
    int32_t hb[6] = {0, 0, 0, 0, 0, 0};
    CSX_HIP(hipMemcpyAsync(hb, bounds, sizeof hb, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    // (a class of 48 since late in round 5: a block of 33 .. 48 columns padded to 64 did 1.8 times the terms it does padded to 48)
    static const int kBS[5] = {8, 16, 32, 48, 64};
    for (int c = 0; c < 5; c++) {
        CholPlan::ExactClass &X = P->xc[c];
        X.count = hb[c + 1] - hb[c];
        if (X.count <= 0) continue;
        const int BS = kBS[c];
        const size_t rows = (size_t)X.count * BS, terms = (size_t)X.count * (BS * (BS - 1) / 2);
        if (terms > 0x7fffffffull) return CSX_OK;            // (32-bit program offsets: the general plan takes over; nothing built is used)
        CSX_TRY(dalloc(&X.trees, (size_t)X.count));
        CSX_TRY(dalloc(&X.nodes, rows));
        CSX_TRY(dalloc(&X.f_ptr, rows + 1));
        CSX_TRY(dalloc(&X.b_ptr, rows + 1));
        CSX_TRY(dalloc(&X.diagk, rows));
        CSX_TRY(dalloc(&X.diagb, rows));
        CSX_TRY(dalloc(&X.f_val, terms + 128));
        CSX_TRY(dalloc(&X.dense_b, terms + 128));
#define CSX_XCP(B_)                                                                                                                \
    hipLaunchKernelGGL(k_xc_plan<B_>, dim3((unsigned)X.count), dim3(256), 0, s, list + hb[c], P->trees, P->L->p, P->L->x, X.trees, \
                       X.nodes, X.f_ptr, X.b_ptr, X.f_val, X.dense_b, X.diagk, X.diagb, X.count)
        switch (BS) {
            case 8: CSX_XCP(8); break;
            case 16: CSX_XCP(16); break;
            case 32: CSX_XCP(32); break;
            case 48: CSX_XCP(48); break;
            default: CSX_XCP(64); break;
        }
#undef CSX_XCP
        CSX_LAUNCH_CHECK();
    }
    CSX_HIP(hipStreamSynchronize(s));       // (list is a temporary of this function)
    P->xc_built = true;
    return CSX_OK;
}

static int launch_exact_dpp(int BS, const Tree *trees, int32_t ntrees, const int32_t *nodes, const int32_t *perm, const int32_t *f_ptr,
                            const double *f_val, const int32_t *b_ptr, const double *dense_b, const double *diagk, const double *diagb,
                            double *B, int32_t nrhs, bool mix, const double *packed_lx = nullptr) {
    // packed_lx (with mix): L.x of a factor that is nothing but these equal blocks, block t at t BS (BS + 1) / 2 -- no programs needed
    hipStream_t s = ctx().stream;
    const int32_t chunks = (nrhs + 63) / 64;
    const int share = chunks % 4 == 0 ? 1 : chunks % 2 == 0 ? 2 : 4;
    const int64_t groups = ((int64_t)ntrees + share - 1) / share * (chunks / (4 / share));
    const dim3 grid((unsigned)groups);
    if (packed_lx) f_val = packed_lx;
#define CSX_DPP_X(BS_, SH, MX, PK)                                                                                                     \
    hipLaunchKernelGGL((k_cholsol_dense_exact_dpp<BS_, SH, MX, PK>), grid, dim3(256), 0, s, trees, ntrees, nodes, perm, f_ptr, f_val, \
                       b_ptr, dense_b, diagk, diagb, B, nrhs, chunks)
#define CSX_DPP_S(BS_, MX, PK)                      \
    if (share == 1) CSX_DPP_X(BS_, 1, MX, PK);      \
    else if (share == 2) CSX_DPP_X(BS_, 2, MX, PK); \
    else CSX_DPP_X(BS_, 4, MX, PK)
#define CSX_DPP_V(BS_)                                \
    if (mix && packed_lx) { CSX_DPP_S(BS_, 1, true); } \
    else if (mix) { CSX_DPP_S(BS_, 1, false); }        \
    else { CSX_DPP_S(BS_, 0, false); }
    switch (BS) {
        case 8: CSX_DPP_V(8); break;
        case 16: CSX_DPP_V(16); break;
        case 32: CSX_DPP_V(32); break;
        case 48: { CSX_DPP_S(48, 1, false); } break;      // (the padded size classes only: programs, the default mix)
        default: CSX_DPP_V(64); break;
    }
#undef CSX_DPP_V
#undef CSX_DPP_S
#undef CSX_DPP_X
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

static int cholsol_solve(CholPlan *P, double *B, int32_t nrhs) {
    hipStream_t s = ctx().stream;
    const int32_t n = P->n;
    if (n == 0 || nrhs == 0) return CSX_OK;
    if (P->lite && P->lite_cliques && !(P->relaxed && P->rag) && ctx().opt.cholsol_dense_blocks) {
        // the exact order on cliques of unequal sizes: size classes padded with the identity, the register-resident exact kernel
        CSX_TRY(cholsol_exact_classes_build(P));
        if (P->xc_built) {
            static const int kBS[5] = {8, 16, 32, 48, 64};
            for (int c = 0; c < 5; c++) {
                const CholPlan::ExactClass &X = P->xc[c];
                if (X.count > 0)
                    CSX_TRY(launch_exact_dpp(kBS[c], X.trees, X.count, X.nodes, nullptr, X.f_ptr, X.f_val, X.b_ptr, X.dense_b, X.diagk,
                                             X.diagb, B, nrhs, true));
            }
            return CSX_OK;
        }
    }
    if (P->lite) {
        if (P->relaxed && P->rag && ctx().opt.cholsol_dense_blocks) return ragged_solve(P->rag, P->tree_nodes, nullptr, false, 2, B, nrhs, P->n);
        if (!P->full) CSX_TRY(cholsol_plan(P->L, nullptr, &P->full));
        P->full->relaxed = false;          // (what `full` is for: the exact order, or the order the guard left)
        return cholsol_solve(P->full, B, nrhs);
    }
    const int32_t *Gp = nullptr, *Gi = nullptr;
    const double *Gx = nullptr, *Gd = nullptr;
    if (!P->clique) tri_gather_arrays(P->fwd, &Gp, &Gi, &Gx, &Gd);
    if (P->local && (Gd != nullptr || P->clique)) {
        // zero pivots were detected by the analysis; report like the reference (ZeroDivisionError)
        if (P->clique) {
            // a plan csx_cholsol_factor made holds the matrix-core operands only (written by the block kernel beside L.x); what the
            // substitution kernels read is cut out of L.x the first time one of them is asked for
            // ... unless the kernel asked for takes L.x as it is: the matrix cores (W tiles beside it), and -- round 5 -- the
            // default exact kernel, which copies the block's packed columns to LDS once for both passes
            const bool cores = P->dense_bs && P->relaxed && P->frag_f && ctx().opt.cholsol_dense_blocks;
            const int wantv = ctx().opt.cholsol_exact_variant;
            const bool lx_aligned = (reinterpret_cast<uintptr_t>(P->L->x) & 15) == 0;      // (the copies to LDS move 16 bytes a lane)
            const bool packed_exact = P->dense_bs && !P->relaxed && ctx().opt.cholsol_dense_blocks && !(wantv >= 1 && wantv <= 6 && wantv != 5) && lx_aligned;
            if (!cores && !packed_exact && !P->f_val) CSX_TRY(cholsol_plan_clique(P, P->dense_bs));
            if (P->clique_zero_pivot) return CSX_EZEROPIVOT;
            if (!ctx().opt.cholsol_dense_blocks) CSX_TRY(cholsol_clique_local(P));
        } else {
            int st = tri_solve_raw(P->fwd, B, 0, false);
            if (st != CSX_OK) return st;
        }
        // Forests of dense blocks: the default (exact) order runs the substitution kernel that keeps the reference's
        // operations and their order; the rounding-equal order the FMA / matrix-core kernels.
        if (P->dense_bs && !P->relaxed && ctx().opt.cholsol_dense_blocks) {
            // Variants ("cholsol.exact_variant"; measurements on G-spd, 5M rows, 128 right-hand sides, in
            // profiles/r03_ablation.md section 3): 0 / 5 = the L values by DPP row broadcast, one term in four by an LDS
            // broadcast read (the default for every block size: 4.6 - 4.8 / 3.0 / 2.2 / 2.0 ms at blocks of 64 / 32 / 16 / 8);
            // 6 = by DPP only; the LDS-broadcast forms they replaced: 1 = one fence per row / one right-hand side per
            // lane, 2 = the L values through a ring of registers, 3 = rows / two per lane, 4 = ring / two per lane (two per
            // lane only for blocks <= 32; at blocks of 32 these four compile to scratch).
            const int want = ctx().opt.cholsol_exact_variant;
            int variant = 5;
            if (want >= 1 && want <= 6) variant = want;
            if (variant >= 5) {
                // the L values by DPP row broadcast: one right-hand side per lane; waves that solve the same block share its
                // LDS copy (as many as divide the number of 64-wide chunks of right-hand sides)
                // (a plan on all the columns of L -- `clique` -- hands the kernel L.x itself)
                return launch_exact_dpp(P->dense_bs, P->trees, P->ntrees, P->tree_nodes, P->perm, P->f_ptr, P->f_val, P->b_ptr, P->dense_b,
                                        P->diagk, P->diagb, B, nrhs, variant == 5,
                                        P->clique && variant == 5 && (reinterpret_cast<uintptr_t>(P->L->x) & 15) == 0 ? P->L->x : nullptr);
            }
            if (P->dense_bs == 64 && variant > 2) variant -= 2;
            if (nrhs <= 64 && variant > 2) variant -= 2;
            const int R = variant > 2 ? 2 : 1;
            const bool ring = variant == 2 || variant == 4;
            const int32_t chunks = (nrhs + 64 * R - 1) / (64 * R);
            const int64_t tasks = (int64_t)P->ntrees * chunks;
            const dim3 grid((unsigned)((tasks + 3) / 4));
#define CSX_DENSE_X(BS, RR, RG)                                                                                                 \
    hipLaunchKernelGGL((k_cholsol_dense_exact<BS, RR, RG>), grid, dim3(256), 0, s, P->trees, P->ntrees, P->tree_nodes, P->perm, \
                       P->f_ptr, P->f_val, P->b_ptr, P->dense_b, P->diagk, P->diagb, B, nrhs, chunks)
#define CSX_DENSE_V(BS)                                \
    if (variant == 1) CSX_DENSE_X(BS, 1, false);       \
    else if (variant == 2) CSX_DENSE_X(BS, 1, true);   \
    else if (variant == 3) CSX_DENSE_X(BS, 2, false);  \
    else CSX_DENSE_X(BS, 2, true)
            switch (P->dense_bs) {
                case 8: CSX_DENSE_V(8); break;
                case 16: CSX_DENSE_V(16); break;
                case 32: CSX_DENSE_V(32); break;
                default:
                    if (variant == 1) CSX_DENSE_X(64, 1, false);
                    else CSX_DENSE_X(64, 1, true);
                    break;
            }
#undef CSX_DENSE_V
#undef CSX_DENSE_X
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
        if (P->dense_bs && P->relaxed && ctx().opt.cholsol_dense_blocks) {
            const int32_t chunks = (nrhs + 63) / 64;
            const int64_t tasks = (int64_t)P->ntrees * chunks;
            const dim3 grid((unsigned)((tasks + 3) / 4));
            if (P->frag_f) {
                // (equal dense blocks on ALL the columns of L -- `clique` -- lie in L.x as they do in a copy: block t at t bs (bs + 1) / 2)
                const double *lv = P->lcopy ? P->lcopy : P->L->x;
                const int share = chunks % 4 == 0 ? 1 : chunks % 2 == 0 ? 2 : 4;
                const dim3 g2((unsigned)(((int64_t)P->ntrees + share - 1) / share * (chunks / (4 / share))));
#define CSX_MF_X(NB_, SH) \
    hipLaunchKernelGGL((k_cholsol_mfma<NB_, SH>), g2, dim3(256), 0, s, P->trees, P->ntrees, P->tree_nodes, P->perm, lv, P->frag_f, B, nrhs, chunks)
#define CSX_MF(NB_)                       \
    if (share == 1) CSX_MF_X(NB_, 1);     \
    else if (share == 2) CSX_MF_X(NB_, 2); \
    else CSX_MF_X(NB_, 4)
                switch (P->dense_bs) {
                    case 16: CSX_MF(1); break;
                    case 32: CSX_MF(2); break;
                    default: CSX_MF(4); break;
                }
#undef CSX_MF
#undef CSX_MF_X
                CSX_LAUNCH_CHECK();
                return CSX_OK;
            }
#define CSX_DENSE(BS)                                                                                          \
    hipLaunchKernelGGL(k_cholsol_dense<BS>, grid, dim3(256), 0, s, P->trees, P->ntrees, P->tree_nodes, P->perm, \
                       P->f_ptr, P->f_val, P->b_ptr, P->dense_b, P->diagk, P->diagb, B, nrhs, chunks)
            switch (P->dense_bs) {
                case 8: CSX_DENSE(8); break;
                case 16: CSX_DENSE(16); break;
                case 32: CSX_DENSE(32); break;
                default: CSX_DENSE(64); break;
            }
#undef CSX_DENSE
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
        if (P->relaxed && P->rag && ctx().opt.cholsol_dense_blocks)    // trees of any shape, rounding-equal order: on the matrix cores
            return ragged_solve(P->rag, P->tree_nodes, P->perm, false, 2, B, nrhs, P->n);
        const size_t per_wave = (size_t)P->max_nodes * 64 * sizeof(double);
        const int waves = tile_waves_per_workgroup(per_wave, CH_WAVES);
        const int32_t chunks = (nrhs + 63) / 64;
        const int64_t tasks = (int64_t)P->ntrees * chunks;
        const size_t lds = per_wave * (size_t)waves;
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cholsol_local),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        if (!P->trees_by_size) CSX_TRY(trees_biggest_first(P->trees, P->ntrees, P->max_nodes, &P->trees_by_size));
        hipLaunchKernelGGL(k_cholsol_local, dim3((unsigned)((tasks + waves - 1) / waves)), dim3(64 * waves), lds, s,
                           P->trees_by_size, P->ntrees, P->tree_nodes, P->perm, P->f_ptr, P->f_idx, P->f_val, P->b_ptr, P->b_idx,
                           P->b_val, P->diagk, P->diagb, B, nrhs, chunks, P->max_nodes, waves);
        CSX_LAUNCH_CHECK();
        return CSX_OK;
    }
    double *X = B;
    if (P->perm) {
        const int64_t need = (int64_t)n * nrhs;
        if (P->scratch_len < need) {
            dfree(P->scratch);
            P->scratch = nullptr;
            P->scratch_len = 0;
            CSX_TRY(dalloc(&P->scratch, (size_t)need));
            P->scratch_len = need;
        }
        X = P->scratch;
        hipLaunchKernelGGL(k_perm_rows, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, s, P->perm, B, X, n, nrhs, 1);
    }
    if (P->relaxed && P->sn && ctx().opt.tri_supernodes && sn_usable(P->sn)) {
        CSX_TRY(tri_solve_raw(P->fwd, X, 0, false));          // a zero pivot found by the analysis: ZeroDivisionError
        // "tri.graph": 1 = always; 2 (the default) = when a solve is many launches -- more than 256: a natural-order grid
        // factor is 5 624 -- and this block has been the block of the two solves before this one as well (round 4 captured on
        // the SECOND solve of a block: 9 ms of capture + instantiate in front of a 1.8 ms solve on bcsstk16, which a caller
        // who solves a block twice never gets back); 0 = never.  The replay takes the host 13 - 60 us instead of 0.35 -
        // 17 ms; the device time is the same.  csx_cholsol_graph_info reports the captures and what the last one cost.
        bool graph = ctx().opt.tri_graph == 1;
        if (ctx().opt.tri_graph == 2) {
            int32_t nsn = 0, steps = 0, maxw = 0;
            sn_info(P->sn, &nsn, &steps, &maxw);
            P->same_block_runs = (P->last_X == X && P->last_nrhs == nrhs) ? P->same_block_runs + 1 : 0;
            graph = 6 * (int64_t)steps > 256 && (P->same_block_runs >= 2 || (P->g_exec && P->g_X == X && P->g_nrhs == nrhs));
        }
        P->last_X = X;
        P->last_nrhs = nrhs;
        if (graph) {
            // the two sweeps' launches as one graph, re-used while the block of right-hand sides stays where it is
            CSX_TRY(sn_prepare(P->sn, nrhs));              // (may move the work space: the captured launches hold its address)
            if (!(P->g_exec && P->g_X == X && P->g_nrhs == nrhs && P->g_gen == sn_generation(P->sn) &&
                  P->g_opt == ctx().opt.tri_supernodes)) {
                if (P->g_exec) (void)hipGraphExecDestroy(P->g_exec);
                P->g_exec = nullptr;
                hipGraph_t graph = nullptr;
                const auto t_cap = std::chrono::steady_clock::now();
                CSX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int st = sn_solve(P->sn, true, Gp, Gi, Gx, Gd, P->L, X, nrhs);
                if (st == CSX_OK) st = sn_solve(P->sn, false, Gp, Gi, Gx, Gd, P->L, X, nrhs);
                const hipError_t ec = hipStreamEndCapture(s, &graph);
                if (st != CSX_OK || ec != hipSuccess) {
                    if (graph) (void)hipGraphDestroy(graph);
                    set_error("cholsol: capturing the supernodal solve failed");
                    return st != CSX_OK ? st : CSX_ERUNTIME;
                }
                const hipError_t ei = hipGraphInstantiate(&P->g_exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ei != hipSuccess) {
                    P->g_exec = nullptr;
                    set_error("cholsol: hipGraphInstantiate: %s", hipGetErrorString(ei));
                    return CSX_ERUNTIME;
                }
                P->g_X = X;
                P->g_nrhs = nrhs;
                P->g_gen = sn_generation(P->sn);
                P->g_opt = ctx().opt.tri_supernodes;
                P->g_captures++;
                P->g_capture_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cap).count();
            }
            CSX_HIP(hipGraphLaunch(P->g_exec, s));
        } else {
            CSX_TRY(sn_solve(P->sn, true, Gp, Gi, Gx, Gd, P->L, X, nrhs));
            CSX_TRY(sn_solve(P->sn, false, Gp, Gi, Gx, Gd, P->L, X, nrhs));
        }
    } else {
        CSX_TRY(tri_solve_raw(P->fwd, X, nrhs, P->relaxed));
        CSX_TRY(tri_solve_raw(P->bwd, X, nrhs, P->relaxed));
    }
    if (P->perm) {
        const int64_t need = (int64_t)n * nrhs;
        hipLaunchKernelGGL(k_perm_rows, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, s, P->perm, X, B, n, nrhs, 0);
    }
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_chol(csx_handle_t hA, const int32_t *parent, const int32_t *cp, const int32_t *pinv,
                        csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !A->x || A->m != A->n || !parent || !cp || !out) return CSX_EINVAL;
    if (cp[0] != 0 || cp[A->n] < A->n) return CSX_EINVAL;
    Csc *L = new Csc();
    int st = chol_device(A, parent, cp, pinv, L);
    if (st != CSX_OK) {
        free_csc(L);
        return st;
    }
    *out = put(K_CSC, L);
    return CSX_OK;
}

namespace csx {
// cs_cholsol's factor sequence, natural order (csparse.py:636-639: S = cs_schol(0, A); N = cs_chol(A, S)), and the solve plan of
// csparse.py:640-643 in ONE call with S never leaving the device.  A forest of cliques / of small sparse trees on consecutive columns:
// one wait for the analysis (clique_forest), the block kernel, one wait for its flags; for EQUAL dense blocks of 16 / 32 / 64 columns
// in the rounding-equal order the block kernel writes the matrix-core solve's operands itself (CliqueEmit) and the plan is complete
// when it ends.  Anything else: csx_schol + csx_chol + csx_cholsol_plan behind this one entry, S in host vectors the caller never sees.
static int g_factor_path = -1;
static double g_factor_ms[3] = {0.0, 0.0, 0.0};   // analysis / numeric kernel (HIP events) / whole call (host clock)

static int cholsol_factor_device(csx_handle_t hA, Csc *A, bool exact, Csc *L, CholPlan **Pout) {   // (A->clique may be replaced)
    hipStream_t s = ctx().stream;
    const int32_t n = A->n;
    const auto t_call = std::chrono::steady_clock::now();
    g_factor_path = -1;
    g_factor_ms[0] = g_factor_ms[1] = g_factor_ms[2] = 0.0;
    auto since = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    if (n > 0 && A->nnz > 0 && ctx().opt.chol_clique && ctx().opt.chol_dense_trees) {
        CliqueForest F;
        bool ok = false;
        const bool cached = A->clique != nullptr && (!A->clique->sparse || ctx().opt.chol_forest);
        if (cached) {
            F = *A->clique;
            ok = true;
        } else {
            CSX_TRY(clique_forest(A, &F, &ok));
        }
        g_factor_ms[0] = since(t_call);
        struct Release {   // the forest's arrays, unless L took them
            CliqueForest *F;
            bool on;
            ~Release() {
                if (on) free_clique(F);
            }
        } release{&F, ok && !cached};
        if (ok && F.ascending && F.max_bs <= CLIQUE_MAX_BLOCK) {
            const int32_t bs = F.max_bs;
            const bool emit = !exact && !F.sparse && F.min_bs == bs && (bs == 16 || bs == 32 || bs == 64) && ctx().opt.cholsol_dense_blocks;
            L->m = L->n = n;
            L->nnz = (int32_t)F.lnz;
            L->owns = true;
            // L gets a copy of the forest's column pointers; the finding itself (tree, counts, block list: pattern only) stays on the
            // matrix for the next factorisation of it -- a refactorisation loop pays for the analysis once, like csx_schol followed by
            // many csx_chol (csx_csc_invalidate drops it with every other cached plan)
            CSX_TRY(dalloc(&L->p, (size_t)n + 1));
            CSX_HIP(hipMemcpyAsync(L->p, F.cp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            if (!cached) {
                free_clique_cache(A->clique);
                A->clique = new CliqueForest(F);
                release.on = false;   // the matrix owns the arrays now
            }
            CSX_TRY(dalloc(&L->x, (size_t)L->nnz));
            // (L.i: allocated below, once it is known whether the kernel writes it -- with the emission it does not: rows j, j + 1, ...
            // in every column of a clique, Csc::rows_pending)
            CholPlan *P = nullptr;
            CliqueEmit em;
            DevScope tmp;
            int *d_flags = nullptr;          // [0] first column with a non-positive pivot, [2..3] the guard's measure
            CSX_TRY(tmp.alloc(&d_flags, 4));
            CSX_HIP(hipMemsetAsync(d_flags, 0x7f, sizeof(int), s));
            CSX_HIP(hipMemsetAsync(d_flags + 2, 0, 2 * sizeof(int), s));
            // cliques of UNEQUAL sizes, rounding-equal order: the block kernel writes the matrix-core operands of csx_trimfma.hip's
            // size classes itself (the class-ordered list, descriptors and fragment storage are made first: the kernel needs to know
            // where each block's fragments go)
            const bool emit_ragged = !emit && !exact && !F.sparse && ctx().opt.cholsol_dense_blocks &&
                                     !(F.min_bs == bs && (bs == 8 || bs == 16 || bs == 32 || bs == 64));
            RaggedMfma *Rg = nullptr;
            int64_t *frag_off = nullptr;
            struct RagGuard {
                RaggedMfma *&R;
                int64_t *&off;
                ~RagGuard() {
                    ragged_free(R);
                    dfree(off);
                }
            } rag_guard{Rg, frag_off};
            if (emit_ragged) {
                P = new CholPlan();
                *Pout = P;
                P->n = n;
                P->L = L;
                P->lite = true;
                P->lite_cliques = true;
                P->local = true;
                P->relaxed = true;
                P->ntrees = F.nblocks;
                P->max_nodes = bs;
                CSX_TRY(dalloc(&P->trees, (size_t)F.nblocks));
                CSX_TRY(dalloc(&P->tree_nodes, (size_t)n));
                CSX_TRY(ragged_blocks(F.start, F.nblocks, n, P->trees, P->tree_nodes));
                CSX_TRY(ragged_prepare_emit(P->trees, P->ntrees, bs, P->tree_nodes, &Rg, &frag_off));
                if (Rg) {
                    em.frag = Rg->frag;
                    em.frag_off = frag_off;
                    em.cond_bits = (unsigned long long *)(d_flags + 2);
                    em.list = Rg->list;
                    for (int c = 0; c <= RAG_CLASSES; c++) em.cls_start[c] = Rg->cls_start[c];
                }
            }
            const bool emit_any = emit || (emit_ragged && Rg);
            if (!emit_any) CSX_TRY(dalloc(&L->i, (size_t)L->nnz));
            if (emit) {
                P = new CholPlan();
                *Pout = P;            // (the caller frees it on any error)
                P->n = n;
                P->L = L;
                CSX_TRY(dalloc(&P->frag_f, (size_t)F.nblocks * (size_t)(bs / 16) * 256));   // the W tiles; the rest is L.x
                CSX_TRY(dalloc(&P->trees, (size_t)F.nblocks));
                CSX_TRY(dalloc(&P->tree_nodes, (size_t)n));
                em.frag = P->frag_f;
                em.cond_bits = (unsigned long long *)(d_flags + 2);
                em.trees = P->trees;
                em.tree_nodes = P->tree_nodes;
            }
            hipEvent_t ev_a = nullptr, ev_b = nullptr;
            if (hipEventCreate(&ev_a) != hipSuccess || hipEventCreate(&ev_b) != hipSuccess) {
                if (ev_a) (void)hipEventDestroy(ev_a);
                return CSX_ERUNTIME;
            }
            (void)hipEventRecord(ev_a, s);
            int st = chol_clique_numeric(A, F, L, d_flags, emit_any ? &em : nullptr, !ctx().opt.chol_exact);
            (void)hipEventRecord(ev_b, s);
            int h[4] = {0, 0, 0, 0};
            if (hipMemcpyAsync(h, d_flags, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess) st = st == CSX_OK ? CSX_ERUNTIME : st;
            if (hipStreamSynchronize(s) != hipSuccess) st = st == CSX_OK ? CSX_ERUNTIME : st;   // (also on failure: L's arrays go back to the pool)
            float ms = 0.0f;
            if (st == CSX_OK && hipEventElapsedTime(&ms, ev_a, ev_b) == hipSuccess) g_factor_ms[1] = ms;
            (void)hipEventDestroy(ev_a);
            (void)hipEventDestroy(ev_b);
            CSX_TRY(st);
            if (h[0] != 0x7f7f7f7f) return CSX_ENOTSPD;
            g_chol_path = F.sparse ? 2 : 1;
            g_chol_numeric_ms = g_factor_ms[1];
            if (emit) {
                L->rows_pending = true;       // rows j, j + 1, ... in every column: made from L.p when a handle to L is resolved
                double growth = 0.0;
                std::memcpy(&growth, h + 2, sizeof growth);
                P->clique = true;
                P->clique_zero_pivot = false;  // every pivot is a square root of a positive number
                P->ntrees = F.nblocks;
                P->max_nodes = bs;
                P->local = true;
                P->dense_bs = bs;
                P->relaxed = true;
                P->mfma_tried = true;
                P->mfma_growth = growth;
                if (!(growth <= MFMA_GROWTH_LIMIT)) {     // the guard refuses the explicit inverses (a NaN too): substitution, from L.x
                    dfree(P->frag_f);
                    P->frag_f = nullptr;
                }
                g_factor_path = 3;
            } else if (emit_ragged) {
                // the plan was made before the kernel ran (above); its matrix-core operands are in place
                if (Rg) {
                    L->rows_pending = L->i == nullptr;
                    double growth = 0.0;
                    std::memcpy(&growth, h + 2, sizeof growth);
                    Rg->growth = growth;
                    P->mfma_growth = growth;
                    P->rag_tried = true;
                    if (growth <= MFMA_GROWTH_LIMIT) {       // (max|L| max|W| of a block, the equal-block path's measure; a NaN fails)
                        P->rag = Rg;
                        Rg = nullptr;
                    }
                } else if (!L->i) {
                    L->rows_pending = true;
                }
                g_factor_path = 1;
            } else if (ctx().opt.cholsol_dense_blocks && (!exact || !F.sparse) &&
                       !(!F.sparse && F.min_bs == bs && (bs == 8 || bs == 16 || bs == 32 || bs == 64))) {
                // a forest of UNEQUAL cliques (either order) or of small sparse trees (rounding-equal order): the plan is the block
                // list from the forest's starts; the matrix-core operands come straight from L's columns (now, or when the order is
                // switched), the exact order of cliques runs on padded size classes cut out of L.x at the first such solve
                // (cholsol_exact_classes_build) -- no general plan unless a solve needs one
                P = new CholPlan();
                *Pout = P;
                P->n = n;
                P->L = L;
                P->lite = true;
                P->lite_cliques = !F.sparse;
                P->local = true;
                P->relaxed = !exact;
                P->ntrees = F.nblocks;
                P->max_nodes = bs;
                CSX_TRY(dalloc(&P->trees, (size_t)F.nblocks));
                CSX_TRY(dalloc(&P->tree_nodes, (size_t)n));
                CSX_TRY(ragged_blocks(F.start, F.nblocks, n, P->trees, P->tree_nodes));
                if (!exact) CSX_TRY(cholsol_build_ragged(P));
                g_factor_path = F.sparse ? 2 : 1;
            } else {
                if (!F.sparse && F.min_bs == bs && (bs == 8 || bs == 16 || bs == 32 || bs == 64) && ctx().opt.cholsol_dense_blocks) {
                    // equal dense blocks (the forest's record says so: no k_clique_factor_shape over L.i): the programs straight from L.x
                    P = new CholPlan();
                    *Pout = P;
                    P->n = n;
                    P->L = L;
                    if (exact) {
                        // every solve in the reference's order: the default exact kernel reads L.x itself (k_cholsol_dense_exact_dpp<PACKED>),
                        // so the plan is the block list; programs are cut out of L.x only if another kernel is asked for (cholsol_solve)
                        CSX_TRY(dalloc(&P->trees, (size_t)F.nblocks));
                        CSX_TRY(dalloc(&P->tree_nodes, (size_t)n));
                        CSX_TRY(ragged_blocks(F.start, F.nblocks, n, P->trees, P->tree_nodes));
                        P->clique = true;
                        P->clique_zero_pivot = false;      // every pivot is a square root of a positive number
                        P->ntrees = F.nblocks;
                        P->max_nodes = bs;
                        P->local = true;
                        P->dense_bs = bs;
                    } else {
                        CSX_TRY(cholsol_plan_clique(P, bs));
                    }
                } else {
                    CSX_TRY(cholsol_plan(L, nullptr, Pout));
                    P = *Pout;
                }
                if (!exact) {
                    P->relaxed = true;
                    CSX_TRY(cholsol_build_mfma(P));
                    CSX_TRY(cholsol_build_sn(P));
                }
                g_factor_path = F.sparse ? 2 : 1;
            }
            g_factor_ms[2] = since(t_call);
            return CSX_OK;
        }
    }
    // the general path: the three calls behind one entry
    std::vector<int32_t> parent((size_t)std::max(n, 1)), cp((size_t)n + 1);
    CSX_TRY(csx_schol(hA, parent.data(), cp.data()));
    g_factor_ms[0] = since(t_call);
    CSX_TRY(chol_device(A, parent.data(), cp.data(), nullptr, L));
    g_factor_ms[1] = g_chol_numeric_ms;
    CSX_TRY(cholsol_plan(L, nullptr, Pout));
    if (!exact) {
        (*Pout)->relaxed = true;
        CSX_TRY(cholsol_build_mfma(*Pout));
        CSX_TRY(cholsol_build_sn(*Pout));
    }
    g_factor_path = 0;
    g_factor_ms[2] = since(t_call);
    return CSX_OK;
}
}  // namespace csx

extern "C" int csx_cholsol_factor(csx_handle_t hA, int exact, csx_handle_t *outL, csx_handle_t *outPlan) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !A->x || A->m != A->n || !outL || !outPlan) return CSX_EINVAL;
    Csc *L = new Csc();
    CholPlan *P = nullptr;
    const int st = cholsol_factor_device(hA, A, exact != 0, L, &P);
    if (st != CSX_OK) {
        free_cholplan(P);
        free_csc(L);
        return st;
    }
    *outL = put(K_CSC, L);
    *outPlan = put(K_CHOLPLAN, P);
    return CSX_OK;
}

extern "C" int csx_cholsol_factor_info(int32_t *path, double *analysis_ms, double *numeric_ms, double *call_ms) {
    if (g_factor_path < 0) return CSX_EINVAL;   // no csx_cholsol_factor has completed yet
    if (path) *path = g_factor_path;
    if (analysis_ms) *analysis_ms = g_factor_ms[0];
    if (numeric_ms) *numeric_ms = g_factor_ms[1];
    if (call_ms) *call_ms = g_factor_ms[2];
    return CSX_OK;
}

extern "C" int csx_chol_info(int32_t *path, double *numeric_ms) {
    if (g_chol_path < 0) return CSX_EINVAL;   // no csx_chol has completed yet
    if (path) *path = g_chol_path;
    if (numeric_ms) *numeric_ms = g_chol_numeric_ms;
    return CSX_OK;
}

extern "C" int csx_cholsol_plan(csx_handle_t hL, const int32_t *pinv, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *L = csc(hL);
    if (!L || !L->x || L->m != L->n || !out) return CSX_EINVAL;
    CholPlan *P = nullptr;
    int st = cholsol_plan(L, pinv, &P);
    if (st != CSX_OK) {
        free_cholplan(P);
        return st;
    }
    *out = put(K_CHOLPLAN, P);
    return CSX_OK;
}

extern "C" int csx_cholsol_info(csx_handle_t h, int32_t *local, int32_t *ntrees, int32_t *max_nodes) {
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    if (!P) return CSX_EINVAL;
    // 0 level-scheduled, 1 fused in LDS, 2 dense blocks (substitution), 3 dense blocks on the matrix cores, 4 supernodal schedule,
    // 5 small trees of any shape made dense by size class on the matrix cores (csx_trimfma.hip)
    if (P->lite && P->lite_cliques && !(P->relaxed && P->rag) && ctx().opt.cholsol_dense_blocks) {
        if (local) *local = 2;               // dense-block substitution (padded size classes), the exact order
        if (ntrees) *ntrees = P->ntrees;
        if (max_nodes) *max_nodes = P->max_nodes;
        return CSX_OK;
    }
    if (P->lite && !(P->relaxed && P->rag)) {   // the general plan answers (made now if it has to be)
        if (!P->full) CSX_TRY(cholsol_plan(P->L, nullptr, &P->full));
        P->full->relaxed = false;
        P = P->full;
    }
    if (local) *local = P->local ? (P->dense_bs ? (P->relaxed && P->frag_f ? 3 : 2) : (P->relaxed && P->rag ? 5 : 1)) : (P->relaxed && P->sn && sn_usable(P->sn) ? 4 : 0);
    if (ntrees) *ntrees = P->ntrees;
    if (max_nodes) *max_nodes = P->max_nodes;
    return CSX_OK;
}

extern "C" int csx_cholsol_set_order(csx_handle_t h, int exact) {
    CSX_TRY(require_ready());
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    if (!P) return CSX_EINVAL;
    P->relaxed = exact == 0;
    if (P->relaxed) CSX_TRY(cholsol_build_mfma(P));
    if (P->relaxed) CSX_TRY(cholsol_build_sn(P));
    return CSX_OK;
}

extern "C" int csx_cholsol_growth(csx_handle_t h, double *growth) {
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    if (!P || !growth) return CSX_EINVAL;
    *growth = P->mfma_growth;
    return CSX_OK;
}

extern "C" int csx_cholsol_sn_info(csx_handle_t h, int32_t *supernodes, int32_t *steps, int32_t *max_width, int32_t *matrix_cores,
                                   double *growth) {
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    if (!P) return CSX_EINVAL;
    int32_t a = 0, b = 0, c = 0, mc = 0;
    double g = 0.0;
    if (P->sn) {
        sn_info(P->sn, &a, &b, &c);
        sn_info2(P->sn, &mc, &g);
    }
    if (supernodes) *supernodes = a;
    if (steps) *steps = b;
    if (max_width) *max_width = c;
    if (matrix_cores) *matrix_cores = mc;
    if (growth) *growth = g;
    return CSX_OK;
}

// "tri.host_chains" (opt-in): the solve phase of cs_cholsol (csparse.py:640-643) for ONE host right-hand side when L is a chain:
// x = P b, L \ x, L' \ x, b = P' x with the reference's loops on the host (csx_tri_solve_list's rule, applied to both sweeps)
extern "C" int csx_cholsol_solve_list(csx_handle_t h, double *b, int *taken) {
    CSX_TRY(require_ready());
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    if (!P || !b || !taken) return CSX_EINVAL;
    *taken = 0;
    if (!ctx().opt.tri_host_chains || P->local || P->lite || !P->fwd || !P->bwd || P->n == 0) return CSX_OK;
    const int32_t n = P->n;
    std::vector<double> x((size_t)n);
    std::vector<int32_t> perm;
    if (P->perm) {
        perm.resize((size_t)n);
        CSX_HIP(hipMemcpyAsync(perm.data(), P->perm, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx().stream));
        CSX_HIP(hipStreamSynchronize(ctx().stream));
        for (int32_t j = 0; j < n; j++) x[(size_t)j] = b[perm[(size_t)j]];      // x = P b  (cs_ipvec, csparse.py:640)
    } else {
        std::memcpy(x.data(), b, (size_t)n * sizeof(double));
    }
    bool t1 = false, t2 = false;
    CSX_TRY(tri_solve_host_raw(P->fwd, x.data(), &t1));
    if (!t1) return CSX_OK;                       // no chain: nothing was changed, the caller goes to the device
    CSX_TRY(tri_solve_host_raw(P->bwd, x.data(), &t2));
    if (!t2) return CSX_ERUNTIME;                 // (both sweeps share one level structure: cannot happen)
    if (P->perm) {
        for (int32_t j = 0; j < n; j++) b[perm[(size_t)j]] = x[(size_t)j];      // b = P' x  (cs_pvec, csparse.py:643)
    } else {
        std::memcpy(b, x.data(), (size_t)n * sizeof(double));
    }
    *taken = 1;
    return CSX_OK;
}

extern "C" int csx_cholsol_graph_info(csx_handle_t h, int32_t *captures, double *last_capture_ms) {
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    if (!P) return CSX_EINVAL;
    if (captures) *captures = P->g_captures;
    if (last_capture_ms) *last_capture_ms = P->g_capture_ms;
    return CSX_OK;
}

extern "C" int csx_cholsol_solve(csx_handle_t h, csx_handle_t hB, int32_t nrhs) {
    CSX_TRY(require_ready());
    CholPlan *P = (CholPlan *)get(h, K_CHOLPLAN);
    Vec *B = vec(hB);
    if (!P || !B || nrhs < 0 || B->len < (int64_t)P->n * nrhs) return CSX_EINVAL;
    return cholsol_solve(P, (double *)B->d, nrhs);
}
