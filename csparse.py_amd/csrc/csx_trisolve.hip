// cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve (csparse.py:1330-1345, :1348-1365,
// :2368-2385, :2460-2475) and cs_ipvec / cs_pvec (:1264-1277, :1779-1792) on the
// device, for n-by-nrhs row-major blocks of right-hand sides.
//
// Every solve is executed in GATHER form, one unknown at a time:
//     x[r] = ( x[r] - sum_q val[q] * x[idx[q]] ) / diag[r]
// with the terms q in exactly the order in which the reference updates x[r]:
//   L  (forward, column push):  ascending source column, then storage order
//        -> stable transpose of L with the first entry of every column removed
//   U  (backward, column push): descending source column, then storage order
//        -> stable transpose of U with the last entry of every column removed
//           and the columns taken in reverse
//   LT / UT (column gather):    the column itself, storage order, no copy
// Each (unknown, right-hand side) pair is one lane running that loop with the
// multiply and the subtract rounded separately, so the result is bit-identical
// to the reference for every right-hand side; lanes of a wavefront hold
// neighbouring right-hand sides (row-major X => coalesced).
//
// Scheduling: level sets (unknowns whose inputs are all in earlier levels).
// Wide levels are one launch each (measured on W, 80 dependent launches, 0.52 ms: replaying the chain
// from a hipGraph changed nothing, so the cost is device-side dependency between dispatches, not host
// launch calls; and one cooperative launch with a grid barrier per level was SLOWER, 0.79 ms at 10
// workgroups and 5.4 ms at 512: an agent-scope release/acquire across the 8 XCD L2s costs more than a
// dispatch; and for pure chains -- bcsstk16's factor, 4810 one-row levels, 7.3 us per level here -- a
// workgroup whose waves take the levels in rotation and prefetch their rows' terms into registers,
// broadcasting them with v_readlane, was slower too: 10.8 us per level, the per-term scalar overhead
// outweighs the saved round trip); runs of narrow levels are executed by ONE
// workgroup that walks them with a workgroup barrier in between, so a chain
// (thousands of one-row levels) costs one launch, not thousands.
//
// Structures the reference would mis-solve (an entry above the diagonal in L,
// ...) are run by a literal one-lane-per-right-hand-side transcription of the
// reference loop, so the answer matches even then.
#include <algorithm>
#include <cstdlib>

#include "csx_internal.h"
#include "csx_sweep.h"
#include "csx_trimfma.h"

namespace csx {

struct Segment {
    int32_t l0, l1;   // levels [l0, l1)
    bool one_wg;
};

struct TriPlan {
    int kind = 0;
    int32_t n = 0, nlevels = 0;
    bool sequential = false, zero_pivot = false;
    bool scheduled = false, forward = true;  // level sets are computed on first use
    int32_t gnnz = 0;
    bool owns_g = false;
    int skip_first = 0, skip_last = 0;
    int32_t *ptr = nullptr, *idx = nullptr;
    double *val = nullptr;
    double *diag = nullptr;
    int32_t *order = nullptr, *level_ptr = nullptr;  // device
    // blocked chain walker (k_tri_chain): per position in `order` the number of leading terms that come from
    // before the row's 16-row block, and per term the block slot of its source (-1: outside the block)
    int32_t *npre = nullptr, *nin = nullptr;   // nin: number of in-block sources of the row (relaxed order)
    int8_t *tslot = nullptr;
    bool chain_ok = false;
    std::vector<int32_t> level_ptr_h;
    std::vector<Segment> segs;
    const int32_t *Tp = nullptr, *Ti = nullptr;  // the analysed matrix (not owned)
    const double *Tx = nullptr;
    // component path (k_tri_local): the dependency graph falls into many small connected components, each
    // solved by one wave with its X tile in LDS.  Built on the device (analyse_components).
    bool comp_tried = false, comp_ok = false;
    int32_t ncomp = 0, comp_max = 0;
    Tree *comps = nullptr;           // [ncomp] (first, count) into comp_nodes
    Tree *comps_by_size = nullptr;   // the same list, biggest first (stable): what the many-right-hand-side sweeps launch by
    int32_t *comp_nodes = nullptr;   // [n] rows grouped by component, ascending inside one
    int32_t *prog_ptr = nullptr, *prog_idx = nullptr;   // per sweep position: terms (local row * 64, value)
    double *prog_val = nullptr, *prog_diag = nullptr;
    int col_state = 0;               // k_tri_columns: 0 not examined, 1 usable, 2 not (duplicate rows in a column)
    // windowed column kernels (k_tri_wcolumns / k_tri_wcolchain): band half-width of T (max |row - column|, -1: not
    // measured yet) and the number of columns with an entry next to the diagonal (a chain link)
    int32_t band = -1, links = 0;
    TriPlan *mate = nullptr;         // plan of the transposed solve on the same matrix (cholsol: L for L'), not owned
    std::vector<int32_t> level_hint; // levels proposed by the caller (cholsol: elimination-tree heights / depths); verified before use
    // two-phase runs of narrow levels (k_tri_run_prefix64): level of every row, and per row where its chain resumes
    int32_t *level_of = nullptr, *resume = nullptr;
    int32_t *cptr = nullptr, *cidx = nullptr;   // push kinds (L, U): per sweep position the COLUMN's entries
    double *cval = nullptr, *cdiag = nullptr;
    int32_t push_terms = 0;          // most terms of one component (0: no push program)
    int few_cpw = 0;                 // components per wave of the all-in-LDS kernel (0: its tiles do not fit)
    int32_t few_rows = 0, few_terms = 0;
    // rounding-equal order (csx_tri_set_order(plan, 0)): the components made dense, bucketed by size class and solved on the
    // matrix cores (csx_trimfma.hip) -- components of at most 80 rows whose diagonal tiles pass the guard; the exact order and
    // every other plan shape are untouched
    // "tri.host_chains" (opt-in): a host copy of the analysed matrix for ONE host right-hand side on a chain-like factor
    std::vector<int32_t> hTp, hTi;
    std::vector<double> hTx;
    bool rounding_equal = false, rag_tried = false;
    RaggedMfma *rag = nullptr;
    double rag_growth = 0.0;
};

void free_triplan(TriPlan *t) {
    if (!t) return;
    if (t->owns_g) {
        dfree(t->ptr);
        dfree(t->idx);
        dfree(t->val);
    }
    dfree(t->diag);
    dfree(t->order);
    dfree(t->level_ptr);
    dfree(t->npre);
    dfree(t->nin);
    dfree(t->tslot);
    dfree(t->level_of);
    dfree(t->resume);
    dfree(t->comps);
    dfree(t->comps_by_size);
    dfree(t->comp_nodes);
    dfree(t->prog_ptr);
    dfree(t->prog_idx);
    dfree(t->prog_val);
    dfree(t->prog_diag);
    dfree(t->cptr);
    dfree(t->cidx);
    dfree(t->cval);
    dfree(t->cdiag);
    ragged_free(t->rag);
    delete t;
}

constexpr int NARROW = 512;   // (rows in level) * nrhs at or below this: level joins a one-workgroup run

// ---- building the gather layout ----------------------------------------------
__global__ __launch_bounds__(256) void k_check_columns(int32_t n, const int32_t *Tp, int *bad) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && Tp[j + 1] <= Tp[j]) *bad = 1;
}

// out = T without the first entry of each column (same column order); G lanes to a column (4 where columns are short: a wave
// per column of three entries is 61 idle lanes, 1.05 ms at 4.8M columns)
template <int G>
__global__ __launch_bounds__(256) void k_strip_first(int32_t n, const int32_t *Tp, const int32_t *Ti, const double *Tx,
                                                     int32_t *op, int32_t *oi, double *ox, double *diag) {
    const int lane = threadIdx.x & (G - 1);
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / G;
    for (int64_t j = wave; j <= n; j += nwaves) {
        if (j == n) {
            if (lane == 0) op[n] = Tp[n] - n;
            continue;
        }
        const int32_t b = Tp[j], e = Tp[j + 1], ob = b - (int32_t)j;
        if (lane == 0) {
            op[j] = ob;
            diag[j] = Tx[b];
        }
        for (int32_t p = b + 1 + lane; p < e; p += G) {
            oi[ob + (p - b - 1)] = Ti[p];
            ox[ob + (p - b - 1)] = Tx[p];
        }
    }
}

// out column c = column n-1-c of T without its last entry
__global__ __launch_bounds__(256) void k_strip_last_reverse(int32_t n, const int32_t *Tp, const int32_t *Ti,
                                                            const double *Tx, int32_t *op, int32_t *oi, double *ox,
                                                            double *diag) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int32_t nnz = Tp[n];
    for (int64_t c = wave; c <= n; c += nwaves) {
        const int32_t ob = nnz - Tp[n - c] - (int32_t)c;
        if (c == n) {
            if (lane == 0) op[n] = ob;
            continue;
        }
        const int32_t j = n - 1 - (int32_t)c, b = Tp[j], e = Tp[j + 1];
        if (lane == 0) {
            op[c] = ob;
            diag[j] = Tx[e - 1];
        }
        for (int32_t p = b + lane; p < e - 1; p += 64) {
            oi[ob + (p - b)] = Ti[p];
            ox[ob + (p - b)] = Tx[p];
        }
    }
}

__global__ __launch_bounds__(256) void k_diag_direct(int32_t n, const int32_t *Tp, const double *Tx, int last,
                                                     double *diag) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) diag[j] = last ? Tx[Tp[j + 1] - 1] : Tx[Tp[j]];
}

__global__ __launch_bounds__(256) void k_reverse_index(int64_t count, int32_t n, int32_t *idx) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < count) idx[q] = n - 1 - idx[q];
}

__global__ __launch_bounds__(256) void k_any_zero(int32_t n, const double *diag, int *flag) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && diag[j] == 0.0) *flag = 1;
}

// ---- solve kernels ---------------------------------------------------------------
#pragma clang fp contract(off)

__device__ __forceinline__ void solve_one(int32_t row, int r, int nrhs, const int32_t *__restrict__ ptr,
                                          const int32_t *__restrict__ idx, const double *__restrict__ val,
                                          const double *__restrict__ diag, int skip_first, int skip_last, double *X) {
    const int32_t b = ptr[row] + skip_first, e = ptr[row + 1] - skip_last;
    double acc = X[(int64_t)row * nrhs + r];
    // Terms are SUBTRACTED strictly in order (that is the parity contract), but their operands are
    // fetched eight at a time: one round trip for eight (index, value) pairs, one for the eight x's,
    // instead of two dependent round trips per term.  Slots past the row end re-read its last term
    // (a valid address) and are skipped in the arithmetic.
    constexpr int TB = 8;
    int32_t c[TB];
    double v[TB];
#pragma unroll
    for (int u = 0; u < TB; u++) {
        const int32_t q = b + u < e ? b + u : (e > b ? e - 1 : 0);
        c[u] = e > b ? idx[q] : 0;
        v[u] = e > b ? val[q] : 0.0;
    }
    for (int32_t q0 = b; q0 < e; q0 += TB) {
        double xv[TB];
#pragma unroll
        for (int u = 0; u < TB; u++) xv[u] = X[(int64_t)c[u] * nrhs + r];
        // the next batch's (index, value) pairs are requested behind the gathers and arrive while those are awaited
        int32_t cn[TB];
        double vn[TB];
#pragma unroll
        for (int u = 0; u < TB; u++) {
            const int32_t q = q0 + TB + u < e ? q0 + TB + u : e - 1;
            cn[u] = idx[q];
            vn[u] = val[q];
        }
#pragma unroll
        for (int u = 0; u < TB; u++) {
            const double t = v[u] * xv[u];
            acc = q0 + u < e ? acc - t : acc;
        }
#pragma unroll
        for (int u = 0; u < TB; u++) {
            c[u] = cn[u];
            v[u] = vn[u];
        }
    }
    X[(int64_t)row * nrhs + r] = acc / diag[row];
}

__global__ __launch_bounds__(256) void k_tri_level(const int32_t *__restrict__ order, int32_t first, int32_t count,
                                                   const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                   const double *__restrict__ val, const double *__restrict__ diag,
                                                   int skip_first, int skip_last, double *X, int nrhs) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)count * nrhs) return;
    solve_one(order[first + t / nrhs], (int)(t % nrhs), nrhs, ptr, idx, val, diag, skip_first, skip_last, X);
}

__global__ __launch_bounds__(1024) void k_tri_levels_one_wg(const int32_t *__restrict__ order,
                                                            const int32_t *__restrict__ level_ptr, int32_t l0,
                                                            int32_t l1, const int32_t *__restrict__ ptr,
                                                            const int32_t *__restrict__ idx,
                                                            const double *__restrict__ val,
                                                            const double *__restrict__ diag, int skip_first,
                                                            int skip_last, double *X, int nrhs) {
    for (int32_t l = l0; l < l1; l++) {
        const int32_t first = level_ptr[l], count = level_ptr[l + 1] - first;
        for (int32_t t = threadIdx.x; t < count * nrhs; t += 1024)
            solve_one(order[first + t / nrhs], t % nrhs, nrhs, ptr, idx, val, diag, skip_first, skip_last, X);
        __syncthreads();  // workgroup-scope release/acquire: the next level reads these x
    }
}

// ---- narrow levels, sixteen rows at a time --------------------------------------------------------------
// A run of narrow levels is a dependency chain; walking it one level per workgroup barrier costs a full
// round trip (or several: a row of 125 terms is 16 batches) per level.  But most of a row's terms do not
// depend on its immediate predecessors.  Rows are taken in blocks of 16 consecutive positions of the
// level order (a topological order): a term whose source lies BEFORE the block is final when the block
// starts, so the leading such terms of all 16 rows -- for a banded factor all but the last few -- are
// subtracted by 16 x 64 threads at once (phase A, one thread per (row, right-hand side), batched loads).
// What remains of a row, from its first in-block source on, waits for the rows above it: phase B runs
// the rows that have such a remainder one after another, their sources read from an LDS copy of the
// block's x, one workgroup barrier each.  Every (row, right-hand side) still subtracts its terms strictly
// in the reference's order -- prefix first, remainder after -- so x stays bit-identical.
constexpr int CHB = 16, CH_SUF = 32;   // rows per block; remainder terms staged in LDS per row

__global__ __launch_bounds__(64 * CHB) void k_tri_chain(const int32_t *__restrict__ order, int32_t p0, int32_t p1,
                                                        const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                        const double *__restrict__ val, const double *__restrict__ diag,
                                                        int skip_first, int skip_last, const int32_t *__restrict__ npre,
                                                        const int32_t *__restrict__ nin, const int8_t *__restrict__ tslot,
                                                        double *X, int nrhs, int relaxed) {
    __shared__ double xblk[CHB][64];
    __shared__ double sval[CHB][CH_SUF];
    __shared__ int32_t sidx[CHB][CH_SUF];
    __shared__ int8_t sslot[CHB][CH_SUF];
    __shared__ int32_t nsuf[CHB];
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    for (int32_t blk = p0 & ~(CHB - 1); blk < p1; blk += CHB) {
        const int32_t pos = blk + slot;
        const bool active = pos >= p0 && pos < p1;
        const int32_t row = pos < p1 ? order[pos] : 0;       // positions before p0 are solved rows of this block
        int32_t b = 0, e = 0, pre = 0;
        double dg = 1.0;
        if (active) {
            b = ptr[row] + skip_first;
            e = ptr[row + 1] - skip_last;
            pre = relaxed ? 0 : npre[pos];
            dg = diag[row];
        }
        // exact order: remainder = everything from the first in-block source on.  Relaxed order (opt-in, the
        // batched cholsol solve): remainder = the in-block sources only, all other terms go first -- a different
        // association of the same sum, equal to the reference's to rounding.
        const int32_t ns = (relaxed && active) ? nin[pos] : e - b - pre;
        for (int c0 = 0; c0 < nrhs; c0 += 64) {
            const int rhs = c0 + lane;
            const bool live = rhs < nrhs;
            const int rl = live ? rhs : nrhs - 1;
            double acc = 0.0;
            // ---- phase A ----
            if (active) {
                if (!relaxed) {
                    if (lane < CH_SUF && lane < ns) {         // the row's remainder: value, source, source's slot
                        sval[slot][lane] = val[b + pre + lane];
                        sidx[slot][lane] = idx[b + pre + lane];
                        sslot[slot][lane] = tslot[b + pre + lane];
                    }
                } else {                                      // compact the in-block terms, order kept
                    int base_ = 0;
                    for (int32_t q0 = b; q0 < e; q0 += 64) {
                        const int32_t q = q0 + lane;
                        const int sl = q < e ? (int)tslot[q] : -1;
                        const unsigned long long bal = __ballot(sl >= 0);
                        const int at = base_ + __popcll(bal & ((1ull << lane) - 1ull));
                        if (sl >= 0 && at < CH_SUF) {
                            sval[slot][at] = val[q];
                            sidx[slot][at] = idx[q];
                            sslot[slot][at] = (int8_t)sl;
                        }
                        base_ += __popcll(bal);
                    }
                }
                if (lane == 0) nsuf[slot] = ns;
                acc = X[(int64_t)row * nrhs + rl];
                constexpr int TB = 8;
                const int32_t pe = relaxed ? e : b + pre;
                int32_t c[TB];
                double v[TB];
                bool inb[TB];   // relaxed order: the term's source is in the block -> it belongs to phase B
#pragma unroll
                for (int u = 0; u < TB; u++) {
                    const int32_t q = b + u < pe ? b + u : (pe > b ? pe - 1 : 0);
                    c[u] = pe > b ? idx[q] : 0;
                    v[u] = pe > b ? val[q] : 0.0;
                    inb[u] = relaxed && pe > b && tslot[q] >= 0;
                }
                for (int32_t q0 = b; q0 < pe; q0 += TB) {
                    double xv[TB];
#pragma unroll
                    for (int u = 0; u < TB; u++) xv[u] = X[(int64_t)c[u] * nrhs + rl];
                    int32_t cn[TB];
                    double vn[TB];
                    bool inbn[TB];
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        const int32_t q = q0 + TB + u < pe ? q0 + TB + u : pe - 1;
                        cn[u] = idx[q];
                        vn[u] = val[q];
                        inbn[u] = relaxed && tslot[q] >= 0;
                    }
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        const double t = v[u] * xv[u];
                        acc = (q0 + u < pe && !inb[u]) ? acc - t : acc;
                    }
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        c[u] = cn[u];
                        v[u] = vn[u];
                        inb[u] = inbn[u];
                    }
                }
                if (ns == 0) {                                // nothing in-block: this row is done
                    const double xr = acc / dg;
                    xblk[slot][lane] = xr;
                    if (live) X[(int64_t)row * nrhs + rhs] = xr;
                }
            } else {
                if (lane == 0) nsuf[slot] = 0;
                if (pos < p0) xblk[slot][lane] = X[(int64_t)row * nrhs + rl];   // solved by an earlier segment
            }
            __syncthreads();
            // ---- phase B: rows with a remainder, in order ----
            for (int s2 = 0; s2 < CHB; s2++) {
                if (nsuf[s2] == 0) continue;                  // uniform
                if (slot == s2) {
                    for (int32_t t = 0; t < ns; t++) {
                        double v, xv;
                        if (t < CH_SUF) {
                            v = sval[slot][t];
                            const int sl = sslot[slot][t];
                            xv = sl >= 0 ? xblk[sl][lane] : X[(int64_t)sidx[slot][t] * nrhs + rl];
                        } else {                              // very long remainder (exact order only): from memory
                            const int32_t q = b + pre + t;
                            v = val[q];
                            const int sl = tslot[q];
                            xv = sl >= 0 ? xblk[sl][lane] : X[(int64_t)idx[q] * nrhs + rl];
                        }
                        const double tt = v * xv;
                        acc = acc - tt;
                    }
                    const double xr = acc / dg;
                    xblk[slot][lane] = xr;
                    if (live) X[(int64_t)row * nrhs + rhs] = xr;
                }
                __syncthreads();
            }
            __syncthreads();   // the LDS arrays are reused by the next chunk of right-hand sides / next block
        }
    }
}

// literal transcriptions of the reference loops, one lane per right-hand side
__global__ __launch_bounds__(64) void k_tri_sequential(int kind, int32_t n, const int32_t *Tp, const int32_t *Ti,
                                                       const double *Tx, double *X, int nrhs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrhs) return;
#define XV(i) X[(int64_t)(i) * nrhs + r]
    if (kind == CSX_TRI_L) {
        for (int32_t j = 0; j < n; j++) {
            XV(j) = XV(j) / Tx[Tp[j]];
            const double xj = XV(j);
            for (int32_t p = Tp[j] + 1; p < Tp[j + 1]; p++) {
                double t = Tx[p] * xj;
                XV(Ti[p]) = XV(Ti[p]) - t;
            }
        }
    } else if (kind == CSX_TRI_LT) {
        for (int32_t j = n - 1; j >= 0; j--) {
            for (int32_t p = Tp[j] + 1; p < Tp[j + 1]; p++) {
                double t = Tx[p] * XV(Ti[p]);
                XV(j) = XV(j) - t;
            }
            XV(j) = XV(j) / Tx[Tp[j]];
        }
    } else if (kind == CSX_TRI_U) {
        for (int32_t j = n - 1; j >= 0; j--) {
            XV(j) = XV(j) / Tx[Tp[j + 1] - 1];
            const double xj = XV(j);
            for (int32_t p = Tp[j]; p < Tp[j + 1] - 1; p++) {
                double t = Tx[p] * xj;
                XV(Ti[p]) = XV(Ti[p]) - t;
            }
        }
    } else {
        for (int32_t j = 0; j < n; j++) {
            for (int32_t p = Tp[j]; p < Tp[j + 1] - 1; p++) {
                double t = Tx[p] * XV(Ti[p]);
                XV(j) = XV(j) - t;
            }
            XV(j) = XV(j) / Tx[Tp[j + 1] - 1];
        }
    }
#undef XV
}

#pragma clang fp contract(fast)

__global__ __launch_bounds__(256) void k_permute(const int32_t *__restrict__ p, const double *__restrict__ b,
                                                 double *__restrict__ x, int32_t n, int nrhs, int inverse) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * nrhs) return;
    const int64_t k = t / nrhs, r = t % nrhs;
    const int64_t pk = p ? p[k] : k;
    if (inverse) x[pk * nrhs + r] = b[t];
    else x[t] = b[pk * nrhs + r];
}


// ---- connected components of the dependency graph, on the device ---------------------------------------
// W (BASELINE config 3) is 1 493 independent 67 x 67 blocks: its global level sets are ~45 levels each
// ~2 000 rows wide, i.e. ~80 dependent dispatches for L and U of ~6 us each, although no block ever waits for
// another.  Components are found with min-label hooking + pointer jumping (every term is an edge row--source),
// rows are grouped by component (stable sort by root keeps them ascending = a valid sweep order, descending
// for the backward kinds), and each component becomes a packed program for the fused in-LDS sweep of
// csx_sweep.h -- one launch, no level sets, nothing copied to the host but five counters.
__global__ __launch_bounds__(256) void k_cc_local_id(int32_t ncomp, const Tree *__restrict__ comps,
                                                     const uint32_t *__restrict__ srow, int32_t *local_id) {
    const int lane = threadIdx.x & 63;
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= ncomp) return;
    const Tree t = comps[c];
    for (int32_t a = lane; a < t.count; a += 64) local_id[srow[t.first + a]] = a;
}

// program length of sweep position k: forward kinds sweep a component's rows ascending, backward descending
__global__ __launch_bounds__(256) void k_prog_len(int32_t ncomp, const Tree *__restrict__ comps,
                                                  const uint32_t *__restrict__ srow, const int32_t *__restrict__ ptr,
                                                  int sf, int sl, int forward, int32_t *len) {
    const int lane = threadIdx.x & 63;
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= ncomp) return;
    const Tree t = comps[c];
    for (int32_t sp = lane; sp < t.count; sp += 64) {
        const int32_t row = (int32_t)srow[t.first + (forward ? sp : t.count - 1 - sp)];
        len[t.first + sp] = (ptr[row + 1] - sl) - (ptr[row] + sf);
    }
}

__global__ __launch_bounds__(256) void k_prog_fill(int32_t n, const int32_t *__restrict__ comp_of_pos,
                                                   const Tree *__restrict__ comps, const uint32_t *__restrict__ srow,
                                                   const int32_t *__restrict__ local_id, const int32_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                   const double *__restrict__ diag, int sf, int forward,
                                                   const int32_t *__restrict__ prog_ptr, int32_t *prog_idx,
                                                   double *prog_val, double *prog_diag) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // sweep position (global)
    if (k >= n) return;
    const Tree t = comps[comp_of_pos[k]];
    const int32_t sp = (int32_t)k - t.first;
    const int32_t row = (int32_t)srow[t.first + (forward ? sp : t.count - 1 - sp)];
    const int32_t gb = ptr[row] + sf, o = prog_ptr[k], len = prog_ptr[k + 1] - o;
    for (int32_t q = lane; q < len; q += 64) {
        prog_idx[o + q] = local_id[idx[gb + q]] * 64;   // premultiplied: the X tile is [local row][64 lanes]
        prog_val[o + q] = val[gb + q];
    }
    if (lane == 0) prog_diag[k] = diag[row];
}

// One wave = one component x 64 right-hand sides: X tile in LDS, one sweep in the plan's direction, every
// (row, right-hand side) subtracting its terms in the reference's order (csx_sweep.h) -> bit-identical.
constexpr int TL_WAVES_MAX = 4;
template <bool FORWARD>
__global__ __launch_bounds__(64 * TL_WAVES_MAX) void k_tri_local(const Tree *__restrict__ comps, int32_t ncomp,
                                                                 const int32_t *__restrict__ nodes,
                                                                 const int32_t *__restrict__ prog_ptr,
                                                                 const int32_t *__restrict__ prog_idx,
                                                                 const double *__restrict__ prog_val,
                                                                 const double *__restrict__ prog_diag, const double *Bsrc,
                                                                 double *B, const int32_t *__restrict__ load_rows,
                                                                 const int32_t *__restrict__ store_rows, int32_t nrhs,
                                                                 int32_t chunks, int32_t max_nodes, int32_t waves_per_wg) {
    // Bsrc / load_rows, B / store_rows: row j of the system is read from row load_rows[j] of Bsrc and written to row store_rows[j] of
    // B (null: row j; Bsrc == B: in place) -- cs_lusol's two permutations ride on its two sweeps (csx_lusol_solve), the order of
    // the arithmetic is untouched
    extern __shared__ __attribute__((aligned(16))) double xt[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= waves_per_wg) return;
    const int64_t task = (int64_t)blockIdx.x * waves_per_wg + w;
    if (task >= (int64_t)ncomp * chunks) return;
    const int32_t t = (int32_t)(task / chunks), h = (int32_t)(task % chunks);
    const Tree tr = comps[t];
    const int32_t rhs = h * 64 + lane;
    const bool live = rhs < nrhs;
    const int32_t rl = live ? rhs : nrhs - 1;        // clamped: lanes past the last right-hand side load, never store
    double *X = xt + (size_t)w * max_nodes * 64;
    for (int32_t c0 = 0; c0 < tr.count; c0 += 64) {
        const int32_t crow = min(64, tr.count - c0);
        int32_t jrow = lane < crow ? nodes[tr.first + c0 + lane] : 0;
        if (load_rows && lane < crow) jrow = load_rows[jrow];
        for (int32_t r0 = 0; r0 < crow; r0 += 16) {
            double tmp[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int32_t row = __builtin_amdgcn_readlane(jrow, min(r0 + u, crow - 1));
                tmp[u] = Bsrc[(int64_t)row * nrhs + rl];
            }
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (r0 + u < crow) X[(c0 + r0 + u) * 64 + lane] = tmp[u];
        }
    }
    sweep<FORWARD>(tr, prog_ptr, prog_idx, prog_val, prog_diag, X, lane);
    for (int32_t c0 = 0; c0 < tr.count; c0 += 64) {
        const int32_t crow = min(64, tr.count - c0);
        int32_t jrow = lane < crow ? nodes[tr.first + c0 + lane] : 0;
        if (store_rows && lane < crow) jrow = store_rows[jrow];
        for (int32_t r = 0; r < crow; r++) {
            const int32_t row = __builtin_amdgcn_readlane(jrow, r);
            if (live) B[(int64_t)row * nrhs + rhs] = X[(c0 + r) * 64 + lane];
        }
    }
}


// Few right-hand sides (nrhs <= 32): a wave that gives all 64 lanes to the right-hand sides of ONE component
// idles most of them.  Here a lane is a (component, right-hand side) pair: G = nrhs rounded up to a power of
// two lanes per component, 64 / G components per wave, each lane walking its own component's program in
// sweep order.  The solved unknowns of a lane's right-hand side sit in its own column of an LDS tile
// ([component in wave][local row][g], row count padded to an odd number against bank conflicts), so no lane
// ever reads another lane's data and no barrier is needed; a row's terms are fetched eight at a time and
// subtracted strictly in order (multiply and subtract rounded separately): bit-identical.  The next row's
// descriptor, right-hand-side entry and first term batch are requested while the current row is computed.
#pragma clang fp contract(off)
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_tri_few(const Tree *__restrict__ comps, int32_t ncomp,
                                                 const int32_t *__restrict__ nodes,
                                                 const int32_t *__restrict__ prog_ptr,
                                                 const int32_t *__restrict__ prog_idx,
                                                 const double *__restrict__ prog_val,
                                                 const double *__restrict__ prog_diag, double *B, int32_t nrhs, int G,
                                                 int32_t tile_rows, int32_t waves_per_wg) {
    extern __shared__ __attribute__((aligned(16))) double xt[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= waves_per_wg) return;
    const int cpw = 64 / G, cw = lane / G, g = lane % G;
    const int64_t comp = ((int64_t)blockIdx.x * waves_per_wg + w) * cpw + cw;
    if (comp >= ncomp || g >= nrhs) return;
    double *X = xt + ((size_t)w * cpw + cw) * tile_rows * G + g;   // this lane's column: X[local_row * G]
    const Tree tr = comps[comp];
    constexpr int TB = 8;
    // descriptor of the first row
    int32_t k = tr.first;
    int32_t tb = prog_ptr[k], te = prog_ptr[k + 1];
    int32_t row = nodes[FORWARD ? k : tr.first + tr.count - 1];
    double dg = prog_diag[k];
    double acc = B[(int64_t)row * nrhs + g];
    int32_t c[TB];
    double v[TB];
#pragma unroll
    for (int u = 0; u < TB; u++) {
        const int32_t q = tb + u < te ? tb + u : (te > tb ? te - 1 : 0);
        c[u] = te > tb ? prog_idx[q] : 0;
        v[u] = te > tb ? prog_val[q] : 0.0;
    }
    for (int32_t sp = 0; sp < tr.count; sp++) {
        // the next row's descriptor (clamped at the component's end: re-reads the last row, harmless)
        const int32_t kn = sp + 1 < tr.count ? k + 1 : k;
        const int32_t ntb = prog_ptr[kn], nte = prog_ptr[kn + 1];
        const int32_t nrow = nodes[FORWARD ? kn : tr.first + tr.count - 1 - (kn - tr.first)];
        const double ndg = prog_diag[kn];
        const double nacc = B[(int64_t)nrow * nrhs + g];
        for (int32_t q0 = tb; q0 < te; q0 += TB) {
            double xv[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) xv[u] = X[(c[u] >> 6) * G];
            int32_t cn[TB];
            double vn[TB];
            const bool more = q0 + TB < te;   // next batch of this row, else the first batch of the next row
            const int32_t nb_ = more ? q0 + TB : ntb, ne_ = more ? te : nte;
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int32_t q = nb_ + u < ne_ ? nb_ + u : (ne_ > nb_ ? ne_ - 1 : 0);
                cn[u] = ne_ > nb_ ? prog_idx[q] : 0;
                vn[u] = ne_ > nb_ ? prog_val[q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const double t = v[u] * xv[u];
                acc = q0 + u < te ? acc - t : acc;
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                c[u] = cn[u];
                v[u] = vn[u];
            }
        }
        if (te <= tb) {   // a row without terms never entered the loop: fetch the next row's first batch here
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int32_t q = ntb + u < nte ? ntb + u : (nte > ntb ? nte - 1 : 0);
                c[u] = nte > ntb ? prog_idx[q] : 0;
                v[u] = nte > ntb ? prog_val[q] : 0.0;
            }
        }
        const double xr = acc / dg;
        X[(FORWARD ? sp : tr.count - 1 - sp) * G] = xr;
        B[(int64_t)row * nrhs + g] = xr;
        k = kn;
        tb = ntb;
        te = nte;
        row = nrow;
        dg = ndg;
        acc = nacc;
    }
}
#pragma clang fp contract(fast)


// The same lane = (component, right-hand side) scheme with everything in LDS: a wave takes CPW consecutive
// components, whose packed programs are one contiguous piece of the plan, copies that piece and the rows'
// right-hand-side entries into LDS with coalesced loads, and then every lane walks its own component out of
// LDS (64 / CPW right-hand sides at a time).  Global memory sees two coalesced sweeps; the dependent chain
// of a component -- row after row, term after term, in the reference's order -- runs at LDS latency.
// W at one right-hand side: 241 us -> see profiles/r02_configs.jsonl.
#pragma clang fp contract(off)
struct __attribute__((aligned(16))) FewTerm {   // one term of a row, one 16-byte LDS read
    double v;
    int32_t src, pad;
};
struct __attribute__((aligned(16))) FewRow {    // one row: diagonal and extent of its terms, one 16-byte LDS read
    double diag;
    int32_t tb, te;
};

template <bool FORWARD, int CPW>
__global__ __launch_bounds__(64) void k_tri_few_lds(const Tree *__restrict__ comps, int32_t ncomp,
                                                    const int32_t *__restrict__ nodes,
                                                    const int32_t *__restrict__ prog_ptr,
                                                    const int32_t *__restrict__ prog_idx,
                                                    const double *__restrict__ prog_val,
                                                    const double *__restrict__ prog_diag, double *B, int32_t nrhs,
                                                    int32_t rows_cap, int32_t terms_cap) {
    constexpr int G = 64 / CPW;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    FewTerm *s_term = reinterpret_cast<FewTerm *>(lds_raw);                    // [terms_cap]
    FewRow *s_row = reinterpret_cast<FewRow *>(s_term + terms_cap);            // [rows_cap]
    double *xs = reinterpret_cast<double *>(s_row + rows_cap);                 // [rows_cap * G]
    const int lane = threadIdx.x;
    const int32_t c0 = blockIdx.x * CPW, c1 = min(c0 + CPW, ncomp);
    const int32_t first_k = comps[c0].first, last_k = comps[c1 - 1].first + comps[c1 - 1].count;
    const int32_t nrows = last_k - first_k, tbase = prog_ptr[first_k], nterms = prog_ptr[last_k] - tbase;
    // staging: four independent loads in flight per lane and pass (a loop that waits for each load in turn
    // costs a memory round trip per 64 elements -- measured 27 us of a 78 us kernel)
    constexpr int SU = 4;
    for (int32_t i0 = lane; i0 < nrows; i0 += 64 * SU) {
        int32_t pb[SU], pe[SU];
        double dd[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int32_t i = min(i0 + 64 * u, nrows - 1);
            pb[u] = prog_ptr[first_k + i];
            pe[u] = prog_ptr[first_k + i + 1];
            dd[u] = prog_diag[first_k + i];
        }
#pragma unroll
        for (int u = 0; u < SU; u++)
            if (i0 + 64 * u < nrows) s_row[i0 + 64 * u] = {dd[u], pb[u] - tbase, pe[u] - tbase};
    }
    for (int32_t q0 = lane; q0 < nterms; q0 += 64 * SU) {
        int32_t ii[SU];
        double vv[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int32_t q = min(q0 + 64 * u, nterms - 1);
            ii[u] = prog_idx[tbase + q];
            vv[u] = prog_val[tbase + q];
        }
#pragma unroll
        for (int u = 0; u < SU; u++)
            if (q0 + 64 * u < nterms) s_term[q0 + 64 * u] = {vv[u], ii[u] >> 6, 0};   // the plan stores local row * 64
    }
    const int cw = lane / G, g = lane % G;
    const int32_t comp = c0 + cw;
    Tree tr = {0, 0};
    if (comp < c1) tr = comps[comp];
    const int32_t rb = tr.first - first_k;                     // this component's first tile row
    for (int32_t r0 = 0; r0 < nrhs; r0 += G) {
        const int32_t gw = min(G, nrhs - r0), nx = nrows * gw;   // only the live right-hand sides are moved
        for (int32_t i0 = lane; i0 < nx; i0 += 64 * SU) {
            int32_t nd[SU];
            double bb[SU];
#pragma unroll
            for (int u = 0; u < SU; u++) nd[u] = nodes[first_k + min(i0 + 64 * u, nx - 1) / gw];
#pragma unroll
            for (int u = 0; u < SU; u++) bb[u] = B[(int64_t)nd[u] * nrhs + r0 + min(i0 + 64 * u, nx - 1) % gw];
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int32_t i = i0 + 64 * u;
                if (i < nx) xs[(i / gw) * G + i % gw] = bb[u];
            }
        }
        __syncthreads();
        if (comp < c1 && r0 + g < nrhs) {
            // The chain that cannot be shortened is: x of the sources (LDS) -> multiply -> ordered subtractions ->
            // division -> x of this row (LDS).  The row record and the next (source, value) records do not depend
            // on x and are read ahead, beside that chain.
            constexpr int TB = 8;
            FewRow cur = s_row[rb];
            FewTerm t[TB];
#pragma unroll
            for (int u = 0; u < TB; u++) t[u] = s_term[cur.tb + u < cur.te ? cur.tb + u : (cur.te > cur.tb ? cur.te - 1 : 0)];
            for (int32_t sp = 0; sp < tr.count; sp++) {
                const FewRow nxt = s_row[rb + (sp + 1 < tr.count ? sp + 1 : sp)];
                const int32_t trow = rb + (FORWARD ? sp : tr.count - 1 - sp);
                double acc = xs[trow * G + g];
                int32_t q0 = cur.tb;
                do {
                    double xv[TB];
#pragma unroll
                    for (int u = 0; u < TB; u++) xv[u] = xs[(rb + t[u].src) * G + g];
                    // the next batch: of this row if it has more terms, else the first one of the next row
                    const bool more = q0 + TB < cur.te;
                    const int32_t nb_ = more ? q0 + TB : nxt.tb, ne_ = more ? cur.te : nxt.te;
                    FewTerm tn[TB];
#pragma unroll
                    for (int u = 0; u < TB; u++) tn[u] = s_term[nb_ + u < ne_ ? nb_ + u : (ne_ > nb_ ? ne_ - 1 : 0)];
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        const double pr = t[u].v * xv[u];
                        acc = q0 + u < cur.te ? acc - pr : acc;
                    }
#pragma unroll
                    for (int u = 0; u < TB; u++) t[u] = tn[u];
                    q0 += TB;
                } while (q0 < cur.te);
                xs[trow * G + g] = acc / cur.diag;
                cur = nxt;
            }
        }
        __syncthreads();
        for (int32_t i = lane; i < nx; i += 64) B[(int64_t)nodes[first_k + i / gw] * nrhs + r0 + i % gw] = xs[(i / gw) * G + i % gw];
        __syncthreads();
    }
}
#pragma clang fp contract(fast)


// ---- components, column-push form (kinds L and U, few right-hand sides) ----------------------------------
// In gather form a row waits for its ~7 terms one after the other; in the reference's own COLUMN-PUSH form the
// ~7 updates a finished x[j] causes go to distinct rows and can be done by as many lanes at once, each row still
// receiving its updates in ascending (L) / descending (U) column order -- bit-identical.  One wave per
// component, its column program (local row, value) and its x copied into LDS, lanes = (entry of the column,
// right-hand side).  The chain that remains is one division and one LDS round trip per column.
#pragma clang fp contract(off)
template <bool FORWARD>
__global__ __launch_bounds__(64) void k_tri_comp_push(const Tree *__restrict__ comps, int32_t ncomp,
                                                      const int32_t *__restrict__ nodes,
                                                      const int32_t *__restrict__ cptr, const int32_t *__restrict__ cidx,
                                                      const double *__restrict__ cval, const double *__restrict__ cdiag,
                                                      double *B, int32_t nrhs, int G, int32_t rows_cap, int32_t terms_cap) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    FewTerm *s_term = reinterpret_cast<FewTerm *>(lds_raw);                    // [terms_cap]
    FewRow *s_col = reinterpret_cast<FewRow *>(s_term + terms_cap);            // [rows_cap]: diagonal, extent
    double *xs = reinterpret_cast<double *>(s_col + rows_cap);                 // [rows_cap * G]
    const int lane = threadIdx.x;
    const int32_t comp = blockIdx.x;
    if (comp >= ncomp) return;
    const Tree tr = comps[comp];
    const int32_t first_k = tr.first, nrows = tr.count, tbase = cptr[first_k], nterms = cptr[first_k + nrows] - tbase;
    constexpr int SU = 4;
    for (int32_t i0 = lane; i0 < nrows; i0 += 64 * SU) {
        int32_t pb[SU], pe[SU];
        double dd[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int32_t i = min(i0 + 64 * u, nrows - 1);
            pb[u] = cptr[first_k + i];
            pe[u] = cptr[first_k + i + 1];
            dd[u] = cdiag[first_k + i];
        }
#pragma unroll
        for (int u = 0; u < SU; u++)
            if (i0 + 64 * u < nrows) s_col[i0 + 64 * u] = {dd[u], pb[u] - tbase, pe[u] - tbase};
    }
    for (int32_t q0 = lane; q0 < nterms; q0 += 64 * SU) {
        int32_t ii[SU];
        double vv[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int32_t q = min(q0 + 64 * u, nterms - 1);
            ii[u] = cidx[tbase + q];
            vv[u] = cval[tbase + q];
        }
#pragma unroll
        for (int u = 0; u < SU; u++)
            if (q0 + 64 * u < nterms) s_term[q0 + 64 * u] = {vv[u], ii[u], 0};
    }
    const int epl = 64 / G, t = lane / G, g = lane % G;    // entries of a column handled per pass; this lane's entry, RHS
    for (int32_t r0 = 0; r0 < nrhs; r0 += G) {
        const int32_t gw = min(G, nrhs - r0), nx = nrows * gw;
        for (int32_t i0 = lane; i0 < nx; i0 += 64 * SU) {
            int32_t nd[SU];
            double bb[SU];
#pragma unroll
            for (int u = 0; u < SU; u++) nd[u] = nodes[first_k + min(i0 + 64 * u, nx - 1) / gw];
#pragma unroll
            for (int u = 0; u < SU; u++) bb[u] = B[(int64_t)nd[u] * nrhs + r0 + min(i0 + 64 * u, nx - 1) % gw];
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int32_t i = i0 + 64 * u;
                if (i < nx) xs[(i / gw) * G + i % gw] = bb[u];
            }
        }
        __syncthreads();
        if (g < gw) {
            FewRow cur = s_col[0];
            for (int32_t sp = 0; sp < nrows; sp++) {
                const FewRow nxt = s_col[sp + 1 < nrows ? sp + 1 : sp];
                const int32_t a = FORWARD ? sp : nrows - 1 - sp;          // local row of this column's unknown
                const double xj = xs[a * G + g] / cur.diag;               // every entry lane: same operands, same result
                for (int32_t q = cur.tb + t; q < cur.te; q += epl) {
                    const FewTerm e = s_term[q];
                    const double pr = e.v * xj;
                    xs[e.src * G + g] = xs[e.src * G + g] - pr;
                }
                if (t == 0) xs[a * G + g] = xj;                           // after every lane's read of it (same wave, in order)
                cur = nxt;
            }
        }
        __syncthreads();
        for (int32_t i = lane; i < nx; i += 64) B[(int64_t)nodes[first_k + i / gw] * nrhs + r0 + i % gw] = xs[(i / gw) * G + i % gw];
        __syncthreads();
    }
}
#pragma clang fp contract(fast)

// column program of the push kinds: sweep position k of a component <-> column j of T without its diagonal;
// local row = ascending index of the row inside the component
__global__ __launch_bounds__(256) void k_push_len(int32_t ncomp, const Tree *__restrict__ comps,
                                                  const uint32_t *__restrict__ srow, const int32_t *__restrict__ Tp,
                                                  int forward, int32_t *len) {
    const int lane = threadIdx.x & 63;
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= ncomp) return;
    const Tree t = comps[c];
    for (int32_t sp = lane; sp < t.count; sp += 64) {
        const int32_t j = (int32_t)srow[t.first + (forward ? sp : t.count - 1 - sp)];
        len[t.first + sp] = Tp[j + 1] - Tp[j] - 1;
    }
}

__global__ __launch_bounds__(256) void k_push_fill(int32_t n, const int32_t *__restrict__ comp_of_pos,
                                                   const Tree *__restrict__ comps, const uint32_t *__restrict__ srow,
                                                   const int32_t *__restrict__ local_id, const int32_t *__restrict__ Tp,
                                                   const int32_t *__restrict__ Ti, const double *__restrict__ Tx, int forward,
                                                   const int32_t *__restrict__ cptr, int32_t *cidx, double *cval, double *cdiag,
                                                   int *stats) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= n) return;
    const Tree t = comps[comp_of_pos[k]];
    const int32_t sp = (int32_t)k - t.first;
    const int32_t j = (int32_t)srow[t.first + (forward ? sp : t.count - 1 - sp)];
    const int32_t b = Tp[j], e = Tp[j + 1];
    const int32_t lo = forward ? b + 1 : b;           // L: diagonal first; U: diagonal last
    const int32_t o = cptr[k], len = cptr[k + 1] - o;
    for (int32_t q = lane; q < len; q += 64) {
        cidx[o + q] = local_id[Ti[lo + q]];
        cval[o + q] = Tx[lo + q];
    }
    if (lane == 0) {
        cdiag[k] = Tx[forward ? b : e - 1];
        if (sp == 0) atomicMax(&stats[0], cptr[t.first + t.count] - cptr[t.first]);   // most terms in one component
    }
}

// largest number of rows / terms in any group of `cpw` consecutive components
__global__ __launch_bounds__(256) void k_group_caps(int32_t ncomp, int cpw, const Tree *__restrict__ comps,
                                                    const int32_t *__restrict__ prog_ptr, int *caps) {
    const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c0 = gidx * cpw;
    if (c0 >= ncomp) return;
    const int32_t c1 = (int32_t)(c0 + cpw < ncomp ? c0 + cpw : ncomp);
    const int32_t fk = comps[c0].first, lk = comps[c1 - 1].first + comps[c1 - 1].count;
    atomicMax(&caps[0], lk - fk);
    atomicMax(&caps[1], prog_ptr[lk] - prog_ptr[fk]);
}


// ---- small chain-like systems: the reference's loop, one workgroup per right-hand side, x in LDS ------------
// bcsstk16's factor (n = 4 884, 125 entries per column, 4 810 levels) offers a level schedule nothing to
// schedule: 7 us per level made the forward solve 15 ms and the backward one 34 ms, against a third of a
// millisecond for one host core.  When the whole x of one right-hand side fits LDS (n <= 15 360) the workgroup
// simply runs the reference's column loop: for L and U (column push) the column's entries are spread over the
// threads (distinct rows, each x[i] still receives its updates in ascending / descending column order), one
// workgroup barrier per column, the next column's entries already in flight; for L' and U' (column gather) one wave
// forms a column's products, one per lane, and lane 0 subtracts them in storage order while they rotate towards it
// (k_tri_colchain).  Bit-identical, every kind.  Right-hand sides are independent workgroups.  L' on bcsstk16: 256
// threads with the products in LDS and one lane subtracting 9.0 ms; one wave lifting products out with v_readlane
// 9.6 ms, with the v_readlane of the next eight issued ahead 7.7; products rotated by DPP 7.1; entries requested three
// columns ahead instead of one 6.6 ms.
#pragma clang fp contract(off)
constexpr int TC_THREADS = 256;
constexpr int TC_MAX_N = 15360;
constexpr int TR64_RUN = 64, TR64_RUN_MIN = 8;   // levels per two-phase run of narrow levels; shorter runs: one phase
constexpr int TRW_MAX_RHS = 4, TRW_MIN_ROW = 24;   // a wave per row: at most so many right-hand sides, rows at least so long on average

// The gather kinds (L', U') on ONE wave: every lane forms one product of the column, then lane 0 subtracts them in
// storage order while the products rotate towards it through the wave (tch_chain) -- no barrier and no LDS round trip
// for the products between columns.  Lanes past the end of the column hold +0.0, and x - (+0.0) = x for every x, so
// the chain runs in blocks of eight without a test per term.
// (Kept for reference and A/B timing: the kernels now hand the products over through LDS, tch_chain_lds below --
// bcsstk16 L' 6.6 -> 5.9 ms, a band of 200: 3.1 -> 1.8 us per column; what is left is the dependent fp64 subtraction,
// ~19 cycles per term.)
__device__ __forceinline__ double tch_chain(double acc, double p, int cnt) {
    // Only lane 0's chain is the result.  The products are rotated through the wave, one lane per step (DPP
    // wave_rol:1: lane l receives lane l + 1), so lane 0 meets product 0, 1, 2, ... in order; the two rotations of a
    // step do not depend on the subtraction, which is the only chain.  (Lifting the products out with v_readlane
    // costs ~10 cycles per scalar write, 30 cycles per term; rotating in place 26; eight rotations issued ahead into
    // registers of their own 34: a wave_rol is a long instruction and eight in a row serialise.)
    int plo = __double2loint(p), phi = __double2hiint(p);
#pragma unroll
    for (int g = 0; g < 64; g += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = acc - __hiloint2double(phi, plo);
            plo = __builtin_amdgcn_update_dpp(plo, plo, 0x134, 0xf, 0xf, false);
            phi = __builtin_amdgcn_update_dpp(phi, phi, 0x134, 0xf, 0xf, false);
        }
        if (g + 8 >= cnt) break;   // uniform
    }
    return acc;
}

// The same chain with the products handed to lane 0 through LDS instead of the DPP rotation: every lane stores its
// product (one ds_write_b64), then ALL lanes read the products back two at a time from the same addresses (broadcast
// reads, 16 products per block, the next block requested before this block's subtractions), so the only serial chain
// is the subtraction itself.  buf: 64 doubles private to the wave, 16-byte aligned.  LDS operations of one wave complete
// in order, so neither the read-back nor the next call's store needs a barrier.
__device__ __forceinline__ double tch_chain_lds(double acc, double p, int cnt, double *buf, int lane) {
    buf[lane] = p;
    const double2 *b2 = reinterpret_cast<const double2 *>(buf);
    double2 cur[8], nxt[8];
#pragma unroll
    for (int q = 0; q < 8; q++) cur[q] = b2[q];
#pragma unroll
    for (int g = 0; g < 64; g += 16) {
        // the next block is requested whether or not it will be used: with the request under a branch the compiler
        // has to wait for EVERY outstanding read before the first subtraction (the two paths differ in how many there
        // are), which puts one LDS latency per 16 terms back on the chain
#pragma unroll
        for (int q = 0; q < 8; q++) nxt[q] = b2[((g + 16) & 63) / 2 + q];
        __builtin_amdgcn_sched_barrier(0);     // ... and BEFORE this block's subtractions
#pragma unroll
        for (int q = 0; q < 8; q++) {
            acc = acc - cur[q].x;
            acc = acc - cur[q].y;
        }
        if (g + 16 >= cnt) break;              // uniform
#pragma unroll
        for (int q = 0; q < 8; q++) cur[q] = nxt[q];
    }
    return acc;
}

struct TchCol {   // a column's off-diagonal range, its diagonal, and two entries per lane
    int32_t lo, hi;
    double dg;
    int32_t ci0, ci1;
    double cv0, cv1;
};

template <bool DIAG_FIRST>
__device__ __forceinline__ TchCol tch_load(int32_t b, int32_t e, const int32_t *__restrict__ Ti,
                                           const double *__restrict__ Tx, int lane) {
    TchCol c;
    c.lo = DIAG_FIRST ? b + 1 : b;
    c.hi = DIAG_FIRST ? e : e - 1;
    c.dg = Tx[DIAG_FIRST ? b : e - 1];
    c.ci0 = c.lo + lane < c.hi ? Ti[c.lo + lane] : 0;
    c.ci1 = c.lo + 64 + lane < c.hi ? Ti[c.lo + 64 + lane] : 0;
    c.cv0 = c.lo + lane < c.hi ? Tx[c.lo + lane] : 0.0;
    c.cv1 = c.lo + 64 + lane < c.hi ? Tx[c.lo + 64 + lane] : 0.0;
    return c;
}

// The factor does not fit an XCD's L2 (bcsstk16: 7.3 MB), so a column's entries come from the memory side and take
// about as long as a column takes to compute: they are requested THREE columns ahead (pointers four ahead).
template <int KIND>   // CSX_TRI_LT or CSX_TRI_UT
__global__ __launch_bounds__(64) void k_tri_colchain(int32_t n, const int32_t *__restrict__ Tp, const int32_t *__restrict__ Ti,
                                                  const double *__restrict__ Tx, double *X, int nrhs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // n doubles
    __shared__ __attribute__((aligned(16))) double cbuf[64];
    const int lane = threadIdx.x, r = blockIdx.x;
    for (int32_t i = lane; i < n; i += 64) xs[i] = X[(int64_t)i * nrhs + r];
    __syncthreads();
    constexpr bool ASC = KIND == CSX_TRI_UT;
    constexpr bool DIAG_FIRST = KIND == CSX_TRI_LT;
    auto col_at = [&](int32_t step) {   // the column of a step, clamped to the last one
        const int32_t st = step < n ? step : n - 1;
        return ASC ? st : n - 1 - st;
    };
    // A ring of four columns in registers, the loop unrolled by four so that every slot has a fixed name: a load
    // issued at step s is first touched at step s + 3.  (Rotating named variables instead -- cur = n1; n1 = n2; ... --
    // makes the compiler wait for the newest load at the end of every step: the copy needs the data.)
    TchCol ring[4];
    int32_t pb[4], pe[4];                       // column pointers, one step ahead of the entries
#pragma unroll
    for (int u = 0; u < 3; u++) ring[u] = tch_load<DIAG_FIRST>(Tp[col_at(u)], Tp[col_at(u) + 1], Ti, Tx, lane);
    pb[3] = Tp[col_at(3)];
    pe[3] = Tp[col_at(3) + 1];
    for (int32_t base = 0; base < n; base += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t step = base + u;
            if (step >= n) break;                       // uniform
            const int32_t j = col_at(step);
            const int32_t j4 = col_at(step + 4);
            pb[u] = Tp[j4];                             // pointers of the column four steps on (slot u is free: its
            pe[u] = Tp[j4 + 1];                         // pointers were consumed at the previous step)
            ring[(u + 3) & 3] = tch_load<DIAG_FIRST>(pb[(u + 3) & 3], pe[(u + 3) & 3], Ti, Tx, lane);
            const TchCol &cur = ring[u];
            const int32_t len = cur.hi - cur.lo;
            double acc = xs[j];
            const double p0 = lane < len ? cur.cv0 * xs[cur.ci0] : 0.0;
            const double p1 = 64 + lane < len ? cur.cv1 * xs[cur.ci1] : 0.0;
            acc = tch_chain_lds(acc, p0, len, cbuf, lane);
            if (len > 64) acc = tch_chain_lds(acc, p1, len - 64, cbuf, lane);
            for (int32_t q0 = 128; q0 < len; q0 += 64) {   // columns longer than two rounds of the wave
                const double pq = q0 + lane < len ? Tx[cur.lo + q0 + lane] * xs[Ti[cur.lo + q0 + lane]] : 0.0;
                acc = tch_chain_lds(acc, pq, len - q0, cbuf, lane);
            }
            const double xj = acc / cur.dg;                   // lane 0's is the one
            if (lane == 0) {
                xs[j] = xj;
                X[(int64_t)j * nrhs + r] = xj;
            }
        }
    }
}
// ---- level schedule, ONE WAVE PER ROW (few right-hand sides, long rows) ---------------------------------------
// The level kernels above give a thread to every (row, right-hand side): with one right-hand side and rows of
// thousands of terms (the separator rows of a nested-dissection factor) a handful of threads walk them alone.  Here a
// wave takes a row: the lanes fetch 64 terms at a time and multiply them by their x (all final: earlier levels), and
// lane 0 subtracts the products in the reference's order as in k_tri_colchain -- the same bits as solve_one.
// The gathers of x run one block of 64 terms ahead of the chain and the (index, value) loads two blocks ahead.
// PREFIX (first phase of a two-phase run of narrow levels, see k_tri_run_prefix64 below): the chain stops in front of
// the first term whose source lies in level >= l0; the partial sum is stored in place, the position in resume_out.
template <bool PREFIX>
__device__ __forceinline__ void solve_row_wave(int32_t row, int nrhs, const int32_t *__restrict__ ptr,
                                               const int32_t *__restrict__ idx, const double *__restrict__ val,
                                               const double *__restrict__ diag, int skip_first, int skip_last, double *X,
                                               int lane, double *cbuf, const int32_t *__restrict__ resume_in,
                                               const int32_t *__restrict__ level_of, int32_t l0, int32_t *resume_out) {
    const int32_t b = resume_in ? resume_in[row] : ptr[row] + skip_first, e = ptr[row + 1] - skip_last;
    const double dg = PREFIX ? 1.0 : diag[row];
    int32_t stop_at = e;
    for (int r = 0; r < nrhs; r++) {
        double acc = X[(int64_t)row * nrhs + r];
        int32_t ci0 = b + lane < e ? idx[b + lane] : 0, ci1 = b + 64 + lane < e ? idx[b + 64 + lane] : 0;
        double cv0 = b + lane < e ? val[b + lane] : 0.0, cv1 = b + 64 + lane < e ? val[b + 64 + lane] : 0.0;
        double xv0 = X[(int64_t)ci0 * nrhs + r];
        int32_t lv0 = PREFIX ? level_of[ci0] : 0;
        for (int32_t q0 = b; q0 < e; q0 += 64) {
            const int32_t q2 = q0 + 128 + lane;            // two blocks ahead: indices and values
            const int32_t ci2 = q2 < e ? idx[q2] : 0;
            const double cv2 = q2 < e ? val[q2] : 0.0;
            const double xv1 = X[(int64_t)ci1 * nrhs + r];   // one block ahead: the gathers
            const int32_t lv1 = PREFIX ? level_of[ci1] : 0;
            int32_t cnt = e - q0;
            bool last = false;
            if (PREFIX) {
                const unsigned long long m = __ballot(q0 + lane < e && lv0 >= l0);
                if (m) {
                    cnt = __ffsll((long long)m) - 1;
                    stop_at = q0 + cnt;
                    last = true;
                }
            }
            const double p = lane < cnt ? cv0 * xv0 : 0.0;
            if (cnt > 0) acc = tch_chain_lds(acc, p, cnt, cbuf, lane);
            if (last) break;
            ci0 = ci1; cv0 = cv1; xv0 = xv1; lv0 = lv1;
            ci1 = ci2; cv1 = cv2;
        }
        if (lane == 0) X[(int64_t)row * nrhs + r] = PREFIX ? acc : acc / dg;
    }
    if (PREFIX && lane == 0) resume_out[row] = stop_at;
}

__global__ __launch_bounds__(256) void k_tri_level_rows(const int32_t *__restrict__ order, int32_t first, int32_t count,
                                                        const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                        const double *__restrict__ val, const double *__restrict__ diag,
                                                        int skip_first, int skip_last, double *X, int nrhs) {
    __shared__ __attribute__((aligned(16))) double cbuf[4][64];
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= count) return;
    solve_row_wave<false>(order[first + w], nrhs, ptr, idx, val, diag, skip_first, skip_last, X, lane, cbuf[threadIdx.x >> 6],
                          nullptr, nullptr, 0, nullptr);
}

// first phase of a two-phase run of narrow levels, a wave per row of the run (few right-hand sides)
__global__ __launch_bounds__(256) void k_tri_run_prefix_rows(const int32_t *__restrict__ order, int32_t first, int32_t count,
                                                             int32_t l0, const int32_t *__restrict__ level_of,
                                                             const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                             const double *__restrict__ val, int skip_first, int skip_last,
                                                             double *X, int nrhs, int32_t *__restrict__ resume) {
    __shared__ __attribute__((aligned(16))) double cbuf[4][64];
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= count) return;
    solve_row_wave<true>(order[first + w], nrhs, ptr, idx, val, nullptr, skip_first, skip_last, X, lane, cbuf[threadIdx.x >> 6],
                         nullptr, level_of, l0, resume);
}

__global__ __launch_bounds__(1024) void k_tri_levels_rows_one_wg(const int32_t *__restrict__ order,
                                                                 const int32_t *__restrict__ level_ptr, int32_t l0,
                                                                 int32_t l1, const int32_t *__restrict__ ptr,
                                                                 const int32_t *__restrict__ idx,
                                                                 const double *__restrict__ val,
                                                                 const double *__restrict__ diag, int skip_first,
                                                                 int skip_last, double *X, int nrhs,
                                                                 const int32_t *__restrict__ resume) {
    __shared__ __attribute__((aligned(16))) double cbuf[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int32_t l = l0; l < l1; l++) {
        const int32_t first = level_ptr[l], count = level_ptr[l + 1] - first;
        for (int32_t t = w; t < count; t += 16)
            solve_row_wave<false>(order[first + t], nrhs, ptr, idx, val, diag, skip_first, skip_last, X, lane, cbuf[w], resume,
                                  nullptr, 0, nullptr);
        __syncthreads();  // workgroup-scope release/acquire: the next level reads these x
    }
}

// ---- level schedule, a wave per (row, block of 64 right-hand sides) -- many right-hand sides, long rows ----------
// With a thread per (row, right-hand side) every thread fetches the row's (index, value) pairs for itself and keeps
// eight gathers in flight: a separator row of 2 000 terms is 250 dependent round trips.  Here the lanes of a wave are
// 64 right-hand sides of ONE row: the row's indices and values are uniform (scalar loads), a term's x values are one
// coalesced 512-byte row of X, and a chunk of 16 terms is 16 independent loads per lane with the next chunk behind it.
// Same subtractions in the same order as solve_one: same bits.
constexpr int TR64_MIN_RHS = 5;    // 5 .. 15 right-hand sides: the same kernel with idle lanes beats a thread per (row, right-hand side) on long rows (700 x 700 grid, order 1, 8 right-hand sides: 730 -> ~35 ms)

// The chain of one (row, 64 right-hand sides).  A row's terms are fetched 64 at a time, one per lane (coalesced, the
// next 64 requested before these are used), and handed round by v_readlane: a term's source index becomes a scalar
// base for one coalesced 512-byte load of X, its value a scalar factor.  Gathers are issued 16 terms ahead of the
// subtractions.  PREFIX: stop at the first term whose source lies in level >= l0 (see k_tri_run_prefix64), no division.
template <bool PREFIX>
__device__ __forceinline__ void solve_row_rhs64(int32_t row, int rb, int nrhs, const int32_t *__restrict__ ptr,
                                                const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                const double *__restrict__ diag, int skip_first, int skip_last, double *X,
                                                int lane, const int32_t *__restrict__ resume_in,
                                                const int32_t *__restrict__ level_of, int32_t l0, int32_t *resume_out) {
    const int r = rb * 64 + lane;
    const bool live = r < nrhs;
    const int rr = live ? r : nrhs - 1;          // a valid address either way
    const int32_t b = resume_in ? resume_in[row] : ptr[row] + skip_first, e = ptr[row + 1] - skip_last;
    double acc = X[(int64_t)row * nrhs + rr];
    int32_t ci = b + lane < e ? idx[b + lane] : 0;
    double cv = b + lane < e ? val[b + lane] : 0.0;
    int32_t stop_at = e;
    for (int32_t q0 = b; q0 < e; q0 += 64) {
        const int32_t qn = q0 + 64 + lane;
        const int32_t cin = qn < e ? idx[qn] : 0;
        const double cvn = qn < e ? val[qn] : 0.0;
        int cnt = e - q0 < 64 ? e - q0 : 64;     // uniform
        bool last = false;
        if (PREFIX) {
            const bool in_run = lane < cnt && level_of[ci] >= l0;
            const unsigned long long m = __ballot(in_run);
            if (m) {
                const int first_in = __ffsll((long long)m) - 1;
                cnt = first_in;
                stop_at = q0 + first_in;
                last = true;
            }
        }
        const int clo = __double2loint(cv), chi = __double2hiint(cv);
        double xa[16], xb[16];
        auto gather = [&](int g, double *xv) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int32_t c = __builtin_amdgcn_readlane(ci, g + u);
                xv[u] = X[(int64_t)c * nrhs + rr];
            }
        };
        auto chain = [&](int g, const double *xv) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const double v = __hiloint2double(__builtin_amdgcn_readlane(chi, g + u), __builtin_amdgcn_readlane(clo, g + u));
                const double t = v * xv[u];
                acc = g + u < cnt ? acc - t : acc;
            }
        };
        if (cnt > 0) {
            gather(0, xa);
            if (cnt > 16) gather(16, xb);
            chain(0, xa);
            if (cnt > 16) {
                if (cnt > 32) gather(32, xa);
                chain(16, xb);
                if (cnt > 32) {
                    if (cnt > 48) gather(48, xb);
                    chain(32, xa);
                    if (cnt > 48) chain(48, xb);
                }
            }
        }
        if (last) break;
        ci = cin;
        cv = cvn;
    }
    if (PREFIX) {
        if (live) X[(int64_t)row * nrhs + r] = acc;
        if (rb == 0 && lane == 0) resume_out[row] = stop_at;
    } else if (live) {
        X[(int64_t)row * nrhs + r] = acc / diag[row];
    }
}

// A RUN of narrow levels [l0, l1) (the separators at the top of a nested-dissection tree: a few rows per level, each
// with thousands of terms) walked level by level costs rows x terms of serial latency.  But a row's terms come in
// ascending source order and its leading ones -- usually nearly all -- have sources in levels BELOW the run, final
// before the run starts.  So the run is done in two phases, 64 levels at a time: here, one launch over all rows of the
// run at once, every (row, 64 right-hand sides) chain is run up to its first term whose source lies inside the run, the
// partial sum stored in place (nobody reads x[row] before the row's own level) and the position kept in resume[row];
// then the level walker takes each chain up where it stopped.  The chain itself is unchanged: same bits.
__global__ __launch_bounds__(256) void k_tri_run_prefix64(const int32_t *__restrict__ order, int32_t first, int32_t count,
                                                          int32_t l0, const int32_t *__restrict__ level_of,
                                                          const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                          const double *__restrict__ val, int skip_first, int skip_last,
                                                          double *X, int nrhs, int32_t *__restrict__ resume) {
    const int lane = threadIdx.x & 63;
    const int nblk = (nrhs + 63) >> 6;
    const int64_t w = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (w >= (int64_t)count * nblk) return;
    const int32_t row = __builtin_amdgcn_readfirstlane(order[first + w / nblk]);
    solve_row_rhs64<true>(row, (int)(w % nblk), nrhs, ptr, idx, val, nullptr, skip_first, skip_last, X, lane, nullptr,
                          level_of, l0, resume);
}

__global__ __launch_bounds__(256) void k_tri_level_of(int32_t nlevels, const int32_t *__restrict__ level_ptr,
                                                      const int32_t *__restrict__ order, int32_t n, int32_t *level_of) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    int32_t lo = 0, hi = nlevels;                    // level l holds positions [level_ptr[l], level_ptr[l + 1])
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (level_ptr[mid] <= q) lo = mid;
        else hi = mid;
    }
    level_of[order[q]] = lo;
}

__global__ __launch_bounds__(256) void k_tri_level_rows64(const int32_t *__restrict__ order, int32_t first, int32_t count,
                                                          const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                          const double *__restrict__ val, const double *__restrict__ diag,
                                                          int skip_first, int skip_last, double *X, int nrhs) {
    const int lane = threadIdx.x & 63;
    const int nblk = (nrhs + 63) >> 6;
    const int64_t w = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (w >= (int64_t)count * nblk) return;
    const int32_t row = __builtin_amdgcn_readfirstlane(order[first + w / nblk]);
    solve_row_rhs64<false>(row, (int)(w % nblk), nrhs, ptr, idx, val, diag, skip_first, skip_last, X, lane, nullptr, nullptr, 0,
                           nullptr);
}

__global__ __launch_bounds__(1024) void k_tri_levels_rows64_one_wg(const int32_t *__restrict__ order,
                                                                   const int32_t *__restrict__ level_ptr, int32_t l0,
                                                                   int32_t l1, const int32_t *__restrict__ ptr,
                                                                   const int32_t *__restrict__ idx,
                                                                   const double *__restrict__ val,
                                                                   const double *__restrict__ diag, int skip_first,
                                                                   int skip_last, double *X, int nrhs,
                                                                   const int32_t *__restrict__ resume) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nblk = (nrhs + 63) >> 6;
    for (int32_t l = l0; l < l1; l++) {
        const int32_t first = level_ptr[l], count = level_ptr[l + 1] - first;
        for (int32_t t = w; t < count * nblk; t += 16) {
            const int32_t row = __builtin_amdgcn_readfirstlane(order[first + t / nblk]);
            solve_row_rhs64<false>(row, t % nblk, nrhs, ptr, idx, val, diag, skip_first, skip_last, X, lane, resume, nullptr, 0,
                                   nullptr);
        }
        __syncthreads();  // workgroup-scope release/acquire: the next level reads these x
    }
}

template <int KIND>
__global__ __launch_bounds__(TC_THREADS) void k_tri_columns(int32_t n, const int32_t *__restrict__ Tp,
                                                            const int32_t *__restrict__ Ti, const double *__restrict__ Tx,
                                                            double *X, int nrhs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // n doubles
    const int tid = threadIdx.x, r = blockIdx.x;
    for (int32_t i = tid; i < n; i += TC_THREADS) xs[i] = X[(int64_t)i * nrhs + r];
    __syncthreads();
    static_assert(KIND == CSX_TRI_L || KIND == CSX_TRI_U, "the gather kinds run k_tri_colchain");
    constexpr bool ASC = KIND == CSX_TRI_L;
    constexpr bool DIAG_FIRST = KIND == CSX_TRI_L;
    // A ring of four columns in registers, the loop unrolled by four so that every slot has a fixed name: a column's
    // entries are requested three steps before they are used (see k_tri_colchain), and the barrier waits for LDS only.
    auto col_at = [&](int32_t step) {   // the column of a step, clamped to the last one
        const int32_t st = step < n ? step : n - 1;
        return ASC ? st : n - 1 - st;
    };
    struct Col {
        int32_t lo, hi, ci;
        double dg, cv;
    };
    auto load = [&](int32_t b, int32_t e) {
        Col c;
        c.lo = DIAG_FIRST ? b + 1 : b;
        c.hi = DIAG_FIRST ? e : e - 1;
        c.dg = Tx[DIAG_FIRST ? b : e - 1];
        c.ci = c.lo + tid < c.hi ? Ti[c.lo + tid] : 0;
        c.cv = c.lo + tid < c.hi ? Tx[c.lo + tid] : 0.0;
        return c;
    };
    Col ring[4];
    int32_t pb[4], pe[4];
#pragma unroll
    for (int u = 0; u < 3; u++) ring[u] = load(Tp[col_at(u)], Tp[col_at(u) + 1]);
    pb[3] = Tp[col_at(3)];
    pe[3] = Tp[col_at(3) + 1];
    for (int32_t base = 0; base < n; base += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t step = base + u;
            if (step >= n) break;                          // uniform
            const int32_t j = col_at(step), j4 = col_at(step + 4);
            pb[u] = Tp[j4];
            pe[u] = Tp[j4 + 1];
            ring[(u + 3) & 3] = load(pb[(u + 3) & 3], pe[(u + 3) & 3]);
            const Col &c = ring[u];
            const double xj = xs[j] / c.dg;                    // every thread: same operands, same result
            if (tid == 0) X[(int64_t)j * nrhs + r] = xj;       // x[j] is final and not read again
            if (c.lo + tid < c.hi) {
                const double t = c.cv * xj;
                xs[c.ci] = xs[c.ci] - t;
            }
            for (int32_t p = c.lo + tid + TC_THREADS; p < c.hi; p += TC_THREADS) {   // columns longer than the workgroup
                const int32_t i = Ti[p];
                const double t = Tx[p] * xj;
                xs[i] = xs[i] - t;
            }
            lds_barrier();
        }
    }
}

// ---- chain-like systems of ANY size whose band fits LDS: the same two loops on a WINDOW of x -------------------
// A grid Laplacian in natural order factors into a chain elimination tree and a band as wide as the grid (n = 490 000,
// half-width 700, lnz 3.4e8): no level schedule has anything to schedule, and x (3.9 MB) does not fit LDS.  But a
// column only reaches `band` rows ahead, so a circular window of the next Wn > band + 2 rows does: slot = row & (Wn - 1).
// Push kinds (k_tri_wcolumns): the row that enters the window when column j leaves it takes over slot j (fetched three
// steps ahead by thread 0, stored after the step's barrier).  Gather kinds (k_tri_wcolchain): x[j] is written into its
// own slot when it is formed; the right-hand side value comes from memory with the column's entries.
// k_tri_wcolumns takes the matrix as (ptr, idx, val) + a diagonal array + how many leading / trailing entries of a
// column to skip, so it also runs the ROWS of L (the forward plan's gather arrays, diagonal stripped) as the columns of
// L': the backward solve in push form -- x[i] then loses its terms in descending instead of ascending source order, so
// that is used for the rounding-equal order only (csx_cholsol_set_order(plan, 0)).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_tri_wcolumns(int32_t n, const int32_t *__restrict__ Tp,
                                                          const int32_t *__restrict__ Ti, const double *__restrict__ Tx,
                                                          const double *__restrict__ diag, int skip_first, int skip_last,
                                                          int asc, uint32_t mask, double *X, int nrhs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // mask + 1 doubles
    const int tid = threadIdx.x, r = blockIdx.x;
    const int32_t Wn = (int32_t)mask + 1;
    for (int32_t i = tid; i < min(n, Wn); i += THREADS) {
        const int32_t row = asc ? i : n - 1 - i;
        xs[row & mask] = X[(int64_t)row * nrhs + r];
    }
    __syncthreads();
    auto col_at = [&](int32_t step) {   // the column of a step, clamped to the last one
        const int32_t st = step < n ? step : n - 1;
        return asc ? st : n - 1 - st;
    };
    struct Col {
        int32_t lo, hi, ci;
        double dg, cv, xin;
    };
    auto load = [&](int32_t j, int32_t b, int32_t e) {
        Col c;
        c.lo = b + skip_first;
        c.hi = e - skip_last;
        c.dg = diag[j];
        c.ci = c.lo + tid < c.hi ? Ti[c.lo + tid] : 0;
        c.cv = c.lo + tid < c.hi ? Tx[c.lo + tid] : 0.0;
        const int32_t rin = asc ? j + Wn : j - Wn;       // the row that takes over slot j
        c.xin = (tid == 0 && rin >= 0 && rin < n) ? X[(int64_t)rin * nrhs + r] : 0.0;
        return c;
    };
    Col ring[4];
    int32_t pb[4], pe[4];
#pragma unroll
    for (int u = 0; u < 3; u++) ring[u] = load(col_at(u), Tp[col_at(u)], Tp[col_at(u) + 1]);
    pb[3] = Tp[col_at(3)];
    pe[3] = Tp[col_at(3) + 1];
    for (int32_t base = 0; base < n; base += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t step = base + u;
            if (step >= n) break;                          // uniform
            const int32_t j = col_at(step), j4 = col_at(step + 4);
            pb[u] = Tp[j4];
            pe[u] = Tp[j4 + 1];
            ring[(u + 3) & 3] = load(col_at(step + 3), pb[(u + 3) & 3], pe[(u + 3) & 3]);
            const Col &c = ring[u];
            const double xj = xs[j & mask] / c.dg;             // every thread: same operands, same result
            if (tid == 0) X[(int64_t)j * nrhs + r] = xj;       // x[j] is final and not read again
            if (c.lo + tid < c.hi) {
                const double t = c.cv * xj;
                xs[c.ci & mask] = xs[c.ci & mask] - t;
            }
            for (int32_t p = c.lo + tid + THREADS; p < c.hi; p += THREADS) {   // columns longer than the workgroup
                const int32_t i = Ti[p];
                const double t = Tx[p] * xj;
                xs[i & mask] = xs[i & mask] - t;
            }
            lds_barrier();
            const int32_t rin = asc ? j + Wn : j - Wn;
            if (tid == 0 && rin >= 0 && rin < n) xs[j & mask] = c.xin;   // seen after the next barrier, needed later still
        }
    }
}

template <int ROUNDS>
struct TchWide {   // a column's off-diagonal range, its diagonal, the right-hand side value and ROUNDS entries per lane
    int32_t lo, hi;
    double dg, b;
    int32_t ci[ROUNDS];
    double cv[ROUNDS];
};

template <int KIND, int ROUNDS>   // CSX_TRI_LT or CSX_TRI_UT; columns of up to 64 * ROUNDS entries are fetched ahead
__global__ __launch_bounds__(64) void k_tri_wcolchain(int32_t n, const int32_t *__restrict__ Tp,
                                                      const int32_t *__restrict__ Ti, const double *__restrict__ Tx,
                                                      uint32_t mask, double *X, int nrhs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // mask + 1 doubles: x of the last mask + 1 columns
    __shared__ __attribute__((aligned(16))) double cbuf[64];
    const int lane = threadIdx.x, r = blockIdx.x;
    constexpr bool ASC = KIND == CSX_TRI_UT;
    constexpr bool DIAG_FIRST = KIND == CSX_TRI_LT;
    auto col_at = [&](int32_t step) {
        const int32_t st = step < n ? step : n - 1;
        return ASC ? st : n - 1 - st;
    };
    auto load = [&](int32_t j, int32_t b, int32_t e) {
        TchWide<ROUNDS> c;
        c.lo = DIAG_FIRST ? b + 1 : b;
        c.hi = DIAG_FIRST ? e : e - 1;
        c.dg = Tx[DIAG_FIRST ? b : e - 1];
        c.b = X[(int64_t)j * nrhs + r];
#pragma unroll
        for (int q = 0; q < ROUNDS; q++) {
            const int32_t p = c.lo + 64 * q + lane;
            c.ci[q] = p < c.hi ? Ti[p] : 0;
            c.cv[q] = p < c.hi ? Tx[p] : 0.0;
        }
        return c;
    };
    TchWide<ROUNDS> ring[4];
    int32_t pb[4], pe[4];
#pragma unroll
    for (int u = 0; u < 3; u++) ring[u] = load(col_at(u), Tp[col_at(u)], Tp[col_at(u) + 1]);
    pb[3] = Tp[col_at(3)];
    pe[3] = Tp[col_at(3) + 1];
    for (int32_t base = 0; base < n; base += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t step = base + u;
            if (step >= n) break;                       // uniform
            const int32_t j = col_at(step);
            const int32_t j4 = col_at(step + 4);
            pb[u] = Tp[j4];
            pe[u] = Tp[j4 + 1];
            ring[(u + 3) & 3] = load(col_at(step + 3), pb[(u + 3) & 3], pe[(u + 3) & 3]);
            const TchWide<ROUNDS> &cur = ring[u];
            const int32_t len = cur.hi - cur.lo;
            double acc = cur.b;
#pragma unroll
            for (int q = 0; q < ROUNDS; q++) {
                if (64 * q >= len) break;                // uniform
                const double pq = 64 * q + lane < len ? cur.cv[q] * xs[cur.ci[q] & mask] : 0.0;
                acc = tch_chain_lds(acc, pq, len - 64 * q, cbuf, lane);
            }
            for (int32_t q0 = 64 * ROUNDS; q0 < len; q0 += 64) {   // columns longer than what is fetched ahead
                const double pq = q0 + lane < len ? Tx[cur.lo + q0 + lane] * xs[Ti[cur.lo + q0 + lane] & mask] : 0.0;
                acc = tch_chain_lds(acc, pq, len - q0, cbuf, lane);
            }
            const double xj = acc / cur.dg;                   // lane 0's is the one
            if (lane == 0) {
                xs[j & mask] = xj;
                X[(int64_t)j * nrhs + r] = xj;
            }
        }
    }
}

// band half-width of T, the number of columns with an entry right next to the diagonal, and whether T is a proper
// triangle of its kind (diagonal first in L, last in U, every other entry strictly on its side): out[0..2]
__global__ __launch_bounds__(256) void k_tri_band(int32_t n, const int32_t *__restrict__ Tp, const int32_t *__restrict__ Ti,
                                                  int lower, int *out) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int32_t far = 0, links = 0, bad = 0;             // per wave, over its columns (one atomic each at the end)
    for (int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; j < n; j += nwaves) {
        int32_t link = 0;
        const int32_t b = Tp[j], e = Tp[j + 1];
        for (int32_t p = b + lane; p < e; p += 64) {
            const int32_t d = Ti[p] - (int32_t)j;
            const bool is_diag = lower ? p == b : p == e - 1;
            bad |= is_diag ? d != 0 : (lower ? d <= 0 : d >= 0);
            far = max(far, abs(d));
            link |= abs(d) == 1;
        }
        links += __ballot(link) != 0ull;
    }
    for (int o = 32; o > 0; o >>= 1) {
        far = max(far, __shfl_xor(far, o));
        bad |= __shfl_xor(bad, o);
    }
    if (lane == 0) {
        atomicMax(out, far);
        if (links) atomicAdd(out + 1, links);
        if (bad) out[2] = 1;
    }
}
#pragma clang fp contract(fast)

// a column of T with two entries in the same row (the reference's own LU factors have them) must keep their
// order: the push kernels above would race on x[row].  G = stable transpose: such entries are neighbours in a row.
__global__ __launch_bounds__(256) void k_adjacent_equal(int32_t n, const int32_t *__restrict__ ptr,
                                                        const int32_t *__restrict__ idx, int *flag) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n) return;
    for (int32_t q = ptr[r] + 1 + lane; q < ptr[r + 1]; q += 64)
        if (idx[q] == idx[q - 1]) *flag = 1;
}

constexpr int COMP_MAX_ROWS = 256;    // X tile of one component: rows x 64 lanes x 8 B <= 128 KiB of LDS
constexpr int COMP_MIN_COUNT = 64;    // fewer components than this: level scheduling fills the chip better

static size_t few_lds_bytes(int32_t rows, int32_t terms, int G) {
    return (size_t)(terms > 0 ? terms : 1) * 16 + (size_t)rows * (16 + 8 * (size_t)G) + 64;
}

// Returns CSX_OK whether or not the component path applies; P->comp_ok says which.
static int analyse_components(TriPlan *P) {
    if (P->comp_tried) return CSX_OK;
    P->comp_tried = true;
    const int32_t n = P->n;
    if (n < COMP_MIN_COUNT) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *root = nullptr, *local_id = nullptr, *len = nullptr, *comp_of_pos = nullptr;
    uint32_t *srow = nullptr;
    int *flags = nullptr;
    CSX_TRY(tmp.alloc(&root, (size_t)n));
    CSX_TRY(tmp.alloc(&flags, 4));
    bool malformed = false;
    CSX_TRY(connected_components(n, P->ptr, P->idx, P->skip_first, P->skip_last, P->forward ? 1 : 2, root, &malformed));
    if (malformed) return CSX_OK;         // the literal transcription of the reference loop handles it
    Tree *comps = nullptr;
    int32_t ncomp = 0, maxc = 0;
    CSX_TRY(tmp.alloc(&srow, (size_t)n));
    CSX_TRY(tmp.alloc(&comp_of_pos, (size_t)n));
    CSX_TRY(group_by_root(n, root, srow, comp_of_pos, &comps, &ncomp, &maxc));
    P->comps = comps;
    if (ncomp < COMP_MIN_COUNT || maxc > COMP_MAX_ROWS) return CSX_OK;
    const unsigned nbw = (unsigned)(((int64_t)n + 3) / 4);
    int hflags[4] = {maxc, 0, 0, 0};
    P->ncomp = (int32_t)ncomp;
    P->comp_max = hflags[0];
    CSX_TRY(tmp.alloc(&local_id, (size_t)n));
    CSX_TRY(tmp.alloc(&len, (size_t)n + 1));
    const unsigned ncw = (unsigned)((ncomp + 3) / 4);
    hipLaunchKernelGGL(k_cc_local_id, dim3(ncw), dim3(256), 0, s, P->ncomp, comps, srow, local_id);
    hipLaunchKernelGGL(k_prog_len, dim3(ncw), dim3(256), 0, s, P->ncomp, comps, srow, P->ptr, P->skip_first, P->skip_last,
                       P->forward ? 1 : 0, len);
    CSX_TRY(dalloc(&P->prog_ptr, (size_t)n + 1));
    int64_t total = 0;
    CSX_TRY(scan_exclusive_i32(len, P->prog_ptr, n, &total));
    CSX_TRY(dalloc(&P->prog_idx, (size_t)total + 64));
    CSX_TRY(dalloc(&P->prog_val, (size_t)total + 64));
    CSX_TRY(dalloc(&P->prog_diag, (size_t)n));
    CSX_TRY(dalloc(&P->comp_nodes, (size_t)n));
    hipLaunchKernelGGL(k_prog_fill, dim3(nbw), dim3(256), 0, s, n, comp_of_pos, comps, srow, local_id, P->ptr, P->idx, P->val,
                       P->diag, P->skip_first, P->forward ? 1 : 0, P->prog_ptr, P->prog_idx, P->prog_val, P->prog_diag);
    CSX_HIP(hipMemcpyAsync(P->comp_nodes, srow, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    CSX_LAUNCH_CHECK();
    if (P->kind == CSX_TRI_L || P->kind == CSX_TRI_U) {
        // column program for k_tri_comp_push -- unless some column holds the same row twice (the reference's own LU
        // factors do): the lanes of a column would race on it; the gather kernels keep such entries in order
        int hdup = 0;
        CSX_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_adjacent_equal, dim3(nbw), dim3(256), 0, s, n, P->ptr, P->idx, flags);
        CSX_HIP(hipMemcpyAsync(&hdup, flags, sizeof(int), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        if (!hdup) {
            int32_t *clen = nullptr;
            CSX_TRY(tmp.alloc(&clen, (size_t)n + 1));
            hipLaunchKernelGGL(k_push_len, dim3(ncw), dim3(256), 0, s, P->ncomp, comps, srow, P->Tp, P->forward ? 1 : 0, clen);
            CSX_TRY(dalloc(&P->cptr, (size_t)n + 1));
            int64_t ctotal = 0;
            CSX_TRY(scan_exclusive_i32(clen, P->cptr, n, &ctotal));
            CSX_TRY(dalloc(&P->cidx, (size_t)ctotal + 64));
            CSX_TRY(dalloc(&P->cval, (size_t)ctotal + 64));
            CSX_TRY(dalloc(&P->cdiag, (size_t)n));
            CSX_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), s));
            hipLaunchKernelGGL(k_push_fill, dim3(nbw), dim3(256), 0, s, n, comp_of_pos, comps, srow, local_id, P->Tp, P->Ti,
                               P->Tx, P->forward ? 1 : 0, P->cptr, P->cidx, P->cval, P->cdiag, flags);
            int hmax = 0;
            CSX_HIP(hipMemcpyAsync(&hmax, flags, sizeof(int), hipMemcpyDeviceToHost, s));
            CSX_HIP(hipStreamSynchronize(s));
            P->push_terms = hmax > 0 ? hmax : 1;
        }
    }
    for (int cpw : {4}) {   // LDS need of the all-in-LDS kernel: 12 B per term, (12 + 8 G) B per row, G = 64 / cpw = 16
        int hcaps[2] = {0, 0};
        CSX_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), s));
        const int64_t groups = (ncomp + cpw - 1) / cpw;
        hipLaunchKernelGGL(k_group_caps, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, P->ncomp, cpw, comps,
                           P->prog_ptr, flags);
        CSX_HIP(hipMemcpyAsync(hcaps, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        const size_t need = few_lds_bytes(hcaps[0], hcaps[1], 64 / cpw);
        if (need <= 120 * 1024) {
            P->few_cpw = cpw;
            P->few_rows = hcaps[0];
            P->few_terms = hcaps[1];
            break;
        }
    }
    // The sweeps of many right-hand sides (k_tri_local) launch one single-wave workgroup per (component, chunk).  Workgroups are dealt
    // to the XCDs and their shader engines in a fixed rotation, so a PERIODIC pattern of work in launch order lands on the hardware as
    // an imbalance: W's U is a 65-row component and a singleton per block -- sixteen big workgroups, sixteen tiny ones, and so on: half
    // of the engines got the big ones and its sweep took exactly twice its L's (2.30 against 1.14 ms), with or without the singletons'
    // work.  Biggest first (stable): 1.28 ms.
    CSX_TRY(trees_biggest_first(comps, P->ncomp, maxc, &P->comps_by_size));
    CSX_HIP(hipStreamSynchronize(s));
    P->comp_ok = true;
    return CSX_OK;
}

// the components made dense for the matrix cores, the first time the rounding-equal order asks (P->rag stays null when the guard refuses)
static int components_ragged(TriPlan *P) {
    if (P->rag_tried) return CSX_OK;
    P->rag_tried = true;
    RaggedMfma *R = nullptr;
    CSX_TRY(ragged_build(P->comps, P->ncomp, P->comp_max, P->comp_nodes, P->prog_ptr, P->prog_idx, P->prog_val, P->prog_diag, !P->forward, &R));
    if (R) {
        P->rag_growth = R->growth;
        if (R->growth <= RAG_GROWTH_LIMIT) P->rag = R;     // (a NaN fails the comparison)
        else ragged_free(R);
    }
    return CSX_OK;
}

// io (csx_lusol_solve, exact order): the block read and its row map, the row map of the block written (X); honoured by the kernel
// of MANY right-hand sides only -- *io_taken says whether (false: nothing was launched, the caller runs the separate steps)
struct TriIO {
    const double *src;
    const int32_t *load_rows, *store_rows;
};
static int solve_components(TriPlan *P, double *X, int32_t nrhs, const TriIO *io = nullptr, bool *io_taken = nullptr) {
    hipStream_t s = ctx().stream;
    if (io) {
        *io_taken = false;
        if ((P->rounding_equal && nrhs > 8) || nrhs <= 32) return CSX_OK;      // (other kernels' territory)
    }
    if (P->rounding_equal && nrhs > 8) {
        // the caller granted rounding (csx_tri_set_order): dense components on the matrix cores, one sweep in position order
        CSX_TRY(components_ragged(P));
        if (P->rag) return ragged_solve(P->rag, P->comp_nodes, nullptr, !P->forward, 1, X, nrhs, P->n);
    }
    // L, U with up to 8 right-hand sides: one wave per component, column-push form (W, L + U pair: 43 us at 1 RHS,
    // 78 us at 8; at 64 the entry-parallel lanes are gone and it loses to k_tri_local, 483 against 272 us)
    if (nrhs <= 8 && P->push_terms > 0 && ctx().opt.tri_push) {
        int G = 1;
        while (G < nrhs) G <<= 1;
        const size_t lds = (size_t)P->push_terms * 16 + (size_t)P->comp_max * (16 + 8 * (size_t)G) + 64;
        if (lds <= 120 * 1024) {
            if (P->forward) {
                CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_comp_push<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                hipLaunchKernelGGL(k_tri_comp_push<true>, dim3((unsigned)P->ncomp), dim3(64), lds, s, P->comps, P->ncomp,
                                   P->comp_nodes, P->cptr, P->cidx, P->cval, P->cdiag, X, nrhs, G, P->comp_max, P->push_terms);
            } else {
                CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_comp_push<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                hipLaunchKernelGGL(k_tri_comp_push<false>, dim3((unsigned)P->ncomp), dim3(64), lds, s, P->comps, P->ncomp,
                                   P->comp_nodes, P->cptr, P->cidx, P->cval, P->cdiag, X, nrhs, G, P->comp_max, P->push_terms);
            }
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
    }
    if (nrhs <= 32 && P->few_cpw) {   // lanes = (component, right-hand side) pairs, programs and X in LDS
        const int cpw = P->few_cpw;
        const size_t lds = few_lds_bytes(P->few_rows, P->few_terms, 64 / cpw);
        const dim3 grid((unsigned)((P->ncomp + cpw - 1) / cpw));
#define CSX_FEW(FWD, CPW)                                                                                            \
    hipLaunchKernelGGL((k_tri_few_lds<FWD, CPW>), grid, dim3(64), lds, s, P->comps, P->ncomp, P->comp_nodes, P->prog_ptr, \
                       P->prog_idx, P->prog_val, P->prog_diag, X, nrhs, P->few_rows, P->few_terms)
        if (P->forward) {
            CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_few_lds<true, 4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            CSX_FEW(true, 4);
        } else {
            CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_few_lds<false, 4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            CSX_FEW(false, 4);
        }
#undef CSX_FEW
        CSX_LAUNCH_CHECK();
        return CSX_OK;
    }
    if (nrhs <= 32) {   // the same with the programs read from memory (components too big for the LDS copy)
        int G = 1;
        while (G < nrhs) G <<= 1;
        const int cpw = 64 / G;
        const int32_t tile_rows = P->comp_max | 1;
        const size_t per_wave_f = (size_t)64 * tile_rows * sizeof(double);
        int wv = (int)std::min<size_t>(4, (128 * 1024) / per_wave_f);
        if (wv < 1) wv = 1;
        const int64_t waves_needed = ((int64_t)P->ncomp + cpw - 1) / cpw;
        const dim3 grid((unsigned)((waves_needed + wv - 1) / wv));
        if (P->forward) {
            CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_few<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            hipLaunchKernelGGL(k_tri_few<true>, grid, dim3(256), per_wave_f * wv, s, P->comps, P->ncomp, P->comp_nodes,
                               P->prog_ptr, P->prog_idx, P->prog_val, P->prog_diag, X, nrhs, G, tile_rows, wv);
        } else {
            CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_few<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            hipLaunchKernelGGL(k_tri_few<false>, grid, dim3(256), per_wave_f * wv, s, P->comps, P->ncomp, P->comp_nodes,
                               P->prog_ptr, P->prog_idx, P->prog_val, P->prog_diag, X, nrhs, G, tile_rows, wv);
        }
        CSX_LAUNCH_CHECK();
        return CSX_OK;
    }
    const size_t per_wave = (size_t)P->comp_max * 64 * sizeof(double);
    const int waves = tile_waves_per_workgroup(per_wave, TL_WAVES_MAX);
    const int32_t chunks = (nrhs + 63) / 64;
    const int64_t tasks = (int64_t)P->ncomp * chunks;
    const size_t lds = per_wave * (size_t)waves;
    const dim3 grid((unsigned)((tasks + waves - 1) / waves));
    if (P->forward) {
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_local<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        hipLaunchKernelGGL(k_tri_local<true>, grid, dim3(64 * waves), lds, s, P->comps_by_size, P->ncomp, P->comp_nodes,
                           P->prog_ptr, P->prog_idx, P->prog_val, P->prog_diag, io ? io->src : (const double *)X, X,
                           io ? io->load_rows : nullptr, io ? io->store_rows : nullptr, nrhs, chunks, P->comp_max, waves);
    } else {
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_local<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        hipLaunchKernelGGL(k_tri_local<false>, grid, dim3(64 * waves), lds, s, P->comps_by_size, P->ncomp, P->comp_nodes,
                           P->prog_ptr, P->prog_idx, P->prog_val, P->prog_diag, io ? io->src : (const double *)X, X,
                           io ? io->load_rows : nullptr, io ? io->store_rows : nullptr, nrhs, chunks, P->comp_max, waves);
    }
    CSX_LAUNCH_CHECK();
    if (io) *io_taken = true;
    return CSX_OK;
}

// ---- analysis ---------------------------------------------------------------------
static int download_i32(std::vector<int32_t> &h, const int32_t *d, size_t count) {
    h.resize(count);
    if (count) CSX_HIP(hipMemcpyAsync(h.data(), d, count * sizeof(int32_t), hipMemcpyDeviceToHost, ctx().stream));
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    return CSX_OK;
}

// Level of every unknown from the gather structure.  forward: inputs have smaller
// index.  Returns false if some input does not precede its unknown (malformed).
static bool compute_levels(int32_t n, const std::vector<int32_t> &ptr, const std::vector<int32_t> &idx, int sf, int sl,
                           bool forward, std::vector<int32_t> &level, int32_t &nlevels) {
    level.assign((size_t)n, 0);
    nlevels = 0;
    for (int32_t s = 0; s < n; s++) {
        const int32_t r = forward ? s : n - 1 - s;
        int32_t lv = 0;
        for (int32_t q = ptr[r] + sf; q < ptr[r + 1] - sl; q++) {
            const int32_t j = idx[q];
            if (forward ? j >= r : j <= r) return false;
            lv = std::max(lv, level[j] + 1);
        }
        level[r] = lv;
        nlevels = std::max(nlevels, lv + 1);
    }
    return true;
}

static int analyse(const Csc *T, int kind, TriPlan **out) {
    hipStream_t s = ctx().stream;
    const int32_t n = T->n;
    TriPlan *P = new TriPlan();
    *out = P;
    P->kind = kind;
    P->n = n;
    P->Tp = T->p;
    P->Ti = T->i;
    P->Tx = T->x;
    if (n == 0) return CSX_OK;
    int *flag = nullptr;
    CSX_TRY(dalloc(&flag, 2));
    CSX_HIP(hipMemsetAsync(flag, 0, 2 * sizeof(int), s));
    const unsigned nb = (unsigned)(((int64_t)n + 255) / 256);
    hipLaunchKernelGGL(k_check_columns, dim3(nb), dim3(256), 0, s, n, T->p, flag);
    int hflag[2] = {0, 0};
    CSX_HIP(hipMemcpyAsync(hflag, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (hflag[0]) {  // a column without entries has no diagonal: the reference would index past it
        dfree(flag);
        return CSX_EINVAL;
    }
    CSX_TRY(dalloc(&P->diag, (size_t)n));
    const bool forward = (kind == CSX_TRI_L || kind == CSX_TRI_UT);
    if (kind == CSX_TRI_LT || kind == CSX_TRI_UT) {
        P->ptr = T->p;
        P->idx = T->i;
        P->val = T->x;
        P->skip_first = kind == CSX_TRI_LT ? 1 : 0;
        P->skip_last = kind == CSX_TRI_UT ? 1 : 0;
        hipLaunchKernelGGL(k_diag_direct, dim3(nb), dim3(256), 0, s, n, T->p, T->x, kind == CSX_TRI_UT ? 1 : 0, P->diag);
    } else {
        Csc S;  // stripped copy
        S.m = S.n = n;
        S.nnz = T->nnz - n;
        int st = dalloc(&S.p, (size_t)n + 1);
        if (st == CSX_OK) st = dalloc(&S.i, (size_t)S.nnz);
        if (st == CSX_OK) st = dalloc(&S.x, (size_t)S.nnz);
        if (st == CSX_OK) {
            int64_t blocks = std::min<int64_t>(((int64_t)n + 4) / 4, 65536);
            const bool short_cols = (int64_t)T->nnz < 8 * (int64_t)n;
            if (kind == CSX_TRI_L && short_cols)
                hipLaunchKernelGGL(k_strip_first<4>, dim3((unsigned)std::min<int64_t>(((int64_t)n + 64) / 64, 65536)), dim3(256), 0, s, n,
                                   T->p, T->i, T->x, S.p, S.i, S.x, P->diag);
            else if (kind == CSX_TRI_L)
                hipLaunchKernelGGL(k_strip_first<64>, dim3((unsigned)blocks), dim3(256), 0, s, n, T->p, T->i, T->x, S.p, S.i,
                                   S.x, P->diag);
            else
                hipLaunchKernelGGL(k_strip_last_reverse, dim3((unsigned)blocks), dim3(256), 0, s, n, T->p, T->i, T->x,
                                   S.p, S.i, S.x, P->diag);
            if (hipGetLastError() != hipSuccess) st = CSX_ERUNTIME;
        }
        Csc G;
        if (st == CSX_OK) st = transpose_device(&S, true, &G);
        dfree(S.p);
        dfree(S.i);
        dfree(S.x);
        if (st != CSX_OK) {
            dfree(G.p);
            dfree(G.i);
            dfree(G.x);
            dfree(flag);
            return st;
        }
        P->ptr = G.p;
        P->idx = G.i;
        P->val = G.x;
        P->owns_g = true;
        if (kind == CSX_TRI_U && G.nnz > 0) {
            hipLaunchKernelGGL(k_reverse_index, dim3((unsigned)(((int64_t)G.nnz + 255) / 256)), dim3(256), 0, s,
                               (int64_t)G.nnz, n, P->idx);
        }
    }
    hipLaunchKernelGGL(k_any_zero, dim3(nb), dim3(256), 0, s, n, P->diag, flag + 1);
    CSX_HIP(hipMemcpyAsync(hflag, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    dfree(flag);
    P->zero_pivot = hflag[1] != 0;
    P->gnnz = (kind == CSX_TRI_LT || kind == CSX_TRI_UT) ? T->nnz : T->nnz - n;
    P->forward = forward;
    return CSX_OK;
}


// ---- level sets and chain-walker tables on the device ----------------------------------------------------
// For big factors the host analysis below starts with a copy of the whole gather structure (2 GB over PCIe at
// lnz = 1.6e8) and then walks it on one core.  On the device: level by level, every unresolved row looks at its
// sources; it joins level L when all of them are in levels < L (pull form: only the gather structure the plan
// already has; a row at level l is inspected l + 1 times, which is cheap for the wide, shallow graphs big
// factors have -- deep chains of small factors stay with the host pass, ensure_schedule picks by size).
__global__ __launch_bounds__(256) void k_lv_init(int32_t n, int32_t *level) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) level[r] = -1;
}

// one wave per row; stats[0] += rows resolved in this round, stats[1] |= malformed (a source does not precede)
__global__ __launch_bounds__(256) void k_lv_round(int32_t n, const int32_t *__restrict__ ptr,
                                                  const int32_t *__restrict__ idx, int sf, int sl, int forward,
                                                  int32_t L, int32_t *level, int *stats) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n) return;
    if (level[r] >= 0) return;                              // wave-uniform
    const int32_t b = ptr[r] + sf, e = ptr[r + 1] - sl;
    bool ready = true, bad = false;
    for (int32_t q = b + lane; q < e; q += 64) {
        const int32_t j = idx[q];
        if (j < 0 || j >= n || (forward ? j >= r : j <= r)) {
            bad = true;
            continue;
        }
        const int32_t lj = level[j];
        if (lj < 0 || lj >= L) ready = false;               // unresolved, or resolved in THIS round by another wave
    }
    if (__ballot(bad) != 0ull) {
        if (lane == 0) stats[1] = 1;
        return;
    }
    if (__ballot(!ready) == 0ull && lane == 0) {
        level[r] = L;
        atomicAdd(&stats[0], 1);
    }
}

__global__ __launch_bounds__(256) void k_iota_u32_tri(int32_t n, uint32_t *v) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) v[r] = (uint32_t)r;
}

__global__ __launch_bounds__(256) void k_lv_pos_of(int32_t n, const uint32_t *__restrict__ order, int32_t *pos_of) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) pos_of[order[q]] = (int32_t)q;
}

// the tables of the blocked chain walker (see k_tri_chain): one thread per position of the level order
__global__ __launch_bounds__(256) void k_lv_chain_tables(int32_t n, const uint32_t *__restrict__ order,
                                                         const int32_t *__restrict__ pos_of,
                                                         const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                         int sf, int sl, int32_t *npre, int32_t *nin, int8_t *tslot,
                                                         unsigned long long *sums) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long suffix = 0, all = 0;
    if (q < n) {
        const int32_t r = (int32_t)order[q], blk = (int32_t)q & ~(CHB - 1);
        const int32_t b = ptr[r] + sf, e = ptr[r + 1] - sl;
        int32_t pre = 0, inb = 0;
        bool in_prefix = true;
        for (int32_t t = b; t < e; t++) {
            const int32_t sp = pos_of[idx[t]];
            const bool inside = sp >= blk;
            tslot[t] = inside ? (int8_t)(sp - blk) : (int8_t)-1;
            if (inside) {
                in_prefix = false;
                inb++;
            }
            if (in_prefix) pre++;
        }
        npre[q] = pre;
        nin[q] = inb;
        suffix = (unsigned long long)((e - b) - pre);
        all = (unsigned long long)(e - b);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        suffix += __shfl_xor(suffix, d, 64);
        all += __shfl_xor(all, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && all) {
        atomicAdd(&sums[0], suffix);
        atomicAdd(&sums[1], all);
    }
}

constexpr int64_t LV_DEVICE_MIN_TERMS = 4 << 20;   // below this the copy is small and deep chains favour the host pass
constexpr int LV_MAX_ROUNDS = 512;     // a round costs ~80 us whatever it finds: deeper than this, one sequential host pass wins

// Every term's source must lie in a lower level than its row (and on the proper side of the diagonal): what makes a
// proposed level array a valid schedule, whether or not it is the shallowest one.  stats[0]: rows with a violation.
__global__ __launch_bounds__(256) void k_lv_verify(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                   int sf, int sl, int forward, const int32_t *__restrict__ level, int *stats) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n) return;
    const int32_t lr = level[r];
    bool bad = lr < 0;
    for (int32_t q = ptr[r] + sf + lane; q < ptr[r + 1] - sl; q += 64) {
        const int32_t j = idx[q];
        if (j < 0 || j >= n || (forward ? j >= r : j <= r)) bad = true;
        else if (level[j] >= lr) bad = true;
    }
    if (__ballot(bad) != 0ull && lane == 0) stats[0] = 1;
}

// rows by level, level boundaries and the chain walker's tables from a device array of levels (L of them)
static int schedule_from_levels(TriPlan *P, const int32_t *level, int32_t L) {
    hipStream_t s = ctx().stream;
    const int32_t n = P->n;
    DevScope tmp;
    int32_t *pos_of = nullptr;
    uint32_t *rows = nullptr, *slevel = nullptr, *order = nullptr;
    unsigned long long *sums = nullptr;
    const unsigned nb = (unsigned)(((int64_t)n + 255) / 256);
    P->nlevels = L;
    // rows by level, ascending row inside a level (stable sort of 0..n-1 by level)
    CSX_TRY(tmp.alloc(&rows, (size_t)n));
    CSX_TRY(tmp.alloc(&slevel, (size_t)n));
    CSX_TRY(dalloc(&order, (size_t)n));
    P->order = (int32_t *)order;
    hipLaunchKernelGGL(k_iota_u32_tri, dim3(nb), dim3(256), 0, s, n, rows);
    CSX_TRY(stable_sort_by_key((const uint32_t *)level, rows, nullptr, n, (uint32_t)L, slevel, order, nullptr));
    CSX_TRY(dalloc(&P->level_ptr, (size_t)L + 1));
    CSX_TRY(boundaries_from_sorted(slevel, n, L, P->level_ptr));
    P->level_ptr_h.resize((size_t)L + 1);
    CSX_HIP(hipMemcpyAsync(P->level_ptr_h.data(), P->level_ptr, ((size_t)L + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    // chain-walker tables
    CSX_TRY(tmp.alloc(&pos_of, (size_t)n));
    CSX_TRY(tmp.alloc(&sums, 2));
    CSX_TRY(dalloc(&P->npre, (size_t)n));
    CSX_TRY(dalloc(&P->nin, (size_t)n));
    CSX_TRY(dalloc(&P->tslot, (size_t)P->gnnz + 1));
    CSX_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_lv_pos_of, dim3(nb), dim3(256), 0, s, n, order, pos_of);
    hipLaunchKernelGGL(k_lv_chain_tables, dim3(nb), dim3(256), 0, s, n, order, pos_of, P->ptr, P->idx, P->skip_first,
                       P->skip_last, P->npre, P->nin, P->tslot, sums);
    unsigned long long hs[2] = {0, 0};
    CSX_HIP(hipMemcpyAsync(hs, sums, sizeof hs, hipMemcpyDeviceToHost, s));
    CSX_LAUNCH_CHECK();
    CSX_HIP(hipStreamSynchronize(s));
    P->chain_ok = hs[1] > 0 && hs[0] * 4 <= hs[1];
    P->scheduled = true;
    return CSX_OK;
}

// Levels proposed by the caller (cholsol_plan: heights / depths in the elimination tree, which ARE the level sets of a
// Cholesky factor's two solves): one verification pass over the pattern instead of one round per level.
static int schedule_from_hint(TriPlan *P, bool *done) {
    *done = false;
    hipStream_t s = ctx().stream;
    const int32_t n = P->n;
    if ((int32_t)P->level_hint.size() != n) return CSX_OK;
    int32_t L = 0;
    for (int32_t v : P->level_hint) {
        if (v < 0) return CSX_OK;
        L = std::max(L, v + 1);
    }
    DevScope tmp;
    int32_t *level = nullptr;
    int *stats = nullptr;
    CSX_TRY(tmp.alloc(&level, (size_t)n));
    CSX_TRY(tmp.alloc(&stats, 1));
    CSX_HIP(hipMemcpyAsync(level, P->level_hint.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemsetAsync(stats, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_lv_verify, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, P->ptr, P->idx, P->skip_first,
                       P->skip_last, P->forward ? 1 : 0, level, stats);
    int bad = 0;
    CSX_HIP(hipMemcpyAsync(&bad, stats, sizeof(int), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    std::vector<int32_t>().swap(P->level_hint);
    if (bad) return CSX_OK;                 // not the factor the hint was made for: the general analysis runs
    CSX_TRY(schedule_from_levels(P, level, L));
    *done = true;
    return CSX_OK;
}

// *done = false: not attempted or gave up (too deep) -> the host pass runs instead
static int schedule_on_device(TriPlan *P, bool *done) {
    *done = false;
    hipStream_t s = ctx().stream;
    const int32_t n = P->n;
    DevScope tmp;
    int32_t *level = nullptr;
    int *stats = nullptr;
    CSX_TRY(tmp.alloc(&level, (size_t)n));
    CSX_TRY(tmp.alloc(&stats, 2));
    const unsigned nb = (unsigned)(((int64_t)n + 255) / 256), nbw = (unsigned)(((int64_t)n + 3) / 4);
    hipLaunchKernelGGL(k_lv_init, dim3(nb), dim3(256), 0, s, n, level);
    int64_t resolved = 0;
    int32_t L = 0;
    for (; resolved < n; L++) {
        if (L >= LV_MAX_ROUNDS) return CSX_OK;                 // a deep chain: the host pass is the better tool
        int h[2] = {0, 0};
        CSX_HIP(hipMemsetAsync(stats, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(k_lv_round, dim3(nbw), dim3(256), 0, s, n, P->ptr, P->idx, P->skip_first, P->skip_last,
                           P->forward ? 1 : 0, L, level, stats);
        CSX_HIP(hipMemcpyAsync(h, stats, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        if (h[1] || h[0] == 0) {                               // malformed triangle: literal transcription of the loop
            P->scheduled = true;
            P->sequential = true;
            P->nlevels = n;
            *done = true;
            return CSX_OK;
        }
        resolved += h[0];
    }
    CSX_TRY(schedule_from_levels(P, level, L));
    *done = true;
    return CSX_OK;
}

// Level sets (host, O(nnz)): deferred until a level-scheduled solve needs them, so plans that
// only feed the fused in-LDS cholsol kernel never pay for the download.
static int ensure_schedule(TriPlan *P) {
    if (P->scheduled || P->n == 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    const int32_t n = P->n;
    const int where = ctx().opt.tri_levels_where;   // 0: by size, 1: host, 2: device
    if (!P->level_hint.empty() && where != 1) {
        bool done = false;
        CSX_TRY(schedule_from_hint(P, &done));
        if (done) return CSX_OK;
    }
    if (where == 2 || (where == 0 && (int64_t)P->gnnz >= LV_DEVICE_MIN_TERMS)) {
        bool done = false;
        CSX_TRY(schedule_on_device(P, &done));
        if (done) return CSX_OK;
        dfree(P->order);          // gave up part way (deeper than LV_MAX_ROUNDS): start over on the host
        P->order = nullptr;
    }
    std::vector<int32_t> hptr, hidx;
    CSX_TRY(download_i32(hptr, P->ptr, (size_t)n + 1));
    CSX_TRY(download_i32(hidx, P->idx, (size_t)P->gnnz));
    std::vector<int32_t> level;
    P->scheduled = true;
    if (!compute_levels(n, hptr, hidx, P->skip_first, P->skip_last, P->forward, level, P->nlevels)) {
        P->sequential = true;
        P->nlevels = n;
        return CSX_OK;
    }
    // rows sorted by level (counting sort; inside a level ascending row)
    P->level_ptr_h.assign((size_t)P->nlevels + 1, 0);
    for (int32_t r = 0; r < n; r++) P->level_ptr_h[(size_t)level[r] + 1]++;
    for (int32_t l = 0; l < P->nlevels; l++) P->level_ptr_h[(size_t)l + 1] += P->level_ptr_h[(size_t)l];
    std::vector<int32_t> order((size_t)n), fill(P->level_ptr_h.begin(), P->level_ptr_h.end() - 1);
    for (int32_t r = 0; r < n; r++) order[(size_t)fill[(size_t)level[r]]++] = r;
    CSX_TRY(dalloc(&P->order, (size_t)n));
    CSX_TRY(dalloc(&P->level_ptr, (size_t)P->nlevels + 1));
    CSX_HIP(hipMemcpyAsync(P->order, order.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(P->level_ptr, P->level_ptr_h.data(), ((size_t)P->nlevels + 1) * sizeof(int32_t),
                           hipMemcpyHostToDevice, s));
    // blocked chain walker: blocks of CHB consecutive positions of `order`
    std::vector<int32_t> pos_of((size_t)n), hnpre((size_t)n, 0), hnin((size_t)n, 0);
    int64_t suffix_terms = 0, all_terms = 0;
    std::vector<int8_t> hslot((size_t)P->gnnz + 1, (int8_t)-1);
    for (int32_t q = 0; q < n; q++) pos_of[(size_t)order[(size_t)q]] = q;
    for (int32_t q = 0; q < n; q++) {
        const int32_t r = order[(size_t)q], blk = q & ~(CHB - 1);
        const int32_t b = hptr[(size_t)r] + P->skip_first, e = hptr[(size_t)r + 1] - P->skip_last;
        int32_t pre = 0;
        bool in_prefix = true;
        for (int32_t t = b; t < e; t++) {
            const int32_t sp = pos_of[(size_t)hidx[(size_t)t]];
            const bool inside = sp >= blk;              // sp < q always: sources have a lower level
            hslot[(size_t)t] = inside ? (int8_t)(sp - blk) : (int8_t)-1;
            if (inside) in_prefix = false, hnin[(size_t)q]++;
            if (in_prefix) pre++;
        }
        hnpre[(size_t)q] = pre;
        suffix_terms += (e - b) - pre;
        all_terms += e - b;
    }
    // the chain walker pays off when most of a row can be done ahead of its in-block sources (forward
    // solves: the in-block sources are a row's LAST terms); when they come first (L' x = b in the
    // reference's order) nearly every term would wait in phase B and the level walker is the better one
    P->chain_ok = all_terms > 0 && suffix_terms * 4 <= all_terms;
    CSX_TRY(dalloc(&P->npre, (size_t)n));
    CSX_TRY(dalloc(&P->nin, (size_t)n));
    CSX_HIP(hipMemcpyAsync(P->nin, hnin.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_TRY(dalloc(&P->tslot, (size_t)P->gnnz + 1));
    CSX_HIP(hipMemcpyAsync(P->npre, hnpre.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(P->tslot, hslot.data(), (size_t)P->gnnz + 1, hipMemcpyHostToDevice, s));
    CSX_HIP(hipStreamSynchronize(s));
    return CSX_OK;
}

static void make_segments(TriPlan *P, int nrhs) {
    P->segs.clear();
    int32_t l = 0;
    while (l < P->nlevels) {
        const int64_t w = (int64_t)(P->level_ptr_h[(size_t)l + 1] - P->level_ptr_h[(size_t)l]) * nrhs;
        if (w > NARROW) {
            P->segs.push_back({l, l + 1, false});
            l++;
            continue;
        }
        int32_t e = l + 1;
        while (e < P->nlevels && (int64_t)(P->level_ptr_h[(size_t)e + 1] - P->level_ptr_h[(size_t)e]) * nrhs <= NARROW) e++;
        P->segs.push_back({l, e, true});
        l = e;
    }
}

int tri_solve_raw(TriPlan *P, double *X, int32_t nrhs, bool relaxed) {
    hipStream_t s = ctx().stream;
    if (P->zero_pivot) return CSX_EZEROPIVOT;
    if (P->n == 0 || nrhs == 0) return CSX_OK;
    if (ctx().opt.tri_components) CSX_TRY(analyse_components(P));
    if (P->comp_ok && ctx().opt.tri_components) return solve_components(P, X, nrhs);
    // One pass over the pattern tells a banded chain from the rest: band width, share of columns with a neighbour link,
    // proper triangle.  Such a system goes to the column loops without any level analysis -- x in LDS when it fits,
    // a window of x otherwise.
    if (ctx().opt.tri_columns) {
        if (P->band < 0) {
            DevScope tmp;
            int *o = nullptr;
            int h[3] = {0, 0, 0};
            CSX_TRY(tmp.alloc(&o, 3));
            CSX_HIP(hipMemsetAsync(o, 0, 3 * sizeof(int), s));
            hipLaunchKernelGGL(k_tri_band, dim3((unsigned)std::min<int64_t>(((int64_t)P->n + 3) / 4, 2048)), dim3(256), 0, s, P->n, P->Tp, P->Ti,
                               (P->kind == CSX_TRI_L || P->kind == CSX_TRI_LT) ? 1 : 0, o);
            CSX_HIP(hipMemcpyAsync(h, o, sizeof h, hipMemcpyDeviceToHost, s));
            CSX_HIP(hipStreamSynchronize(s));
            P->band = h[2] ? 0x7fffffff : h[0];     // not a proper triangle: the schedule's literal loop handles it
            P->links = h[1];
            if (P->band != 0x7fffffff && (P->kind == CSX_TRI_L || P->kind == CSX_TRI_U)) {
                int hd = 0;                          // the same row twice in a column: the lanes of a column would race
                CSX_HIP(hipMemsetAsync(o, 0, sizeof(int), s));
                hipLaunchKernelGGL(k_adjacent_equal, dim3((unsigned)(((int64_t)P->n + 3) / 4)), dim3(256), 0, s, P->n, P->ptr,
                                   P->idx, o);
                CSX_HIP(hipMemcpyAsync(&hd, o, sizeof(int), hipMemcpyDeviceToHost, s));
                CSX_HIP(hipStreamSynchronize(s));
                P->col_state = hd ? 2 : 1;
            }
        }
        uint32_t Wn = 64;
        while ((int64_t)Wn < (int64_t)P->band + 4 && Wn < (1u << 20)) Wn <<= 1;
        const bool chainlike = (int64_t)P->links * 12 > (int64_t)P->n * 11;   // nearly every column hands on to its neighbour
        if (chainlike && P->band != 0x7fffffff && (size_t)Wn * sizeof(double) <= 128 * 1024) {
            const size_t lds = (size_t)Wn * sizeof(double);
            const bool big = P->n > TC_MAX_N;       // x of one right-hand side does not fit LDS: only the window kernels apply
            const bool push = P->kind == CSX_TRI_L || P->kind == CSX_TRI_U;
            const TriPlan *M = P->mate;
            // rounding-equal order, L' (any size): the rows of L (the mate's gather arrays) pushed as the columns of L'
            const bool mate_push = !push && relaxed && P->kind == CSX_TRI_LT && M && M->kind == CSX_TRI_L && M->owns_g &&
                                   M->col_state == 1 && M->n == P->n;
            if ((big && push && P->col_state == 1) || mate_push) {
                const int32_t *cp = mate_push ? M->ptr : P->Tp, *ci = mate_push ? M->idx : P->Ti;
                const double *cx = mate_push ? M->val : P->Tx, *cd = mate_push ? M->diag : P->diag;
                const int sf = (!mate_push && P->kind == CSX_TRI_L) ? 1 : 0, sl = (!mate_push && P->kind == CSX_TRI_U) ? 1 : 0;
                const int asc = (!mate_push && P->kind == CSX_TRI_L) ? 1 : 0;
                const int64_t terms = mate_push ? (int64_t)M->gnnz : (int64_t)P->gnnz;
                if (terms > (int64_t)192 * P->n) {
                    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_wcolumns<1024>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                    hipLaunchKernelGGL(k_tri_wcolumns<1024>, dim3((unsigned)nrhs), dim3(1024), lds, s, P->n, cp, ci, cx, cd, sf,
                                       sl, asc, Wn - 1, X, nrhs);
                } else {
                    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_wcolumns<256>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                    hipLaunchKernelGGL(k_tri_wcolumns<256>, dim3((unsigned)nrhs), dim3(256), lds, s, P->n, cp, ci, cx, cd, sf, sl,
                                       asc, Wn - 1, X, nrhs);
                }
                CSX_LAUNCH_CHECK();
                return CSX_OK;
            }
            if (big && !push) {
#define CSX_WCH(K, R)                                                                                              \
    {                                                                                                              \
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_wcolchain<K, R>),                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));               \
        hipLaunchKernelGGL((k_tri_wcolchain<K, R>), dim3((unsigned)nrhs), dim3(64), lds, s, P->n, P->Tp, P->Ti, P->Tx, \
                           Wn - 1, X, nrhs);                                                                       \
    }
                const bool wide = (int64_t)P->gnnz > (int64_t)160 * P->n;   // columns of more than two rounds on average
                if (P->kind == CSX_TRI_LT) {
                    if (wide) CSX_WCH(CSX_TRI_LT, 12) else CSX_WCH(CSX_TRI_LT, 2)
                } else {
                    if (wide) CSX_WCH(CSX_TRI_UT, 12) else CSX_WCH(CSX_TRI_UT, 2)
                }
#undef CSX_WCH
                CSX_LAUNCH_CHECK();
                return CSX_OK;
            }
        }
    }
    // a proper triangle whose columns nearly all hand on to their neighbour is a chain whatever its level sets say
    const bool chain_by_links = ctx().opt.tri_columns && P->band >= 0 && P->band != 0x7fffffff &&
                                (int64_t)P->links * 12 > (int64_t)P->n * 11;
    auto schedule_or_literal = [&](bool *done) -> int {
        *done = false;
        CSX_TRY(ensure_schedule(P));
        if (P->sequential) {
            hipLaunchKernelGGL(k_tri_sequential, dim3((unsigned)((nrhs + 63) / 64)), dim3(64), 0, s, P->kind, P->n, P->Tp,
                               P->Ti, P->Tx, X, nrhs);
            CSX_LAUNCH_CHECK();
            *done = true;
        }
        return CSX_OK;
    };
    bool done = false;
    if (!(chain_by_links && P->n <= TC_MAX_N)) {
        CSX_TRY(schedule_or_literal(&done));
        if (done) return CSX_OK;
    }
    // small and chain-like: the column loop, one wave per right-hand side, beats any schedule
    if (ctx().opt.tri_columns && P->n <= TC_MAX_N && (chain_by_links || (int64_t)P->nlevels * 12 > P->n)) {
        if (P->col_state == 0) {
            P->col_state = 1;
            if (P->kind == CSX_TRI_L || P->kind == CSX_TRI_U) {   // gather structure of a push kind = stable transpose
                DevScope tmp;
                int *flag = nullptr;
                int h = 0;
                CSX_TRY(tmp.alloc(&flag, 1));
                CSX_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
                hipLaunchKernelGGL(k_adjacent_equal, dim3((unsigned)(((int64_t)P->n + 3) / 4)), dim3(256), 0, s, P->n, P->ptr,
                                   P->idx, flag);
                CSX_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
                CSX_HIP(hipStreamSynchronize(s));
                if (h) P->col_state = 2;
            }
        }
        if (P->col_state == 1 && ((size_t)P->n + 8) * sizeof(double) <= 150 * 1024) {
            const size_t lds = ((size_t)P->n + 8) * sizeof(double);
#define CSX_TC(K)                                                                                                  \
    {                                                                                                              \
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_columns<K>),                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));                \
        hipLaunchKernelGGL(k_tri_columns<K>, dim3((unsigned)nrhs), dim3(TC_THREADS), lds, s, P->n, P->Tp, P->Ti, P->Tx, \
                           X, nrhs);                                                                               \
    }
#define CSX_TCH(K)                                                                                                 \
    {                                                                                                              \
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tri_colchain<K>),                               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));               \
        hipLaunchKernelGGL(k_tri_colchain<K>, dim3((unsigned)nrhs), dim3(64), (size_t)P->n * sizeof(double), s, P->n, P->Tp, \
                           P->Ti, P->Tx, X, nrhs);                                                                 \
    }
            switch (P->kind) {
                case CSX_TRI_L: CSX_TC(CSX_TRI_L) break;
                case CSX_TRI_LT: CSX_TCH(CSX_TRI_LT) break;
                case CSX_TRI_U: CSX_TC(CSX_TRI_U) break;
                default: CSX_TCH(CSX_TRI_UT) break;
            }
#undef CSX_TC
#undef CSX_TCH
            CSX_LAUNCH_CHECK();
            return CSX_OK;
        }
    }
    if (!P->scheduled) {                 // the column loops did not apply after all (duplicate rows, x too long for LDS)
        CSX_TRY(schedule_or_literal(&done));
        if (done) return CSX_OK;
    }
    // the exact chain walker is already the fast one when in-block sources come last: relax only the others
    relaxed = relaxed && !P->chain_ok;
    make_segments(P, nrhs);
    const bool no_chain = !ctx().opt.tri_chain_walker;
    // few right-hand sides and long rows: a wave per row (k_tri_level_rows) instead of a thread per (row, right-hand side)
    const bool by_rows = nrhs <= TRW_MAX_RHS && P->n > 0 && (int64_t)P->gnnz >= (int64_t)TRW_MIN_ROW * P->n &&
                         ctx().opt.tri_row_waves;
    // many right-hand sides and long rows: a wave per (row, 64 right-hand sides) (k_tri_level_rows64)
    const bool by_rows64 = nrhs >= TR64_MIN_RHS && P->n > 0 && (int64_t)P->gnnz >= (int64_t)TRW_MIN_ROW * P->n &&
                           ctx().opt.tri_row_waves;
    for (const Segment &g : P->segs) {
        if (by_rows64 && g.one_wg) {
            // runs of narrow levels in two phases, TR64_RUN levels at a time (see k_tri_run_prefix64)
            if (!P->level_of && g.l1 - g.l0 >= TR64_RUN_MIN) {
                CSX_TRY(dalloc(&P->level_of, (size_t)P->n));
                CSX_TRY(dalloc(&P->resume, (size_t)P->n));
                hipLaunchKernelGGL(k_tri_level_of, dim3((unsigned)(((int64_t)P->n + 255) / 256)), dim3(256), 0, s, P->nlevels,
                                   P->level_ptr, P->order, P->n, P->level_of);
            }
            for (int32_t a = g.l0; a < g.l1;) {
                const int32_t b = std::min(g.l1, a + TR64_RUN);
                const bool two_phase = b - a >= TR64_RUN_MIN;
                if (two_phase) {
                    const int32_t first = P->level_ptr_h[(size_t)a], count = P->level_ptr_h[(size_t)b] - first;
                    const int64_t waves = (int64_t)count * ((nrhs + 63) / 64);
                    hipLaunchKernelGGL(k_tri_run_prefix64, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P->order, first,
                                       count, a, P->level_of, P->ptr, P->idx, P->val, P->skip_first, P->skip_last, X, nrhs,
                                       P->resume);
                }
                hipLaunchKernelGGL(k_tri_levels_rows64_one_wg, dim3(1), dim3(1024), 0, s, P->order, P->level_ptr, a, b, P->ptr,
                                   P->idx, P->val, P->diag, P->skip_first, P->skip_last, X, nrhs,
                                   two_phase ? P->resume : nullptr);
                a = b;
            }
        } else if (by_rows64) {
            const int32_t first = P->level_ptr_h[(size_t)g.l0];
            const int32_t count = P->level_ptr_h[(size_t)g.l0 + 1] - first;
            const int64_t waves = (int64_t)count * ((nrhs + 63) / 64);
            hipLaunchKernelGGL(k_tri_level_rows64, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P->order, first, count,
                               P->ptr, P->idx, P->val, P->diag, P->skip_first, P->skip_last, X, nrhs);
        } else if (by_rows && g.one_wg) {
            if (!P->level_of && g.l1 - g.l0 >= TR64_RUN_MIN) {
                CSX_TRY(dalloc(&P->level_of, (size_t)P->n));
                CSX_TRY(dalloc(&P->resume, (size_t)P->n));
                hipLaunchKernelGGL(k_tri_level_of, dim3((unsigned)(((int64_t)P->n + 255) / 256)), dim3(256), 0, s, P->nlevels,
                                   P->level_ptr, P->order, P->n, P->level_of);
            }
            for (int32_t a = g.l0; a < g.l1;) {
                const int32_t b = std::min(g.l1, a + TR64_RUN);
                const bool two_phase = b - a >= TR64_RUN_MIN;
                if (two_phase) {
                    const int32_t first = P->level_ptr_h[(size_t)a], count = P->level_ptr_h[(size_t)b] - first;
                    hipLaunchKernelGGL(k_tri_run_prefix_rows, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, P->order, first,
                                       count, a, P->level_of, P->ptr, P->idx, P->val, P->skip_first, P->skip_last, X, nrhs,
                                       P->resume);
                }
                hipLaunchKernelGGL(k_tri_levels_rows_one_wg, dim3(1), dim3(1024), 0, s, P->order, P->level_ptr, a, b, P->ptr,
                                   P->idx, P->val, P->diag, P->skip_first, P->skip_last, X, nrhs,
                                   two_phase ? P->resume : nullptr);
                a = b;
            }
        } else if (by_rows) {
            const int32_t first = P->level_ptr_h[(size_t)g.l0];
            const int32_t count = P->level_ptr_h[(size_t)g.l0 + 1] - first;
            hipLaunchKernelGGL(k_tri_level_rows, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, P->order, first, count,
                               P->ptr, P->idx, P->val, P->diag, P->skip_first, P->skip_last, X, nrhs);
        } else if (g.one_wg && !no_chain && (P->chain_ok || relaxed)) {
            hipLaunchKernelGGL(k_tri_chain, dim3(1), dim3(64 * CHB), 0, s, P->order, P->level_ptr_h[(size_t)g.l0],
                               P->level_ptr_h[(size_t)g.l1], P->ptr, P->idx, P->val, P->diag, P->skip_first, P->skip_last,
                               P->npre, P->nin, P->tslot, X, nrhs, relaxed ? 1 : 0);
        } else if (g.one_wg) {
            hipLaunchKernelGGL(k_tri_levels_one_wg, dim3(1), dim3(1024), 0, s, P->order, P->level_ptr, g.l0, g.l1, P->ptr,
                               P->idx, P->val, P->diag, P->skip_first, P->skip_last, X, nrhs);
        } else {
            const int32_t first = P->level_ptr_h[(size_t)g.l0];
            const int32_t count = P->level_ptr_h[(size_t)g.l0 + 1] - first;
            const int64_t threads = (int64_t)count * nrhs;
            hipLaunchKernelGGL(k_tri_level, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, P->order, first,
                               count, P->ptr, P->idx, P->val, P->diag, P->skip_first, P->skip_last, X, nrhs);
        }
    }
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

int tri_analyse_raw(const Csc *T, int kind, TriPlan **out) { return analyse(T, kind, out); }
void tri_set_mate(TriPlan *P, TriPlan *mate) { P->mate = mate; }
void tri_set_level_hint(TriPlan *P, std::vector<int32_t> &&level) { P->level_hint = std::move(level); }

void tri_gather_arrays(const TriPlan *P, const int32_t **ptr, const int32_t **idx, const double **val,
                       const double **diag) {
    *ptr = P->ptr;
    *idx = P->idx;
    *val = P->val;
    *diag = P->diag;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_tri_analyse(csx_handle_t hT, int kind, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *T = csc(hT);
    if (!T || !out || !T->x || T->m != T->n || kind < 0 || kind > 3) return CSX_EINVAL;
    TriPlan *P = nullptr;
    int st = analyse(T, kind, &P);
    if (st != CSX_OK) {
        free_triplan(P);
        return st;
    }
    *out = put(K_TRIPLAN, P);
    return CSX_OK;
}

extern "C" int csx_tri_info(csx_handle_t h, int32_t *n, int32_t *levels, int32_t *sequential) {
    TriPlan *P = (TriPlan *)get(h, K_TRIPLAN);
    if (!P) return CSX_EINVAL;
    CSX_TRY(ensure_schedule(P));
    if (n) *n = P->n;
    if (levels) *levels = P->nlevels;
    if (sequential) *sequential = P->sequential ? 1 : 0;
    return CSX_OK;
}

extern "C" int csx_tri_components(csx_handle_t h, int32_t *ncomp) {
    TriPlan *P = (TriPlan *)get(h, K_TRIPLAN);
    if (!P || !ncomp) return CSX_EINVAL;
    *ncomp = P->comp_ok ? P->ncomp : 0;
    return CSX_OK;
}

namespace csx {
// ---- "tri.host_chains" (opt-in, default 0) -----------------------------------------------------------------------------------------
// One right-hand side in HOST memory on a factor whose dependency graph is a chain (levels > n / 4) leaves the device nothing to do
// in parallel: the exact order is one dependent subtraction per term (1.07 us per level measured; bcsstk16: 6.5 ms against 0.65 ms
// for the same loop on one host core; SURVEY 8d's W-chain: 67 against 1.6 ms).  With the option set such a call runs the
// reference's loop (csparse.py:1330-1365, :2368-2385, :2460-2475; multiply and subtract rounded separately: the same bits) on the
// host, on a copy of the factor the plan downloads once.  It is a dispatch decision inside the library for a case the device
// loses, OFF by default: every call runs the HIP path unless the caller asks otherwise, and nothing here stands in for a missing GPU
// (the plan it hangs off exists only on a device).
#pragma clang fp contract(off)
static void tri_host_loop(int kind, int32_t n, const int32_t *Tp, const int32_t *Ti, const double *Tx, double *x) {
    switch (kind) {
        case CSX_TRI_L:
            for (int32_t j = 0; j < n; j++) {
                x[j] /= Tx[Tp[j]];
                for (int32_t p = Tp[j] + 1; p < Tp[j + 1]; p++) {
                    const double t = Tx[p] * x[j];
                    x[Ti[p]] = x[Ti[p]] - t;
                }
            }
            break;
        case CSX_TRI_LT:
            for (int32_t j = n - 1; j >= 0; j--) {
                for (int32_t p = Tp[j] + 1; p < Tp[j + 1]; p++) {
                    const double t = Tx[p] * x[Ti[p]];
                    x[j] = x[j] - t;
                }
                x[j] /= Tx[Tp[j]];
            }
            break;
        case CSX_TRI_U:
            for (int32_t j = n - 1; j >= 0; j--) {
                x[j] /= Tx[Tp[j + 1] - 1];
                for (int32_t p = Tp[j]; p < Tp[j + 1] - 1; p++) {
                    const double t = Tx[p] * x[j];
                    x[Ti[p]] = x[Ti[p]] - t;
                }
            }
            break;
        default:
            for (int32_t j = 0; j < n; j++) {
                for (int32_t p = Tp[j]; p < Tp[j + 1] - 1; p++) {
                    const double t = Tx[p] * x[Ti[p]];
                    x[j] = x[j] - t;
                }
                x[j] /= Tx[Tp[j + 1] - 1];
            }
            break;
    }
}
#pragma clang fp contract(fast)

// *taken = false: the option is off or the factor is no chain (the caller goes to the device as always)
int tri_solve_host_raw(TriPlan *P, double *x, bool *taken) {
    *taken = false;
    if (!ctx().opt.tri_host_chains || P->n == 0) return CSX_OK;
    if (P->zero_pivot) return CSX_EZEROPIVOT;
    if ((int64_t)P->gnnz >= 50000000) return CSX_OK;
    CSX_TRY(ensure_schedule(P));
    if (!P->sequential && (int64_t)P->nlevels * 4 <= (int64_t)P->n) return CSX_OK;
    const int32_t n = P->n;
    if (P->hTp.empty()) {
        hipStream_t s = ctx().stream;
        P->hTp.resize((size_t)n + 1);
        CSX_HIP(hipMemcpyAsync(P->hTp.data(), P->Tp, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        const size_t nnz = (size_t)P->hTp[(size_t)n];
        P->hTi.resize(nnz);
        P->hTx.resize(nnz);
        if (nnz) {
            CSX_HIP(hipMemcpyAsync(P->hTi.data(), P->Ti, nnz * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            CSX_HIP(hipMemcpyAsync(P->hTx.data(), P->Tx, nnz * sizeof(double), hipMemcpyDeviceToHost, s));
            CSX_HIP(hipStreamSynchronize(s));
        }
        for (size_t q = 0; q < nnz; q++)
            if (P->hTi[q] < 0 || P->hTi[q] >= n) {      // (a wrapped matrix is validated before analysis; belt and braces)
                P->hTp.clear();
                return CSX_EINVAL;
            }
    }
    tri_host_loop(P->kind, n, P->hTp.data(), P->hTi.data(), P->hTx.data(), x);
    *taken = true;
    return CSX_OK;
}
}  // namespace csx

extern "C" int csx_tri_solve_list(csx_handle_t h, double *x, int *taken) {
    CSX_TRY(require_ready());
    TriPlan *P = (TriPlan *)get(h, K_TRIPLAN);
    if (!P || !x || !taken) return CSX_EINVAL;
    bool t = false;
    const int st = tri_solve_host_raw(P, x, &t);
    *taken = t ? 1 : 0;
    return st;
}

extern "C" int csx_tri_set_order(csx_handle_t h, int exact) {
    TriPlan *P = (TriPlan *)get(h, K_TRIPLAN);
    if (!P) return CSX_EINVAL;
    P->rounding_equal = exact == 0;
    return CSX_OK;
}

extern "C" int csx_tri_order_info(csx_handle_t h, int32_t *matrix_cores, double *growth) {
    TriPlan *P = (TriPlan *)get(h, K_TRIPLAN);
    if (!P) return CSX_EINVAL;
    if (matrix_cores) *matrix_cores = (P->rounding_equal && P->rag) ? 1 : 0;
    if (growth) *growth = P->rag_growth;
    return CSX_OK;
}

extern "C" int csx_tri_solve(csx_handle_t h, csx_handle_t hX, int32_t nrhs) {
    CSX_TRY(require_ready());
    TriPlan *P = (TriPlan *)get(h, K_TRIPLAN);
    Vec *X = vec(hX);
    if (!P || !X || nrhs < 0 || X->len < (int64_t)P->n * nrhs) return CSX_EINVAL;
    static const bool relaxed_env = ablation_env("CSX_TRI_RELAXED") != nullptr;   // experiments only
    return tri_solve_raw(P, (double *)X->d, nrhs, relaxed_env);
}

extern "C" int csx_permute_vec(csx_handle_t hp, csx_handle_t hb, csx_handle_t hx, int32_t n, int32_t nrhs,
                               int inverse) {
    CSX_TRY(require_ready());
    Vec *b = vec(hb), *x = vec(hx);
    Vec *p = hp ? ivec(hp) : nullptr;
    if (!b || !x || n < 0 || nrhs < 0 || (hp && (!p || p->len < n))) return CSX_EINVAL;
    if (b->len < (int64_t)n * nrhs || x->len < (int64_t)n * nrhs || b->d == x->d) return CSX_EINVAL;
    const int64_t total = (int64_t)n * nrhs;
    if (total == 0) return CSX_OK;
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx().stream,
                       p ? (const int32_t *)p->d : nullptr, (const double *)b->d, (double *)x->d, n, nrhs, inverse);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

__global__ __launch_bounds__(256) void k_invert_perm(const int32_t *__restrict__ p, int32_t n, int32_t *__restrict__ inv) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && (uint32_t)p[k] < (uint32_t)n) inv[p[k]] = (int32_t)k;      // (an entry out of range is the caller's error: no write outside)
}

// cs_lusol's solve phase for a block of right-hand sides (csparse.py:1470-1473): x = P b (cs_ipvec with pinv), L x = x, U x = x,
// b = Q x (cs_ipvec with q), B overwritten with the solutions.  When both factors are forests of small components in the
// rounding-equal order (csx_tri_set_order) the permutations are FUSED into the sweeps: the sweep over L gathers its rows out of B
// through the inverse of pinv and leaves x in `work`, the sweep over U reads `work` and scatters through q into B -- four passes over
// the block instead of eight ("on device must be fused into the batched solve").  Otherwise: the four steps, one
// after the other, exactly what csx_permute_vec + csx_tri_solve + csx_tri_solve + csx_permute_vec do.  *fused says which.
extern "C" int csx_lusol_solve(csx_handle_t hL, csx_handle_t hU, csx_handle_t hpinv, csx_handle_t hq, csx_handle_t hB,
                               csx_handle_t hWork, int32_t nrhs, int *fused) {
    CSX_TRY(require_ready());
    TriPlan *PL = (TriPlan *)get(hL, K_TRIPLAN), *PU = (TriPlan *)get(hU, K_TRIPLAN);
    Vec *B = vec(hB), *W = vec(hWork);
    Vec *pv = hpinv ? ivec(hpinv) : nullptr, *qv = hq ? ivec(hq) : nullptr;
    if (fused) *fused = 0;
    if (!PL || !PU || !B || !W || nrhs < 0 || PL->n != PU->n || B->d == W->d) return CSX_EINVAL;
    const int32_t n = PL->n;
    if ((hpinv && (!pv || pv->len < n)) || (hq && (!qv || qv->len < n))) return CSX_EINVAL;
    if (B->len < (int64_t)n * nrhs || W->len < (int64_t)n * nrhs) return CSX_EINVAL;
    if (n == 0 || nrhs == 0) return CSX_OK;
    if (PL->zero_pivot || PU->zero_pivot) return CSX_EZEROPIVOT;
    hipStream_t s = ctx().stream;
    const int32_t *pinv = pv ? (const int32_t *)pv->d : nullptr, *q = qv ? (const int32_t *)qv->d : nullptr;
    double *b = (double *)B->d, *x = (double *)W->d;
    const int64_t total = (int64_t)n * nrhs;
    if (ctx().opt.tri_components && PL->rounding_equal && PU->rounding_equal && nrhs > 8) {
        CSX_TRY(analyse_components(PL));
        CSX_TRY(analyse_components(PU));
        if (PL->comp_ok && PU->comp_ok) {
            CSX_TRY(components_ragged(PL));
            CSX_TRY(components_ragged(PU));
        }
        if (PL->comp_ok && PU->comp_ok && PL->rag && PU->rag) {
            DevScope tmp;
            int32_t *invp = nullptr;
            if (pinv) {
                CSX_TRY(tmp.alloc(&invp, (size_t)n));
                CSX_HIP(hipMemsetAsync(invp, 0, (size_t)n * sizeof(int32_t), s));      // (not a permutation: rows never named read row 0)
                hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, pinv, n, invp);
                CSX_LAUNCH_CHECK();
            }
            CSX_TRY(ragged_solve_io(PL->rag, PL->comp_nodes, invp, nullptr, !PL->forward, 1, b, x, nrhs, n));
            CSX_TRY(ragged_solve_io(PU->rag, PU->comp_nodes, nullptr, q, !PU->forward, 1, x, b, nrhs, n));
            if (fused) *fused = 1;
            return CSX_OK;
        }
    }
    if (ctx().opt.tri_components && !PL->rounding_equal && !PU->rounding_equal && nrhs > 32) {
        // the exact order on forests of small components: the same fusion through the in-LDS sweeps (k_tri_local reads and writes a
        // component's rows through the row maps); the arithmetic and its order are those of the separate steps, bit for bit
        CSX_TRY(analyse_components(PL));
        CSX_TRY(analyse_components(PU));
        if (PL->comp_ok && PU->comp_ok) {
            DevScope tmp;
            int32_t *invp = nullptr;
            if (pinv) {
                CSX_TRY(tmp.alloc(&invp, (size_t)n));
                CSX_HIP(hipMemsetAsync(invp, 0, (size_t)n * sizeof(int32_t), s));
                hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, pinv, n, invp);
                CSX_LAUNCH_CHECK();
            }
            bool took = false;
            const TriIO ioL{b, invp, nullptr};
            CSX_TRY(solve_components(PL, x, nrhs, &ioL, &took));
            if (took) {
                const TriIO ioU{x, nullptr, q};
                CSX_TRY(solve_components(PU, b, nrhs, &ioU, &took));     // (same plan shape, same nrhs: taken too)
                if (took) {
                    if (fused) *fused = 1;
                    return CSX_OK;
                }
                // (not taken after all: x holds L's solution of the permuted block -- finish with the separate steps)
                CSX_TRY(tri_solve_raw(PU, x, nrhs, false));
                hipLaunchKernelGGL(k_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, (const double *)x, b, n, nrhs, 1);
                CSX_LAUNCH_CHECK();
                return CSX_OK;
            }
        }
    }
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pinv, (const double *)b, x, n, nrhs, 1);
    CSX_LAUNCH_CHECK();
    static const bool relaxed_env = ablation_env("CSX_TRI_RELAXED") != nullptr;   // experiments only
    CSX_TRY(tri_solve_raw(PL, x, nrhs, relaxed_env));
    CSX_TRY(tri_solve_raw(PU, x, nrhs, relaxed_env));
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, (const double *)x, b, n, nrhs, 1);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}
