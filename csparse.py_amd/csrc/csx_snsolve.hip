// Supernodal triangular solves with a Cholesky factor: cs_lsolve / cs_ltsolve (csparse.py:1341-1344, :1360-1364) in the
// rounding-equal order of a cholsol plan (csx_cholsol_set_order(plan, 0)), for connected factors whose elimination tree
// is deep in columns but shallow in SUPERNODES -- nested-dissection factors: the separators are dense trapezoids of L
// (w consecutive columns with the same rows below them), which cs_chol already factors as dense blocks (csx_cholband.hip).
//
// The column-level schedule gives every column of a separator a level of its own (a 700 x 700 grid in nested dissection:
// ~2 000 levels, each a launch or a workgroup barrier, each row's thousands of terms walked one dependent subtraction at
// a time).  Here the schedule has three kinds of unit:
//   * LEAF SUBTREES: maximal subtrees of the elimination tree with at most 64 columns (the small regions nested
//     dissection stops at: most of the columns, almost none of the work), all of them in one launch: first of all in the
//     forward solve, last of all in the backward one (after one launch has taken every leaf column's terms of rows
//     OUTSIDE its subtree -- ancestors, final by then).  A subtree is a triangle of at most 64 x 64 made dense and solved on
//     the matrix cores like B below, or (when the guard refuses that) a packed program walked out of LDS (k_sn_leaf).
//   * SUPERNODES of the rest, by height (forward) / depth (backward) in the supernodal elimination tree: FUNDAMENTAL ones
//     (dense trapezoids) and RELAXED ones (runs of at most 128 columns of a chain of the tree; see sn_build).  Per level,
//       A: every row (column) of the level's supernodes takes its terms from OUTSIDE its supernode -- all final: they
//          belong to descendants (ancestors) -- as tasks, a wave or a workgroup each per 64 right-hand sides; a line of
//          more than SN_WHOLE terms is cut into pieces of SN_SEG whose partial sums are added in a fixed order;
//       B: the triangle of every (chunk of a) supernode: a blocked TRSM on the matrix cores with explicit inverses of the
//          16 x 16 diagonal blocks (k_sn_mfma; fragments built with the plan, guarded by the blocks' condition), or by
//          substitution out of LDS in panels of 16 (k_sn_tri: "tri.supernodes" = 2, or a block past the guard).
//   * WIDE supernodes are cut into chunks of 128 columns (64 with the substitution triangles) that are solved one after the other (a blocked,
//     right-looking dense triangular solve): after chunk q, every remaining row of the supernode takes its 64 terms of
//     that chunk in one launch across the chip (the same kernel as A), then chunk q + 1 is solved.
// Forward reads the row-major copy of L the forward plan holds (terms of a row in ascending column order, so the terms
// inside the row's own supernode are its LAST ones); backward reads L itself (a column: diagonal, the rows inside its
// supernode, the rows below it).  In A a lane is a right-hand side: a term is one coalesced 512-byte load of a row of X
// and one FMA.
//
// Equal to the reference to rounding (the sums are regrouped), never used by the exact order; the same bits on every
// run.  tests/test_gpu_cholesky.py compares with the plain-C restatement of the reference at 1e-10 and with the exact order.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "csx_internal.h"

namespace csx {

constexpr int SN_SEG = 256;      // terms per piece of a long row / column
constexpr int SN_WHOLE = 512;    // a supernode's line of up to this many outside terms stays one task (no partial slots, no combine
                                 // launch: a banded chain's rows of 300 terms were cut in two in every one of its n / 64 steps)
constexpr int SN_PANEL = 16;
constexpr int SN_CHUNK = 64;     // columns per chunk of a wide supernode
constexpr int SN_LEAF = 64;      // columns of a leaf subtree (its x tile: SN_LEAF x 64 doubles of LDS per wave)

struct SnTask {
    int32_t line;   // row (forward) or column (backward)
    int32_t b, e;   // term range
    int32_t out;    // -1: subtract from X[line]; else partial slot
};

struct SnStep {
    int32_t t0, tc;   // tasks
    int32_t c0, cc;   // lines of this step that were cut into pieces (k_sn_combine)
    int32_t s0, sc;   // narrow virtual supernodes (at most 16 columns: one wave each)
    int32_t m0, mc;   // up to 32 columns: two waves, a quarter of the LDS of the wide ones
    int32_t b0, bc;   // the others (four waves each)
    int32_t q0, qc;   // all of them in one list for the matrix-core kernel (SnDir::tri4)
    int64_t terms;    // terms of the step's tasks together
};

struct SnDir {
    std::vector<SnStep> steps;       // host
    SnTask *tasks = nullptr;         // device
    int32_t *comb = nullptr;         // device: lines with partial slots, by step
    int32_t *narrow = nullptr;       // device: virtual supernode ids by step
    int32_t *medium = nullptr;       // device
    int32_t *wide = nullptr;         // device
    int32_t *part_ptr = nullptr;     // device [n + 1]: partial slots of a row / column
    int4 *tri4 = nullptr;            // device: (first column, width, first fragment, 0) of every virtual supernode, by step
    std::vector<int4> tri_h;         // the same on the host: a step of one triangle passes its entry with the launch
    int32_t nslots = 0;
};

struct SnPlan {
    int32_t n = 0, nsn = 0, max_w = 0, nleaf = 0, nleafcols = 0;
    int32_t *vs_a = nullptr, *vs_w = nullptr;   // device: first column and width of every virtual supernode (chunk)
    // leaf subtrees: columns by subtree (ascending inside), packed forward / backward programs by position in leaf_cols
    int32_t *leaf_ptr = nullptr, *leaf_cols = nullptr;
    int32_t *lf_ptr = nullptr, *lf_idx = nullptr, *lb_ptr = nullptr, *lb_idx = nullptr;
    double *lf_val = nullptr, *lb_val = nullptr, *ldiag = nullptr;
    SnTask *leaf_tasks = nullptr;               // backward: the leaf columns' rows outside their subtrees, in pieces
    int32_t nleaftasks = 0, nleafslots = 0;
    int32_t *lslot_ptr = nullptr;               // [nleafcols + 1] partial slots of a leaf column (after the backward plan's slots)
    SnDir fwd, bwd;
    double *partial = nullptr;
    int64_t partial_len = 0;
    int4 *leaf4 = nullptr;                      // (first position in leaf_cols, columns, first fragment, 0) of every leaf subtree
    int64_t frag_toff = 0;                      // the transposed tiles (backward sweep) lie this many doubles behind the forward ones
    double *frags = nullptr;                    // matrix-core fragments of every virtual supernode's and leaf subtree's triangle
    bool mfma = false;                          // fragments built and every block inverse tame: k_sn_mfma solves the triangles
    double growth = 0.0;                        // the guard's measure (k_sn_frags)
    int chunk = 64;                             // columns per chunk of a wide supernode / per relaxed run (128: matrix cores only)
    bool has_relaxed = false;                   // some supernodes are runs of a chain, not dense trapezoids: matrix cores only
    int64_t generation = 0;                     // counts the moves of `partial` (a captured solve holds its address)
};

void free_snplan(SnPlan *P) {
    if (!P) return;
    for (void *p : {(void *)P->vs_a, (void *)P->vs_w, (void *)P->leaf_ptr, (void *)P->leaf_cols, (void *)P->lf_ptr, (void *)P->lf_idx,
                    (void *)P->lb_ptr, (void *)P->lb_idx, (void *)P->lf_val, (void *)P->lb_val, (void *)P->ldiag, (void *)P->leaf_tasks,
                    (void *)P->lslot_ptr, (void *)P->partial, (void *)P->frags, (void *)P->leaf4})
        dfree(p);
    for (SnDir *d : {&P->fwd, &P->bwd}) {
        dfree(d->tasks);
        dfree(d->comb);
        dfree(d->narrow);
        dfree(d->medium);
        dfree(d->wide);
        dfree(d->part_ptr);
        dfree(d->tri4);
    }
    delete P;
}

namespace {

template <class T>
int up(T **d, const std::vector<T> &h) {
    CSX_TRY(dalloc(d, std::max<size_t>(h.size(), 1)));
    if (!h.empty())
        CSX_HIP(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
    return CSX_OK;
}

// sum over q in [b, e) of val[q] * X[idx[q], r]: lane = right-hand side, a term's index and value handed round by
// v_readlane, sixteen row loads of X in flight ahead of the multiply-adds
__device__ __forceinline__ double sn_dot(int32_t b, int32_t e, const int32_t *__restrict__ idx, const double *__restrict__ val,
                                         const double *X, int nrhs, int rr, int lane) {
    double acc0 = 0.0, acc1 = 0.0;
    int32_t ci = b + lane < e ? idx[b + lane] : 0;
    double cv = b + lane < e ? val[b + lane] : 0.0;
    for (int32_t q0 = b; q0 < e; q0 += 64) {
        const int32_t qn = q0 + 64 + lane;
        const int32_t cin = qn < e ? idx[qn] : 0;
        const double cvn = qn < e ? val[qn] : 0.0;
        const int cnt = e - q0 < 64 ? e - q0 : 64;   // uniform
        const int clo = __double2loint(cv), chi = __double2hiint(cv);
        double xa[16], xb[16];
        auto gather = [&](int g, double *xv) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int32_t c = __builtin_amdgcn_readlane(ci, g + u);   // lanes past cnt hold row 0: a valid address
                xv[u] = X[(int64_t)c * nrhs + rr];
            }
        };
        auto fma16 = [&](int g, const double *xv) {
#pragma unroll
            for (int u = 0; u < 16; u += 2) {                             // lanes past cnt hold value 0.0
                const double v0 = __hiloint2double(__builtin_amdgcn_readlane(chi, g + u), __builtin_amdgcn_readlane(clo, g + u));
                const double v1 = __hiloint2double(__builtin_amdgcn_readlane(chi, g + u + 1), __builtin_amdgcn_readlane(clo, g + u + 1));
                acc0 = fma(v0, xv[u], acc0);
                acc1 = fma(v1, xv[u + 1], acc1);
            }
        };
        // all the loads of the 64 terms first (a line of a banded chain is five such rounds, each a memory round trip,
        // and the step waits for the longest line), then the sums in term order
        double xc[16], xd[16];
        gather(0, xa);
        if (cnt > 16) gather(16, xb);
        if (cnt > 32) gather(32, xc);
        if (cnt > 48) gather(48, xd);
        fma16(0, xa);
        if (cnt > 16) fma16(16, xb);
        if (cnt > 32) fma16(32, xc);
        if (cnt > 48) fma16(48, xd);
        ci = cin;
        cv = cvn;
    }
    return acc0 + acc1;
}

// A: one wave per (task, 64 right-hand sides)
__global__ __launch_bounds__(256) void k_sn_outside(const SnTask *__restrict__ tasks, int32_t first, int32_t count,
                                                    const int32_t *__restrict__ idx, const double *__restrict__ val, double *X,
                                                    double *partial, int nrhs) {
    const int lane = threadIdx.x & 63;
    const int nblk = (nrhs + 63) >> 6;
    const int64_t wv = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wv >= (int64_t)count * nblk) return;
    const SnTask t = tasks[first + wv / nblk];
    if (t.e <= t.b) return;
    const int r = (int)(wv % nblk) * 64 + lane;
    const bool live = r < nrhs;
    const int rr = live ? r : nrhs - 1;
    const double d = sn_dot(t.b, t.e, idx, val, X, nrhs, rr, lane);
    if (!live) return;
    if (t.out < 0) X[(int64_t)t.line * nrhs + r] -= d;
    else partial[(int64_t)t.out * nrhs + r] = d;
}

// A for MANY tasks and FEW right-hand sides: R lanes per task (R = the number of right-hand sides rounded up to a power of
// two, at most 32), 64 / R tasks per wave, every lane walking its task's terms itself.  With a wave per task the launch of
// the leaf columns' outside terms on the 700 x 700 grid -- 378 000 tasks of a handful of terms -- is bound by the rate at
// which waves can be started (375 us for one right-hand side); this form starts 64 / R times fewer.  The sum is formed
// exactly as sn_dot forms it (even terms into one accumulator, odd ones into the other, the last group of 16 padded with
// zero products), so a right-hand side gets the same bits whichever kernel ran.
template <int R>
__global__ __launch_bounds__(256) void k_sn_outside_thin(const SnTask *__restrict__ tasks, int32_t first, int32_t count,
                                                         const int32_t *__restrict__ idx, const double *__restrict__ val, double *X,
                                                         double *partial, int nrhs) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t task = gid / R;
    const int r = (int)(gid % R);
    if (task >= count || r >= nrhs) return;
    const SnTask t = tasks[first + task];
    if (t.e <= t.b) return;
    double acc0 = 0.0, acc1 = 0.0;
    int32_t q = t.b;
    for (; q + 1 < t.e; q += 2) {
        acc0 = fma(val[q], X[(int64_t)idx[q] * nrhs + r], acc0);
        acc1 = fma(val[q + 1], X[(int64_t)idx[q + 1] * nrhs + r], acc1);
    }
    if (q < t.e) {
        acc0 = fma(val[q], X[(int64_t)idx[q] * nrhs + r], acc0);
        q++;
        if ((q - t.b) & 15) acc1 = fma(0.0, X[r], acc1);     // sn_dot's padding: pairs (0.0, row 0) up to the end of the group of 16
        q++;
    }
    for (; (q - t.b) & 15; q += 2) {
        acc0 = fma(0.0, X[r], acc0);
        acc1 = fma(0.0, X[r], acc1);
    }
    const double d = acc0 + acc1;
    if (t.out < 0) X[(int64_t)t.line * nrhs + r] -= d;
    else partial[(int64_t)t.out * nrhs + r] = d;
}

// A for steps of few tasks: a WORKGROUP per (task, 64 right-hand sides).  The task's terms are dealt to the four waves in
// runs of 64 (wave w: runs w, w + 4, ...), the four sums meet in LDS and are added in wave order.  A task of 300 terms --
// a row of a banded chain -- is then two memory round trips deep instead of five, and in such a factor the step waits for
// exactly that.  (Steps with thousands of tasks keep a wave per task: there the chip is full either way.)
__global__ __launch_bounds__(256) void k_sn_outside_wg(const SnTask *__restrict__ tasks, int32_t first, int32_t count,
                                                       const int32_t *__restrict__ idx, const double *__restrict__ val, double *X,
                                                       double *partial, int nrhs) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nblk = (nrhs + 63) >> 6;
    const SnTask t = tasks[first + blockIdx.x / nblk];
    const int r = (int)(blockIdx.x % nblk) * 64 + lane;
    const bool live = r < nrhs;
    const int rr = live ? r : nrhs - 1;
    double d = 0.0;
    for (int32_t q = t.b + 64 * w; q < t.e; q += 256) d += sn_dot(q, q + 64 < t.e ? q + 64 : t.e, idx, val, X, nrhs, rr, lane);
    part[w][lane] = d;
    __syncthreads();
    if (w != 0 || !live || t.e <= t.b) return;
    const double sum = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (t.out < 0) X[(int64_t)t.line * nrhs + r] -= sum;
    else partial[(int64_t)t.out * nrhs + r] = sum;
}

// Between A and B: a line cut into pieces takes its partial sums, in piece order, eight loads in flight at a time
// (inside B they were one dependent load after the other per line: 26 - 47 us per chunk on bcsstk16 instead of 14).
// entry k of the list: line = lines[k]; its slots [slot_ptr[j], slot_ptr[j + 1]) with j = k (by_pos) or the line.
__global__ __launch_bounds__(256) void k_sn_combine(const int32_t *__restrict__ lines, int32_t first, int32_t count,
                                                    const int32_t *__restrict__ slot_ptr, int by_pos,
                                                    const double *__restrict__ partial, double *X, int nrhs) {
    const int lane = threadIdx.x & 63;
    const int nblk = (nrhs + 63) >> 6;
    const int64_t wv = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wv >= (int64_t)count * nblk) return;
    const int32_t k = first + (int32_t)(wv / nblk), line = lines[k];
    const int32_t j = by_pos ? k : line;
    const int32_t s0 = slot_ptr[j], s1 = slot_ptr[j + 1];
    if (s1 <= s0) return;
    const int r = (int)(wv % nblk) * 64 + lane;
    if (r >= nrhs) return;
    double acc = X[(int64_t)line * nrhs + r];
    for (int32_t s = s0; s < s1; s += 8) {
        double pv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) pv[u] = s + u < s1 ? partial[(int64_t)(s + u) * nrhs + r] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) acc -= pv[u];
    }
    X[(int64_t)line * nrhs + r] = acc;
}

// B: (virtual) supernode columns [a, a + w), w <= W, for 64 right-hand sides: the triangle (Ls[i][t] = L(a + i, a + t), i > t)
// and the w rows of X (less the partial sums of their outside terms) staged in LDS; solved there in panels of 16 rows,
// the panel's triangle by one wave with its rows in registers, the update of the remaining rows by all waves.
// FWD: rows from the forward plan's row-major copy (row a + v: its terms of columns a .. a + v - 1 are the last v of its
// gather range), panels ascending.  !FWD: columns of L (column a + v: diagonal, then rows a + v + 1 ..), panels descending.
template <int NW, int W, bool FWD>
__global__ __launch_bounds__(64 * NW) void k_sn_tri(const int32_t *__restrict__ list, int32_t first,
                                                    const int32_t *__restrict__ vs_a, const int32_t *__restrict__ vs_w,
                                                    const int32_t *__restrict__ Tp, const double *__restrict__ Tx,
                                                    const double *__restrict__ Td, const int32_t *__restrict__ part_ptr,
                                                    const double *__restrict__ partial, double *X, int nrhs) {
    extern __shared__ __attribute__((aligned(16))) double sn_smem[];     // sn_tri_lds<W>() bytes
    double(*Ls)[W + 1] = reinterpret_cast<double(*)[W + 1]>(sn_smem);
    double(*xt)[64] = reinterpret_cast<double(*)[64]>(sn_smem + W * (W + 1));
    double *dg = sn_smem + W * (W + 1) + W * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nblk = (nrhs + 63) >> 6;
    const int32_t S = list[first + blockIdx.x / nblk];
    const int r = (int)(blockIdx.x % nblk) * 64 + lane;
    const bool live = r < nrhs;
    const int rr = live ? r : nrhs - 1;
    const int32_t a = vs_a[S], w = vs_w[S];
    // Staging: lane v holds the pointers of line a + v, so that every load below has its address at once (v_readlane)
    // and a wave's W / NW lines are all in flight together instead of one dependent chain per line.
    const int32_t mp = lane < w ? Tp[a + lane + (FWD ? 1 : 0)] : 0;
    // (part_ptr == nullptr: the partial sums were already taken off X by k_sn_combine)
    const int32_t ps = part_ptr && lane < w ? part_ptr[a + lane] : 0, pe = part_ptr && lane < w ? part_ptr[a + lane + 1] : 0;
    if (wave == 0 && lane < w) dg[lane] = 1.0 / (FWD ? Td[a + lane] : Tx[mp]);   // one division per line, off the chain below
    constexpr int PER = W / NW;
    double xv[PER], lv[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = wave + NW * i;
        xv[i] = 0.0;
        lv[i] = 0.0;
        if (v < w) {
            const int32_t m = __builtin_amdgcn_readlane(mp, v);
            xv[i] = X[(int64_t)(a + v) * nrhs + rr];
            if (FWD) {
                if (lane < v) lv[i] = Tx[m - v + lane];              // row a + v: its last v terms are columns a .. a + v - 1
            } else {
                if (lane > v && lane < w) lv[i] = Tx[m + lane - v];  // column a + v: row a + lane at m + lane - v
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int v = wave + NW * i;
        if (v < w) {
            const int32_t s0 = __builtin_amdgcn_readlane(ps, v), s1 = __builtin_amdgcn_readlane(pe, v);
            double acc = xv[i];
            for (int32_t s = s0; s < s1; s++) acc -= partial[(int64_t)s * nrhs + rr];
            xt[v][lane] = acc;
            if (FWD) {
                if (lane < v) Ls[v][lane] = lv[i];
            } else {
                if (lane > v && lane < w) Ls[lane][v] = lv[i];
            }
        }
    }
    __syncthreads();
    const int npan = (w + SN_PANEL - 1) / SN_PANEL;
    for (int pp = 0; pp < npan; pp++) {
        const int p = FWD ? pp : npan - 1 - pp;
        const int p0 = p * SN_PANEL;
        const int np = w - p0 < SN_PANEL ? w - p0 : SN_PANEL;
        if (wave == p % NW) {
            double acc[SN_PANEL];
#pragma unroll
            for (int t = 0; t < SN_PANEL; t++) acc[t] = t < np ? xt[p0 + t][lane] : 0.0;
            if (FWD) {
#pragma unroll
                for (int t = 0; t < SN_PANEL; t++)
                    if (t < np) {
                        const double y = acc[t] * dg[p0 + t];
                        acc[t] = y;
#pragma unroll
                        for (int s = t + 1; s < SN_PANEL; s++)
                            if (s < np) acc[s] = fma(-Ls[p0 + s][p0 + t], y, acc[s]);
                    }
            } else {
#pragma unroll
                for (int t = SN_PANEL - 1; t >= 0; t--)
                    if (t < np) {
                        const double y = acc[t] * dg[p0 + t];
                        acc[t] = y;
#pragma unroll
                        for (int s = 0; s < t; s++) acc[s] = fma(-Ls[p0 + t][p0 + s], y, acc[s]);
                    }
            }
#pragma unroll
            for (int t = 0; t < SN_PANEL; t++)
                if (t < np) {
                    xt[p0 + t][lane] = acc[t];
                    if (live) X[(int64_t)(a + p0 + t) * nrhs + r] = acc[t];
                }
        }
        if (pp + 1 == npan) break;
        __syncthreads();
        // the rows still to be solved take their terms of this panel
        const int lo = FWD ? p0 + np : 0, hi = FWD ? w : p0;
        for (int v = lo + wave; v < hi; v += NW) {
            double acc0 = xt[v][lane], acc1 = 0.0;
#pragma unroll
            for (int t = 0; t < SN_PANEL; t += 2) {
                if (t < np) acc0 = fma(-(FWD ? Ls[v][p0 + t] : Ls[p0 + t][v]), xt[p0 + t][lane], acc0);
                if (t + 1 < np) acc1 = fma(-(FWD ? Ls[v][p0 + t + 1] : Ls[p0 + t + 1][v]), xt[p0 + t + 1][lane], acc1);
            }
            xt[v][lane] = acc0 + acc1;
        }
        __syncthreads();
    }
}

template <int W>
constexpr size_t sn_tri_lds() {
    return (size_t)(W * (W + 1) + W * 64 + W) * sizeof(double);
}

// ---- B on the matrix cores --------------------------------------------------------------------------------------
// The triangle of a virtual supernode (w <= 64 columns) as a blocked TRSM on 16 x 16 tiles, the scheme of the dense-block
// cholsol kernel (csx_chol.hip, k_cholsol_mfma):  forward X_i <- W_ii (X_i - sum_{j<i} L_ij X_j), backward
// X_i <- W_ii' (X_i - sum_{j>i} L_ji' X_j), W_ii = inv(L_ii) formed with the plan.  One wave per (virtual supernode, 64
// right-hand sides): the nb x 4 tiles of X live in registers (accumulator layout = B-operand layout, so a finished tile
// feeds the next product as it stands), a tile of L reaches the pipe as one double per lane from a fragment array that
// is read once, coalesced.  What it replaces (k_sn_tri) stages the triangle in LDS and walks 64 dependent rows: 33 - 47 us
// per 64-column chunk on bcsstk16 against ~10 here, and that chain of steps IS the solve time of such a factor.
// Fragments of a virtual supernode: tiles in the order (0,0) (1,0) (1,1) (2,0) .. of its nb = ceil(w / 16) block rows,
// four k-steps of 64 doubles each; off-diagonal tiles hold -L_ij, diagonal ones W_ii; rows and columns past w are
// those of the identity.  The backward sweep reads the same fragments transposed (lane and k-step exchange roles).
typedef double sn_f64x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int sn_tiles(int nb) { return nb * (nb + 1) / 2; }

// one wave per triangle: list entry (a, w, first fragment, -).
// LEAF == false: a virtual supernode, columns a .. a + w - 1 of L (dense inside a supernode: column a + t is its diagonal, then
// rows a + t + 1 ..).  LEAF == true: a leaf subtree, positions a .. a + w - 1 of leaf_cols; column k's in-subtree entries come
// from the packed backward program (position of the row in the subtree, value) -- the subtree's triangle made dense, the
// entries its pattern lacks being zeros.
// MAXW: the widest triangle of the list (64; 128 for the chunks of a plan built with chunks of 128 columns)
template <int MAXW>
constexpr size_t sn_frags_lds() {
    return (size_t)(MAXW * (MAXW + 1) + (MAXW / 16) * 16 * 17 + MAXW) * sizeof(double);
}

template <bool LEAF, int MAXW>
__global__ __launch_bounds__(64) void k_sn_frags(const int4 *__restrict__ list, int32_t count, const int32_t *__restrict__ Lp,
                                                 const int32_t *__restrict__ Li, const double *__restrict__ Lx,
                                                 const int32_t *__restrict__ lb_ptr,
                                                 const int32_t *__restrict__ lb_idx, const double *__restrict__ lb_val,
                                                 const double *__restrict__ ldiag, double *__restrict__ frags,
                                                 const int64_t t_off, unsigned long long *cond_bits) {
    extern __shared__ __attribute__((aligned(16))) double sn_fr_smem[];     // sn_frags_lds<MAXW>() bytes
    double(*Ls)[MAXW + 1] = reinterpret_cast<double(*)[MAXW + 1]>(sn_fr_smem);
    double(*Wm)[16][17] = reinterpret_cast<double(*)[16][17]>(sn_fr_smem + MAXW * (MAXW + 1));
    double *rsum = sn_fr_smem + MAXW * (MAXW + 1) + (MAXW / 16) * 16 * 17;
    const int lane = threadIdx.x;
    if ((int32_t)blockIdx.x >= count) return;
    const int4 ent = list[blockIdx.x];
    const int32_t a = ent.x, w = ent.y;
    const int nb = (w + 15) >> 4;
    for (int e = lane; e < MAXW * MAXW; e += 64) Ls[e / MAXW][e % MAXW] = (e / MAXW) == (e % MAXW) ? 1.0 : 0.0;
    __syncthreads();
    if (LEAF) {
        for (int k = 0; k < w; k++) {                   // column k of the subtree: distinct rows, so no two lanes meet
            const int32_t b = lb_ptr[a + k], l = lb_ptr[a + k + 1] - b;
            if (lane < l) Ls[lb_idx[b + lane]][k] = lb_val[b + lane];
            if (lane == 0) Ls[k][k] = ldiag[a + k];
        }
    } else {
        for (int t = 0; t < w; t++) {                   // column a + t: diagonal, then its rows inside the triangle (fewer
            const int32_t b = Lp[a + t], cnt = Lp[a + t + 1] - b;   // than MAXW, the first of the column: rows ascend)
            for (int q = lane; q < cnt && q < MAXW; q += 64) {
                const int32_t i = Li[b + q] - a;        // a fundamental supernode has all of t .. w - 1, a relaxed one some
                if (i < w) Ls[i][t] = Lx[b + q];
            }
        }
    }
    __syncthreads();
    for (int pass = 0; pass < MAXW / 64; pass++) {
        const int blk = (lane >> 4) + 4 * pass, col = lane & 15;
        double rs = 0.0;                                    // row sum of |L_ii| for row `col` of diagonal block blk
        if (blk < nb) {
            double wcol[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                double sres = r == col ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < r; q++) sres -= Ls[16 * blk + r][16 * blk + q] * (q >= col ? wcol[q] : 0.0);
                wcol[r] = r >= col ? sres / Ls[16 * blk + r][16 * blk + r] : 0.0;
                Wm[blk][r][col] = wcol[r];
                rs += fabs(Ls[16 * blk + col][16 * blk + r]);
            }
        }
        rsum[lane + 64 * pass] = rs;
    }
    __syncthreads();
    // What an explicit inverse costs in accuracy: X_i <- W_ii r with r = L_ii x carries a relative error of about
    // eps || |W_ii| |L_ii| ||_inf (the condition of the 16 x 16 diagonal block alone, whatever the scaling of its rows;
    // off-diagonal tiles enter as plain products).  Row `col` of that matrix sums to sum_k |W(col, k)| rsum(k).
    double growth = 0.0;
    for (int pass = 0; pass < MAXW / 64; pass++) {
        const int blk = (lane >> 4) + 4 * pass, col = lane & 15;
        double g = 0.0;
        if (blk < nb)
#pragma unroll
            for (int k = 0; k < 16; k++) g += fabs(Wm[blk][col][k]) * rsum[16 * blk + k];
        growth = (g > growth || !(g >= 0.0)) ? g : growth;   // a NaN wins
    }
    const int m = lane & 15, kq = lane >> 4;
    // forward fragments, and at the same tile positions t_off doubles further on the TRANSPOSED tiles for the backward sweep
    // (read out of the forward ones with lane and k-step exchanged they cost a 16-way LDS bank conflict per read: 21 us
    // against 11 for a 128-column triangle)
    double *F = frags + (size_t)ent.z * 64 + lane, *T = F + t_off;
    int f = 0;
    for (int i = 0; i < nb; i++) {
        for (int j = 0; j < i; j++)
            for (int sx = 0; sx < 4; sx++) {
                F[64 * f] = -Ls[16 * i + m][16 * j + 4 * sx + kq];
                T[64 * f++] = -Ls[16 * i + 4 * sx + kq][16 * j + m];
            }
        for (int sx = 0; sx < 4; sx++) {
            F[64 * f] = Wm[i][m][4 * sx + kq];
            T[64 * f++] = Wm[i][4 * sx + kq][m];
        }
    }
    if (!(growth >= 0.0)) growth = __longlong_as_double(0x7ff0000000000000ll);   // a NaN (zero pivot): refuse
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) growth = fmax(growth, __shfl_xor(growth, d, 64));
    if (lane == 0) atomicMax(cond_bits, (unsigned long long)__double_as_longlong(growth));   // ordered bits of a non-negative double
}

// GATHER: the rows of X are rows[a .. a + w) (a leaf subtree's columns) instead of a .. a + w - 1
// CT = 4: one wave per (triangle, 64 right-hand sides), four column tiles in its registers.  CT = 1: a wave per column tile of
// 16 right-hand sides (a workgroup of up to four waves per block of 64): each wave's chain of dependent matrix
// instructions is a quarter as long, which is what a step with a handful of triangles is made of.
// use0: the step has ONE triangle and its descriptor came with the launch (ent0) instead of through a load.
// NBMAX: block rows of 16 a triangle can have (4; 8 for the 128-column chunks of a plan built with such chunks, CT = 1 then).
template <bool FWD, bool GATHER, int CT, int NBMAX>
__global__ __launch_bounds__(256) void k_sn_mfma(const int4 *__restrict__ list, int32_t first, const int4 ent0, const int use0,
                                                 const int32_t *__restrict__ rows, const double *__restrict__ frags,
                                                 const int64_t t_off, double *X, int nrhs) {
    // The triangle's fragments are staged in LDS by global -> LDS DMA, all of them requested at once by the workgroup's
    // waves together: read straight into registers they came in a dozen dependent batches (as many as the registers held),
    // each a memory round trip -- 15 - 24 us for the 72 KB of a 128-column triangle, 7 - 9 us for the 20 KB of a 64-column one.
    extern __shared__ __attribute__((aligned(16))) double sn_fr[];      // sn_tiles(nb) * 256 doubles
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = (int)(blockDim.x >> 6);
    const int cbase = CT == 4 ? 0 : wv;
    const int nblk = (nrhs + 63) >> 6;
    const int h = (int)(blockIdx.x % nblk);
    const bool active = h * 64 + 16 * cbase < nrhs;     // (a column tile past the last right-hand side still helps to stage)
    const int4 ent = use0 ? ent0 : list[first + blockIdx.x / nblk];
    const int32_t a = ent.x, w = ent.y;
    const int nb = (w + 15) >> 4;                       // uniform
    const int col = lane & 15, rq = lane >> 4;
    bool live[CT];
    int32_t cidx[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int32_t rhs = h * 64 + 16 * (cbase + c) + col;
        live[c] = rhs < nrhs;
        cidx[c] = live[c] ? rhs : nrhs - 1;             // clamped: loaded, never stored
    }
    // lane (rq, col), register r of tile (i, c): row 16 i + rq + 4 r of the supernode, right-hand side 16 c + col
    sn_f64x4 Xt[NBMAX][CT];
#pragma unroll
    for (int i = 0; i < NBMAX; i++)
#pragma unroll
        for (int c = 0; c < CT; c++) Xt[i][c] = sn_f64x4{0.0, 0.0, 0.0, 0.0};
    int64_t roff[NBMAX][4];
#pragma unroll
    for (int i = 0; i < NBMAX; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * i + rq + 4 * r;
            const int32_t at = a + (row < w ? row : 0);
            roff[i][r] = (int64_t)(GATHER ? (i < nb ? rows[at] : 0) : at) * nrhs;
        }
#pragma unroll
    for (int i = 0; i < NBMAX; i++)
        if (i < nb) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * i + rq + 4 * r;
                const double *src = X + roff[i][r];
#pragma unroll
                for (int c = 0; c < CT; c++) {
                    const double v = src[cidx[c]];
                    Xt[i][c][r] = row < w ? v : 0.0;    // rows past w: those of the identity block, kept at zero
                }
            }
        }
    {
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const double *gF = frags + (FWD ? 0 : t_off) + (size_t)ent.z * 64;   // backward: the transposed tiles
        const int nk = sn_tiles(nb) * 2;                // pieces of 128 doubles (64 lanes x 16 bytes)
        for (int k = wv; k < nk; k += nwv)
            __builtin_amdgcn_global_load_lds((gptr_t)(gF + k * 128 + 2 * lane), (lptr_t)(sn_fr + k * 128), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!active) return;
    const double *F = sn_fr + lane;
    auto tile_at = [](int p, int q) { return (p * (p + 1) / 2 + q) * 4; };   // first of the tile's four fragments
    if (FWD) {
        int f = 0;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
#pragma unroll
                for (int j = 0; j < i; j++)
#pragma unroll
                    for (int sx = 0; sx < 4; sx++) {
                        const double av = F[64 * f++];
#pragma unroll
                        for (int c = 0; c < CT; c++) Xt[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Xt[j][c][sx], Xt[i][c], 0, 0, 0);
                    }
                sn_f64x4 Y[CT];
#pragma unroll
                for (int c = 0; c < CT; c++) Y[c] = sn_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sx = 0; sx < 4; sx++) {
                    const double av = F[64 * f++];
#pragma unroll
                    for (int c = 0; c < CT; c++) Y[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Xt[i][c][sx], Y[c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < CT; c++) Xt[i][c] = Y[c];
            }
    } else {
#pragma unroll
        for (int i = NBMAX - 1; i >= 0; i--)
            if (i < nb) {
#pragma unroll
                for (int j = i + 1; j < NBMAX; j++)
                    if (j < nb) {
#pragma unroll
                        for (int sx = 0; sx < 4; sx++) {
                            const double av = F[64 * (tile_at(j, i) + sx)];            // -L_ji'
#pragma unroll
                            for (int c = 0; c < CT; c++)
                                Xt[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Xt[j][c][sx], Xt[i][c], 0, 0, 0);
                        }
                    }
                sn_f64x4 Y[CT];
#pragma unroll
                for (int c = 0; c < CT; c++) Y[c] = sn_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sx = 0; sx < 4; sx++) {
                    const double av = F[64 * (tile_at(i, i) + sx)];                    // W_ii'
#pragma unroll
                    for (int c = 0; c < CT; c++) Y[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Xt[i][c][sx], Y[c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < CT; c++) Xt[i][c] = Y[c];
            }
    }
#pragma unroll
    for (int i = 0; i < NBMAX; i++)
        if (i < nb) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * i + rq + 4 * r;
                if (row < w) {
                    double *dst = X + roff[i][r];
#pragma unroll
                    for (int c = 0; c < CT; c++)
                        if (live[c]) dst[cidx[c]] = Xt[i][c][r];
                }
            }
        }
}

// Leaf subtrees: one wave per (subtree, 64 right-hand sides), the subtree's x in LDS (position in the subtree * 64 + lane),
// the terms from a packed program: line k of the subtree (a row forward, a column backward) has its in-subtree terms at
// [ptr[c0 + k], ptr[c0 + k + 1]) as (position of the source in the subtree, value); at most 63 of them, so one load per
// lane fetches a line, and the next line's load is in flight while this one is summed.  FWD: lines ascending.
template <bool FWD>
__global__ __launch_bounds__(64) void k_sn_leaf(const int32_t *__restrict__ leaf_ptr, const int32_t *__restrict__ leaf_cols,
                                                const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                const double *__restrict__ val, const double *__restrict__ ldiag,
                                                const int32_t *__restrict__ slot_ptr, const double *__restrict__ partial, double *X,
                                                int nrhs) {
    __shared__ double xs[SN_LEAF * 64];
    const int lane = threadIdx.x;
    const int nblk = (nrhs + 63) >> 6;
    const int32_t t = blockIdx.x / nblk;
    const int r = (int)(blockIdx.x % nblk) * 64 + lane;
    const bool live = r < nrhs;
    const int rr = live ? r : nrhs - 1;
    const int32_t c0 = leaf_ptr[t], cnt = leaf_ptr[t + 1] - c0;
    const int32_t myptr = lane < cnt ? ptr[c0 + lane] : 0;
    const int32_t mylen = lane < cnt ? ptr[c0 + lane + 1] - myptr : 0;
    const int32_t myline = lane < cnt ? leaf_cols[c0 + lane] : 0;
    const double mydg = lane < cnt ? ldiag[c0 + lane] : 1.0;
    const int dlo = __double2loint(mydg), dhi = __double2hiint(mydg);
    // backward: partial sums of a long column's rows outside the subtree (k_sn_outside wrote them), slots [s0, s1)
    const int32_t mys0 = slot_ptr && lane < cnt ? slot_ptr[c0 + lane] : 0, mys1 = slot_ptr && lane < cnt ? slot_ptr[c0 + lane + 1] : 0;
    struct Line {
        int32_t li;
        double cv, xb;
    };
    auto load = [&](int k) {
        const int32_t p = __builtin_amdgcn_readlane(myptr, k), l = __builtin_amdgcn_readlane(mylen, k);
        const int32_t line = __builtin_amdgcn_readlane(myline, k);
        Line L;
        L.li = lane < l ? idx[p + lane] : 0;
        L.cv = lane < l ? val[p + lane] : 0.0;
        L.xb = X[(int64_t)line * nrhs + rr];
        const int32_t s0 = __builtin_amdgcn_readlane(mys0, k), s1 = __builtin_amdgcn_readlane(mys1, k);
        for (int32_t sl = s0; sl < s1; sl++) L.xb -= partial[(int64_t)sl * nrhs + rr];
        return L;
    };
    auto compute = [&](const Line &L, int k) {
        const int32_t l = __builtin_amdgcn_readlane(mylen, k), line = __builtin_amdgcn_readlane(myline, k);
        const int clo = __double2loint(L.cv), chi = __double2hiint(L.cv);
        double acc0 = L.xb, acc1 = 0.0;
        int u = 0;
        for (; u + 1 < l; u += 2) {
            const int32_t ca = __builtin_amdgcn_readlane(L.li, u), cb = __builtin_amdgcn_readlane(L.li, u + 1);
            const double va = __hiloint2double(__builtin_amdgcn_readlane(chi, u), __builtin_amdgcn_readlane(clo, u));
            const double vb = __hiloint2double(__builtin_amdgcn_readlane(chi, u + 1), __builtin_amdgcn_readlane(clo, u + 1));
            acc0 = fma(-va, xs[ca * 64 + lane], acc0);
            acc1 = fma(-vb, xs[cb * 64 + lane], acc1);
        }
        if (u < l) {
            const int32_t ca = __builtin_amdgcn_readlane(L.li, u);
            const double va = __hiloint2double(__builtin_amdgcn_readlane(chi, u), __builtin_amdgcn_readlane(clo, u));
            acc0 = fma(-va, xs[ca * 64 + lane], acc0);
        }
        const double d = __hiloint2double(__builtin_amdgcn_readlane(dhi, k), __builtin_amdgcn_readlane(dlo, k));
        const double y = (acc0 + acc1) / d;
        xs[k * 64 + lane] = y;
        if (live) X[(int64_t)line * nrhs + r] = y;
    };
    // lines in order (forward: 0 .. cnt - 1; backward: cnt - 1 .. 0), two register sets so that a load is always ahead
    auto at = [&](int step) { return FWD ? step : cnt - 1 - step; };
    if (cnt <= 0) return;
    Line R0 = load(at(0)), R1 = R0, R2 = R0, R3 = R0;     // a ring of four: three lines' loads in flight behind the sums
    if (cnt > 1) R1 = load(at(1));
    if (cnt > 2) R2 = load(at(2));
    for (int step = 0; step < cnt; step += 4) {
        if (step + 3 < cnt) R3 = load(at(step + 3));
        compute(R0, at(step));
        if (step + 4 < cnt) R0 = load(at(step + 4));
        if (step + 1 < cnt) compute(R1, at(step + 1));
        if (step + 5 < cnt) R1 = load(at(step + 5));
        if (step + 2 < cnt) compute(R2, at(step + 2));
        if (step + 6 < cnt) R2 = load(at(step + 6));
        if (step + 3 < cnt) compute(R3, at(step + 3));
    }
}

// ---- plan-time kernels of the leaf subtrees ----
// position k in leaf_cols (column j of subtree t): number of forward terms (the whole row: all of it lies in the subtree),
// number of backward terms inside the subtree (a column's rows are ancestors: those of the subtree come first), the task
// for the rest of the column, the diagonal.
__global__ void k_sn_leaf_meta(int32_t ncols, const int32_t *__restrict__ leaf_cols, const int32_t *__restrict__ sub_of,
                               const int32_t *__restrict__ Gp, const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                               const double *__restrict__ Lx, int32_t *lenf, int32_t *lenb, int32_t *split, int32_t *ntask,
                               int32_t *nslot, double *ldiag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int32_t j = leaf_cols[k], t = sub_of[j];
    lenf[k] = Gp[j + 1] - Gp[j];
    int32_t lo = Lp[j] + 1, hi = Lp[j + 1];
    const int32_t end = hi;
    while (lo < hi) {                      // first row that is not of subtree t
        const int32_t mid = (lo + hi) >> 1;
        if (sub_of[Li[mid]] == t) lo = mid + 1;
        else hi = mid;
    }
    lenb[k] = lo - (Lp[j] + 1);
    split[k] = lo;
    const int32_t len = end - lo, pieces = (len + SN_SEG - 1) / SN_SEG;
    ntask[k] = pieces;
    nslot[k] = pieces > 1 ? pieces : 0;
    ldiag[k] = Lx[Lp[j]];
}

// the tasks of the leaf columns' rows outside their subtrees: one per piece of SN_SEG terms; a column of several pieces
// sums into partial slots (slot0 + its slot numbers), a column of one piece straight into X
__global__ void k_sn_leaf_tasks(int32_t ncols, const int32_t *__restrict__ leaf_cols, const int32_t *__restrict__ Lp,
                                const int32_t *__restrict__ split, const int32_t *__restrict__ task_ptr,
                                const int32_t *__restrict__ slot_ptr, int32_t slot0, SnTask *tasks) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int32_t j = leaf_cols[k], b = split[k], e = Lp[j + 1];
    int32_t t = task_ptr[k], sl = slot0 + slot_ptr[k];
    const bool many = task_ptr[k + 1] - t > 1;
    for (int32_t q = b; q < e; q += SN_SEG) tasks[t++] = SnTask{j, q, q + SN_SEG < e ? q + SN_SEG : e, many ? sl++ : -1};
}

__global__ __launch_bounds__(256) void k_sn_leaf_fill(int32_t ncols, const int32_t *__restrict__ leaf_cols,
                                                      const int32_t *__restrict__ local_of, const int32_t *__restrict__ Gp,
                                                      const int32_t *__restrict__ Gi, const double *__restrict__ Gx,
                                                      const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                                      const double *__restrict__ Lx, const int32_t *__restrict__ lf_ptr,
                                                      int32_t *lf_idx, double *lf_val, const int32_t *__restrict__ lb_ptr,
                                                      int32_t *lb_idx, double *lb_val) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= ncols) return;
    const int32_t j = leaf_cols[k];
    const int32_t fb = lf_ptr[k], fl = lf_ptr[k + 1] - fb, gb = Gp[j];
    for (int32_t q = lane; q < fl; q += 64) {
        lf_idx[fb + q] = local_of[Gi[gb + q]];
        lf_val[fb + q] = Gx[gb + q];
    }
    const int32_t bb = lb_ptr[k], bl = lb_ptr[k + 1] - bb, cb = Lp[j] + 1;
    for (int32_t q = lane; q < bl; q += 64) {
        lb_idx[bb + q] = local_of[Li[cb + q]];
        lb_val[bb + q] = Lx[cb + q];
    }
}

// The schedule rests on properties every Cholesky factor has and an arbitrary lower triangle may not: the columns of
// a supernode share their rows (column j's rows are column j - 1's minus its diagonal), a column's rows are ancestors
// of the column (greater forward level, smaller backward level; inside a leaf subtree: the same subtree or outside
// any), rows ascending without duplicates.  One pass over the pattern checks all of it.  fl / bl: forward / backward
// level of every column (leaf subtrees: -1 / INT_MAX), sn: supernode of a column (leaf: -1 - subtree).
__global__ __launch_bounds__(256) void k_sn_verify(int32_t n, const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                                   const int32_t *__restrict__ sn, const int32_t *__restrict__ joins,
                                                   const int32_t *__restrict__ fl, const int32_t *__restrict__ bl, int *bad) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int32_t b = Lp[j], e = Lp[j + 1], S = sn[j];
    const bool jn = joins[j] == 1;                   // (2: a relaxed join -- the columns need not share their rows)
    const int32_t pb = jn ? Lp[j - 1] : 0;
    bool wrong = e - b < 1 || Li[b] != (int32_t)j;
    for (int32_t q = b + lane; q < e; q += 64) {
        const int32_t i = Li[q];
        if ((uint32_t)i >= (uint32_t)n) {
            wrong = true;
            continue;
        }
        if (jn && Li[pb + (q - b) + 1] != i) wrong = true;
        if (q > b) {
            if (i <= Li[q - 1]) wrong = true;            // rows ascending, no duplicates
            const int32_t T = sn[i];
            if (T != S && (fl[i] <= fl[j] || bl[i] >= bl[j])) wrong = true;
        }
    }
    if (wrong) atomicOr(bad, 1);
}

// Line j of a supernode with columns [a, e): fin[j] = terms of row j (forward plan's row-major copy, ascending columns) that
// lie inside the supernode (columns >= a: the tail of the row); bin[j] = entries of column j below its diagonal that lie
// inside (rows < e: the head of the column).  In a fundamental supernode these are j - a and e - 1 - j; in a relaxed one
// (a run of a chain of the tree whose columns do not share their rows) they have to be looked up.
__global__ void k_sn_incount(int32_t n, const int32_t *__restrict__ sn_a, const int32_t *__restrict__ sn_e,
                             const int32_t *__restrict__ Gp, const int32_t *__restrict__ Gi, const int32_t *__restrict__ Lp,
                             const int32_t *__restrict__ Li, int32_t *fin, int32_t *bin) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t a = sn_a[j], e = sn_e[j];
    if (a < 0) {
        fin[j] = 0;
        bin[j] = 0;
        return;
    }
    int32_t lo = Gp[j], hi = Gp[j + 1];               // first term with column >= a
    const int32_t gend = hi;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (Gi[mid] < a) lo = mid + 1;
        else hi = mid;
    }
    fin[j] = gend - lo;
    lo = Lp[j] + 1;                                   // first row >= e
    hi = Lp[j + 1];
    const int32_t cb = lo;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (Li[mid] < e) lo = mid + 1;
        else hi = mid;
    }
    bin[j] = lo - cb;
}

}  // namespace

/* Build the supernodal schedule of a Cholesky factor.  parent: elimination tree (host, n), Lp_h / Gp_h: host copies of
 * L's column pointers and of the forward plan's row pointers (off-diagonal terms of each row); G*: that row-major copy
 * on the device.  *out = nullptr when the factor gains nothing from it (no supernodes to speak of, or a chain). */
static int sn_build_with(const Csc *L, const int32_t *parent, const int32_t *Lp_h, const int32_t *Gp_h, const int32_t *Gp,
                         const int32_t *Gi, const double *Gx, int32_t col_levels, const int chunk, SnPlan **out) {
    const int32_t n = L->n;
    *out = nullptr;
    if (n < 2) return CSX_OK;
    const bool say = std::getenv("CSX_CHOL_TIMING") != nullptr;
    const bool allow_relaxed = ctx().opt.tri_supernodes == 1;
    // (twigs of fewer than eight columns are left to the relaxed ranges below: as leaves they would cut the chain they hang off)
    const int32_t leaf_min = allow_relaxed ? 8 : 1;
    // ---- leaf subtrees: maximal subtrees of at most SN_LEAF columns ----
    std::vector<int32_t> size((size_t)n, 1), sub_of((size_t)n, -1), local_of((size_t)n, -1);
    for (int32_t j = 0; j < n; j++)
        if (parent[j] >= 0) size[(size_t)parent[j]] += size[(size_t)j];
    std::vector<int32_t> leaf_ptr, leaf_cols;
    int32_t nleaf = 0;
    for (int32_t j = n - 1; j >= 0; j--) {            // parents before children
        const int32_t pj = parent[j];
        if (pj >= 0 && sub_of[(size_t)pj] >= 0) sub_of[(size_t)j] = sub_of[(size_t)pj];
        else if (size[(size_t)j] <= SN_LEAF && size[(size_t)j] >= leaf_min) sub_of[(size_t)j] = nleaf++;
    }
    {
        std::vector<int32_t> cnt((size_t)nleaf + 1, 0);
        for (int32_t j = 0; j < n; j++)
            if (sub_of[(size_t)j] >= 0) cnt[(size_t)sub_of[(size_t)j] + 1]++;
        for (int32_t t = 0; t < nleaf; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
        leaf_ptr = cnt;
        leaf_cols.assign((size_t)cnt[(size_t)nleaf], 0);
        std::vector<int32_t> at(cnt.begin(), cnt.end() - 1);
        for (int32_t j = 0; j < n; j++)                // ascending inside a subtree: a topological order
            if (sub_of[(size_t)j] >= 0) {
                const int32_t t = sub_of[(size_t)j];
                local_of[(size_t)j] = at[(size_t)t] - leaf_ptr[(size_t)t];
                leaf_cols[(size_t)at[(size_t)t]++] = j;
            }
    }
    // ---- supernodes of the columns outside the leaf subtrees ----
    // A column joins the supernode of its predecessor when it is that column's parent and
    //   (1) FUNDAMENTAL: its count is the predecessor's less one -- the two share their rows, the supernode is a dense
    //       trapezoid of any width (cut into chunks of `chunk` columns below); or
    //   (2) RELAXED (only with the matrix-core triangles): nothing more -- a run of at most `chunk` columns of a CHAIN of the
    //       tree.  Its triangle is made dense in the fragments (zeros where the pattern has none), the split of a row /
    //       column into its part inside and outside the run is looked up (k_sn_incount).  This is what gives a banded
    //       factor in natural order (one chain, no two columns with the same rows) a schedule of n / 64 steps.
    // A supernode is of one kind: a fundamental one does not continue with relaxed joins, nor the other way round.
    // (The relaxed kind is a RANGE of consecutive columns [a, e), e - a <= chunk, every one of which but the last has its
    // parent inside the range: a chain, or a chain with the twigs that hang off it -- bcsstk16 in natural order is 4 810
    // levels of chain for 4 884 columns with twigs of one to three columns every few steps.  Inside such a range every
    // source of a row with index >= a is in the range, so the in / out split of k_sn_incount holds.  A fundamental run of 32
    // or more columns stays fundamental (its chunks then need no looking up), shorter ones may end up inside a range.)
    std::vector<int32_t> first, sn_of((size_t)n, -1), joins((size_t)n, 0);
    bool any_relaxed = false;
    {
        auto fund_join = [&](int32_t j) {                        // column j continues a fundamental supernode of j - 1
            return j > 0 && sub_of[(size_t)j] < 0 && sub_of[(size_t)j - 1] < 0 && parent[j - 1] == j &&
                   (Lp_h[j + 1] - Lp_h[j]) == (Lp_h[j] - Lp_h[j - 1]) - 1;
        };
        std::vector<int8_t> frun((size_t)n + 1, 1);              // width of the fundamental run that starts at a column, capped at 32
        for (int32_t c = n - 2; c >= 0; c--) frun[(size_t)c] = fund_join(c + 1) ? (int8_t)std::min(32, frun[(size_t)c + 1] + 1) : (int8_t)1;
        int kind = 0;                                            // of the supernode being grown
        int32_t rel_end = -1;                                    // end of the relaxed range being filled
        for (int32_t j = 0; j < n; j++) {
            if (sub_of[(size_t)j] >= 0) continue;
            int jn = 0;
            if (j < rel_end) jn = 2;
            else if (kind != 2 && fund_join(j)) jn = 1;
            if (!jn) {
                first.push_back(j);
                kind = 0;
                rel_end = -1;
                if (allow_relaxed) {
                    if (frun[(size_t)j] < 32) {
                        // the longest range [j, e) whose columns before the last all have their parent inside; it stops
                        // short of a wide fundamental supernode
                        int32_t best = j + 1, maxpar = parent[j];
                        for (int32_t c = j + 1; c < n && c - j < chunk && sub_of[(size_t)c] < 0 && maxpar >= 0; c++) {
                            if (frun[(size_t)c] >= 32 && !fund_join(c)) break;
                            if (maxpar <= c) best = c + 1;       // [j, c + 1): the parents of j .. c - 1 lie at or before c
                            maxpar = std::max(maxpar, parent[c] < 0 ? 0x7fffffff : parent[c]);
                            if (parent[c] < 0) break;            // a root: nothing can follow it inside a range
                        }
                        if (best > j + 1) rel_end = best;
                    }
                }
            } else {
                kind = jn;
                any_relaxed |= jn == 2;
            }
            joins[(size_t)j] = jn;
            sn_of[(size_t)j] = (int32_t)first.size() - 1;
        }
    }
    const int32_t nsn = (int32_t)first.size();
    if (nsn == 0) return CSX_OK;                       // a forest of small trees: the fused per-tree kernels' business
    std::vector<int32_t> width((size_t)nsn, 0);
    for (int32_t j = 0; j < n; j++)
        if (sn_of[(size_t)j] >= 0) width[(size_t)sn_of[(size_t)j]]++;
    std::vector<int32_t> sparent((size_t)nsn, -1), height((size_t)nsn, 0), depth((size_t)nsn, 0);
    int32_t max_w = 0;
    for (int32_t S = 0; S < nsn; S++) {
        const int32_t last = first[(size_t)S] + width[(size_t)S] - 1;
        max_w = std::max(max_w, width[(size_t)S]);
        if (parent[last] >= 0) sparent[(size_t)S] = sn_of[(size_t)parent[last]];
    }
    for (int32_t S = 0; S < nsn; S++)
        if (sparent[(size_t)S] >= 0) height[(size_t)sparent[(size_t)S]] = std::max(height[(size_t)sparent[(size_t)S]], height[(size_t)S] + 1);
    for (int32_t S = nsn - 1; S >= 0; S--)
        if (sparent[(size_t)S] >= 0) depth[(size_t)S] = depth[(size_t)sparent[(size_t)S]] + 1;
    int32_t nlev = 0;
    for (int32_t S = 0; S < nsn; S++) nlev = std::max(nlev, height[(size_t)S] + 1);
    // worth it?  the supernodal schedule must be much shorter than the column one
    int64_t est_steps = 0;
    {
        std::vector<int32_t> lev_w((size_t)nlev, 0);
        for (int32_t S = 0; S < nsn; S++) lev_w[(size_t)height[(size_t)S]] = std::max(lev_w[(size_t)height[(size_t)S]], width[(size_t)S]);
        for (int32_t l = 0; l < nlev; l++) est_steps += (lev_w[(size_t)l] + chunk - 1) / chunk;
    }
    if (say)
        std::fprintf(stderr, "sn_build: n %d: %d leaf subtrees (%d columns), %d supernodes in %d levels, ~%d steps (column levels: %d), max width %d\n",
                     n, nleaf, (int)leaf_cols.size(), nsn, nlev, (int)est_steps, col_levels, max_w);
    // (each step is two or three dependent launches of ~10 us: a chain of a thousand narrow supernodes -- bcsstk16 in natural
    // order: ~1 500 steps for 4 810 column levels -- took 38 ms this way against 3.5 ms for the chain walkers)
    if (est_steps * 8 > col_levels) return CSX_OK;
    SnPlan *P = new SnPlan();
    P->n = n;
    P->chunk = chunk;
    P->nsn = nsn;
    P->max_w = max_w;
    P->nleaf = nleaf;
    P->nleafcols = (int32_t)leaf_cols.size();
    hipStream_t s = ctx().stream;
    int st = CSX_OK;
    {   // is L shaped like a Cholesky factor with these supernodes and subtrees?
        std::vector<int32_t> snc((size_t)n), fl((size_t)n), bl((size_t)n);
        for (int32_t j = 0; j < n; j++) {
            const int32_t S = sn_of[(size_t)j];
            snc[(size_t)j] = S >= 0 ? S : -1 - sub_of[(size_t)j];
            fl[(size_t)j] = S >= 0 ? height[(size_t)S] : -1;
            bl[(size_t)j] = S >= 0 ? depth[(size_t)S] : 0x7fffffff;
        }
        DevScope tmp;
        int32_t *d_sn = nullptr, *d_j = nullptr, *d_f = nullptr, *d_b = nullptr;
        int *d_bad = nullptr, h_bad = 0;
        const std::pair<int32_t **, std::vector<int32_t> *> ups[] = {{&d_sn, &snc}, {&d_j, &joins}, {&d_f, &fl}, {&d_b, &bl}};
        for (const auto &pr : ups) {
            if (st == CSX_OK) st = tmp.alloc(pr.first, (size_t)n);
            if (st == CSX_OK && hipMemcpyAsync(*pr.first, pr.second->data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
        if (st == CSX_OK) st = tmp.alloc(&d_bad, 1);
        if (st == CSX_OK && hipMemsetAsync(d_bad, 0, sizeof(int), s) != hipSuccess) st = CSX_ERUNTIME;
        if (st == CSX_OK) {
            hipLaunchKernelGGL(k_sn_verify, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, L->p, L->i, d_sn, d_j, d_f, d_b,
                               d_bad);
            if (hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
        if (say) std::fprintf(stderr, "sn_build: verify %s\n", h_bad ? "FAILED" : "ok");
        if (st != CSX_OK || h_bad) {            // not such a factor: the level-scheduled plans stay in charge
            free_snplan(P);
            return st;
        }
    }
    // how much of every row / column lies inside its supernode
    std::vector<int32_t> fin_h((size_t)n, 0), bin_h((size_t)n, 0);
    P->has_relaxed = any_relaxed;
    if (!any_relaxed) {
        for (int32_t S = 0; S < nsn; S++)
            for (int32_t v = 0; v < width[(size_t)S]; v++) {
                fin_h[(size_t)(first[(size_t)S] + v)] = v;
                bin_h[(size_t)(first[(size_t)S] + v)] = width[(size_t)S] - v - 1;
            }
    } else {
        std::vector<int32_t> sa((size_t)n, -1), se((size_t)n, -1);
        for (int32_t S = 0; S < nsn; S++)
            for (int32_t v = 0; v < width[(size_t)S]; v++) {
                sa[(size_t)(first[(size_t)S] + v)] = first[(size_t)S];
                se[(size_t)(first[(size_t)S] + v)] = first[(size_t)S] + width[(size_t)S];
            }
        DevScope tmp;
        int32_t *d_a = nullptr, *d_e = nullptr, *d_f = nullptr, *d_b = nullptr;
        for (int32_t **a : {&d_a, &d_e, &d_f, &d_b})
            if (st == CSX_OK) st = tmp.alloc(a, (size_t)n);
        if (st == CSX_OK &&
            (hipMemcpyAsync(d_a, sa.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
             hipMemcpyAsync(d_e, se.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess))
            st = CSX_ERUNTIME;
        if (st == CSX_OK) {
            hipLaunchKernelGGL(k_sn_incount, dim3((unsigned)(((int64_t)n + 255) / 256)), dim3(256), 0, s, n, d_a, d_e, Gp, Gi, L->p, L->i, d_f,
                               d_b);
            if (hipMemcpyAsync(fin_h.data(), d_f, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipMemcpyAsync(bin_h.data(), d_b, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                st = CSX_ERUNTIME;
        }
        if (st != CSX_OK) {
            free_snplan(P);
            return st;
        }
    }
    // partial slots of the backward schedule (a column whose rows below its supernode are cut into pieces): counted here
    // because the leaf columns' slots are numbered behind them
    int32_t bwd_slots = 0;
    for (int32_t S = 0; S < nsn; S++)
        for (int32_t v = 0; v < width[(size_t)S]; v++) {
            const int32_t col = first[(size_t)S] + v, c = Lp_h[col + 1] - (Lp_h[col] + 1 + bin_h[(size_t)col]);
            if (c > SN_WHOLE) bwd_slots += (c + SN_SEG - 1) / SN_SEG;
        }
    // ---- leaf subtrees: packed programs ----
    st = up(&P->leaf_ptr, leaf_ptr);
    if (st == CSX_OK) st = up(&P->leaf_cols, leaf_cols);
    if (st == CSX_OK && P->nleafcols > 0) {
        DevScope tmp;
        int32_t *d_sub = nullptr, *d_local = nullptr, *lenf = nullptr, *lenb = nullptr, *split = nullptr, *ntask = nullptr,
                *nslot = nullptr, *task_ptr = nullptr;
        const int32_t nc = P->nleafcols;
        st = tmp.alloc(&d_sub, (size_t)n);
        if (st == CSX_OK) st = tmp.alloc(&d_local, (size_t)n);
        for (int32_t **a : {&lenf, &lenb, &split, &ntask, &nslot, &task_ptr})
            if (st == CSX_OK) st = tmp.alloc(a, (size_t)nc + 1);
        if (st == CSX_OK) st = dalloc(&P->lf_ptr, (size_t)nc + 1);
        if (st == CSX_OK) st = dalloc(&P->lb_ptr, (size_t)nc + 1);
        if (st == CSX_OK) st = dalloc(&P->lslot_ptr, (size_t)nc + 1);
        if (st == CSX_OK) st = dalloc(&P->ldiag, (size_t)nc);
        if (st == CSX_OK &&
            (hipMemcpyAsync(d_sub, sub_of.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
             hipMemcpyAsync(d_local, local_of.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess))
            st = CSX_ERUNTIME;
        int64_t ftot = 0, btot = 0, ttot = 0, stot = 0;
        if (st == CSX_OK) {
            hipLaunchKernelGGL(k_sn_leaf_meta, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, nc, P->leaf_cols, d_sub, Gp, L->p,
                               L->i, L->x, lenf, lenb, split, ntask, nslot, P->ldiag);
            st = scan_exclusive_i32(lenf, P->lf_ptr, nc, &ftot);
            if (st == CSX_OK) st = scan_exclusive_i32(lenb, P->lb_ptr, nc, &btot);
            if (st == CSX_OK) st = scan_exclusive_i32(ntask, task_ptr, nc, &ttot);
            if (st == CSX_OK) st = scan_exclusive_i32(nslot, P->lslot_ptr, nc, &stot);
        }
        P->nleaftasks = (int32_t)ttot;
        P->nleafslots = (int32_t)stot;
        if (st == CSX_OK) st = dalloc(&P->leaf_tasks, (size_t)ttot + 1);
        // the leaf columns' partial slots come behind the backward schedule's own (bwd_slots: counted above)
        if (st == CSX_OK)
            hipLaunchKernelGGL(k_sn_leaf_tasks, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, nc, P->leaf_cols, L->p, split,
                               task_ptr, P->lslot_ptr, bwd_slots, P->leaf_tasks);
        if (st == CSX_OK) st = dalloc(&P->lf_idx, (size_t)ftot + 64);
        if (st == CSX_OK) st = dalloc(&P->lf_val, (size_t)ftot + 64);
        if (st == CSX_OK) st = dalloc(&P->lb_idx, (size_t)btot + 64);
        if (st == CSX_OK) st = dalloc(&P->lb_val, (size_t)btot + 64);
        if (st == CSX_OK) {
            hipLaunchKernelGGL(k_sn_leaf_fill, dim3((unsigned)(((int64_t)nc + 3) / 4)), dim3(256), 0, s, nc, P->leaf_cols, d_local, Gp, Gi,
                               Gx, L->p, L->i, L->x, P->lf_ptr, P->lf_idx, P->lf_val, P->lb_ptr, P->lb_idx, P->lb_val);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
        }
    }
    // ---- virtual supernodes: every supernode cut into chunks of `chunk` columns ----
    std::vector<int32_t> vs_a, vs_w, vs0((size_t)nsn + 1, 0);
    for (int32_t S = 0; S < nsn; S++) {
        vs0[(size_t)S] = (int32_t)vs_a.size();
        for (int32_t c = 0; c < width[(size_t)S]; c += chunk) {
            vs_a.push_back(first[(size_t)S] + c);
            vs_w.push_back(std::min(chunk, width[(size_t)S] - c));
        }
    }
    vs0[(size_t)nsn] = (int32_t)vs_a.size();
    std::vector<int32_t> vs_f(vs_a.size() + 1, 0);      // first fragment (64 doubles each) of every virtual supernode
    for (size_t c = 0; c < vs_a.size(); c++) vs_f[c + 1] = vs_f[c] + sn_tiles((vs_w[c] + 15) / 16) * 4;
    if (st == CSX_OK) st = up(&P->vs_a, vs_a);
    if (st == CSX_OK) st = up(&P->vs_w, vs_w);
    auto build = [&](SnDir &D, const std::vector<int32_t> &level, bool forward) -> int {
        int32_t Lv = 0;
        for (int32_t S = 0; S < nsn; S++) Lv = std::max(Lv, level[(size_t)S] + 1);
        std::vector<std::vector<int32_t>> by((size_t)Lv);
        for (int32_t S = 0; S < nsn; S++) by[(size_t)level[(size_t)S]].push_back(S);
        std::vector<SnTask> tasks;
        std::vector<int32_t> comb, narrow, medium, wide, part((size_t)n + 1, 0);
        std::vector<int4> tri;
        // terms of line a + v (v: position in its supernode of width w) that lie outside the supernode
        auto outside = [&](int32_t line, int32_t v, int32_t w, int32_t *b, int32_t *e) {
            (void)v;
            (void)w;
            if (forward) {                       // row: all but its terms inside the supernode (fundamental: the last v)
                *b = Gp_h[line];
                *e = Gp_h[line + 1] - fin_h[(size_t)line];
            } else {                             // column: the rows below the supernode (fundamental: from Lp[col] + (w - v) on)
                *b = Lp_h[line] + 1 + bin_h[(size_t)line];
                *e = Lp_h[line + 1];
            }
        };
        for (int32_t S = 0; S < nsn; S++)
            for (int32_t v = 0; v < width[(size_t)S]; v++) {
                int32_t b, e;
                outside(first[(size_t)S] + v, v, width[(size_t)S], &b, &e);
                const int32_t c = e - b;
                part[(size_t)(first[(size_t)S] + v) + 1] = c > SN_WHOLE ? (c + SN_SEG - 1) / SN_SEG : 0;
            }
        for (int32_t j = 0; j < n; j++) part[(size_t)j + 1] += part[(size_t)j];
        D.nslots = part[(size_t)n];
        for (int32_t l = 0; l < Lv; l++) {
            int32_t nsub = 1;
            for (int32_t S : by[(size_t)l]) nsub = std::max(nsub, vs0[(size_t)S + 1] - vs0[(size_t)S]);
            for (int32_t q = 0; q < nsub; q++) {
                SnStep stp{(int32_t)tasks.size(), 0, (int32_t)comb.size(), 0, (int32_t)narrow.size(), 0, (int32_t)medium.size(), 0,
                           (int32_t)wide.size(), 0, (int32_t)tri.size(), 0, 0};
                for (int32_t S : by[(size_t)l]) {
                    const int32_t a = first[(size_t)S], w = width[(size_t)S], nch = vs0[(size_t)S + 1] - vs0[(size_t)S];
                    if (q >= nch) continue;
                    if (q == 0) {
                        // the terms from outside the supernode, for all its rows / columns
                        for (int32_t v = 0; v < w; v++) {
                            int32_t b, e;
                            outside(a + v, v, w, &b, &e);
                            if (e <= b) continue;
                            if (e - b <= SN_WHOLE) {
                                tasks.push_back({a + v, b, e, -1});
                            } else {
                                int32_t slot = part[(size_t)(a + v)];
                                for (int32_t t = b; t < e; t += SN_SEG) tasks.push_back({a + v, t, std::min(e, t + SN_SEG), slot++});
                                comb.push_back(a + v);
                            }
                        }
                    } else if (forward) {
                        // chunk q - 1 is solved: every later row takes its terms of that chunk's columns
                        const int32_t c0 = (q - 1) * chunk;
                        for (int32_t v = q * chunk; v < w; v++) {
                            const int32_t in0 = Gp_h[a + v + 1] - v;
                            tasks.push_back({a + v, in0 + c0, in0 + c0 + chunk, -1});
                        }
                    } else {
                        // chunk c = nch - q is solved: every earlier column takes its terms of that chunk's rows
                        const int32_t c = nch - q, r0 = c * chunk, r1 = std::min(w, r0 + chunk);
                        for (int32_t v = 0; v < r0; v++) {
                            const int32_t base = Lp_h[a + v] - v;
                            tasks.push_back({a + v, base + r0, base + r1, -1});
                        }
                    }
                    const int32_t chunk = vs0[(size_t)S] + (forward ? q : nch - 1 - q);
                    const int32_t cw = vs_w[(size_t)chunk];
                    (cw > 32 ? wide : cw > SN_PANEL ? medium : narrow).push_back(chunk);
                    tri.push_back(make_int4(vs_a[(size_t)chunk], cw, vs_f[(size_t)chunk], 0));
                }
                stp.tc = (int32_t)tasks.size() - stp.t0;
                for (int32_t t = stp.t0; t < stp.t0 + stp.tc; t++) stp.terms += tasks[(size_t)t].e - tasks[(size_t)t].b;
                stp.cc = (int32_t)comb.size() - stp.c0;
                stp.sc = (int32_t)narrow.size() - stp.s0;
                stp.mc = (int32_t)medium.size() - stp.m0;
                stp.bc = (int32_t)wide.size() - stp.b0;
                stp.qc = (int32_t)tri.size() - stp.q0;
                // the widest first: a launch is as long as its longest wave
                std::stable_sort(tri.begin() + stp.q0, tri.end(), [](const int4 &x, const int4 &y) { return x.y > y.y; });
                D.steps.push_back(stp);
            }
        }
        CSX_TRY(up(&D.tasks, tasks));
        CSX_TRY(up(&D.comb, comb));
        CSX_TRY(up(&D.narrow, narrow));
        CSX_TRY(up(&D.medium, medium));
        CSX_TRY(up(&D.wide, wide));
        CSX_TRY(up(&D.part_ptr, part));
        CSX_TRY(up(&D.tri4, tri));
        D.tri_h = std::move(tri);
        return CSX_OK;
    };
    if (st == CSX_OK) st = build(P->fwd, height, true);
    if (st == CSX_OK) st = build(P->bwd, depth, false);
    if (st == CSX_OK && P->bwd.nslots != bwd_slots) {
        set_error("sn_build: slot count mismatch (%d / %d)", P->bwd.nslots, bwd_slots);
        st = CSX_ERUNTIME;
    }
    // ---- matrix-core fragments of every triangle (every virtual supernode is in exactly one forward step) ----
    double growth = 0.0;
    if (st == CSX_OK && ctx().opt.tri_supernodes == 1) {
        const size_t nv = vs_a.size();
        std::vector<int4> leaf4((size_t)nleaf);
        int64_t at = vs_f.back();
        for (int32_t t = 0; t < nleaf; t++) {
            const int32_t cnt = leaf_ptr[(size_t)t + 1] - leaf_ptr[(size_t)t];
            leaf4[(size_t)t] = make_int4(leaf_ptr[(size_t)t], cnt, (int32_t)at, 0);
            at += sn_tiles((cnt + 15) / 16) * 4;
        }
        const size_t nfrag = (size_t)at;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        if (nfrag * 1024 <= free_b / 4 && at < 0x7fffffff) {
            DevScope tmp;
            unsigned long long *d_cond = nullptr, h_cond = 0;
            st = dalloc(&P->frags, 2 * nfrag * 64 + 64);
            P->frag_toff = (int64_t)nfrag * 64;
            if (st == CSX_OK) st = up(&P->leaf4, leaf4);
            if (st == CSX_OK) st = tmp.alloc(&d_cond, 1);
            if (st == CSX_OK && hipMemsetAsync(d_cond, 0, sizeof(unsigned long long), s) != hipSuccess) st = CSX_ERUNTIME;
            if (st == CSX_OK) {
                static bool frag_lds_set = false;
                if (!frag_lds_set) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_frags<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)sn_frags_lds<64>());
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_frags<true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)sn_frags_lds<64>());
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_frags<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)sn_frags_lds<128>());
                    frag_lds_set = true;
                }
                if (chunk > 64)
                    hipLaunchKernelGGL((k_sn_frags<false, 128>), dim3((unsigned)nv), dim3(64), sn_frags_lds<128>(), s, P->fwd.tri4, (int32_t)nv,
                                       L->p, L->i, L->x, (const int32_t *)nullptr, (const int32_t *)nullptr, (const double *)nullptr,
                                       (const double *)nullptr, P->frags, P->frag_toff, d_cond);
                else
                    hipLaunchKernelGGL((k_sn_frags<false, 64>), dim3((unsigned)nv), dim3(64), sn_frags_lds<64>(), s, P->fwd.tri4, (int32_t)nv,
                                       L->p, L->i, L->x, (const int32_t *)nullptr, (const int32_t *)nullptr, (const double *)nullptr,
                                       (const double *)nullptr, P->frags, P->frag_toff, d_cond);
                if (nleaf > 0)
                    hipLaunchKernelGGL((k_sn_frags<true, 64>), dim3((unsigned)nleaf), dim3(64), sn_frags_lds<64>(), s, P->leaf4, nleaf, L->p,
                                       L->i, L->x, P->lb_ptr, P->lb_idx, P->lb_val, P->ldiag, P->frags, P->frag_toff, d_cond);
                if (hipMemcpyAsync(&h_cond, d_cond, sizeof(h_cond), hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess)
                    st = CSX_ERUNTIME;
            }
            if (st == CSX_OK) {
                std::memcpy(&growth, &h_cond, sizeof(double));
                // the largest || |W_ii| |L_ii| ||_inf over all diagonal blocks (k_sn_frags): past 1e3 -- an error of
                // ~1e-13 per block -- the triangles stay with substitution (k_sn_tri); a zero pivot reports infinity
                P->growth = growth;
                P->mfma = growth <= 1e3;
                if (!P->mfma) {
                    dfree(P->frags);
                    P->frags = nullptr;
                }
            }
        }
    }
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    if (st == CSX_OK && (P->has_relaxed || P->chunk > 64) && !P->mfma) {
        // relaxed supernodes exist only as matrix-core fragments (the substitution kernel reads dense trapezoids): without
        // them -- a diagonal block past the guard, no memory for the fragments -- the level-scheduled plans keep the factor
        if (say) std::fprintf(stderr, "sn_build: relaxed supernodes but no matrix-core triangles (%.3g): no supernodal plan\n", growth);
        free_snplan(P);
        return CSX_OK;
    }
    if (st != CSX_OK) {
        free_snplan(P);
        return st;
    }
    if (say)
        std::fprintf(stderr, "sn_build: %d forward steps, %d backward steps, triangles %s (largest || |W_ii| |L_ii| || = %.3g)\n", (int)P->fwd.steps.size(),
                     (int)P->bwd.steps.size(), P->mfma ? "on the matrix cores" : "by substitution", growth);
    *out = P;
    return CSX_OK;
}

/* Chunks (and relaxed runs) of 128 columns when the triangles can go to the matrix cores -- half the steps of a chain, a
 * step costing half as much again -- else of 64, the widest the substitution kernel stages in LDS. */
int sn_build(const Csc *L, const int32_t *parent, const int32_t *Lp_h, const int32_t *Gp_h, const int32_t *Gp, const int32_t *Gi,
             const double *Gx, int32_t col_levels, SnPlan **out) {
    if (ctx().opt.tri_supernodes == 1) {
        const int st = sn_build_with(L, parent, Lp_h, Gp_h, Gp, Gi, Gx, col_levels, 128, out);
        if (st == CSX_OK && *out) return CSX_OK;
        if (st != CSX_OK) {                 // (its fragment builder wants 150 KB of LDS: should a device refuse, chunks of 64 remain)
            (void)hipGetLastError();
            *out = nullptr;
        }
    }
    return sn_build_with(L, parent, Lp_h, Gp_h, Gp, Gi, Gx, col_levels, SN_CHUNK, out);
}

void sn_info(const SnPlan *P, int32_t *nsn, int32_t *levels, int32_t *max_w) {
    if (nsn) *nsn = P->nsn;
    if (levels) *levels = (int32_t)P->fwd.steps.size();
    if (max_w) *max_w = P->max_w;
}

bool sn_usable(const SnPlan *P) { return !(P->has_relaxed || P->chunk > 64) || (P->mfma && ctx().opt.tri_supernodes == 1); }

void sn_info2(const SnPlan *P, int32_t *matrix_cores, double *growth) {
    if (matrix_cores) *matrix_cores = P->mfma && ctx().opt.tri_supernodes == 1 ? 1 : 0;
    if (growth) *growth = P->growth;
}

/* X (n x nrhs, row-major) <- inv(L) X (forward) or inv(L') X.  G*: the forward plan's row-major copy of L (off-diagonal
 * terms of every row in ascending column order, diagonal apart); L: the factor. */
int64_t sn_generation(const SnPlan *P) { return P->generation; }

int sn_prepare(SnPlan *P, int32_t nrhs) {
    const int64_t need = (int64_t)std::max(P->fwd.nslots, P->bwd.nslots + P->nleafslots) * nrhs;
    if (P->partial_len < need) {
        dfree(P->partial);
        P->partial = nullptr;
        P->partial_len = 0;
        CSX_TRY(dalloc(&P->partial, (size_t)need));
        P->partial_len = need;
        P->generation++;
    }
    static bool lds_set = false;
    if (!lds_set) {     // the 64-column triangle + its rows of X: 66 KB of LDS, past the static limit
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_tri<4, SN_CHUNK, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sn_tri_lds<SN_CHUNK>()));
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_tri<4, SN_CHUNK, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sn_tri_lds<SN_CHUNK>()));
        // the fragments of a 128-column triangle: 72 KB
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_mfma<true, false, 1, 8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sn_tiles(8) * 256 * sizeof(double))));
        CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_mfma<false, false, 1, 8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sn_tiles(8) * 256 * sizeof(double))));
        lds_set = true;
    }
    return CSX_OK;
}

int sn_solve(SnPlan *P, bool forward, const int32_t *Gp, const int32_t *Gi, const double *Gx, const double *Gd, const Csc *L,
             double *X, int32_t nrhs) {
    hipStream_t s = ctx().stream;
    const SnDir &D = forward ? P->fwd : P->bwd;
    CSX_TRY(sn_prepare(P, nrhs));
    const int nblk = (nrhs + 63) / 64;
    const int32_t *idx = forward ? Gi : L->i;
    const double *val = forward ? Gx : L->x;
    const bool cores = P->mfma && ctx().opt.tri_supernodes == 1;
    // a launch of many tasks: a wave per (task, 64 right-hand sides), or R lanes per task for up to 32 right-hand sides
    auto many_tasks = [&](const SnTask *tasks, int32_t t0, int32_t tc, const int32_t *ti, const double *tv, bool short_tasks) {
        // (lanes per task pay when the tasks are many and SHORT: a lane walks its task one dependent gather after the other --
        // 22 528 tasks of up to 512 terms took 100 us that way against 35 - 60 with a wave each)
        // (and when there are so many that a wave each is bound by the rate at which waves start, whatever their length)
        if (nrhs <= 32 && tc >= 8192 && (short_tasks || tc >= 50000)) {
            int R = 1;
            while (R < nrhs) R *= 2;
            const unsigned grid = (unsigned)(((int64_t)tc * R + 255) / 256);
            switch (R) {
                case 1: hipLaunchKernelGGL(k_sn_outside_thin<1>, dim3(grid), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs); break;
                case 2: hipLaunchKernelGGL(k_sn_outside_thin<2>, dim3(grid), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs); break;
                case 4: hipLaunchKernelGGL(k_sn_outside_thin<4>, dim3(grid), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs); break;
                case 8: hipLaunchKernelGGL(k_sn_outside_thin<8>, dim3(grid), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs); break;
                case 16: hipLaunchKernelGGL(k_sn_outside_thin<16>, dim3(grid), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs); break;
                default: hipLaunchKernelGGL(k_sn_outside_thin<32>, dim3(grid), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs); break;
            }
        } else {
            const int64_t waves = (int64_t)tc * nblk;
            hipLaunchKernelGGL(k_sn_outside, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, tasks, t0, tc, ti, tv, X, P->partial, nrhs);
        }
    };
    // `count` triangles of a list from `first` on: few of them -> a wave per column tile of 16 right-hand sides (the dependent
    // chain of a step is what counts), many -> a wave per 64 (fewer workgroups, every fragment read once)
    auto triangles = [&](auto fwd_tag, auto gather_tag, const int4 *list, const std::vector<int4> *host, int32_t first, int32_t count,
                         const int32_t *rows) {
        constexpr bool F = decltype(fwd_tag)::value, G = decltype(gather_tag)::value;
        const int4 ent0 = host && count == 1 ? (*host)[(size_t)first] : make_int4(0, 0, 0, 0);
        const int use0 = host && count == 1 ? 1 : 0;
        const unsigned grid = (unsigned)((int64_t)count * nblk);
        const int tiles = std::min(4, (nrhs + 15) / 16);
        constexpr size_t lds4 = (size_t)sn_tiles(4) * 256 * sizeof(double), lds8 = (size_t)sn_tiles(8) * 256 * sizeof(double);
        if (!G && P->chunk > 64) {          // triangles of up to 128 columns: eight block rows of X, one column tile per wave
            hipLaunchKernelGGL((k_sn_mfma<F, false, 1, 8>), dim3(grid), dim3(64 * tiles), lds8, s, list, first, ent0, use0, rows, P->frags,
                               P->frag_toff, X, nrhs);
        } else if ((int64_t)count * nblk <= 512) {
            hipLaunchKernelGGL((k_sn_mfma<F, G, 1, 4>), dim3(grid), dim3(64 * tiles), lds4, s, list, first, ent0, use0, rows, P->frags, P->frag_toff,
                               X, nrhs);
        } else {
            hipLaunchKernelGGL((k_sn_mfma<F, G, 4, 4>), dim3(grid), dim3(64), lds4, s, list, first, ent0, use0, rows, P->frags, P->frag_toff, X, nrhs);
        }
    };
    if (forward && P->nleaf && cores)
        triangles(std::true_type{}, std::true_type{}, P->leaf4, nullptr, 0, P->nleaf, P->leaf_cols);
    else if (forward && P->nleaf)
        hipLaunchKernelGGL(k_sn_leaf<true>, dim3((unsigned)(P->nleaf * nblk)), dim3(64), 0, s, P->leaf_ptr, P->leaf_cols, P->lf_ptr,
                           P->lf_idx, P->lf_val, P->ldiag, (const int32_t *)nullptr, (const double *)nullptr, X, nrhs);
    for (const SnStep &t : D.steps) {
        if (t.tc > 0 && t.tc <= 16384) {      // (by the step's size alone: the same sums for any number of right-hand sides)
            hipLaunchKernelGGL(k_sn_outside_wg, dim3((unsigned)((int64_t)t.tc * nblk)), dim3(256), 0, s, D.tasks, t.t0, t.tc, idx, val, X,
                               P->partial, nrhs);
        } else if (t.tc > 0) {
            many_tasks(D.tasks, t.t0, t.tc, idx, val, t.terms <= (int64_t)48 * t.tc);
        }
        if (t.cc > 0) {
            const int64_t waves = (int64_t)t.cc * nblk;
            hipLaunchKernelGGL(k_sn_combine, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, D.comb, t.c0, t.cc, D.part_ptr, 0,
                               P->partial, X, nrhs);
        }
        if (cores) {
            if (t.qc > 0) {
                if (forward) triangles(std::true_type{}, std::false_type{}, D.tri4, &D.tri_h, t.q0, t.qc, nullptr);
                else triangles(std::false_type{}, std::false_type{}, D.tri4, &D.tri_h, t.q0, t.qc, nullptr);
            }
        } else if (forward) {
            if (t.sc > 0)
                hipLaunchKernelGGL((k_sn_tri<1, SN_PANEL, true>), dim3((unsigned)(t.sc * nblk)), dim3(64), sn_tri_lds<SN_PANEL>(), s, D.narrow,
                                   t.s0, P->vs_a, P->vs_w, Gp, Gx, Gd, (const int32_t *)nullptr, P->partial, X, nrhs);
            if (t.mc > 0)
                hipLaunchKernelGGL((k_sn_tri<2, 32, true>), dim3((unsigned)(t.mc * nblk)), dim3(128), sn_tri_lds<32>(), s, D.medium, t.m0,
                                   P->vs_a, P->vs_w, Gp, Gx, Gd, (const int32_t *)nullptr, P->partial, X, nrhs);
            if (t.bc > 0)
                hipLaunchKernelGGL((k_sn_tri<4, SN_CHUNK, true>), dim3((unsigned)(t.bc * nblk)), dim3(256), sn_tri_lds<SN_CHUNK>(), s, D.wide,
                                   t.b0, P->vs_a, P->vs_w, Gp, Gx, Gd, (const int32_t *)nullptr, P->partial, X, nrhs);
        } else {
            if (t.sc > 0)
                hipLaunchKernelGGL((k_sn_tri<1, SN_PANEL, false>), dim3((unsigned)(t.sc * nblk)), dim3(64), sn_tri_lds<SN_PANEL>(), s, D.narrow,
                                   t.s0, P->vs_a, P->vs_w, L->p, L->x, (const double *)nullptr, (const int32_t *)nullptr, P->partial, X, nrhs);
            if (t.mc > 0)
                hipLaunchKernelGGL((k_sn_tri<2, 32, false>), dim3((unsigned)(t.mc * nblk)), dim3(128), sn_tri_lds<32>(), s, D.medium, t.m0,
                                   P->vs_a, P->vs_w, L->p, L->x, (const double *)nullptr, (const int32_t *)nullptr, P->partial, X, nrhs);
            if (t.bc > 0)
                hipLaunchKernelGGL((k_sn_tri<4, SN_CHUNK, false>), dim3((unsigned)(t.bc * nblk)), dim3(256), sn_tri_lds<SN_CHUNK>(), s, D.wide,
                                   t.b0, P->vs_a, P->vs_w, L->p, L->x, (const double *)nullptr, (const int32_t *)nullptr, P->partial, X, nrhs);
        }
    }
    if (!forward && P->nleaf) {
        const int64_t waves = (int64_t)P->nleaftasks * nblk;    // every leaf column's rows outside its subtree (ancestors: final)
        if (waves > 0) many_tasks(P->leaf_tasks, 0, P->nleaftasks, L->i, L->x, true);   // a leaf column's few rows above its subtree
        if (P->nleafslots > 0) {                                // the leaf columns that were cut into pieces
            const int64_t cw = (int64_t)P->nleafcols * nblk;
            hipLaunchKernelGGL(k_sn_combine, dim3((unsigned)((cw + 3) / 4)), dim3(256), 0, s, P->leaf_cols, 0, P->nleafcols, P->lslot_ptr, 1,
                               P->partial + (int64_t)P->bwd.nslots * nrhs, X, nrhs);
        }
        if (cores)
            triangles(std::false_type{}, std::true_type{}, P->leaf4, nullptr, 0, P->nleaf, P->leaf_cols);
        else
            hipLaunchKernelGGL(k_sn_leaf<false>, dim3((unsigned)(P->nleaf * nblk)), dim3(64), 0, s, P->leaf_ptr, P->leaf_cols, P->lb_ptr,
                               P->lb_idx, P->lb_val, P->ldiag, (const int32_t *)nullptr, (const double *)nullptr, X, nrhs);
    }
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // namespace csx
