// cs_chol for chain-like factors with a WIDE band (reference: csparse.py:561-619, the up-looking row solve).
//
// A 2-D grid Laplacian in natural order (the only order the reference's own cs_cholsol(0, ...) offers) has an
// elimination tree that is one chain and a factor that is dense inside a band of half-width g: a 700 x 700 grid is
// n = 490 000 columns of 701 entries (lnz 3.4e8), every column its own level, 700 updates of 700 terms each.  The
// column kernels of csx_chol.hip spend 350 us on such a column (171 s for the matrix); the register-window kernel
// (k_chol_band) holds the live triangle of half-widths up to 176 in one workgroup's registers and stops there.
//
// Here the band is worked on in a DENSE BAND ARRAY in memory -- element (r, c), c <= r <= c + bw, at
// W[c * (bw + 1) + r - c], 2.7 GB for the grid above, the live (bw + 1)^2 / 2 window of it (2 MB) staying in L2 --
// by a right-looking BLOCKED factorisation, NB columns per step, two launches per step:
//   k_wband_panel   the NB x NB diagonal block is factored by one wave (lane = row, the block's row in registers,
//                   L(c, j) handed round by v_readlane), every workgroup doing it for itself; then each thread
//                   takes one of the <= bw rows below it and solves its NB entries against the block;
//   k_wband_update  64 x 64 tiles of the live window (one workgroup each, 4 x 4 elements per thread) subtract the
//                   panel's NB products l(r, k) l(c, k) from their elements, the two l strips staged in LDS.
// Every element receives its products in ascending column order k, multiply and subtract rounded separately, then
// one division by the pivot: on a chain tree that is the reference's operation sequence (cs_ereach hands the row
// solve its columns in ascending order), so L.x is bit-identical to it; structural zeros inside the band stay +0.0
// and change nothing.  On other trees the result agrees to rounding, like the general column kernels.
#include "csx_internal.h"

namespace csx {

#pragma clang fp contract(off)

// one wave per column: dense band array <- L.x (scatter = true) or L.x <- dense band array.  On the way back the entries
// of a column inside its own NB x NB diagonal block come from `dfac` (panel-major, [panel][column of the block][row of
// the block]): the factored blocks are NOT stored into the band array while the factorisation runs -- every workgroup of
// a panel's launch reads the block as the earlier launches left it and factors it for itself, so the one that stores the
// result must not store it where the others read.
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_wband_copy(int32_t n, int32_t ld, const int32_t *__restrict__ Lp,
                                                    const int32_t *__restrict__ Li, double *Lx, double *W,
                                                    const double *__restrict__ dfac, int nb) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int32_t p0 = (int32_t)(j / nb) * nb;            // first column / row of j's diagonal block
    for (int32_t q = Lp[j] + lane; q < Lp[j + 1]; q += 64) {
        const int32_t r = Li[q];
        const int64_t at = j * ld + (r - (int32_t)j);
        if (SCATTER) W[at] = Lx[q];
        else Lx[q] = r < p0 + nb ? dfac[(int64_t)(j / nb) * nb * nb + (int64_t)(j - p0) * nb + (r - p0)] : W[at];
    }
}

constexpr int WB_PANEL_ROWS = 256;      // rows below the diagonal block per workgroup of k_wband_panel
constexpr int WB_TILE = 64;             // k_wband_update: tile side

// Panel [c0, c0 + NB): wave 0 of every workgroup factors the diagonal block (redundantly; workgroup 0 stores it),
// waves 1 .. 4 of workgroup g solve rows c0 + NB + 256 g ... against it.  Element (r, c0 + t) of the panel sits at
// Wp[t * ld + (r - c0 - t)], Wp = W + c0 * ld: 32-bit offsets from one uniform base.
template <int NB>
__global__ __launch_bounds__(64 + WB_PANEL_ROWS) void k_wband_panel(int32_t n, int32_t bw, int32_t c0, double *W,
                                                                   double *__restrict__ dfac, int *notspd) {
    __shared__ double D[NB * NB];       // the factored block: D[s * NB + t] = L(c0 + t, c0 + s), s <= t
    const uint32_t ld = (uint32_t)bw + 1;
    double *Wp = W + (int64_t)c0 * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t r = c0 + NB + (int32_t)blockIdx.x * WB_PANEL_ROWS + (tid - 64);   // this thread's row below the block
    const int32_t rmax = min(n - 1, c0 + NB - 1 + bw);
    const bool mine = wave > 0 && r <= rmax;
    const uint32_t ro = (uint32_t)(r - c0);
    double x[NB];
    if (wave == 0) {
        // ---- diagonal block, one wave, lane = row i of the block, a[c] = element (i, c) ----
        const bool row = lane < NB && c0 + lane < n;
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = (row && c <= lane && lane - c <= bw) ? Wp[(uint32_t)c * ld + (uint32_t)(lane - c)] : 0.0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            // columns past the end of the matrix: an identity block (pivot 1, nothing below)
            const bool live = c0 + j < n;
            const int dlo = __builtin_amdgcn_readlane(__double2loint(a[j]), j);
            const int dhi = __builtin_amdgcn_readlane(__double2hiint(a[j]), j);
            const double d = live ? __hiloint2double(dhi, dlo) : 1.0;
            if (live && !(d > 0.0) && lane == 0 && blockIdx.x == 0) atomicMin(notspd, c0 + j);   // csparse.py:612
            const double ljj = sqrt(d);
            double lij = 0.0;
            if (row && lane > j) lij = a[j] / ljj;
            if (lane == j) a[j] = ljj;
            else a[j] = lij;
#pragma unroll
            for (int c = j + 1; c < NB; c++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(lij), c);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(lij), c);
                const double lcj = __hiloint2double(hi, lo);
                const double t = lij * lcj;
                if (lane >= c) a[c] = a[c] - t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB; c++) {
                D[c * NB + lane] = a[c];
                if (blockIdx.x == 0 && row && c <= lane) dfac[(int64_t)(c0 / NB) * NB * NB + c * NB + lane] = a[c];   // see k_wband_copy
            }
        }
    } else {
        // the row's entries in the panel's columns (outside the band or the matrix: 0), in flight while wave 0 works
#pragma unroll
        for (int t = 0; t < NB; t++)
            x[t] = (mine && c0 + t < n && ro - (uint32_t)t <= (uint32_t)bw) ? Wp[(uint32_t)t * ld + (ro - (uint32_t)t)] : 0.0;
    }
    __syncthreads();
    if (!mine) return;
    // ---- this row against the block: x[t] = (x[t] - sum_{s < t} x[s] L(c0 + t, c0 + s)) / L(c0 + t, c0 + t) ----
#pragma unroll
    for (int t = 0; t < NB; t++) {
        double v = x[t];
#pragma unroll
        for (int s = 0; s < t; s++) {
            const double p = x[s] * D[s * NB + t];
            v = v - p;
        }
        x[t] = v / D[t * NB + t];
        asm volatile("" : "+v"(x[t]) : : "memory");   // x[t] is formed HERE: the block's values of one row, not of all rows, in registers
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NB; t++)
        if (c0 + t < n && ro - (uint32_t)t <= (uint32_t)bw) Wp[(uint32_t)t * ld + (ro - (uint32_t)t)] = x[t];
}

// Trailing update by panel [c0, c0 + NB): element (r, c), c0 + NB <= c <= r <= min(n - 1, c0 + NB - 1 + bw),
// loses sum_k l(r, k) l(c, k), k ascending.  Tile (blockIdx.x = row tile, blockIdx.y = column tile), 256 threads,
// 4 x 4 elements per thread.
template <int NB>
__global__ __launch_bounds__(256) void k_wband_update(int32_t n, int32_t bw, int32_t c0, double *W) {
    if (blockIdx.y > blockIdx.x) return;
    __shared__ __attribute__((aligned(16))) double lr[NB][WB_TILE], lc[NB][WB_TILE];
    const int64_t ld = (int64_t)bw + 1;
    const int32_t first = c0 + NB;
    const int32_t rmax = min(n - 1, c0 + NB - 1 + bw);
    const int32_t R0 = first + (int32_t)blockIdx.x * WB_TILE, C0 = first + (int32_t)blockIdx.y * WB_TILE;
    const int tid = threadIdx.x;
    // the two strips: l(R0 + i, c0 + t) and l(C0 + i, c0 + t), zero outside the band / past rmax
    for (int e = tid; e < NB * WB_TILE; e += 256) {
        const int t = e / WB_TILE, i = e % WB_TILE;
        const int32_t k = c0 + t;
        const int32_t ra = R0 + i, rb = C0 + i;
        lr[t][i] = (ra <= rmax && ra - k <= bw) ? W[k * ld + (ra - k)] : 0.0;
        lc[t][i] = (rb <= rmax && rb - k <= bw) ? W[k * ld + (rb - k)] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    const int32_t r0 = R0 + 4 * tx, cc0 = C0 + 4 * ty;
    if (r0 > rmax || cc0 > rmax || r0 + 3 < cc0) return;     // nothing of this thread's 4 x 4 is on or below the diagonal
    double acc[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int32_t rr = r0 + a, c = cc0 + b;
            acc[b][a] = (rr <= rmax && c <= rr) ? W[c * ld + (rr - c)] : 0.0;
        }
#pragma unroll 4
    for (int t = 0; t < NB; t++) {
        double ra[4], cb[4];
#pragma unroll
        for (int a = 0; a < 4; a++) ra[a] = lr[t][4 * tx + a];
#pragma unroll
        for (int b = 0; b < 4; b++) cb[b] = lc[t][4 * ty + b];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const double p = ra[a] * cb[b];
                acc[b][a] = acc[b][a] - p;
            }
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int32_t rr = r0 + a, c = cc0 + b;
            if (rr <= rmax && c <= rr) W[c * ld + (rr - c)] = acc[b][a];
        }
}

// ---- one launch per panel: panel k + 1 is factored WHILE panel k's update of the rest of the window runs ----------
// Launch for panel [c1, c1 + NB), previous panel [c0, c0 + NB) = [c1 - NB, c1) (has_prev):
//   role A (the first gA workgroups, 64 + 256 threads): the panel's own elements first lose the previous panel's
//          products (the part of its trailing update that falls on these NB columns), then the diagonal block is
//          factored by wave 0 and every row below is solved against it, as in k_wband_panel;
//   role B (the other workgroups): 64 x 64 tiles of the window BEYOND the panel (columns >= c1 + NB) lose the previous
//          panel's products, as in k_wband_update.
// Both read the previous panel's columns (final since the launch before) and write disjoint elements, so the two
// launches per panel become one and the serial part (block factor) hides behind the wide one.  Per-element order of
// operations is unchanged: earlier columns first, ascending.
template <int NB>
__global__ __launch_bounds__(64 + WB_PANEL_ROWS) void k_wband_step(int32_t n, int32_t bw, int32_t c1, int has_prev,
                                                                  int32_t gA, int32_t ntB, double *W,
                                                                  double *__restrict__ dfac, int *notspd) {
    __shared__ __attribute__((aligned(16))) double smem[2 * NB * WB_TILE > 3 * NB * NB ? 2 * NB * WB_TILE : 3 * NB * NB];
    const uint32_t ld = (uint32_t)bw + 1;
    const int32_t c0 = c1 - NB;
    const int tid = threadIdx.x;
    if ((int32_t)blockIdx.x >= gA) {
        // ---- role B: tile (bi, bj), bj <= bi, of the window beyond the panel ----
        if (tid >= 256) return;
        const int32_t tb = (int32_t)blockIdx.x - gA;
        const int32_t bi = tb / ntB, bj = tb % ntB;
        if (bj > bi) return;
        double (*lr)[WB_TILE] = reinterpret_cast<double (*)[WB_TILE]>(smem);
        double (*lc)[WB_TILE] = reinterpret_cast<double (*)[WB_TILE]>(smem + NB * WB_TILE);
        const int64_t ldl = ld;
        const int32_t first = c1 + NB;
        const int32_t rmax = min(n - 1, c0 + NB - 1 + bw);
        const int32_t R0 = first + bi * WB_TILE, C0 = first + bj * WB_TILE;
        const int tx = tid & 15, ty = tid >> 4;
        const int32_t r0 = R0 + 4 * tx, cc0 = C0 + 4 * ty;
        const bool any = !(r0 > rmax || cc0 > rmax || r0 + 3 < cc0);   // something of this thread's 4 x 4 on / below the diagonal
        double acc[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int32_t rr = r0 + a, c = cc0 + b;
                acc[b][a] = (any && rr <= rmax && c <= rr) ? W[c * ldl + (rr - c)] : 0.0;
            }
        for (int e = tid; e < NB * WB_TILE; e += 256) {
            const int t = e / WB_TILE, i = e % WB_TILE;
            const int32_t k = c0 + t;
            const int32_t ra = R0 + i, rb = C0 + i;
            lr[t][i] = (ra <= rmax && ra - k <= bw) ? W[k * ldl + (ra - k)] : 0.0;
            lc[t][i] = (rb <= rmax && rb - k <= bw) ? W[k * ldl + (rb - k)] : 0.0;
        }
        __syncthreads();
        if (!any) return;
#pragma unroll 4
        for (int t = 0; t < NB; t++) {
            double ra[4], cb[4];
#pragma unroll
            for (int a = 0; a < 4; a++) ra[a] = lr[t][4 * tx + a];
#pragma unroll
            for (int b = 0; b < 4; b++) cb[b] = lc[t][4 * ty + b];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const double p = ra[a] * cb[b];
                    acc[b][a] = acc[b][a] - p;
                }
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int32_t rr = r0 + a, c = cc0 + b;
                if (rr <= rmax && c <= rr) W[c * ldl + (rr - c)] = acc[b][a];
            }
        return;
    }
    // ---- role A ----
    double *E = smem;                 // E[t * NB + i] = l(c1 + i, c0 + t): the panel's diagonal rows in the previous panel's columns
    double *Dt = smem + NB * NB;      // Dt[c * NB + i]: the diagonal block after the previous panel's update
    double *D = smem + 2 * NB * NB;   // D[s * NB + t] = L(c1 + t, c1 + s): the factored block
    double *Wp = W + (int64_t)c1 * ld;                       // element (r, c1 + t) at Wp[t * ld + (r - c1 - t)]
    double *Wq = W + (int64_t)(has_prev ? c0 : c1) * ld;     // element (r, c0 + t) at Wq[t * ld + (r - c0 - t)]
    const int lane = tid & 63, wave = tid >> 6;
    const int32_t r = c1 + NB + (int32_t)blockIdx.x * WB_PANEL_ROWS + (tid - 64);   // this thread's row below the block
    const int32_t rmax = min(n - 1, c1 + NB - 1 + bw);
    const bool mine = wave > 0 && r <= rmax;
    const uint32_t ro = (uint32_t)(r - c1), rq = (uint32_t)(r - c0);
    double x[NB], lrow[NB];
    constexpr int THREADS = 64 + WB_PANEL_ROWS;
    for (int e = tid; e < NB * NB; e += THREADS) {
        const int ec = e / NB, ei = e % NB;
        // E: t = ec, i = ei: row c1 + ei, column c0 + ec, distance NB + ei - ec
        const int32_t dist = NB + ei - ec;
        E[e] = (has_prev && c1 + ei < n && dist <= bw) ? Wq[(uint32_t)ec * ld + (uint32_t)dist] : 0.0;
        // Dt: element (c1 + ei, c1 + ec) of the block as stored (zero above the diagonal / outside the band or matrix)
        Dt[e] = (ei >= ec && c1 + ei < n && ei - ec <= bw) ? Wp[(uint32_t)ec * ld + (uint32_t)(ei - ec)] : 0.0;
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            x[t] = (mine && c1 + t < n && ro - (uint32_t)t <= (uint32_t)bw) ? Wp[(uint32_t)t * ld + (ro - (uint32_t)t)] : 0.0;
            lrow[t] = (mine && has_prev && rq - (uint32_t)t <= (uint32_t)bw) ? Wq[(uint32_t)t * ld + (rq - (uint32_t)t)] : 0.0;
        }
    }
    __syncthreads();
    // the previous panel's products, ascending t
    if (has_prev) {
        for (int e = tid; e < NB * NB; e += THREADS) {       // the thread that stored Dt[e] updates it
            const int ec = e / NB, ei = e % NB;
            if (ei < ec) continue;
            double v = Dt[e];
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const double p = E[t * NB + ei] * E[t * NB + ec];
                v = v - p;
            }
            Dt[e] = v;
        }
    }
    if (mine && has_prev) {
#pragma unroll
        for (int tq = 0; tq < NB; tq++) {
            double v = x[tq];
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const double p = lrow[t] * E[t * NB + tq];
                v = v - p;
            }
            x[tq] = v;
            asm volatile("" : "+v"(x[tq]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // ---- diagonal block, one wave, lane = row i of the block, a[c] = element (i, c) ----
        const bool row = lane < NB && c1 + lane < n;
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = (lane < NB && c <= lane) ? Dt[c * NB + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const bool live = c1 + j < n;                     // past the end of the matrix: an identity block
            const int dlo = __builtin_amdgcn_readlane(__double2loint(a[j]), j);
            const int dhi = __builtin_amdgcn_readlane(__double2hiint(a[j]), j);
            const double d = live ? __hiloint2double(dhi, dlo) : 1.0;
            if (live && !(d > 0.0) && lane == 0 && blockIdx.x == 0) atomicMin(notspd, c1 + j);   // csparse.py:612
            const double ljj = sqrt(d);
            double lij = 0.0;
            if (row && lane > j) lij = a[j] / ljj;
            if (lane == j) a[j] = ljj;
            else a[j] = lij;
#pragma unroll
            for (int c = j + 1; c < NB; c++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(lij), c);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(lij), c);
                const double lcj = __hiloint2double(hi, lo);
                const double t = lij * lcj;
                if (lane >= c) a[c] = a[c] - t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB; c++) {
                D[c * NB + lane] = a[c];
                if (blockIdx.x == 0 && row && c <= lane) dfac[(int64_t)(c1 / NB) * NB * NB + c * NB + lane] = a[c];   // see k_wband_copy
            }
        }
    }
    __syncthreads();
    if (!mine) return;
#pragma unroll
    for (int t = 0; t < NB; t++) {
        double v = x[t];
#pragma unroll
        for (int s = 0; s < t; s++) {
            const double p = x[s] * D[s * NB + t];
            v = v - p;
        }
        x[t] = v / D[t * NB + t];
        asm volatile("" : "+v"(x[t]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NB; t++)
        if (c1 + t < n && ro - (uint32_t)t <= (uint32_t)bw) Wp[(uint32_t)t * ld + (ro - (uint32_t)t)] = x[t];
}

// ---- dense fronts of a bushy tree: fundamental supernodes factored in place ------------------------------------------
// A separator of a nested-dissection ordering is a FUNDAMENTAL SUPERNODE of L: w consecutive columns a .. a + w - 1, each
// the only child of the next, column a + t holding rows a + t .. a + w - 1 followed by the SAME r rows below the block.
// Its part of L.x is therefore a dense trapezoid stored column after column, element (v, t) -- v = t .. w + r - 1 a
// "virtual" row: the triangle's rows, then the r common rows -- at Lx[Lp[a + t] + v - t].  Every update from outside the
// supernode comes from below its FIRST column in the tree, so once those are in (k_chol_coop, mode 1, all w columns in one
// launch) the rest is a dense right-looking factorisation of the trapezoid: the blocked scheme of k_wband_step with
// nothing clipped by a band, run where the entries lie.  Supernodes of one tree level are independent: blockIdx.y.
struct SnDesc {
    int32_t a, w, r;     // first column, columns, rows below the triangular block
};

template <int NB>
__global__ __launch_bounds__(64 + WB_PANEL_ROWS) void k_sn_step(const SnDesc *__restrict__ sns, int32_t c1, int32_t gA,
                                                               int32_t ntR, int32_t ntC, const int32_t *__restrict__ Lp,
                                                               double *Lx, double *__restrict__ dfac, int32_t dstride,
                                                               int *notspd) {
    __shared__ __attribute__((aligned(16))) double smem[2 * NB * WB_TILE > 3 * NB * NB ? 2 * NB * WB_TILE : 3 * NB * NB];
    const SnDesc sn = sns[blockIdx.y];
    const int32_t w = sn.w, nrows = sn.w + sn.r;
    if (c1 >= w) return;                                  // this supernode is finished
    const int32_t c0 = c1 - NB;
    const bool has_prev = c1 > 0;
    const int tid = threadIdx.x;
    auto colp = [&](int32_t c) { return Lx + Lp[sn.a + c] - c; };   // element (v, c) at colp(c)[v]
    if ((int32_t)blockIdx.x >= gA) {
        // ---- role B: tile (bi over rows, bj over columns) of what lies beyond the panel ----
        if (tid >= 256 || !has_prev) return;
        const int32_t tb = (int32_t)blockIdx.x - gA;
        if (tb >= ntR * ntC) return;
        const int32_t bi = tb / ntC, bj = tb % ntC;
        const int32_t first = c1 + NB;
        const int32_t R0 = first + bi * WB_TILE, C0 = first + bj * WB_TILE;
        if (C0 >= w || R0 >= nrows || R0 + WB_TILE - 1 < C0) return;   // no column / no row / wholly above the diagonal
        double (*lr)[WB_TILE] = reinterpret_cast<double (*)[WB_TILE]>(smem);
        double (*lc)[WB_TILE] = reinterpret_cast<double (*)[WB_TILE]>(smem + NB * WB_TILE);
        const int tx = tid & 15, ty = tid >> 4;
        const int32_t r0 = R0 + 4 * tx, cc0 = C0 + 4 * ty;
        const bool any = r0 < nrows && cc0 < w && r0 + 3 >= cc0;
        double acc[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int32_t rr = r0 + a, c = cc0 + b;
                acc[b][a] = (any && rr < nrows && c < w && c <= rr) ? colp(c)[rr] : 0.0;
            }
        for (int e = tid; e < NB * WB_TILE; e += 256) {
            const int t = e / WB_TILE, i = e % WB_TILE;
            const double *cp = colp(c0 + t);
            const int32_t ra = R0 + i, rb = C0 + i;
            lr[t][i] = ra < nrows ? cp[ra] : 0.0;
            lc[t][i] = rb < w ? cp[rb] : 0.0;             // rb >= first > c0 + t: below the diagonal of column c0 + t
        }
        __syncthreads();
        if (!any) return;
#pragma unroll 4
        for (int t = 0; t < NB; t++) {
            double ra[4], cb[4];
#pragma unroll
            for (int a = 0; a < 4; a++) ra[a] = lr[t][4 * tx + a];
#pragma unroll
            for (int b = 0; b < 4; b++) cb[b] = lc[t][4 * ty + b];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const double p = ra[a] * cb[b];
                    acc[b][a] = acc[b][a] - p;
                }
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int32_t rr = r0 + a, c = cc0 + b;
                if (rr < nrows && c < w && c <= rr) colp(c)[rr] = acc[b][a];
            }
        return;
    }
    // ---- role A: the panel [c1, c1 + NB) of the supernode ----
    double *E = smem;                 // E[t * NB + i] = l(c1 + i, c0 + t)
    double *Dt = smem + NB * NB;      // Dt[c * NB + i]: the diagonal block after the previous panel's update
    double *D = smem + 2 * NB * NB;   // D[s * NB + t] = L(c1 + t, c1 + s)
    const int lane = tid & 63, wave = tid >> 6;
    const int32_t v = c1 + NB + (int32_t)blockIdx.x * WB_PANEL_ROWS + (tid - 64);   // this thread's (virtual) row below the block
    const bool mine = wave > 0 && v < nrows;
    if ((int32_t)blockIdx.x * WB_PANEL_ROWS >= nrows - (c1 + NB) && blockIdx.x > 0) return;   // no rows for this workgroup
    double x[NB], lrow[NB];
    constexpr int THREADS = 64 + WB_PANEL_ROWS;
    for (int e = tid; e < NB * NB; e += THREADS) {
        const int ec = e / NB, ei = e % NB;
        // block rows c1 + ei exist as virtual rows while c1 + ei < nrows; block COLUMNS only while c1 + ec < w
        E[e] = (has_prev && c1 + ei < nrows) ? colp(c0 + ec)[c1 + ei] : 0.0;
        Dt[e] = (ei >= ec && c1 + ec < w && c1 + ei < nrows) ? colp(c1 + ec)[c1 + ei] : 0.0;
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            x[t] = (mine && c1 + t < w) ? colp(c1 + t)[v] : 0.0;
            lrow[t] = (mine && has_prev) ? colp(c0 + t)[v] : 0.0;
        }
    }
    __syncthreads();
    if (has_prev) {
        for (int e = tid; e < NB * NB; e += THREADS) {
            const int ec = e / NB, ei = e % NB;
            if (ei < ec) continue;
            double vv = Dt[e];
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const double p = E[t * NB + ei] * E[t * NB + ec];
                vv = vv - p;
            }
            Dt[e] = vv;
        }
    }
    if (mine && has_prev) {
#pragma unroll
        for (int tq = 0; tq < NB; tq++) {
            double vv = x[tq];
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const double p = lrow[t] * E[t * NB + tq];
                vv = vv - p;
            }
            x[tq] = vv;
            asm volatile("" : "+v"(x[tq]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // the block's COLUMNS past the supernode's width do not exist: identity there (pivot 1, nothing below); its ROWS
        // past w are rows of the rectangular part: they take part as rows of the block (entries in the existing columns)
        const bool row = lane < NB && c1 + lane < nrows;
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = (row && c <= lane) ? Dt[c * NB + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const bool live = c1 + j < w;
            const int dlo = __builtin_amdgcn_readlane(__double2loint(a[j]), j);
            const int dhi = __builtin_amdgcn_readlane(__double2hiint(a[j]), j);
            const double d = live ? __hiloint2double(dhi, dlo) : 1.0;
            if (live && !(d > 0.0) && lane == 0 && blockIdx.x == 0) atomicMin(notspd, sn.a + c1 + j);   // csparse.py:612
            const double ljj = sqrt(d);
            double lij = 0.0;
            if (row && lane > j) lij = a[j] / ljj;
            if (lane == j) a[j] = ljj;
            else a[j] = lij;
#pragma unroll
            for (int c = j + 1; c < NB; c++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(lij), c);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(lij), c);
                const double lcj = __hiloint2double(hi, lo);
                const double t = lij * lcj;
                if (lane >= c) a[c] = a[c] - t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB; c++) {
                D[c * NB + lane] = a[c];
                // the factored block goes to dfac (k_sn_put_blocks moves it into place when every panel is done): the
                // other workgroups of this launch read the block as the earlier launches left it
                if (blockIdx.x == 0 && row && c <= lane && c1 + c < w)
                    dfac[(int64_t)blockIdx.y * dstride + (int64_t)(c1 / NB) * NB * NB + c * NB + lane] = a[c];
            }
        }
    }
    __syncthreads();
    if (!mine) return;
#pragma unroll
    for (int t = 0; t < NB; t++) {
        double vv = x[t];
#pragma unroll
        for (int s2 = 0; s2 < t; s2++) {
            const double p = x[s2] * D[s2 * NB + t];
            vv = vv - p;
        }
        x[t] = vv / D[t * NB + t];
        asm volatile("" : "+v"(x[t]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NB; t++)
        if (c1 + t < w) colp(c1 + t)[v] = x[t];
}

#pragma clang fp contract(fast)

// the factored diagonal blocks into their places in L.x: grid (panels of the widest supernode, supernodes), NB * NB threads
template <int NB>
__global__ __launch_bounds__(NB * NB) void k_sn_put_blocks(const SnDesc *__restrict__ sns, const int32_t *__restrict__ Lp,
                                                           double *Lx, const double *__restrict__ dfac, int32_t dstride) {
    const SnDesc sn = sns[blockIdx.y];
    const int32_t c1 = (int32_t)blockIdx.x * NB;
    if (c1 >= sn.w) return;
    const int c = threadIdx.x / NB, i = threadIdx.x % NB;      // element (row c1 + i, column c1 + c) of the trapezoid
    if (i < c || c1 + c >= sn.w || c1 + i >= sn.w + sn.r) return;
    Lx[Lp[sn.a + c1 + c] + (i - c)] = dfac[(int64_t)blockIdx.y * dstride + (int64_t)blockIdx.x * NB * NB + c * NB + i];
}

// Factor nsn supernodes (descriptors on the device, independent of each other, outside updates already applied) in
// place: max_w / max_rows = the widest supernode / the most virtual rows (w + r) among them.
int chol_supernodes(const void *d_sns, int32_t nsn, int32_t max_w, int32_t max_rows, const int32_t *Lp, double *Lx,
                    int *notspd) {
    constexpr int NB = 16;
    hipStream_t s = ctx().stream;
    const int32_t npan = (max_w + NB - 1) / NB;
    const int32_t dstride = npan * NB * NB;
    DevScope tmp;
    double *dfac = nullptr;
    CSX_TRY(tmp.alloc(&dfac, (size_t)nsn * dstride));
    for (int32_t c1 = 0; c1 < max_w; c1 += NB) {
        const int32_t below = max_rows - (c1 + NB);
        const int32_t gA = std::max(1, (below + WB_PANEL_ROWS - 1) / WB_PANEL_ROWS);
        const int32_t ntR = below > 0 && c1 > 0 ? (below + WB_TILE - 1) / WB_TILE : 0;
        const int32_t ntC = max_w - (c1 + NB) > 0 && c1 > 0 ? (max_w - (c1 + NB) + WB_TILE - 1) / WB_TILE : 0;
        hipLaunchKernelGGL((k_sn_step<NB>), dim3((unsigned)(gA + ntR * ntC), (unsigned)nsn), dim3(64 + WB_PANEL_ROWS), 0, s,
                           (const SnDesc *)d_sns, c1, gA, ntR, ntC, Lp, Lx, dfac, dstride, notspd);
    }
    hipLaunchKernelGGL((k_sn_put_blocks<NB>), dim3((unsigned)npan, (unsigned)nsn), dim3(NB * NB), 0, s, (const SnDesc *)d_sns,
                       Lp, Lx, dfac, dstride);
    CSX_LAUNCH_CHECK();
    // dfac goes back to the cache when this returns: every use of device memory is ordered on the one stream, and the
    // cache hands a block out again only to work enqueued after this
    return CSX_OK;
}

// Bytes of the dense band array for a factor of n columns and half-width bw.
size_t chol_wide_band_bytes(int32_t n, int32_t bw) { return (size_t)n * ((size_t)bw + 1) * sizeof(double); }

// L.x holds A scattered into the pattern of L (k_chol_init); on return it holds the factor.  *notspd (device)
// receives the first column with a non-positive pivot (atomicMin), as in the other kernels.
template <int NB>
static int wide_band_run(int32_t n, int32_t bw, double *W, double *dfac, int *notspd) {
    hipStream_t s = ctx().stream;
    for (int32_t c0 = 0; c0 < n; c0 += NB) {
        const int32_t below = std::min(n - 1, c0 + NB - 1 + bw) - (c0 + NB) + 1;   // rows under the block (may be <= 0)
        const unsigned g2 = (unsigned)std::max(1, (below + WB_PANEL_ROWS - 1) / WB_PANEL_ROWS);
        hipLaunchKernelGGL((k_wband_panel<NB>), dim3(g2), dim3(64 + WB_PANEL_ROWS), 0, s, n, bw, c0, W, dfac, notspd);
        if (below > 0) {
            const unsigned nt = (unsigned)((below + WB_TILE - 1) / WB_TILE);
            hipLaunchKernelGGL((k_wband_update<NB>), dim3(nt, nt), dim3(256), 0, s, n, bw, c0, W);
        }
    }
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

template <int NB>
static int wide_band_run_fused(int32_t n, int32_t bw, double *W, double *dfac, int *notspd) {
    hipStream_t s = ctx().stream;
    for (int32_t c1 = 0; c1 < n; c1 += NB) {
        const int32_t below = std::min(n - 1, c1 + NB - 1 + bw) - (c1 + NB) + 1;    // rows under this panel's block
        const int32_t gA = std::max(1, (below + WB_PANEL_ROWS - 1) / WB_PANEL_ROWS);
        int32_t ntB = 0;
        if (c1 > 0) {
            const int32_t beyond = std::min(n - 1, c1 - 1 + bw) - (c1 + NB) + 1;     // window of the previous panel beyond this one
            if (beyond > 0) ntB = (beyond + WB_TILE - 1) / WB_TILE;
        }
        hipLaunchKernelGGL((k_wband_step<NB>), dim3((unsigned)(gA + ntB * ntB)), dim3(64 + WB_PANEL_ROWS), 0, s, n, bw, c1,
                           c1 > 0 ? 1 : 0, gA, ntB, W, dfac, notspd);
    }
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

int chol_wide_band(int32_t n, int32_t bw, const int32_t *Lp, const int32_t *Li, double *Lx, int *notspd, int nb) {
    hipStream_t s = ctx().stream;
    if (n <= 0) return CSX_OK;
    DevScope tmp;
    double *W = nullptr;
    const size_t count = (size_t)n * ((size_t)bw + 1);
    CSX_TRY(tmp.alloc(&W, count));
    CSX_HIP(hipMemsetAsync(W, 0, count * sizeof(double), s));
    const unsigned gw = (unsigned)(((int64_t)n + 3) / 4);
    const int NBv = (nb == 32 || nb == -32) ? 32 : 16;
    double *dfac = nullptr;                              // the factored diagonal blocks, panel-major
    CSX_TRY(tmp.alloc(&dfac, ((size_t)n / NBv + 1) * NBv * NBv));
    hipLaunchKernelGGL((k_wband_copy<true>), dim3(gw), dim3(256), 0, s, n, bw + 1, Lp, Li, Lx, W, dfac, NBv);
    if (nb == 16) CSX_TRY(wide_band_run_fused<16>(n, bw, W, dfac, notspd));
    else if (nb == 32) CSX_TRY(wide_band_run_fused<32>(n, bw, W, dfac, notspd));
    else if (nb == -16) CSX_TRY(wide_band_run<16>(n, bw, W, dfac, notspd));     // two launches per panel (tests, timing)
    else CSX_TRY(wide_band_run<32>(n, bw, W, dfac, notspd));
    hipLaunchKernelGGL((k_wband_copy<false>), dim3(gw), dim3(256), 0, s, n, bw + 1, Lp, Li, Lx, W, dfac, NBv);
    CSX_LAUNCH_CHECK();
    CSX_HIP(hipStreamSynchronize(s));   // W is released when this returns
    return CSX_OK;
}

}  // namespace csx
