// Shared device code of the fused in-LDS triangular sweeps: one wave = one small dependency component (an
// elimination tree of cs_chol's L, or a connected component of any triangular factor) x 64 right-hand sides.
// Used by csx_chol.hip (k_cholsol_local: forward + backward in one kernel) and csx_trisolve.hip (k_tri_local).
#ifndef CSX_SWEEP_H
#define CSX_SWEEP_H

#include "csx_internal.h"

namespace csx {

struct Tree {
    int32_t first, count;  // a component's rows: positions [first, first + count) of the node list
};

// One wave = one tree x 64 right-hand sides.  X tile in LDS: [node][lane].  The terms of a row
// are loaded coalesced (one term per lane) and broadcast with v_readlane; each lane applies them
// to its own right-hand side in the reference's order (multiply and subtract rounded
// separately), so the result is bit-identical to cs_lsolve + cs_ltsolve on this L.
// csx_components.hip
int connected_components(int32_t n, const int32_t *ptr, const int32_t *idx, int sf, int sl, int order, int32_t *root,
                         bool *malformed);
int group_by_root(int32_t n, const int32_t *root, uint32_t *nodes, int32_t *comp_of_pos, Tree **comps_out,
                  int32_t *ncomp_out, int32_t *max_count);

// A wave of the fused sweeps owns an X tile of `per_wave` bytes of LDS.  Waves per workgroup (1 .. maxw) that put the most
// waves on a CU (160 KB of LDS); of equal choices the smallest workgroup (the last round of a launch fills better).  W's
// 67-row components (34 KB a wave): 1 x 4 workgroups instead of 3 x 1 -- the old rule capped a workgroup at 128 KB.
// a copy of a component list ordered biggest first (stable; counts in [0, max_count]): the order single-wave-per-component kernels launch
// by -- workgroups go to the XCDs and shader engines in a fixed rotation, so a periodic pattern of big and small components in launch
// order becomes an imbalance between engines (csx_trisolve.hip: analyse_components)
int trees_biggest_first(const Tree *trees, int32_t ntrees, int32_t max_count, Tree **out);

inline int tile_waves_per_workgroup(size_t per_wave, int maxw) {
    const size_t cu = 160 * 1024 - 1024;
    int best = 1;
    size_t best_total = 0;
    for (int w = 1; w <= maxw; w++) {
        const size_t per_wg = per_wave * (size_t)w;
        if (per_wg > cu) break;
        const size_t total = (cu / per_wg) * (size_t)w;
        if (total > best_total) {
            best_total = total;
            best = w;
        }
    }
    return best;
}

#pragma clang fp contract(off)
struct TermRegs {  // up to 64 terms of one row, one per lane
    int32_t i;
    double v;
};

__device__ __forceinline__ TermRegs load_terms(const int32_t *__restrict__ idx, const double *__restrict__ val,
                                               int32_t q, int32_t qe, int lane) {
    TermRegs t;
    const bool in = q + lane < qe;
    t.i = in ? idx[q + lane] : 0;
    t.v = in ? val[q + lane] : 0.0;
    return t;
}

__device__ __forceinline__ double bcast_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// acc -= val[u] * X[idx[u] + lane] for the terms u0 <= u < u1 held one per lane in `t`, in order
__device__ __forceinline__ double apply_terms(double acc, const TermRegs &t, int u0, int u1, const double *X,
                                              int lane) {
    int u = u0;
    for (; u + 8 <= u1; u += 8) {  // 8 LDS reads in flight, then the (inherently serial) subtract chain
        double xx[8], vv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) xx[k] = X[__builtin_amdgcn_readlane(t.i, u + k) + lane];
#pragma unroll
        for (int k = 0; k < 8; k++) vv[k] = bcast_f64(t.v, u + k) * xx[k];
#pragma unroll
        for (int k = 0; k < 8; k++) acc = acc - vv[k];
    }
    for (; u < u1; u++) {
        const double p = bcast_f64(t.v, u) * X[__builtin_amdgcn_readlane(t.i, u) + lane];
        acc = acc - p;
    }
    return acc;
}

constexpr int SW_SLOTS = 32;  // 32 slots x 64 lanes = a window of 2048 terms in registers

// One sweep over a tree's rows in sweep order (forward: rows ascending; backward: descending --
// the backward program is packed in that order, so both sweeps stream their terms front to back).
// A window of up to 2048 terms is requested with 64 back-to-back coalesced loads and then walked
// row by row: term u of slot s is broadcast with v_readlane, every lane applies it to its own
// right-hand side.  Memory latency is paid once per window, not per row or per term.
template <bool FORWARD>
__device__ __forceinline__ void sweep(const Tree tr, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                      const double *__restrict__ val, const double *__restrict__ diag, double *X,
                                      int lane) {
    const int32_t count = tr.count;
    if (count == 0) return;
    const int32_t tbase = ptr[tr.first], tend = ptr[tr.first + count];
    // row ends and diagonals of 64 sweep positions at a time, handed out by v_readlane
    int32_t c0 = 0;
    int32_t pe_v = lane < count ? ptr[tr.first + lane + 1] : 0;
    double dg_v = lane < count ? diag[tr.first + lane] : 1.0;
    int32_t sp = 0;
    int32_t rend = __builtin_amdgcn_readlane(pe_v, 0);
    double acc = X[(FORWARD ? 0 : count - 1) * 64 + lane];
#define CSX_FINALIZE_ROW                                                                  \
    {                                                                                     \
        X[(FORWARD ? sp : count - 1 - sp) * 64 + lane] = acc / bcast_f64(dg_v, sp - c0);  \
        sp++;                                                                             \
        if (sp < count) {                                                                 \
            if (sp - c0 == 64) {                                                          \
                c0 = sp;                                                                  \
                pe_v = c0 + lane < count ? ptr[tr.first + c0 + lane + 1] : 0;             \
                dg_v = c0 + lane < count ? diag[tr.first + c0 + lane] : 1.0;              \
            }                                                                             \
            acc = X[(FORWARD ? sp : count - 1 - sp) * 64 + lane];                         \
            rend = __builtin_amdgcn_readlane(pe_v, sp - c0);                              \
        } else {                                                                          \
            rend = 0x7fffffff;                                                            \
        }                                                                                 \
    }
    for (int32_t w0 = tbase; w0 < tend; w0 += 64 * SW_SLOTS) {
        TermRegs T[SW_SLOTS];
        // (slots past the program's end are skipped by a wave-uniform branch, not by 64 idle lanes: a tree of 45 terms has one)
#pragma unroll
        for (int sl = 0; sl < SW_SLOTS; sl++)
            if (w0 + 64 * sl < tend) T[sl] = load_terms(idx, val, w0 + 64 * sl, tend, lane);
#pragma unroll
        for (int sl = 0; sl < SW_SLOTS; sl++) {
            const int32_t s0 = w0 + 64 * sl;
            if (s0 >= tend) break;
            const int ulim = min(64, tend - s0);
            int u = 0;
            while (u < ulim) {
                while (s0 + u == rend) CSX_FINALIZE_ROW
                const int run = min(ulim, rend - s0);
                acc = apply_terms(acc, T[sl], u, run, X, lane);
                u = run;
            }
        }
    }
    while (sp < count) CSX_FINALIZE_ROW
#undef CSX_FINALIZE_ROW
}

#pragma clang fp contract(fast)

}  // namespace csx
#endif
