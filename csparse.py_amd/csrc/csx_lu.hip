// cs_lu (csparse.py:1370-1451), natural column order, for matrices that are a BATCH OF SMALL INDEPENDENT BLOCKS
// (BASELINE config 3's W: 1 493 blocks of 67 x 67; "batches of independent matrices" of the north star).
//
// The reference factors left-looking, one column after the other: reach of A(:,k) in the graph of L (DFS), sparse
// triangular solve, threshold pivot search.  Inside one matrix that is sequential (csx_lu_host does it in host
// C++).  But when the graph of A falls into many small connected components (csx_components.hip) the components
// factor independently, and the pivot found in a component never depends on another one: ONE LANE PER COMPONENT
// runs the very same loop (same DFS order, same order of the updates, same pivot rule with its first-maximum
// tie-break, same order of the entries it appends to L and U), 64 components to a wave, the small work arrays in
// LDS, the factors appended to a private strip of a scratch buffer and then moved to their place in the global L
// and U once the column counts are scanned.  The result is the host code's, bit for bit: L (unit diagonal first),
// U (diagonal last), pinv -- tests/test_gpu_lu_blocks.py.  W: 95 ms on one host core -> see DESIGN.md.
#include "csx_internal.h"
#include "csx_sweep.h"

namespace csx {

constexpr int LU_MAX_M = 96;          // rows of a component (work arrays: 24 B per row and lane in LDS)
constexpr int LU_MIN_COMPONENTS = 64;

// flags: [0] a singular block (smallest such column, as min)
#pragma clang fp contract(off)   // the host code rounds multiply and subtract separately (x86-64 baseline, no FMA)
__global__ __launch_bounds__(64) void k_lu_blocks(const Tree *__restrict__ comps, int32_t ncomp,
                                                  const uint32_t *__restrict__ nodes,
                                                  const int32_t *__restrict__ local_id, const int32_t *__restrict__ Ap,
                                                  const int32_t *__restrict__ Ai, const double *__restrict__ Ax,
                                                  double tol, int32_t ld, int32_t *sLi, double *sLx, int32_t *sUi,
                                                  double *sUx, int32_t *sLp, int32_t *sUp, int32_t *lcount,
                                                  int32_t *ucount, int32_t *pinv_out, int *flags) {
    extern __shared__ __attribute__((aligned(16))) double lu_smem[];
    const int lane = threadIdx.x;
    const int32_t c = blockIdx.x * 64 + lane;
    // per-lane work arrays, element e of lane l at [e * 64 + l] (no bank conflicts between lanes)
    double *x = lu_smem;                                               // [ld * 64]
    int32_t *reach = reinterpret_cast<int32_t *>(x + (size_t)ld * 64); // [ld * 64]
    int32_t *stack = reach + (size_t)ld * 64;
    int32_t *pos = stack + (size_t)ld * 64;
    int32_t *pinv = pos + (size_t)ld * 64;                             // local: row -> pivot position, -1
    unsigned char *seen = reinterpret_cast<unsigned char *>(pinv + (size_t)ld * 64);
#define AT(arr, e) arr[(size_t)(e) * 64 + lane]
    if (c >= ncomp) return;
    const Tree tr = comps[c];
    const int32_t m = tr.count;
    // this component's strips: column pointers (m + 1) and up to m * m entries of each factor
    int32_t *Lp = sLp + (size_t)tr.first + c, *Up = sUp + (size_t)tr.first + c;   // (m + 1) slots per component
    int32_t *Li = sLi + (size_t)tr.first * ld, *Ui = sUi + (size_t)tr.first * ld;
    double *Lx = sLx + (size_t)tr.first * ld, *Ux = sUx + (size_t)tr.first * ld;
    for (int32_t i = 0; i < m; i++) {
        AT(x, i) = 0.0;
        AT(pinv, i) = -1;
        AT(seen, i) = 0;
    }
    int32_t lnz = 0, unz = 0;
    bool singular = false;
    for (int32_t k = 0; k < m && !singular; k++) {
        const int32_t j = (int32_t)nodes[tr.first + k];
        Lp[k] = lnz;
        Up[k] = unz;
        // reach of A(:,k) in the graph of L: depth-first search, topological order in reach[top..m-1]
        int32_t top = m;
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int32_t r0 = local_id[Ai[p]];
            if (AT(seen, r0)) continue;
            int32_t head = 0;
            AT(stack, 0) = r0;
            while (head >= 0) {
                const int32_t jj = AT(stack, head);
                const int32_t col = AT(pinv, jj);
                if (!AT(seen, jj)) {
                    AT(seen, jj) = 1;
                    AT(pos, head) = col < 0 ? 0 : Lp[col];
                }
                bool done = true;
                const int32_t end = col < 0 ? 0 : (col == k ? lnz : Lp[col + 1]);
                for (int32_t q = AT(pos, head); q < end; q++) {
                    const int32_t i = Li[q];
                    if (AT(seen, i)) continue;
                    AT(pos, head) = q;
                    AT(stack, ++head) = i;
                    done = false;
                    break;
                }
                if (done) {
                    head--;
                    AT(reach, --top) = jj;
                }
            }
        }
        for (int32_t p = top; p < m; p++) {
            AT(seen, AT(reach, p)) = 0;
            AT(x, AT(reach, p)) = 0.0;
        }
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) AT(x, local_id[Ai[p]]) = Ax[p];
        // sparse triangular solve x = L \ A(:,k) along the reach
        for (int32_t px = top; px < m; px++) {
            const int32_t jj = AT(reach, px), col = AT(pinv, jj);
            if (col < 0) continue;
            AT(x, jj) = AT(x, jj) / Lx[Lp[col]];
            const double xj = AT(x, jj);
            for (int32_t q = Lp[col] + 1; q < Lp[col + 1]; q++) {
                const double t = Lx[q] * xj;
                AT(x, Li[q]) = AT(x, Li[q]) - t;
            }
        }
        // pivot search among the non-pivotal rows (first maximum in reach order); pivotal rows go to U
        int32_t ipiv = -1;
        double a = -1.0;
        for (int32_t p = top; p < m; p++) {
            const int32_t i = AT(reach, p);
            if (AT(pinv, i) < 0) {
                const double t = fabs(AT(x, i));
                if (t > a) {
                    a = t;
                    ipiv = i;
                }
            } else {
                Ui[unz] = AT(pinv, i);
                Ux[unz++] = AT(x, i);
            }
        }
        if (ipiv == -1 || a <= 0) {
            singular = true;
            break;
        }
        if (AT(pinv, k) < 0 && fabs(AT(x, k)) >= a * tol) ipiv = k;
        const double pivot = AT(x, ipiv);
        Ui[unz] = k;
        Ux[unz++] = pivot;
        AT(pinv, ipiv) = k;
        Li[lnz] = ipiv;
        Lx[lnz++] = 1.0;
        for (int32_t p = top; p < m; p++) {
            const int32_t i = AT(reach, p);
            if (AT(pinv, i) < 0) {
                Li[lnz] = i;
                Lx[lnz++] = AT(x, i) / pivot;
            }
            AT(x, i) = 0.0;
        }
        Lp[k + 1] = lnz;           // L's column k is complete only now (the search above used lnz as its end)
    }
    if (singular) {
        atomicMin(&flags[0], (int)nodes[tr.first]);
        return;
    }
    Lp[m] = lnz;
    Up[m] = unz;
    for (int32_t k = 0; k < m; k++) {
        const int32_t j = (int32_t)nodes[tr.first + k];
        lcount[j] = Lp[k + 1] - Lp[k];
        ucount[j] = Up[k + 1] - Up[k];
        pinv_out[j] = (int32_t)nodes[tr.first + AT(pinv, k)];
    }
    // the reference's last step, Li = pinv[Li] (csparse.py:1447-1448), in local numbering; the fill kernel maps to global
    for (int32_t q = 0; q < lnz; q++) Li[q] = AT(pinv, Li[q]);
#undef AT
}
#pragma clang fp contract(fast)

// one wave per column: copy its strip entries to their place, local pivot positions -> global indices
__global__ __launch_bounds__(256) void k_lu_fill(int32_t n, const int32_t *__restrict__ comp_of_pos,
                                                 const Tree *__restrict__ comps, const uint32_t *__restrict__ nodes,
                                                 int32_t ld, const int32_t *__restrict__ sLi, const double *__restrict__ sLx,
                                                 const int32_t *__restrict__ sUi, const double *__restrict__ sUx,
                                                 const int32_t *__restrict__ sLp, const int32_t *__restrict__ sUp,
                                                 const int32_t *__restrict__ Lp, int32_t *Li, double *Lx,
                                                 const int32_t *__restrict__ Up, int32_t *Ui, double *Ux) {
    const int lane = threadIdx.x & 63;
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // position in the node list
    if (k >= n) return;
    const int32_t c = comp_of_pos[k];
    const Tree tr = comps[c];
    const int32_t cc = (int32_t)k - tr.first, j = (int32_t)nodes[k];
    const int32_t *lp = sLp + (size_t)tr.first + c, *up = sUp + (size_t)tr.first + c;
    const size_t base = (size_t)tr.first * ld;
    for (int32_t q = lp[cc] + lane; q < lp[cc + 1]; q += 64) {
        Li[Lp[j] + (q - lp[cc])] = (int32_t)nodes[tr.first + sLi[base + q]];
        Lx[Lp[j] + (q - lp[cc])] = sLx[base + q];
    }
    for (int32_t q = up[cc] + lane; q < up[cc + 1]; q += 64) {
        Ui[Up[j] + (q - up[cc])] = (int32_t)nodes[tr.first + sUi[base + q]];
        Ux[Up[j] + (q - up[cc])] = sUx[base + q];
    }
}

__global__ __launch_bounds__(256) void k_lu_local_id(int32_t ncomp, const Tree *__restrict__ comps,
                                                     const uint32_t *__restrict__ nodes, int32_t *local_id) {
    const int lane = threadIdx.x & 63;
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (c >= ncomp) return;
    const Tree t = comps[c];
    for (int32_t a = lane; a < t.count; a += 64) local_id[nodes[t.first + a]] = a;
}


// ---- cs_qr (csparse.py:1797-1870, with cs_house :1238-1261 and cs_happly :1216-1235) for a BATCH OF SMALL
// INDEPENDENT BLOCKS, on the device ---------------------------------------------------------------------------
// Column k's Householder vector depends on the columns before it along the column elimination tree: inside one
// matrix the factorisation is a sequence (csx_qr_host does it in host C++).  When the matrix is square, has no
// fictitious rows (m2 == m) and falls into many small connected components, the components factor independently and
// ONE LANE PER COMPONENT runs the host code's loop statement for statement -- same order of the reflections applied
// to a column, same order of every sum, multiply and add rounded separately -- with its work arrays (stamp, order,
// path: 4 B, work: 8 B per row) interleaved in LDS and V, R appended to private strips that are moved to their place
// once the column counts are scanned (k_lu_fill of csx_lu.hip does that for any pair of factors).  V, R and beta come
// out bit-identical to csx_qr_host (tests/test_gpu_qr_device.py).
constexpr int QR_MAX_M = 96;
constexpr int QR_MIN_COMPONENTS = 64;

#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void k_qr_blocks(const Tree *__restrict__ comps, int32_t ncomp,
                                                  const uint32_t *__restrict__ nodes,
                                                  const int32_t *__restrict__ local_id, const int32_t *__restrict__ root,
                                                  const int32_t *__restrict__ Ap, const int32_t *__restrict__ Ai,
                                                  const double *__restrict__ Ax, const int32_t *__restrict__ parent,
                                                  const int32_t *__restrict__ pinv, const int32_t *__restrict__ leftmost,
                                                  int32_t ld, int32_t *sVi, double *sVx, int32_t *sRi, double *sRx,
                                                  int32_t *sVp, int32_t *sRp, int32_t *vcount, int32_t *rcount,
                                                  double *beta, int *flags) {
    extern __shared__ __attribute__((aligned(16))) double qr_smem[];
    const int lane = threadIdx.x;
    const int32_t c = blockIdx.x * 64 + lane;
    double *work = qr_smem;                                                    // [ld * 64]
    int32_t *stamp = reinterpret_cast<int32_t *>(work + (size_t)ld * 64);      // [ld * 64] each
    int32_t *order = stamp + (size_t)ld * 64;
    int32_t *path = order + (size_t)ld * 64;
#define AT(arr, e) arr[(size_t)(e) * 64 + lane]
    if (c >= ncomp) return;
    const Tree tr = comps[c];
    const int32_t m = tr.count;
    const int32_t myroot = root[nodes[tr.first]];
    int32_t *Vp = sVp + (size_t)tr.first + c, *Rp = sRp + (size_t)tr.first + c;      // (m + 1) slots per component
    int32_t *Vi = sVi + (size_t)tr.first * ld, *Ri = sRi + (size_t)tr.first * ld;    // local indices
    double *Vx = sVx + (size_t)tr.first * ld, *Rx = sRx + (size_t)tr.first * ld;
    for (int32_t i = 0; i < m; i++) {
        AT(stamp, i) = -1;
        AT(work, i) = 0.0;
    }
    int32_t vnz = 0, rnz = 0;
    bool bad = false;
    for (int32_t k = 0; k < m && !bad; k++) {
        const int32_t j = (int32_t)nodes[tr.first + k];
        Rp[k] = rnz;
        Vp[k] = vnz;
        const int32_t vstart = vnz;
        AT(stamp, k) = k;                                   // the diagonal position leads V(:,k)
        Vi[vnz++] = k;
        int32_t front = m;                                  // order[front..m) = reflections to apply, children before parents
        for (int32_t p = Ap[j]; p < Ap[j + 1] && !bad; p++) {
            const int32_t i = Ai[p];
            int32_t len = 0;
            for (int32_t cc = leftmost[i];; cc = parent[cc]) {
                if (cc < 0 || root[cc] != myroot) {         // symbolic data that does not fit the block structure
                    bad = true;
                    break;
                }
                const int32_t lc = local_id[cc];
                if (AT(stamp, lc) == k) break;
                AT(path, len++) = lc;
                AT(stamp, lc) = k;
            }
            if (bad) break;
            while (len > 0) AT(order, --front) = AT(path, --len);
            const int32_t rg = pinv[i];
            if (rg < 0 || root[rg] != myroot) {
                bad = true;
                break;
            }
            const int32_t r = local_id[rg];
            AT(work, r) = Ax[p];
            if (r > k && AT(stamp, r) < k) {                // a row below the diagonal not yet in V(:,k)'s pattern
                Vi[vnz++] = r;
                AT(stamp, r) = k;
            }
        }
        if (bad) break;
        for (int32_t t = front; t < m; t++) {
            const int32_t cl = AT(order, t);                // a column before k, local
            const int32_t cg = (int32_t)nodes[tr.first + cl];
            const double b = beta[cg];
            double dot = 0.0;                               // y <- (I - b v v') y for the stored reflection cl
            for (int32_t p = Vp[cl]; p < Vp[cl + 1]; p++) {
                const double pr = Vx[p] * AT(work, Vi[p]);
                dot = dot + pr;
            }
            dot = dot * b;
            for (int32_t p = Vp[cl]; p < Vp[cl + 1]; p++) {
                const double pr = Vx[p] * dot;
                AT(work, Vi[p]) = AT(work, Vi[p]) - pr;
            }
            Ri[rnz] = cl;
            Rx[rnz++] = AT(work, cl);
            AT(work, cl) = 0.0;
            if (parent[cg] == j) {                          // V(:,cl)'s rows below cl pass on to V(:,k)
                for (int32_t p = Vp[cl]; p < Vp[cl + 1]; p++) {
                    const int32_t r = Vi[p];
                    if (AT(stamp, r) < k) {
                        AT(stamp, r) = k;
                        Vi[vnz++] = r;
                    }
                }
            }
        }
        for (int32_t p = vstart; p < vnz; p++) {
            Vx[p] = AT(work, Vi[p]);
            AT(work, Vi[p]) = 0.0;
        }
        // Householder vector of Vx[vstart..vnz): afterwards (I - beta v v') x = s e1 (csparse.py:1238-1261)
        double tail2 = 0.0;
        for (int32_t p = vstart + 1; p < vnz; p++) {
            const double sq = Vx[p] * Vx[p];
            tail2 = tail2 + sq;
        }
        const double head = Vx[vstart];
        double sn, b;
        if (tail2 == 0.0) {
            sn = fabs(head);
            b = head <= 0.0 ? 2.0 : 0.0;
            Vx[vstart] = 1.0;
        } else {
            const double hh = head * head;
            sn = sqrt(hh + tail2);
            Vx[vstart] = head <= 0.0 ? head - sn : -tail2 / (head + sn);
            const double den = sn * Vx[vstart];
            b = -1.0 / den;
        }
        Ri[rnz] = k;
        Rx[rnz++] = sn;
        beta[j] = b;
        Vp[k + 1] = vnz;                                    // later columns read V(:,k) through Vp[k + 1]
    }
    if (bad) {
        flags[0] = 1;
        return;
    }
    Vp[m] = vnz;
    Rp[m] = rnz;
    for (int32_t k = 0; k < m; k++) {
        const int32_t j = (int32_t)nodes[tr.first + k];
        vcount[j] = Vp[k + 1] - Vp[k];
        rcount[j] = Rp[k + 1] - Rp[k];
    }
#undef AT
}
#pragma clang fp contract(fast)

}  // namespace csx

using namespace csx;

// *done = 0: the matrix is not a batch of small blocks: use csx_lu_host.
extern "C" int csx_lu_blocks(csx_handle_t hA, double tol, csx_handle_t *hL, csx_handle_t *hU, int32_t *pinv_host,
                             int *done) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !A->x || A->m != A->n || !hL || !hU || !pinv_host || !done) return CSX_EINVAL;
    *done = 0;
    const int32_t n = A->n;
    if (n < LU_MIN_COMPONENTS) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *root = nullptr, *comp_of_pos = nullptr, *local_id = nullptr, *lcount = nullptr, *ucount = nullptr,
            *d_pinv = nullptr;
    uint32_t *nodes = nullptr;
    int *flags = nullptr;
    bool bad = false;
    CSX_TRY(tmp.alloc(&root, (size_t)n));
    CSX_TRY(connected_components(n, A->p, A->i, 0, 0, 0, root, &bad));
    if (bad) return CSX_EINVAL;
    CSX_TRY(tmp.alloc(&nodes, (size_t)n));
    CSX_TRY(tmp.alloc(&comp_of_pos, (size_t)n));
    Tree *comps = nullptr;
    int32_t ncomp = 0, maxc = 0;
    int st = group_by_root(n, root, nodes, comp_of_pos, &comps, &ncomp, &maxc);
    tmp.held.push_back(comps);
    CSX_TRY(st);
    if (ncomp < LU_MIN_COMPONENTS || maxc > LU_MAX_M) return CSX_OK;
    const int32_t ld = maxc;                            // a component of m rows owns a strip of m * ld >= m * m entries
    int32_t *sLi = nullptr, *sUi = nullptr, *sLp = nullptr, *sUp = nullptr;
    double *sLx = nullptr, *sUx = nullptr;
    CSX_TRY(tmp.alloc(&local_id, (size_t)n));
    CSX_TRY(tmp.alloc(&lcount, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&ucount, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&d_pinv, (size_t)n));
    CSX_TRY(tmp.alloc(&sLi, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sLx, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sUi, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sUx, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sLp, (size_t)n + ncomp + 1));
    CSX_TRY(tmp.alloc(&sUp, (size_t)n + ncomp + 1));
    CSX_TRY(tmp.alloc(&flags, 2));
    int hflags[2] = {0x7fffffff, 0};
    CSX_HIP(hipMemcpyAsync(flags, hflags, sizeof hflags, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_lu_local_id, dim3((unsigned)((ncomp + 3) / 4)), dim3(256), 0, s, ncomp, comps, nodes, local_id);
    const size_t lds = (size_t)ld * 64 * (sizeof(double) + 4 * sizeof(int32_t) + 1) + 64;
    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lu_blocks), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256));
    hipLaunchKernelGGL(k_lu_blocks, dim3((unsigned)((ncomp + 63) / 64)), dim3(64), lds, s, comps, ncomp, nodes, local_id, A->p,
                       A->i, A->x, tol, ld, sLi, sLx, sUi, sUx, sLp, sUp, lcount, ucount, d_pinv, flags);
    CSX_LAUNCH_CHECK();
    CSX_HIP(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (hflags[0] != 0x7fffffff) return CSX_ENOTSPD;    // a singular block: the reference returns None
    Csc *L = new Csc(), *U = new Csc();
    L->m = L->n = U->m = U->n = n;
    int64_t lnz = 0, unz = 0;
    st = dalloc(&L->p, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&U->p, (size_t)n + 1);
    if (st == CSX_OK) st = scan_exclusive_i32(lcount, L->p, n, &lnz);
    if (st == CSX_OK) st = scan_exclusive_i32(ucount, U->p, n, &unz);
    if (st == CSX_OK) {
        L->nnz = (int32_t)lnz;
        U->nnz = (int32_t)unz;
        st = dalloc(&L->i, (size_t)lnz);
    }
    if (st == CSX_OK) st = dalloc(&L->x, (size_t)lnz);
    if (st == CSX_OK) st = dalloc(&U->i, (size_t)unz);
    if (st == CSX_OK) st = dalloc(&U->x, (size_t)unz);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_lu_fill, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, comp_of_pos, comps, nodes, ld, sLi,
                           sLx, sUi, sUx, sLp, sUp, L->p, L->i, L->x, U->p, U->i, U->x);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(pinv_host, d_pinv, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            st = CSX_ERUNTIME;
    }
    if (st != CSX_OK) {
        free_csc(L);
        free_csc(U);
        return st;
    }
    *hL = put(K_CSC, L);
    *hU = put(K_CSC, U);
    *done = 1;
    return CSX_OK;
}

// *done = 0: not a batch of small blocks (or fictitious rows / a column order): use csx_qr_host.
extern "C" int csx_qr_blocks(csx_handle_t hA, const int32_t *parent_host, const int32_t *pinv_host,
                             const int32_t *leftmost_host, int32_t m2, csx_handle_t *hV, csx_handle_t *hR,
                             double *beta_host, int *done) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !A->x || !parent_host || !pinv_host || !leftmost_host || !hV || !hR || !beta_host || !done) return CSX_EINVAL;
    *done = 0;
    const int32_t n = A->n;
    if (A->m != n || m2 != n || n < QR_MIN_COMPONENTS) return CSX_OK;
    for (int32_t i = 0; i < n; i++)
        if (parent_host[i] >= n || pinv_host[i] >= n || leftmost_host[i] >= n) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *root = nullptr, *comp_of_pos = nullptr, *local_id = nullptr, *vcount = nullptr, *rcount = nullptr,
            *d_parent = nullptr, *d_pinv = nullptr, *d_left = nullptr;
    uint32_t *nodes = nullptr;
    double *d_beta = nullptr;
    int *flags = nullptr;
    bool bad = false;
    CSX_TRY(tmp.alloc(&root, (size_t)n));
    CSX_TRY(connected_components(n, A->p, A->i, 0, 0, 0, root, &bad));
    if (bad) return CSX_EINVAL;
    CSX_TRY(tmp.alloc(&nodes, (size_t)n));
    CSX_TRY(tmp.alloc(&comp_of_pos, (size_t)n));
    Tree *comps = nullptr;
    int32_t ncomp = 0, maxc = 0;
    int st = group_by_root(n, root, nodes, comp_of_pos, &comps, &ncomp, &maxc);
    tmp.held.push_back(comps);
    CSX_TRY(st);
    if (ncomp < QR_MIN_COMPONENTS || maxc > QR_MAX_M) return CSX_OK;
    const int32_t ld = maxc;
    int32_t *sVi = nullptr, *sRi = nullptr, *sVp = nullptr, *sRp = nullptr;
    double *sVx = nullptr, *sRx = nullptr;
    CSX_TRY(tmp.alloc(&local_id, (size_t)n));
    CSX_TRY(tmp.alloc(&vcount, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&rcount, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&d_parent, (size_t)n));
    CSX_TRY(tmp.alloc(&d_pinv, (size_t)n));
    CSX_TRY(tmp.alloc(&d_left, (size_t)n));
    CSX_TRY(tmp.alloc(&d_beta, (size_t)n));
    CSX_TRY(tmp.alloc(&sVi, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sVx, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sRi, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sRx, (size_t)n * ld));
    CSX_TRY(tmp.alloc(&sVp, (size_t)n + ncomp + 1));
    CSX_TRY(tmp.alloc(&sRp, (size_t)n + ncomp + 1));
    CSX_TRY(tmp.alloc(&flags, 2));
    CSX_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(int), s));
    CSX_HIP(hipMemcpyAsync(d_parent, parent_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(d_pinv, pinv_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    CSX_HIP(hipMemcpyAsync(d_left, leftmost_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_lu_local_id, dim3((unsigned)((ncomp + 3) / 4)), dim3(256), 0, s, ncomp, comps, nodes, local_id);
    const size_t lds = (size_t)ld * 64 * (sizeof(double) + 3 * sizeof(int32_t)) + 64;
    CSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_qr_blocks), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256));
    hipLaunchKernelGGL(k_qr_blocks, dim3((unsigned)((ncomp + 63) / 64)), dim3(64), lds, s, comps, ncomp, nodes, local_id, root,
                       A->p, A->i, A->x, d_parent, d_pinv, d_left, ld, sVi, sVx, sRi, sRx, sVp, sRp, vcount, rcount, d_beta,
                       flags);
    CSX_LAUNCH_CHECK();
    int hflags[2] = {0, 0};
    CSX_HIP(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (hflags[0]) return CSX_OK;                       // the analysis does not follow the blocks: host code
    Csc *V = new Csc(), *R = new Csc();
    V->m = V->n = R->m = R->n = n;
    int64_t vnz = 0, rnz = 0;
    st = dalloc(&V->p, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&R->p, (size_t)n + 1);
    if (st == CSX_OK) st = scan_exclusive_i32(vcount, V->p, n, &vnz);
    if (st == CSX_OK) st = scan_exclusive_i32(rcount, R->p, n, &rnz);
    if (st == CSX_OK) {
        V->nnz = (int32_t)vnz;
        R->nnz = (int32_t)rnz;
        st = dalloc(&V->i, (size_t)vnz);
    }
    if (st == CSX_OK) st = dalloc(&V->x, (size_t)vnz);
    if (st == CSX_OK) st = dalloc(&R->i, (size_t)rnz);
    if (st == CSX_OK) st = dalloc(&R->x, (size_t)rnz);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_lu_fill, dim3((unsigned)(((int64_t)n + 3) / 4)), dim3(256), 0, s, n, comp_of_pos, comps, nodes, ld, sVi,
                           sVx, sRi, sRx, sVp, sRp, V->p, V->i, V->x, R->p, R->i, R->x);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(beta_host, d_beta, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            st = CSX_ERUNTIME;
    }
    if (st != CSX_OK) {
        free_csc(V);
        free_csc(R);
        return st;
    }
    *hV = put(K_CSC, V);
    *hR = put(K_CSC, R);
    *done = 1;
    return CSX_OK;
}
