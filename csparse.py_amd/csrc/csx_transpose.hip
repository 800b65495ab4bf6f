// cs_transpose (csparse.py:2292-2315) on the device.
//
// The reference transposes with a counting sort by row: row counts
// (:2305-2306), cs_cumsum (:2307), then a fill in ascending (column, position)
// order (:2308-2314).  That is a STABLE sort of the entries by row index.  Here:
// stable LSD radix sort of (row key, column, value) records (csx_sort.hip); the
// first pass derives each entry's column from its position, the last pass leaves
// the first slot of every row, from which the column pointers of the result follow.  p[] and i[] come out bit-identical to the reference,
// x[] is a pure permutation (no arithmetic).
//
// Algorithmic bytes: read 12 nnz + 4(n+1), write 12 nnz + 4(m+1).  The radix
// passes move more than that (one 16-byte record read + write per 8 key bits);
// DESIGN.md states the real traffic.
#include "csx_internal.h"

namespace csx {

int transpose_device(const Csc *A, bool values, Csc *C) {
    hipStream_t s = ctx().stream;
    const bool with_values = values && A->x != nullptr;
    C->m = A->n;
    C->n = A->m;
    C->nnz = A->nnz;
    C->owns = true;
    CSX_TRY(dalloc(&C->p, (size_t)C->n + 1));
    CSX_TRY(dalloc(&C->i, (size_t)C->nnz));
    if (with_values) CSX_TRY(dalloc(&C->x, (size_t)C->nnz));
    if (A->nnz == 0) {
        CSX_HIP(hipMemsetAsync(C->p, 0, ((size_t)C->n + 1) * sizeof(int32_t), s));
        return CSX_OK;
    }
    // one stable sort of the entries by row: columns are derived from positions in its first pass, the column
    // pointers of the result come out of its last pass
    SortExtra ex{A->p, A->n, C->p, A->m};
    int st = stable_sort_by_key_ex((const uint32_t *)A->i, nullptr, with_values ? A->x : nullptr, A->nnz, (uint32_t)A->m,
                                   nullptr, (uint32_t *)C->i, C->x, &ex);
    if (st == CSX_OK && hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;
    return st;
}

int build_row_gather(Csc *A) {
    if (A->rows) return CSX_OK;
    if (!A->x) return CSX_EINVAL;
    Csc T;
    int st = transpose_device(A, true, &T);
    if (st != CSX_OK) {
        dfree(T.p);
        dfree(T.i);
        dfree(T.x);
        return st;
    }
    Gather *g = new Gather();
    g->rows = A->m;
    g->ptr = T.p;
    g->idx = T.i;
    g->val = T.x;
    A->rows = g;
    return CSX_OK;
}

}  // namespace csx

using namespace csx;

extern "C" int csx_transpose(csx_handle_t hA, int values, csx_handle_t *out) {
    CSX_TRY(require_ready());
    Csc *A = csc(hA);
    if (!A || !out) return CSX_EINVAL;
    Csc *C = new Csc();
    int st = transpose_device(A, values != 0, C);
    if (st != CSX_OK) {
        free_csc(C);
        return st;
    }
    *out = put(K_CSC, C);
    return CSX_OK;
}
