// Small dependency components on the matrix cores (csx_trimfma.hip): the rounding-equal order of the fused in-LDS sweeps.
#ifndef CSX_TRIMFMA_H
#define CSX_TRIMFMA_H

#include "csx_internal.h"
#include "csx_sweep.h"

namespace csx {

constexpr int RAG_MAX_ROWS = 80;           // 5 tiles of 16
constexpr int RAG_CLASSES = RAG_MAX_ROWS / 16;
constexpr double RAG_GROWTH_LIMIT = 1e3;   // || |inv(T_ii)| |T_ii| ||_inf of a diagonal tile (the supernodal plan's guard, csx_snsolve.hip)

struct RaggedMfma {
    int32_t ntrees = 0;
    int32_t *list = nullptr;                     // device [ntrees]: component ids, ordered by size class (stable)
    void *desc = nullptr;                        // device [ntrees] int4 {first, count, base row or -1, id} in that order: what a solve reads
    int32_t cls_start[RAG_CLASSES + 1] = {0};    // class c (components of 16 c + 1 .. 16 (c + 1) rows): list[cls_start[c] .. cls_start[c + 1])
    size_t cls_frag[RAG_CLASSES + 1] = {0};      // first double of class c's fragments
    double *frag = nullptr;
    double growth = 0.0;                         // the guard's measure over all diagonal tiles
    bool cls_consecutive[RAG_CLASSES] = {false}; // every component of the class is consecutive rows of X (every desc base >= 0)
};

// The components' packed sweep programs (csx_sweep.h: per sweep position the terms (local row * 64, value), the diagonal) made
// dense in POSITION order -- position sp of a component is its row sp (forward sweeps) or count - 1 - sp (backward sweeps) --
// zero where the pattern has none, the identity on the padding, cut into 16 x 16 tiles: off-diagonal tiles negated, diagonal
// tiles inverted, fragment by fragment in use order (tile (a, b), b < a, then the inverse of diagonal tile a; 4 fragments of 64 doubles a tile).  *out = nullptr when a component has more than RAG_MAX_ROWS rows.
// from_factor (or null): the components are blocks of consecutive columns of this Cholesky-shaped factor (trees[b] = {first column,
// columns}, nodes the identity) and the dense triangles are read from its columns instead of from sweep programs (ptr .. diag unused).
int ragged_build(const Tree *trees, int32_t ntrees, int32_t max_rows, const int32_t *nodes, const int32_t *ptr, const int32_t *idx,
                 const double *val, const double *diag, bool reverse, RaggedMfma **out, const Csc *from_factor = nullptr);
// For a producer that writes the fragments itself (k_chol_clique with CliqueEmit::frag_off: cs_chol of a forest of cliques of unequal
// sizes): the class-ordered list, descriptors and fragment storage, and frag_off[t] = the first double of component t's fragments
// (device array the caller frees; layout per component as above, its class = ceil(rows / 16) tiles).  R->growth is the caller's to set.
int ragged_prepare_emit(const Tree *trees, int32_t ntrees, int32_t max_rows, const int32_t *nodes, RaggedMfma **out, int64_t **frag_off);
// trees[b] = {start[b], start[b + 1] - start[b]}, nodes = the identity (n entries): the block list of such a factor
int ragged_blocks(const int32_t *start, int32_t nblocks, int32_t n, Tree *trees, int32_t *nodes);
// X (n-by-nrhs, row-major) <- the sweep applied to every component: a blocked substitution in position order on the matrix cores.
// passes = 1: that sweep; passes = 2: then the TRANSPOSED system backwards (cs_cholsol's L then L', the fragments read transposed).
// perm (or null): row j of the components is row perm[j] of X.
int ragged_solve(const RaggedMfma *R, const int32_t *nodes, const int32_t *perm, bool reverse, int passes, double *X, int32_t nrhs,
                 int32_t n_rows = 0);
// The same from one block into another (src == dst: in place), with a row permutation on either side: position p of a component
// (node j = nodes[first + p]) is read from row load_rows[j] of src (null: row j) and written to row store_rows[j] of dst (null: row j).
// n_rows: rows of the blocks (the fused permutations of cs_lusol address a whole block through one 32-bit resource; 0: unknown).
int ragged_solve_io(const RaggedMfma *R, const int32_t *nodes, const int32_t *load_rows, const int32_t *store_rows, bool reverse, int passes,
                    const double *src, double *dst, int32_t nrhs, int32_t n_rows);
void ragged_free(RaggedMfma *R);

}  // namespace csx
#endif
