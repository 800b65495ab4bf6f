// Forests of cliques on consecutive columns (csx_cholclique.hip): symbolic analysis and numeric factorisation of
// block-diagonal SPD matrices with dense blocks without the general pattern machine.
#ifndef CSX_CHOLCLIQUE_H
#define CSX_CHOLCLIQUE_H

#include <cstring>

#include "csx_internal.h"

namespace csx {

struct CliqueForest {
    int32_t n = 0;
    int32_t nblocks = 0, max_bs = 0;
    int32_t min_bs = 0;          // narrowest block (min_bs == max_bs: a forest of EQUAL blocks, what the dense-block solve kernels want)
    int64_t lnz = 0;
    bool ascending = true;       // the upper part of every column is strictly ascending (what k_chol_clique needs)
    bool dense_in_front = false; // ... and is rows u[k] .. k, one each, stored before any lower entry: k_chol_clique skips A.i
    bool sparse = false;         // the blocks are small TREES, not cliques (columns of L shorter than the block): parent / cp come
                                 // from the symbolic elimination on row masks (k_forest_symbolic), k_chol_clique stores compacted
    int32_t *parent = nullptr;   // device [n]: elimination tree (csparse.py:1136-1169)
    int32_t *cp = nullptr;       // device [n + 1]: column pointers of L (csparse.py:2069-2071)
    int32_t *start = nullptr;    // device [nblocks + 1]: first column of every block, then n
    unsigned long long *colmask = nullptr;   // device [n], sparse only: the rows of column k of L as bits (bit r = row start + r)
    int32_t *order = nullptr;    // device [nblocks], blocks of UNEQUAL sizes only: the blocks biggest first (stable) -- the order the
                                 // block kernel takes them in, so that the four waves of a workgroup hold blocks of like size
};

void free_clique(CliqueForest *F);
// *ok = A's elimination forest is a set of cliques on consecutive columns (F filled; the caller frees it)
int clique_forest(const Csc *A, CliqueForest *F, bool *ok);
// Are the host arrays parent[n], cp[n + 1] F's?  Three steps so that the caller's kernel runs beside the upload: _begin takes
// the temporaries, _run (after the caller has queued its own work) uploads and compares on a side stream, _end waits and answers.
struct CliqueCompare {
    const CliqueForest *F = nullptr;
    const int32_t *parent = nullptr, *cp = nullptr;
    int32_t *dp = nullptr, *dc = nullptr;
    int *bad = nullptr;
    int h = 1;
    hipEvent_t ev = nullptr;     // recorded on the context's stream when the temporaries were taken: the side stream waits for it
};
int clique_matches_begin(const CliqueForest &F, const int32_t *parent, const int32_t *cp, CliqueCompare *c);
int clique_matches_run(CliqueCompare *c);
int clique_matches_end(CliqueCompare *c, bool *same);
// What the block kernel can write BESIDE L.x while it has a finished block in registers (csx_cholsol_factor: factor -> plan in one
// kernel instead of k_clique_factor_shape + k_clique_plan + k_mfma_frags reading L back): what the matrix-core solve needs beyond L.x
// for a forest of EQUAL dense blocks of 16 / 32 / 64 columns -- per block the inverses W_ii = inv(L_ii) of its diagonal tiles as A
// fragments (k_cholsol_mfma reads the off-diagonal tiles in L.x itself) -- the guard's measure max|L| max|W| over the forest, and the
// plan's block list.  With `emit`, L.i is NOT written (L->i may be null: Csc::rows_pending).
struct Tree;
struct CliqueEmit {
    double *frag = nullptr;               // equal blocks: [nblocks * (bs / 16) * 256], tile i of block t at (t bs / 16 + i) * 256
    unsigned long long *cond_bits = nullptr;   // ordered bits of the largest max|L| max|W| of a block (zeroed by the caller)
    Tree *trees = nullptr;                // [nblocks]  {first column, columns}   (null: the caller has made the block list)
    int32_t *tree_nodes = nullptr;        // [n]        the identity node list
    const int64_t *frag_off = nullptr;    // blocks of UNEQUAL sizes: frag_off[t] = first double of block t's fragments (csx_trimfma.h:
                                          // ragged_prepare_emit); a block of bs columns then takes ceil(bs / 16) tiles a side, its last
                                          // tile row / column padded with the identity, and its fragments are ALL its tiles (-L_ij
                                          // and W_ii, clique_frags_per_block of them).  null: equal blocks
    const int32_t *order = nullptr;       // the blocks biggest first (CliqueForest::order; null: matrix order) -- set by chol_clique_numeric
    const int32_t *list = nullptr;        // with frag_off: the blocks ordered by size class (ceil(bs / 16) - 1), class c at
    int32_t cls_start[6] = {0, 0, 0, 0, 0, 0};   // list[cls_start[c] .. cls_start[c + 1]): what the matrix-core block kernel launches by
};
// values and row indices of L (L->p = F.cp already in place, L->i / L->x allocated); blocks of at most 64 columns
// relaxed ("chol.exact" = 0): fused multiply-adds and reciprocal square roots -- L.x equal to the exact kernel's to rounding
int chol_clique_numeric(const Csc *A, const CliqueForest &F, Csc *L, int *d_notspd, const CliqueEmit *emit = nullptr, bool relaxed = false);
constexpr int clique_frags_per_block(int nb16) { return (nb16 * (nb16 - 1) / 2 + nb16) * 4; }

// Column `col` of the inverse of a 16 x 16 lower-triangular tile T (element (r, q) at T[r * ld + q]): w[r] = inv(T)(r, col), zero
// above the diagonal.  Shared by k_mfma_frags (csx_chol.hip: fragments from a plan's programs) and k_chol_clique (fragments from
// the block in registers), so that both give a plan the same bits.
__device__ __forceinline__ void tile_inverse_column(const double *T, int ld, int col, double *w) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        double sres = r == col ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < r; q++) sres -= T[r * ld + q] * (q >= col ? w[q] : 0.0);
        w[r] = r >= col ? sres / T[r * ld + r] : 0.0;
    }
}
constexpr int CLIQUE_MAX_BLOCK = 64;
// *bs = the block size when L is the factor of a forest of equal dense blocks on consecutive columns, else 0
int clique_factor_block_size(const Csc *L, int32_t *bs);

}  // namespace csx
#endif
