/* csx.h -- C ABI of libcsx.so, the MI355X (gfx950) implementation of the
 * CSparse.py sparse-direct hot path.
 *
 * The reference (rwl/CSparse.py) is one pure-Python module with no FFI: its
 * "plugin interface" for this path is the module namespace (cs_gaxpy, cs_multiply,
 * cs_transpose, cs_lsolve ... called as `csparse.cs_gaxpy(A, x, y)`,
 * csparse_test.py:24,146,262).  This header is the C boundary the drop-in module
 * csparse.py_amd/csparse.py binds with ctypes; every compute entry point names the
 * reference function (csparse.py:LINE) whose loop it replaces.  INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain C types only.  Indices are int32_t, values are double
 *     (IEEE binary64, what a Python float is); nnz < 2^31.
 *   - Host buffers belong to the caller.  Device buffers belong to the library,
 *     behind opaque 64-bit handles, unless created with a *_wrap call (then the
 *     caller keeps ownership of the device memory: torch tensors can be passed
 *     as raw device pointers this way).
 *   - Dense right-hand-side blocks are n-by-k, row-major (element (i, r) at
 *     i*k + r): the k values of one matrix row are contiguous.
 *   - Every function returns a status.  No exceptions, no exit().  On
 *     CSX_ERUNTIME the message is in csx_last_error().
 *   - One context per process (one process per GPU).  Calls are serialised on
 *     the context's HIP stream; results are complete after the call returns only
 *     for functions that copy to host, otherwise after csx_sync().
 */
#ifndef CSX_H
#define CSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t csx_handle_t;

enum {
    CSX_OK = 0,
    CSX_EINVAL = 1,     /* bad argument: the Python layer returns False / None / -1 */
    CSX_EZEROPIVOT = 2, /* zero diagonal in a triangular solve: Python raises ZeroDivisionError */
    CSX_ENOTSPD = 3,    /* non-positive pivot in cs_chol: Python returns None (csparse.py:612) */
    CSX_ERUNTIME = 4    /* HIP runtime failure; see csx_last_error() */
};

/* triangular-solve kinds (csparse.py:1330, :1348, :2368, :2460) */
enum { CSX_TRI_L = 0, CSX_TRI_LT = 1, CSX_TRI_U = 2, CSX_TRI_UT = 3 };

/* cs_gaxpy execution modes */
enum {
    CSX_GAXPY_AUTO = 0,  /* best available plan for the matrix */
    CSX_GAXPY_EXACT = 1, /* reference summation order, no FMA: bit-identical to csparse.py:1211-1212 */
    CSX_GAXPY_WAVE = 2,  /* one wavefront (or sub-wave group) per row of the cached transpose */
    CSX_GAXPY_TILED = 3, /* LDS-tiled plan for matrices without locality (G-rand) */
    CSX_GAXPY_ATOMIC = 4 /* direct CSC scatter with fp64 atomics, no plan */
};

/* ---- context ---------------------------------------------------------- */
int csx_init(int device);                 /* idempotent; selects the HIP device, creates the stream */
int csx_finalize(void);                   /* frees every handle and the stream */
const char *csx_last_error(void);
int csx_sync(void);                       /* wait for the context's stream */
int csx_set_stream(void *hip_stream);     /* run on a caller-provided hipStream_t (NULL: own stream) */
int csx_device_info(char *name, int name_cap, int *compute_units, int64_t *hbm_bytes);
/* Device memory is served by a caching allocator (freed blocks are reused instead of returned to the
 * driver; cap = 1/4 of the device, CSX_POOL_LIMIT_MB / CSX_NO_POOL=1 override).  csx_mem_trim returns
 * every idle block to the driver; csx_mem_info reports idle bytes, bytes in use, and hipMemGetInfo's free.
 * Reuse is ordered on the context's stream: a block whose pointer was exported (csx_vec_ptr, csx_csc_ptrs)
 * must not be freed while work on another stream still uses it (csx_free waits for the context's stream only). */
int csx_mem_trim(void);
int csx_mem_info(int64_t *cached_bytes, int64_t *live_bytes, int64_t *device_free_bytes);
/* Kernel-selection overrides, for tests that must reach a kernel the planner would not pick for a given input.
 * Every setting computes correct results; no environment variable changes which kernel runs or what it computes
 * (the environment is read for the allocator's cap above and for CSX_CHOL_TIMING=1, which prints cs_chol's phase
 * times to stderr).  Names (default 1): "chol.dense_trees", "chol.band", "cholsol.dense_blocks", "spgemm.one_pass",
 * "tri.chain_walker", "tri.components", "tri.columns", "tri.push", "tri.row_waves", "gaxpy.keys24"; "gaxpy.tune_shape" (default 0); "tri.levels_where" (default 0: level
 * analysis of a triangular plan on the device for big factors and on the host for small ones; 1 = host, 2 = device);
 * "chol.wband" (blocked dense-band cs_chol for chain-like factors: default 1 = for half-widths above 80, 0 = never,
 * 2 = whenever the tree is chain-like) and "chol.wband_nb" (columns per step: 16 (default) or 32; negative: two
 * launches per step instead of one); "chol.supernodes" (default 1); "pool.limit_mb" (cap of the device-memory cache in MB,
 * 0 = the default quarter of the device).  Round 3: "tri.supernodes" (supernodal schedule of a cholsol plan in the
 * rounding-equal order: 1 = yes, triangles on the matrix cores where their guard allows (default); 2 = yes, triangles by
 * substitution out of LDS, no relaxed supernodes; 0 = never), "tri.graph" (the launches of a supernodal solve captured
 * into a hipGraph and replayed while the block of right-hand sides stays in place: 2 (default) = when a solve is more than
 * 256 launches and the block has been the block of the two solves before it as well (round 5; round 4 captured on the second
 * solve of a block, which cost more than it saved), 1 = always, 0 = never),
 * "cholsol.exact_variant" (the exact dense-block kernel: 0 = default = 5: L values by DPP row broadcast, one term in four
 * by an LDS broadcast read; 6: by DPP only; 1 - 4: the LDS-broadcast forms), "spgemm.ordered" (default 0; 1 = cs_multiply
 * sums every entry's products in the reference's order: bit-identical values, about twenty times the time),
 * "spgemm.chunks" (default 1; >= 2: hash and compaction of column chunks on two streams -- measured slower),
 * "sort.short_keys" (default 1: a transpose with values carries 16-bit keys between its radix passes where the matrix allows),
 * "lu.etree" (cs_lu inside one connected matrix by levels of the column elimination tree: 0 = never (default since round
 * 4: at best a tie with one host core), 1 = shallow trees with short columns, 2 = always).  Round 4:
 * "chol.clique" (default 1: csx_schol / csx_chol / csx_cholsol_plan recognise forests of cliques on consecutive columns
 * -- block-diagonal matrices with dense blocks -- from the matrix itself and skip the general pattern machine; 0 = the
 * general path), "chol.forest" (default 1: where that rule fails, csx_schol / csx_chol look for blocks of <= 64 consecutive
 * columns closed under their upper entries -- forests of small sparse trees -- and analyse / factor a block in one wave;
 * 0 = the general path for them).  Round 5: "chol.exact" (default 1: cs_chol's block kernel keeps the reference's operations and
 * their order, L.x bit-identical; 0, opt-in: fused multiply-adds and refined reciprocal square roots in that kernel -- forests of
 * dense blocks only -- L.x equal to rounding; equal blocks of 16 / 32 / 48 / 64 columns are then factored on the matrix cores, 1.8x
 * the rate), "tri.host_chains" (default 0; 1: csx_tri_solve_list / csx_cholsol_solve_list take one host right-hand side on a
 * chain-like factor to the host).  Unknown name: CSX_EINVAL. */
int csx_set_option(const char *name, int value);
int csx_get_option(const char *name, int *value);   /* the value in force (after csx_set_option's normalisation) */
int csx_timer_start(void);                /* hipEvent on the context's stream */
int csx_timer_stop(double *ms);           /* second hipEvent, synchronises, elapsed ms */

/* ---- CSC matrices: the reference's `cs` object with nz == -1 (csparse.py:37-54) ---- */
int csx_csc_upload(int32_t m, int32_t n, const int32_t *p, const int32_t *i, const double *x /* or NULL */,
                   csx_handle_t *out);
int csx_csc_alloc(int32_t m, int32_t n, int32_t nnz, int values, csx_handle_t *out);
int csx_csc_wrap(int32_t m, int32_t n, int32_t nnz, void *d_p, void *d_i, void *d_x /* or NULL */,
                 csx_handle_t *out);
int csx_csc_info(csx_handle_t A, int32_t *m, int32_t *n, int32_t *nnz, int *has_values);
int csx_csc_download(csx_handle_t A, int32_t *p, int32_t *i, double *x /* or NULL */);
int csx_csc_ptrs(csx_handle_t A, void **d_p, void **d_i, void **d_x);
/* csx_gaxpy caches plans on the matrix (a row-major copy, the LDS-tiled regrouping) that hold COPIES of its
 * values and structure; csx_schol leaves its finding there when the matrix is a forest of cliques (tree, counts, block list: pattern
 * only), for the csx_chol that follows.  Arrays handed out by csx_csc_ptrs, or wrapped by csx_csc_wrap, must not be changed
 * in place without telling the library: call csx_csc_invalidate afterwards (drops the cached plans; the next
 * csx_gaxpy rebuilds them).  Plans that are handles of their own (csx_tri_analyse, csx_cholsol_plan) also copy
 * the values they need: rebuild them after a change. */
int csx_csc_invalidate(csx_handle_t A);
int csx_free(csx_handle_t h);             /* any handle kind; handles carry a generation, a stale one is CSX_EINVAL */

/* ---- dense vectors / row-major blocks (float64) and index vectors (int32) ---- */
int csx_vec_alloc(int64_t len, csx_handle_t *out);          /* zero-filled */
int csx_vec_upload(const double *src, int64_t len, csx_handle_t *out);
int csx_vec_wrap(void *d_ptr, int64_t len, csx_handle_t *out);
int csx_vec_download(csx_handle_t v, double *dst, int64_t len);
int csx_vec_write(csx_handle_t v, const double *src, int64_t len);
int csx_vec_fill(csx_handle_t v, double value);
int csx_vec_copy(csx_handle_t src, csx_handle_t dst);
int csx_vec_ptr(csx_handle_t v, void **d_ptr, int64_t *len);
int csx_ivec_upload(const int32_t *src, int64_t len, csx_handle_t *out);
int csx_ivec_download(csx_handle_t v, int32_t *dst, int64_t len);

/* ---- hot path --------------------------------------------------------- */

/* cs_gaxpy, csparse.py:1199-1213: y += A x.  x has >= n, y >= m entries. */
int csx_gaxpy(csx_handle_t A, csx_handle_t x, csx_handle_t y, int mode);
/* Build (and cache on A) the plan `mode` needs, outside any timed region. */
int csx_gaxpy_prepare(csx_handle_t A, int mode);
/* Which plans the matrix holds: the row-major copy (EXACT / WAVE), the LDS-tiled regrouping (TILED), and the bytes
 * per key of the latter: 4 (column, row packed) or 3 (row + 9-bit column offset inside a run of 64 column-sorted
 * entries; chosen when every run is narrower than 512 columns); 0 without a tiled plan.  Any pointer may be NULL. */
int csx_gaxpy_plan_info(csx_handle_t A, int *has_rows, int *has_tiled, int *key_bytes);
/* With csx_set_option("gaxpy.tune_shape", 1) (default 0) the tiled plan times its launch shapes (waves per workgroup
 * x groups per wave and step: same kernel, plan and results) on the device when it is built and keeps the fastest:
 * *shape = 0 (4 x 5), 1 (2 x 10), 2 (8 x 4), 3 (2 x 8), or -1 (not timed: 4 x 5); ms4[0..3] = the candidates' ms per
 * pass (0 when not timed).  CSX_EINVAL without a tiled plan. */
int csx_gaxpy_plan_shape(csx_handle_t A, int *shape, double *ms4);
/* One-shot form for host arrays (the reference's list signature): y[0..m) += A x in the reference's
 * summation order (bit-identical), nothing left on the device.  x (values) must be present. */
int csx_gaxpy_host(int32_t m, int32_t n, const int32_t *p, const int32_t *i, const double *x, const double *xv,
                   double *yv);

/* cs_transpose, csparse.py:2292-2315: stable counting sort by row. */
int csx_transpose(csx_handle_t A, int values, csx_handle_t *out);

/* cs_cumsum, csparse.py:767-784, on int32 device vectors: p[0..n] = exclusive
 * scan of c[0..n-1], c overwritten with p[0..n-1]; *total = sum. */
int csx_cumsum(csx_handle_t p, csx_handle_t c, int64_t n, int64_t *total);

/* cs_multiply (+ cs_scatter), csparse.py:1608-1642, :1961-1989: C = A*B with
 * each column of C in first-touch order; pattern only if A or B has no values.
 * p[] and i[] are exactly the reference's.  x[]: the products of an entry are summed by LDS / memory atomics in the
 * order they arrive, not in the reference's order, so x[] equals the reference's to rounding (tests: 1e-10 relative to
 * the sum of |products|) and may differ in the last bits from one run to the next -- the one result of this library
 * that is not reproducible bit for bit.  The same holds for the sums cs_dupl and cs_add form from duplicate entries
 * (csx_dupl, csx_add); entries without duplicates come out exact. */
int csx_multiply(csx_handle_t A, csx_handle_t B, csx_handle_t *out);

/* cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve, csparse.py:1330-1365,
 * :2368-2385, :2460-2475.  Analyse once (level sets, gather layout in the
 * reference's update order, zero-pivot check), then solve any number of
 * right-hand sides in place: X is n-by-nrhs, row-major. */
int csx_tri_analyse(csx_handle_t T, int kind, csx_handle_t *plan);
int csx_tri_info(csx_handle_t plan, int32_t *n, int32_t *levels, int32_t *sequential);
int csx_tri_solve(csx_handle_t plan, csx_handle_t X, int32_t nrhs);
/* The order of a plan's solves.  exact = 1 (the default of every plan): the reference's operations in the reference's order,
 * bit-identical to cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve (csparse.py:1330-1365, :2368-2385, :2460-2475).  exact = 0
 * (round 5): equal to rounding (x[] within 1e-10).  It changes the solve of a factor that falls into many small independent
 * components of at most 80 rows with more than 8 right-hand sides: every component is made dense in sweep order, padded to a
 * multiple of 16 and solved as a blocked substitution on the matrix cores (16 x 16 tiles, explicit inverses of the diagonal
 * tiles, built at the first such solve) -- unless || |inv(T_ii)| |T_ii| ||_inf of a diagonal tile exceeds 1e3, then the exact
 * kernels stay.  Every other plan shape solves exactly in either order.  csx_tri_order_info: whether the matrix-core form is in
 * use (after the first solve in that order) and the guard's measure; either pointer may be NULL. */
/* "tri.host_chains" = 1 (csx_set_option; opt-in, default 0): ONE right-hand side in host memory on a factor whose dependency graph
 * is a chain (more than n / 4 levels, fewer than 5e7 entries) is solved by the reference's own loop on the host, on a copy of the
 * factor the plan downloads once -- the same operations in the same order, the same bits, 10 - 40x sooner than one dependent
 * subtraction per term on the device.  *taken = 0: the option is off or the factor is no chain: x untouched, call csx_tri_solve.
 * csx_cholsol_solve_list: the same for cs_cholsol's whole solve sequence (csparse.py:640-643) on b[n]. */
int csx_tri_solve_list(csx_handle_t plan, double *x, int *taken);
int csx_cholsol_solve_list(csx_handle_t plan, double *b, int *taken);
int csx_tri_set_order(csx_handle_t plan, int exact);
int csx_tri_order_info(csx_handle_t plan, int32_t *matrix_cores, double *growth);
/* After the first solve: the number of connected components of the dependency graph when the plan solves
 * them one wave each (many components of <= 256 rows: block-diagonal factors), 0 when it level-schedules. */
int csx_tri_components(csx_handle_t plan, int32_t *ncomp);

/* cs_ipvec / cs_pvec, csparse.py:1264-1277, :1779-1792, on n-by-nrhs blocks:
 * inverse != 0: x[p[k], :] = b[k, :] (ipvec); else x[k, :] = b[p[k], :] (pvec).
 * p == 0 is the identity permutation. */
int csx_permute_vec(csx_handle_t p, csx_handle_t b, csx_handle_t x, int32_t n, int32_t nrhs, int inverse);
/* The solve phase of cs_lusol for an n-by-nrhs block, csparse.py:1470-1473 (cs_ipvec(pinv), cs_lsolve, cs_usolve, cs_ipvec(q)):
 * b overwritten with the solutions; planL / planU: csx_tri_analyse plans of L (CSX_TRI_L) and U (CSX_TRI_U); pinv, q: int vectors
 * or 0 (identity); work: another n-by-nrhs block.  Each plan solves in the order csx_tri_set_order gave it.  When both are
 * in the rounding-equal order and both factors are forests of small components, the two permutations are fused into the two
 * sweeps (four passes over the block instead of eight): *fused = 1.  Otherwise the four steps run one after the other. */
int csx_lusol_solve(csx_handle_t planL, csx_handle_t planU, csx_handle_t pinv, csx_handle_t q, csx_handle_t b, csx_handle_t work,
                    int32_t nrhs, int *fused);

/* cs_schol (natural order), csparse.py:2051-2072: host C++ symbolic analysis of
 * the upper triangle of a host CSC pattern.  parent[n], cp[n+1]. */
int csx_schol_host(int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent, int32_t *cp);
/* cs_counts, csparse.py:703-764: column counts of chol(A) (ata = 0; A square, upper triangle used) or chol(A'A) (ata != 0;
 * A m-by-n) from the elimination tree parent[n] and its postorder post[n] (cs_etree / cs_post of the same matrix), host C++.
 * colcount[n].  CSX_EINVAL for an index out of range, a parent outside [-1, n), a post that is no permutation. */
int csx_counts_host(int32_t m, int32_t n, const int32_t *Ap, const int32_t *Ai, const int32_t *parent, const int32_t *post,
                    int ata, int32_t *colcount);
/* The same for a device-resident square matrix.  A forest of cliques on consecutive columns (block-diagonal with dense
 * blocks; recognised in one pass over A's pattern from the smallest upper row of every column): tree and counts on the
 * device, no pattern of L formed.  Otherwise: elimination tree on the device for many small components, else on the
 * host; column counts of L on the device (the row-subtree walks of csx_chol).  parent[n] and cp[n+1] are host arrays. */
int csx_schol(csx_handle_t A, int32_t *parent, int32_t *cp);

/* A fill-reducing ordering for order = 1 (Cholesky; the reference's cs_amd, csparse.py:214-556, does not
 * run): nested dissection of the graph of A + A' by breadth-first level separators, which yields the wide
 * elimination-tree levels the device kernels want.  Host arrays; perm[k] = original index of the k-th
 * row/column of P A P'. */
int csx_order_nd_host(int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *perm);

/* cs_chol numeric, csparse.py:561-619.  A: device CSC (upper triangle used);
 * parent/cp: host arrays from csx_schol / csx_schol_host; pinv: host permutation or NULL.
 * Output L (device CSC, diagonal first, rows ascending).  CSX_EINVAL when parent / cp are not A's. */
int csx_chol(csx_handle_t A, const int32_t *parent, const int32_t *cp, const int32_t *pinv,
             csx_handle_t *L);
/* What the last successful csx_chol of this process did.  *path: 1 = A's elimination forest is a set of cliques on
 * consecutive columns (block-diagonal with dense blocks of <= 64 columns; recognised from A itself, L.p / L.i follow from
 * the counts, every block factored in the registers of one wave, L.x bit-identical to csparse.py:587-617), 2 = blocks of
 * <= 64 consecutive columns closed under their upper entries whose trees are NOT cliques (symbolic elimination on 64-bit row
 * masks in one wave per block, the same block kernel with a compacted store; L.x bit-identical where the trees are chains,
 * equal to rounding where they branch -- as path 0; "chol.forest" = 0 switches this recognition off), 0 = the general
 * path (pattern of L by row-subtree walks and sorts, column kernels by tree level).  *numeric_ms: HIP-event time of the
 * numeric part (path 1: the block kernel alone; path 0: everything after the pattern of L).  Either pointer may be NULL;
 * CSX_EINVAL before the first csx_chol.  "chol.clique" = 0 (csx_set_option) forces path 0. */
int csx_chol_info(int32_t *path, double *numeric_ms);

/* The solve phase of cs_cholsol, csparse.py:640-643, for nrhs right-hand sides:
 * B (n-by-nrhs, row-major) is overwritten with the solutions.  The plan of a factor that is a forest of equal dense
 * blocks of 8 / 16 / 32 / 64 columns (recognised from L itself, also from an L that came over the wire or from the host)
 * is the block list: the default exact kernel and the matrix-core kernel read the blocks' packed columns in L.x itself (round 5);
 * any other factor gets two triangular-solve analyses and the forest partition.  The plan BORROWS L's arrays either way (lazy
 * builds read them too): free the plan before L. */
int csx_cholsol_plan(csx_handle_t L, const int32_t *pinv /* host, or NULL */, csx_handle_t *plan);
/* cs_cholsol's factor sequence in natural order -- S = cs_schol(0, A); N = cs_chol(A, S), csparse.py:636-639 -- and the solve
 * plan of csparse.py:640-643 in ONE call, the symbolic analysis never leaving the device (round 5; replaces csx_schol's 40 MB of
 * parent / cp going to the host at 5M columns, csx_chol's upload-and-compare of the same arrays, and for forests of equal dense
 * blocks the three kernels that read L back to re-arrange it).  A: device CSC, square, upper triangle used.  exact: the order the
 * plan starts in (csx_cholsol_set_order changes it later).  *L: the factor (device CSC, diagonal first, rows ascending; for a forest
 * of equal dense blocks factored with exact = 0 the row indices are written when a handle to L is first used, its values and
 * column pointers at once); *plan: the solve plan, which borrows L's arrays -- free the plan before L.  CSX_ENOTSPD as csx_chol.
 * The analysis it implies is cs_schol(0, A)'s: S.cp = L.p, S.parent[j] = the first row below the diagonal of column j of L. */
int csx_cholsol_factor(csx_handle_t A, int exact, csx_handle_t *L, csx_handle_t *plan);
/* What the last successful csx_cholsol_factor did.  *path: 3 = forest of equal dense blocks of 16 / 32 / 64 columns, rounding-equal
 * order: the block kernel wrote the matrix-core solve's operands beside L.x, no other kernel touched L; 1 = forest of cliques (block
 * kernel, then the plan cut out of L.x); 2 = forest of small sparse trees; 0 = the general path (csx_schol + csx_chol +
 * csx_cholsol_plan behind the one entry).  *analysis_ms: host clock up to the end of the symbolic analysis; *numeric_ms: HIP-event
 * time of the numeric kernel(s); *call_ms: host clock of the whole call.  Any pointer may be NULL. */
int csx_cholsol_factor_info(int32_t *path, double *analysis_ms, double *numeric_ms, double *call_ms);
/* *path, for the plan's current order (csx_cholsol_set_order): 0 = level-scheduled generic, 1 = fused per-tree
 * kernel (X tile in LDS; the only forest path of the default, exact order), 2 = dense-block FMA substitution,
 * 3 = dense blocks as a blocked TRSM on the matrix cores (fp64 MFMA; blocks of 16/32/64 whose block inverses
 * are benign); 4 = supernodal schedule of a big elimination tree (csx_cholsol_sn_info); 5 = a forest of small trees that are not
 * equal dense blocks (cliques of unequal sizes, small sparse trees, at most 80 columns each) made dense tree by tree, bucketed by
 * size class (16 / 32 / 48 / 64 / 80) and solved on the matrix cores (round 5; guard: || |inv(T_ii)| |T_ii| ||_inf <= 1e3 for every
 * diagonal tile); 2 - 5 only in the rounding-equal order */
int csx_cholsol_info(csx_handle_t plan, int32_t *path, int32_t *ntrees, int32_t *max_nodes);
int csx_cholsol_solve(csx_handle_t plan, csx_handle_t B, int32_t nrhs);
/* exact = 1 (the default of every plan): every right-hand side is solved in the reference's operation order,
 * bit-identical to cs_lsolve + cs_ltsolve on the same L, on every path (substitution kernels).
 * exact = 0: equal to the reference to rounding (1e-10 budget).  Forests of dense blocks go to the dense-block
 * kernels: blocks of 16/32/64 as a blocked TRSM on the matrix cores with explicit inverses of the diagonal
 * tiles (built by this call), unless an inverse is large (max|inv(L_ii)| max|L| > 1e3), then -- like blocks of
 * 8 -- FMA substitution with the unknowns in registers; on big elimination trees the blocked
 * chain walker may subtract a row's out-of-block terms first.  csx_cholsol_info reports the path in use;
 * csx_cholsol_growth the guard's measure (0 before the first exact = 0). */
int csx_cholsol_set_order(csx_handle_t plan, int exact);
int csx_cholsol_growth(csx_handle_t plan, double *growth);
/* The supernodal schedule of a plan in the rounding-equal order (path 4 of csx_cholsol_info; zeros when the plan has
 * none): number of supernodes outside the leaf subtrees, dependent steps of the forward solve, widest supernode,
 * whether the triangles of the supernodes are solved on the matrix cores (blocked TRSM with explicit inverses of the
 * 16 x 16 diagonal blocks) and the guard's measure for that: the largest || |inv(L_ii)| |L_ii| ||_inf over all diagonal
 * blocks; past 1e3, or with "tri.supernodes" = 2, the triangles are solved by substitution out of LDS.  Any pointer
 * may be NULL. */
int csx_cholsol_sn_info(csx_handle_t plan, int32_t *supernodes, int32_t *steps, int32_t *max_width, int32_t *matrix_cores,
                        double *growth);

/* "tri.graph": how many times this plan's supernodal solve has been captured into a hipGraph so far, and the host time of the
 * last capture + instantiate in ms (the option's default, 2, captures on the third consecutive solve of one block).  Either
 * pointer may be NULL. */
int csx_cholsol_graph_info(csx_handle_t plan, int32_t *captures, double *last_capture_ms);

/* ---- assembly and reshaping around the hot path ---------
 * Every function returns a NEW matrix handle.  p[] / i[] bit-identical to the reference's result.
 * cs_compress, csparse.py:647-673: host triplets (row Ti[k], column Tj[k], value Tx[k] or NULL) -> CSC,
 *   entries of a column in triplet order (stable counting sort by column).
 * cs_add, csparse.py:163-192: alpha*A + beta*B; column pattern in first-touch order over A(:,j) then B(:,j).
 * cs_dupl, csparse.py:1035-1065: duplicates summed into their first occurrence.
 * cs_dropzeros / cs_droptol, csparse.py:1019-1031 / 1002-1014: mode 0 keeps a != 0, mode 1 keeps |a| > tol;
 *   order preserved.  Needs values.
 * cs_permute, csparse.py:1666-1693: C = P A Q (pinv: host, length m, or NULL; q: host, length n, or NULL).
 * cs_symperm, csparse.py:2220-2255: upper triangle of P A P' (pinv host permutation or NULL), A square. */
int csx_compress(int32_t m, int32_t n, int64_t nz, const int32_t *Ti, const int32_t *Tj, const double *Tx,
                 csx_handle_t *out);
int csx_add(csx_handle_t A, csx_handle_t B, double alpha, double beta, csx_handle_t *out);
int csx_dupl(csx_handle_t A, csx_handle_t *out);
int csx_drop(csx_handle_t A, int mode, double tol, csx_handle_t *out);
int csx_permute(csx_handle_t A, const int32_t *pinv, const int32_t *q, int values, csx_handle_t *out);
int csx_symperm(csx_handle_t A, const int32_t *pinv, int values, csx_handle_t *out);
/* cs_norm, csparse.py:1647-1663: 1-norm (largest column sum of |a|), column sums in storage order (the
 * reference's bits).  Needs values. */
int csx_norm1(csx_handle_t A, double *out);
/* Columns [first, first + count) of A as a new m x count matrix: the unit of a column-sharded SpMV. */
int csx_csc_col_block(csx_handle_t A, int32_t first, int32_t count, csx_handle_t *out);

/* cs_updown, csparse.py:2318-2365: L L' + sigma w w' (sigma = +1 update, -1 downdate) applied to the device
 * factor L in place; w is given as host arrays (rows Ci[0..cnz), values Cx), parent = elimination tree (host,
 * length n).  *ok = 0 when a downdate is not positive definite (L is then changed exactly as far as the
 * reference's loop gets).  Solve plans built from L before the call hold stale values: rebuild them. */
int csx_updown(csx_handle_t L, int sigma, int32_t cnz, const int32_t *Ci, const double *Cx, const int32_t *parent,
               int *ok);

/* cs_lu, csparse.py:1370-1451 (+ cs_spsolve :2078-2113), natural column order: host C++
 * left-looking LU with threshold partial pivoting.  It produces the L (unit diagonal first)
 * and U (diagonal last) that cs_lsolve / cs_usolve consume in cs_lusol (csparse.py:1474-1477).
 * Output arrays are malloc'ed here; release each with csx_host_free.  Returns CSX_ENOTSPD for a
 * singular matrix (the reference returns None, csparse.py:1423). */
int csx_lu_host(int32_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax, double tol,
                int32_t **Lp, int32_t **Li, double **Lx, int32_t **Up, int32_t **Ui, double **Ux,
                int32_t *pinv);
void csx_host_free(void *p);

/* cs_qr numeric phase, csparse.py:1797-1870 (+ cs_house :1238-1261, cs_happly :1216-1235): sparse Householder QR on
 * the host (C++), A m-by-n with the symbolic analysis of cs_sqr(order 0, qr): parent (column etree of A'A), pinv
 * (length m2), leftmost (length m), m2 rows incl. fictitious ones.  V (m2-by-n) and R (diagonal last in every column)
 * are written into caller arrays of capacity vcap / rcap entries (cs_sqr's counts), beta has n entries.
 * csx_qr_apply_host: x <- Q' x (transpose != 0) or Q x, x of length m2. */
int csx_qr_host(int32_t m, int32_t n, int32_t m2, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                const int32_t *q /* or NULL */, const int32_t *parent, const int32_t *pinv, const int32_t *leftmost,
                int32_t vcap, int32_t rcap, int32_t *Vp, int32_t *Vi, double *Vx, int32_t *Rp, int32_t *Ri, double *Rx,
                double *beta);
int csx_qr_apply_host(int32_t n, const int32_t *Vp, const int32_t *Vi, const double *Vx, const double *beta, int transpose,
                      double *x);
/* cs_happly (csparse.py:1216-1235) applied for every reflection of V to a block of nrhs vectors on the device: X (length
 * V.m * nrhs, row-major: row r of all vectors contiguous) <- Q' X (transpose != 0: reflections 0 .. n-1) or Q X
 * (n-1 .. 0); beta: device vector of V.n entries.  One lane per vector runs the reference's loops: bit-identical to
 * cs_happly called reflection by reflection on each column. */
int csx_happly(csx_handle_t V, csx_handle_t beta, csx_handle_t X, int32_t nrhs, int transpose);
/* cs_sqr for QR, natural column order (csparse.py:2187-2217 with cs_etree of A'A :1136-1169, cs_post :1711-1742,
 * cs_counts :703-764, cs_vcount :2118-2184), on the host: parent and cp (column counts of R) have n entries, pinv m + n,
 * leftmost m; *m2 = rows of V including the fictitious ones, *vnz / *rnz = entries of V / R. */
int csx_sqr_host(int32_t m, int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent, int32_t *cp, int32_t *pinv,
                 int32_t *leftmost, int32_t *m2, int64_t *vnz, int64_t *rnz);
/* cs_qr's numeric phase on the device for a SQUARE matrix that is a batch of small independent blocks (connected
 * components of at most 96 rows, at least 64 of them), natural column order, no fictitious rows (m2 == m): one lane per
 * block runs csx_qr_host's loop.  parent / pinv / leftmost: host arrays of n entries from cs_sqr.  V, R: new device
 * matrices; beta: host, n entries.  All three bit-identical to csx_qr_host.  *done = 0 when the matrix (or the
 * analysis) is not of that shape: use csx_qr_host. */
int csx_qr_blocks(csx_handle_t A, const int32_t *parent, const int32_t *pinv, const int32_t *leftmost, int32_t m2,
                  csx_handle_t *V, csx_handle_t *R, double *beta, int *done);
/* The same factorisation on the device for a matrix that is a batch of small independent blocks (many connected
 * components of at most 96 rows, found on the device): one workgroup per block, dense in LDS, the reference's
 * pivot rule.  *done = 0 when the matrix is not of that shape (or has duplicate entries): use csx_lu_host.
 * L, U: new device matrices (unit diagonal first / diagonal last, row indices in pivot order); pinv: host, n.
 * Values equal the left-looking code's to rounding.  CSX_ENOTSPD: a singular block. */
int csx_lu_blocks(csx_handle_t A, double tol, csx_handle_t *L, csx_handle_t *U, int32_t *pinv, int *done);

/* cs_lu of ONE connected matrix on the device, natural column order: the columns are scheduled by the column elimination
 * tree (the tree of A'A, csparse.py:1136-1169): columns that are not ancestor and descendant reach disjoint rows, so a
 * level of the tree is one launch with one lane per column running csx_lu_host's loop.  L (unit diagonal first), U
 * (diagonal last), pinv bit-identical to csx_lu_host.  *done = 0 when the tree is too deep for its size (a chain: a
 * banded matrix in natural order), n < 2048 or tol <= 0: use csx_lu_host.  CSX_ENOTSPD: singular. */
int csx_lu_etree(csx_handle_t A, double tol, csx_handle_t *L, csx_handle_t *U, int32_t *pinv, int *done);

/* cs_spsolve (csparse.py:2078-2113) with cs_reach (:1939-1958) and cs_dfs (:789-829), for every column of B at once:
 * X(:,k) solves G X(:,k) = B(:,k), G n-by-n lower (lo != 0, diagonal first in every column) or upper (diagonal last)
 * triangular, B n-by-nb sparse.  pinv (host, n entries, or NULL): column pinv[j] of G belongs to node j, a negative
 * entry = no column (the partial permutation cs_lu hands over).  Column k of X lists the reach of B(:,k) in the
 * reference's xi[top..n-1] order with the solution beside it -- pattern order and values bit-identical to calling the
 * reference column by column.  values == 0: pattern only (cs_reach for every column; G and B need no values).
 * One lane per column of B runs the reference's own loops; work space n * 17 bytes per column in flight. */
int csx_spsolve(csx_handle_t G, csx_handle_t B, const int32_t *pinv /* or NULL */, int lo, int values, csx_handle_t *X);

/* ---- multi-GPU exchange: RCCL over xGMI inside the library, one process per GPU ----------------
 * The reference is one Python process (no collective call site); what shards are its sequences
 * csparse.py:640-643 (cs_ipvec, cs_lsolve, cs_ltsolve, cs_pvec: right-hand-side blocks are independent) and
 * csparse.py:1210-1212 (cs_gaxpy: column blocks give partial y vectors that are summed).
 * Set-up: rank 0 calls csx_comm_unique_id, the launcher's side channel (csparse.py_amd/shard.py: a TCP hand-shake on
 * MASTER_ADDR) carries the 128 bytes to the other ranks, every rank calls csx_comm_init(rank, world, id) after csx_init.
 * RCCL (librccl.so.1) is bound by csx_comm_init, not when libcsx is loaded.  world == 1 with id == NULL makes no RCCL
 * call at all (every exchange is a device copy); world == 1 with an id is a real RCCL communicator of one rank.
 * Every exchange works on device buffers behind handles and is enqueued on the context's stream, in order with the
 * kernels around it: no synchronisation between a kernel and the collective that ships its result.  Host results
 * (csx_comm_allreduce_host, csx_comm_bcast_host, csx_comm_barrier) are complete when the call returns. */
#define CSX_COMM_ID_BYTES 128
int csx_comm_unique_id(uint8_t *id128);
int csx_comm_init(int rank, int world, const uint8_t *id128 /* NULL: world of one, no RCCL */);
int csx_comm_finalize(void);
int csx_comm_info(int *rank, int *world, int *uses_rccl);
int csx_comm_barrier(void);                                        /* stream drained + every rank arrived */
int csx_comm_allreduce_host(double *vals, int count, int op);      /* op 0 = sum, 1 = max; count <= 1024 (timings, checks) */
int csx_comm_bcast_host(void *buf, int64_t bytes, int root);       /* small control data */
/* Factor once on `root`, ship the factor: *A (the root's matrix) arrives as a NEW matrix handle on every other rank
 * (sizes, then p, i, x: three ncclBroadcast).  The root's handle is unchanged. */
int csx_comm_bcast_csc(csx_handle_t *A, int root);
int csx_comm_bcast_vec(csx_handle_t v, int root);                  /* in place, same length on every rank */
/* `full` has world * len(out) entries; out = entries [rank len, (rank + 1) len) of the sum over ranks (ncclReduceScatter) */
int csx_comm_reduce_scatter_vec(csx_handle_t full, csx_handle_t out);
int csx_comm_allreduce_vec(csx_handle_t v);                        /* in place sum */
/* Right-hand-side blocks leave the root / solution blocks return to it: `src` / `out` (root only) hold world blocks of
 * len doubles in rank order; world - 1 point-to-point transfers in one RCCL group, the root's own block a device copy. */
int csx_comm_scatter_blocks(csx_handle_t src, csx_handle_t dst, int64_t len, int root);
int csx_comm_gather_blocks(csx_handle_t block, csx_handle_t out, int64_t len, int root);
/* Columns [c0, c0 + k) of the n x K row-major block B as a contiguous n x k block `out` (back == 0), or `out` written
 * back into those columns of B (back != 0): a rank's share of a batch of right-hand sides. */
int csx_block_cols(csx_handle_t B, int64_t n, int32_t K, int32_t c0, int32_t k, csx_handle_t out, int back);
/* ONE cs_gaxpy sharded by columns: rank r holds A_r = columns [first_r, first_r + count_r) of an m x n matrix
 * (csx_csc_col_block) and the matching slice x_r of x; y = sum_r A_r x_r, and rank r ends up with rows
 * [r chunk, min((r + 1) chunk, m)), chunk = ceil(m / world) (csx_gaxpy_sharded_rows).
 * csx_gaxpy_sharded(plan, x_r, y_mine, how): y_mine (chunk entries) += this rank's rows of the sum.
 *   how 0: one SpMV into a full-length partial y, then one ncclReduceScatter;
 *   how 1: the block is cut by rows into the world pieces y is owned in; rank r computes the piece of rank r + 1
 *          first and its own last, and a finished piece leaves for its owner over the direct link (ncclSend /
 *          ncclRecv on a second stream) while the next piece is computed; the owner adds the world partial pieces
 *          in ascending rank order, so the result has the same bits on every run.
 * The plan keeps a pointer to the block: free the plan first. */
int csx_gaxpy_sharded_plan(csx_handle_t block, csx_handle_t *plan);
int csx_gaxpy_sharded_rows(csx_handle_t plan, int32_t *first, int32_t *count);
int csx_gaxpy_sharded(csx_handle_t plan, csx_handle_t x, csx_handle_t y_mine, int how);
/* The steps of how == 1 one at a time, for a transport that is not RCCL (shard.py's host stand-in, which carries the
 * N > 1 tests on a one-GPU box): a plan cut for `world` ranks; piece q's partial sums into the plan's work buffer
 * (piece q at work + q * chunk); the buffers (recv: world - 1 arrival slots of chunk doubles, senders in ascending
 * rank order); the owner's sum of the world pieces in ascending rank order into y_mine. */
int csx_gaxpy_sharded_plan_for(csx_handle_t block, int world, csx_handle_t *plan);
int csx_gaxpy_sharded_piece(csx_handle_t plan, int q, csx_handle_t x);
int csx_gaxpy_sharded_buffers(csx_handle_t plan, void **work, void **recv, int64_t *chunk);
int csx_gaxpy_sharded_sum(csx_handle_t plan, int rank, int world, csx_handle_t y_mine);

/* ---- synthetic inputs of the benchmark configs, generated on
 * the device from a counter-based hash so host and device agree bit for bit ---- */
int csx_gen_grand(int32_t n, int32_t per_col, uint64_t seed, csx_handle_t *out);
/* G-rand exactly as SURVEY 8d words it: per_col (<= 64) distinct rows per column drawn uniformly from [0, n),
 * ascending.  csx_gen_grand above draws one row per stratum of n/per_col rows instead (same marginal
 * distribution, row-block loads almost exactly equal); bench.py reports both. */
int csx_gen_grand_uniform(int32_t n, int32_t per_col, uint64_t seed, csx_handle_t *out);
int csx_gen_gspd(int32_t nblocks, int32_t bs, uint64_t seed, csx_handle_t *out);
int csx_gen_vec(int64_t len, uint64_t seed, double lo, double hi, csx_handle_t *out);
int csx_gen_rhs(int32_t n, int32_t nrhs, int32_t col0, csx_handle_t *out);

#ifdef __cplusplus
}
#endif
#endif /* CSX_H */
