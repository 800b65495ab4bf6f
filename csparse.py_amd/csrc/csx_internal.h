// Internal declarations shared by the libcsx translation units (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/csx.h"

namespace csx {

// Experiment switches exist only in the -DCSX_ABLATION build (libcsx_ablation.so, build.py --ablation): in the
// shipped library no environment variable can change which kernel runs or what it computes.
inline const char *ablation_env(const char *name) {
#ifdef CSX_ABLATION
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}


void set_error(const char *fmt, ...);

#define CSX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            csx::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
            return CSX_ERUNTIME;                                                           \
        }                                                                                  \
    } while (0)

#define CSX_TRY(call)            \
    do {                         \
        int _s = (call);         \
        if (_s != CSX_OK) return _s; \
    } while (0)

#define CSX_LAUNCH_CHECK() CSX_HIP(hipGetLastError())

enum Kind : int { K_FREE = 0, K_CSC, K_VEC, K_IVEC, K_TRIPLAN, K_CHOLPLAN, K_SHARDPLAN };

struct Csc;

// Row-gather layout used by the wave/exact SpMV and the triangular solves:
// for row r, entries [ptr[r], ptr[r+1]) hold (idx, val) in the order the
// reference updates that row.
struct Gather {
    int32_t rows = 0;
    int32_t *ptr = nullptr;
    int32_t *idx = nullptr;
    double *val = nullptr;
};

// LDS-resident SpMV plan (csx_gaxpy_tiled.hip): entries regrouped into (row block, column slab)
// tiles, row-block-major, column order kept inside a tile, stored in interleaved groups of 256.
struct TiledPlan {
    int32_t row_block = 0;         // rows per LDS tile (one tile per workgroup)
    int32_t nrb = 0;               // number of row blocks
    int32_t nslab = 0;             // number of column slabs
    int32_t slab_cols = 0;         // columns per slab
    int32_t *tile_ptr = nullptr;   // [nrb + 1] first group of every row block
    int32_t *tile_len = nullptr;   // [ngroups] group info: (slab << 9) | entries in the group
    uint32_t *tile_key = nullptr;  // packed (local col << rb_bits) | local row; null when the 3-byte keys are in use
    double *tile_val = nullptr;
    int rb_bits = 0;
    // 3-byte keys (row | column offset << 15), used when every run of 64 column-sorted entries spans < 512 columns:
    uint8_t *tile_key24 = nullptr; // [ngroups * 768]
    uint32_t *tile_base = nullptr; // [ngroups * 4] slab-local column of the first entry of each 64-entry run
    // launch shape (waves per workgroup x groups per wave and step), picked when the plan is built by timing the
    // candidates on this device (gaxpy_tiled_prepare); -1: the default shape
    int shape = -1;
    float shape_ms[4] = {0, 0, 0, 0};
};

// Order in which the reflections of a matrix of Householder vectors can be applied in parallel (csx_qr.hip):
// reflections of one level touch disjoint rows.
struct HouseLevels {
    int32_t nlevels = 0;
    std::vector<int32_t> ptr;      // host: [nlevels + 1] into cols
    int32_t *cols = nullptr;       // device: columns ordered by level, ascending inside a level
};

struct CliqueForest;   // csx_cholclique.h

struct Csc {
    int32_t m = 0, n = 0, nnz = 0;
    int32_t *p = nullptr;
    int32_t *i = nullptr;
    double *x = nullptr;  // nullptr: pattern only
    bool owns = true;
    bool trusted = true;  // structure made or checked by the library; false for csx_csc_wrap until csc_validate has passed
    // cached plans (built on demand, freed with the matrix)
    Gather *rows = nullptr;   // stable transpose = rows of A in ascending column order
    TiledPlan *tiled = nullptr;
    HouseLevels *house = nullptr;   // csx_happly's level schedule (pattern only; dropped by csx_csc_invalidate too)
    CliqueForest *clique = nullptr; // csx_schol's finding "a forest of cliques on consecutive columns" (tree, counts, block list on the
                                    // device), kept for the csx_chol that follows; pattern only, dropped by csx_csc_invalidate too
    bool rows_pending = false;      // i == nullptr ON PURPOSE: every column holds the consecutive rows j, j + 1, ... (the factor of a
                                    // forest of cliques, csx_cholsol_factor), so i[] follows from p[] alone and is written by
                                    // csc_fill_rows the first time a handle to the matrix is resolved (csc() below)
};

struct Vec {
    int64_t len = 0;
    void *d = nullptr;
    bool owns = true;
};

struct TriPlan;   // csx_trisolve.hip
struct CholPlan;  // csx_chol.hip
struct ShardPlan; // csx_comm.hip
struct SnPlan;    // csx_snsolve.hip

struct Object {
    Kind kind = K_FREE;
    void *ptr = nullptr;
    uint32_t gen = 0;   // bumped on every reuse of the slot (upper half of the handle)
};

// Kernel-selection overrides for TESTS of the kernels a plan would not pick by itself (csx_set_option).  Every
// setting computes correct results; none is read from the environment.
struct Options {
    int chol_band = 1;                // cs_chol: register-window kernel for chain-like banded factors
    int chol_wband = 1;               // cs_chol: blocked dense-band kernels for chain-like factors: 0 never, 1 for half-widths
                                      // above 80 (below, the register-window kernel), 2 whenever the tree is chain-like
    int chol_wband_nb = 16;           // ... columns per step (16 or 32)
    int chol_supernodes = 1;      // cs_chol: fundamental supernodes of >= 8 columns factored as dense trapezoids in place
    int chol_dense_trees = 1;     // cs_chol: LDS dense-block kernel for trees that are dense blocks
    int chol_forest = 1;          // ... and forests of small SPARSE trees on consecutive columns (blocks of <= 64 columns closed under
                                  // their upper entries): symbolic analysis on 64-bit row masks in registers, the block kernel with
                                  // a compacted store (needs chol.clique)
    int chol_exact = 1;           // cs_chol on forests of dense blocks: 1 = the reference's operations in the reference's order, L.x
                                  // bit-identical (default); 0 = fused multiply-adds and refined reciprocal square roots in the block
                                  // kernel: L.x equal to rounding, 1.5x the rate (opt-in; BASELINE grants x[] 1e-10)
    int chol_clique = 1;          // cs_schol / cs_chol / cholsol plan: forests of cliques on consecutive columns recognised from
                                  // A (or L) itself and handled without the general pattern machine (csx_cholclique.hip)
    int cholsol_dense_blocks = 1; // cholsol: dense-block kernels (false: the fused per-tree kernel)
    int spgemm_one_pass = 1;      // cs_multiply: one-walk LDS hash kernel (false: the two-pass kernel)
    int tri_chain_walker = 1;     // tri-solve: blocked chain walker for runs of narrow levels
    int tri_components = 1;       // tri-solve: one wave per small connected component (false: level sets)
    int tri_push = 1;             // tri-solve: component kernels in column-push form for L / U with few RHS
    int tri_columns = 1;          // tri-solve: small chain-like systems by the column loop, x in LDS
    int gaxpy_keys24 = 1;         // tiled cs_gaxpy plan: 3-byte keys when the matrix allows them
    int gaxpy_tune_shape = 0;    // tiled cs_gaxpy plan: time the launch shapes when the plan is built and keep the fastest
    int tri_row_waves = 1;        // level-scheduled solves: a wave per row for few right-hand sides and long rows
    int tri_levels_where = 0;         // level analysis: 0 = device for big factors, host for small; 1 = host; 2 = device
    int sort_short_keys = 1;          // cs_transpose: 16-bit keys between the radix passes where the matrix allows (0: always 32-bit)
    int tri_host_chains = 0;          // ONE host right-hand side on a chain-like factor (levels > n / 4, < 5e7 entries): 1 = the reference's
                                      // loop on the host (same bits), a case the device loses 10 - 40x; 0 (default) = the device, always
    int tri_graph = 2;                // supernodal solves: replay the launches of a solve as a hipGraph while the block of right-hand
                                      // sides stays in place: 0 never, 1 always, 2 when a solve is more than 256 launches and the
                                      // block has been the block of the two solves before it too (the third consecutive solve captures)
    int tri_supernodes = 1;           // cholsol: supernodal forward / backward solves on factors with supernodes (0 never, 1 yes,
                                      // 2 yes but the triangles by substitution out of LDS instead of on the matrix cores)
    int spgemm_ordered = 0;           // cs_multiply: sum every entry's products in the reference's order (bit-identical x)
    int spgemm_chunks = 1;            // cs_multiply: column chunks whose compaction overlaps the next chunk's hashing on a second stream
                                      // (1 = off, the default: measured slower, profiles/r03_ablation.md section 2)
    int cholsol_exact_variant = 0;    // exact dense-block cholsol: 0 = the measured choice per block size, 1 - 6 force a variant (tests, ablation)
    int lu_etree = 0;                 // cs_lu of one connected matrix on the device, columns scheduled by the column etree:
                                      // 0 never (the default since round 4: measured at best a tie with one host core, on the
                                      // shape it was made for -- profiles/r04_ablation.md), 1 for shallow trees with short
                                      // columns, 2 always (tests: L, U, pinv bit-identical to the host loop and the CPU restatement of the reference loop)
};

struct Context {
    Options opt;
    bool ready = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;      // a second stream for copies that run beside a kernel of `stream` (csx_chol's check of S); made
                                     // and used once in csx_init: a stream's first copy from pageable memory costs tens of ms
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cus = 0;
    std::vector<Object> objects;  // handle = generation << 32 | index + 1 (csx_core.hip: put / get)
};

Context &ctx();
int require_ready();
csx_handle_t put(Kind k, void *ptr);
void *get(csx_handle_t h, Kind k);
int csc_fill_rows(Csc *A);   // csx_cholclique.hip: the row indices of a matrix with rows_pending
inline Csc *csc(csx_handle_t h) {
    Csc *A = (Csc *)get(h, K_CSC);
    if (A && A->rows_pending && csc_fill_rows(A) != CSX_OK) return nullptr;
    return A;
}
inline Vec *vec(csx_handle_t h) { return (Vec *)get(h, K_VEC); }
inline Vec *ivec(csx_handle_t h) { return (Vec *)get(h, K_IVEC); }

// device allocation helpers (bytes may be 0 -> still returns a valid pointer)
int dmalloc(void **p, size_t bytes);
template <class T>
inline int dalloc(T **p, size_t count) { return dmalloc((void **)p, count * sizeof(T)); }
void dfree(void *p);
void pool_trim();                               // release every idle block to the driver
void pool_stats(size_t *cached, size_t *live);  // bytes idle in the cache / handed out
void pool_set_limit(size_t bytes);              // cap of the cache (0: the default quarter of the device)

void free_gather(Gather *g);
void free_clique_cache(CliqueForest *F);   // csx_cholclique.hip
void free_tiled(TiledPlan *t);
void free_csc(Csc *A);
void free_triplan(TriPlan *t);
void free_cholplan(CholPlan *t);
void free_shardplan(ShardPlan *t);
void free_snplan(SnPlan *t);
// csx_snsolve.hip: supernodal schedule of a Cholesky-shaped factor for the rounding-equal order of a cholsol plan
int sn_build(const Csc *L, const int32_t *parent, const int32_t *Lp_h, const int32_t *Gp_h, const int32_t *Gp, const int32_t *Gi,
             const double *Gx, int32_t col_levels, SnPlan **out);
int sn_solve(SnPlan *P, bool forward, const int32_t *Gp, const int32_t *Gi, const double *Gx, const double *Gd, const Csc *L,
             double *X, int32_t nrhs);
int64_t sn_generation(const SnPlan *P);   // changes whenever sn_prepare moved the work space
int sn_prepare(SnPlan *P, int32_t nrhs);   // work space of a solve with nrhs right-hand sides (sn_solve calls it; a capture calls it first)
void sn_info(const SnPlan *P, int32_t *nsn, int32_t *levels, int32_t *max_w);
void sn_info2(const SnPlan *P, int32_t *matrix_cores, double *growth);
bool sn_usable(const SnPlan *P);   // with the options in force (a plan with relaxed supernodes needs the matrix-core triangles)

// Device temporaries of a host function with several exits: freed when the guard leaves scope.
struct DevScope {
    std::vector<void *> held;
    ~DevScope() {
        for (void *p : held) dfree(p);
    }
    template <class T>
    int alloc(T **p, size_t count) {
        const int st = dalloc(p, count);
        if (st == CSX_OK) held.push_back((void *)*p);
        return st;
    }
};

// ---- device primitives (csx_scan.hip, csx_sort.hip) ----
// out[k] = sum(in[0..k-1]) for k in [0, n]; out has n+1 slots; in may alias out
// (then the last slot is written too).  total (host) optional.
int scan_exclusive_i32(const int32_t *in, int32_t *out, int64_t n, int64_t *total_host);
// col[p] = j for p in [Ap[j], Ap[j+1])
int expand_columns(const int32_t *Ap, int32_t n, int32_t nnz, int32_t *col);
// Stable sort of `count` records by key in [0, key_limit).  a = 32-bit payload,
// v = 64-bit payload (nullptr: none).  Inputs are not modified; outputs may not
// alias inputs.  out_key may be nullptr.
int stable_sort_by_key(const uint32_t *key, const uint32_t *a, const double *v, int64_t count,
                       uint32_t key_limit, uint32_t *out_key, uint32_t *out_a, double *out_v);
// The same sort with what cs_transpose adds to it (both optional): expand_ptr -- the 32-bit payload is not read but
// is the column of the record's position under these column pointers (expand_n columns); out_ptr -- the segment
// starts of the sorted keys, out_ptr[r] = first slot with key >= r for r in [0, nkeys], written by the last pass
// itself (then out_key may be null and no pass over the sorted keys is needed).
struct SortExtra {
    const int32_t *expand_ptr;
    int32_t expand_n;
    int32_t *out_ptr;
    int32_t nkeys;
};
int stable_sort_by_key_ex(const uint32_t *key, const uint32_t *a, const double *v, int64_t count, uint32_t key_limit,
                          uint32_t *out_key, uint32_t *out_a, double *out_v, const SortExtra *ex);
// p[r] = min(p[r], ..., p[n-1]) in place (reverse running minimum)
int suffix_min_i32(int32_t *p, int64_t n);
// ptr[r] = first position q with sorted_key[q] >= r, r in [0, nkeys]; ptr has nkeys+1 slots
int boundaries_from_sorted(const uint32_t *sorted_key, int64_t count, int32_t nkeys, int32_t *ptr);

// ---- building blocks shared across files ----
int build_row_gather(Csc *A);   // fills A->rows (values required)
// Structure check of a matrix whose arrays the library did not make (csx_csc_wrap): p non-decreasing from 0 to nnz,
// every row index in [0, m), in one device pass.  CSX_EINVAL (IndexError in Python) otherwise; remembered in A->trusted.
int csc_validate(Csc *A);
int transpose_device(const Csc *A, bool values, Csc *C);  // C fields allocated here
int gaxpy_device(Csc *A, const double *x, double *y, int mode);                 // csx_gaxpy.hip: y += A x, raw pointers
int gaxpy_prepare_device(Csc *A, int mode);                                     // ... the plan `mode` needs, cached on A
int col_block_device(const Csc *A, int32_t first, int32_t count, Csc *C);      // csx_assemble.hip: columns [first, first + count)

// Workgroup barrier that orders LDS only.  __syncthreads() carries a fence over global memory as well: it waits for
// every global load the wave has in flight (s_waitcnt vmcnt(0)), which puts the latency of software-pipelined loads
// back on the critical path of a kernel that synchronises once or twice per step.  Use where only LDS is shared
// across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sizes
constexpr int WAVE = 64;

}  // namespace csx
