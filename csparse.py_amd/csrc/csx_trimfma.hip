// Small dependency components on the matrix cores: the ROUNDING-EQUAL order of the fused in-LDS sweeps.
//
// cs_lsolve / cs_ltsolve / cs_usolve / cs_utsolve (csparse.py:1330-1365, :2368-2385, :2460-2475) on a factor that falls into many
// small independent components -- a forest of small elimination trees (cs_cholsol on batches of independent matrices), the L and
// U of a block-diagonal cs_lu (BASELINE config 3: 1 493 components of 67 rows) -- run, in the exact order, as one wave per
// (component, 64 right-hand sides) with the X tile in LDS and the terms applied one by one in the reference's order
// (csx_sweep.h): a chain of dependent LDS round trips, 0.12 - 0.39 of the HBM roofline (profiles/r04_ablation.md section 8).
// Where the caller grants rounding (x[] within 1e-10, BASELINE.json north_star) the same solve is a DENSE triangular system per
// component: its rows in sweep-position order, zeros where the pattern has none, padded with the identity to a multiple of 16,
// solved as a blocked substitution on 16 x 16 tiles with v_mfma_f64_16x16x4_f64 -- X_i <- W_ii (X_i - sum_{j<i} T_ij X_j),
// W_ii = inv(T_ii) formed when the plan is built -- the scheme of k_cholsol_mfma (csx_chol.hip) for components of UNEQUAL sizes
// and any pattern: components are bucketed by size class (16 / 32 / 48 / 64 / 80 rows), one launch per class, X in registers,
// the fragments read once and coalesced.  A component of 67 rows costs 4.5x the flops of its sparse program and a tenth of the
// issue slots; fp64 MFMA is chosen for how it takes its operands, not for its rate.
#include <algorithm>
#include <cstring>

#include "csx_internal.h"
#include "csx_sweep.h"
#include "csx_cholclique.h"   // tile_inverse_column
#include "csx_trimfma.h"

namespace csx {

typedef double rg_f64x4 __attribute__((ext_vector_type(4)));
typedef double rg_f64x2 __attribute__((ext_vector_type(2)));

template <int NB>
constexpr int rag_frags() { return (NB * (NB - 1) / 2 + NB) * 4; }
static inline int rag_frags_of(int nb) { return (nb * (nb - 1) / 2 + nb) * 4; }

__global__ __launch_bounds__(256) void k_rag_class(const Tree *__restrict__ trees, int32_t ntrees, uint32_t *__restrict__ key,
                                                   uint32_t *__restrict__ id) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntrees) return;
    const int32_t c = trees[t].count;
    key[t] = (uint32_t)(c <= 16 ? 0 : (c - 1) >> 4);
    id[t] = (uint32_t)t;
}

// What a solve needs to know of the component in slot q of the class-ordered list, in ONE 16-byte load: {first, count, base, id}.
// base >= 0: the component's rows are the consecutive rows base, base + 1, ... of X (blocks of a block-diagonal matrix: every
// forest this path was made for) and a lane computes its rows; base = -1: it looks them up in the node list.  (The first version
// went list -> trees -> nodes -> X: three dependent memory round trips in front of the first byte of X, 3.39 ms for 128
// right-hand sides on 5M rows of cliques of 8 .. 64 columns; with the descriptor and computed rows: see profiles/r05_ablation.md.)
__global__ __launch_bounds__(256) void k_rag_desc(const int32_t *__restrict__ list, int32_t ntrees, const Tree *__restrict__ trees,
                                                  const int32_t *__restrict__ nodes, int4 *__restrict__ desc, int32_t *scattered) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ntrees) return;
    const int32_t t = list[q];
    const Tree tr = trees[t];
    int32_t base = tr.count > 0 ? nodes[tr.first] : 0;
    for (int32_t a = 1; a < tr.count; a++)
        if (nodes[tr.first + a] != base + a) {
            base = -1;
            break;
        }
    if (base < 0) scattered[min(max((tr.count + 15) / 16 - 1, 0), RAG_CLASSES - 1)] = 1;     // (its size class: k_rag_class's rule)
    desc[q] = make_int4(tr.first, tr.count, base, t);
}

// one wave per component: the dense position-order matrix in LDS, the inverses of its diagonal tiles, the fragments, the guard
template <int NB>
__global__ __launch_bounds__(64) void k_rag_frags(const int32_t *__restrict__ list, const Tree *__restrict__ trees,
                                                  const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                  const double *__restrict__ val, const double *__restrict__ diag, int reverse,
                                                  double *__restrict__ frag, unsigned long long *cond_bits) {
    constexpr int BS = 16 * NB;
    __shared__ double M[BS][BS + 1];
    __shared__ double W[NB][16][17];
    __shared__ int32_t rp[BS + 1];
    const int lane = threadIdx.x;
    const int32_t t = list[blockIdx.x], first = trees[t].first, count = trees[t].count;
    for (int e = lane; e < BS * (BS + 1); e += 64) (&M[0][0])[e] = 0.0;
    for (int sp = lane; sp <= count; sp += 64) rp[sp] = ptr[first + sp];
    __syncthreads();
    for (int sp = lane; sp < BS; sp += 64) M[sp][sp] = sp < count ? diag[first + sp] : 1.0;
    // every term of the component, 64 at a time; its sweep position by a search over the row starts.  (A row that names a source
    // twice -- cs_lu's L may, SURVEY D7 -- subtracts both products: the coefficients add.)
    const int32_t tb = rp[0], te = rp[count];
    for (int32_t q = tb + lane; q < te; q += 64) {
        int lo = 0, hi = count;                 // largest sp with rp[sp] <= q
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rp[mid] <= q) lo = mid;
            else hi = mid;
        }
        const int32_t src = idx[q] >> 6;
        const int32_t sp2 = reverse ? count - 1 - src : src;
        if (sp2 >= 0 && sp2 < lo) atomicAdd(&M[lo][sp2], val[q]);
    }
    __syncthreads();
    double wmax = 0.0;
    {
        const int blk = lane >> 4, col = lane & 15;
        for (int b0 = 0; b0 < NB; b0 += 4) {
            const int b = b0 + blk;
            if (b < NB) {
                double wcol[16];
                tile_inverse_column(&M[16 * b][16 * b], BS + 1, col, wcol);
#pragma unroll
                for (int r = 0; r < 16; r++) W[b][r][col] = wcol[r];
            }
        }
    }
    __syncthreads();
    // the guard: || |W_bb| |T_bb| ||_inf of every diagonal tile (scaling-invariant; an explicit inverse costs a relative error of
    // about eps times this)
    {
        const int r = lane & 15, cg = lane >> 4;
        for (int b = 0; b < NB; b++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 4; c++)
                for (int k = 0; k < 16; k++) s += fabs(W[b][r][k]) * fabs(M[16 * b + k][16 * b + 4 * cg + c]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            wmax = fmax(wmax, s);
        }
    }
    const int m = lane & 15, kq = lane >> 4;
    double *F = frag + (size_t)blockIdx.x * rag_frags<NB>() * 64 + lane;
    int f = 0;
    for (int i = 0; i < NB; i++) {
        for (int j = 0; j < i; j++)
            for (int sx = 0; sx < 4; sx++) F[64 * f++] = -M[16 * i + m][16 * j + 4 * sx + kq];
        for (int sx = 0; sx < 4; sx++) F[64 * f++] = W[i][m][4 * sx + kq];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, d, 64));
    if (lane == 0) {
        // (a NaN must raise the flag too: its bits are above every finite number's)
        const unsigned long long bits = (unsigned long long)__double_as_longlong(wmax != wmax ? __longlong_as_double(0x7ff8000000000000ll) : wmax);
        if (!(bits <= *(volatile unsigned long long *)cond_bits)) atomicMax(cond_bits, bits);
    }
}

// The same fragments straight from a Cholesky-shaped factor L whose components are BLOCKS of consecutive columns closed under
// their rows (a forest of cliques or of small trees as csx_cholclique.hip recognises it): component = columns [first, first + count),
// element (r, c) of its dense triangle = L(first + r, first + c).  No solve plan in between (csx_cholsol_factor: the general plan of
// such a factor -- two triangular analyses, partition, packing -- took 12 ms at 5M rows in front of a 2.8 ms solve).
template <int NB>
__global__ __launch_bounds__(64) void k_rag_frags_csc(const int32_t *__restrict__ list, const Tree *__restrict__ trees,
                                                      const int32_t *__restrict__ Lp, const int32_t *__restrict__ Li,
                                                      const double *__restrict__ Lx, double *__restrict__ frag,
                                                      unsigned long long *cond_bits) {
    constexpr int BS = 16 * NB;
    __shared__ double M[BS][BS + 1];
    __shared__ double W[NB][16][17];
    __shared__ int32_t cp[BS + 1];
    const int lane = threadIdx.x;
    const int32_t t = list[blockIdx.x], first = trees[t].first, count = trees[t].count;
    for (int e = lane; e < BS * (BS + 1); e += 64) (&M[0][0])[e] = 0.0;
    for (int c = lane; c <= count; c += 64) cp[c] = Lp[first + c];
    __syncthreads();
    for (int sp = count + lane; sp < BS; sp += 64) M[sp][sp] = 1.0;
    const int32_t tb = cp[0], te = cp[count];
    for (int32_t q = tb + lane; q < te; q += 64) {
        int lo = 0, hi = count;                 // the column of entry q: largest c with cp[c] <= q
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cp[mid] <= q) lo = mid;
            else hi = mid;
        }
        const int32_t r = Li[q] - first;
        if (r >= lo && r < count) atomicAdd(&M[r][lo], Lx[q]);
    }
    __syncthreads();
    double wmax = 0.0;
    {
        const int blk = lane >> 4, col = lane & 15;
        for (int b0 = 0; b0 < NB; b0 += 4) {
            const int b = b0 + blk;
            if (b < NB) {
                double wcol[16];
                tile_inverse_column(&M[16 * b][16 * b], BS + 1, col, wcol);
#pragma unroll
                for (int r = 0; r < 16; r++) W[b][r][col] = wcol[r];
            }
        }
    }
    __syncthreads();
    {
        const int r = lane & 15, cg = lane >> 4;
        for (int b = 0; b < NB; b++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 4; c++)
                for (int k = 0; k < 16; k++) s += fabs(W[b][r][k]) * fabs(M[16 * b + k][16 * b + 4 * cg + c]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            wmax = fmax(wmax, s);
        }
    }
    const int m = lane & 15, kq = lane >> 4;
    double *F = frag + (size_t)blockIdx.x * rag_frags<NB>() * 64 + lane;
    int f = 0;
    for (int i = 0; i < NB; i++) {
        for (int j = 0; j < i; j++)
            for (int sx = 0; sx < 4; sx++) F[64 * f++] = -M[16 * i + m][16 * j + 4 * sx + kq];
        for (int sx = 0; sx < 4; sx++) F[64 * f++] = W[i][m][4 * sx + kq];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, d, 64));
    if (lane == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(wmax != wmax ? __longlong_as_double(0x7ff8000000000000ll) : wmax);
        if (!(bits <= *(volatile unsigned long long *)cond_bits)) atomicMax(cond_bits, bits);
    }
}

// the plan's block list from the forest's block starts: component b = columns [start[b], start[b + 1]), the identity node list
__global__ __launch_bounds__(256) void k_rag_blocks(const int32_t *__restrict__ start, int32_t nblocks, int32_t n, Tree *__restrict__ trees,
                                                    int32_t *__restrict__ nodes) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nblocks) trees[q] = Tree{start[q], start[q + 1] - start[q]};
    if (q < n) nodes[q] = (int32_t)q;
}

int ragged_blocks(const int32_t *start, int32_t nblocks, int32_t n, Tree *trees, int32_t *nodes) {
    const int64_t m = std::max<int64_t>(nblocks, n);
    if (m <= 0) return CSX_OK;
    hipLaunchKernelGGL(k_rag_blocks, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx().stream, start, nblocks, n, trees, nodes);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

// One wave = one component x 64 right-hand sides.  Lane (rq, col): rows 16 i + rq + 4 r of the position order, right-hand sides
// col (+ 16 c) of the chunk -- the f64 accumulator layout, which is also the B-operand layout of k-step r: a finished tile feeds
// the next product from its registers.  Positions past the component's rows are padding: zero in X, the identity in T.
// SHARE > 0: the component's fragments go to LDS first (global_load_lds, every line once per workgroup; the backward sweep's
// transposed reads, 8 bytes out of each of sixteen lines an instruction straight from memory, then cost LDS cycles instead of cache
// lines -- what took k_cholsol_mfma from 2.37 to 2.07 ms): SHARE components per workgroup, 4 / SHARE waves (chunks of 64
// right-hand sides) per component; the host picks the SHARE whose waves-per-component divides the number of chunks, or 0 (every
// wave reads its fragments from memory, as before round 5's last week) where the copies would not leave two workgroups to a CU.
typedef __attribute__((address_space(1))) const void *rg_gptr;
typedef __attribute__((address_space(3))) void *rg_lptr;
// MODE -- how X moves (the arithmetic is the same):
//   0  anything: rows looked up (nodes, then load_rows / store_rows when given), or the buffer-resource path when the component is
//      consecutive unpermuted rows; partial chunks of right-hand sides; decided at run time, four paths in one kernel
//   1  the host vouches that every component is consecutive rows, unpermuted, and every chunk is 64 whole right-hand sides at 16-byte
//      aligned blocks: only the buffer-resource path is compiled.  (With the four ways of moving X in one kernel the paths meet in
//      register moves that wait for the loads one by one, in front of everything that could overlap with them.)
//   2  as 1, but the LOAD gathers: position p of a component takes row load_rows[base + p] of Bsrc (cs_lusol's x = P b fused into the
//      sweep over L: csparse.py:1470, cs_ipvec); one resource over all of Bsrc, a padding position's offset past its end
//   3  as 1, but the STORE scatters: position p goes to row store_rows[base + p] of Bdst (cs_lusol's b = Q x fused into the sweep over U)
//   4  whole chunks, rows LOOKED UP on both sides (nodes, then load_rows / store_rows when given: components that are not consecutive
//      rows, cs_cholsol with a fill-reducing order): one resource over each whole block, a position's two byte offsets computed once
//      and kept (a padding position's past the end) -- the look-ups of MODE 0 without its branches and without its four paths
// Bsrc / Bdst: the block read / the block written (the same block for an in-place solve).
// CT: tiles of 16 right-hand sides a wave takes (a chunk is 16 CT right-hand sides): 4, or 2 for the class of 80 rows -- 5 x 4 tiles of
// unknowns are 320 registers, one wave to a SIMD and nothing to hide a load behind; 5 x 2 leave room for three.
template <int NB, int PASSES, int SHARE, int MODE, int CT>
__global__ __launch_bounds__(256, 2) void k_rag_mfma(const int4 *__restrict__ desc, int32_t ncls,
                                                                     const int32_t *__restrict__ nodes,
                                                                     const int32_t *__restrict__ load_rows,
                                                                     const int32_t *__restrict__ store_rows,
                                                                     const double *__restrict__ frag, int reverse, const double *Bsrc,
                                                                     double *Bdst, int32_t nrhs, int32_t chunks, int32_t n_rows) {
    constexpr bool FAST = MODE != 0;
    constexpr int FSZ = rag_frags<NB>() * 64;               // doubles per component
    constexpr int WPT = SHARE ? 4 / SHARE : 1;
    __shared__ __attribute__((aligned(16))) double s_f[SHARE ? SHARE : 1][SHARE ? FSZ : 2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = SHARE ? w / WPT : 0, sub = SHARE ? w % WPT : 0;
    int32_t q, h;
    bool valid = true;
    if (SHARE) {
        const int32_t cgroups = chunks / WPT;
        const int32_t qg = (int32_t)(blockIdx.x / cgroups), cg = (int32_t)(blockIdx.x % cgroups);
        const int32_t q_raw = qg * SHARE + slot;
        valid = q_raw < ncls;                                // a wave past the last component still copies and meets the barrier
        q = valid ? q_raw : ncls - 1;
        h = cg * WPT + sub;
    } else {
        const int64_t task = (int64_t)blockIdx.x * 4 + w;
        if (task >= (int64_t)ncls * chunks) return;
        q = (int32_t)(task / chunks);
        h = (int32_t)(task % chunks);
    }
    const int4 ds = desc[q];
    const int32_t first = __builtin_amdgcn_readfirstlane(ds.x), count = __builtin_amdgcn_readfirstlane(ds.y);
    const int32_t base = __builtin_amdgcn_readfirstlane(ds.z);
    const int col = lane & 15, rq = lane >> 4;
    // X moves through a BUFFER RESOURCE when the component's rows are consecutive rows of X (base >= 0: blocks of a block-diagonal
    // matrix): scalar base = the component's first row, size = its rows, a lane's 32-bit byte offset = its position's row and
    // right-hand sides.  A padding position lies past the size: the hardware's range check returns zero for its load and drops its
    // store -- no predicate, no branch, no access.  (Tried first: a predicate per load -- the compiler gave every load its own
    // exec-masked basic block, 234 branches in the 64-row kernel: 3.31 ms for 128 right-hand sides on 5M rows of cliques of
    // 8 .. 64 columns; padding positions loading the last row and zeroing afterwards: 4.36 ms, the dummy loads cost what real ones
    // do; this form: profiles/r05_ablation.md.)  Rows that are not consecutive, or permuted (cs_cholsol with a fill-reducing
    // order), are looked up and go through predicated global accesses.
    // (MODE 0: the copy of the fragments is requested FIRST -- it depends on nothing, and behind the loads of X it waited for them:
    // the paths that load X meet in register moves that wait for every load in turn)
    if (SHARE && !FAST) {
        const double *src = frag + (size_t)q * FSZ;
#pragma unroll
        for (int k = 0; k < FSZ / 128; k++)
            if (k % WPT == sub) __builtin_amdgcn_global_load_lds((rg_gptr)(src + k * 128 + 2 * lane), (rg_lptr)(s_f[slot] + k * 128), 16, 0, 0);
    }
    const bool consecutive = FAST || (base >= 0 && !load_rows && !store_rows && (int64_t)(RAG_MAX_ROWS + 16) * nrhs * 8 < (1ll << 31));      // uniform (offsets of padding positions must not wrap back into range)
    rg_f64x4 X[NB][CT];
    bool live[CT];
    int32_t cidx[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int32_t rhs = h * (16 * CT) + 16 * c + col;
        live[c] = rhs < nrhs;
        cidx[c] = live[c] ? rhs : nrhs - 1;   // clamped: loaded, never stored
    }
    // a chunk wholly inside the block (and an even nrhs: 16-byte alignment) moves 16 bytes per lane: lane (rq, col) takes the
    // neighbours 32 c' + 2 col, + 1 of a row and gives them to column chunks 2 c' and 2 c' + 1 (which right-hand side a
    // (chunk, column) pair stands for is free)
    const bool wide = FAST || ((nrhs & 1) == 0 && h * (16 * CT) + 16 * CT <= nrhs && ((reinterpret_cast<uintptr_t>(Bsrc) | reinterpret_cast<uintptr_t>(Bdst)) & 15) == 0);   // uniform
    typedef unsigned int rg_u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int rg_u32x2 __attribute__((ext_vector_type(2)));
    // the component's rows as a resource of the block read and of the block written (MODE 2 / 3: the WHOLE block on the permuted side)
    const __amdgpu_buffer_rsrc_t rs_ld =
        (MODE == 2 || MODE == 4) ? __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Bsrc), 0, (int32_t)((uint32_t)n_rows * (uint32_t)nrhs * 8u), 0x00020000)
                  : __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Bsrc) + (consecutive ? (int64_t)base * nrhs : 0), 0, consecutive ? count * nrhs * 8 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_st =
        (MODE == 3 || MODE == 4) ? __builtin_amdgcn_make_buffer_rsrc(Bdst, 0, (int32_t)((uint32_t)n_rows * (uint32_t)nrhs * 8u), 0x00020000)
                  : __builtin_amdgcn_make_buffer_rsrc(Bdst + (consecutive ? (int64_t)base * nrhs : 0), 0, consecutive ? count * nrhs * 8 : 0, 0x00020000);
    // byte offset of position p's row inside the component (padding: past the size, also when the order is reversed)
    auto pos_off = [&](int i, int r) -> uint32_t {
        const int p = 16 * i + rq + 4 * r;
        return (uint32_t)(reverse ? count - 1 - p : p) * (uint32_t)(nrhs * 8);
    };
    // MODE 2 / 3: byte offset of the permuted row of position p in the whole block (padding: an offset the range check refuses)
    auto perm_off = [&](const int32_t *rows, int i, int r) -> uint32_t {
        const int p = 16 * i + rq + 4 * r;
        const int pc = p < count ? p : count - 1;
        const int32_t row = rows[base + (reverse ? count - 1 - pc : pc)];
        return p < count ? (uint32_t)row * (uint32_t)(nrhs * 8) : 0xc0000000u;    // (the host keeps the block under 3 GB for these modes)
    };
    auto row_of = [&](int i, int r, bool *ok) -> int32_t {       // the general case: look the row up
        const int p = 16 * i + rq + 4 * r;
        *ok = p < count;
        if (!*ok) return 0;
        return nodes[first + (reverse ? count - 1 - p : p)];
    };
    uint32_t off_ld[MODE == 4 ? NB : 1][4], off_st[MODE == 4 ? NB : 1][4];      // MODE 4: the looked-up rows' byte offsets
    if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = 16 * i + rq + 4 * r;
                const bool ok = p < count;
                const int pc = ok ? p : count - 1;
                const int32_t j = nodes[first + (reverse ? count - 1 - pc : pc)];
                const int32_t jl = load_rows ? load_rows[j] : j;
                const int32_t js = store_rows == load_rows ? jl : (store_rows ? store_rows[j] : j);
                off_ld[MODE == 4 ? i : 0][r] = ok ? (uint32_t)jl * (uint32_t)(nrhs * 8) : 0xc0000000u;
                off_st[MODE == 4 ? i : 0][r] = ok ? (uint32_t)js * (uint32_t)(nrhs * 8) : 0xc0000000u;
            }
    }
    if (MODE == 2 || MODE == 4) {
        const uint32_t coff = (uint32_t)(h * (16 * CT) + 2 * col) * 8u;
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t ro = (MODE == 4 ? off_ld[MODE == 4 ? i : 0][r] : perm_off(load_rows, i, r)) + coff;
#pragma unroll
                for (int cp = 0; cp < CT / 2; cp++) {
                    const rg_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_ld, ro, 32 * cp * 8, 2);
                    const rg_f64x2 v = __builtin_bit_cast(rg_f64x2, u);
                    X[i][2 * cp][r] = v.x;
                    X[i][2 * cp + 1][r] = v.y;
                }
            }
    } else
    if (consecutive && wide) {
        const uint32_t coff = (uint32_t)(h * (16 * CT) + 2 * col) * 8u;
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int cp = 0; cp < CT / 2; cp++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const rg_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_ld, pos_off(i, r) + coff, 32 * cp * 8, 2);   // (aux 2 = nt: X goes through once)
                    const rg_f64x2 v = __builtin_bit_cast(rg_f64x2, u);
                    X[i][2 * cp][r] = v.x;
                    X[i][2 * cp + 1][r] = v.y;
                }
    } else if (consecutive) {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int c = 0; c < CT; c++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    X[i][c][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_ld, pos_off(i, r) + (uint32_t)cidx[c] * 8u, 0, 0));
    } else if (wide) {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                bool ok;
                int32_t jr = row_of(i, r, &ok);
                if (ok && load_rows) jr = load_rows[jr];
                const int64_t ro = (int64_t)jr * nrhs;
#pragma unroll
                for (int cp = 0; cp < CT / 2; cp++) {
                    rg_f64x2 v = rg_f64x2{0.0, 0.0};
                    if (ok) v = *reinterpret_cast<const rg_f64x2 *>(Bsrc + ro + h * (16 * CT) + 32 * cp + 2 * col);
                    X[i][2 * cp][r] = v.x;
                    X[i][2 * cp + 1][r] = v.y;
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                bool ok;
                int32_t jr = row_of(i, r, &ok);
                if (ok && load_rows) jr = load_rows[jr];
                const int64_t ro = (int64_t)jr * nrhs;
#pragma unroll
                for (int c = 0; c < CT; c++) X[i][c][r] = ok ? Bsrc[ro + cidx[c]] : 0.0;
            }
    }
    if (SHARE && FAST) {   // (one way of loading X: behind its loads, where the same copy costs k_cholsol_mfma less)
        const double *src = frag + (size_t)q * FSZ;
#pragma unroll
        for (int k = 0; k < FSZ / 128; k++)
            if (k % WPT == sub) __builtin_amdgcn_global_load_lds((rg_gptr)(src + k * 128 + 2 * lane), (rg_lptr)(s_f[slot] + k * 128), 16, 0, 0);
    }
    if (SHARE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const double *Fb = SHARE ? s_f[slot] : frag + (size_t)q * FSZ;
    const double *F = Fb + lane;
    int f = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) {
#pragma unroll
        for (int j = 0; j < i; j++)
#pragma unroll
            for (int sx = 0; sx < 4; sx++) {
                const double a = F[64 * f++];
#pragma unroll
                for (int c = 0; c < CT; c++) X[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[j][c][sx], X[i][c], 0, 0, 0);
            }
        rg_f64x4 Y[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) Y[c] = rg_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sx = 0; sx < 4; sx++) {
            const double a = F[64 * f++];
#pragma unroll
            for (int c = 0; c < CT; c++) Y[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[i][c][sx], Y[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < CT; c++) X[i][c] = Y[c];
    }
    if (PASSES == 2) {
        // the transposed system, backwards: the A fragment of tile' for (lane = (m, kq), k-step sx) is element (4 sx + kq, m) of
        // the tile, which the forward layout keeps in k-step m >> 2 at lane (m & 3) * 16 + 4 sx + kq (k_cholsol_mfma)
        const double *Ft = Fb + (col >> 2) * 64 + (col & 3) * 16 + rq;
        auto tile_at = [](int a, int b) { return (a * (a + 1) / 2 + b) * 4; };
#pragma unroll
        for (int i = NB - 1; i >= 0; i--) {
#pragma unroll
            for (int j = i + 1; j < NB; j++)
#pragma unroll
                for (int sx = 0; sx < 4; sx++) {
                    const double a = Ft[(size_t)tile_at(j, i) * 64 + 4 * sx];
#pragma unroll
                    for (int c = 0; c < CT; c++) X[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[j][c][sx], X[i][c], 0, 0, 0);
                }
            rg_f64x4 Y[CT];
#pragma unroll
            for (int c = 0; c < CT; c++) Y[c] = rg_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sx = 0; sx < 4; sx++) {
                const double a = Ft[(size_t)tile_at(i, i) * 64 + 4 * sx];
#pragma unroll
                for (int c = 0; c < CT; c++) Y[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[i][c][sx], Y[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < CT; c++) X[i][c] = Y[c];
        }
    }
    if (!valid) return;
    if (MODE == 3 || MODE == 4) {
        const uint32_t coff = (uint32_t)(h * (16 * CT) + 2 * col) * 8u;
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t ro = (MODE == 4 ? off_st[MODE == 4 ? i : 0][r] : perm_off(store_rows, i, r)) + coff;
#pragma unroll
                for (int cp = 0; cp < CT / 2; cp++) {
                    rg_f64x2 v;
                    v.x = X[i][2 * cp][r];
                    v.y = X[i][2 * cp + 1][r];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rg_u32x4, v), rs_st, ro + 32u * cp * 8u, 0, 2);   // (soffset 0: see below)
                }
            }
        return;
    }
    if (consecutive && wide) {
        const uint32_t coff = (uint32_t)(h * (16 * CT) + 2 * col) * 8u;
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int cp = 0; cp < CT / 2; cp++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    rg_f64x2 v;
                    v.x = X[i][2 * cp][r];
                    v.y = X[i][2 * cp + 1][r];
                    // (the constant goes into the LANE offset, not the instruction's scalar offset: a 16-byte buffer store whose
                    // soffset is an SGPR gets no wait states from the compiler before a vector instruction overwrites its data
                    // registers -- GCNHazardRecognizer::createsVALUHazard exempts MUBUF stores with a register soffset -- and on this
                    // chip the store then took the NEW low dword of its first register in about one block in a hundred: solutions
                    // wrong in the 7th digit, in columns 56 .. 63 of a chunk only, differently from run to run)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rg_u32x4, v), rs_st, pos_off(i, r) + coff + 32u * cp * 8u, 0, 2);
                }
        return;
    }
    if (consecutive) {
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int c = 0; c < CT; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double xv = X[i][c][r];      // (a bit_cast of the vector ELEMENT expression itself reads element 0 of the vector)
                    // a right-hand side past the block: an offset the range check refuses (0xfffffff8, not 0xffffffff: the access
                    // is two dwords, each checked on its own, and the second one's offset would wrap to 3)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rg_u32x2, xv), rs_st,
                                                          live[c] ? pos_off(i, r) + (uint32_t)cidx[c] * 8u : 0xfffffff8u, 0, 0);
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            bool ok;
            int32_t jr = row_of(i, r, &ok);
            if (!ok) continue;
            if (store_rows) jr = store_rows[jr];
            const int64_t ro = (int64_t)jr * nrhs;
            if (wide) {
#pragma unroll
                for (int cp = 0; cp < CT / 2; cp++) {
                    rg_f64x2 v;
                    v.x = X[i][2 * cp][r];
                    v.y = X[i][2 * cp + 1][r];
                    *reinterpret_cast<rg_f64x2 *>(Bdst + ro + h * (16 * CT) + 32 * cp + 2 * col) = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < CT; c++)
                    if (live[c]) Bdst[ro + cidx[c]] = X[i][c][r];
            }
        }
}

void ragged_free(RaggedMfma *R) {
    if (!R) return;
    dfree(R->desc);
    dfree(R->list);
    dfree(R->frag);
    delete R;
}

// the class-ordered list, the descriptors and room for the fragments (not yet written)
static int ragged_skeleton(const Tree *trees, int32_t ntrees, const int32_t *nodes, RaggedMfma *R) {
    hipStream_t s = ctx().stream;
    R->ntrees = ntrees;
    DevScope tmp;
    uint32_t *key = nullptr, *id = nullptr, *skey = nullptr;
    int32_t *bounds = nullptr;
    CSX_TRY(tmp.alloc(&key, (size_t)ntrees));
    CSX_TRY(tmp.alloc(&id, (size_t)ntrees));
    CSX_TRY(tmp.alloc(&skey, (size_t)ntrees));
    CSX_TRY(tmp.alloc(&bounds, 2 * RAG_CLASSES + 1));         // (+ per class: "a component's rows are not consecutive")
    CSX_HIP(hipMemsetAsync(bounds + RAG_CLASSES + 1, 0, RAG_CLASSES * sizeof(int32_t), s));
    CSX_TRY(dalloc(&R->list, (size_t)ntrees));
    hipLaunchKernelGGL(k_rag_class, dim3((unsigned)((ntrees + 255) / 256)), dim3(256), 0, s, trees, ntrees, key, id);
    CSX_LAUNCH_CHECK();
    CSX_TRY(stable_sort_by_key(key, id, nullptr, ntrees, RAG_CLASSES, skey, (uint32_t *)R->list, nullptr));
    CSX_TRY(boundaries_from_sorted(skey, ntrees, RAG_CLASSES, bounds));
    CSX_TRY(dalloc((int4 **)&R->desc, (size_t)ntrees));
    hipLaunchKernelGGL(k_rag_desc, dim3((unsigned)((ntrees + 255) / 256)), dim3(256), 0, s, R->list, ntrees, trees, nodes, (int4 *)R->desc,
                       bounds + RAG_CLASSES + 1);
    CSX_LAUNCH_CHECK();
    int32_t hb[2 * RAG_CLASSES + 1];
    CSX_HIP(hipMemcpyAsync(hb, bounds, sizeof hb, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    for (int c = 0; c <= RAG_CLASSES; c++) R->cls_start[c] = hb[c];
    for (int c = 0; c < RAG_CLASSES; c++) R->cls_consecutive[c] = hb[RAG_CLASSES + 1 + c] == 0;
    size_t total = 0;
    for (int c = 0; c < RAG_CLASSES; c++) {
        R->cls_frag[c] = total;
        total += (size_t)(R->cls_start[c + 1] - R->cls_start[c]) * (size_t)rag_frags_of(c + 1) * 64;
    }
    R->cls_frag[RAG_CLASSES] = total;
    CSX_TRY(dalloc(&R->frag, total));
    return CSX_OK;
}

// frag_off[t] = where component t's fragments start (doubles from R->frag): its class's base + its slot in the class
struct RagBases {
    int32_t start[RAG_CLASSES + 1];
    int64_t frag[RAG_CLASSES + 1];
};
__global__ __launch_bounds__(256) void k_rag_fragoff(const int32_t *__restrict__ list, int32_t ntrees, RagBases b, int64_t *__restrict__ off) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ntrees) return;
    int c = 0;
    while (c + 1 < RAG_CLASSES && q >= b.start[c + 1]) c++;
    const int nb = c + 1;
    off[list[q]] = b.frag[c] + (q - b.start[c]) * (int64_t)((nb * (nb - 1) / 2 + nb) * 4 * 64);
}

int ragged_prepare_emit(const Tree *trees, int32_t ntrees, int32_t max_rows, const int32_t *nodes, RaggedMfma **out, int64_t **frag_off) {
    *out = nullptr;
    *frag_off = nullptr;
    if (ntrees <= 0 || max_rows > RAG_MAX_ROWS) return CSX_OK;
    RaggedMfma *R = new RaggedMfma();
    struct Guard {
        RaggedMfma *R;
        ~Guard() { ragged_free(R); }
    } guard{R};
    CSX_TRY(ragged_skeleton(trees, ntrees, nodes, R));
    int64_t *off = nullptr;
    CSX_TRY(dalloc(&off, (size_t)ntrees));
    RagBases b;
    for (int c = 0; c <= RAG_CLASSES; c++) {
        b.start[c] = R->cls_start[c];
        b.frag[c] = (int64_t)R->cls_frag[c];
    }
    hipLaunchKernelGGL(k_rag_fragoff, dim3((unsigned)((ntrees + 255) / 256)), dim3(256), 0, ctx().stream, R->list, ntrees, b, off);
    if (hipGetLastError() != hipSuccess) {
        dfree(off);
        return CSX_ERUNTIME;
    }
    guard.R = nullptr;
    *out = R;
    *frag_off = off;
    return CSX_OK;
}

int ragged_build(const Tree *trees, int32_t ntrees, int32_t max_rows, const int32_t *nodes, const int32_t *ptr, const int32_t *idx,
                 const double *val, const double *diag, bool reverse, RaggedMfma **out, const Csc *from_factor) {
    *out = nullptr;
    if (ntrees <= 0 || max_rows > RAG_MAX_ROWS) return CSX_OK;
    hipStream_t s = ctx().stream;
    RaggedMfma *R = new RaggedMfma();
    struct Guard {
        RaggedMfma *R;
        ~Guard() { ragged_free(R); }
    } guard{R};
    CSX_TRY(ragged_skeleton(trees, ntrees, nodes, R));
    DevScope tmp;
    unsigned long long *cond = nullptr;
    CSX_TRY(tmp.alloc(&cond, 1));
    CSX_HIP(hipMemsetAsync(cond, 0, sizeof(unsigned long long), s));
    for (int c = 0; c < RAG_CLASSES; c++) {
        const int32_t cnt = R->cls_start[c + 1] - R->cls_start[c];
        if (cnt <= 0) continue;
        const int32_t *lst = R->list + R->cls_start[c];
        double *fr = R->frag + R->cls_frag[c];
#define CSX_RF(NB)                                                                                                                   \
    if (from_factor)                                                                                                                \
        hipLaunchKernelGGL(k_rag_frags_csc<NB>, dim3((unsigned)cnt), dim3(64), 0, s, lst, trees, from_factor->p, from_factor->i,    \
                           from_factor->x, fr, cond);                                                                               \
    else                                                                                                                            \
        hipLaunchKernelGGL(k_rag_frags<NB>, dim3((unsigned)cnt), dim3(64), 0, s, lst, trees, ptr, idx, val, diag, reverse ? 1 : 0, fr, cond)
        switch (c) {
            case 0: CSX_RF(1); break;
            case 1: CSX_RF(2); break;
            case 2: CSX_RF(3); break;
            case 3: CSX_RF(4); break;
            default: CSX_RF(5); break;
        }
#undef CSX_RF
        CSX_LAUNCH_CHECK();
    }
    unsigned long long hcond = 0;
    CSX_HIP(hipMemcpyAsync(&hcond, cond, sizeof hcond, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    std::memcpy(&R->growth, &hcond, sizeof R->growth);
    guard.R = nullptr;
    *out = R;
    return CSX_OK;
}

int ragged_solve(const RaggedMfma *R, const int32_t *nodes, const int32_t *perm, bool reverse, int passes, double *X, int32_t nrhs,
                 int32_t n_rows) {
    return ragged_solve_io(R, nodes, perm, perm, reverse, passes, X, X, nrhs, n_rows);
}

int ragged_solve_io(const RaggedMfma *R, const int32_t *nodes, const int32_t *load_rows, const int32_t *store_rows, bool reverse, int passes,
                    const double *src, double *dst, int32_t nrhs, int32_t n_rows) {
    hipStream_t s = ctx().stream;
    const int rev = reverse ? 1 : 0;
    const bool whole_rhs = nrhs % 64 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                       (int64_t)(RAG_MAX_ROWS + 16) * nrhs * 8 < (1ll << 31);
    const bool small_block = (int64_t)n_rows * nrhs * 8 < 0xc0000000ll && n_rows > 0;
    for (int c = 0; c < RAG_CLASSES; c++) {
        const int32_t cnt = R->cls_start[c + 1] - R->cls_start[c];
        if (cnt <= 0) continue;
        const bool whole = whole_rhs && R->cls_consecutive[c];
        int mode = 0;
        if (whole && !load_rows && !store_rows) mode = 1;
        else if (whole && load_rows && !store_rows && passes == 1 && small_block) mode = 2;
        else if (whole && !load_rows && store_rows && passes == 1 && small_block) mode = 3;
        else if (whole_rhs && small_block) mode = 4;
        const int ct = c >= 4 ? 2 : 4;                          // tiles of 16 right-hand sides to a wave (see the kernel)
        const int32_t chunks = (nrhs + 16 * ct - 1) / (16 * ct);
        const int64_t tasks = (int64_t)cnt * chunks;
        const int4 *dsc = (const int4 *)R->desc + R->cls_start[c];
        const double *fr = R->frag + R->cls_frag[c];
        // fragments through LDS, shared by the waves of a component (see the kernel); classes of 64 / 80 rows (20 / 30 KB a component)
        // with an odd number of chunks keep the direct reads: four private copies would leave one workgroup to a CU
        int share = chunks % 4 == 0 ? 1 : chunks % 2 == 0 ? 2 : 4;
        if (share == 4 && c >= 3) share = 0;
        const dim3 grid(share ? (unsigned)(((int64_t)cnt + share - 1) / share * (chunks / (4 / share))) : (unsigned)((tasks + 3) / 4));
#define CSX_RK(NB, PS, SH, MD)                                                                                                          \
    hipLaunchKernelGGL((k_rag_mfma<NB, PS, SH, MD, (NB >= 5 ? 2 : 4)>), grid, dim3(256), 0, s, dsc, cnt, nodes, load_rows, store_rows, fr, rev, src, dst, \
                       nrhs, chunks, n_rows)
#define CSX_RS1(NB, SH)                              \
    if (mode == 1) CSX_RK(NB, 1, SH, 1);             \
    else if (mode == 2) CSX_RK(NB, 1, SH, 2);        \
    else if (mode == 3) CSX_RK(NB, 1, SH, 3);        \
    else if (mode == 4) CSX_RK(NB, 1, SH, 4);        \
    else CSX_RK(NB, 1, SH, 0)
#define CSX_RS2(NB, SH)                       \
    if (mode == 1) CSX_RK(NB, 2, SH, 1);      \
    else if (mode == 4) CSX_RK(NB, 2, SH, 4); \
    else CSX_RK(NB, 2, SH, 0)
#define CSX_RSS(NB, SH)                      \
    if (passes == 2) { CSX_RS2(NB, SH); }    \
    else { CSX_RS1(NB, SH); }
#define CSX_RSP(NB)                            \
    if (share == 1) { CSX_RSS(NB, 1); }        \
    else if (share == 2) { CSX_RSS(NB, 2); }   \
    else if (share == 4) { CSX_RSS(NB, 4); }   \
    else { CSX_RSS(NB, 0); }
        switch (c) {
            case 0: CSX_RSP(1); break;
            case 1: CSX_RSP(2); break;
            case 2: CSX_RSP(3); break;
            case 3: CSX_RSP(4); break;
            default: CSX_RSP(5); break;
        }
#undef CSX_RSP
#undef CSX_RSS
#undef CSX_RS2
#undef CSX_RS1
#undef CSX_RK
        CSX_LAUNCH_CHECK();
    }
    return CSX_OK;
}

}  // namespace csx
