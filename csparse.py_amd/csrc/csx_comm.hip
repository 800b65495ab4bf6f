// Multi-GPU exchange of the sharded hot path, inside libcsx: RCCL over xGMI on the library's own streams.
//
// One process per GPU.  Rank 0 makes a unique id (csx_comm_unique_id), the launcher's side channel carries its 128
// bytes to the other ranks (csparse.py_amd/shard.py: a TCP hand-shake on MASTER_ADDR), every rank calls
// csx_comm_init(rank, world, id).  From then on every exchange is a call on device buffers behind csx handles,
// enqueued on the context's stream like any kernel of this library: no other runtime, no torch tensor, no
// synchronisation between "the kernel's stream" and "the collective's stream" -- a collective that follows a kernel
// is simply the next thing on the stream.
//
// RCCL is bound at csx_comm_init time (dlopen of librccl.so.1, the ROCm installation's), not at load time: a
// single-GPU user of libcsx never loads it, and world == 1 with id == NULL makes no RCCL call at all (every
// exchange is then a device copy).  world == 1 WITH an id is a real RCCL communicator of one rank (the API
// rehearsal a one-GPU box can run).
//
// The reference has no counterpart (one Python process); what is sharded are its sequences
// csparse.py:640-643 (ipvec, lsolve, ltsolve, pvec: right-hand-side blocks are independent) and :1210-1212
// (cs_gaxpy: column blocks give partial y vectors that must be summed).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>

#include "csx_internal.h"

namespace csx {

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

struct Comm {
    bool up = false;
    bool rccl = false;          // false: world of one without RCCL, every exchange a local copy
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t xs = nullptr;   // exchange stream of the overlapped sharded SpMV (everything else: the context's stream)
    hipEvent_t ev_k = nullptr, ev_x = nullptr;
    double *stage_d = nullptr;  // 1024 doubles: control-plane reductions / broadcasts of small host data
};

Rccl g_rccl;
Comm g_comm;

int load_rccl() {
    if (g_rccl.lib) return CSX_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        set_error("csx_comm: cannot load librccl.so.1 (%s)", dlerror());
        return CSX_ERUNTIME;
    }
#define CSX_SYM(field, name)                                              \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                \
    if (!g_rccl.field) {                                                  \
        set_error("csx_comm: %s missing from librccl", name);             \
        dlclose(h);                                                       \
        return CSX_ERUNTIME;                                              \
    }
    CSX_SYM(GetUniqueId, "ncclGetUniqueId")
    CSX_SYM(CommInitRank, "ncclCommInitRank")
    CSX_SYM(CommDestroy, "ncclCommDestroy")
    CSX_SYM(CommAbort, "ncclCommAbort")
    CSX_SYM(GetErrorString, "ncclGetErrorString")
    CSX_SYM(Broadcast, "ncclBroadcast")
    CSX_SYM(AllReduce, "ncclAllReduce")
    CSX_SYM(ReduceScatter, "ncclReduceScatter")
    CSX_SYM(Send, "ncclSend")
    CSX_SYM(Recv, "ncclRecv")
    CSX_SYM(GroupStart, "ncclGroupStart")
    CSX_SYM(GroupEnd, "ncclGroupEnd")
#undef CSX_SYM
    g_rccl.lib = h;
    return CSX_OK;
}

#define CSX_NCCL(call)                                                                                      \
    do {                                                                                                    \
        ncclResult_t _r = (call);                                                                           \
        if (_r != ncclSuccess) {                                                                            \
            csx::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(_r));         \
            return CSX_ERUNTIME;                                                                            \
        }                                                                                                   \
    } while (0)

int require_comm() {
    CSX_TRY(require_ready());
    if (!g_comm.up) {
        set_error("csx_comm: csx_comm_init has not been called");
        return CSX_EINVAL;
    }
    return CSX_OK;
}

__global__ void k_add_into(double *__restrict__ y, const double *__restrict__ a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a[i];
}

// y[i] += sum over q = 0 .. world-1 of part_q[i], in ascending rank order (a fixed order: the same bits on every run);
// part_q = own (the rank's own piece) for q == rank, recv + slot(q) * stride otherwise (slot: arrival slots in rank order)
__global__ void k_sum_parts(double *__restrict__ y, const double *__restrict__ own, const double *__restrict__ recv,
                            int64_t stride, int rank, int world, int64_t n) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < world; q++) s += q == rank ? own[i] : recv[(int64_t)(q < rank ? q : q - 1) * stride + i];
    y[i] += s;
}

// rows [r0, r0 + cnt) of B (n x K row-major) columns [c0, c0 + k) -> out (n x k row-major), or back
__global__ void k_block_cols(const double *__restrict__ B, int64_t n, int32_t K, int32_t c0, int32_t k, double *__restrict__ out,
                             int back) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    const int64_t i = t / k;
    const int32_t c = (int32_t)(t - i * k);
    if (back) const_cast<double *>(B)[i * K + c0 + c] = out[t];
    else out[t] = B[i * K + c0 + c];
}

}  // namespace

// One SpMV sharded by columns over the ranks: this rank's m x count block, split by rows into the pieces each rank
// will own of y.
struct ShardPlan {
    csx_handle_t hblock = 0;             // the rank's column block (m x count): NOT owned, so kept as its handle and resolved
                                         // at every call -- a block freed behind the plan's back is CSX_EINVAL, not a dangling pointer
    int32_t m = 0, n = 0, chunk = 0;     // rows, columns; rows per rank = ceil(m / world)
    std::vector<Csc *> pieces;           // world row pieces of the block (owned; empty when world == 1)
    double *work = nullptr;              // world * chunk partial y
    double *recv = nullptr;              // (world - 1) * chunk pieces received
};

void free_shardplan(ShardPlan *P) {
    if (!P) return;
    for (Csc *c : P->pieces) free_csc(c);
    dfree(P->work);
    dfree(P->recv);
    delete P;
}

}  // namespace csx

using namespace csx;

extern "C" {

int csx_comm_unique_id(uint8_t *id128) {
    if (!id128) return CSX_EINVAL;
    CSX_TRY(require_ready());
    CSX_TRY(load_rccl());
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    CSX_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, 128);
    return CSX_OK;
}

int csx_comm_init(int rank, int world, const uint8_t *id128) {
    CSX_TRY(require_ready());
    if (g_comm.up) {
        // the same call again is fine; another rank / world, or the other transport (a local world of one is up and RCCL is
        // asked for, or the reverse), is not: csx_comm_finalize first
        if (g_comm.rank == rank && g_comm.world == world && g_comm.rccl == (id128 != nullptr)) return CSX_OK;
        set_error("csx_comm_init: already initialised as rank %d of %d (%s); csx_comm_finalize first", g_comm.rank, g_comm.world,
                  g_comm.rccl ? "RCCL" : "local");
        return CSX_EINVAL;
    }
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return CSX_EINVAL;
    Comm c;
    c.rank = rank;
    c.world = world;
    auto build = [&]() -> int {
        if (id128) {
            CSX_TRY(load_rccl());
            ncclUniqueId id;
            std::memcpy(&id, id128, 128);
            CSX_NCCL(g_rccl.CommInitRank(&c.comm, world, id, rank));
            c.rccl = true;
        }
        CSX_HIP(hipStreamCreateWithFlags(&c.xs, hipStreamNonBlocking));
        CSX_HIP(hipEventCreateWithFlags(&c.ev_k, hipEventDisableTiming));
        CSX_HIP(hipEventCreateWithFlags(&c.ev_x, hipEventDisableTiming));
        CSX_TRY(dalloc(&c.stage_d, 1024));
        return CSX_OK;
    };
    const int st = build();
    if (st != CSX_OK) {   // whatever was made before the failing step goes back
        if (c.rccl && c.comm) (void)g_rccl.CommDestroy(c.comm);
        if (c.ev_k) (void)hipEventDestroy(c.ev_k);
        if (c.ev_x) (void)hipEventDestroy(c.ev_x);
        if (c.xs) (void)hipStreamDestroy(c.xs);
        dfree(c.stage_d);
        return st;
    }
    c.up = true;
    g_comm = c;
    return CSX_OK;
}

int csx_comm_finalize(void) {
    if (!g_comm.up) return CSX_OK;
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    (void)hipStreamSynchronize(g_comm.xs);
    if (g_comm.rccl && g_comm.comm) (void)g_rccl.CommDestroy(g_comm.comm);
    (void)hipEventDestroy(g_comm.ev_k);
    (void)hipEventDestroy(g_comm.ev_x);
    (void)hipStreamDestroy(g_comm.xs);
    dfree(g_comm.stage_d);
    g_comm = Comm();
    return CSX_OK;
}

int csx_comm_info(int *rank, int *world, int *uses_rccl) {
    if (!g_comm.up) return CSX_EINVAL;
    if (rank) *rank = g_comm.rank;
    if (world) *world = g_comm.world;
    if (uses_rccl) *uses_rccl = g_comm.rccl ? 1 : 0;
    return CSX_OK;
}

/* vals[0..count) <- sum (op 0) or max (op 1) over the ranks; count <= 1024.  Control plane (timings, checks). */
int csx_comm_allreduce_host(double *vals, int count, int op) {
    CSX_TRY(require_comm());
    if (!vals || count < 0 || count > 1024 || (op != 0 && op != 1)) return CSX_EINVAL;
    if (!g_comm.rccl || count == 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    CSX_HIP(hipMemcpyAsync(g_comm.stage_d, vals, (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
    CSX_NCCL(g_rccl.AllReduce(g_comm.stage_d, g_comm.stage_d, (size_t)count, ncclDouble, op ? ncclMax : ncclSum, g_comm.comm, s));
    CSX_HIP(hipMemcpyAsync(vals, g_comm.stage_d, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    return CSX_OK;
}

/* Every rank has reached this call and finished the work on its context's stream. */
int csx_comm_barrier(void) {
    double one = 1.0;
    CSX_TRY(require_comm());
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    return csx_comm_allreduce_host(&one, 1, 0);
}

/* bytes of host memory from `root` to everyone (small control data: a pickled choice, a checksum list). */
int csx_comm_bcast_host(void *buf, int64_t bytes, int root) {
    CSX_TRY(require_comm());
    if (bytes < 0 || (bytes > 0 && !buf) || root < 0 || root >= g_comm.world) return CSX_EINVAL;
    if (!g_comm.rccl || bytes == 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    char *d = nullptr;
    CSX_TRY(tmp.alloc(&d, (size_t)bytes));
    if (g_comm.rank == root) CSX_HIP(hipMemcpyAsync(d, buf, (size_t)bytes, hipMemcpyHostToDevice, s));
    CSX_NCCL(g_rccl.Broadcast(d, d, (size_t)bytes, ncclChar, root, g_comm.comm, s));
    if (g_comm.rank != root) CSX_HIP(hipMemcpyAsync(buf, d, (size_t)bytes, hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    return CSX_OK;
}

/* "Factor once, ship the factor" (SURVEY 8e): the CSC matrix *A of `root` arrives as a new matrix on every other rank
 * (three broadcasts: p, i, x; sizes first).  On the root *A is unchanged. */
int csx_comm_bcast_csc(csx_handle_t *A, int root) {
    CSX_TRY(require_comm());
    if (!A || root < 0 || root >= g_comm.world) return CSX_EINVAL;
    const bool am_root = g_comm.rank == root;
    Csc *M = am_root ? csc(*A) : nullptr;
    if (am_root && !M) return CSX_EINVAL;
    int64_t meta[4] = {0, 0, 0, 0};
    if (am_root) {
        meta[0] = M->m;
        meta[1] = M->n;
        meta[2] = M->nnz;
        meta[3] = M->x != nullptr;
    }
    CSX_TRY(csx_comm_bcast_host(meta, sizeof meta, root));
    if (!am_root) {
        csx_handle_t h = 0;
        CSX_TRY(csx_csc_alloc((int32_t)meta[0], (int32_t)meta[1], (int32_t)meta[2], (int)meta[3], &h));
        *A = h;
        M = csc(h);
    }
    if (!g_comm.rccl) return CSX_OK;
    hipStream_t s = ctx().stream;
    CSX_NCCL(g_rccl.Broadcast(M->p, M->p, (size_t)M->n + 1, ncclInt32, root, g_comm.comm, s));
    if (M->nnz) CSX_NCCL(g_rccl.Broadcast(M->i, M->i, (size_t)M->nnz, ncclInt32, root, g_comm.comm, s));
    if (M->nnz && M->x) CSX_NCCL(g_rccl.Broadcast(M->x, M->x, (size_t)M->nnz, ncclDouble, root, g_comm.comm, s));
    return CSX_OK;
}

int csx_comm_bcast_vec(csx_handle_t hv, int root) {
    CSX_TRY(require_comm());
    Vec *v = vec(hv);
    if (!v || root < 0 || root >= g_comm.world) return CSX_EINVAL;
    if (!g_comm.rccl || v->len == 0) return CSX_OK;
    CSX_NCCL(g_rccl.Broadcast(v->d, v->d, (size_t)v->len, ncclDouble, root, g_comm.comm, ctx().stream));
    return CSX_OK;
}

/* out = this rank's piece of the sum of the ranks' `full` vectors: full has world * len(out) entries, rank r receives
 * entries [r len, (r + 1) len).  One ncclReduceScatter on the context's stream. */
int csx_comm_reduce_scatter_vec(csx_handle_t hfull, csx_handle_t hout) {
    CSX_TRY(require_comm());
    Vec *f = vec(hfull), *o = vec(hout);
    if (!f || !o || f->len != o->len * g_comm.world) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    if (!g_comm.rccl) {
        CSX_HIP(hipMemcpyAsync(o->d, f->d, (size_t)o->len * sizeof(double), hipMemcpyDeviceToDevice, s));
        return CSX_OK;
    }
    CSX_NCCL(g_rccl.ReduceScatter(f->d, o->d, (size_t)o->len, ncclDouble, ncclSum, g_comm.comm, s));
    return CSX_OK;
}

int csx_comm_allreduce_vec(csx_handle_t hv) {
    CSX_TRY(require_comm());
    Vec *v = vec(hv);
    if (!v) return CSX_EINVAL;
    if (!g_comm.rccl || v->len == 0) return CSX_OK;
    CSX_NCCL(g_rccl.AllReduce(v->d, v->d, (size_t)v->len, ncclDouble, ncclSum, g_comm.comm, ctx().stream));
    return CSX_OK;
}

/* Right-hand-side blocks leave the root: `src` (root only; world blocks of `len` doubles, rank order) -> every rank's
 * `dst` (len doubles).  world - 1 point-to-point sends in one RCCL group; the root's own block is a device copy. */
int csx_comm_scatter_blocks(csx_handle_t hsrc, csx_handle_t hdst, int64_t len, int root) {
    CSX_TRY(require_comm());
    Vec *d = vec(hdst);
    const bool am_root = g_comm.rank == root;
    Vec *sv = am_root ? vec(hsrc) : nullptr;
    if (!d || d->len < len || len < 0 || root < 0 || root >= g_comm.world || (am_root && (!sv || sv->len < len * g_comm.world)))
        return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    if (am_root)
        CSX_HIP(hipMemcpyAsync(d->d, (const double *)sv->d + (int64_t)root * len, (size_t)len * sizeof(double),
                               hipMemcpyDeviceToDevice, s));
    if (!g_comm.rccl || g_comm.world == 1 || len == 0) return CSX_OK;
    CSX_NCCL(g_rccl.GroupStart());
    if (am_root) {
        for (int r = 0; r < g_comm.world; r++)
            if (r != root) CSX_NCCL(g_rccl.Send((const double *)sv->d + (int64_t)r * len, (size_t)len, ncclDouble, r, g_comm.comm, s));
    } else {
        CSX_NCCL(g_rccl.Recv(d->d, (size_t)len, ncclDouble, root, g_comm.comm, s));
    }
    CSX_NCCL(g_rccl.GroupEnd());
    return CSX_OK;
}

/* Solution blocks back to the root: every rank's `block` (len doubles) -> `out` on the root (world * len, rank order).
 * The root's inbound links are the bound. */
int csx_comm_gather_blocks(csx_handle_t hblock, csx_handle_t hout, int64_t len, int root) {
    CSX_TRY(require_comm());
    Vec *b = vec(hblock);
    const bool am_root = g_comm.rank == root;
    Vec *o = am_root ? vec(hout) : nullptr;
    if (!b || b->len < len || len < 0 || root < 0 || root >= g_comm.world || (am_root && (!o || o->len < len * g_comm.world)))
        return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    if (am_root)
        CSX_HIP(hipMemcpyAsync((double *)o->d + (int64_t)root * len, b->d, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (!g_comm.rccl || g_comm.world == 1 || len == 0) return CSX_OK;
    CSX_NCCL(g_rccl.GroupStart());
    if (am_root) {
        for (int r = 0; r < g_comm.world; r++)
            if (r != root) CSX_NCCL(g_rccl.Recv((double *)o->d + (int64_t)r * len, (size_t)len, ncclDouble, r, g_comm.comm, s));
    } else {
        CSX_NCCL(g_rccl.Send(b->d, (size_t)len, ncclDouble, root, g_comm.comm, s));
    }
    CSX_NCCL(g_rccl.GroupEnd());
    return CSX_OK;
}

/* Columns [c0, c0 + k) of the n x K row-major block B as a contiguous n x k block (back == 0), or written back into B
 * (back != 0): how a rank's share of a batch of right-hand sides is cut out of / returned to the caller's block. */
int csx_block_cols(csx_handle_t hB, int64_t n, int32_t K, int32_t c0, int32_t k, csx_handle_t hout, int back) {
    CSX_TRY(require_ready());
    Vec *B = vec(hB), *o = vec(hout);
    if (!B || !o || n < 0 || K <= 0 || k < 0 || c0 < 0 || c0 + k > K || B->len < n * K || o->len < n * k) return CSX_EINVAL;
    if (n * k == 0) return CSX_OK;
    hipLaunchKernelGGL(k_block_cols, dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, ctx().stream, (const double *)B->d, n, K,
                       c0, k, (double *)o->d, back);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

/* ---- one SpMV sharded by columns (second bullet; csparse.py:1210-1212) ---------------------------------
 * Rank r holds A_r = columns [first_r, first_r + count_r) of an m x n matrix (csx_csc_col_block) and the matching
 * slice x_r.  y = sum_r A_r x_r; rank r ends up with rows [r chunk, min((r + 1) chunk, m)), chunk = ceil(m / world).
 * The plan cuts A_r by rows into the world pieces y is owned in (through two stable transposes: a piece's columns come
 * out with ascending rows), so that a piece's partial sums can leave while the next piece is computed. */
static int sharded_plan(csx_handle_t hblock, int W, csx_handle_t *out);

int csx_gaxpy_sharded_plan(csx_handle_t hblock, csx_handle_t *out) {
    CSX_TRY(require_comm());
    return sharded_plan(hblock, g_comm.world, out);
}

/* The same plan cut for a world the communicator does not have (the stand-in transport's ranks). */
int csx_gaxpy_sharded_plan_for(csx_handle_t hblock, int world, csx_handle_t *out) {
    CSX_TRY(require_comm());
    if (world < 1) return CSX_EINVAL;
    return sharded_plan(hblock, world, out);
}

static int sharded_plan(csx_handle_t hblock, int W, csx_handle_t *out) {
    Csc *A = csc(hblock);
    if (!A || !out || !A->x) return CSX_EINVAL;
    ShardPlan *P = new ShardPlan();
    P->hblock = hblock;
    P->m = A->m;
    P->n = A->n;
    P->chunk = (A->m + W - 1) / W;
    int st = dalloc(&P->work, (size_t)std::max<int64_t>((int64_t)P->chunk * W, 1));
    if (st == CSX_OK && W > 1) st = dalloc(&P->recv, (size_t)std::max<int64_t>((int64_t)P->chunk * (W - 1), 1));
    if (st == CSX_OK && W > 1) {
        // rows of A_r as columns (stable transpose), cut, transposed back: piece q = rows [q chunk, ...) of A_r, rebased
        Csc T;
        st = transpose_device(A, true, &T);
        for (int q = 0; q < W && st == CSX_OK; q++) {
            const int32_t r0 = std::min<int64_t>((int64_t)q * P->chunk, A->m);
            const int32_t cnt = (int32_t)std::min<int64_t>(P->chunk, (int64_t)A->m - r0);
            Csc Tq;
            st = col_block_device(&T, r0, cnt, &Tq);
            Csc *piece = new Csc();
            if (st == CSX_OK) st = transpose_device(&Tq, true, piece);
            dfree(Tq.p);
            dfree(Tq.i);
            dfree(Tq.x);
            P->pieces.push_back(piece);
            if (st == CSX_OK && piece->nnz) st = gaxpy_prepare_device(piece, CSX_GAXPY_AUTO);
        }
        dfree(T.p);
        dfree(T.i);
        dfree(T.x);
    }
    if (st != CSX_OK) {
        free_shardplan(P);
        return st;
    }
    *out = put(K_SHARDPLAN, P);
    return CSX_OK;
}

int csx_gaxpy_sharded_rows(csx_handle_t hplan, int32_t *first, int32_t *count) {
    CSX_TRY(require_comm());
    ShardPlan *P = (ShardPlan *)get(hplan, K_SHARDPLAN);
    if (!P) return CSX_EINVAL;
    const int32_t r0 = (int32_t)std::min<int64_t>((int64_t)g_comm.rank * P->chunk, P->m);
    if (first) *first = r0;
    if (count) *count = (int32_t)std::min<int64_t>(P->chunk, (int64_t)P->m - r0);
    return CSX_OK;
}

/* y_mine += (sum over ranks of A_r x_r)[my rows].  x: this rank's slice (count entries); y_mine: chunk entries.
 * how == 0: the whole block in one SpMV into a full-length partial y, then ONE ncclReduceScatter (RCCL's own algorithm
 *           and summation order).
 * how == 1: piece by piece in rotated order (rank r computes the piece of rank r + 1 first, its own last); as soon as
 *           a piece's kernel has finished its partial sums go straight to their owner over the direct xGMI link
 *           (ncclSend / ncclRecv on the exchange stream) while the next piece is computed; the owner adds the world
 *           partial pieces in ascending rank order (reproducible bits).  Per link and rank: chunk * 8 bytes, all
 *           seven links busy at once -- the one-shot exchange SURVEY 8e asks for on a fully connected mesh. */
int csx_gaxpy_sharded(csx_handle_t hplan, csx_handle_t hx, csx_handle_t hy, int how) {
    CSX_TRY(require_comm());
    ShardPlan *P = (ShardPlan *)get(hplan, K_SHARDPLAN);
    Vec *x = vec(hx), *y = vec(hy);
    Csc *block = P ? csc(P->hblock) : nullptr;      // handles carry a generation: a freed (or recycled) block resolves to null
    if (!P || !block || block->m != P->m || block->n != P->n || !x || !y || x->len < P->n || y->len < P->chunk ||
        (how != 0 && how != 1))
        return CSX_EINVAL;
    const int W = g_comm.world, rank = g_comm.rank;
    hipStream_t s = ctx().stream;
    const double *xd = (const double *)x->d;
    double *yd = (double *)y->d;
    const int64_t chunk = P->chunk;
    const int32_t mine = (int32_t)std::min<int64_t>(chunk, std::max<int64_t>(0, (int64_t)P->m - (int64_t)rank * chunk));
    if (W == 1 && !g_comm.rccl) return gaxpy_device(block, xd, yd, CSX_GAXPY_AUTO);   // nothing to exchange
    CSX_HIP(hipMemsetAsync(P->work, 0, (size_t)(chunk * W) * sizeof(double), s));
    if (how == 0 || W == 1) {
        CSX_TRY(gaxpy_device(block, xd, P->work, CSX_GAXPY_AUTO));
        double *piece = P->recv ? P->recv : P->work;     // world of one under RCCL: in place
        CSX_NCCL(g_rccl.ReduceScatter(P->work, piece, (size_t)chunk, ncclDouble, ncclSum, g_comm.comm, s));
        if (mine) {
            hipLaunchKernelGGL(k_add_into, dim3((unsigned)((mine + 255) / 256)), dim3(256), 0, s, yd, piece, (int64_t)mine);
            CSX_LAUNCH_CHECK();
        }
        return CSX_OK;
    }
    // how == 1: the exchange stream starts behind everything already on the context's stream
    CSX_HIP(hipEventRecord(g_comm.ev_k, s));
    CSX_HIP(hipStreamWaitEvent(g_comm.xs, g_comm.ev_k, 0));
    for (int step = 1; step <= W; step++) {
        const int q = (rank + step) % W;                 // the piece computed now; step == W: my own
        CSX_TRY(gaxpy_device(P->pieces[q], xd, P->work + (int64_t)q * chunk, CSX_GAXPY_AUTO));
        if (step == W) break;
        const int from = (rank - step + W) % W;          // who computes MY piece at this step
        CSX_HIP(hipEventRecord(g_comm.ev_k, s));
        CSX_HIP(hipStreamWaitEvent(g_comm.xs, g_comm.ev_k, 0));
        CSX_NCCL(g_rccl.GroupStart());
        CSX_NCCL(g_rccl.Send(P->work + (int64_t)q * chunk, (size_t)chunk, ncclDouble, q, g_comm.comm, g_comm.xs));
        CSX_NCCL(g_rccl.Recv(P->recv + (int64_t)(from < rank ? from : from - 1) * chunk, (size_t)chunk, ncclDouble, from, g_comm.comm,
                             g_comm.xs));
        CSX_NCCL(g_rccl.GroupEnd());
    }
    CSX_HIP(hipEventRecord(g_comm.ev_x, g_comm.xs));
    CSX_HIP(hipStreamWaitEvent(s, g_comm.ev_x, 0));      // the context's stream continues behind the last arrival
    if (mine) {
        hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((mine + 255) / 256)), dim3(256), 0, s, yd, P->work + (int64_t)rank * chunk,
                           P->recv, chunk, rank, W, (int64_t)mine);
        CSX_LAUNCH_CHECK();
    }
    return CSX_OK;
}

/* The steps of how == 1 one at a time, for a transport that is not RCCL (the host stand-in of shard.py, which carries
 * the N > 1 tests on a one-GPU box): piece q's partial sums into the plan's work buffer; the buffers themselves
 * (work: world pieces of chunk doubles, piece q at q * chunk; recv: world - 1 arrival slots in ascending rank order
 * of the senders); the owner's final sum over the world pieces in ascending rank order. */
int csx_gaxpy_sharded_piece(csx_handle_t hplan, int q, csx_handle_t hx) {
    CSX_TRY(require_comm());
    ShardPlan *P = (ShardPlan *)get(hplan, K_SHARDPLAN);
    Vec *x = vec(hx);
    if (!P || !x || x->len < P->n || q < 0 || q >= (int)P->pieces.size()) return CSX_EINVAL;
    double *dst = P->work + (int64_t)q * P->chunk;
    CSX_HIP(hipMemsetAsync(dst, 0, (size_t)P->chunk * sizeof(double), ctx().stream));
    return gaxpy_device(P->pieces[q], (const double *)x->d, dst, CSX_GAXPY_AUTO);
}

int csx_gaxpy_sharded_buffers(csx_handle_t hplan, void **work, void **recv, int64_t *chunk) {
    ShardPlan *P = (ShardPlan *)get(hplan, K_SHARDPLAN);
    if (!P) return CSX_EINVAL;
    if (work) *work = P->work;
    if (recv) *recv = P->recv;
    if (chunk) *chunk = P->chunk;
    return CSX_OK;
}

int csx_gaxpy_sharded_sum(csx_handle_t hplan, int rank, int world, csx_handle_t hy) {
    CSX_TRY(require_ready());
    ShardPlan *P = (ShardPlan *)get(hplan, K_SHARDPLAN);
    Vec *y = vec(hy);
    if (!P || !y || y->len < P->chunk || world != (int)P->pieces.size() || rank < 0 || rank >= world) return CSX_EINVAL;
    const int64_t chunk = P->chunk;
    const int32_t mine = (int32_t)std::min<int64_t>(chunk, std::max<int64_t>(0, (int64_t)P->m - (int64_t)rank * chunk));
    if (!mine) return CSX_OK;
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((mine + 255) / 256)), dim3(256), 0, ctx().stream, (double *)y->d,
                       P->work + (int64_t)rank * chunk, P->recv, chunk, rank, world, (int64_t)mine);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

}  // extern "C"
