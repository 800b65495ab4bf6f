// Context, handles, host<->device transfers, timers.
#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "csx_internal.h"

namespace csx {

static thread_local std::string g_error;
static Context g_ctx;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

Context &ctx() { return g_ctx; }

int require_ready() {
    if (!g_ctx.ready) {
        set_error("csx_init() has not been called");
        return CSX_ERUNTIME;
    }
    return CSX_OK;
}

// Handle = (generation << 32) | (slot + 1).  The table is guarded by a mutex: ctypes releases the GIL and
// Python finalisers call csx_free from whichever thread collects garbage.  A slot's generation changes every
// time it is reused, so a stale handle never aliases a newer object of the same kind.  (The mutex protects
// the TABLE; the library still runs one operation at a time per context -- one stream.)
static std::mutex g_objects_mu;
static std::vector<size_t> g_free_slots;

csx_handle_t put(Kind k, void *ptr) {
    std::lock_guard<std::mutex> lock(g_objects_mu);
    auto &objs = g_ctx.objects;
    size_t i;
    if (!g_free_slots.empty()) {
        i = g_free_slots.back();
        g_free_slots.pop_back();
    } else {
        objs.push_back(Object());
        i = objs.size() - 1;
    }
    const uint32_t gen = objs[i].gen + 1 ? objs[i].gen + 1 : 1;
    objs[i].kind = k;
    objs[i].ptr = ptr;
    objs[i].gen = gen;
    return ((csx_handle_t)gen << 32) | (csx_handle_t)(i + 1);
}

static Object *slot_of(csx_handle_t h) {   // caller holds g_objects_mu
    auto &objs = g_ctx.objects;
    const uint64_t idx = h & 0xffffffffull;
    if (idx == 0 || idx > objs.size()) return nullptr;
    Object &o = objs[idx - 1];
    return (o.kind != K_FREE && o.gen == (uint32_t)(h >> 32)) ? &o : nullptr;
}

void *get(csx_handle_t h, Kind k) {
    std::lock_guard<std::mutex> lock(g_objects_mu);
    Object *o = slot_of(h);
    return (o && o->kind == k) ? o->ptr : nullptr;
}

// ---- device memory: a caching allocator -------------------------------------------------------------
// hipMalloc / hipFree of multi-GB blocks cost milliseconds and hipFree synchronises the device, which
// would dominate the short operations here (a transpose or product allocates several work arrays).
// Freed blocks are kept on a free list keyed by size and handed out again; every use of device memory in
// this library is ordered on the context's single stream (csx_set_stream synchronises when it changes),
// so a block can be reused as soon as the host has released it.  Blocks carry at least 64 bytes of slack
// past the requested size (aligned vector loads may read a few entries past an array's end).  The cache is capped (default: a quarter
// of the device's memory, CSX_POOL_LIMIT_MB overrides, CSX_NO_POOL=1 disables) and emptied when an
// allocation fails.
namespace {
struct Pool {
    std::mutex mu;
    struct Idle {
        void *p;
        uint64_t stamp;                            // when it was released: the oldest goes first when room is needed
    };
    std::multimap<size_t, Idle> idle;            // size -> block
    uint64_t clock = 0;
    std::unordered_map<void *, size_t> size_of;  // every block handed out or idle
    size_t cached = 0, live = 0, limit = 0;
    bool enabled = true, configured = false;
};
Pool g_pool;

size_t pool_round(size_t bytes) {
    if (bytes < 512) return 512;
    if (bytes <= (1u << 20)) {
        size_t r = 512;
        while (r < bytes) r <<= 1;
        return r;
    }
    const size_t g = (size_t)2 << 20;
    return (bytes + g - 1) / g * g;
}

void pool_configure() {
    if (g_pool.configured) return;
    g_pool.configured = true;
    g_pool.enabled = !getenv("CSX_NO_POOL");
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
    g_pool.limit = total_b / 4;
    if (const char *e = getenv("CSX_POOL_LIMIT_MB")) g_pool.limit = (size_t)strtoull(e, nullptr, 10) << 20;
}

void pool_release_locked() {
    for (auto &kv : g_pool.idle) {
        g_pool.size_of.erase(kv.second.p);
        (void)hipFree(kv.second.p);
    }
    g_pool.idle.clear();
    g_pool.cached = 0;
}
}  // namespace

void pool_trim() {
    std::lock_guard<std::mutex> lock(g_pool.mu);
    pool_release_locked();
}

// cap of the cache in bytes (tests; 0 restores the default quarter of the device)
void pool_set_limit(size_t bytes) {
    std::lock_guard<std::mutex> lock(g_pool.mu);
    pool_configure();
    if (bytes == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
        bytes = total_b / 4;
    }
    g_pool.limit = bytes;
}

void pool_stats(size_t *cached, size_t *live) {
    std::lock_guard<std::mutex> lock(g_pool.mu);
    if (cached) *cached = g_pool.cached;
    if (live) *live = g_pool.live;
}

int dmalloc(void **p, size_t bytes) {
    *p = nullptr;
    std::lock_guard<std::mutex> lock(g_pool.mu);
    pool_configure();
    const size_t want = pool_round(bytes + 64);   // every block has >= 64 readable bytes past its logical end
    if (g_pool.enabled) {
        // an idle block of at most 25 % more; for a big request (>= 64 MB) up to three times as much: fresh gigabytes from the
        // driver cost tens of milliseconds (30 ms for the 2 GB of a 5M-row factor inside its first csx_chol when the pool
        // happened to hold nothing of that size), and a block that idles in the cache serves nobody
        auto it = g_pool.idle.lower_bound(want);
        // (the slack of a big block is capped in absolute terms: a 2 GB factor must not pin a 6 GB block for its lifetime)
        const size_t most = want >= ((size_t)64 << 20) ? std::min(3 * want, want + ((size_t)1 << 30)) : want + want / 4;
        if (it != g_pool.idle.end() && it->first <= most) {
            *p = it->second.p;
            g_pool.cached -= it->first;
            g_pool.live += it->first;
            g_pool.idle.erase(it);
            return CSX_OK;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess && !g_pool.idle.empty()) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        pool_release_locked();
        e = hipMalloc(p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        set_error("device allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return CSX_ERUNTIME;
    }
    g_pool.size_of[*p] = want;
    g_pool.live += want;
    return CSX_OK;
}

void dfree(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pool.mu);
    auto it = g_pool.size_of.find(p);
    if (it == g_pool.size_of.end()) {   // not ours (never happens for owned blocks); be safe
        (void)hipFree(p);
        return;
    }
    const size_t sz = it->second;
    g_pool.live -= sz;
    if (g_pool.enabled && sz <= g_pool.limit) {
        // The block just released is the likeliest to be asked for again (a loop's temporaries): when the cache is
        // full it is the OLDEST idle blocks that go back to the driver, not this one.  (Freeing the newcomer instead
        // made a 12 GB work array of cs_multiply cost a hipFree + hipMalloc per call once an earlier phase had filled
        // the cache: 590 ms per multiply instead of 15.)
        // Small blocks may overshoot the cap by up to 1/16 of it instead of evicting: handing a multi-GB block back to
        // the driver takes tens of milliseconds, which is no price for caching a few megabytes.
        const size_t cap = sz < ((size_t)64 << 20) ? g_pool.limit + g_pool.limit / 16 : g_pool.limit;
        while (g_pool.cached + sz > cap && !g_pool.idle.empty()) {
            auto old = g_pool.idle.begin();
            for (auto k = g_pool.idle.begin(); k != g_pool.idle.end(); ++k)
                if (k->second.stamp < old->second.stamp) old = k;
            g_pool.cached -= old->first;
            g_pool.size_of.erase(old->second.p);
            (void)hipFree(old->second.p);
            g_pool.idle.erase(old);
        }
        g_pool.idle.emplace(sz, Pool::Idle{p, ++g_pool.clock});
        g_pool.cached += sz;
        return;
    }
    g_pool.size_of.erase(it);
    (void)hipFree(p);
}

void free_gather(Gather *g) {
    if (!g) return;
    dfree(g->ptr);
    dfree(g->idx);
    dfree(g->val);
    delete g;
}

void free_tiled(TiledPlan *t) {
    if (!t) return;
    dfree(t->tile_ptr);
    dfree(t->tile_len);
    dfree(t->tile_key);
    dfree(t->tile_key24);
    dfree(t->tile_base);
    dfree(t->tile_val);
    delete t;
}

void free_csc(Csc *A) {
    if (!A) return;
    if (A->owns) {
        dfree(A->p);
        dfree(A->i);
        dfree(A->x);
    }
    free_gather(A->rows);
    free_tiled(A->tiled);
    if (A->house) {
        dfree(A->house->cols);
        delete A->house;
    }
    free_clique_cache(A->clique);
    delete A;
}

static void free_vec(Vec *v) {
    if (!v) return;
    if (v->owns) dfree(v->d);
    delete v;
}

static void free_object(Object &o) {
    switch (o.kind) {
        case K_CSC: free_csc((Csc *)o.ptr); break;
        case K_VEC:
        case K_IVEC: free_vec((Vec *)o.ptr); break;
        case K_TRIPLAN: free_triplan((TriPlan *)o.ptr); break;
        case K_CHOLPLAN: free_cholplan((CholPlan *)o.ptr); break;
        case K_SHARDPLAN: free_shardplan((ShardPlan *)o.ptr); break;
        default: break;
    }
    o.kind = K_FREE;   // the generation stays: the next put() of this slot bumps it
    o.ptr = nullptr;
}

__global__ void k_fill_f64(double *p, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

}  // namespace csx

using namespace csx;

extern "C" {

__global__ void k_warm(int *p) { if (threadIdx.x == 0) *p += 1; }

int csx_init(int device) {
    Context &c = ctx();
    if (c.ready) {
        if (c.device == device) return CSX_OK;
        set_error("csx_init: context already bound to device %d", c.device);
        return CSX_EINVAL;
    }
    int count = 0;
    CSX_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        set_error("csx_init: device %d out of range (%d visible)", device, count);
        return CSX_EINVAL;
    }
    CSX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CSX_HIP(hipGetDeviceProperties(&prop, device));
    c.cus = prop.multiProcessorCount;
    // every exit below that is not the last one gives back what was made so far (streams, events, the warm-up block)
    int *d_warm = nullptr;
    auto undo = [&](hipError_t e, const char *what) {
        set_error("csx_init: %s -> %s", what, hipGetErrorString(e));
        if (d_warm) (void)hipFree(d_warm);
        if (c.ev0) (void)hipEventDestroy(c.ev0);
        if (c.ev1) (void)hipEventDestroy(c.ev1);
        if (c.side) (void)hipStreamDestroy(c.side);
        if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
        c = Context();
        return CSX_ERUNTIME;
    };
#define CSX_INIT_STEP(call)                           \
    do {                                              \
        const hipError_t _e = (call);                 \
        if (_e != hipSuccess) return undo(_e, #call); \
    } while (0)
    CSX_INIT_STEP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    CSX_INIT_STEP(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
    {
        // the first LARGE copy from pageable memory on a stream sets up its copy path (6 ms seen inside the first csx_chol)
        std::vector<int> warm((size_t)1 << 20, 0);
        CSX_INIT_STEP(hipMalloc(&d_warm, warm.size() * sizeof(int)));
        CSX_INIT_STEP(hipMemcpyAsync(d_warm, warm.data(), warm.size() * sizeof(int), hipMemcpyHostToDevice, c.side));
        hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, c.side, d_warm);      // (and a stream's first kernel makes its queue)
        CSX_INIT_STEP(hipMemcpyAsync(warm.data(), d_warm, warm.size() * sizeof(int), hipMemcpyDeviceToHost, c.side));
        CSX_INIT_STEP(hipStreamSynchronize(c.side));
        CSX_INIT_STEP(hipFree(d_warm));
        d_warm = nullptr;
    }
    CSX_INIT_STEP(hipEventCreate(&c.ev0));
    CSX_INIT_STEP(hipEventCreate(&c.ev1));
#undef CSX_INIT_STEP
    c.device = device;
    c.ready = true;
    return CSX_OK;
}

int csx_finalize(void) {
    Context &c = ctx();
    if (!c.ready) return CSX_OK;
    (void)hipStreamSynchronize(c.stream);
    {
        std::lock_guard<std::mutex> lock(g_objects_mu);
        for (auto &o : c.objects) free_object(o);
        c.objects.clear();
        g_free_slots.clear();
    }
    pool_trim();
    (void)hipEventDestroy(c.ev0);
    (void)hipEventDestroy(c.ev1);
    (void)hipStreamDestroy(c.own_stream);
    (void)hipStreamDestroy(c.side);
    c = Context();
    return CSX_OK;
}

const char *csx_last_error(void) { return g_error.c_str(); }

int csx_sync(void) {
    CSX_TRY(require_ready());
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    return CSX_OK;
}

int csx_set_stream(void *hip_stream) {
    CSX_TRY(require_ready());
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    ctx().stream = hip_stream ? (hipStream_t)hip_stream : ctx().own_stream;
    return CSX_OK;
}

int csx_mem_trim(void) {
    CSX_TRY(require_ready());
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    pool_trim();
    return CSX_OK;
}

int csx_mem_info(int64_t *cached_bytes, int64_t *live_bytes, int64_t *device_free_bytes) {
    CSX_TRY(require_ready());
    size_t c = 0, l = 0, f = 0, t = 0;
    pool_stats(&c, &l);
    CSX_HIP(hipMemGetInfo(&f, &t));
    if (cached_bytes) *cached_bytes = (int64_t)c;
    if (live_bytes) *live_bytes = (int64_t)l;
    if (device_free_bytes) *device_free_bytes = (int64_t)f;
    return CSX_OK;
}

int csx_device_info(char *name, int name_cap, int *compute_units, int64_t *hbm_bytes) {
    CSX_TRY(require_ready());
    hipDeviceProp_t prop;
    CSX_HIP(hipGetDeviceProperties(&prop, ctx().device));
    if (name && name_cap > 0) {
        std::snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return CSX_OK;
}

int csx_timer_start(void) {
    CSX_TRY(require_ready());
    CSX_HIP(hipEventRecord(ctx().ev0, ctx().stream));
    return CSX_OK;
}

int csx_timer_stop(double *ms) {
    CSX_TRY(require_ready());
    CSX_HIP(hipEventRecord(ctx().ev1, ctx().stream));
    CSX_HIP(hipEventSynchronize(ctx().ev1));
    float t = 0.f;
    CSX_HIP(hipEventElapsedTime(&t, ctx().ev0, ctx().ev1));
    if (ms) *ms = (double)t;
    return CSX_OK;
}

// ---- CSC ----------------------------------------------------------------

int csx_csc_alloc(int32_t m, int32_t n, int32_t nnz, int values, csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (m < 0 || n < 0 || nnz < 0 || !out) return CSX_EINVAL;
    Csc *A = new Csc();
    A->m = m;
    A->n = n;
    A->nnz = nnz;
    int st = dalloc(&A->p, (size_t)n + 1);
    if (st == CSX_OK) st = dalloc(&A->i, (size_t)nnz);
    if (st == CSX_OK && values) st = dalloc(&A->x, (size_t)nnz);
    if (st != CSX_OK) {
        free_csc(A);
        return st;
    }
    *out = put(K_CSC, A);
    return CSX_OK;
}

int csx_csc_upload(int32_t m, int32_t n, const int32_t *p, const int32_t *i, const double *x,
                   csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (m < 0 || n < 0 || !p || !out) return CSX_EINVAL;
    int32_t nnz = p[n];
    if (nnz < 0 || p[0] != 0 || (nnz > 0 && !i)) return CSX_EINVAL;
    for (int32_t j = 0; j < n; j++)
        if (p[j] > p[j + 1]) return CSX_EINVAL;
    for (int32_t q = 0; q < nnz; q++)
        if (i[q] < 0 || i[q] >= m) return CSX_EINVAL;  // the reference would raise IndexError
    csx_handle_t h;
    CSX_TRY(csx_csc_alloc(m, n, nnz, x != nullptr, &h));
    Csc *A = csc(h);
    hipStream_t s = ctx().stream;
    CSX_HIP(hipMemcpyAsync(A->p, p, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (nnz) CSX_HIP(hipMemcpyAsync(A->i, i, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (nnz && x) CSX_HIP(hipMemcpyAsync(A->x, x, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, s));
    CSX_HIP(hipStreamSynchronize(s));
    *out = h;
    return CSX_OK;
}

int csx_csc_wrap(int32_t m, int32_t n, int32_t nnz, void *d_p, void *d_i, void *d_x, csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (m < 0 || n < 0 || nnz < 0 || !d_p || (nnz > 0 && !d_i) || !out) return CSX_EINVAL;
    Csc *A = new Csc();
    A->m = m;
    A->n = n;
    A->nnz = nnz;
    A->p = (int32_t *)d_p;
    A->i = (int32_t *)d_i;
    A->x = (double *)d_x;
    A->owns = false;
    A->trusted = false;   // caller's arrays: checked (csc_validate) by the kernels that index work space with them
    *out = put(K_CSC, A);
    return CSX_OK;
}

}  // extern "C"

namespace csx {
__global__ void k_csc_validate(int32_t m, int32_t n, int32_t nnz, const int32_t *__restrict__ p,
                               const int32_t *__restrict__ i, int32_t *bad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    bool b = false;
    for (int64_t j = t; j < n; j += stride) b |= p[j] > p[j + 1] || p[j] < 0;
    if (t == 0) b |= p[0] != 0 || p[n] != nnz;
    for (int64_t q = t; q < nnz; q += stride) b |= (uint32_t)i[q] >= (uint32_t)m;
    if (b) atomicOr(bad, 1);
}

int csc_validate(Csc *A) {
    if (A->trusted) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    int32_t *bad = nullptr, h = 0;
    CSX_TRY(tmp.alloc(&bad, 1));
    CSX_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), s));
    const int64_t work = std::max<int64_t>((int64_t)A->n, (int64_t)A->nnz);
    const unsigned grid = (unsigned)std::min<int64_t>(4096, (work + 255) / 256 + 1);
    hipLaunchKernelGGL(k_csc_validate, dim3(grid), dim3(256), 0, s, A->m, A->n, A->nnz, A->p, A->i, bad);
    CSX_LAUNCH_CHECK();
    CSX_HIP(hipMemcpyAsync(&h, bad, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    if (h) return CSX_EINVAL;
    A->trusted = true;
    return CSX_OK;
}
}  // namespace csx

extern "C" {

int csx_csc_info(csx_handle_t h, int32_t *m, int32_t *n, int32_t *nnz, int *has_values) {
    Csc *A = csc(h);
    if (!A) return CSX_EINVAL;
    if (m) *m = A->m;
    if (n) *n = A->n;
    if (nnz) *nnz = A->nnz;
    if (has_values) *has_values = A->x != nullptr;
    return CSX_OK;
}

int csx_csc_download(csx_handle_t h, int32_t *p, int32_t *i, double *x) {
    CSX_TRY(require_ready());
    Csc *A = csc(h);
    if (!A) return CSX_EINVAL;
    hipStream_t s = ctx().stream;
    if (p) CSX_HIP(hipMemcpyAsync(p, A->p, ((size_t)A->n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (i && A->nnz) CSX_HIP(hipMemcpyAsync(i, A->i, (size_t)A->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (x && A->x && A->nnz)
        CSX_HIP(hipMemcpyAsync(x, A->x, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    return CSX_OK;
}

int csx_csc_ptrs(csx_handle_t h, void **d_p, void **d_i, void **d_x) {
    Csc *A = csc(h);
    if (!A) return CSX_EINVAL;
    if (d_p) *d_p = A->p;
    if (d_i) *d_i = A->i;
    if (d_x) *d_x = A->x;
    return CSX_OK;
}

int csx_free(csx_handle_t h) {
    Object taken;
    {
        std::lock_guard<std::mutex> lock(g_objects_mu);
        Object *o = slot_of(h);
        if (!o) return CSX_EINVAL;
        taken = *o;              // unlink under the lock, release the memory outside it
        o->kind = K_FREE;
        o->ptr = nullptr;
        g_free_slots.push_back((size_t)((h & 0xffffffffull) - 1));
    }
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    free_object(taken);
    return CSX_OK;
}

/* Kernel-selection overrides for tests (see Options in csx_internal.h). */
/* name -> (slot, kind): kind 0 = flag (0 / 1), otherwise the option's own value rule (normalise()). */
namespace {
struct OptSlot {
    const char *name;
    int Options::*field;
    int kind;
};
const OptSlot kOptSlots[] = {
    {"chol.dense_trees", &Options::chol_dense_trees, 0}, {"chol.band", &Options::chol_band, 0},
    {"chol.supernodes", &Options::chol_supernodes, 0},   {"chol.wband", &Options::chol_wband, 1},
    {"chol.wband_nb", &Options::chol_wband_nb, 2},       {"cholsol.dense_blocks", &Options::cholsol_dense_blocks, 0},
    {"spgemm.one_pass", &Options::spgemm_one_pass, 0},   {"tri.chain_walker", &Options::tri_chain_walker, 0},
    {"tri.components", &Options::tri_components, 0},     {"tri.columns", &Options::tri_columns, 0},
    {"gaxpy.keys24", &Options::gaxpy_keys24, 0},         {"gaxpy.tune_shape", &Options::gaxpy_tune_shape, 0},
    {"tri.row_waves", &Options::tri_row_waves, 0},       {"tri.push", &Options::tri_push, 0},
    {"tri.levels_where", &Options::tri_levels_where, 3}, {"tri.supernodes", &Options::tri_supernodes, 5},
    {"spgemm.ordered", &Options::spgemm_ordered, 0},     {"spgemm.chunks", &Options::spgemm_chunks, 4},
    {"lu.etree", &Options::lu_etree, 5},                 {"tri.graph", &Options::tri_graph, 6},
    {"sort.short_keys", &Options::sort_short_keys, 0},   {"chol.clique", &Options::chol_clique, 0},
    {"chol.forest", &Options::chol_forest, 0},           {"chol.exact", &Options::chol_exact, 0},
    {"tri.host_chains", &Options::tri_host_chains, 0},
                    {"cholsol.exact_variant", &Options::cholsol_exact_variant, 4},
};
int normalise(int kind, int value) {
    switch (kind) {
    case 0: return value != 0;
    case 1: return (value == 0 || value == 2) ? value : 1;
    case 2: return (value == 32 || value == -16 || value == -32) ? value : 16;   // negative: two launches per panel
    case 3: return (value == 1 || value == 2) ? value : 0;
    case 4: return value < 0 ? 0 : (value > 64 ? 64 : value);
    case 5: return (value == 0 || value == 2) ? value : 1;
    case 6: return (value == 0 || value == 1) ? value : 2;
    }
    return value;
}
int g_pool_limit_mb = 0;
}  // namespace

int csx_set_option(const char *name, int value) {
    if (!name) return CSX_EINVAL;
    const std::string n(name);
    if (n == "pool.limit_mb") {
        g_pool_limit_mb = value > 0 ? value : 0;
        pool_set_limit(value > 0 ? (size_t)value << 20 : 0);   // 0: the default
        return CSX_OK;
    }
    for (const OptSlot &s : kOptSlots)
        if (n == s.name) {
            ctx().opt.*(s.field) = normalise(s.kind, value);
            return CSX_OK;
        }
    return CSX_EINVAL;
}

int csx_get_option(const char *name, int *value) {
    if (!name || !value) return CSX_EINVAL;
    const std::string n(name);
    if (n == "pool.limit_mb") {
        *value = g_pool_limit_mb;
        return CSX_OK;
    }
    for (const OptSlot &s : kOptSlots)
        if (n == s.name) {
            *value = ctx().opt.*(s.field);
            return CSX_OK;
        }
    return CSX_EINVAL;
}

/* The SpMV plans cached on a matrix (the row-major copy, the LDS-tiled regrouping) hold COPIES of its values.
 * After the arrays behind csx_csc_ptrs / csx_csc_wrap have been changed in place, drop them: the next
 * csx_gaxpy rebuilds what it needs. */
int csx_csc_invalidate(csx_handle_t h) {
    CSX_TRY(require_ready());
    Csc *A = csc(h);
    if (!A) return CSX_EINVAL;
    (void)hipStreamSynchronize(ctx().stream);
    if (!A->owns) A->trusted = false;   // the caller changed its arrays: check them again
    free_gather(A->rows);
    A->rows = nullptr;
    free_tiled(A->tiled);
    A->tiled = nullptr;
    if (A->house) {
        dfree(A->house->cols);
        delete A->house;
        A->house = nullptr;
    }
    free_clique_cache(A->clique);
    A->clique = nullptr;
    return CSX_OK;
}

// ---- vectors ------------------------------------------------------------

static int vec_new(Kind k, int64_t len, size_t elem, csx_handle_t *out, Vec **pv) {
    CSX_TRY(require_ready());
    if (len < 0 || !out) return CSX_EINVAL;
    Vec *v = new Vec();
    v->len = len;
    int st = dmalloc(&v->d, (size_t)len * elem);
    if (st != CSX_OK) {
        delete v;
        return st;
    }
    *out = put(k, v);
    *pv = v;
    return CSX_OK;
}

int csx_vec_alloc(int64_t len, csx_handle_t *out) {
    Vec *v;
    CSX_TRY(vec_new(K_VEC, len, sizeof(double), out, &v));
    CSX_HIP(hipMemsetAsync(v->d, 0, (size_t)len * sizeof(double), ctx().stream));
    return CSX_OK;
}

int csx_vec_upload(const double *src, int64_t len, csx_handle_t *out) {
    if (!src && len > 0) return CSX_EINVAL;
    Vec *v;
    CSX_TRY(vec_new(K_VEC, len, sizeof(double), out, &v));
    if (len) {
        CSX_HIP(hipMemcpyAsync(v->d, src, (size_t)len * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
        CSX_HIP(hipStreamSynchronize(ctx().stream));
    }
    return CSX_OK;
}

int csx_vec_wrap(void *d_ptr, int64_t len, csx_handle_t *out) {
    CSX_TRY(require_ready());
    if (len < 0 || (!d_ptr && len > 0) || !out) return CSX_EINVAL;
    Vec *v = new Vec();
    v->len = len;
    v->d = d_ptr;
    v->owns = false;
    *out = put(K_VEC, v);
    return CSX_OK;
}

int csx_vec_download(csx_handle_t h, double *dst, int64_t len) {
    CSX_TRY(require_ready());
    Vec *v = vec(h);
    if (!v || len < 0 || len > v->len || (!dst && len > 0)) return CSX_EINVAL;
    if (len) CSX_HIP(hipMemcpyAsync(dst, v->d, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    return CSX_OK;
}

int csx_vec_write(csx_handle_t h, const double *src, int64_t len) {
    CSX_TRY(require_ready());
    Vec *v = vec(h);
    if (!v || len < 0 || len > v->len || (!src && len > 0)) return CSX_EINVAL;
    if (len) {
        CSX_HIP(hipMemcpyAsync(v->d, src, (size_t)len * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
        CSX_HIP(hipStreamSynchronize(ctx().stream));
    }
    return CSX_OK;
}

int csx_vec_fill(csx_handle_t h, double value) {
    CSX_TRY(require_ready());
    Vec *v = vec(h);
    if (!v) return CSX_EINVAL;
    if (v->len == 0) return CSX_OK;
    int64_t blocks = (v->len + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, (double *)v->d, v->len, value);
    CSX_LAUNCH_CHECK();
    return CSX_OK;
}

int csx_vec_copy(csx_handle_t hs, csx_handle_t hd) {
    CSX_TRY(require_ready());
    Vec *s = vec(hs), *d = vec(hd);
    if (!s || !d || d->len < s->len) return CSX_EINVAL;
    if (s->len)
        CSX_HIP(hipMemcpyAsync(d->d, s->d, (size_t)s->len * sizeof(double), hipMemcpyDeviceToDevice, ctx().stream));
    return CSX_OK;
}

int csx_vec_ptr(csx_handle_t h, void **d_ptr, int64_t *len) {
    Vec *v = vec(h);
    if (!v) v = ivec(h);
    if (!v) return CSX_EINVAL;
    if (d_ptr) *d_ptr = v->d;
    if (len) *len = v->len;
    return CSX_OK;
}

int csx_ivec_upload(const int32_t *src, int64_t len, csx_handle_t *out) {
    if (!src && len > 0) return CSX_EINVAL;
    Vec *v;
    CSX_TRY(vec_new(K_IVEC, len, sizeof(int32_t), out, &v));
    if (len) {
        CSX_HIP(hipMemcpyAsync(v->d, src, (size_t)len * sizeof(int32_t), hipMemcpyHostToDevice, ctx().stream));
        CSX_HIP(hipStreamSynchronize(ctx().stream));
    }
    return CSX_OK;
}

int csx_ivec_download(csx_handle_t h, int32_t *dst, int64_t len) {
    CSX_TRY(require_ready());
    Vec *v = ivec(h);
    if (!v || len < 0 || len > v->len || (!dst && len > 0)) return CSX_EINVAL;
    if (len) CSX_HIP(hipMemcpyAsync(dst, v->d, (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost, ctx().stream));
    CSX_HIP(hipStreamSynchronize(ctx().stream));
    return CSX_OK;
}

}  // extern "C"
