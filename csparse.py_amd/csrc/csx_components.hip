// Connected components of an index graph on the device, and grouping of the vertices by component.
// Used by the triangular-solve planner (csx_trisolve.hip: a factor that falls into many small independent
// blocks is solved one wave per block) and by the symbolic Cholesky analysis (csx_cholsym.hip: the elimination
// tree of a matrix with many small components is built one thread per component, csparse.py:1136-1169).
//
// Method: min-label hooking with atomicMin + pointer jumping.  parent[x] <= x always, so chains strictly
// descend and every walk ends; a hook attaches the larger of two roots under the smaller; rounds repeat until
// one passes without a hook (a handful for block-diagonal inputs).  The root of a component is its smallest
// vertex.
#include "csx_internal.h"
#include "csx_sweep.h"

namespace csx {

__global__ __launch_bounds__(256) void k_cc_init(int32_t n, int32_t *parent) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) parent[r] = (int32_t)r;
}

__device__ __forceinline__ int32_t cc_root(const int32_t *parent, int32_t r) {
    int32_t p = parent[r];
    while (p != r) {
        r = p;
        p = parent[r];
    }
    return r;
}

// one wave per row r, an edge (r, idx[q]) for q in [ptr[r] + sf, ptr[r + 1] - sl).  flags[0] |= some hook
// happened; flags[1] |= malformed: an index out of range, or (order 1) a neighbour >= r, (order 2) <= r
__global__ __launch_bounds__(256) void k_cc_hook(int32_t n, const int32_t *__restrict__ ptr,
                                                 const int32_t *__restrict__ idx, int sf, int sl, int order,
                                                 int32_t *parent, int *flags) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n) return;
    const int32_t b = ptr[r] + sf, e = ptr[r + 1] - sl;
    int32_t rr = -1;
    for (int32_t q = b + lane; q < e; q += 64) {
        const int32_t j = idx[q];
        if (j < 0 || j >= n || (order == 1 && j >= r) || (order == 2 && j <= r)) {
            flags[1] = 1;
            continue;
        }
        if (rr < 0) rr = cc_root(parent, (int32_t)r);
        int32_t a = rr, c = cc_root(parent, j);
        while (a != c) {                     // hook the larger root under the smaller one
            const int32_t hi = a > c ? a : c, lo = a > c ? c : a;
            const int32_t old = atomicMin(&parent[hi], lo);
            if (old == hi) {
                flags[0] = 1;
                break;
            }
            a = cc_root(parent, old < lo ? old : lo);   // somebody else re-parented hi: merge with that tree
            c = cc_root(parent, old < lo ? lo : old);
            flags[0] = 1;
        }
        rr = a < c ? a : c;
    }
}

__global__ __launch_bounds__(256) void k_cc_flatten(int32_t n, int32_t *parent) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) parent[r] = cc_root(parent, (int32_t)r);
}

__global__ __launch_bounds__(256) void k_iota_u32(int32_t n, uint32_t *v) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) v[r] = (uint32_t)r;
}

// sorted_root[k] starts a component when it differs from its left neighbour
__global__ __launch_bounds__(256) void k_cc_heads(int32_t n, const uint32_t *__restrict__ sroot, int32_t *head) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) head[k] = (k == 0 || sroot[k] != sroot[k - 1]) ? 1 : 0;
}

// component c starts at the c-th head; its size is the distance to the next head
__global__ __launch_bounds__(256) void k_cc_first(int32_t n, const int32_t *__restrict__ head,
                                                  const int32_t *__restrict__ hscan, Tree *comps) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && head[k]) comps[hscan[k]].first = (int32_t)k;
}

__global__ __launch_bounds__(256) void k_cc_count(int32_t n, int32_t ncomp, Tree *comps, int *stats) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncomp) return;
    const int32_t cnt = (c + 1 < ncomp ? comps[c + 1].first : n) - comps[c].first;
    comps[c].count = cnt;
    atomicMax(&stats[0], cnt);
}


__global__ __launch_bounds__(256) void k_comp_of_pos(int32_t n, const int32_t *__restrict__ head,
                                                     const int32_t *__restrict__ hscan, int32_t *comp_of_pos) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) comp_of_pos[k] = hscan[k] + head[k] - 1;
}

int connected_components(int32_t n, const int32_t *ptr, const int32_t *idx, int sf, int sl, int order, int32_t *root,
                         bool *malformed) {
    hipStream_t s = ctx().stream;
    *malformed = false;
    if (n == 0) return CSX_OK;
    DevScope tmp;
    int *flags = nullptr;
    CSX_TRY(tmp.alloc(&flags, 4));
    const unsigned nb = (unsigned)(((int64_t)n + 255) / 256), nbw = (unsigned)(((int64_t)n + 3) / 4);
    hipLaunchKernelGGL(k_cc_init, dim3(nb), dim3(256), 0, s, n, root);
    int hflags[2] = {0, 0};
    for (int it = 0; it < 64; it++) {
        CSX_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_cc_hook, dim3(nbw), dim3(256), 0, s, n, ptr, idx, sf, sl, order, root, flags);
        hipLaunchKernelGGL(k_cc_flatten, dim3(nb), dim3(256), 0, s, n, root);
        CSX_HIP(hipMemcpyAsync(hflags, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        CSX_HIP(hipStreamSynchronize(s));
        if (hflags[1]) {
            *malformed = true;
            return CSX_OK;
        }
        if (!hflags[0]) return CSX_OK;
    }
    set_error("connected_components: did not settle in 64 rounds");
    return CSX_ERUNTIME;
}

// nodes[k]: the vertices grouped by root, ascending inside a group (stable sort); comps[c] = (first, count)
// into nodes (allocated here, caller frees); comp_of_pos[k] = component of position k (optional).
int group_by_root(int32_t n, const int32_t *root, uint32_t *nodes, int32_t *comp_of_pos, Tree **comps_out,
                  int32_t *ncomp_out, int32_t *max_count) {
    hipStream_t s = ctx().stream;
    *comps_out = nullptr;
    *ncomp_out = 0;
    *max_count = 0;
    if (n == 0) return CSX_OK;
    DevScope tmp;
    uint32_t *iota = nullptr, *sroot = nullptr;
    int32_t *head = nullptr, *hscan = nullptr;
    int *stat = nullptr;
    CSX_TRY(tmp.alloc(&iota, (size_t)n));
    CSX_TRY(tmp.alloc(&sroot, (size_t)n));
    CSX_TRY(tmp.alloc(&head, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&hscan, (size_t)n + 1));
    CSX_TRY(tmp.alloc(&stat, 1));
    const unsigned nb = (unsigned)(((int64_t)n + 255) / 256);
    hipLaunchKernelGGL(k_iota_u32, dim3(nb), dim3(256), 0, s, n, iota);
    CSX_TRY(stable_sort_by_key((const uint32_t *)root, iota, nullptr, n, (uint32_t)n, sroot, nodes, nullptr));
    hipLaunchKernelGGL(k_cc_heads, dim3(nb), dim3(256), 0, s, n, sroot, head);
    int64_t ncomp = 0;
    CSX_TRY(scan_exclusive_i32(head, hscan, n, &ncomp));
    Tree *comps = nullptr;
    CSX_TRY(dalloc(&comps, (size_t)ncomp));
    *comps_out = comps;
    CSX_HIP(hipMemsetAsync(stat, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_cc_first, dim3(nb), dim3(256), 0, s, n, head, hscan, comps);
    hipLaunchKernelGGL(k_cc_count, dim3((unsigned)((ncomp + 255) / 256)), dim3(256), 0, s, n, (int32_t)ncomp, comps, stat);
    if (comp_of_pos) hipLaunchKernelGGL(k_comp_of_pos, dim3(nb), dim3(256), 0, s, n, head, hscan, comp_of_pos);
    int hmax = 0;
    CSX_HIP(hipMemcpyAsync(&hmax, stat, sizeof(int), hipMemcpyDeviceToHost, s));
    CSX_HIP(hipStreamSynchronize(s));
    *ncomp_out = (int32_t)ncomp;
    *max_count = hmax;
    return CSX_OK;
}

__global__ __launch_bounds__(256) void k_comp_size_key(const Tree *__restrict__ comps, int32_t ncomp, int32_t maxc, uint32_t *__restrict__ key,
                                                       uint32_t *__restrict__ id) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncomp) return;
    key[q] = (uint32_t)min(max(maxc - comps[q].count, 0), maxc);       // biggest first (clamped: a count outside [0, maxc] must not leave the key range)
    id[q] = (uint32_t)q;
}
__global__ __launch_bounds__(256) void k_comp_gather(const Tree *__restrict__ comps, const uint32_t *__restrict__ list, int32_t ncomp,
                                                     Tree *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < ncomp) out[q] = comps[list[q]];
}


int trees_biggest_first(const Tree *trees, int32_t ntrees, int32_t max_count, Tree **out) {
    *out = nullptr;
    if (ntrees <= 0) return CSX_OK;
    hipStream_t s = ctx().stream;
    DevScope tmp;
    uint32_t *key = nullptr, *id = nullptr, *list = nullptr;
    CSX_TRY(tmp.alloc(&key, (size_t)ntrees));
    CSX_TRY(tmp.alloc(&id, (size_t)ntrees));
    CSX_TRY(tmp.alloc(&list, (size_t)ntrees));
    Tree *o = nullptr;
    CSX_TRY(dalloc(&o, (size_t)ntrees));
    const unsigned g = (unsigned)((ntrees + 255) / 256);
    hipLaunchKernelGGL(k_comp_size_key, dim3(g), dim3(256), 0, s, trees, ntrees, max_count, key, id);
    int st = hipGetLastError() == hipSuccess ? CSX_OK : CSX_ERUNTIME;
    if (st == CSX_OK) st = stable_sort_by_key(key, id, nullptr, ntrees, (uint32_t)max_count + 1, nullptr, list, nullptr);
    if (st == CSX_OK) {
        hipLaunchKernelGGL(k_comp_gather, dim3(g), dim3(256), 0, s, trees, list, ntrees, o);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = CSX_ERUNTIME;     // (list is a temporary)
    }
    if (st != CSX_OK) {
        dfree(o);
        return st;
    }
    *out = o;
    return CSX_OK;
}

}  // namespace csx
