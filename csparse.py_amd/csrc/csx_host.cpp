// Host-side (CPU, C++) steps that feed the device hot path.  These are the
// integer / data-dependent phases the reference also runs before its hot loops:
//   csx_schol_host   cs_schol, natural order          (csparse.py:2051-2072)
//   symbolic_fill    pattern of L from cs_ereach walks (csparse.py:1094-1131, 606-617)
//   csx_lu_host      cs_lu with partial pivoting       (csparse.py:1370-1451, 2078-2113)
// They are written from the algorithms, not transcribed: the elimination tree
// uses the same ancestor path compression idea, the column counts come from the
// row-pattern walk (O(|L|)) that the numeric phase needs anyway.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "csx_internal.h"

namespace csx {

// Elimination tree of the symmetric matrix whose upper triangle is (Ap, Ai)
// (entries with row > col are ignored).  With a permutation, node ids are the
// permuted ones: entry (i, j) of A is entry (min, max) of (pinv[i], pinv[j]).
// up_ptr/up_idx receive the upper-triangular pattern of the permuted matrix by column.
void upper_pattern(int32_t n, const int32_t *Ap, const int32_t *Ai, const int32_t *pinv, std::vector<int32_t> &up_ptr,
                   std::vector<int32_t> &up_idx) {
    up_ptr.assign((size_t)n + 1, 0);
    for (int32_t j = 0; j < n; j++) {
        const int32_t j2 = pinv ? pinv[j] : j;
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int32_t i = Ai[p];
            if (i > j) continue;
            const int32_t i2 = pinv ? pinv[i] : i;
            up_ptr[(size_t)std::max(i2, j2) + 1]++;
        }
    }
    for (int32_t j = 0; j < n; j++) up_ptr[(size_t)j + 1] += up_ptr[(size_t)j];
    up_idx.resize((size_t)up_ptr[(size_t)n]);
    std::vector<int32_t> fill(up_ptr.begin(), up_ptr.end() - 1);
    for (int32_t j = 0; j < n; j++) {
        const int32_t j2 = pinv ? pinv[j] : j;
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int32_t i = Ai[p];
            if (i > j) continue;
            const int32_t i2 = pinv ? pinv[i] : i;
            up_idx[(size_t)fill[(size_t)std::max(i2, j2)]++] = std::min(i2, j2);
        }
    }
}

void etree_of_upper(int32_t n, const std::vector<int32_t> &up_ptr, const std::vector<int32_t> &up_idx,
                    std::vector<int32_t> &parent) {
    parent.assign((size_t)n, -1);
    std::vector<int32_t> anc((size_t)n, -1);
    for (int32_t k = 0; k < n; k++) {
        for (int32_t p = up_ptr[(size_t)k]; p < up_ptr[(size_t)k + 1]; p++) {
            int32_t i = up_idx[(size_t)p];
            while (i != -1 && i < k) {
                const int32_t next = anc[(size_t)i];
                anc[(size_t)i] = k;
                if (next == -1) parent[(size_t)i] = k;
                i = next;
            }
        }
    }
}

// the same from a full CSC pattern: only entries above the diagonal of each column take part
void etree_of_csc(int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent) {
    std::vector<int32_t> anc((size_t)n, -1);
    for (int32_t k = 0; k < n; k++) parent[k] = -1;
    for (int32_t k = 0; k < n; k++) {
        for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) {
            int32_t i = Ai[p];
            while (i != -1 && i < k) {
                const int32_t next = anc[(size_t)i];
                anc[(size_t)i] = k;
                if (next == -1) parent[i] = k;
                i = next;
            }
        }
    }
}

// Row patterns of L below the diagonal: for row k the columns i < k with L(k,i) != 0 are
// the nodes on the etree paths from the entries of column k of the upper pattern up to k.
// visit(k, i) is called once per such pair, k ascending.
template <class Visit>
static void walk_row_patterns(int32_t n, const std::vector<int32_t> &up_ptr, const std::vector<int32_t> &up_idx,
                              const int32_t *parent, Visit visit) {
    std::vector<int32_t> mark((size_t)n, -1);
    for (int32_t k = 0; k < n; k++) {
        mark[(size_t)k] = k;
        for (int32_t p = up_ptr[(size_t)k]; p < up_ptr[(size_t)k + 1]; p++) {
            for (int32_t i = up_idx[(size_t)p]; i != -1 && i < k && mark[(size_t)i] != k; i = parent[i]) {
                mark[(size_t)i] = k;
                visit(k, i);
            }
        }
    }
}

// ---- a fill-reducing ordering that suits level scheduling: nested dissection --------------------------
// The reference's cs_amd does not run, so order = 1 has no answer to match; what the
// device wants from an ordering is a BUSHY elimination tree (wide levels), and nested dissection gives
// exactly that: split the graph of A + A' by a vertex separator, order the two halves first (recursively,
// they are independent subtrees) and the separator last.  Separators here are the middle level of a
// breadth-first level structure rooted at a pseudo-peripheral vertex (two sweeps) -- crude but O(nnz log n),
// deterministic, and good on the mesh-like matrices sparse Cholesky is used for.  Parts of <= ND_LEAF
// vertices, and parts whose level structure is too shallow to cut, are numbered in breadth-first order.
// perm[k] = the original index of the k-th row/column of P A P' (cs_amd's convention, csparse.py:214).
namespace {
constexpr int32_t ND_LEAF = 96;

struct NdGraph {
    int32_t n;
    std::vector<int32_t> ptr, adj;
};

void nd_build_graph(int32_t n, const int32_t *Ap, const int32_t *Ai, NdGraph &G) {
    G.n = n;
    G.ptr.assign((size_t)n + 1, 0);
    for (int32_t j = 0; j < n; j++)
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int32_t i = Ai[p];
            if (i == j) continue;
            G.ptr[(size_t)i + 1]++;
            G.ptr[(size_t)j + 1]++;
        }
    for (int32_t v = 0; v < n; v++) G.ptr[(size_t)v + 1] += G.ptr[(size_t)v];
    G.adj.assign((size_t)G.ptr[(size_t)n], 0);
    std::vector<int32_t> fill(G.ptr.begin(), G.ptr.end() - 1);
    for (int32_t j = 0; j < n; j++)
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int32_t i = Ai[p];
            if (i == j) continue;
            G.adj[(size_t)fill[(size_t)i]++] = j;
            G.adj[(size_t)fill[(size_t)j]++] = i;
        }
}

// breadth-first search inside the part `tag` (part[v] == tag); fills order[] / level_start[]; returns the number of
// levels.  part / mark are shared by the worker threads: a thread writes only vertices of the part it owns, and reads
// of other parts' entries only ever compare unequal (tags and stamps are unique), hence relaxed atomics.
using AtomicVec = std::unique_ptr<std::atomic<int32_t>[]>;
inline int32_t ld(const AtomicVec &a, int32_t i) { return a[(size_t)i].load(std::memory_order_relaxed); }
inline void st(AtomicVec &a, int32_t i, int32_t v) { a[(size_t)i].store(v, std::memory_order_relaxed); }

int32_t nd_bfs(const NdGraph &G, const AtomicVec &part, int32_t tag, int32_t root, AtomicVec &mark, int32_t stamp,
               std::vector<int32_t> &order, std::vector<int32_t> &level_start) {
    order.clear();
    level_start.clear();
    order.push_back(root);
    st(mark, root, stamp);
    size_t head = 0;
    while (head < order.size()) {
        level_start.push_back((int32_t)head);
        const size_t end = order.size();
        for (; head < end; head++) {
            const int32_t v = order[head];
            for (int32_t q = G.ptr[(size_t)v]; q < G.ptr[(size_t)v + 1]; q++) {
                const int32_t u = G.adj[(size_t)q];
                if (ld(part, u) == tag && ld(mark, u) != stamp) {
                    st(mark, u, stamp);
                    order.push_back(u);
                }
            }
        }
    }
    level_start.push_back((int32_t)order.size());
    return (int32_t)level_start.size() - 1;
}

struct NdJob {
    int32_t tag, out_lo;             // the part and where its vertices go in perm
    std::vector<int32_t> verts;
};

// The recursion as a pool of jobs: the two halves of a cut are independent, so after a few levels every worker has
// parts of its own.  The ordering does not depend on which thread takes which part (a part's numbering is a function
// of its vertex list and the graph alone).
struct NdPool {
    const NdGraph &G;
    int32_t *perm;
    AtomicVec part, mark;
    std::atomic<int32_t> next_tag{1}, stamp{0};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<NdJob> jobs;
    int pending = 0;                 // jobs queued or being worked on

    NdPool(const NdGraph &g, int32_t *p) : G(g), perm(p) {
        part.reset(new std::atomic<int32_t>[(size_t)std::max(g.n, 1)]);
        mark.reset(new std::atomic<int32_t>[(size_t)std::max(g.n, 1)]);
        for (int32_t v = 0; v < g.n; v++) {
            st(part, v, 0);
            st(mark, v, -1);
        }
    }
    void push(NdJob &&j) {
        std::lock_guard<std::mutex> lock(mu);
        jobs.push_back(std::move(j));
        pending++;
        cv.notify_one();
    }
    bool pop(NdJob &j) {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return !jobs.empty() || pending == 0; });
        if (jobs.empty()) return false;
        j = std::move(jobs.back());
        jobs.pop_back();
        return true;
    }
    void finished() {
        std::lock_guard<std::mutex> lock(mu);
        if (--pending == 0) cv.notify_all();
    }
    void work() {
        std::vector<int32_t> order, level_start;
        NdJob job;
        while (pop(job)) {
            run(job, order, level_start);
            finished();
        }
    }
    void run(NdJob &job, std::vector<int32_t> &order, std::vector<int32_t> &level_start) {
        const int32_t sz = (int32_t)job.verts.size();
        if (sz == 0) return;
        // one connected component at a time: the rest goes back to the pool as its own part
        int32_t stp = ++stamp;
        nd_bfs(G, part, job.tag, job.verts[0], mark, stp, order, level_start);
        if ((int32_t)order.size() < sz) {
            NdJob rest;
            rest.tag = next_tag++;
            rest.out_lo = job.out_lo + (int32_t)order.size();
            for (int32_t v : job.verts)
                if (ld(mark, v) != stp) {
                    rest.verts.push_back(v);
                    st(part, v, rest.tag);
                }
            push(std::move(rest));
        }
        const int32_t csz = (int32_t)order.size();
        // pseudo-peripheral root: restart from a vertex of the last level (smallest degree), twice
        for (int sweep = 0; sweep < 2 && csz > ND_LEAF; sweep++) {
            const int32_t nl = (int32_t)level_start.size() - 1;
            int32_t far = order[(size_t)level_start[(size_t)nl - 1]];
            for (int32_t q = level_start[(size_t)nl - 1]; q < level_start[(size_t)nl]; q++) {
                const int32_t v = order[(size_t)q];
                if (G.ptr[(size_t)v + 1] - G.ptr[(size_t)v] < G.ptr[(size_t)far + 1] - G.ptr[(size_t)far]) far = v;
            }
            stp = ++stamp;
            nd_bfs(G, part, job.tag, far, mark, stp, order, level_start);
        }
        const int32_t nl = (int32_t)level_start.size() - 1;
        if (csz <= ND_LEAF || nl < 3) {   // leaf (or too shallow to cut): breadth-first numbering
            for (int32_t q = 0; q < csz; q++) perm[job.out_lo + q] = order[(size_t)q];
            return;
        }
        // separator = the level at which half of the component has been passed (never the first or last level)
        int32_t s = 1;
        while (s < nl - 2 && level_start[(size_t)s + 1] < csz / 2) s++;
        const int32_t a_end = level_start[(size_t)s], b_begin = level_start[(size_t)s + 1];
        // thin the separator: only the vertices of level s that touch level s + 1 have to be in it; the others
        // stay with the first half (every path into the second half leaves level s through such a vertex)
        const int32_t next_mark = -2 - stp;
        for (int32_t q = b_begin; q < level_start[(size_t)s + 2]; q++) st(mark, order[(size_t)q], next_mark);  // level s+1
        NdJob A, B;
        A.tag = next_tag++;
        B.tag = next_tag++;
        A.verts.assign(order.begin(), order.begin() + a_end);
        std::vector<int32_t> sep;
        for (int32_t q = a_end; q < b_begin; q++) {
            const int32_t v = order[(size_t)q];
            bool touches = false;
            for (int32_t e = G.ptr[(size_t)v]; e < G.ptr[(size_t)v + 1] && !touches; e++)
                touches = ld(mark, G.adj[(size_t)e]) == next_mark;
            if (touches) sep.push_back(v);
            else A.verts.push_back(v);
        }
        B.verts.assign(order.begin() + b_begin, order.end());
        A.out_lo = job.out_lo;
        B.out_lo = job.out_lo + (int32_t)A.verts.size();
        for (int32_t v : A.verts) st(part, v, A.tag);
        for (int32_t v : B.verts) st(part, v, B.tag);
        int32_t out = B.out_lo + (int32_t)B.verts.size();     // the separator goes last
        for (int32_t v : sep) {
            st(part, v, -1);
            perm[out++] = v;
        }
        push(std::move(A));
        push(std::move(B));
    }
};
}  // namespace

}  // namespace csx

using namespace csx;

extern "C" int csx_order_nd_host(int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *perm) {
    if (n < 0 || !Ap || (!Ai && Ap[n] > 0) || !perm) return CSX_EINVAL;
    for (int32_t j = 0; j < n; j++)
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] < 0 || Ai[p] >= n) return CSX_EINVAL;
    if (n == 0) return CSX_OK;
    NdGraph G;
    nd_build_graph(n, Ap, Ai, G);
    NdPool pool(G, perm);
    {
        NdJob all;
        all.tag = 0;
        all.out_lo = 0;
        all.verts.resize((size_t)n);
        for (int32_t v = 0; v < n; v++) all.verts[(size_t)v] = v;
        pool.push(std::move(all));
    }
    // small problems: one thread (the pool costs more than it saves); else up to eight workers
    unsigned workers = 1;
    if (n >= 20000 || Ap[n] >= 200000) workers = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::thread> threads;
    for (unsigned t = 1; t < workers; t++) threads.emplace_back([&pool] { pool.work(); });
    pool.work();
    for (auto &t : threads) t.join();
    return CSX_OK;
}

namespace csx {

}  // namespace csx

using namespace csx;

// cs_counts (csparse.py:703-764, with cs_leaf :1280-1304 and the row lists of _init_ata :677-700): the column counts of
// chol(A) (ata = 0: A square, its upper triangle used) or of chol(A'A) (ata != 0: A m-by-n) from the elimination tree
// `parent` and its postorder `post` (both n entries, the caller's -- cs_etree / cs_post), without forming L.  For every node j
// in postorder the rows of the columns filed under j (ata: the rows whose leftmost column, by postorder rank, is j; else
// column j itself) are run through the skeleton test: A(i, j) counts where j is a leaf of row i's subtree, and the least
// common ancestor of two consecutive leaves loses the overlap; the deltas are then summed up the tree.
extern "C" int csx_counts_host(int32_t m, int32_t n, const int32_t *Ap, const int32_t *Ai, const int32_t *parent,
                               const int32_t *post, int ata, int32_t *colcount) {
    if (m < 0 || n < 0 || !Ap || (!Ai && Ap[n] > 0) || !parent || !post || !colcount) return CSX_EINVAL;
    if (!ata && m != n) return CSX_EINVAL;
    const int64_t nnz = Ap[n];
    for (int64_t p = 0; p < nnz; p++)
        if (Ai[p] < 0 || Ai[p] >= m) return CSX_EINVAL;
    {   // parent must be a forest on n nodes with parents above their children's... any value in [-1, n); post a permutation
        std::vector<char> seen((size_t)n, 0);
        for (int32_t k = 0; k < n; k++) {
            if (parent[k] < -1 || parent[k] >= n || post[k] < 0 || post[k] >= n || seen[(size_t)post[k]]) return CSX_EINVAL;
            seen[(size_t)post[k]] = 1;
        }
    }
    // the pattern of A' (row i of A lists its columns in ascending order): cs_transpose(A, False), csparse.py:721
    std::vector<int32_t> ATp((size_t)m + 1, 0), ATi((size_t)nnz);
    for (int64_t p = 0; p < nnz; p++) ATp[(size_t)Ai[p] + 1]++;
    for (int32_t i = 0; i < m; i++) ATp[(size_t)i + 1] += ATp[(size_t)i];
    {
        std::vector<int32_t> fill(ATp.begin(), ATp.end() - 1);
        for (int32_t k = 0; k < n; k++)
            for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) ATi[(size_t)fill[(size_t)Ai[p]]++] = k;
    }
    std::vector<int32_t> first((size_t)n, -1), maxfirst((size_t)n, -1), prevleaf((size_t)n, -1), anc((size_t)n);
    int32_t *delta = colcount;
    for (int32_t k = 0; k < n; k++) {                 // first[j] = postorder rank of the first descendant of j
        int32_t j = post[k];
        delta[j] = first[(size_t)j] == -1 ? 1 : 0;    // 1 for a leaf
        int32_t guard = 0;
        while (j != -1 && first[(size_t)j] == -1) {
            first[(size_t)j] = k;
            j = parent[j];
            if (++guard > n) return CSX_EINVAL;       // a cycle in `parent`
        }
    }
    std::vector<int32_t> head, nxt;
    if (ata) {                                        // _init_ata: row i filed under the smallest postorder rank among its columns
        std::vector<int32_t> rank((size_t)n, 0);
        head.assign((size_t)n + 1, -1);
        nxt.assign((size_t)m, -1);
        for (int32_t k = 0; k < n; k++) rank[(size_t)post[k]] = k;
        for (int32_t i = 0; i < m; i++) {
            int32_t k = n;
            for (int32_t p = ATp[(size_t)i]; p < ATp[(size_t)i + 1]; p++) k = std::min(k, rank[(size_t)ATi[(size_t)p]]);
            nxt[(size_t)i] = head[(size_t)k];
            head[(size_t)k] = i;
        }
    }
    for (int32_t j = 0; j < n; j++) anc[(size_t)j] = j;
    for (int32_t k = 0; k < n; k++) {
        const int32_t j = post[k];
        if (parent[j] != -1) delta[parent[j]]--;      // j is not a root
        for (int32_t J = ata ? head[(size_t)k] : j; J != -1; J = ata ? nxt[(size_t)J] : -1) {
            for (int32_t p = ATp[(size_t)J]; p < ATp[(size_t)J + 1]; p++) {
                const int32_t i = ATi[(size_t)p];
                if (i <= j || first[(size_t)j] <= maxfirst[(size_t)i]) continue;      // cs_leaf: j is not a leaf of row i's subtree
                maxfirst[(size_t)i] = first[(size_t)j];
                const int32_t jprev = prevleaf[(size_t)i];
                prevleaf[(size_t)i] = j;
                delta[j]++;                           // A(i, j) is in the skeleton
                if (jprev != -1) {                    // a later leaf: the least common ancestor of the two loses the overlap
                    int32_t q = jprev;
                    while (q != anc[(size_t)q]) q = anc[(size_t)q];
                    for (int32_t s = jprev; s != q;) {
                        const int32_t sp = anc[(size_t)s];
                        anc[(size_t)s] = q;
                        s = sp;
                    }
                    delta[q]--;
                }
            }
        }
        if (parent[j] != -1) anc[(size_t)j] = parent[j];
    }
    for (int32_t j = 0; j < n; j++)
        if (parent[j] != -1) colcount[parent[j]] += colcount[j];
    return CSX_OK;
}

extern "C" int csx_schol_host(int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent, int32_t *cp) {
    if (n < 0 || !Ap || (!Ai && Ap[n] > 0) || !parent || !cp) return CSX_EINVAL;
    for (int32_t j = 0; j < n; j++)
        for (int32_t p = Ap[j]; p < Ap[j + 1]; p++)
            if (Ai[p] < 0 || Ai[p] >= n) return CSX_EINVAL;
    std::vector<int32_t> up_ptr, up_idx, par;
    upper_pattern(n, Ap, Ai, nullptr, up_ptr, up_idx);
    etree_of_upper(n, up_ptr, up_idx, par);
    std::vector<int32_t> count((size_t)n, 1);  // the diagonal
    walk_row_patterns(n, up_ptr, up_idx, par.data(), [&](int32_t, int32_t i) { count[(size_t)i]++; });
    int64_t run = 0;
    for (int32_t j = 0; j < n; j++) {
        parent[j] = par[(size_t)j];
        cp[j] = (int32_t)run;
        run += count[(size_t)j];
        if (run > 2147483647ll) return CSX_EINVAL;
    }
    cp[n] = (int32_t)run;
    return CSX_OK;
}

// ---- LU, left-looking, threshold partial pivoting, natural column order -----------------
// Output: L (unit diagonal first in each column, row indices in pivot order), U (diagonal last),
// pinv (row i of A is row pinv[i] of L U).  Arrays are malloc'ed; free with csx_host_free.
extern "C" int csx_lu_host(int32_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax, double tol,
                           int32_t **Lp_out, int32_t **Li_out, double **Lx_out, int32_t **Up_out, int32_t **Ui_out,
                           double **Ux_out, int32_t *pinv) {
    if (n < 0 || !Ap || !Ai || !Ax || !Lp_out || !Li_out || !Lx_out || !Up_out || !Ui_out || !Ux_out || !pinv)
        return CSX_EINVAL;
    std::vector<int32_t> Lp((size_t)n + 1, 0), Up((size_t)n + 1, 0), Li, Ui;
    std::vector<double> Lx, Ux, x((size_t)n, 0.0);
    std::vector<int32_t> reach((size_t)n), stack((size_t)n), pos((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    for (int32_t i = 0; i < n; i++) pinv[i] = -1;
    for (int32_t k = 0; k < n; k++) {
        Lp[(size_t)k] = (int32_t)Li.size();
        Up[(size_t)k] = (int32_t)Ui.size();
        // reach of A(:,k) in the graph of L (rows already pivotal map to columns of L): DFS,
        // topological order in reach[top..n-1]
        int32_t top = n;
        for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) {
            if (seen[(size_t)Ai[p]]) continue;
            int32_t head = 0;
            stack[0] = Ai[p];
            while (head >= 0) {
                const int32_t j = stack[(size_t)head];
                const int32_t col = pinv[j];
                if (!seen[(size_t)j]) {
                    seen[(size_t)j] = 1;
                    pos[(size_t)head] = col < 0 ? 0 : Lp[(size_t)col];
                }
                bool done = true;
                const int32_t end = col < 0 ? 0 : (col == k ? (int32_t)Li.size() : Lp[(size_t)col + 1]);
                for (int32_t q = pos[(size_t)head]; q < end; q++) {
                    const int32_t i = Li[(size_t)q];
                    if (seen[(size_t)i]) continue;
                    pos[(size_t)head] = q;
                    stack[(size_t)++head] = i;
                    done = false;
                    break;
                }
                if (done) {
                    head--;
                    reach[(size_t)--top] = j;
                }
            }
        }
        for (int32_t p = top; p < n; p++) {
            seen[(size_t)reach[(size_t)p]] = 0;
            x[(size_t)reach[(size_t)p]] = 0.0;
        }
        for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) x[(size_t)Ai[p]] = Ax[p];
        // sparse triangular solve x = L \ A(:,k) along the reach
        for (int32_t px = top; px < n; px++) {
            const int32_t j = reach[(size_t)px], col = pinv[j];
            if (col < 0) continue;
            x[(size_t)j] /= Lx[(size_t)Lp[(size_t)col]];
            const double xj = x[(size_t)j];
            for (int32_t q = Lp[(size_t)col] + 1; q < Lp[(size_t)col + 1]; q++) x[(size_t)Li[(size_t)q]] -= Lx[(size_t)q] * xj;
        }
        // pivot search among non-pivotal rows; pivotal rows go to U
        int32_t ipiv = -1;
        double a = -1.0;
        for (int32_t p = top; p < n; p++) {
            const int32_t i = reach[(size_t)p];
            if (pinv[i] < 0) {
                const double t = std::fabs(x[(size_t)i]);
                if (t > a) {
                    a = t;
                    ipiv = i;
                }
            } else {
                Ui.push_back(pinv[i]);
                Ux.push_back(x[(size_t)i]);
            }
        }
        if (ipiv == -1 || a <= 0) return CSX_ENOTSPD;  // singular: the reference returns None (csparse.py:1423)
        if (pinv[k] < 0 && std::fabs(x[(size_t)k]) >= a * tol) ipiv = k;
        const double pivot = x[(size_t)ipiv];
        Ui.push_back(k);
        Ux.push_back(pivot);
        pinv[ipiv] = k;
        Li.push_back(ipiv);
        Lx.push_back(1.0);
        for (int32_t p = top; p < n; p++) {
            const int32_t i = reach[(size_t)p];
            if (pinv[i] < 0) {
                Li.push_back(i);
                Lx.push_back(x[(size_t)i] / pivot);
            }
            x[(size_t)i] = 0.0;
        }
        // L's column k is complete only now; DFS above used Li.size() as its end while building
        Lp[(size_t)k + 1] = (int32_t)Li.size();
    }
    Lp[(size_t)n] = (int32_t)Li.size();
    Up[(size_t)n] = (int32_t)Ui.size();
    for (auto &r : Li) r = pinv[r];
    auto dup_i = [](const std::vector<int32_t> &v) {
        int32_t *p = (int32_t *)std::malloc((v.size() + 1) * sizeof(int32_t));
        if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(int32_t));
        return p;
    };
    auto dup_d = [](const std::vector<double> &v) {
        double *p = (double *)std::malloc((v.size() + 1) * sizeof(double));
        if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(double));
        return p;
    };
    *Lp_out = dup_i(Lp);
    *Li_out = dup_i(Li);
    *Lx_out = dup_d(Lx);
    *Up_out = dup_i(Up);
    *Ui_out = dup_i(Ui);
    *Ux_out = dup_d(Ux);
    return CSX_OK;
}

extern "C" void csx_host_free(void *p) { std::free(p); }

// ---- sparse Householder QR, numeric phase (csparse.py:1797-1870 with :1216-1261) -------------------------------
// Left-looking: column k of R and the Householder vector V(:,k) come from column q[k] of A after the earlier
// reflections that touch it have been applied.  Which ones touch it is read off the column elimination tree:
// every nonzero row i of the column starts a walk from leftmost[i] (the first column that has an entry in row i)
// towards the root; the walk stops at a column already collected for this k.  The collected columns, each path
// kept in root-ward order and later paths placed in front of earlier ones, are the nonzero pattern of R(:,k) in
// an order in which every reflection precedes its ancestors -- the order the reflections are applied in and the
// order the entries of R(:,k) are stored in (diagonal last).  V(:,k)'s pattern is the permuted rows of the column
// below k together with the patterns of the V columns whose tree parent is k.
//
// Symbolic input from cs_sqr: parent (column etree of A'A), pinv (row permutation, length m2), leftmost (length m),
// m2 (rows incl. fictitious ones), capacities vcap / rcap.  Output arrays are the caller's (sized by cs_sqr's counts).
extern "C" int csx_qr_host(int32_t m, int32_t n, int32_t m2, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                           const int32_t *q, const int32_t *parent, const int32_t *pinv, const int32_t *leftmost,
                           int32_t vcap, int32_t rcap, int32_t *Vp, int32_t *Vi, double *Vx, int32_t *Rp, int32_t *Ri,
                           double *Rx, double *beta) {
    if (m < 0 || n < 0 || m2 < m || !Ap || !Ai || !Ax || !parent || !pinv || !leftmost || !Vp || !Vi || !Vx || !Rp ||
        !Ri || !Rx || !beta)
        return CSX_EINVAL;
    std::vector<int32_t> stamp((size_t)m2, -1);      // stamp[r] == k: row / column r already collected for column k
    std::vector<int32_t> order((size_t)n > 0 ? (size_t)n : 1), path((size_t)n > 0 ? (size_t)n : 1);
    std::vector<double> work((size_t)m2, 0.0);       // the column being reduced, in permuted row numbering
    int32_t vnz = 0, rnz = 0;
    // y <- (I - b v v') y for a stored reflection
    auto reflect = [&](int32_t col, double b) {
        double dot = 0.0;
        for (int32_t p = Vp[col]; p < Vp[col + 1]; p++) dot += Vx[p] * work[(size_t)Vi[p]];
        dot *= b;
        for (int32_t p = Vp[col]; p < Vp[col + 1]; p++) work[(size_t)Vi[p]] -= Vx[p] * dot;
    };
    for (int32_t k = 0; k < n; k++) {
        Rp[k] = rnz;
        Vp[k] = vnz;
        const int32_t vstart = vnz;
        if (vnz >= vcap) return CSX_EINVAL;
        stamp[(size_t)k] = k;                        // the diagonal position leads V(:,k)
        Vi[vnz++] = k;
        int32_t front = n;                           // order[front..n) = reflections to apply, children before parents
        const int32_t col = q ? q[k] : k;
        for (int32_t p = Ap[col]; p < Ap[col + 1]; p++) {
            int32_t len = 0;
            for (int32_t c = leftmost[Ai[p]]; stamp[(size_t)c] != k; c = parent[c]) {
                path[(size_t)len++] = c;
                stamp[(size_t)c] = k;
            }
            while (len > 0) order[(size_t)--front] = path[(size_t)--len];
            const int32_t r = pinv[Ai[p]];
            work[(size_t)r] = Ax[p];
            if (r > k && stamp[(size_t)r] < k) {     // a row below the diagonal not yet in V(:,k)'s pattern
                if (vnz >= vcap) return CSX_EINVAL;
                Vi[vnz++] = r;
                stamp[(size_t)r] = k;
            }
        }
        for (int32_t t = front; t < n; t++) {
            const int32_t c = order[(size_t)t];
            reflect(c, beta[c]);
            if (rnz >= rcap) return CSX_EINVAL;
            Ri[rnz] = c;
            Rx[rnz++] = work[(size_t)c];
            work[(size_t)c] = 0.0;
            if (parent[c] == k) {                    // V(:,c)'s rows below c pass on to V(:,k)
                for (int32_t p = Vp[c]; p < Vp[c + 1]; p++) {
                    const int32_t r = Vi[p];
                    if (stamp[(size_t)r] < k) {
                        if (vnz >= vcap) return CSX_EINVAL;
                        stamp[(size_t)r] = k;
                        Vi[vnz++] = r;
                    }
                }
            }
        }
        for (int32_t p = vstart; p < vnz; p++) {
            Vx[p] = work[(size_t)Vi[p]];
            work[(size_t)Vi[p]] = 0.0;
        }
        // Householder vector of Vx[vstart..vnz): afterwards (I - beta v v') x = s e1 (csparse.py:1238-1261)
        double tail2 = 0.0;
        for (int32_t p = vstart + 1; p < vnz; p++) tail2 += Vx[p] * Vx[p];
        const double head = Vx[vstart];
        double s, b;
        if (tail2 == 0.0) {
            s = std::fabs(head);
            b = head <= 0.0 ? 2.0 : 0.0;
            Vx[vstart] = 1.0;
        } else {
            s = std::sqrt(head * head + tail2);
            Vx[vstart] = head <= 0.0 ? head - s : -tail2 / (head + s);
            b = -1.0 / (s * Vx[vstart]);
        }
        if (rnz >= rcap) return CSX_EINVAL;
        Ri[rnz] = k;
        Rx[rnz++] = s;
        beta[k] = b;
    }
    Rp[n] = rnz;
    Vp[n] = vnz;
    return CSX_OK;
}

// x <- Q' x (transpose = 1: reflections 0 .. n-1 in order) or x <- Q x (transpose = 0: n-1 .. 0), csparse.py:1216-1235
extern "C" int csx_qr_apply_host(int32_t n, const int32_t *Vp, const int32_t *Vi, const double *Vx, const double *beta,
                                 int transpose, double *x) {
    if (n < 0 || !Vp || !Vi || !Vx || !beta || !x) return CSX_EINVAL;
    for (int32_t t = 0; t < n; t++) {
        const int32_t k = transpose ? t : n - 1 - t;
        double dot = 0.0;
        for (int32_t p = Vp[k]; p < Vp[k + 1]; p++) dot += Vx[p] * x[Vi[p]];
        dot *= beta[k];
        for (int32_t p = Vp[k]; p < Vp[k + 1]; p++) x[Vi[p]] -= Vx[p] * dot;
    }
    return CSX_OK;
}

// cs_sqr for QR with the natural column order (csparse.py:2187-2217): the column elimination tree (cs_etree of A'A,
// :1136-1169), its postorder (:1711-1742), the column counts of R = chol(A'A) (cs_counts with ata, :703-764 and
// :677-700) and cs_vcount (:2118-2184: leftmost[], the row permutation pinv, the rows of V, m2 with the
// fictitious rows of a structurally rank-deficient A).  Written from the algorithms; the same results as the
// Python versions in csparse.py, which stay as the list-level functions cs_etree / cs_post.
// parent, cp: n entries; pinv: m + n; leftmost: m.
extern "C" int csx_sqr_host(int32_t m, int32_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent, int32_t *cp,
                            int32_t *pinv, int32_t *leftmost, int32_t *m2_out, int64_t *vnz_out, int64_t *rnz_out) {
    if (m < 0 || n < 0 || !Ap || !Ai || !parent || !cp || !pinv || !leftmost || !m2_out || !vnz_out || !rnz_out)
        return CSX_EINVAL;
    const int64_t nnz = Ap[n];
    for (int64_t p = 0; p < nnz; p++)
        if (Ai[p] < 0 || Ai[p] >= m) return CSX_EINVAL;
    // ---- column elimination tree: the tree of A'A without forming it (a row links the columns it touches) ----
    {
        std::vector<int32_t> anc((size_t)n, -1), prev((size_t)m, -1);
        for (int32_t k = 0; k < n; k++) {
            parent[k] = -1;
            for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) {
                int32_t i = prev[(size_t)Ai[p]];
                while (i != -1 && i < k) {
                    const int32_t up = anc[(size_t)i];
                    anc[(size_t)i] = k;
                    if (up == -1) parent[i] = k;
                    i = up;
                }
                prev[(size_t)Ai[p]] = k;
            }
        }
    }
    // ---- postorder (children in ascending order, roots in ascending order) ----
    std::vector<int32_t> post;
    post.reserve((size_t)n);
    {
        std::vector<int32_t> first_child((size_t)n, -1), sibling((size_t)n, -1), stack;
        for (int32_t j = n - 1; j >= 0; j--)
            if (parent[j] != -1) {
                sibling[(size_t)j] = first_child[(size_t)parent[j]];
                first_child[(size_t)parent[j]] = j;
            }
        for (int32_t root = 0; root < n; root++) {
            if (parent[root] != -1) continue;
            stack.assign(1, root);
            while (!stack.empty()) {
                const int32_t top = stack.back();
                const int32_t c = first_child[(size_t)top];
                if (c == -1) {
                    post.push_back(top);
                    stack.pop_back();
                } else {
                    first_child[(size_t)top] = sibling[(size_t)c];
                    stack.push_back(c);
                }
            }
        }
    }
    // ---- rows of A (pattern of A'), by a counting sort: row i lists its columns in ascending order ----
    std::vector<int32_t> ATp((size_t)m + 1, 0), ATi((size_t)nnz);
    for (int64_t p = 0; p < nnz; p++) ATp[(size_t)Ai[p] + 1]++;
    for (int32_t i = 0; i < m; i++) ATp[(size_t)i + 1] += ATp[(size_t)i];
    {
        std::vector<int32_t> fill(ATp.begin(), ATp.end() - 1);
        for (int32_t k = 0; k < n; k++)
            for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) ATi[(size_t)fill[(size_t)Ai[p]]++] = k;
    }
    // ---- column counts of chol(A'A): rows grouped by the postorder rank of their leftmost column, then the
    //      skeleton / leaf counting over those rows ----
    {
        std::vector<int32_t> rank((size_t)n, 0), head((size_t)n + 1, -1), nxt((size_t)m, -1);
        for (int32_t k = 0; k < n; k++) rank[(size_t)post[(size_t)k]] = k;
        for (int32_t i = 0; i < m; i++) {
            int32_t k = n;
            for (int32_t p = ATp[(size_t)i]; p < ATp[(size_t)i + 1]; p++) k = std::min(k, rank[(size_t)ATi[(size_t)p]]);
            nxt[(size_t)i] = head[(size_t)k];
            head[(size_t)k] = i;
        }
        std::vector<int32_t> first((size_t)n, -1), maxfirst((size_t)n, -1), prevleaf((size_t)n, -1), anc((size_t)n);
        for (int32_t k = 0; k < n; k++) {
            int32_t j = post[(size_t)k];
            cp[j] = first[(size_t)j] == -1 ? 1 : 0;                  // cp doubles as delta
            while (j != -1 && first[(size_t)j] == -1) {
                first[(size_t)j] = k;
                j = parent[j];
            }
        }
        for (int32_t j = 0; j < n; j++) anc[(size_t)j] = j;
        for (int32_t k = 0; k < n; k++) {
            const int32_t j = post[(size_t)k];
            if (parent[j] != -1) cp[parent[j]]--;
            for (int32_t J = head[(size_t)k]; J != -1; J = nxt[(size_t)J]) {
                for (int32_t p = ATp[(size_t)J]; p < ATp[(size_t)J + 1]; p++) {
                    const int32_t i = ATi[(size_t)p];
                    if (i <= j || first[(size_t)j] <= maxfirst[(size_t)i]) continue;
                    maxfirst[(size_t)i] = first[(size_t)j];
                    const int32_t jprev = prevleaf[(size_t)i];
                    prevleaf[(size_t)i] = j;
                    cp[j]++;
                    if (jprev != -1) {
                        int32_t q = jprev;
                        while (q != anc[(size_t)q]) q = anc[(size_t)q];
                        for (int32_t s = jprev; s != q;) {
                            const int32_t sp = anc[(size_t)s];
                            anc[(size_t)s] = q;
                            s = sp;
                        }
                        cp[q]--;
                    }
                }
            }
            if (parent[j] != -1) anc[(size_t)j] = parent[j];
        }
        for (int32_t j = 0; j < n; j++)
            if (parent[j] != -1) cp[parent[j]] += cp[j];
    }
    int64_t rnz = 0;
    for (int32_t j = 0; j < n; j++) rnz += cp[j];
    // ---- cs_vcount: leftmost column of every row, rows handed to the columns along the tree, pinv, m2, nnz(V) ----
    for (int32_t i = 0; i < m + n; i++) pinv[i] = -1;
    for (int32_t i = 0; i < m; i++) leftmost[i] = -1;
    for (int32_t k = n - 1; k >= 0; k--)
        for (int32_t p = Ap[k]; p < Ap[k + 1]; p++) leftmost[Ai[p]] = k;
    std::vector<int32_t> head((size_t)n, -1), tail((size_t)n, -1), count((size_t)n, 0), nxt((size_t)m, -1);
    for (int32_t i = m - 1; i >= 0; i--) {
        const int32_t k = leftmost[i];
        if (k == -1) continue;
        if (count[(size_t)k]++ == 0) tail[(size_t)k] = i;
        nxt[(size_t)i] = head[(size_t)k];
        head[(size_t)k] = i;
    }
    int64_t vnz = 0;
    int32_t m2 = m;
    for (int32_t k = 0; k < n; k++) {
        int32_t i = head[(size_t)k];
        vnz++;
        if (i < 0) i = m2++;                                       // a fictitious row for a column without one
        pinv[i] = k;
        if (--count[(size_t)k] <= 0) continue;
        vnz += count[(size_t)k];
        const int32_t pa = parent[k];
        if (pa != -1) {
            if (count[(size_t)pa] == 0) tail[(size_t)pa] = tail[(size_t)k];
            nxt[(size_t)tail[(size_t)k]] = head[(size_t)pa];
            head[(size_t)pa] = nxt[(size_t)i];
            count[(size_t)pa] += count[(size_t)k];
        }
    }
    for (int32_t i = 0, k = n; i < m; i++)
        if (pinv[i] < 0) pinv[i] = k++;
    *m2_out = m2;
    *vnz_out = vnz;
    *rnz_out = rnz;
    return CSX_OK;
}
