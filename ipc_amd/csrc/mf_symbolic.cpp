#include "mf_symbolic.h"
#include <algorithm>
#include <memory>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cmath>
#include <numeric>
#include <mutex>
#include <stdexcept>
#include <system_error>
#include <thread>

namespace ipcgpu {

namespace {

struct Graph {
    int nn;
    std::vector<int> ptr, adj;
};

// body(0) ... body(n - 1) on n host threads (body(0) on the caller's).  A thread that cannot be started has its share run by the caller; the first exception
// of any share is rethrown once every thread has been joined.
template <class Body>
void run_threads(int n, Body&& body)
{
    std::exception_ptr err;
    std::mutex errLock;
    auto guarded = [&](int t) {
        try {
            body(t);
        }
        catch (...) {
            std::lock_guard<std::mutex> lk(errLock);
            if (!err) err = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(std::max(n - 1, 0));
    std::vector<int> inlineShares;
    for (int t = 1; t < n; ++t) {
        try {
            pool.emplace_back(guarded, t);
        }
        catch (const std::system_error&) {
            inlineShares.push_back(t);
        }
    }
    if (n > 0) guarded(0);
    for (int t : inlineShares) guarded(t);
    for (auto& th : pool) th.join();
    if (err) std::rethrow_exception(err);
}

// Patterns built by LinSysSolver::set_pattern are block structured: the three scalar rows of a node list the same neighbour
// blocks (row 3u + d is row 3u minus its first d entries).  Then row 3u alone gives the node graph, already sorted: the
// neighbours below u arrive in ascending order from the earlier rows, the ones above u from its own row.
bool block_structured(int n, const int* ia, const int* ja)
{
    for (int r = 0; r < n; r += 3) {
        const int L = ia[r + 1] - ia[r];
        if (L % 3 != 0 || ia[r + 2] - ia[r + 1] != L - 1 || ia[r + 3] - ia[r + 2] != L - 2) return false;
        for (int k = ia[r]; k < ia[r + 1]; k += 3)
            if (ja[k] % 3 != 0 || ja[k + 1] != ja[k] + 1 || ja[k + 2] != ja[k] + 2) return false;
    }
    return true;
}

Graph node_graph(int n, const int* ia, const int* ja)
{
    Graph g;
    g.nn = n / 3;
    if (block_structured(n, ia, ja)) {
        // adj(u) = neighbours below u in ascending order (they come from the rows of those neighbours) followed by the ones above u (its own row).
        // A counting sort over node ranges on a few threads: thread t counts, per target w, the rows of its range that list w; the offsets of
        // the threads inside adj(w) follow in thread order, i.e. ascending u -- the same lists a single pass over the rows produces.
        const int nn = g.nn;
        const int nThreads = nn < 4096 ? 1 : std::max(1, std::min(4, (int)std::thread::hardware_concurrency())); // starting a thread costs what 1 000 nodes do
        auto lo = [&](int t) { return (int)((int64_t)nn * t / nThreads); };
        std::vector<std::vector<int>> below(nThreads, std::vector<int>(nn, 0)); // below[t][w]: rows u of range t with w in row u (u < w)
        auto parallel = [&](auto&& body) { run_threads(nThreads, body); };
        parallel([&](int t) {
            int* c = below[t].data();
            for (int u = lo(t); u < lo(t + 1); ++u)
                for (int k = ia[3 * u] + 3; k < ia[3 * u + 1]; k += 3) c[ja[k] / 3]++; // first block = the diagonal
        });
        g.ptr.assign(nn + 1, 0);
        for (int w = 0; w < nn; ++w) {
            int tot = 0;
            for (int t = 0; t < nThreads; ++t) {
                const int c = below[t][w];
                below[t][w] = tot; // offset of thread t's rows inside the lower part of adj(w)
                tot += c;
            }
            g.ptr[w + 1] = g.ptr[w] + tot + (ia[3 * w + 1] - ia[3 * w]) / 3 - 1;
        }
        g.adj.resize(g.ptr[nn]);
        parallel([&](int t) {
            int* off = below[t].data();
            for (int u = lo(t); u < lo(t + 1); ++u) {
                // the upper part of adj(u) sits behind the lower part, whose length is what is left of the list
                int* dst = g.adj.data() + g.ptr[u + 1] - ((ia[3 * u + 1] - ia[3 * u]) / 3 - 1);
                for (int k = ia[3 * u] + 3; k < ia[3 * u + 1]; k += 3) {
                    const int w = ja[k] / 3;
                    g.adj[g.ptr[w] + off[w]++] = u;
                    *dst++ = w;
                }
            }
        });
        return g;
    }
    // general scalar CSR (set_pattern_csr): every (u, w) block that holds at least one entry
    std::vector<std::pair<int, int>> edges;
    edges.reserve((size_t)ia[n] / 4);
    for (int r = 0; r < n; ++r) {
        int u = r / 3, last = -1;
        for (int k = ia[r]; k < ia[r + 1]; ++k) {
            int w = ja[k] / 3;
            if (w != u && w != last) {
                edges.emplace_back(u, w);
                last = w;
            }
        }
    }
    std::vector<std::vector<int>> tmp(g.nn);
    for (auto& e : edges) {
        tmp[e.first].push_back(e.second);
        tmp[e.second].push_back(e.first);
    }
    g.ptr.assign(g.nn + 1, 0);
    for (int u = 0; u < g.nn; ++u) {
        auto& a = tmp[u];
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
        g.ptr[u + 1] = g.ptr[u] + (int)a.size();
    }
    g.adj.resize(g.ptr[g.nn]);
    for (int u = 0; u < g.nn; ++u) std::copy(tmp[u].begin(), tmp[u].end(), g.adj.begin() + g.ptr[u]);
    return g;
}

// Nested dissection.  Produces groups of old node ids in elimination order (domains first, separators last).
class Dissector {
public:
    Dissector(const Graph& g, const double* coords, int leaf)
        : g_(g), xyz_(coords), leaf_(leaf), mark_(new std::atomic<int>[std::max(g.nn, 1)]), key_(g.nn, 0.0), seen_(g.nn, -1)
    {
        for (int i = 0; i < g.nn; ++i) mark_[i].store(-1, std::memory_order_relaxed);
    }

    // taskOf[i]: groups of one subproblem that ran on a single thread share an id >= 0 and are contiguous; -1 for the separators above those
    // subproblems.  Nothing outside such a subproblem except its ancestors' separators is adjacent to it, which is what lets the caller build
    // the front structures of different subproblems on different threads.
    std::vector<std::vector<int>> run(std::vector<int>& taskOf)
    {
        std::vector<int> all(g_.nn);
        std::iota(all.begin(), all.end(), 0);
        std::vector<std::vector<int>> groups;
        taskOf.clear();
        split(all, groups, taskOf, 0, -1);
        return groups;
    }

private:
    // The two halves of a cut are independent: the first levels of the recursion run them on separate threads.  A node belongs
    // to exactly one active subproblem, so key_ / seen_ are only touched by their owner; mark_ is also READ for neighbours
    // that may belong to another subproblem (where it holds that subproblem's tag, never ours) -- hence relaxed atomics; the tags
    // come from atomic counters and are unique.
#ifndef MF_ND_PAR_DEPTH
#define MF_ND_PAR_DEPTH 4 // (3 / 8192 until round 6: profiles/r06_nd_threads_ab.txt)
#endif
#ifndef MF_ND_PAR_MIN
#define MF_ND_PAR_MIN 2048 // nodes below which a subproblem stays on its thread
#endif
    static constexpr int PAR_DEPTH = MF_ND_PAR_DEPTH;
    const Graph& g_;
    const double* xyz_;
    int leaf_;
    std::unique_ptr<std::atomic<int>[]> mark_;
    std::vector<double> key_;
    std::vector<int> seen_;
    std::atomic<int> tag_{ 0 }, seenTag_{ 0 }, task_{ 0 };
    int markOf(int v) const { return mark_[v].load(std::memory_order_relaxed); }
    void setMark(int v, int t) { mark_[v].store(t, std::memory_order_relaxed); }

    // Per axis, the smallest and largest coordinate among a node's neighbours (whole graph): a node whose neighbours all lie on its own side of the
    // cut plane cannot be on the boundary of the cut, and that is nearly every node -- the adjacency scan of the separator search then only runs for
    // the few next to the plane.  (A necessary condition only: the separators are the same sets as without it.)  Built on first use of an axis.
    std::vector<double> nbMin_[3], nbMax_[3];
    std::once_flag nbOnce_[3];
    void ensureNeighbourRange(int ax)
    {
        std::call_once(nbOnce_[ax], [&] {
            const int nn = g_.nn;
            nbMin_[ax].resize(nn);
            nbMax_[ax].resize(nn);
            const int nThreads = nn < 16384 ? 1 : std::max(1, std::min(4, (int)std::thread::hardware_concurrency()));
            auto range = [&](int t) {
                for (int v = (int)((int64_t)nn * t / nThreads); v < (int)((int64_t)nn * (t + 1) / nThreads); ++v) {
                    double mn = 1e300, mx = -1e300;
                    for (int k = g_.ptr[v]; k < g_.ptr[v + 1]; ++k) {
                        const double c = xyz_[3 * (size_t)g_.adj[k] + ax];
                        mn = std::min(mn, c);
                        mx = std::max(mx, c);
                    }
                    nbMin_[ax][v] = mn;
                    nbMax_[ax][v] = mx;
                }
            };
            run_threads(nThreads, range);
        });
    }

    // key_[v] = coordinate along the longest bbox axis (geometric) or BFS depth from a pseudo-peripheral node; returns the axis, -1 for BFS keys
    int compute_keys(const std::vector<int>& S, int t)
    {
        if (xyz_) {
            double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
            for (int v : S)
                for (int c = 0; c < 3; ++c) {
                    lo[c] = std::min(lo[c], xyz_[3 * (size_t)v + c]);
                    hi[c] = std::max(hi[c], xyz_[3 * (size_t)v + c]);
                }
            int ax = 0;
            for (int c = 1; c < 3; ++c)
                if (hi[c] - lo[c] > hi[ax] - lo[ax]) ax = c;
            for (int v : S) key_[v] = xyz_[3 * (size_t)v + ax];
            return ax;
        }
        std::vector<int> order;
        bfs(S, S[0], t, order);
        int far = order.back();
        bfs(S, far, t, order);
        return -1;
    }
    void bfs(const std::vector<int>& S, int start, int t, std::vector<int>& order)
    {
        order.clear();
        const int st = ++seenTag_;
        auto run = [&](int s0, double d0) {
            size_t head = order.size();
            order.push_back(s0);
            seen_[s0] = st;
            key_[s0] = d0;
            while (head < order.size()) {
                int u = order[head++];
                for (int k = g_.ptr[u]; k < g_.ptr[u + 1]; ++k) {
                    int w = g_.adj[k];
                    if (markOf(w) == t && seen_[w] != st) {
                        seen_[w] = st;
                        key_[w] = key_[u] + 1.0;
                        order.push_back(w);
                    }
                }
            }
        };
        run(start, 0.0);
        for (int v : S)
            if (seen_[v] != st) run(v, key_[order.back()] + 1.0);
    }

    void split(std::vector<int>& S, std::vector<std::vector<int>>& out, std::vector<int>& outTask, int depth, int task)
    {
        if ((int)S.size() <= leaf_) {
            if (!S.empty()) {
                out.push_back(S);
                outTask.push_back(task);
            }
            return;
        }
        const int t = ++tag_;
        for (int v : S) setMark(v, t);
        const int ax = compute_keys(S, t);
        // median cut on the key; ties (same BFS level / same coordinate plane) stay on one side.  left = { key < kcut }.
        // The result depends on S only as a set, so the geometric case avoids the full sort: median by selection (O(n)), one
        // counting pass for the tie range, one classification pass.  (BFS keys depend on the traversal start, i.e. on the
        // order of S: that case keeps the sorted order.)
        std::vector<int> sorted = S;
        size_t lo, hi;
        double kmid;
        const size_t half = sorted.size() / 2;
        if (xyz_) {
            std::vector<double> ks(S.size());
            for (size_t i = 0; i < S.size(); ++i) ks[i] = key_[S[i]];
            std::nth_element(ks.begin(), ks.begin() + half, ks.end());
            kmid = ks[half];
            lo = hi = 0;
            for (int v : S) {
                lo += key_[v] < kmid;
                hi += key_[v] <= kmid;
            }
        }
        else {
            std::sort(sorted.begin(), sorted.end(), [&](int a, int b) { return key_[a] < key_[b] || (key_[a] == key_[b] && a < b); });
            kmid = key_[sorted[half]];
            lo = hi = half;
            while (lo > 0 && key_[sorted[lo - 1]] == kmid) --lo;
            while (hi < sorted.size() && key_[sorted[hi]] == kmid) ++hi;
        }
        const bool cutBelow = (half - lo <= hi - half && lo > 0); // left = keys < kmid, else keys <= kmid
        const size_t cut = cutBelow ? lo : hi;
        if (cut == 0 || cut >= sorted.size()) { // cannot be split on this key
            out.push_back(S);
            outTask.push_back(task);
            return;
        }
        auto isLeftOf = [&](int v) { return cutBelow ? key_[v] < kmid : key_[v] <= kmid; };
        // can v have a neighbour on the other side at all?  (geometric keys: from the coordinate range of its neighbours; BFS keys: neighbours are at
        // most one level apart)
        const double* nbMin = nullptr;
        const double* nbMax = nullptr;
        if (ax >= 0) {
            ensureNeighbourRange(ax);
            nbMin = nbMin_[ax].data();
            nbMax = nbMax_[ax].data();
        }
        auto mayTouch = [&](int v, bool isLeft) {
            if (ax < 0) return std::fabs(key_[v] - kmid) <= 1.0;
            if (isLeft) return cutBelow ? nbMax[v] >= kmid : nbMax[v] > kmid;
            return cutBelow ? nbMin[v] < kmid : nbMin[v] <= kmid;
        };
        // vertex separator: the smaller of the two one-sided boundaries
        std::vector<int> bl, br;
        for (size_t i = 0; i < sorted.size(); ++i) {
            int v = sorted[i];
            const bool isLeft = isLeftOf(v);
            if (!mayTouch(v, isLeft)) continue;
            bool touches = false;
            for (int k = g_.ptr[v]; k < g_.ptr[v + 1] && !touches; ++k) {
                int w = g_.adj[k];
                if (markOf(w) == t && (isLeftOf(w) != isLeft)) touches = true;
            }
            if (touches) (isLeft ? bl : br).push_back(v);
        }
        const bool useLeft = bl.size() <= br.size();
        const std::vector<int>& sep = useLeft ? bl : br;
        std::vector<char> inSep; // local flags through seen_ reuse would clash with bfs; use a tag on mark_
        const int sepTag = ++tag_;
        for (int v : sep) setMark(v, sepTag);
        std::vector<int> left, right;
        for (size_t i = 0; i < sorted.size(); ++i) {
            int v = sorted[i];
            if (markOf(v) == sepTag) continue;
            (isLeftOf(v) ? left : right).push_back(v);
        }
        std::vector<int> sepCopy = sep;
        if (depth < PAR_DEPTH && sorted.size() > MF_ND_PAR_MIN) {
            std::vector<std::vector<int>> gr;
            std::vector<int> grTask;
            std::exception_ptr err;
            std::thread th([&] {
                try {
                    split(right, gr, grTask, depth + 1, -1);
                }
                catch (...) {
                    err = std::current_exception();
                }
            });
            split(left, out, outTask, depth + 1, -1);
            th.join();
            if (err) std::rethrow_exception(err);
            for (auto& grp : gr) out.push_back(std::move(grp));
            outTask.insert(outTask.end(), grTask.begin(), grTask.end());
        }
        else {
            if (task < 0) task = task_++; // from here down one thread
            split(left, out, outTask, depth + 1, task);
            split(right, out, outTask, depth + 1, task);
        }
        if (!sepCopy.empty()) {
            out.push_back(std::move(sepCopy));
            outTask.push_back(task);
        }
    }
};

} // namespace

// user-matrix entry -> front slot (lower triangle of the permuted matrix): aDst / aFront of `o` from its ordering and front structures.  The product computes
// the same two arrays on the device (MfNumeric::setup, k_entry_dst: round 5); this host version is what mf_analyze(..., withEntryDestinations = true) fills in --
// the tests of the analysis and the GPU test that pins the device kernel on it.
void mf_entry_destinations(int n, const int* ia, const int* ja, MfSymbolic& o)
{
    std::vector<int> frontOfNode(o.nn);
    for (int s = 0; s < o.ns; ++s)
        for (int v = o.firstNode[s]; v < o.firstNode[s + 1]; ++v) frontOfNode[v] = s;
    // room for the blocks a later (contact) pattern adds: growing inside the capacity faults in only the new pages, a reallocation all of them
    if (o.aDst.capacity() < (size_t)ia[n]) {
        o.aDst.reserve((size_t)ia[n] + (size_t)ia[n] / 2);
        o.aFront.reserve((size_t)ia[n] + (size_t)ia[n] / 2);
    }
    o.aDst.resize(ia[n]);
    o.aFront.resize(ia[n]);
    auto slot = [&](int r, int c, int* owner = nullptr) -> int64_t {
        const int pr = 3 * o.newOf[r / 3] + r % 3, pc = 3 * o.newOf[c / 3] + c % 3;
        const int i = std::max(pr, pc), j = std::min(pr, pc);
        const int s = frontOfNode[j / 3];
        if (owner) *owner = s;
        const int f = o.firstNode[s], l = o.firstNode[s + 1];
        const int64_t N = o.N(s);
        int64_t lr;
        if (i / 3 < l) lr = i - 3 * f;
        else {
            const int* b = o.idx.data() + o.idxPtr[s] + (l - f);
            const int* e = o.idx.data() + o.idxPtr[s + 1];
            const int* it = std::lower_bound(b, e, i / 3);
            if (it == e || *it != i / 3) throw std::logic_error("mf_analyze: matrix entry outside the symbolic structure");
            lr = 3 * (int64_t)((l - f) + (it - b)) + i % 3;
        }
        return o.frontOff[s] + lr + N * (int64_t)(j - 3 * f);
    };
    if (block_structured(n, ia, ja)) {
        // one lookup per 3 x 3 node block instead of one per scalar entry: inside a block the destination moves by 1 per
        // row of the front and by N per column.  Node ranges are independent: a few host threads.
        auto fillRange = [&](int u0, int u1) {
        for (int u = u0; u < u1; ++u) {
            const int base = ia[3 * u], L = ia[3 * u + 1] - base;
            const int row1 = ia[3 * u + 1], row2 = ia[3 * u + 2];
            // diagonal block: upper entries (a, b), a <= b, of node u -> lower entries (b, a) of its front
            {
                const int pu = o.newOf[u], s = frontOfNode[pu];
                const int64_t N = o.N(s), d0 = o.frontOff[s] + 3 * (int64_t)(pu - o.firstNode[s]) * (N + 1);
                o.aDst[base] = d0;
                o.aDst[base + 1] = d0 + 1;
                o.aDst[base + 2] = d0 + 2;
                o.aDst[row1] = d0 + N + 1;
                o.aDst[row1 + 1] = d0 + N + 2;
                o.aDst[row2] = d0 + 2 * N + 2;
                o.aFront[base] = o.aFront[base + 1] = o.aFront[base + 2] = o.aFront[row1] = o.aFront[row1 + 1] = o.aFront[row2] = s;
            }
            for (int q = 3; q < L; q += 3) {
                const int w = ja[base + q] / 3;
                const int pu = o.newOf[u], pw = o.newOf[w];
                const int64_t d0 = slot(3 * u, 3 * w); // entry (row 0 of u, column 0 of w)
                const int s = frontOfNode[std::min(pu, pw)];
                const int64_t N = o.N(s);
                // scalar (a of u, b of w): if u is the row node of the front (pu > pw) the row index follows a, the column b
                const int64_t da = pu > pw ? 1 : N, db = pu > pw ? N : 1;
                for (int b = 0; b < 3; ++b) {
                    o.aDst[base + q + b] = d0 + db * b;
                    o.aDst[row1 + q - 1 + b] = d0 + da + db * b;
                    o.aDst[row2 + q - 2 + b] = d0 + 2 * da + db * b;
                    o.aFront[base + q + b] = o.aFront[row1 + q - 1 + b] = o.aFront[row2 + q - 2 + b] = s;
                }
            }
        }
        };
        const int nThreads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
        run_threads(nThreads, [&](int t) { fillRange((int)((int64_t)o.nn * t / nThreads), (int)((int64_t)o.nn * (t + 1) / nThreads)); });
    }
    else {
        for (int r = 0; r < n; ++r)
            for (int k = ia[r]; k < ia[r + 1]; ++k) o.aDst[k] = slot(r, ja[k], &o.aFront[k]);
    }
}

void mf_analyze(int n, const int* ia, const int* ja, const double* coords, int leafSize, MfSymbolic& o, bool withEntryDestinations)
{
    if (n % 3 != 0) throw std::invalid_argument("mf_analyze: row count must be a multiple of 3 (3x3 node blocks)");
    // every vector below is re-assigned in full: keep the capacity of a previous analysis (tens of MB that would otherwise be
    // unmapped and page-faulted back in on every pattern change)
    o.nnzL = 0;
    o.flops = 0;
    o.maxN = 0;
    o.n = n;
    static const bool timeIt = std::getenv("IPCGPU_MF_SETUP_TIMES") != nullptr;
    auto tLap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timeIt) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mf analyze %-26s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tLap).count());
        tLap = now;
    };
    Graph g = node_graph(n, ia, ja);
    lap("node graph");
    o.nn = g.nn;
    Dissector nd(g, coords, leafSize);
    std::vector<int> taskOf;
    std::vector<std::vector<int>> groups = nd.run(taskOf);
    lap("nested dissection");
    o.ns = (int)groups.size();
    o.newOf.assign(o.nn, -1);
    o.oldOf.assign(o.nn, -1);
    o.firstNode.assign(o.ns + 1, 0);
    std::vector<int> frontOfNode(o.nn);
    int next = 0;
    for (int s = 0; s < o.ns; ++s) {
        o.firstNode[s] = next;
        std::sort(groups[s].begin(), groups[s].end());
        for (int v : groups[s]) {
            o.newOf[v] = next;
            o.oldOf[next] = v;
            frontOfNode[next] = s;
            ++next;
        }
    }
    o.firstNode[o.ns] = next;
    if (next != o.nn) throw std::logic_error("mf_analyze: ordering lost nodes");

    // symbolic factorisation on the front level: struct(s) = (adj(s) U struct(children)) \ {nodes < end(s)}.
    // The subproblems the dissection ran on one thread each (taskOf) are closed under "child of": a front in one of them has its structure
    // inside the subproblem or in the separators above it, so its parent is never a front of another subproblem.  They are built on
    // separate threads (own stamp array; children whose parent lies above the subproblem are handed over afterwards); the separators above
    // follow in order.  Should a parent ever turn up inside another subproblem, everything is redone on one thread.
    std::vector<std::vector<int>> st(o.ns), kids(o.ns);
    o.parent.assign(o.ns, -1);
    auto buildRange = [&](int s0, int s1, bool all, std::vector<int>& stamp, std::vector<std::pair<int, int>>* handOver) {
        // fronts [s0, s1) (all) or those of [s0, s1) with taskOf < 0
        for (int s = s0; s < s1; ++s) {
            if (!all && taskOf[s] >= 0) continue;
            const int end = o.firstNode[s + 1];
            std::vector<int>& r = st[s];
            r.clear();
            for (int v = o.firstNode[s]; v < end; ++v) {
                const int ov = o.oldOf[v];
                for (int k = g.ptr[ov]; k < g.ptr[ov + 1]; ++k) {
                    const int w = o.newOf[g.adj[k]];
                    if (w >= end && stamp[w] != s) {
                        stamp[w] = s;
                        r.push_back(w);
                    }
                }
            }
            for (int c : kids[s])
                for (int w : st[c])
                    if (w >= end && stamp[w] != s) {
                        stamp[w] = s;
                        r.push_back(w);
                    }
            std::sort(r.begin(), r.end());
            if (!r.empty()) {
                const int p = frontOfNode[r[0]];
                o.parent[s] = p;
                if (handOver && p >= s1) handOver->emplace_back(p, s);
                else kids[p].push_back(s);
            }
        }
    };
    std::vector<std::pair<int, int>> ranges; // [first, end) of the single-thread subproblems
    for (int s = 0; s < o.ns;) {
        int e = s + 1;
        if (taskOf[s] >= 0) {
            while (e < o.ns && taskOf[e] == taskOf[s]) ++e;
            ranges.emplace_back(s, e);
        }
        s = e;
    }
    bool closed = true;
    if (ranges.size() > 1) {
        std::vector<std::vector<std::pair<int, int>>> handOver(ranges.size());
        run_threads((int)ranges.size(), [&](int t) {
            std::vector<int> stamp(o.nn, -1);
            buildRange(ranges[t].first, ranges[t].second, true, stamp, &handOver[t]);
        });
        for (auto& h : handOver)
            for (auto& pc : h) {
                if (taskOf[pc.first] >= 0) closed = false; // parent inside another subproblem: that front was built without this child
                kids[pc.first].push_back(pc.second);
            }
    }
    std::vector<int> stamp(o.nn, -1); // the children's structures overlap almost completely: dedupe before sorting
    if (closed && ranges.size() > 1) {
        buildRange(0, o.ns, false, stamp, nullptr);
        for (auto& k : kids) std::sort(k.begin(), k.end()); // ascending, as a single pass over the fronts lists them
    }
    else {
        for (auto& k : kids) k.clear();
        o.parent.assign(o.ns, -1);
        buildRange(0, o.ns, true, stamp, nullptr);
    }
    if (timeIt) fprintf(stderr, "mf analyze   (%d fronts, %zu single-thread subproblems, closed %d)\n", o.ns, ranges.size(), (int)closed);
    lap("front structures");
    o.level.assign(o.ns, 0);
    int maxLevel = 0;
    for (int s = 0; s < o.ns; ++s) {
        int lv = 0;
        for (int c : kids[s]) lv = std::max(lv, o.level[c] + 1);
        o.level[s] = lv;
        maxLevel = std::max(maxLevel, lv);
    }
    o.idxPtr.assign(o.ns + 1, 0);
    for (int s = 0; s < o.ns; ++s) o.idxPtr[s + 1] = o.idxPtr[s] + (o.firstNode[s + 1] - o.firstNode[s]) + (int)st[s].size();
    o.idx.resize(o.idxPtr[o.ns]);
    o.frontOff.assign(o.ns + 1, 0);
    o.wOff.assign(o.ns + 1, 0);
    o.childPtr.assign(o.ns + 1, 0);
    for (int s = 0; s < o.ns; ++s) {
        int p = o.idxPtr[s];
        for (int v = o.firstNode[s]; v < o.firstNode[s + 1]; ++v) o.idx[p++] = v;
        for (int w : st[s]) o.idx[p++] = w;
        const int64_t N = o.N(s), nc = o.nc(s);
        o.frontOff[s + 1] = o.frontOff[s] + N * N;
        o.wOff[s + 1] = o.wOff[s] + N;
        o.maxN = std::max<int>(o.maxN, (int)N);
        o.nnzL += nc * (nc + 1) / 2 + nc * (N - nc);
        for (int64_t j = 0; j < nc; ++j) {
            double m = double(N - j - 1);
            o.flops += m * m + 2 * m + 1;
        }
        o.childPtr[s + 1] = o.childPtr[s] + (int)kids[s].size();
    }
    o.child.resize(o.childPtr[o.ns]);
    for (int s = 0; s < o.ns; ++s) std::copy(kids[s].begin(), kids[s].end(), o.child.begin() + o.childPtr[s]);
    // inverse relative indices: for child c of p, inv[invPtr[c] + I] = position of parent-local node I in struct(c) or -1
    o.invPtr.assign(o.ns + 1, 0);
    for (int s = 0; s < o.ns; ++s) {
        int len = 0;
        if (o.parent[s] >= 0) len = o.idxPtr[o.parent[s] + 1] - o.idxPtr[o.parent[s]];
        o.invPtr[s + 1] = o.invPtr[s] + len;
    }
    o.inv.assign(o.invPtr[o.ns], -1);
    for (int s = 0; s < o.ns; ++s) {
        const int p = o.parent[s];
        if (p < 0) continue;
        const int* pIdx = o.idx.data() + o.idxPtr[p];
        const int pLen = o.idxPtr[p + 1] - o.idxPtr[p];
        const std::vector<int>& cs = st[s];
        int* out = o.inv.data() + o.invPtr[s];
        size_t j = 0;
        for (int I = 0; I < pLen && j < cs.size(); ++I) {
            if (pIdx[I] == cs[j]) {
                out[I] = (int)j;
                ++j;
            }
            else if (pIdx[I] > cs[j]) throw std::logic_error("mf_analyze: child struct not contained in parent front");
        }
        if (j != cs.size()) throw std::logic_error("mf_analyze: child struct not contained in parent front (tail)");
    }
    // levels
    o.levelPtr.assign(maxLevel + 2, 0);
    for (int s = 0; s < o.ns; ++s) o.levelPtr[o.level[s] + 1]++;
    for (int l = 0; l <= maxLevel; ++l) o.levelPtr[l + 1] += o.levelPtr[l];
    o.levelFronts.resize(o.ns);
    {
        std::vector<int> pos(o.levelPtr.begin(), o.levelPtr.end() - 1);
        for (int s = 0; s < o.ns; ++s) o.levelFronts[pos[o.level[s]]++] = s;
    }
    lap("index + inverse maps, levels");
    if (withEntryDestinations) {
        mf_entry_destinations(n, ia, ja, o);
        lap("entry destinations");
    }
    else {
        o.aDst.clear();
        o.aFront.clear();
    }
}

double mf_assign_owners(const MfSymbolic& sym, int world, std::vector<int>& owner)
{
    const int ns = sym.ns;
    owner.assign(ns, 0);
    if (world <= 1) return 0.0;
    std::vector<double> own(ns, 0.0), cost(ns, 0.0);
    for (int s = 0; s < ns; ++s) {
        const double N = sym.N(s), nc = sym.nc(s);
        for (int j = 0; j < (int)nc; ++j) own[s] += (N - j - 1) * (N - j - 1);
        cost[s] += own[s];
        if (sym.parent[s] >= 0) cost[sym.parent[s]] += cost[s]; // children precede their parent in the elimination order
    }
    std::vector<int> frontier;
    for (int s = 0; s < ns; ++s)
        if (sym.parent[s] < 0) frontier.push_back(s);
    std::vector<char> shared(ns, 0);
    while ((int)frontier.size() < world) { // open the most expensive subtree that still has children
        int best = -1;
        for (size_t i = 0; i < frontier.size(); ++i) {
            const int f = frontier[i];
            if (sym.childPtr[f + 1] > sym.childPtr[f] && (best < 0 || cost[f] > cost[frontier[best]])) best = (int)i;
        }
        if (best < 0) break;
        const int f = frontier[best];
        frontier.erase(frontier.begin() + best);
        shared[f] = 1;
        for (int q = sym.childPtr[f]; q < sym.childPtr[f + 1]; ++q) frontier.push_back(sym.child[q]);
    }
    std::sort(frontier.begin(), frontier.end(), [&](int a, int b) { return cost[a] > cost[b] || (cost[a] == cost[b] && a < b); });
    std::vector<double> load(world, 0.0);
    std::vector<int> rootOwner(ns, -1);
    for (int f : frontier) {
        const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        load[r] += cost[f];
        rootOwner[f] = r;
    }
    // fronts in descending order: a front inherits its parent's owner unless it is a subtree root or above the cut
    double tot = 0.0, sh = 0.0;
    for (int s = ns - 1; s >= 0; --s) {
        if (shared[s]) owner[s] = -1;
        else if (rootOwner[s] >= 0) owner[s] = rootOwner[s];
        else owner[s] = owner[sym.parent[s]];
        tot += own[s];
        if (shared[s]) sh += own[s];
    }
    return tot > 0 ? sh / tot : 0.0;
}

void mf_assign_executors(const MfSymbolic& sym, const std::vector<int>& owner, std::vector<int>& exec, std::vector<unsigned long long>& group)
{
    const int ns = sym.ns;
    exec.assign(ns, 0);
    group.assign(ns, 0ull);
    std::vector<double> cost(ns, 0.0);
    for (int s = 0; s < ns; ++s) { // children precede their parent in the elimination order
        const double N = sym.N(s), nc = sym.nc(s);
        for (int j = 0; j < (int)nc; ++j) cost[s] += (N - j - 1) * (N - j - 1);
        if (owner[s] >= 0) exec[s] = owner[s];
        else {
            int best = -1;
            for (int q = sym.childPtr[s]; q < sym.childPtr[s + 1]; ++q) {
                const int c = sym.child[q];
                if (best < 0 || cost[c] > cost[best]) best = c;
            }
            exec[s] = best >= 0 ? exec[best] : 0; // (a front above the cut always has children: only opened fronts are shared)
        }
        for (int q = sym.childPtr[s]; q < sym.childPtr[s + 1]; ++q) {
            const int c = sym.child[q];
            cost[s] += cost[c];
            group[s] |= group[c];
        }
        group[s] |= 1ull << exec[s];
    }
}

void mf_exchange_plan(const MfSymbolic& sym, const std::vector<int>& owner, const std::vector<int>& exec, const std::vector<unsigned long long>& group, int rank,
    int world, std::vector<MfExchangeLevel>& plan)
{
    const int nLevels = (int)sym.levelPtr.size() - 1;
    plan.assign(nLevels, MfExchangeLevel());
    for (int l = 0; l < nLevels; ++l) {
        MfExchangeLevel& X = plan[l];
        long long off = 0;
        int offW = 0;
        for (int i = sym.levelPtr[l]; i < sym.levelPtr[l + 1]; ++i) {
            const int s = sym.levelFronts[i];
            const int p = sym.parent[s];
            if (p >= 0 && exec[p] != exec[s]) {
                const long long m = sym.N(s) - sym.nc(s);
                if (exec[s] == rank) X.send.push_back(MfExchangeItem{ s, off, offW, exec[p] });
                if (exec[p] == rank) X.recv.push_back(MfExchangeItem{ s, off, offW, exec[s] });
                off += m * (m + 1) / 2;
                offW += (int)m;
            }
            if (owner[s] < 0) { // above the cut: its solution entries go to the ranks that execute fronts below it
                if (exec[s] == rank) {
                    for (int r = 0; r < world; ++r)
                        if (r != rank && (group[s] >> r & 1ull)) X.xsSend.push_back(MfExchangeItem{ s, 0, 0, r });
                }
                else if (group[s] >> rank & 1ull) X.xsRecv.push_back(MfExchangeItem{ s, 0, 0, exec[s] });
            }
        }
        X.count = off;
        X.countW = offW;
    }
}

void mf_L_pattern_csr(const MfSymbolic& sym, std::vector<int>& ptrT, std::vector<int>& indT, std::vector<int>& pivQ)
{
    const int n = sym.n;
    std::vector<int64_t> cnt(n + 1, 0);
    // column structure of front s: every column j of the front has rows {j..end of front cols} U struct rows
    for (int s = 0; s < sym.ns; ++s) {
        const int N = sym.N(s), nc = sym.nc(s);
        const int* idx = sym.idx.data() + sym.idxPtr[s];
        for (int j = 0; j < nc; ++j)
            for (int i = j; i < N; ++i) cnt[3 * idx[i / 3] + i % 3 + 1]++;
    }
    if (sym.nnzL > INT32_MAX) throw std::overflow_error("L pattern exceeds int32 (rocSOLVER csrrf limit)");
    ptrT.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) ptrT[i + 1] = ptrT[i] + (int)cnt[i + 1];
    indT.resize(ptrT[n]);
    std::vector<int> pos(ptrT.begin(), ptrT.end() - 1);
    // fronts are visited in ascending column order, so each row receives ascending column indices
    for (int s = 0; s < sym.ns; ++s) {
        const int N = sym.N(s), nc = sym.nc(s);
        const int* idx = sym.idx.data() + sym.idxPtr[s];
        const int col0 = 3 * sym.firstNode[s];
        for (int j = 0; j < nc; ++j)
            for (int i = j; i < N; ++i) {
                const int row = 3 * idx[i / 3] + i % 3;
                indT[pos[row]++] = col0 + j;
            }
    }
    pivQ.resize(n);
    for (int v = 0; v < sym.nn; ++v)
        for (int d = 0; d < 3; ++d) pivQ[3 * v + d] = 3 * sym.oldOf[v] + d;
}

} // namespace ipcgpu
