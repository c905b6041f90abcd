#include "mf_symbolic.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <stdexcept>

namespace ipcgpu {

namespace {

struct Graph {
    int nn;
    std::vector<int> ptr, adj;
};

Graph node_graph(int n, const int* ia, const int* ja)
{
    Graph g;
    g.nn = n / 3;
    std::vector<int> deg(g.nn, 0);
    // entries come in 3x3 blocks; count every (u,w) block once per scalar entry and dedupe below
    std::vector<std::pair<int, int>> edges;
    edges.reserve((size_t)ia[n] / 4);
    for (int r = 0; r < n; r += 3) { // the first row of a block row lists every neighbour block
        int u = r / 3, last = -1;
        for (int k = ia[r]; k < ia[r + 1]; ++k) {
            int w = ja[k] / 3;
            if (w != u && w != last) {
                edges.emplace_back(u, w);
                last = w;
            }
        }
    }
    // rows 3u+1, 3u+2 carry the same neighbour blocks for patterns built by set_pattern; for a general CSR
    // handed in through set_pattern_csr scan them too
    for (int r = 0; r < n; ++r) {
        if (r % 3 == 0) continue;
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
    Dissector(const Graph& g, const double* coords, int leaf) : g_(g), xyz_(coords), leaf_(leaf), mark_(g.nn, -1), key_(g.nn, 0.0), seen_(g.nn, -1) {}

    std::vector<std::vector<int>> run()
    {
        std::vector<int> all(g_.nn);
        std::iota(all.begin(), all.end(), 0);
        split(all);
        return std::move(groups_);
    }

private:
    const Graph& g_;
    const double* xyz_;
    int leaf_;
    std::vector<int> mark_;
    std::vector<double> key_;
    std::vector<int> seen_;
    int tag_ = 0, seenTag_ = 0;
    std::vector<std::vector<int>> groups_;

    // key_[v] = coordinate along the longest bbox axis (geometric) or BFS depth from a pseudo-peripheral node
    void compute_keys(const std::vector<int>& S, int t)
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
            return;
        }
        std::vector<int> order;
        bfs(S, S[0], t, order);
        int far = order.back();
        bfs(S, far, t, order);
    }
    void bfs(const std::vector<int>& S, int start, int t, std::vector<int>& order)
    {
        order.clear();
        ++seenTag_;
        auto run = [&](int s0, double d0) {
            size_t head = order.size();
            order.push_back(s0);
            seen_[s0] = seenTag_;
            key_[s0] = d0;
            while (head < order.size()) {
                int u = order[head++];
                for (int k = g_.ptr[u]; k < g_.ptr[u + 1]; ++k) {
                    int w = g_.adj[k];
                    if (mark_[w] == t && seen_[w] != seenTag_) {
                        seen_[w] = seenTag_;
                        key_[w] = key_[u] + 1.0;
                        order.push_back(w);
                    }
                }
            }
        };
        run(start, 0.0);
        for (int v : S)
            if (seen_[v] != seenTag_) run(v, key_[order.back()] + 1.0);
    }

    void split(std::vector<int>& S)
    {
        if ((int)S.size() <= leaf_) {
            if (!S.empty()) groups_.push_back(S);
            return;
        }
        const int t = ++tag_;
        for (int v : S) mark_[v] = t;
        compute_keys(S, t);
        // median cut on the key; ties (same BFS level / same coordinate plane) stay on one side
        std::vector<int> sorted = S;
        std::sort(sorted.begin(), sorted.end(), [&](int a, int b) { return key_[a] < key_[b] || (key_[a] == key_[b] && a < b); });
        const size_t half = sorted.size() / 2;
        const double kmid = key_[sorted[half]];
        size_t lo = half, hi = half;
        while (lo > 0 && key_[sorted[lo - 1]] == kmid) --lo;
        while (hi < sorted.size() && key_[sorted[hi]] == kmid) ++hi;
        size_t cut = (half - lo <= hi - half && lo > 0) ? lo : hi; // left = sorted[0..cut)
        if (cut == 0 || cut >= sorted.size()) { // cannot be split on this key
            groups_.push_back(S);
            return;
        }
        const double kcut = key_[sorted[cut]]; // left: key < kcut
        // vertex separator: the smaller of the two one-sided boundaries
        std::vector<int> bl, br;
        for (size_t i = 0; i < sorted.size(); ++i) {
            int v = sorted[i];
            bool isLeft = i < cut;
            bool touches = false;
            for (int k = g_.ptr[v]; k < g_.ptr[v + 1] && !touches; ++k) {
                int w = g_.adj[k];
                if (mark_[w] == t && ((key_[w] < kcut) != isLeft)) touches = true;
            }
            if (touches) (isLeft ? bl : br).push_back(v);
        }
        const bool useLeft = bl.size() <= br.size();
        const std::vector<int>& sep = useLeft ? bl : br;
        std::vector<char> inSep; // local flags through seen_ reuse would clash with bfs; use a tag on mark_
        const int sepTag = ++tag_;
        for (int v : sep) mark_[v] = sepTag;
        std::vector<int> left, right;
        for (size_t i = 0; i < sorted.size(); ++i) {
            int v = sorted[i];
            if (mark_[v] == sepTag) continue;
            (i < cut ? left : right).push_back(v);
        }
        std::vector<int> sepCopy = sep;
        split(left);
        split(right);
        if (!sepCopy.empty()) groups_.push_back(std::move(sepCopy));
    }
};

} // namespace

void mf_analyze(int n, const int* ia, const int* ja, const double* coords, int leafSize, MfSymbolic& o)
{
    if (n % 3 != 0) throw std::invalid_argument("mf_analyze: row count must be a multiple of 3 (3x3 node blocks)");
    o = MfSymbolic();
    o.n = n;
    Graph g = node_graph(n, ia, ja);
    o.nn = g.nn;
    Dissector nd(g, coords, leafSize);
    std::vector<std::vector<int>> groups = nd.run();
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

    // symbolic factorisation on the front level: struct(s) = (adj(s) U struct(children)) \ {nodes < end(s)}
    std::vector<std::vector<int>> st(o.ns), kids(o.ns);
    o.parent.assign(o.ns, -1);
    for (int s = 0; s < o.ns; ++s) {
        const int end = o.firstNode[s + 1];
        std::vector<int>& r = st[s];
        for (int v = o.firstNode[s]; v < end; ++v) {
            const int ov = o.oldOf[v];
            for (int k = g.ptr[ov]; k < g.ptr[ov + 1]; ++k) {
                const int w = o.newOf[g.adj[k]];
                if (w >= end) r.push_back(w);
            }
        }
        for (int c : kids[s])
            for (int w : st[c])
                if (w >= end) r.push_back(w);
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
        if (!r.empty()) {
            o.parent[s] = frontOfNode[r[0]];
            kids[o.parent[s]].push_back(s);
        }
    }
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
    // user-matrix entry -> front slot (lower triangle of the permuted matrix)
    o.aDst.resize(ia[n]);
    for (int r = 0; r < n; ++r)
        for (int k = ia[r]; k < ia[r + 1]; ++k) {
            const int c = ja[k];
            const int pr = 3 * o.newOf[r / 3] + r % 3, pc = 3 * o.newOf[c / 3] + c % 3;
            const int i = std::max(pr, pc), j = std::min(pr, pc);
            const int s = frontOfNode[j / 3];
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
            o.aDst[k] = o.frontOff[s] + lr + N * (int64_t)(j - 3 * f);
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
