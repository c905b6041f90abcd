// TEST INFRASTRUCTURE (host only): the symbolic analysis of the multifrontal solver (ipc_amd/csrc/mf_symbolic.cpp) behind a plain C call, so that
// tests/test_mf_symbolic.py can check its fronts against an independent restatement in Python.  Built by the test with g++.
#include "../../ipc_amd/csrc/mf_symbolic.h"
#include <algorithm>
#include <cstring>
using namespace ipcgpu;
static MfSymbolic g_sym;
extern "C" int shim_analyze(int n, const int* ia, const int* ja, const double* coords, int leaf, int* sizes4)
{
    try {
        mf_analyze(n, ia, ja, coords, leaf, g_sym);
    }
    catch (...) {
        return 1;
    }
    sizes4[0] = g_sym.ns;
    sizes4[1] = g_sym.nn;
    sizes4[2] = (int)g_sym.idx.size();
    sizes4[3] = (int)g_sym.child.size();
    return 0;
}
extern "C" void shim_fetch(int* newOf, int* firstNode, int* parent, int* idxPtr, int* idx, int* childPtr, int* child, long long* aDst, int* aFront, long long* frontOff)
{
    auto cp = [](auto* dst, const auto& v) { std::copy(v.begin(), v.end(), dst); };
    cp(newOf, g_sym.newOf);
    cp(firstNode, g_sym.firstNode);
    cp(parent, g_sym.parent);
    cp(idxPtr, g_sym.idxPtr);
    cp(idx, g_sym.idx);
    cp(childPtr, g_sym.childPtr);
    cp(child, g_sym.child);
    cp(aDst, g_sym.aDst);
    cp(aFront, g_sym.aFront);
    cp(frontOff, g_sym.frontOff);
}
// owner of every front (rank, -1 above the cut) and of every node (old numbering) for `world` ranks; returns the share of the flops above the cut
extern "C" double shim_owners(int world, int* frontOwner, int* nodeOwner)
{
    std::vector<int> owner;
    const double shared = mf_assign_owners(g_sym, world, owner);
    std::copy(owner.begin(), owner.end(), frontOwner);
    for (int s = 0; s < g_sym.ns; ++s)
        for (int v = g_sym.firstNode[s]; v < g_sym.firstNode[s + 1]; ++v) nodeOwner[g_sym.oldOf[v]] = owner[s];
    return shared;
}
// round 5: executors, groups and the point-to-point exchange plan of one rank.  level[s] of every front; records of 7 ints:
// (level, kind, front, off lo, off hi, offW, peer), kind 0 = send, 1 = receive (update matrix + vector), 2 = solution segment out, 3 = solution segment in.
// Returns the number of records (-1: more than `cap`).
extern "C" int shim_exchange_plan(int world, int rank, int* exec, unsigned long long* group, int* level, int* rec, int cap)
{
    std::vector<int> owner, ex;
    std::vector<unsigned long long> gr;
    mf_assign_owners(g_sym, world, owner);
    mf_assign_executors(g_sym, owner, ex, gr);
    std::copy(ex.begin(), ex.end(), exec);
    std::copy(gr.begin(), gr.end(), group);
    std::copy(g_sym.level.begin(), g_sym.level.end(), level);
    std::vector<MfExchangeLevel> plan;
    mf_exchange_plan(g_sym, owner, ex, gr, rank, world, plan);
    int n = 0;
    auto put = [&](int l, int kind, const MfExchangeItem& it) {
        if (n < cap) {
            int* r = rec + 7 * n;
            r[0] = l;
            r[1] = kind;
            r[2] = it.front;
            r[3] = (int)(unsigned)(it.off & 0xffffffffll);
            r[4] = (int)(it.off >> 32);
            r[5] = it.offW;
            r[6] = it.peer;
        }
        ++n;
    };
    for (int l = 0; l < (int)plan.size(); ++l) {
        for (const auto& it : plan[l].send) put(l, 0, it);
        for (const auto& it : plan[l].recv) put(l, 1, it);
        for (const auto& it : plan[l].xsSend) put(l, 2, it);
        for (const auto& it : plan[l].xsRecv) put(l, 3, it);
    }
    return n <= cap ? n : -1;
}
