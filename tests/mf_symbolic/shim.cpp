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
