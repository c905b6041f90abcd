"""Host side of the sparse Cholesky (`ipc_amd/csrc/mf_symbolic.cpp`, the role `cholmod_analyze` has for the reference,
`CHOLMODSolver.cpp:118-131`): ordering, front structures and entry destinations checked against an independent restatement in Python.

The analysis runs on several host threads (nested dissection, node graph, front structures by dissection subproblem, entry destinations);
the large cases here are big enough to take those paths, the small one stays on one thread.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.spatial import cKDTree

from ipc_amd import scene

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(HERE, "mf_symbolic", "_build", "libmfsym.so")
    srcs = [os.path.join(HERE, "mf_symbolic", "shim.cpp"), os.path.join(ROOT, "ipc_amd", "csrc", "mf_symbolic.cpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread"] + srcs + ["-o", so])
    return C.CDLL(so)


def stacked_pattern(n, contact=True):
    """two matN sheets, the node graph of their tets plus (contact) the pairs of facing surface nodes within 1.6 grid spacings:
    what a contact pattern of the stepper looks like.  Returns V, node adjacency (CSR, symmetric, no diagonal), block-structured upper scalar CSR."""
    V, F, nA = scene.make_mat_stack(n, 2, gap=1.2e-3)
    nn = V.shape[0]
    I = np.concatenate([F[:, a] for a in range(4) for b in range(4)])
    J = np.concatenate([F[:, b] for a in range(4) for b in range(4)])
    if contact:
        top = np.where((np.arange(nn) < nA) & (V[:, 1] > V[:nA, 1].mean()))[0]
        bot = np.where((np.arange(nn) >= nA) & (V[:, 1] < V[nA:, 1].mean()))[0]
        near = cKDTree(V[top][:, [0, 2]]).query_ball_point(V[bot][:, [0, 2]], 1.6 / (n - 1))
        ci = np.concatenate([np.full(len(x), bot[i]) for i, x in enumerate(near)])
        cj = np.concatenate([top[x] for x in near])
        I, J = np.concatenate([I, ci, cj]), np.concatenate([J, cj, ci])
    A = sp.coo_matrix((np.ones(len(I)), (I, J)), shape=(nn, nn)).tocsr()
    A.sum_duplicates()
    U = sp.triu(A, format="csr")  # with the diagonal
    rows = []
    for u in range(nn):
        cols = (3 * U.indices[U.indptr[u]:U.indptr[u + 1]][:, None] + np.arange(3)[None, :]).ravel()
        rows += [cols, cols[1:], cols[2:]]
    ia = np.zeros(3 * nn + 1, np.int32)
    ia[1:] = np.cumsum([len(r) for r in rows])
    ja = np.concatenate(rows).astype(np.int32)
    G = (A - sp.diags(A.diagonal())).tocsr()
    G.eliminate_zeros()
    return np.ascontiguousarray(V, np.float64), G, ia, ja


def analyze(shim, ia, ja, coords, leaf=8):
    sizes = np.zeros(4, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = shim.shim_analyze(C.c_int(len(ia) - 1), p(ia), p(ja), p(coords) if coords is not None else None, C.c_int(leaf), p(sizes))
    assert rc == 0
    ns, nn, nIdx, nChild = [int(x) for x in sizes]
    o = dict(newOf=np.zeros(nn, np.int32), firstNode=np.zeros(ns + 1, np.int32), parent=np.zeros(ns, np.int32), idxPtr=np.zeros(ns + 1, np.int32),
             idx=np.zeros(nIdx, np.int32), childPtr=np.zeros(ns + 1, np.int32), child=np.zeros(max(nChild, 1), np.int32), aDst=np.zeros(ia[-1], np.int64),
             aFront=np.zeros(ia[-1], np.int32), frontOff=np.zeros(ns + 1, np.int64))
    shim.shim_fetch(*[p(o[k]) for k in ("newOf", "firstNode", "parent", "idxPtr", "idx", "childPtr", "child", "aDst", "aFront", "frontOff")])
    o["ns"], o["nn"] = ns, nn
    return o


def check(o, G, ia, ja):
    ns, nn = o["ns"], o["nn"]
    newOf, first = o["newOf"], o["firstNode"]
    assert sorted(newOf.tolist()) == list(range(nn)), "ordering is no permutation"
    oldOf = np.argsort(newOf)
    frontOf = np.repeat(np.arange(ns), np.diff(first))
    # (1) true fill of the node graph under this ordering, by node: struct(j) = (adj(j) U struct(children in the elimination tree)) \ {<= j};
    #     every front must hold the structure of each of its nodes (nothing of L falls outside the fronts)
    above = [None] * nn
    kids = [[] for _ in range(nn)]
    for j in range(nn):
        ov = oldOf[j]
        s = set(int(w) for w in newOf[G.indices[G.indptr[ov]:G.indptr[ov + 1]]] if w > j)
        for c in kids[j]:
            s |= above[c]
            above[c] = None
        s.discard(j)
        above[j] = s
        if s:
            kids[min(s)].append(j)
        f = frontOf[j]
        have = o["idx"][o["idxPtr"][f]:o["idxPtr"][f + 1]]
        assert s <= set(have.tolist()), f"fill of node {j} outside front {f}"
    # (2) the fronts as the analysis defines them (own nodes taken as one clique), restated on the quotient graph; parents and children lists
    st = [None] * ns
    fk = [[] for _ in range(ns)]
    for f in range(ns):
        end = first[f + 1]
        s = set()
        for v in range(first[f], end):
            ov = oldOf[v]
            s.update(int(w) for w in newOf[G.indices[G.indptr[ov]:G.indptr[ov + 1]]] if w >= end)
        for c in fk[f]:
            s.update(w for w in st[c] if w >= end)
        st[f] = s
        have = o["idx"][o["idxPtr"][f]:o["idxPtr"][f + 1]]
        own = np.arange(first[f], end)
        assert np.array_equal(have[:len(own)], own)
        assert have[len(own):].tolist() == sorted(s), f"structure of front {f}"
        par = frontOf[min(s)] if s else -1
        assert o["parent"][f] == par
        if par >= 0:
            fk[par].append(f)
    for f in range(ns):
        assert o["child"][o["childPtr"][f]:o["childPtr"][f + 1]].tolist() == fk[f], f"children of front {f} (ascending)"
    # (3) entry destinations: CSR entry (r, c) of the upper triangle -> slot (row pr, column pc) of the permuted lower triangle, inside the front of pc
    rows = np.repeat(np.arange(len(ia) - 1), np.diff(ia))
    pr = 3 * newOf[rows // 3].astype(np.int64) + rows % 3
    pc = 3 * newOf[ja // 3].astype(np.int64) + ja % 3
    i, j = np.maximum(pr, pc), np.minimum(pr, pc)
    f = frontOf[j // 3]
    assert np.array_equal(o["aFront"], f)
    N = 3 * np.diff(o["idxPtr"]).astype(np.int64)
    rng = np.random.default_rng(5)
    for k in rng.integers(0, len(ja), 4000):
        fr = f[k]
        nodes = o["idx"][o["idxPtr"][fr]:o["idxPtr"][fr + 1]]
        lr = 3 * int(np.searchsorted(nodes, i[k] // 3)) + int(i[k] % 3)
        assert nodes[lr // 3] == i[k] // 3
        lc = int(j[k] - 3 * first[fr])
        assert o["aDst"][k] == o["frontOff"][fr] + lr + N[fr] * lc
    assert len(np.unique(o["aDst"])) == len(ja), "two matrix entries share a slot"


@pytest.mark.parametrize("n,contact,geometric,leaf", [(12, True, True, 8), (70, True, True, 8), (70, True, False, 8), (70, False, True, 32)])
def test_fronts_and_destinations_against_a_python_restatement(shim, n, contact, geometric, leaf):
    V, G, ia, ja = stacked_pattern(n, contact)
    o = analyze(shim, ia, ja, V if geometric else None, leaf)
    check(o, G, ia, ja)


def test_analysis_is_reproducible(shim):
    """threads must not change the result: twice the same arrays"""
    V, G, ia, ja = stacked_pattern(70)
    a = analyze(shim, ia, ja, V)
    b = analyze(shim, ia, ja, V)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("world", [2, 3, 5, 6, 8, 16, 64])
def test_exchange_plans_of_all_ranks_fit_together(shim, world):
    """The point-to-point plan of the sharded solver (mf_assign_owners -> mf_assign_executors -> mf_exchange_plan, ipc_amd/csrc/mf_symbolic.cpp), every rank's copy
    side by side -- world sizes the multi-process tests do not reach, odd ones included:
      * every front has exactly one executor, and it is a member of the front's group; a front below the cut is executed by its owner;
      * level by level and pair of ranks by pair of ranks, what rank a sends to rank b is what rank b receives from rank a -- same fronts, IN THE SAME ORDER (RCCL
        matches the sends and receives of a group between two ranks by their order), for the update matrices and for the solution segments;
      * nobody sends to itself; a child's update travels exactly when its parent is executed elsewhere, and then exactly once."""
    V, G, ia, ja = stacked_pattern(24, contact=True)
    o = analyze(shim, ia, ja, V, leaf=12)
    ns = o["ns"]
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    plans, execs = [], None
    for rank in range(world):
        exec_, group, level = np.zeros(ns, np.int32), np.zeros(ns, np.uint64), np.zeros(ns, np.int32)
        cap = 8 * ns + 64
        rec = np.zeros(7 * cap, np.int32)
        n = shim.shim_exchange_plan(C.c_int(world), C.c_int(rank), p(exec_), p(group), p(level), p(rec), C.c_int(cap))
        assert n >= 0
        plans.append(rec[: 7 * n].reshape(n, 7).copy())
        if execs is None:
            execs, groups, levels = exec_.copy(), group.copy(), level.copy()
        else:
            assert np.array_equal(execs, exec_) and np.array_equal(groups, group)  # every rank derives the same assignment
    owner, nodeOwner = np.zeros(ns, np.int32), np.zeros(o["nn"], np.int32)
    shim.shim_owners.restype = C.c_double
    shim.shim_owners(C.c_int(world), p(owner), p(nodeOwner))
    assert ((execs >= 0) & (execs < world)).all()
    assert all((int(groups[s]) >> int(execs[s])) & 1 for s in range(ns))
    below = owner >= 0
    assert np.array_equal(execs[below], owner[below])
    parent = o["parent"]
    # what has to travel: the update of a child whose parent another rank executes
    must = {(int(s), int(execs[s]), int(execs[parent[s]])) for s in range(ns) if parent[s] >= 0 and execs[parent[s]] != execs[s]}
    sent = set()
    for a in range(world):
        for b in range(world):
            for ks, kr in ((0, 1), (2, 3)):
                out = [(int(r[0]), int(r[2])) for r in plans[a] if r[1] == ks and r[6] == b]
                inn = [(int(r[0]), int(r[2])) for r in plans[b] if r[1] == kr and r[6] == a]
                assert out == inn, (a, b, ks, out[:5], inn[:5])
                if a == b:
                    assert not out
                if ks == 0:
                    for lvl, s in out:
                        assert lvl == levels[s] and (s, a, b) in must and (s, a, b) not in sent
                        sent.add((s, a, b))
    assert sent == must
    if world <= 16:
        assert len(must) > 0  # the cut is not empty on this pattern
