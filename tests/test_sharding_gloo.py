"""world_size-2 gloo test of the multi-GPU sharding protocol (CPU, no GPU needed).

bench.py shards the assembly by element ownership and sums the ranks' partial gradient / CSR values with
torch.distributed all-reduces (RCCL on the GPU box).  The same protocol is run here on the `gloo` backend with
the CPU oracle standing in for each rank's kernels: partial results, summed, must equal the unsharded assembly,
and scalar step bounds combine with MIN / energies with SUM exactly as `ipcgpu_opt_set_allreduce` does.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ipc_amd import scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    V, F = scene.make_bar(8, 2, 2, size=(4.0, 0.5, 1.0))
    Vt = scene.twist_state(scene.jitter(V, F), 0.3)
    left, right = scene.border_verts(V, 0.01)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_dbc(np.concatenate([left, right]), 2)
    m.set_V(Vt)
    ia, ja = m.pattern()
    xt = Vt + 1e-3 * np.random.default_rng(5).normal(size=Vt.shape)
    nT = F.shape[0]
    t0, t1 = nT * rank // world, nT * (rank + 1) // world  # the shard rule of ipcgpu_ctx_set_shard
    a, g = orc.assemble_shard(m, len(ja), 0.025 ** 2, True, t0, t1, rank == 0, xt)
    ta, tg = torch.from_numpy(a), torch.from_numpy(g)
    dist.all_reduce(ta, op=dist.ReduceOp.SUM)
    dist.all_reduce(tg, op=dist.ReduceOp.SUM)
    # scalar reductions: energy (sum of per-shard partial sums) and the inversion step bound (min)
    _, pe = m.elastic_energy(1.0, per_elem=True)
    e = torch.tensor([pe[t0:t1].sum()], dtype=torch.float64)
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    p = np.random.default_rng(6).normal(size=3 * V.shape[0])
    steps = m.inversion_step(p, 0.2)
    s = torch.tensor([steps[t0:t1].min()], dtype=torch.float64)
    dist.all_reduce(s, op=dist.ReduceOp.MIN)
    # timing protocol of bench.py: MAX over ranks
    tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        a_full, g_full = orc.assemble_shard(m, len(ja), 0.025 ** 2, True, 0, nT, True, xt)
        a_ref = m.assemble_hessian(len(ja), 0.025 ** 2, True)
        out.put(dict(
            a_err=float(np.abs(ta.numpy() - a_full).max() / np.abs(a_full).max()),
            a_ref_err=float(np.abs(a_full - a_ref).max() / np.abs(a_ref).max()),
            g_err=float(np.abs(tg.numpy() - g_full).max() / np.abs(g_full).max()),
            e_err=float(abs(e.item() - pe.sum()) / abs(pe.sum())),
            s_ok=bool(s.item() == steps.min()), tmax=float(tmax.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_element_sharded_assembly_sums_to_the_whole():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["a_err"] < 1e-13 and res["g_err"] < 1e-12 and res["e_err"] < 1e-13
    assert res["a_ref_err"] < 1e-13  # the sharded restatement equals computePrecondMtr's assembly
    assert res["s_ok"] and res["tmax"] == 2.0


def test_patch_shards_partition_every_patch_once():
    # the rule used by HipOptimizer::patchShard / ipcgpu_ctx_set_shard
    for n in (1, 7, 256, 4375):
        for w in (1, 2, 4, 8):
            cover = []
            for r in range(w):
                cover += list(range(n * r // w, n * (r + 1) // w))
            assert cover == list(range(n))


# ---- owner-computes (round 4): a rank assembles the complete CSR rows of the nodes its subtrees eliminate plus the separator rows above the cut; its fronts
# read nothing else, so NO matrix value crosses ranks; the gradient is exchanged once, every node contributed by its designated rank.
# (HipOptimizer::ensureOwnerPlan / computePrecondMtr, MfNumeric::setup.)  The analysis and the owner assignment are the product's own host code
# (mf_symbolic.cpp through tests/mf_symbolic/shim.cpp); the CPU oracle's assembly stands in for a rank's kernels, masked to the rows that rank writes.
def _shim_lib():
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "mf_symbolic", "_build", "libmfsym.so")
    srcs = [os.path.join(here, "mf_symbolic", "shim.cpp"), os.path.join(os.path.dirname(here), "ipc_amd", "csrc", "mf_symbolic.cpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread"] + srcs + ["-o", so])
    return C.CDLL(so)


def _owner_plan(world, rank):
    """what one rank knows: mesh, pattern, analysis, owners (all deterministic: every rank computes the same), its masks, its rows"""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import orc
    from test_mf_symbolic import analyze
    V, F, nA = scene.make_mat_stack(14, 2, gap=1.2e-3)
    X = scene.jitter(V, F, rel=3e-2)
    nn = V.shape[0]
    low = np.arange(nA)[V[:nA, 1] > V[:nA, 1].mean()]
    up = nA + np.arange(nn - nA)[V[nA:, 1] < V[nA:, 1].mean()]
    k = min(len(low), len(up))
    pairs = np.stack([low[:k], up[:k]], 1).astype(np.int32)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    m.set_dbc(np.where(V[:, 0] < -0.49)[0].astype(np.int32), 1)
    m.set_V(X)
    ia, ja = m.pattern(extra_edges=pairs)
    ia, ja = np.ascontiguousarray(ia, np.int32), np.ascontiguousarray(ja, np.int32)
    L = _shim_lib()
    L.shim_owners.restype = C.c_double
    o = analyze(L, ia, ja, np.ascontiguousarray(V, np.float64), 8)
    frontOwner, nodeOwner = np.zeros(o["ns"], np.int32), np.zeros(nn, np.int32)
    shared = L.shim_owners(C.c_int(world), frontOwner.ctypes.data_as(C.c_void_p), nodeOwner.ctypes.data_as(C.c_void_p))
    need = (nodeOwner < 0) | (nodeOwner == rank)  # rows this rank assembles (HipOptimizer::ensureOwnerPlan)
    mine = (nodeOwner == rank) | ((nodeOwner < 0) & (rank == 0))  # the rank that contributes a node to an exchange
    a = m.assemble_hessian(len(ja), 0.01 ** 2, True)
    g = m.elastic_gradient(0.01 ** 2, True)
    rowNode = np.repeat(np.arange(nn), np.diff(ia)[0::3] + np.diff(ia)[1::3] + np.diff(ia)[2::3])
    return dict(o=o, frontOwner=frontOwner, nodeOwner=nodeOwner, shared=shared, need=need, mine=mine, a=a, g=g, rowNode=rowNode, nn=nn, ja=ja)


def _owner_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = _owner_plan(world, rank)
    a, g, need, mine, rowNode = P["a"], P["g"], P["need"], P["mine"], P["rowNode"]
    a_r = np.where(need[rowNode], a, 0.0)  # what this rank's assembly writes: complete rows of its nodes and of the shared separators, zeros elsewhere
    g_r = np.where(np.repeat(need, 3), g.reshape(-1), 0.0)
    # (1) everything the fronts this rank factorises read is already there, bit for bit -- nothing to exchange
    fo = P["frontOwner"][P["o"]["aFront"]]
    reads = (fo == rank) | (fo < 0)
    complete = bool(np.array_equal(a_r[reads], a[reads]))
    # (2) the gradient: every node contributed once, by its designated rank; ONE all-reduce of 3 nV doubles
    tg = torch.from_numpy(np.where(np.repeat(mine, 3), g_r, 0.0))
    wire = tg.numel() * 8
    dist.all_reduce(tg, op=dist.ReduceOp.SUM)
    designated = torch.from_numpy(mine.astype(np.float64))
    dist.all_reduce(designated, op=dist.ReduceOp.SUM)
    # (3) a consumer of the WHOLE matrix (HipOptimizer::completeMatrix): designated rows summed
    ta = torch.from_numpy(np.where(mine[rowNode], a_r, 0.0))
    dist.all_reduce(ta, op=dist.ReduceOp.SUM)
    # (4) every entry is read by some rank's front
    cover = torch.from_numpy(reads.astype(np.float64))
    dist.all_reduce(cover, op=dist.ReduceOp.MAX)
    res = dict(rank=rank, complete=complete, g_equal=bool(np.array_equal(tg.numpy(), g.reshape(-1))), once=bool((designated.numpy() == 1.0).all()),
               a_equal=bool(np.array_equal(ta.numpy(), a)), covered=bool((cover.numpy() == 1.0).all()), wire=wire, nn=P["nn"],
               rows=float(need.mean()), shared=float(P["shared"]), need_mine=bool((need | ~mine).all()))
    out.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_owner_computes_protocol_on_gloo(world):
    _shim_lib()  # built once here, not by the ranks side by side
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r["complete"], f"rank {r['rank']}: a front of this rank reads a matrix entry the rank did not assemble"
        assert r["g_equal"] and r["once"] and r["a_equal"] and r["covered"] and r["need_mine"]
        assert r["wire"] == 3 * r["nn"] * 8  # the one exchange of the assembly: the gradient
        assert 0.0 < r["shared"] < 0.9
    # the ranks split the rows: each assembles its subtrees + the shared separators, far from everything at 4 ranks
    assert sum(r["rows"] for r in res) < world * 0.95
    assert max(r["rows"] for r in res) < (0.8 if world == 2 else 0.6)


# ---- round 5: the sharded direct solver sends POINT TO POINT.  A front above the cut is executed by one rank (mf_assign_executors); the packed update matrix of
# a child another rank computed, its update vector and the solution entries of the separators travel to exactly the ranks that need them
# (mf_exchange_plan -> MfNumeric::exchange -> ncclSend / ncclRecv groups).  Here the product's own plan (mf_symbolic.cpp through the shim) drives a dense
# multifrontal factorisation in numpy, one process per rank over gloo: every rank factorises and solves ONLY the fronts the plan gives it and moves ONLY what the
# plan lists.  Anything the plan forgot would be missing (None / NaN) where a front needs it, and the solution would not be the direct one.
def _p2p_worker(rank, world, port, out):
    import ctypes as C
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import orc
    from test_mf_symbolic import analyze
    V, F = scene.make_mat(16)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    ia, ja = m.pattern()
    ia, ja = np.ascontiguousarray(ia, np.int32), np.ascontiguousarray(ja, np.int32)
    L = _shim_lib()
    o = analyze(L, ia, ja, np.ascontiguousarray(V, np.float64), 8)
    ns, nn = o["ns"], o["nn"]
    n = 3 * nn
    exec_, group, level = np.zeros(ns, np.int32), np.zeros(ns, np.uint64), np.zeros(ns, np.int32)
    cap = 8 * ns + 64
    rec = np.zeros(7 * cap, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    nrec = L.shim_exchange_plan(C.c_int(world), C.c_int(rank), p(exec_), p(group), p(level), p(rec), C.c_int(cap))
    assert nrec >= 0
    rec = rec[:7 * nrec].reshape(-1, 7)
    # an SPD matrix on the pattern, the same on every rank; permuted to the elimination order
    rng = np.random.default_rng(11)
    row = np.repeat(np.arange(n), np.diff(ia))
    val = rng.uniform(-1.0, 1.0, len(ja))
    A = np.zeros((n, n))
    A[row, ja] = val
    A = A + A.T
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(1) + 1.0
    b = rng.normal(size=n)
    newS = (3 * o["newOf"][:, None] + np.arange(3)[None, :]).ravel()  # new scalar index of old scalar index
    oldS = np.argsort(newS)
    Ap, bp = A[np.ix_(oldS, oldS)], b[oldS]
    first, parent = o["firstNode"], o["parent"]
    fronts = []
    for s in range(ns):
        nodes = o["idx"][o["idxPtr"][s]:o["idxPtr"][s + 1]]
        assert np.array_equal(nodes[:first[s + 1] - first[s]], np.arange(first[s], first[s + 1]))  # own nodes lead the front's index list
        fronts.append((3 * nodes[:, None] + np.arange(3)[None, :]).ravel())
    nc = 3 * np.diff(first)
    kids = [o["child"][o["childPtr"][s]:o["childPtr"][s + 1]] for s in range(ns)]
    nLevels = int(level.max()) + 1
    mine = exec_ == rank
    U, W, Lf = [None] * ns, [None] * ns, [None] * ns
    moved = dict(sent=0, received=0)

    def group_exchange(items):  # items: (tensor, peer, send)
        if not items:
            return
        reqs = [dist.P2POp(dist.isend if snd else dist.irecv, t, peer) for t, peer, snd in items]
        for r in dist.batch_isend_irecv(reqs):
            r.wait()
        for t, peer, snd in items:
            moved["sent" if snd else "received"] += 8 * t.numel()

    def recs(l, kind):
        return rec[(rec[:, 0] == l) & (rec[:, 1] == kind)]

    def tri(mm):
        jj, ii = np.triu_indices(mm)  # column by column of the lower triangle
        return ii, jj

    # ---- factorisation, level by level
    for l in range(nLevels):
        for s in np.nonzero(mine & (level == l))[0]:
            I, k = fronts[s], nc[s]
            Fm = np.zeros((len(I), len(I)))
            Fm[:, :k] = Ap[np.ix_(I, I[:k])]
            Fm[:k, :] = Fm[:, :k].T
            pos = {int(g): i for i, g in enumerate(I)}
            for c in kids[s]:
                Ic = fronts[c][nc[c]:]
                loc = np.array([pos[int(g)] for g in Ic], dtype=np.int64)
                assert U[c] is not None, f"rank {rank}: front {s} needs the update matrix of child {c}, which nobody sent"
                Fm[np.ix_(loc, loc)] += U[c]
            L11 = np.linalg.cholesky(Fm[:k, :k])
            L21 = np.linalg.solve(L11, Fm[k:, :k].T).T
            U[s] = Fm[k:, k:] - L21 @ L21.T
            Lf[s] = (L11, L21)
        items, unpack = [], []
        off_seen = []
        for r in recs(l, 0):
            s, mm = int(r[2]), len(fronts[int(r[2])]) - nc[int(r[2])]
            assert mine[s] and exec_[parent[s]] == r[6] != rank
            ii, jj = tri(mm)
            items.append((torch.from_numpy(np.ascontiguousarray(U[s][ii, jj])), int(r[6]), True))
            off_seen.append(((int(r[4]) << 32) | (int(r[3]) & 0xffffffff), mm * (mm + 1) // 2))
        for r in recs(l, 1):
            s, mm = int(r[2]), len(fronts[int(r[2])]) - nc[int(r[2])]
            assert exec_[parent[s]] == rank and exec_[s] == r[6] != rank
            t = torch.empty(mm * (mm + 1) // 2, dtype=torch.float64)
            items.append((t, int(r[6]), False))
            unpack.append((s, mm, t))
            off_seen.append(((int(r[4]) << 32) | (int(r[3]) & 0xffffffff), mm * (mm + 1) // 2))
        off_seen.sort()
        for (o0, c0), (o1, _) in zip(off_seen, off_seen[1:]):
            assert o0 + c0 <= o1, "staging regions of one level overlap"
        group_exchange(items)
        for s, mm, t in unpack:
            ii, jj = tri(mm)
            Um = np.zeros((mm, mm))
            Um[ii, jj] = t.numpy()
            U[s] = Um + np.tril(Um, -1).T
    # ---- forward sweep
    y = np.full(n, np.nan)
    for l in range(nLevels):
        for s in np.nonzero(mine & (level == l))[0]:
            I, k = fronts[s], nc[s]
            w = np.zeros(len(I))
            w[:k] = bp[I[:k]]
            pos = {int(g): i for i, g in enumerate(I)}
            for c in kids[s]:
                assert W[c] is not None, f"rank {rank}: front {s} needs the update vector of child {c}"
                loc = np.array([pos[int(g)] for g in fronts[c][nc[c]:]], dtype=np.int64)
                np.add.at(w, loc, W[c])
            L11, L21 = Lf[s]
            ys = np.linalg.solve(L11, w[:k])
            y[I[:k]] = ys
            W[s] = w[k:] - L21 @ ys
        items, unpack = [], []
        for r in recs(l, 0):
            items.append((torch.from_numpy(np.ascontiguousarray(W[int(r[2])])), int(r[6]), True))
        for r in recs(l, 1):
            s = int(r[2])
            t = torch.empty(len(fronts[s]) - nc[s], dtype=torch.float64)
            items.append((t, int(r[6]), False))
            unpack.append((s, t))
        group_exchange(items)
        for s, t in unpack:
            W[s] = t.numpy()
    # ---- backward sweep
    x = np.full(n, np.nan)
    for l in range(nLevels - 1, -1, -1):
        for s in np.nonzero(mine & (level == l))[0]:
            I, k = fronts[s], nc[s]
            L11, L21 = Lf[s]
            x[I[:k]] = np.linalg.solve(L11.T, y[I[:k]] - L21.T @ x[I[k:]])
        items, unpack = [], []
        for r in recs(l, 2):
            s = int(r[2])
            items.append((torch.from_numpy(np.ascontiguousarray(x[fronts[s][:nc[s]]])), int(r[6]), True))
        for r in recs(l, 3):
            s = int(r[2])
            t = torch.empty(nc[s], dtype=torch.float64)
            items.append((t, int(r[6]), False))
            unpack.append((s, t))
        group_exchange(items)
        for s, t in unpack:
            x[fronts[s][:nc[s]]] = t.numpy()
    # ---- the solution every rank ends up with: each entry contributed by its executor (one all-reduce, as in MfNumeric::enqueueBackward)
    nodeExec = np.repeat(exec_, np.diff(first))
    own = np.repeat(nodeExec == rank, 3)
    assert not np.isnan(x[own]).any(), f"rank {rank}: a solution entry of its own fronts was never computed"
    tx = torch.from_numpy(np.where(own, x, 0.0))
    dist.all_reduce(tx, op=dist.ReduceOp.SUM)
    xref = np.linalg.solve(Ap, bp)
    tot = torch.tensor([float(moved["sent"]), float(moved["received"])], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    out.put(dict(rank=rank, err=float(np.abs(tx.numpy() - xref).max() / np.abs(xref).max()), sent=moved["sent"], received=moved["received"],
                 sent_all=float(tot[0]), received_all=float(tot[1]), fronts=int(mine.sum()), ns=ns,
                 above=int(((exec_ == rank) & (group != (np.uint64(1) << np.uint64(rank)))).sum()), full_matrix_bytes=8 * len(ja)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_point_to_point_solver_protocol_on_gloo(world):
    _shim_lib()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r["err"] < 1e-11, r  # the direct solution, on every rank
        assert r["sent_all"] == r["received_all"] > 0  # every byte sent is received exactly once
        assert 0 < r["fronts"] < r["ns"]
    assert sum(r["fronts"] for r in res) == res[0]["ns"]  # every front executed by exactly one rank: nothing above the cut is repeated
    assert sum(r["above"] for r in res) >= 1
