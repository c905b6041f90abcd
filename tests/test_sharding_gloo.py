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
