"""The N > 1 code path of the product on real hardware: two ranks (both on GPU 0, collectives over gloo so that a one-GPU box
can run it) shard the patches, all-reduce gradient / CSR values / scalars through the C-ABI hook and must reproduce the
single-rank trajectory.  bench.py binds the same hook to RCCL."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["IPC_REPO"])
import ipc_amd
from ipc_amd import scene

class DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
V, F = scene.make_bar(12, 2, 2, size=(5.0, 0.5, 1.0))
left, right = scene.border_verts(V, 0.01)
Vs = scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.15)
c = ipc_amd.Context(0)
if world > 1:
    c.set_shard(rank, world)
    def hook(ptr, count, op):
        t = torch.as_tensor(DevPtr(ptr, count), device="cuda:0")
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN)
        t.copy_(h)
        torch.cuda.synchronize()
        return 0
    c.set_allreduce(hook)
c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
c.set_positions(Vs)
c.opt_init(0.025, False)
c.set_twist(left, right)
c.precompute()
iters = []
for step in range(2):
    iters.append(c.solve_timestep(40))
s = c.state()
if rank == 0:
    np.savez(os.environ["OUT"], V=s["V"], E=s["E"], g=s["gradient"], iters=np.array(iters))
c.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def run(world, out):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(WORKER)
        script = f.name
    env = dict(os.environ, IPC_REPO=repo, OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    os.unlink(script)
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]


def test_two_ranks_reproduce_the_single_rank_trajectory():
    with tempfile.TemporaryDirectory() as d:
        one, two = os.path.join(d, "one.npz"), os.path.join(d, "two.npz")
        run(1, one)
        run(2, two)
        a, b = np.load(one), np.load(two)
        assert np.array_equal(a["iters"], b["iters"])
        scale = np.abs(a["V"]).max()
        assert np.abs(a["V"] - b["V"]).max() <= 1e-11 * scale
        assert abs(a["E"] - b["E"]) <= 1e-11 * abs(a["E"])
