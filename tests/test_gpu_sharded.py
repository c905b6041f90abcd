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
mode = os.environ.get("MODE", "assembly")
c = ipc_amd.Context(0)
if world > 1:
    def hook(ptr, count, op):
        t = torch.as_tensor(DevPtr(ptr, count), device="cuda:0")
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN)
        t.copy_(h)
        torch.cuda.synchronize()
        return 0
    def xhook(ops):
        # the solver's point-to-point group (ipcgpu_opt_set_exchange): sends / receives of device buffers, bounced through the host for gloo
        reqs, recvs = [], []
        for ptr, count, peer, send in ops:
            t = torch.as_tensor(DevPtr(ptr, count), device="cuda:0")
            if send:
                reqs.append(dist.P2POp(dist.isend, t.cpu(), peer))
            else:
                h = torch.empty(count, dtype=torch.float64)
                recvs.append((t, h))
                reqs.append(dist.P2POp(dist.irecv, h, peer))
        for r in dist.batch_isend_irecv(reqs):
            r.wait()
        for t, h in recvs:
            t.copy_(h)
        torch.cuda.synchronize()
        return 0
    c.set_exchange(xhook)
    if mode in ("assembly", "contact_sharded", "owner", "capi_contact"):
        c.set_shard(rank, world)  # the assembly sharded: with the solver sharded as well, owner-computes rows (no matrix value crosses ranks)
    c.set_allreduce(hook)
    if mode != "assembly":
        c.set_solver_shard(rank, world)  # subtree-sharded factorisation and solves
extra = {}
if mode in ("assembly", "solver", "owner"):
    n = 12 if mode == "assembly" else 40
    V, F = scene.make_bar(12, 2, 2, size=(5.0, 0.5, 1.0)) if mode == "assembly" else scene.make_mat(n)
    left, right = scene.border_verts(V, 0.01)
    Vs = scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.15)
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.025, False)
    c.set_twist(left, right)
    c.precompute()
    if mode == "solver":  # the solver alone first: residual of one solve, not-PD agreed on by all ranks
        rows, nnz = c.get_dims()
        c.assemble_newton(0.025 ** 2, True, with_gradient=False)
        c.analyze_pattern()
        assert c.factorize()
        b = np.random.default_rng(5).normal(size=rows)
        x = c.solve(b)
        extra["res"] = np.linalg.norm(c.multiply(x) - b) / np.linalg.norm(b)
        extra["x"] = x
        extra["shared"] = c.solver_shard_stats()["shared_flop_fraction"]
        a = c.get_a()
        ia, ja = c.get_pattern()
        k = ia[3 * (rows // 6)]
        c.set_coeff(3 * (rows // 6), 3 * (rows // 6), -abs(a[k]))
        extra["notpd"] = not c.factorize()
        c.set_coeff(3 * (rows // 6), 3 * (rows // 6), a[k])
    steps, cap = 2, 40
else:  # contact: two slabs, the upper one dropped on the clamped lower one (pattern changes -> re-analysis with a new cut)
    def two_blocks(gap, n=3, shift=0.13):
        Va, Fa = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(0, 0, 0))
        Vb, Fb = scene.make_box(n, 1, n, size=(1.0, 0.3, 1.0), origin=(shift, 0.3 + gap, 0.5 * shift))
        return np.vstack([Va, Vb]), np.vstack([Fa, Fb + Va.shape[0]])
    V, F = two_blocks(0.03)
    Vs = scene.jitter(V, F, rel=1e-2)
    SF = scene.surface_tris(F)
    nA = V.shape[0] // 2
    bottom = np.nonzero(V[:nA, 1] < V[:nA, 1].min() + 0.02)[0].astype(np.int32)
    vel = np.zeros_like(V)
    vel[nA:, 1] = -6.0
    c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    c.set_dbc(bottom, 1)
    c.set_positions(Vs)
    c.opt_init(0.02, True)
    c.set_surface(SF)
    c.enable_self_collision(1e-2)
    c.set_velocity(vel)
    c.precompute()
    steps, cap = 4, 60
    if mode == "capi_contact":
        # the direct C entry points on a sharded context: the WHOLE barrier gradient / Hessian blocks, not a rank's share (ADVICE round 3)
        for step in range(3):
            c.solve_timestep(cap)
        extra["nActive0"] = c.contact_state()["nActive"]
        st0 = c.state()
        dHat, kappa = st0["dHat"], st0["kappa"]
        extra["cg"] = c.contact_gradient_add(dHat, kappa, True)
        c.set_zero()
        c.contact_hessian_add(dHat, kappa, True)
        extra["ca"] = c.get_a()
        steps = 0
iters = []
for step in range(steps):
    iters.append(c.solve_timestep(cap))
s = c.state()
if mode.startswith("contact"):
    extra["nActive"] = c.contact_state()["nActive"]
    extra["nPatternChanges"] = c.contact_state()["nPatternChanges"]
cm = c.comm_stats()
rows, nnz = c.get_dims()
xs = c.solver_exchange_stats()
extra.update(stepper_bytes=cm["stepper_bytes"], solver_bytes=cm["solver_bytes"], rows_nodes=cm["rows_assembled_nodes"], nodes=cm["nodes"], nnz=nnz,
             solver_sent=xs["sent_bytes"], solver_received=xs["received_bytes"])
if world > 1:  # what every rank moved through the solver, gathered on rank 0
    allx = [None] * world
    dist.all_gather_object(allx, (xs["sent_bytes"], xs["received_bytes"], cm["solver_bytes"]))
    extra.update(solver_sent_all=np.array([a[0] for a in allx]), solver_received_all=np.array([a[1] for a in allx]), solver_bytes_all=np.array([a[2] for a in allx]))
if rank == 0:
    np.savez(os.environ["OUT"], V=s["V"], E=s["E"], g=s["gradient"], iters=np.array(iters), **extra)
c.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def run(world, out, mode="assembly"):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(WORKER)
        script = f.name
    env = dict(os.environ, IPC_REPO=repo, OUT=out, MODE=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
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


@pytest.mark.parametrize("world", [2, 4])
def test_subtree_sharded_solver_reproduces_the_single_rank_run(world):
    """ipcgpu_linsys_set_shard: the assembly tree cut below its top separators, every rank factorising and solving its own subtrees
    (update matrices / vectors of the subtree roots and the final solution crossing ranks through the hook).  The per-front
    arithmetic is the single-rank one, so the solution and the whole Newton trajectory agree to round-off."""
    with tempfile.TemporaryDirectory() as d:
        one, many = os.path.join(d, "one.npz"), os.path.join(d, "many.npz")
        run(1, one, "solver")
        run(world, many, "solver")
        a, b = np.load(one), np.load(many)
        assert a["res"] < 1e-11 and b["res"] < 1e-11
        assert np.abs(a["x"] - b["x"]).max() <= 1e-12 * np.abs(a["x"]).max()
        assert bool(a["notpd"]) and bool(b["notpd"])
        assert 0.0 < float(b["shared"]) < 0.7 and float(a["shared"]) == 0.0  # a real cut: part of the tree, not all of it, is repeated
        assert np.array_equal(a["iters"], b["iters"])
        assert np.abs(a["V"] - b["V"]).max() <= 1e-11 * np.abs(a["V"]).max()
        assert abs(a["E"] - b["E"]) <= 1e-11 * abs(a["E"])


def test_two_ranks_with_contact_reproduce_the_single_rank_trajectory():
    """The same with self-contact: constraint sets, barrier terms and CCD replicated, the solver sharded, the pattern (and with it
    the cut of the assembly tree) changing while the slabs come into contact."""
    with tempfile.TemporaryDirectory() as d:
        one, two = os.path.join(d, "one.npz"), os.path.join(d, "two.npz")
        run(1, one, "contact")
        run(2, two, "contact")
        a, b = np.load(one), np.load(two)
        assert int(a["nActive"]) > 0 and int(a["nActive"]) == int(b["nActive"])
        assert int(a["nPatternChanges"]) >= 1
        assert np.array_equal(a["iters"], b["iters"])
        assert np.abs(a["V"] - b["V"]).max() <= 1e-10 * np.abs(a["V"]).max()


@pytest.mark.parametrize("world", [2, 4])
def test_contact_pair_lists_sharded_reproduce_the_single_rank_trajectory(world):
    """ipcgpu_ctx_set_shard with self-contact on top of the subtree-sharded solver: the elements AND the contact-pair lists are split over the
    ranks by row ownership -- a rank evaluates the stencils that touch a node whose rows it holds (stencils on a cut by both sides) and adds
    their blocks to those rows; the barrier energy is split by index and summed as a scalar.  Same sets, same Newton counts, positions to
    the round-off of a different summation order."""
    with tempfile.TemporaryDirectory() as d:
        one, many = os.path.join(d, "one.npz"), os.path.join(d, "many.npz")
        run(1, one, "contact_sharded")
        run(world, many, "contact_sharded")
        a, b = np.load(one), np.load(many)
        assert int(a["nActive"]) > 0 and int(a["nActive"]) == int(b["nActive"])
        assert np.array_equal(a["iters"], b["iters"])
        assert np.abs(a["V"] - b["V"]).max() <= 1e-9 * np.abs(a["V"]).max()
        # owner-computes (round 4): the barrier blocks are added to the rows a rank holds, nothing of the matrix is exchanged -- per Newton iteration
        # the time stepper moves nodal vectors and scalars only
        its = int(b["iters"].sum())
        assert float(b["stepper_bytes"]) / max(its, 1) <= 0.5 * 8 * float(b["nnz"]), (float(b["stepper_bytes"]) / its, 8 * float(b["nnz"]))


@pytest.mark.parametrize("world", [2, 4])
def test_owner_computes_rows_no_matrix_value_crosses_ranks(world):
    """Round 4, SURVEY.md 8e as the north star states it: with the assembly and the solver sharded, a rank assembles exactly the CSR rows its
    fronts read (its subtrees' nodes + the separator rows above the cut) and NO matrix value crosses ranks.  Same Newton counts and positions
    as the single-rank run; what the time stepper all-reduces per Newton iteration is a few nodal vectors and scalars -- far less than one
    copy of the CSR values, which the older scheme summed every iteration -- and every rank assembles only part of the rows."""
    with tempfile.TemporaryDirectory() as d:
        one, many = os.path.join(d, "one.npz"), os.path.join(d, "many.npz")
        run(1, one, "owner")
        run(world, many, "owner")
        a, b = np.load(one), np.load(many)
        assert np.array_equal(a["iters"], b["iters"])
        assert np.abs(a["V"] - b["V"]).max() <= 1e-11 * np.abs(a["V"]).max()
        assert abs(a["E"] - b["E"]) <= 1e-11 * abs(a["E"])
        its = int(b["iters"].sum())
        per_iter = float(b["stepper_bytes"]) / max(its, 1)
        nodal = 3 * 8 * float(b["nodes"])
        assert per_iter <= 6 * nodal + 4096, (per_iter, nodal)  # gradient (+ the line search's scalars): a handful of nodal vectors, never the matrix
        assert per_iter <= 0.25 * 8 * float(b["nnz"]), (per_iter, 8 * float(b["nnz"]))
        assert 0 < int(b["rows_nodes"]) < int(b["nodes"])  # rank 0 holds its subtrees' rows and the shared separator rows, not all of them
        assert int(a["stepper_bytes"]) == 0 and int(a["solver_bytes"]) == 0
        # round 5: the solver sends point to point -- every byte sent is received exactly once, and what crosses per Newton iteration (update matrices of
        # the children another rank computed, their update vectors, the separators' solution entries, + the all-reduced solution vector and pivot flag)
        # stays below one copy of the CSR values on every rank
        sent, recv = b["solver_sent_all"].astype(float), b["solver_received_all"].astype(float)
        assert sent.sum() > 0 and sent.sum() == recv.sum(), (sent, recv)
        assert (b["solver_bytes_all"].astype(float) / max(its, 1)).max() <= 8 * float(b["nnz"]), (b["solver_bytes_all"] / max(its, 1), 8 * float(b["nnz"]))


def test_contact_entry_points_return_whole_sums_on_a_sharded_context():
    """ipcgpu_contact_gradient_add / ipcgpu_contact_hessian_add called directly on a context that is sharded over two ranks return the whole
    barrier gradient and all Hessian blocks -- which stencils a rank evaluates is the time stepper's owner-computes plan, never an index range
    hidden in the handler (round 3 returned a rank's share there without saying so)."""
    with tempfile.TemporaryDirectory() as d:
        one, two = os.path.join(d, "one.npz"), os.path.join(d, "two.npz")
        run(1, one, "capi_contact")
        run(2, two, "capi_contact")
        a, b = np.load(one), np.load(two)
        assert int(a["nActive0"]) > 0 and int(a["nActive0"]) == int(b["nActive0"])
        assert np.abs(a["cg"] - b["cg"]).max() <= 1e-9 * np.abs(a["cg"]).max()
        assert np.abs(a["ca"] - b["ca"]).max() <= 1e-9 * np.abs(a["ca"]).max()


@pytest.mark.gpu
def test_rccl_binding_from_c_on_one_rank(gpu_lib):
    """include/adapters/ipcgpu_rccl.cpp: ncclCommInitRank on the context's device and the stream-ordered all-reduce hook, driven from C
    (libipcgpu_rccl.so).  A one-GPU box can only form a communicator of one rank (RCCL refuses two ranks on one device); what is
    checked is the binding itself -- unique id, attach, an all-reduce (sum and min) enqueued on the context's stream, detach."""
    c = gpu_lib.Context(0)
    uid = gpu_lib.Context.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    c.rccl_attach(0, 1, uid)
    assert c.rccl_selftest(0, 4096, 0) == 1.0 and c.rccl_selftest(0, 4096, 1) == 1.0
    assert c.rccl_selftest_p2p(0, 1, 4096) == 1.0  # (world 1: the hook is installed, the shift is a copy)
    c.rccl_detach()
    c.close()


RCCL_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["IPC_REPO"])
import ipc_amd
from ipc_amd import scene

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)  # bootstrap only: carries the 128 bytes of the unique id
c = ipc_amd.Context(rank)
uid = [ipc_amd.Context.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, 0)
c.rccl_attach(rank, world, uid[0])
out = dict(sum=c.rccl_selftest(rank, 1 << 16, 0), min=c.rccl_selftest(rank, 1 << 16, 1), shift=c.rccl_selftest_p2p(rank, world, 1 << 16))
c.set_shard(rank, world)
c.set_solver_shard(rank, world)
V, F = scene.make_mat(40)
left, right = scene.border_verts(V, 0.01)
c.set_mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
c.set_positions(scene.twist_state(scene.jitter(V, F, rel=2e-2), 0.15))
c.opt_init(0.025, False)
c.set_twist(left, right)
c.precompute()
iters = [c.solve_timestep(40) for _ in range(2)]
s = c.state()
xs, cm = c.solver_exchange_stats(), c.comm_stats()
allx = [None] * world
dist.all_gather_object(allx, (xs["sent_bytes"], xs["received_bytes"]))
if rank == 0:
    np.savez(os.environ["OUT"], V=s["V"], E=s["E"], iters=np.array(iters), sent=np.array([a[0] for a in allx]), received=np.array([a[1] for a in allx]),
             shared=c.solver_shard_stats()["shared_flop_fraction"], **{k: np.array(v) for k, v in out.items()})
c.rccl_detach()
c.close()
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_rccl_over_two_or_more_gpus():
    """The real thing, whenever the box has more than one GPU (skipped on a one-GPU box): one process per GPU, RCCL bound to the contexts from C
    (ipcgpu_rccl_attach: all-reduce AND point-to-point hooks on the context's own stream), owner-computes assembly + the subtree-sharded solver whose
    update matrices travel as ncclSend / ncclRecv groups over xGMI.  Same Newton counts and positions as the single-rank run on GPU 0."""
    import ctypes
    # (not through torch: importing it into THIS process after libipcgpu.so brings its bundled rocBLAS / rocSPARSE in beside the system ones the library
    # is linked to, and the rocSOLVER comparison back end of a later test then fails inside its analysis -- seen when this file ran in front of test_gpu_parity.py)
    cnt = ctypes.c_int(0)
    ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(cnt))
    ngpu = cnt.value
    if ngpu < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device (the multi-rank paths run over gloo on one device in the tests above)")
    world = 4 if ngpu >= 4 else 2
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        one, many = os.path.join(d, "one.npz"), os.path.join(d, "many.npz")
        run(1, one, "owner")  # the single-rank reference (gloo worker with world 1: no hooks)
        with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
            f.write(RCCL_WORKER)
            script = f.name
        env = dict(os.environ, IPC_REPO=repo, OUT=many, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(world)]
        logs = [p.communicate(timeout=600)[0] for p in procs]
        os.unlink(script)
        for p, log in zip(procs, logs):
            assert p.returncode == 0, log[-3000:]
        a, b = np.load(one), np.load(many)
        assert float(b["sum"]) == world * (world + 1) / 2 and float(b["min"]) == 1.0 and float(b["shift"]) == world  # rank 0 hears from rank world - 1
        assert np.array_equal(a["iters"], b["iters"])
        assert np.abs(a["V"] - b["V"]).max() <= 1e-11 * np.abs(a["V"]).max()
        assert b["sent"].sum() == b["received"].sum() > 0


@pytest.mark.gpu
def test_bench_line_at_two_ranks_on_one_device():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per "GPU"), with --single-device-test so that a one-GPU box can run it:
    both ranks on cuda:0, the hooks over gloo.  What this pins is everything of the N > 1 bench path except the wire itself: the launch contract (RANK /
    LOCAL_RANK / WORLD_SIZE from the environment), attach_transport, the sharded set-up, the barrier-bracketed timing with the maximum over ranks, ONE JSON line
    from rank 0 with the contract's keys, the communication record, and the second (>= 1 M-tet sized in the real run, small here) workload."""
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--single-device-test", "--size", "60", "--large-size", "80"]
    r = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    assert "gloo" in d["transport"]
    c = d["comm_per_iter"]
    assert c["solver_p2p_sent_bytes_rank0"] + c["solver_p2p_received_bytes_rank0"] > 0  # the solver really was sharded: update matrices crossed ranks
    assert 0.3 < c["rows_assembled_on_rank0"] < 0.8  # owner-computes rows: about half of the matrix each
    lw = d["large_workload"]
    assert lw["value"] > 0 and lw["solver_sharded"] and 0.0 < lw["shared_flop_fraction"] < 1.0
    # round 6: the strong-scaling model of DESIGN.md section 6 evaluated beside the measured value, for both workloads, and the time a rank waits above the cut
    for rec in (d, lw):
        mdl = rec["expected_speedup_model"]
        assert 0 < mdl["steps_above_cut"] < mdl["steps_on_critical_path"] and 0.4 < mdl["largest_rank_share_below_cut"] <= 1.0
        assert 1.0 < mdl["expected_speedup"] < 2.0 and mdl["amdahl_bound_on_factorisation_flops"] <= 2.0
        assert mdl["single_rank_measured_in_this_job"]["factor_ms"] > 0
    assert c["rank_wait_ms"] >= 0.0 and lw["comm_per_iter_rank0"]["rank_wait_ms"] >= 0.0
    assert len(d["ms_per_step_min_median_max"]) == 3 and d["ms_per_step_min_median_max"][0] <= d["ms_per_step_min_median_max"][2]
