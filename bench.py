#!/usr/bin/env python
"""bench.py -- Newton iterations/sec of the IPC time-step hot path on MI355X.

Workload (BASELINE.json configs[1]): matTwist -- a 150 x 150 x 2-node neo-Hookean sheet
(45 000 nodes, 133 206 tets, 135 000 dofs; stand-in for the missing mat150x150t40.msh,
SURVEY.md 8d), E = 2e4, nu = 0.4, rho = 1000, dt = 0.04, no gravity, both x-extreme node
columns scripted to twist at +-0.4 pi rad/s (input/paperExamples/14_matTwist.txt,
AnimScripter.cpp:555-572), self-contact off: element assembly + sparse Cholesky + line search.
One "step" = one pass of the solveSub_IP loop (Optimizer.cpp:1829-2204) = one Newton iteration,
time-step boundaries (scripted DBC motion, BE velocity update) included as they occur.

  python bench.py --gpus N --steps K --warmup W
(N > 1: launched by torch.distributed.run, one rank per GPU over RCCL: the direct solver sharded by subtrees, elements and contact-pair lists by the
CSR rows those subtrees read -- no matrix value crosses ranks; the line's `comm_per_iter` says what does.  See DESIGN.md section 6.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_scene(n):
    from ipc_amd import scene
    V, F = scene.make_mat(n)
    left, right = scene.border_verts(V, 0.01)
    return V, F, left, right


class DevPtr:
    """__cuda_array_interface__ view of a raw device pointer (float64 vector)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}


def torch_hooks(local_rank, host_bounce):
    """The two host-level hooks of the C ABI (ipcgpu_opt_set_allreduce / ipcgpu_opt_set_exchange) on torch.distributed.  The library drains its stream before it
    calls them and expects the data in place when they return.  host_bounce = True: gloo through host tensors (--single-device-test, plumbing only);
    False: the process group's RCCL communicator on the device buffers themselves (the fall-back transport of attach_transport)."""
    import torch
    import torch.distributed as dist
    dev = f"cuda:{local_rank}"

    def allreduce(ptr, count, op):
        t = torch.as_tensor(DevPtr(ptr, count), device=dev)
        red = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN
        if host_bounce:
            h = t.cpu()
            dist.all_reduce(h, op=red)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=red)
        torch.cuda.synchronize()
        return 0

    def exchange(ops):
        reqs, recvs = [], []
        for ptr, count, peer, send in ops:
            t = torch.as_tensor(DevPtr(ptr, count), device=dev)
            if send:
                reqs.append(dist.P2POp(dist.isend, t.cpu() if host_bounce else t, peer))
            elif host_bounce:
                h = torch.empty(count, dtype=torch.float64)
                recvs.append((t, h))
                reqs.append(dist.P2POp(dist.irecv, h, peer))
            else:
                reqs.append(dist.P2POp(dist.irecv, t, peer))
        for r in dist.batch_isend_irecv(reqs):
            r.wait()
        for t, h in recvs:
            t.copy_(h)
        torch.cuda.synchronize()
        return 0
    return allreduce, exchange


def attach_transport(ctx, args, rank, world, local_rank):
    """What carries the bytes between the ranks; returns its name (printed in the JSON line as `transport`).
      * default: RCCL called from C on the context's own stream (include/ipcgpu_rccl.h, include/adapters/ipcgpu_rccl.cpp): rank 0 draws the unique id,
        torch.distributed is only the bootstrap that carries its 128 bytes; no collective of the data path goes through Python after this.  A ring of
        ncclSend / ncclRecv (ipcgpu_rccl_selftest_p2p) is pushed through the new communicator before the solver relies on it;
      * if that fails on ANY rank (agreed by an all-reduce of a flag, so that all ranks take the same branch) and --transport is `auto`: the process group's own
        RCCL communicator through the host-level hooks -- the same bytes, with a stream drain around every exchange;
      * --single-device-test: gloo through a host bounce (plumbing check on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    import ipc_amd
    if args.single_device_test:
        ar, xc = torch_hooks(local_rank, host_bounce=True)
        ctx.set_allreduce(ar)
        ctx.set_exchange(xc)
        return "gloo through a host bounce (plumbing test)"
    dev = f"cuda:{local_rank}"
    if args.transport in ("auto", "rccl"):
        ok, why = 1, ""
        # Preconditions are agreed BEFORE any collective of the binding: ncclCommInitRank inside rccl_attach and the ring self-test are collectives themselves, so
        # a rank that cannot even load the binding must say so while the others can still hear it (an all-reduce of the process group, which is up).  What the
        # `auto` fall-back covers are failures every rank sees or that are known up front; a rank that dies INSIDE ncclCommInitRank leaves the others waiting in
        # RCCL until the launcher's own timeout ends the job (torch.distributed.run tears the group down when one rank exits).
        try:
            ipc_amd.Context.rccl_unique_id()
        except Exception as e:  # noqa: BLE001
            ok, why = 0, repr(e)
        pre = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(pre, op=dist.ReduceOp.MIN)
        if int(pre.item()) == 0:
            if why:
                print(f"[bench] rank {rank}: RCCL binding not loadable: {why}", file=sys.stderr, flush=True)
            if args.transport == "rccl":
                raise SystemExit("--transport rccl: the RCCL binding cannot be loaded on at least one rank")
            ar, xc = torch_hooks(local_rank, host_bounce=False)
            ctx.set_allreduce(ar)
            ctx.set_exchange(xc)
            return "torch.distributed nccl backend through the host-level hooks (stream drained around every exchange)"
        try:
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(ipc_amd.Context.rccl_unique_id()), dtype=torch.uint8))
        except Exception as e:  # noqa: BLE001 -- the other ranks are waiting in the broadcast: take part in it, report afterwards
            ok, why = 0, repr(e)
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        dist.broadcast(idt, 0)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            try:
                ctx.rccl_attach(rank, world, bytes(idt.cpu().numpy().tobytes()))
                got = ctx.rccl_selftest_p2p(rank, world)
                if got != float((rank + world - 1) % world + 1):
                    raise RuntimeError(f"ring self-test: received {got} from rank {(rank + world - 1) % world}")
            except Exception as e:  # noqa: BLE001
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return "RCCL from C on the context's stream (ncclAllReduce; ncclSend / ncclRecv groups)"
        if why:
            print(f"[bench] rank {rank}: RCCL binding not usable: {why}", file=sys.stderr, flush=True)
        if args.transport == "rccl":
            raise SystemExit("--transport rccl: the RCCL binding failed on at least one rank")
        ctx.rccl_detach()
    ar, xc = torch_hooks(local_rank, host_bounce=False)
    ctx.set_allreduce(ar)
    ctx.set_exchange(xc)
    return "torch.distributed nccl backend through the host-level hooks (stream drained around every exchange)"


# (rounds 2-3 switched the sharded assembly on only from 4 M tets: its partial matrices were summed by an all-reduce of the CSR values.  Round 4: with the solver
# sharded as well the assembly is owner-computes -- a rank assembles the rows its fronts read, no matrix value crosses ranks -- so every size shards.)


def pmc_traffic(size):
    """HBM bytes per launch of the assembly kernel as measured with the PMC counters (same workload), or None."""
    here = os.path.dirname(os.path.abspath(__file__))
    path = None
    for rnd in ("r06", "r05", "r04", "r03"):  # the newest measurement of this kernel (the patch size changed in round 4; round 5 left the kernel alone and measured again)
        name = f"{rnd}_pmc_assembly_traffic.json" if size == 150 else f"{rnd}_pmc_assembly_traffic_mat{size}.json"
        if os.path.exists(os.path.join(here, "profiles", name)):
            path = os.path.join(here, "profiles", name)
            break
    if path is None:
        return None
    with open(path) as f:
        return float(json.load(f)["traffic_bytes"])


_T0 = time.time()


def progress(msg):
    """phase marks on stderr (the JSON line on stdout stays alone): what a run that is cut short was doing"""
    print(f"[bench {time.time() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=150, help="mat N (N x N x 2 nodes); 150 = BASELINE config[1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-contact", action="store_true", help="skip the contact sub-records (2 x mat100 stack with self-collision; BASELINE's contact "
                    "configurations as shipped: 4_rodsTwist, 12_sphereOnMat; N = 1 only)")
    ap.add_argument("--no-large", action="store_true", help="skip the 1.12 M-tet sub-record `roofline_large` (mat433, N = 1 only)")
    ap.add_argument("--large-size", type=int, default=433, help="N > 1: mat size of the second, >= 1 M-tet workload reported under 'large_workload' (0 = off)")
    ap.add_argument("--cpu-iters", type=int, default=40)
    ap.add_argument("--solver", type=int, default=0, help="0 = GPU multifrontal, 1 = rocSOLVER csrrf")
    ap.add_argument("--solver-shard", choices=["on", "off"], default="on",
                    help="N > 1: subtree-sharded factorisation and solves (ipcgpu_linsys_set_shard); off = every rank repeats the whole solve")
    ap.add_argument("--single-device-test", action="store_true",
                    help="plumbing check on a one-GPU box: all ranks on cuda:0, collectives over gloo through a host bounce (numbers are meaningless)")
    ap.add_argument("--transport", choices=["auto", "rccl", "torch"], default="auto",
                    help="N > 1: rccl = RCCL called from C on the context's stream (include/ipcgpu_rccl.h); torch = the process group's communicator through the "
                         "host-level hooks; auto = rccl, torch if the binding fails on any rank (attach_transport)")
    ap.add_argument("--shard", choices=["auto", "on", "off"], default="auto",
                    help="N > 1: shard the element assembly over the ranks (all-reduce of gradient + CSR values per iteration); "
                         "auto = only when the mesh is big enough for that to pay (see DESIGN.md section 6)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.single_device_test else "nccl", rank=rank, world_size=world)

    import ipc_amd

    V, F, left, right = build_scene(args.size)
    ctx = ipc_amd.Context(local_rank, solver=args.solver)
    # Elements and contact-pair lists shard with the solver's subtrees (owner-computes rows, ipcgpu_opt_comm_stats): what crosses ranks per Newton iteration is
    # the nodal gradient, scalars, and the solver's update matrices / vectors above its cut -- no CSR value.  --shard off / --solver-shard off are A/B switches.
    solver_sharded = distributed and args.solver_shard == "on"
    sharded = distributed and (args.shard == "on" or (args.shard == "auto" and solver_sharded))
    transport = None
    if distributed:
        if sharded:
            ctx.set_shard(rank, world)  # elements and contact pairs by node ownership: a rank assembles the rows its fronts read
        transport = attach_transport(ctx, args, rank, world, local_rank)
        if solver_sharded:
            # the direct solver is what an iteration consists of: the assembly tree is cut below its top separators, every rank
            # factorises / solves its own subtrees, a front above the cut is executed by ONE rank and fed point to point
            ctx.set_solver_shard(rank, world)
    ctx.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    ctx.opt_init(dt=0.04, gravity=False)
    ctx.set_twist(left, right, 0.4 * np.pi)
    t0 = time.time()
    ctx.precompute()
    t_pre = time.time() - t0
    n_rows, nnz = ctx.get_dims()

    state = {"in_step": False, "steps_done": 0}

    def one_iteration():
        # exactly one pass of the solveSub_IP loop; converged passes roll over into the next time step
        while True:
            if not state["in_step"]:
                ctx.begin_timestep()
                state["in_step"] = True
            if ctx.newton_iter():
                ctx.end_timestep()
                state["in_step"] = False
                state["steps_done"] += 1
                continue
            return

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    progress(f"precompute done ({t_pre:.1f} s); {args.warmup} warm-up + {args.steps} timed Newton iterations")
    for _ in range(args.warmup):
        one_iteration()
    comm0 = ctx.comm_stats()
    xs0 = ctx.solver_exchange_stats()
    t_before = ctx.timers().copy()
    barrier()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        one_iteration()
        stamps.append(time.perf_counter())  # (host time at the iteration's one synchronisation; the assembly enqueued ahead belongs to the next one)
    barrier()
    elapsed = time.perf_counter() - t0
    per_iter = 1e3 * np.diff(np.array(stamps))
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    timers = ctx.timers() - t_before
    comm1 = ctx.comm_stats()
    xs1 = ctx.solver_exchange_stats()

    progress(f"timed region done: {args.steps / elapsed:.1f} it/s")
    # collective when the solver is sharded: every rank takes part
    f_ms, s_ms = ctx.bench_factor_solve(3)
    model = None
    if distributed and solver_sharded:
        cp = ctx.solver_critical_path()
        wait_ms = (xs1["wait_ms"] - xs0["wait_ms"]) / args.steps
        one_rank = single_rank_reference(args, ipc_amd, local_rank, args.size, max(20, args.steps // 4), 5)  # every rank, on its own GPU: no communication inside
        model = expected_speedup_model(one_rank, cp, ctx.solver_shard_stats()["shared_flop_fraction"], (xs1["received_bytes"] - xs0["received_bytes"]) / args.steps, world)
        model["rank_wait_ms_per_iter_rank0"] = wait_ms
    out = None
    if rank == 0:
        K = args.steps
        split = {  # main.cpp:1326-1340 bucket map
            "assembly_ms": 1e3 * (timers[0] + timers[1] + timers[12]) / K,
            "solve_ms": 1e3 * (timers[2] + timers[3] + timers[4]) / K,
            "ccd_linesearch_ms": 1e3 * (timers[13] + timers[14] + timers[5] + timers[9]) / K,
            "factor_ms": 1e3 * timers[3] / K,
            "backsolve_ms": 1e3 * timers[4] / K,
            "timestep_ms": 1e3 * timers[11] / K,
        }
        # dominant HBM-bound kernel: fused element assembly (gradient + projected Hessian -> CSR)
        ms_asm, bytes_asm = ctx.bench_assembly(0.04 ** 2, reps=20)
        ach = bytes_asm / (ms_asm * 1e-3) / 1e9
        stream_gbs = ctx.bench_stream(1 << 30, 10)
        st = ctx.linsys_stats()
        out = {
            "metric": "newton_iterations_per_sec",
            "value": K / elapsed,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K,
            # dispersion over the K timed iterations of THIS run (an iteration that ends a time step carries the step boundary; box-to-box spread is ~3 %)
            "ms_per_step_min_median_max": [float(per_iter.min()), float(np.median(per_iter)), float(per_iter.max())],
            "higher_is_better": True,
            # the metric's workload is fixed (strong scaling); with both shards switched off every rank runs the whole iteration and
            # nothing is divided: say so instead of letting N replicas read as an N-GPU strong-scaling point
            "scaling": "strong" if (world == 1 or sharded or solver_sharded) else "none (replicated: every rank runs the whole iteration)",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"matTwist mat{args.size}: {V.shape[0]} nodes / {F.shape[0]} tets neo-Hookean sheet, twist DBC, "
                            "self-contact off, BE dt=0.04, E=2e4 nu=0.4 rho=1000 (BASELINE configs[1])",
                "n_nodes": int(V.shape[0]), "n_tets": int(F.shape[0]), "n_dofs": int(n_rows), "nnz_upper_csr": int(nnz),
                "linear_solver": "gpu-multifrontal-llt" if args.solver == 0 else "rocsolver-csrrf",
                "parallelism": "single GPU" if world == 1 else (
                    f"{world} GPUs: " + ("owner-computes assembly: a rank assembles the CSR rows of the nodes its subtrees eliminate + the separator rows above "
                                         "the cut (elements and contact stencils on a cut evaluated by both sides), no matrix value crosses ranks; all-reduce "
                                         "of the nodal gradient and of scalars" if sharded and solver_sharded else
                                         ("element-sharded assembly, partial matrices summed by an all-reduce of the CSR values" if sharded else
                                          "assembly repeated on every rank"))
                    + "; " + (f"subtree-sharded multifrontal factorisation and solves: {100 * ctx.solver_shard_stats()['shared_flop_fraction']:.0f} % of the "
                              "factorisation flops lie above the cut (each front there executed by one rank, the ranks below it waiting for its result); update "
                              "matrices / vectors of children on other ranks and the separators' solution entries go point to point (one send / receive group per level of the cut), "
                              "the pivot flag and the solution vector by all-reduce; what carries the bytes: see `transport`" if solver_sharded else "factorisation and solves repeated on every rank")),
                "time_steps_completed": state["steps_done"],
            },
            "transport": transport,
            # N > 1: what the partitioning can buy by the model of DESIGN.md section 6, beside the measured `value` (the driver's SCALE record adjudicates it)
            "expected_speedup_model": model,
            "comm_per_iter": {"stepper_allreduce_bytes": (comm1["stepper_bytes"] - comm0["stepper_bytes"]) / K,
                              "stepper_allreduce_calls": (comm1["stepper_calls"] - comm0["stepper_calls"]) / K,
                              "solver_bytes_rank0": (comm1["solver_bytes"] - comm0["solver_bytes"]) / K,  # sent + received point to point + all-reduced buffers, this rank
                              "solver_p2p_sent_bytes_rank0": (xs1["sent_bytes"] - xs0["sent_bytes"]) / K,
                              "solver_p2p_received_bytes_rank0": (xs1["received_bytes"] - xs0["received_bytes"]) / K,
                              "solver_collective_calls": (comm1["solver_calls"] - comm0["solver_calls"]) / K,
                              # time rank 0's stream spent inside the solver's point-to-point groups (HIP events around each): waiting for the rank that executes a front above the cut
                              "rank_wait_ms": (xs1["wait_ms"] - xs0["wait_ms"]) / K,
                              "csr_value_bytes": 8 * int(nnz), "nodal_vector_bytes": 24 * int(V.shape[0]),
                              "rows_assembled_on_rank0": comm1["rows_assembled_nodes"] / max(comm1["nodes"], 1)},
            "split_ms_per_iter": split,
            "split_note": "factor_ms holds the numeric factorisation AND both triangular sweeps (the forward sweep of a level runs on its own stream beside the "
                          "factorisation of the levels above, MfNumeric::factorizeSolve); backsolve_ms is what follows them in the same bucket: the batched "
                          "read-back of |p|_inf, the inversion step filter and E -- and the first trial of the line search (step, inversion flag, E at the trial "
                          "point), taken on the device behind the solve: ccd_linesearch_ms is what a rejected trial costs.  solver.factor_ms / solver.solve_ms below time the two separately "
                          "(ipcgpu_bench_factor_solve, HIP events)",
            "roofline": {
                "kernel": "k_assemble_patch<true> (fused NH gradient + PSD-projected Hessian + mass/DBC diagonal -> symmetric-upper CSR, atomic-free)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.size),
                "traffic_source": "NOT a live counter: read from profiles/r06_pmc_assembly_traffic*.json (r05_*, r04_*, r03_* when absent), measured on this kernel with "
                                  "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, calibrated in-run on a 1 GiB device copy "
                                  "(tools/pmc_traffic.py, tools/gpu_pmc_traffic.sh); bytes per launch",
                "algorithmic_bytes": bytes_asm, "avg_launch_ms": ms_asm,
                "measured_stream_copy_GBs": stream_gbs,
            },
            # the step is dominated by the sparse Cholesky, not by the HBM-bound assembly kernel above: its two kernel groups with
            # their own bounds.  Factorisation: flops of the symbolic analysis / HIP-event time of the whole launch sequence, against
            # the fp64 matrix-core rate (the guide lists no fp64 peak; 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 32 MAC/cycle x 2.4 GHz is the
            # vendor figure).  Triangular solves: every entry of L is read once per sweep (2 x 8 nnz(L) bytes) against HBM.
            "roofline_solver": [
                {"kernel": "multifrontal factorisation (k_front_fused, k_big_step, k_big_schur, k_extend_add, k_xinv_gemm)",
                 "bound": "mfma", "achieved": st["flops"] / 1e12 / (f_ms * 1e-3), "peak": 78.6, "unit": "TFLOP/s",
                 "frac": st["flops"] / 1e12 / (f_ms * 1e-3) / 78.6, "traffic": None,
                 "note": "latency-bound: ~90 dependent 32-column pivot steps on the critical path of the assembly tree"},
                {"kernel": "triangular solves (k_fwd_level, k_bwd_level, k_big_fwd_rect, k_big_bwd_init, k_xinv_fwd, k_xinv_bwd)",
                 "bound": "hbm", "achieved": 2 * 8 * st["nnzL"] / 1e9 / (s_ms * 1e-3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": 2 * 8 * st["nnzL"] / 1e9 / (s_ms * 1e-3) / HBM_PEAK_GBS, "traffic": None,
                 "note": "launch-bound: two to four dependent launches per level of the assembly tree, 15 levels, both directions"},
            ],
            "solver": {"nnzL": st["nnzL"], "factor_gflop": st["flops"] / 1e9, "fronts": st["fronts"], "levels": st["levels"],
                       "factor_ms": f_ms, "solve_ms": s_ms, "factor_gflops_per_s": st["flops"] / 1e9 / (f_ms * 1e-3),
                       "precompute_s": t_pre},
        }
    ctx.close()
    if rank == 0 and world == 1 and not args.no_contact:
        # the contact half of the path, timed by the same process: two stacked mat100 sheets with self-collision on (barrier terms,
        # constraint sets, CCD, pattern changes); BASELINE's metric is quoted on the contact-free matTwist above, this is a sub-record
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_contact
        progress("contact sub-record (2 x mat100 stack)")
        r = bench_contact.run(n=100, layers=2, steps=12, max_iter=12)
        out["contact"] = {"workload": r["scene"] + f": {r['n_nodes']} nodes / {r['n_tets']} tets, {r['n_surface_tris']} surface triangles, dt 0.01, 12 time steps",
                          "newton_iterations": r["newton_iterations"], "value": r["iters_per_s"], "unit": "iter/s", "ms_per_iter": r["ms_per_iter_wall"],
                          "split_ms_per_iter": r["split_ms_per_iter"],
                          "active_constraints_per_step": [c["nActive"] for c in r["contact_state_per_step"]],
                          "pattern_changes": r["contact_state_per_step"][-1]["nPatternChanges"], "intersected_at_end": r["intersected_at_end"]}
    if rank == 0 and world == 1 and not args.no_contact:
        # BASELINE configs[1]'s scene AS SHIPPED: 14_matTwist.txt:15 says `selfCollisionOn` (the headline above follows BASELINE and strips it).  Two windows: the first
        # steps (contact machinery running, nothing active) and the steps after the sheet has wrapped onto itself (tools/bench_mat_twist.py)
        try:
            import bench_mat_twist
            progress("mat_twist_as_shipped (mat150, selfCollisionOn)")
            out["mat_twist_as_shipped"] = bench_mat_twist.run(args.size)
        except Exception as e:  # noqa: BLE001
            out["mat_twist_as_shipped"] = {"value": None, "note": f"not measured: {e!r}"[:300]}
    if rank == 0 and world == 1 and not args.no_contact and not args.no_large:
        # BASELINE configs[4] scale ("~1M tets: full pipeline") on ONE GPU, SURVEY 8d item 5's stand-in: three mat250 sheets stacked with gaps < sqrt(dHat) --
        # 1.12 M tets, 1.36 M active constraints + 0.25 M mollified pairs at the first step.  Same split as `contact`.
        try:
            progress("contact_large (3 x mat250 stack, 1.12 M tets)")
            r = bench_contact.run(n=250, layers=3, steps=2, max_iter=4)
            out["contact_large"] = {"workload": r["scene"] + f": {r['n_nodes']} nodes / {r['n_tets']} tets, {r['n_surface_tris']} surface triangles, dt 0.01, 2 time steps of at most 4 iterations",
                                    "newton_iterations": r["newton_iterations"], "value": r["iters_per_s"], "unit": "iter/s", "ms_per_iter": r["ms_per_iter_wall"],
                                    "split_ms_per_iter": r["split_ms_per_iter"], "precompute_s": r["precompute_s"],
                                    "active_constraints_per_step": [c["nActive"] for c in r["contact_state_per_step"]],
                                    "mollified_pairs_per_step": [c["nPara"] for c in r["contact_state_per_step"]],
                                    "candidate_pairs_per_step": [c["nCand"] for c in r["contact_state_per_step"]],
                                    "pattern_changes": r["contact_state_per_step"][-1]["nPatternChanges"], "intersected_at_end": r["intersected_at_end"],
                                    "solver": r.get("solver")}
        except Exception as e:  # noqa: BLE001
            out["contact_large"] = {"value": None, "note": f"not measured: {e!r}"[:300]}
    if rank == 0 and world == 1 and not args.no_contact:
        # BASELINE configs[3] and configs[2] AS SHIPPED, from the fixtures the reference's own main() produced (tests/golden/ref_scene_*.npz): timed, and the
        # Newton iteration count of every step compared with the reference's in the same record (tools/bench_scene.py)
        import bench_scene
        for name, steps, first, what in (
                ("rods_twist", 2, 1, "BASELINE configs[3]: input/paperExamples/4_rodsTwist.txt as shipped (4 x rod300x33.msh, self-contact on, script twist), time steps 1-2"),
                ("sphere_on_mat", 36, 29, "BASELINE configs[2]: input/paperExamples/12_sphereOnMat.txt as shipped (stiff ball on a mat being stretched, half-space, self-contact on): "
                                          "all 36 steps run, the contact steps 29-36 timed")):
            try:
                progress(f"{name} (fixture of the reference's own run)")
                out[name] = bench_scene.run(name, steps, first, what)
            except Exception as e:  # noqa: BLE001  (a sub-record must not take the bench line down)
                out[name] = {"value": None, "note": f"not measured: {e!r}"[:300]}
    if rank == 0 and world == 1 and not args.no_large and args.size != args.large_size and args.large_size:
        # the same twist scene at 1.12 M tets (mat433) on ONE GPU: where the latency-bound steps amortise, the fractions are what the kernels do when fed
        try:
            progress(f"roofline_large (mat{args.large_size})")
            out["roofline_large"] = large_single(args, ipc_amd)
        except Exception as e:  # noqa: BLE001
            out["roofline_large"] = {"value": None, "note": f"not measured: {e!r}"[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The CPU baselines come LAST: their thread pools (OpenMP workers of the port, the std::thread pool behind the reference's parallel_for) stay alive in this
        # process, and with them in the background the host side of the contact sub-record above measured 1.0 ms per iteration slower (symbolic analysis 1.11 -> 1.52,
        # step bounds 0.63 -> 1.13 ms: profiles/r05_bench_line.json against r05_entry_lists_on_the_device_ab.txt).  Nothing on the GPU runs beside them.
        progress("cpu_baseline (port, 16 host cores)")
        out["cpu_baseline"] = cpu_baseline(V, F, left, right, args.cpu_iters)
        progress("cpu_reference (the reference's sources, 16 threads)")
        out["cpu_reference"] = cpu_reference(V, F)
    if distributed and args.large_size and args.size != args.large_size:
        # a second, >= 1 M-tet strong-scaling point for the curve (mat150's 2.9 ms iteration is mostly the dependent pivot chain of
        # its top separators, which does not shard): same script, same measurement, fewer steps
        big = large_workload(args, rank, local_rank, world, torch, dist, ipc_amd)
        if rank == 0:
            out["large_workload"] = big
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        progress("done")
        print(json.dumps(out))


XGMI_LINK_GBS = 64.0  # what one xGMI link is assumed to deliver to a point-to-point transfer in the model below (MI355X_MICROARCH.md: 7 links per GPU, ~153 GB/s peak
#                       each; ring collectives and single transfers see well under half of that) -- an input of the MODEL, stated in the line, not a measurement


def single_rank_reference(args, ipc_amd, local_rank, size, steps, warmup):
    """N > 1: the same workload on ONE rank (this rank's own GPU, no communication), measured inside the same job: the denominator of the model below."""
    V, F, left, right = build_scene(size)
    ctx = ipc_amd.Context(local_rank, solver=args.solver)
    ctx.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    ctx.opt_init(dt=0.04, gravity=False)
    ctx.set_twist(left, right, 0.4 * np.pi)
    ctx.precompute()
    state = {"in": False}

    def one():
        while True:
            if not state["in"]:
                ctx.begin_timestep()
                state["in"] = True
            if ctx.newton_iter():
                ctx.end_timestep()
                state["in"] = False
                continue
            return
    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    f_ms, s_ms = ctx.bench_factor_solve(2)
    ctx.close()
    return {"ms_per_step": ms, "factor_ms": f_ms, "solve_ms": s_ms}


def expected_speedup_model(one, cp, shared_flop_fraction, p2p_received_bytes, world):
    """DESIGN.md section 6, evaluated: the factorisation is a chain of dependent 32-column pivot steps (latency-bound), so its time divides like its STEPS, not like its
    flops.  The steps of the fronts above the cut run on one rank while the others wait; the rest divides by the ranks (by the largest rank's share of the flops
    below the cut); the update matrices received point to point are added at the assumed link rate.  Everything outside the factorisation is taken as it is on one rank
    (assembly, sweeps, line search: latency-bound at these sizes), which makes the figure an upper bound on what the partitioning can buy.
        factor_N = factor_1 * (a + (1 - a) * share) + bytes / link,   a = steps above the cut / steps,   iteration_N = iteration_1 - factor_1 + factor_N"""
    a = cp["steps_above_cut"] / max(cp["steps"], 1.0)
    factor_n = one["factor_ms"] * (a + (1.0 - a) * cp["max_rank_share_below"]) + 1e3 * p2p_received_bytes / (XGMI_LINK_GBS * 1e9)
    iter_n = one["ms_per_step"] - one["factor_ms"] + factor_n
    amdahl = 1.0 / (shared_flop_fraction + (1.0 - shared_flop_fraction) / world) if shared_flop_fraction is not None else None
    return {"formula": "factor_N = factor_1 * (a + (1 - a) * share) + p2p_bytes / link; iteration_N = iteration_1 - factor_1 + factor_N; speedup = iteration_1 / iteration_N",
            "single_rank_measured_in_this_job": one, "steps_on_critical_path": cp["steps"], "steps_above_cut": cp["steps_above_cut"], "a": a,
            "levels": cp["levels"], "levels_above_cut": cp["levels_above_cut"], "largest_rank_share_below_cut": cp["max_rank_share_below"],
            "p2p_received_bytes_per_iter_rank0": p2p_received_bytes, "assumed_link_GBs": XGMI_LINK_GBS,
            "model_factor_ms": factor_n, "model_ms_per_step": iter_n, "expected_speedup": one["ms_per_step"] / iter_n,
            "amdahl_bound_on_factorisation_flops": amdahl,
            "note": "the >= 6 x at 8 GPUs of the north star is excluded for an exact direct solver at these sizes: the dependent pivot chain of the separators above an "
                    "8-way cut (a of the steps, 28 % of the flops at mat150) runs on one rank whatever the links do -- compare `value` / the N = 1 line with expected_speedup"}


def large_single(args, ipc_amd, steps=12, warmup=3):
    """N = 1: the twist scene at --large-size (mat433 = 1.12 M tets): Newton iterations per second and the three roofline fractions at a size where the
    kernels are fed (the assembly kernel live with HIP events, factorisation and sweeps through ipcgpu_bench_factor_solve)."""
    V, F, left, right = build_scene(args.large_size)
    ctx = ipc_amd.Context(0, solver=args.solver)
    ctx.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    ctx.opt_init(dt=0.04, gravity=False)
    ctx.set_twist(left, right, 0.4 * np.pi)
    t0 = time.time()
    ctx.precompute()
    t_pre = time.time() - t0
    state = {"in": False}

    def one():
        while True:
            if not state["in"]:
                ctx.begin_timestep()
                state["in"] = True
            if ctx.newton_iter():
                ctx.end_timestep()
                state["in"] = False
                continue
            return
    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    el = time.perf_counter() - t0
    f_ms, s_ms = ctx.bench_factor_solve(2)
    ms_asm, bytes_asm = ctx.bench_assembly(0.04 ** 2, reps=10)
    st = ctx.linsys_stats()
    ctx.close()
    ach = bytes_asm / (ms_asm * 1e-3) / 1e9
    return {"workload": f"matTwist mat{args.large_size}: {V.shape[0]} nodes / {F.shape[0]} tets, one GPU", "value": steps / el, "unit": "iter/s", "ms_per_step": 1e3 * el / steps,
            "steps": steps, "warmup": warmup, "precompute_s": t_pre,
            "roofline": {"kernel": "k_assemble_patch<true>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes": bytes_asm, "avg_launch_ms": ms_asm, "traffic": pmc_traffic(args.large_size)},
            "roofline_solver": [
                {"kernel": "multifrontal factorisation", "bound": "mfma", "achieved": st["flops"] / 1e12 / (f_ms * 1e-3), "peak": 78.6, "unit": "TFLOP/s",
                 "frac": st["flops"] / 1e12 / (f_ms * 1e-3) / 78.6, "factor_ms": f_ms, "factor_gflop": st["flops"] / 1e9},
                {"kernel": "triangular solves", "bound": "hbm", "achieved": 2 * 8 * st["nnzL"] / 1e9 / (s_ms * 1e-3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": 2 * 8 * st["nnzL"] / 1e9 / (s_ms * 1e-3) / HBM_PEAK_GBS, "solve_ms": s_ms}]}


def large_workload(args, rank, local_rank, world, torch, dist, ipc_amd):
    """bench.py --gpus N: the same twist scene at --large-size (mat433 = 1.12 M tets), sharded the same way, timed like the main line."""
    V, F, left, right = build_scene(args.large_size)
    ctx = ipc_amd.Context(local_rank, solver=args.solver)
    sharded = args.shard == "on" or (args.shard == "auto" and args.solver_shard == "on")
    if sharded:
        ctx.set_shard(rank, world)
    attach_transport(ctx, args, rank, world, local_rank)
    if args.solver_shard == "on":
        ctx.set_solver_shard(rank, world)
    ctx.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    ctx.opt_init(dt=0.04, gravity=False)
    ctx.set_twist(left, right, 0.4 * np.pi)
    ctx.precompute()
    state = {"in": False}

    def one():
        while True:
            if not state["in"]:
                ctx.begin_timestep()
                state["in"] = True
            if ctx.newton_iter():
                ctx.end_timestep()
                state["in"] = False
                continue
            return
    K, W = max(4, args.steps // 5), 2
    for _ in range(W):
        one()
    xs0, cm0 = ctx.solver_exchange_stats(), ctx.comm_stats()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        one()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{local_rank}")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    rec = {"workload": f"matTwist mat{args.large_size}: {V.shape[0]} nodes / {F.shape[0]} tets", "steps": K, "warmup": W, "value": K / float(el.item()),
           "unit": "iter/s", "ms_per_step": 1e3 * float(el.item()) / K, "element_assembly_sharded": bool(sharded),
           "solver_sharded": args.solver_shard == "on",
           # the share of the factorisation flops above the cut: executed once each (round 5), by one rank while the ranks below it wait -- the serial part
           # of the strong-scaling model of DESIGN.md section 6.  This >= 1 M-tet workload is the one on which the subtrees dominate; on the headline
           # size (45 K nodes) the dependent pivot chain of the top separators caps strong scaling near 1.5-2 x whatever the number of GPUs.
           "shared_flop_fraction": ctx.solver_shard_stats()["shared_flop_fraction"] if args.solver_shard == "on" else None}
    xs1, cm1 = ctx.solver_exchange_stats(), ctx.comm_stats()
    rec["comm_per_iter_rank0"] = {"solver_p2p_sent_bytes": (xs1["sent_bytes"] - xs0["sent_bytes"]) / K, "solver_p2p_received_bytes": (xs1["received_bytes"] - xs0["received_bytes"]) / K,
                                  "solver_bytes": (cm1["solver_bytes"] - cm0["solver_bytes"]) / K, "stepper_allreduce_bytes": (cm1["stepper_bytes"] - cm0["stepper_bytes"]) / K,
                                  "rank_wait_ms": (xs1["wait_ms"] - xs0["wait_ms"]) / K}
    cp = ctx.solver_critical_path() if args.solver_shard == "on" else None
    ctx.close()
    if cp is not None:
        one_rank = single_rank_reference(args, ipc_amd, local_rank, args.large_size, K, W)
        rec["expected_speedup_model"] = expected_speedup_model(one_rank, cp, rec["shared_flop_fraction"], rec["comm_per_iter_rank0"]["solver_p2p_received_bytes"], world)
    return rec


def cpu_baseline(V, F, left, right, iters):
    """The oracle (a from-scratch restatement of the reference's CPU algorithm: OpenMP over the loops the
    reference hands to TBB, own multifrontal Cholesky in place of CHOLMOD) timed on this box's host cores."""
    from oracle import orc
    cores = min(os.cpu_count() or 1, 16)  # the reference's own batch scripts ran 8-12 threads (batch.py:31-46)
    m = orc.Mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    o = orc.Optimizer(m, dt=0.04, gravity=False, nthreads=cores)
    o.set_twist(left, right, 0.4 * np.pi)
    o.precompute()
    o.begin_timestep()
    t0 = time.perf_counter()
    done = 0
    for _ in range(iters):
        if o.newton_iter():
            o.end_timestep()
            o.begin_timestep()
            continue
        done += 1
    el = time.perf_counter() - t0
    t = o.timers()
    return {"value": done / el, "unit": "iter/s", "cores": cores, "kind": "port",
            "sample": f"first {done} Newton iterations of time step 1 of the same mat scene ({el:.1f} s of CPU work)",
            "split_ms_per_iter": {"assembly_ms": 1e3 * (t[0] + t[1] + t[12]) / max(done, 1),
                                  "solve_ms": 1e3 * (t[2] + t[3] + t[4]) / max(done, 1),
                                  "ccd_linesearch_ms": 1e3 * (t[13] + t[14] + t[5] + t[9]) / max(done, 1)}}


def cpu_reference(V, F, steps=4):
    """The reference's OWN code on the same scene, beside the port above: oracle/_ref/libipcref.so is ipc-sim/IPC's main.cpp / Optimizer.cpp /
    Energy / Mesh compiled where they lie by oracle/Makefile.ref (built in the container that holds the reference; the .so travels with the
    repository snapshot), run through its own main() on ALL host cores: its loops are the ones it hands to tbb::parallel_for, executed by the
    std::thread pool of oracle/refshim/tbb/parallel_for.h (IPCREF_THREADS; this image has no oneTBB), and its LinSysSolver is this repository's
    CPU multifrontal Cholesky on the same number of OpenMP threads (this image has no CHOLMOD).  The first `steps` time steps of the bench scene
    (mat150, `script twist`, BE, dt 0.04, no gravity, self-collision off: >= 10 Newton iterations); the time is the reference's own `descent` timer."""
    import re
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "oracle", "_ref", "libipcref.so")
    if not os.path.exists(so):
        return {"value": None, "kind": "reference", "note": "oracle/_ref/libipcref.so not present on this box"}
    sys.path.insert(0, os.path.join(here, "tools"))
    host = os.cpu_count() or 1
    # measured on a 256-core GPU box (round 4): 16 threads 1.87 it/s, 64 threads 0.97 it/s, 256 threads > 120 s for the sample -- the short loops of the
    # reference and the CPU Cholesky behind it stop scaling long before that; 16 is also what the port above uses (the reference's batch scripts ran 8-12)
    cores = int(os.environ.get("IPC_BENCH_REF_THREADS", min(host, 16)))
    saved = os.environ.get("IPCREF_THREADS")
    try:
        import ref_compare as rc
        from ipc_amd import lib as gl
        os.environ["IPCREF_THREADS"] = str(cores)
        with tempfile.TemporaryDirectory(prefix="ipcref_bench_") as tmp:
            gl.save_tet_mesh(os.path.join(tmp, "mat.msh"), V, F)
            with open(os.path.join(tmp, "scene.txt"), "w") as f:
                f.write(f"energy NH\ntimeIntegration BE\ntime {0.04 * steps:.17g} 0.04\ndensity 1000\nstiffness 2e4 0.4\nturnOffGravity\nscript twist\n"
                        f"shapes input 1\n{tmp}/mat.msh 0 0 0  0 0 0  1 1 1\nselfCollisionOff\n")
            t0 = time.perf_counter()
            rcode, log = rc.run_reference(os.path.join(tmp, "scene.txt"), os.path.join(tmp, "out"), timeout=180, cwd=tmp)
            wall = time.perf_counter() - t0
            if rcode != 0:
                return {"value": None, "kind": "reference", "note": "the reference run failed: " + log[-300:]}
            its = int(rc.read_iter_counts(os.path.join(tmp, "out"), steps).sum())
            info = open(os.path.join(tmp, "out", f"info{steps}.txt")).read()
            m = re.search(r"([0-9.eE+-]+) s: descent", info)
            descent = float(m.group(1)) if m else wall
        return {"value": its / descent, "unit": "iter/s", "cores": cores, "host_cores": host, "kind": "reference",
                "sample": f"time steps 1-{steps} of the bench scene through the reference's own main() (libipcref.so): {its} Newton iterations in {descent:.1f} s "
                          f"of its `descent` timer ({wall:.1f} s with set-up)",
                "note": f"the reference's sources on {cores} threads: its tbb::parallel_for loops on a std::thread pool (no oneTBB in this image), its LinSysSolver "
                        "= this repository's CPU multifrontal Cholesky on the same threads (no CHOLMOD in this image); IPCREF_THREADS=1 is the serial build the fixtures come from"}
    except Exception as e:  # noqa: BLE001  (a reported side figure must not take the bench line down)
        return {"value": None, "kind": "reference", "note": f"not measured: {e!r}"[:300]}
    finally:
        if saved is None:
            os.environ.pop("IPCREF_THREADS", None)
        else:
            os.environ["IPCREF_THREADS"] = saved


if __name__ == "__main__":
    main()
