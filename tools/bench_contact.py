"""Contact half of the hot path on hardware: a stack of mats, the upper ones falling on the clamped lowest one, self-collision on.
Reports the per-iteration split by the reference's timer buckets and the constraint-set sizes; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel table committed as profiles/r01_contact_kernel_stats.md.
usage: python tools/bench_contact.py [--n 60] [--layers 2] [--steps 3] [--max-iter 12]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipc_amd import lib, scene  # noqa: E402

def run(n=60, layers=2, steps=3, max_iter=12, gap=1.2e-3, cpu_iters=0, pad=None):
    """One run of the contact scene on cuda:0; returns the record (dict) that the command line prints."""
    class A:
        pass
    args = A()
    args.n, args.layers, args.steps, args.max_iter, args.gap, args.cpu_iters = n, layers, steps, max_iter, gap, cpu_iters

    V, F, nA = scene.make_mat_stack(args.n, args.layers, gap=args.gap)
    Vs = scene.jitter(V, F, rel=2e-3)
    SF = scene.surface_tris(F)
    c = lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    c.set_positions(Vs)
    c.opt_init(0.01, True)
    c.set_surface(SF)
    low = np.arange(nA)
    border = low[(np.abs(V[:nA, 0]) > 0.49) | (np.abs(V[:nA, 2]) > 0.49)].astype(np.int32)
    c.set_dbc(border, 1)
    c.enable_self_collision(1e-3)
    if pad is not None:
        c.set_pattern_lookahead(pad)  # A/B of the look-ahead (ipcgpu_opt_set_pattern_lookahead); None = the library's default
    vel = np.zeros_like(V)
    vel[nA:, 1] = -0.05
    c.set_velocity(vel)
    t0 = time.time()
    c.precompute()
    t_pre = time.time() - t0
    tm0 = c.timers()
    iters, counts = 0, []
    t0 = time.time()
    for step in range(args.steps):
        c.begin_timestep()
        for it in range(args.max_iter):
            if c.newton_iter():
                break
            iters += 1
        c.end_timestep()
        counts.append(c.contact_state())
    wall = time.time() - t0
    tm = c.timers() - tm0
    st = c.linsys_stats()  # of the pattern the last iteration factorised (mesh + contact pairs)
    f_ms, s_ms = c.bench_factor_solve(1)
    solver = {"nnzL": st["nnzL"], "factor_gflop": st["flops"] / 1e9, "fronts": st["fronts"], "levels": st["levels"], "factor_ms": f_ms, "solve_ms": s_ms,
              "factor_tflops_per_s": st["flops"] / 1e12 / (f_ms * 1e-3), "frac_of_78.6_TFLOPs": st["flops"] / 1e12 / (f_ms * 1e-3) / 78.6}
    # timer buckets of the stepper (nested ones taken out of their parent, so that the entries add up to the wall time): bucket 0 is the whole
    # of computePrecondMtr and contains set_pattern (1) and the symbolic analysis (2) of a pattern change, plus the host-side connectivity work
    # of such a change; bucket 3 holds the numeric factorisation and -- overlapped with it -- both triangular sweeps (MfNumeric::factorizeSolve)
    names = {1: "pattern_change:set_pattern", 2: "pattern_change:symbolic_analysis", 3: "factorisation+triangular_sweeps", 4: "search_direction_read_back",
             5: "linesearch_moves+intersection", 9: "energy_evals", 13: "step_bounds(inversion+CCD+CFL)", 14: "constraint_sets", 11: "timestep"}
    split = {"assembly+barrier_hessian(+host connectivity of a pattern change)": 1e3 * (tm[0] - tm[1] - tm[2]) / max(iters, 1)}
    split.update({v: 1e3 * tm[k] / max(iters, 1) for k, v in names.items()})
    cpu = None
    if args.cpu_iters > 0:
        from oracle import orc  # noqa: E402  (the checker, timed beside the GPU path as in bench.py's cpu_baseline)
        nthreads = os.cpu_count() or 1
        m = orc.Mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
        m.set_surface(SF)
        m.set_V(Vs)
        m.set_dbc(border, 1)
        o = orc.Optimizer(m, dt=0.01, gravity=True, nthreads=nthreads)
        orc.opt_enable_self_collision(o, 1e-3)
        orc.opt_set_velocity(o, vel)
        o.precompute()
        n_cpu, t0 = 0, time.time()
        for step in range(args.steps):
            o.begin_timestep()
            for it in range(args.max_iter):
                if o.newton_iter():
                    break
                n_cpu += 1
                if n_cpu >= args.cpu_iters:
                    break
            if n_cpu >= args.cpu_iters:
                break
            o.end_timestep()
        t_cpu = time.time() - t0
        cpu = {"kind": "port", "cores": nthreads, "newton_iterations": n_cpu, "iters_per_s": n_cpu / t_cpu, "ms_per_iter_wall": 1e3 * t_cpu / max(n_cpu, 1)}
    return {
        "cpu_baseline": cpu,
        "scene": f"{args.layers} x mat{args.n} stack, gap {args.gap}, dHat 1e-3, self-collision on", "n_nodes": int(V.shape[0]), "n_tets": int(F.shape[0]),
        "n_surface_tris": int(SF.shape[0]), "newton_iterations": iters, "iters_per_s": iters / wall, "ms_per_iter_wall": 1e3 * wall / max(iters, 1),
        "precompute_s": t_pre, "solver": solver, "split_ms_per_iter": split, "contact_state_per_step": counts, "intersected_at_end": bool(c.is_intersected()),
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--max-iter", type=int, default=12)
    ap.add_argument("--gap", type=float, default=1.2e-3)
    ap.add_argument("--cpu-iters", type=int, default=0, help="also time the CPU oracle on the first N Newton iterations of the same scene")
    ap.add_argument("--pad", type=float, default=None, help="look-ahead of the contact pattern in units of dHat (default: the library's, 4)")
    args = ap.parse_args()
    print(json.dumps(run(args.n, args.layers, args.steps, args.max_iter, args.gap, args.cpu_iters, args.pad)))
