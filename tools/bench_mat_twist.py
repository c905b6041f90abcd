"""matTwist AS SHIPPED (input/paperExamples/14_matTwist.txt:15, 21_scalability/mat150x150_twist.txt: `selfCollisionOn`): the bench scene of bench.py with the
interior-point contact machinery switched on, as an IPC user runs it -- broad phase, constraint sets, CCD and intersection checks in every Newton iteration
from the first step on, barrier terms once the sheet wraps onto itself.  (BASELINE configs[1] strips the self-contact; the headline follows BASELINE, this
is the sub-record `mat_twist_as_shipped` beside it.)

Two timed windows: the first `early_steps` time steps (nothing active yet) and `contact_steps` steps from the first step whose constraint set holds at
least `min_active` stencils (the sheet has wrapped).  usage: python tools/bench_mat_twist.py [--n 150]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipc_amd import lib, scene  # noqa: E402

NAMES = {1: "pattern_change:set_pattern", 2: "pattern_change:symbolic_analysis", 3: "factorisation+triangular_sweeps", 5: "linesearch_moves+intersection",
         9: "energy_evals", 13: "step_bounds(inversion+CCD+CFL)", 14: "constraint_sets", 11: "timestep"}


def make_context(n=150, ctx=None, pad=None, positions=None):
    """the bench scene of bench.py (mat N, twist handles, BE dt 0.04, no gravity) with `selfCollisionOn` (dHat 1e-3 of the bounding-box diagonal, Config.hpp)"""
    V, F = scene.make_mat(n)
    left, right = scene.border_verts(V, 0.01)
    SF = scene.surface_tris(F)
    c = ctx or lib.Context(0)
    c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    if positions is not None:
        c.set_positions(positions(V, F))  # tests: a generic (jittered, pre-twisted) start instead of exact rest, where the reference itself is round-off dependent
    c.opt_init(0.04, False)
    c.set_surface(SF)
    c.set_twist(left, right, 0.4 * np.pi)
    c.enable_self_collision(1e-3)
    if pad is not None:
        c.set_pattern_lookahead(pad)
    return c, dict(V=V, F=F, SF=SF, left=left, right=right)


def one_step(c, max_iter=100):
    c.begin_timestep()
    it = 0
    while it < max_iter:
        if c.newton_iter():
            break
        it += 1
    c.end_timestep()
    return it


def window(c, steps, max_iter=100):
    tm0 = c.timers()
    t0 = time.perf_counter()
    its = [one_step(c, max_iter) for _ in range(steps)]
    wall = time.perf_counter() - t0
    tm = c.timers() - tm0
    n = max(sum(its), 1)
    split = {"assembly+barrier_hessian": 1e3 * (tm[0] - tm[1] - tm[2]) / n}
    split.update({v: 1e3 * tm[k] / n for k, v in NAMES.items()})
    st = c.contact_state()
    return {"time_steps": steps, "newton_iterations": int(sum(its)), "iterations_per_step": its, "value": sum(its) / wall, "unit": "iter/s", "ms_per_iter": 1e3 * wall / n,
            "split_ms_per_iter": split, "active_constraints_at_end": st["nActive"], "mollified_at_end": st["nPara"], "candidates_at_end": st["nCand"],
            "pattern_changes_so_far": st["nPatternChanges"], "full_ccd_so_far": st["nFullCCD"]}


def advance_to_contact(c, min_active=1000, max_steps=200, max_iter=100):
    """untimed: time steps until the constraint set at the end of a step holds >= min_active stencils; returns the number of steps taken"""
    taken = 0
    while taken < max_steps and c.contact_state()["nActive"] < min_active:
        one_step(c, max_iter)
        taken += 1
    return taken


def run(n=150, early_steps=5, contact_steps=5, min_active=1000, max_steps=200, pad=None):
    c, S = make_context(n, pad=pad)
    t0 = time.time()
    c.precompute()
    t_pre = time.time() - t0
    early = window(c, early_steps)
    t0 = time.time()
    skipped = advance_to_contact(c, min_active, max_steps)
    t_adv = time.time() - t0
    first = early_steps + skipped + 1
    rec = {"workload": f"matTwist mat{n} AS SHIPPED (14_matTwist.txt: selfCollisionOn, dHat 1e-3): {S['V'].shape[0]} nodes / {S['F'].shape[0]} tets, "
                       f"{S['SF'].shape[0]} surface triangles, twist DBC, BE dt 0.04", "precompute_s": t_pre,
           "early": dict(window_steps=f"1-{early_steps}", **early)}
    if c.contact_state()["nActive"] >= min_active:
        rec["wrapped"] = dict(window_steps=f"{first}-{first + contact_steps - 1}", untimed_steps_before=skipped, untimed_s=t_adv, **window(c, contact_steps))
        rec["wrapped"]["intersected_at_end"] = bool(c.is_intersected())
    else:
        rec["wrapped"] = {"value": None, "note": f"no step up to {early_steps + skipped} reached {min_active} active stencils"}
    c.close()
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=150)
    ap.add_argument("--early-steps", type=int, default=5)
    ap.add_argument("--contact-steps", type=int, default=5)
    ap.add_argument("--min-active", type=int, default=1000)
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--pad", type=float, default=None, help="look-ahead of the contact pattern in units of dHat (default: the library's)")
    a = ap.parse_args()
    print(json.dumps(run(a.n, a.early_steps, a.contact_steps, a.min_active, a.max_steps, a.pad)))
