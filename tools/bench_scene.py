"""BASELINE's contact configurations AS THE REFERENCE SHIPS THEM, timed on the HIP stepper: a scene fixture under tests/golden/ (the reference's own
scene script + the meshes it names + what the reference's own main() returned: Newton iterations per step, positions; tools/make_golden_ref.py)
is run through the scene tooling on cuda:0; a window of its time steps is timed and the Newton counts of every step are compared with the
reference's in the same record.  Used by bench.py for `rods_twist` (configs[3]: paperExamples/4_rodsTwist.txt, 202 K tets, self-contact on) and
`sphere_on_mat` (configs[2]: paperExamples/12_sphereOnMat.txt, the contact steps 29-36).
usage: python tools/bench_scene.py <fixture> [steps] [first timed step (1-based)]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_ROOT = "/root/reference"  # only the prefix the fixture's mesh keys are relative to: nothing is read from it

BUCKETS = {1: "pattern_change:set_pattern", 2: "pattern_change:symbolic_analysis", 3: "factorisation+triangular_sweeps", 4: "search_direction_read_back",
           5: "linesearch_moves+intersection", 9: "energy_evals", 13: "step_bounds(inversion+CCD+CFL)", 14: "constraint_sets", 11: "timestep"}


def run(name, steps=None, timed_from=1, what=""):
    import ipc_amd
    from ipc_amd import scene_script as ss
    S = np.load(os.path.join(GOLD, f"ref_scene_{name}.npz"))
    meshes = {str(k): (S[f"mesh{i}_V"], S[f"mesh{i}_T"], S[f"mesh{i}_SF"]) for i, k in enumerate(S["mesh_keys"])}
    steps = int(S["steps"]) if steps is None else min(int(steps), int(S["steps"]))
    cfg = ss.SceneConfig.parse(str(S["script"]), REF_ROOT)

    def key(p):
        return os.path.relpath(str(p), REF_ROOT)

    read_obj = ss.read_obj
    ss.read_obj = lambda p: (meshes[key(p)][0].copy(), meshes[key(p)][2].copy())
    try:
        sc = ss.assemble(cfg, lambda p: tuple(a.copy() for a in meshes[key(p)][:3]))
    finally:
        ss.read_obj = read_obj
    t0 = time.perf_counter()
    c = ss.apply(sc, ipc_amd.Context(0))
    t_setup = time.perf_counter() - t0
    its, wall = [], []
    tm0 = None
    for s in range(steps):
        if s + 1 == timed_from:
            tm0 = c.timers().copy()
        sc.before_step(c, s * cfg.dt)
        t0 = time.perf_counter()
        its.append(int(c.solve_timestep(10000)))  # returns after the step's last synchronisation
        wall.append(time.perf_counter() - t0)
    tm = c.timers() - tm0
    n_nodes, n_tets = int(sc.V.shape[0]), int(sc.T.shape[0])
    c.close()
    ref = [int(x) for x in S["iters"][:steps]]
    k = sum(its[timed_from - 1:])
    w = sum(wall[timed_from - 1:])
    split = {"assembly+barrier_hessian(+host connectivity of a pattern change)": 1e3 * (tm[0] - tm[1] - tm[2]) / max(k, 1)}
    split.update({v: 1e3 * tm[b] / max(k, 1) for b, v in BUCKETS.items()})
    return {"workload": what or name, "fixture": f"tests/golden/ref_scene_{name}.npz", "n_nodes": n_nodes, "n_tets": n_tets,
            "steps_run": steps, "steps_timed": [timed_from, steps], "newton_iterations_timed": k, "value": k / w, "unit": "iter/s", "ms_per_iter": 1e3 * w / max(k, 1),
            "split_ms_per_iter": split, "newton_iterations_per_step": its, "reference_iterations_per_step": ref, "counts_equal_the_references": its == ref,
            "setup_s": t_setup}


if __name__ == "__main__":
    a = sys.argv[1:]
    print(json.dumps(run(a[0], int(a[1]) if len(a) > 1 else None, int(a[2]) if len(a) > 2 else 1)))
