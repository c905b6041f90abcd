"""TEST INFRASTRUCTURE -- how well does the REFERENCE ITSELF define the Newton counts and positions of the scenes whose contact begins from exact rest?

tests/test_gpu_vs_reference.py used to give five scenes GPU-only budgets ("5 mismatching counts instead of 3", an end tolerance that went from 2e-2 to
3e-2 when the elimination order changed) with an explanation -- multiply-add contraction in the element kernels -- that round 5 DISPROVED: a library
built with -ffp-contract=off in every file differs from the reference in as many steps (profiles/r05_nofma_parity_study.txt).  What the scenes have in
common is a touch-down from exact rest (F = I up to round-off, faces exactly parallel, corners exactly above corners): closest-feature typing, the
sigma-space clamps and the line search's energy comparisons are decided by the last bits of the state there.

This tool measures that with the reference as its own witness.  The reference-compiled code (oracle/_ref/libipcref.so) runs the scene's first step, is
continued from ITS OWN status1 (a) as is and (b) N times with every coordinate of status1 moved by a random +-1 ulp; the fixture holds, per step, the
smallest and the largest Newton count over the ensemble and the largest position deviation of a perturbed run from (a), relative to the scene's size.
A step in which the reference disagrees with itself under a one-ulp perturbation cannot be held to "equal counts" for ANY second implementation; the
envelope of the ensemble, widened by a stated factor, is the test's criterion instead (check_envelope, tests/test_oracle_vs_reference.py) -- a criterion
that depends on the reference and the scene alone, not on this repository's code.

Round 6 adds a second witness for the same question, asked of the linear solver alone: `--orders` continues the reference from its own status1 with
NOTHING perturbed but the elimination order of its Cholesky (libipcref.so's LinSysSolver is oracle/orc_chol.cpp, oracle/ref_plug.cpp: ORC_CHOL_LEAF = the
leaf size of the nested dissection, ORC_CHOL_SHUFFLE = a seeded shuffle of the node list the dissection starts from).  The same matrix factorised in
another order gives a search direction that differs in the last bits; the fixture records the Newton counts of those runs as `order_iters`.

    python tools/make_golden_ensemble.py [--seeds 24] [--jobs 4] [--orders] [names ...]        (build container only: needs /root/reference and oracle/_ref)
"""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import masonry_perturb as mp  # noqa: E402
import ref_compare as rc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# (fixture name, scene under the reference's input/, appended text, steps of the fixture) -- the same scenes, texts and step counts as in tools/make_golden_ref.py
SCENES = [
    ("two_cubes_fall", "tutorialExamples/2cubesFall.txt", "", 40),
    ("dbc_time_range", "tutorialExamples/BC/2cubesFall_DBC_timeRange.txt", "", 30),
    ("aligned_cubes", "paperExamples/supplementB/SQPBenchmark/12_alignedCubes.txt", "", 30),
    ("aligned_cubes_fric", "paperExamples/supplementB/SQPBenchmark/12_alignedCubes.txt", "\nselfFric 0.3\n", 40),
    ("attach", "tutorialExamples/advanced/2cubesFall_attach.txt", "", 22),
    # BASELINE configs[4]'s chain: ten interlocked tori caught link after link, each from free fall onto the link above (round 6: --seeds 48 --jobs 6 --orders,
    # five minutes on six cores)
    ("chain10", "paperExamples/videoExamples/chain10.txt", "", 30),
]


ORDERS = [("leaf 8", {"ORC_CHOL_LEAF": "8"}), ("leaf 16", {"ORC_CHOL_LEAF": "16"}), ("leaf 24", {"ORC_CHOL_LEAF": "24"}), ("leaf 6", {"ORC_CHOL_LEAF": "6"}),
          ("shuffle 1", {"ORC_CHOL_SHUFFLE": "1"}), ("shuffle 2", {"ORC_CHOL_SHUFFLE": "2"}), ("shuffle 3", {"ORC_CHOL_SHUFFLE": "3"}),
          ("shuffle 4", {"ORC_CHOL_SHUFFLE": "4"}), ("leaf 8 + shuffle 5", {"ORC_CHOL_LEAF": "8", "ORC_CHOL_SHUFFLE": "5"}),
          ("leaf 16 + shuffle 6", {"ORC_CHOL_LEAF": "16", "ORC_CHOL_SHUFFLE": "6"}), ("leaf 10", {"ORC_CHOL_LEAF": "10"}), ("leaf 14", {"ORC_CHOL_LEAF": "14"})]


def _member(args):
    """one continued run of the reference (a process of its own: run_reference starts the reference in a subprocess that inherits this environment)"""
    lines, dt, status, K, tmp, tag, env = args
    for k, v in env.items():
        os.environ[k] = v
    i, p, _ = mp.continue_reference(lines, dt, status, 1, K, tmp, tag)
    return i, p


def ensemble(path, extra, steps, seeds, jobs=1, orders=False):
    from ipc_amd import scene_script as ss
    text = open(os.path.join(mp.REF_ROOT, "input", path)).read() + extra
    lines = [ln for ln in text.split("\n") if not ln.strip().startswith("time ")]
    dt = ss.SceneConfig.parse(text, mp.REF_ROOT).dt
    tmp = tempfile.mkdtemp(prefix="ens_")
    first = os.path.join(tmp, "first.txt")
    open(first, "w").write("\n".join(lines) + f"\ntime {dt:.17g} {dt:.17g}\n")
    rcode, log = rc.run_reference(first, os.path.join(tmp, "ref"), timeout=7200)
    assert rcode == 0, log[-2000:]
    its1 = rc.read_iter_counts(os.path.join(tmp, "ref"), 1)
    status = open(os.path.join(tmp, "ref", "status1")).read()
    K = steps - 1
    base_its, base_pos, _ = mp.continue_reference(lines, dt, status, 1, K, tmp, "base")
    scale = np.abs(base_pos[-1]).max()
    import multiprocessing
    work = [(lines, dt, mp.perturb_status(status, "all", s), K, tmp, f"s{s}", {}) for s in range(seeds)]
    if orders:
        work += [(lines, dt, status, K, tmp, f"o{j}", env) for j, (_, env) in enumerate(ORDERS)]
    with multiprocessing.get_context("spawn").Pool(max(1, jobs)) as pool:
        res = pool.map(_member, work, chunksize=1)
    its = np.array([r[0] for r in res[:seeds]])
    dev = np.array([[float(np.abs(r[1][k] - base_pos[k]).max() / scale) for k in range(K)] for r in res[:seeds]])
    one = np.array([int(its1[0])])
    out = dict(base_iters=np.concatenate([one, base_its]), ens_min=np.concatenate([one, its.min(0)]), ens_max=np.concatenate([one, its.max(0)]),
               ens_dev=np.concatenate([[0.0], dev.max(0)]), ens_mismatches=np.array([int((a != base_its).sum()) for a in its]), ens_iters=its.astype(np.int16),
               seeds=np.array(seeds))
    if orders:
        oi = np.array([np.concatenate([one, r[0]]) for r in res[seeds:]])
        od = np.array([[0.0] + [float(np.abs(r[1][k] - base_pos[k]).max() / scale) for k in range(K)] for r in res[seeds:]])
        out.update(order_iters=oi.astype(np.int16), order_dev=od, order_labels=np.array([lab for lab, _ in ORDERS]))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=24)
    ap.add_argument("--jobs", type=int, default=1, help="members run side by side (each is a single-threaded process)")
    ap.add_argument("--orders", action="store_true", help="also: the unperturbed state with the elimination order of the reference's Cholesky varied")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    for name, path, extra, steps in SCENES:
        if a.names and name not in a.names:
            continue
        E = ensemble(path, extra, steps, a.seeds, a.jobs, a.orders)
        S = np.load(os.path.join(GOLD, f"ref_scene_{name}.npz"))
        assert np.array_equal(E["base_iters"], S["iters"][:steps]), (name, E["base_iters"].tolist(), S["iters"].tolist())  # the continued run IS the fixture's run
        np.savez_compressed(os.path.join(GOLD, f"ref_ensemble_{name}.npz"), **E)
        print(f"{name}: base {E['base_iters'].tolist()}\n   min {E['ens_min'].tolist()}\n   max {E['ens_max'].tolist()}\n   mismatching steps per seed {E['ens_mismatches'].tolist()}\n"
              f"   max deviation per step {' '.join('%.1e' % d for d in E['ens_dev'])}", flush=True)
        if a.orders:
            for lab, row in zip(E["order_labels"], E["order_iters"]):
                print(f"   elimination order [{lab}]: {row.tolist()}", flush=True)
