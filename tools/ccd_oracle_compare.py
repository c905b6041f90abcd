"""TEST INFRASTRUCTURE -- how far is the conservative advancement (oracle/orc_contact.cpp::accd, the per-pair query of the product's CCD)
from the polynomial statement of the published thickened CTCD query (oracle/ccd_poly.py)?  Measured on every point-triangle / edge-edge
candidate pair of a contact scene (two stacked mat sheets, the candidate list of the constraint-set build), for several search
directions, with the thickness the reference passes: eta = (1 - slackness) * current distance, slackness 0.8.

    python tools/ccd_oracle_compare.py [--n 100] [--dirs 4] [--procs 8] [--out profiles/r03_ccd_oracle_compare.txt]"""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ipc_amd import scene  # noqa: E402
from oracle import ccd_poly, orc  # noqa: E402


def one(args):
    kind, X, V = args
    d0 = np.sqrt(orc.unclassified_d2(kind, X))
    ta = orc.accd(kind, X, V, eta=0.2, tmax=1.0)
    tp = ccd_poly.toi(kind, X, V, 0.2 * d0)
    return ta, (tp if tp is not None else np.inf), d0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--dirs", type=int, default=4)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    V, F, nA = scene.make_mat_stack(a.n, 2, gap=1.2e-3)
    Vs = scene.jitter(V, F, rel=2e-3)
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
    m.set_surface(SF)
    m.set_V(Vs)
    dHat = 1e-6 * m.features()["bboxDiag2"]
    sets = orc.Contacts().build(m, dHat)
    cand = sets["cs_ptee"]
    SVI, SFE = orc.mesh_surface(m)
    rng = np.random.default_rng(20260925)
    lines = [f"scene: 2 x mat{a.n} stack, {V.shape[0]} nodes, {len(cand)} point-triangle / edge-edge candidate pairs, dHat = 1e-6 diag^2; "
             f"eta = 0.2 x current distance (slackness 0.8), window [0, 1]"]
    tot = dict(n=0, hits=0, worst_late=0.0, worst_early=0.0, disagree=0)
    t0 = time.time()
    for di in range(a.dirs):
        p = 1e-3 * rng.standard_normal(V.shape)
        p[nA:, 1] -= 2e-3 * (1 + di)  # the upper sheet moves into the lower one
        jobs = []
        for c0, c1 in cand:
            if c0 < 0:  # (-svI - 1, sfI)
                nodes = [SVI[-c0 - 1], *SF[c1]]
                kind = 2
            else:
                nodes = [*SFE[c0], *SFE[c1]]
                kind = 3
            jobs.append((kind, Vs[nodes].copy(), p[nodes].copy()))
        with Pool(a.procs) as pool:
            res = np.array(pool.map(one, jobs, chunksize=512))
        ta, tp, d0 = res[:, 0], res[:, 1], res[:, 2]
        hit_p, hit_a = np.isfinite(tp), ta < 1.0
        both = hit_p & hit_a
        diff = tp[both] - ta[both]  # > 0: the advancement stops earlier (conservative side)
        only = int(np.sum(hit_p != hit_a))
        # a pair one statement reports and the other does not: the first contact sits at the end of the window
        edge = np.abs(np.where(hit_p, tp, 1.0) - np.where(hit_a, ta, 1.0))[hit_p != hit_a]
        lines.append(f"direction {di}: {len(jobs)} pairs, {int(both.sum())} reach the gap inside the window by both statements; t_poly - t_advance: "
                     f"max {diff.max() if diff.size else 0:.3e}, min {diff.min() if diff.size else 0:.3e}, median {np.median(diff) if diff.size else 0:.3e}; "
                     f"{only} pairs hit by one statement only (all within {edge.max() if edge.size else 0:.1e} of the window's end)")
        tot["n"] += len(jobs)
        tot["hits"] += int(both.sum())
        if diff.size:
            tot["worst_early"] = max(tot["worst_early"], float(diff.max()))
            tot["worst_late"] = max(tot["worst_late"], float(-diff.min()))
        tot["disagree"] += only
    lines.append(f"total: {tot['n']} queries, {tot['hits']} hits; the advancement is at most {tot['worst_early']:.3e} earlier and at most {tot['worst_late']:.3e} "
                 f"later (units of the step) than the polynomial statement; {tot['disagree']} window-end disagreements; {time.time() - t0:.0f} s")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
