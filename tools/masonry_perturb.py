"""TEST INFRASTRUCTURE -- the one open parity item of round 3: `otherExamples/friction/masonryArch_25.txt` (`fricIterAmt -1`) continued from the
reference's own status file gave reference [83, 21, 2, 12] against restatement [47, 33, 2, 15] Newton iterations.

This tool answers whether that is a difference of the two implementations or round-off amplification inside the scene, with the REFERENCE
ITSELF as the witness: the reference-compiled code (oracle/_ref/libipcref.so) is continued from its own status<R> file
  (a) as is,
  (b) with ONE coordinate of ONE node moved by one unit in the last place,
  (c) with every coordinate moved by a random +-1 ulp,
and beside them the CPU restatement from the unperturbed file.  If (b) / (c) change the reference's own counts as much as the restatement's
differ from (a), the scene amplifies last-bit differences and no two implementations can agree on it count for count.

    python tools/masonry_perturb.py [--scene input/otherExamples/friction/masonryArch_25.txt] [--restart 4] [--steps 4] [--out profiles/...]

Build container only (needs /root/reference and oracle/_ref)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import ref_compare as rc  # noqa: E402

REF_ROOT = "/root/reference"


def perturb_status(text, mode, seed=0):
    """status file with its `position n dim` block moved by ulps: mode 'one' = the first coordinate of node n/2 by +1 ulp, 'all' = every coordinate by a random +-1 ulp"""
    tok = text.split("\n")
    # find the line that starts the position block
    for li, line in enumerate(tok):
        if line.startswith("position"):
            break
    else:
        raise RuntimeError("no position block")
    n, d = int(line.split()[1]), int(line.split()[2])
    rows = [list(map(float, tok[li + 1 + i].split())) for i in range(n)]
    X = np.array(rows, dtype=np.float64)
    if mode == "one":
        X[n // 2, 0] = np.nextafter(X[n // 2, 0], np.inf)
    else:
        rng = np.random.default_rng(seed)
        up = rng.integers(0, 2, size=X.shape).astype(bool)
        X = np.where(up, np.nextafter(X, np.inf), np.nextafter(X, -np.inf))
    for i in range(n):
        tok[li + 1 + i] = " ".join(repr(float(v)) for v in X[i])
    return "\n".join(tok)


def status_precision_ok(text):
    """the reference writes positions with enough digits to round-trip?  (if not, a 1-ulp perturbation would be lost in the file format)"""
    for line in text.split("\n"):
        if line.startswith("position"):
            continue
    return True


def continue_reference(lines, dt, status_text, R, K, tmp, tag):
    spath = os.path.join(tmp, f"status_{tag}")
    open(spath, "w").write(status_text)
    path2 = os.path.join(tmp, f"scene_{tag}.txt")
    open(path2, "w").write("\n".join(lines) + f"\ntime {(R + K) * dt:.17g} {dt:.17g}\nrestart {spath}\n")
    out = os.path.join(tmp, f"ref_{tag}")
    t0 = time.time()
    rcode, log = rc.run_reference(path2, out, timeout=7200)
    assert rcode == 0, log[-2000:]
    cum = []
    for sN in range(R + 1, R + K + 1):
        with open(os.path.join(out, f"info{sN}.txt")) as fh:
            fh.readline()
            cum.append(int(fh.readline().split()[1]))
    its = np.diff(np.array([0] + cum))
    pos = np.array([rc.read_status_positions(os.path.join(out, f"status{sN}")) for sN in range(R + 1, R + K + 1)])
    return its, pos, time.time() - t0


def continue_oracle(scene_text, status_path, K, nthreads=8):
    from ipc_amd import lib, scene_script as ss
    from oracle import orc
    from test_scene_script import OracleBackend
    cfg = ss.SceneConfig.parse(scene_text, REF_ROOT)
    cfg.restart = status_path
    sc = ss.assemble(cfg, lib.read_tet_mesh)
    be = ss.apply(sc, OracleBackend(orc, nthreads=nthreads))
    pos, its = [], []
    t = float(getattr(sc, "restart_time", 0.0) or 0.0)
    for s in range(K):
        sc.before_step(be, t + s * cfg.dt)
        its.append(be.solve_timestep(100000))
        pos.append(be.state()["V"].copy())
    return np.array(its), np.array(pos)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="input/otherExamples/friction/masonryArch_25.txt")
    ap.add_argument("--restart", type=int, default=4)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    scene_path = os.path.join(REF_ROOT, a.scene)
    text = open(scene_path).read()
    lines = [ln for ln in text.split("\n") if not ln.strip().startswith("time ")]
    from ipc_amd import scene_script as ss
    cfg = ss.SceneConfig.parse(text, REF_ROOT)
    dt = cfg.dt
    R, K = a.restart, a.steps
    tmp = tempfile.mkdtemp(prefix="masonry_")
    report = []

    def say(s):
        print(s, flush=True)
        report.append(s)

    say(f"scene {a.scene}: reference run to step {R}, then continued for {K} steps from its own status{R}")
    path1 = os.path.join(tmp, "scene_first.txt")
    open(path1, "w").write("\n".join(lines) + f"\ntime {R * dt:.17g} {dt:.17g}\n")
    t0 = time.time()
    rcode, log = rc.run_reference(path1, os.path.join(tmp, "ref"), timeout=7200)
    assert rcode == 0, log[-2000:]
    say(f"  first {R} steps by the reference: {time.time() - t0:.0f} s")
    status = open(os.path.join(tmp, "ref", f"status{R}")).read()
    base_its, base_pos, tt = continue_reference(lines, dt, status, R, K, tmp, "base")
    say(f"  (a) reference from its own status{R}:                      Newton iterations {base_its.tolist()}   ({tt:.0f} s)")
    scale = np.abs(base_pos[-1]).max()
    runs = [("one", 0, "(b) reference, ONE coordinate of one node +1 ulp")]
    runs += [("all", s, f"(c) reference, every coordinate +-1 ulp (seed {s})") for s in range(a.seeds)]
    for mode, seed, label in runs:
        its, pos, tt = continue_reference(lines, dt, perturb_status(status, mode, seed), R, K, tmp, f"{mode}{seed}")
        dev = [float(np.abs(pos[s] - base_pos[s]).max() / scale) for s in range(K)]
        say(f"  {label}: Newton iterations {its.tolist()}   positions vs (a) per step {['%.1e' % d for d in dev]}   ({tt:.0f} s)")
    if not a.no_oracle:
        spath = os.path.join(tmp, "status_oracle")
        open(spath, "w").write(status)
        t0 = time.time()
        its, pos = continue_oracle(text, spath, K)
        n = min(pos.shape[1], base_pos.shape[1])
        dev = [float(np.abs(pos[s][:n] - base_pos[s][:n]).max() / scale) for s in range(K)]
        say(f"  (d) CPU restatement from the same status{R}:               Newton iterations {its.tolist()}   positions vs (a) per step {['%.1e' % d for d in dev]}   ({time.time() - t0:.0f} s)")
    if a.out:
        open(a.out, "w").write("\n".join(report) + "\n")
