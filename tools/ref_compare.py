"""TEST INFRASTRUCTURE -- run one scene script through (a) the reference's own main.cpp / Optimizer.cpp, compiled from
/root/reference into oracle/_ref/libipcref.so (oracle/Makefile.ref), and (b) the CPU oracle of this repository, and print how
they compare: Newton iterations per time step and the positions after each step.

    python tools/ref_compare.py <scene.txt> [--steps N] [--keep DIR]

Build container only (needs /root/reference and oracle/_ref)."""
import argparse
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
REF_ROOT = "/root/reference"
LIBREF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libipcref.so")

RUNNER = """
import ctypes, sys
l = ctypes.CDLL(%r)
a = [b"ipc"] + [x.encode() for x in sys.argv[1:]]
sys.exit(l.ipcref_main(len(a), (ctypes.c_char_p * len(a))(*a)))
"""


def run_reference(scene_path, out_dir, timeout=3600, cwd=REF_ROOT):
    """The reference in offline mode (progMode 100); its relative mesh paths resolve against its repository root."""
    os.makedirs(out_dir, exist_ok=True)
    r = subprocess.run([sys.executable, "-c", RUNNER % LIBREF, "100", os.path.abspath(scene_path), "-o", out_dir.rstrip("/") + "/", "--logLevel", "off"],
                       cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return r.returncode, r.stdout.decode(errors="replace")


def read_status_positions(path):
    """`position n dim` section of a status file (Optimizer.cpp:2964-3011)."""
    with open(path) as f:
        tok = f.read().split()
    i = tok.index("position")
    n, d = int(tok[i + 1]), int(tok[i + 2])
    return np.array(tok[i + 3:i + 3 + n * d], dtype=np.float64).reshape(n, d)


def read_iter_counts(out_dir, steps):
    """info<N>.txt line 2: `<time steps so far> <Newton iterations so far> <average>` (main.cpp saveInfoForPresent)."""
    tot = []
    for s in range(1, steps + 1):
        with open(os.path.join(out_dir, f"info{s}.txt")) as f:
            f.readline()
            tot.append(int(f.readline().split()[1]))
    return np.diff(np.array([0] + tot))


def run_oracle(scene_path, steps, nthreads=8):
    from ipc_amd import lib, scene_script as ss
    from oracle import orc
    from test_scene_script import OracleBackend
    cfg = ss.SceneConfig.parse(open(scene_path).read(), REF_ROOT)
    sc = ss.assemble(cfg, lib.read_tet_mesh)
    be = ss.apply(sc, OracleBackend(orc, nthreads=nthreads))
    pos, its = [], []
    for s in range(steps):
        sc.before_step(be, s * cfg.dt)
        its.append(be.solve_timestep(10000))
        pos.append(be.state()["V"].copy())
    return np.array(pos), np.array(its)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--keep", default=None)
    a = ap.parse_args()
    out = a.keep or tempfile.mkdtemp(prefix="ipcref_")
    # limit the reference to the wanted number of steps through a copy of the script with `time` overridden
    txt = open(a.scene).read()
    from ipc_amd import scene_script as ss
    cfg = ss.SceneConfig.parse(txt, REF_ROOT)
    tmp_scene = os.path.join(out, "scene.txt")
    os.makedirs(out, exist_ok=True)
    lines = [ln for ln in txt.splitlines() if not ln.strip().startswith("time ")]
    open(tmp_scene, "w").write("\n".join(lines) + f"\ntime {a.steps * cfg.dt:.17g} {cfg.dt:.17g}\n")
    rc, log = run_reference(tmp_scene, os.path.join(out, "ref"))
    print("reference rc", rc)
    if rc != 0:
        print(log[-3000:])
        sys.exit(1)
    ref_dir = os.path.join(out, "ref")
    cum_r = {}  # Newton iterations after step N, where the reference wrote info<N>.txt (at most 100 per simulated second, main.cpp:408-420)
    have_pos = []
    for sN in range(1, a.steps + 1):
        f = os.path.join(ref_dir, f"info{sN}.txt")
        if os.path.exists(f):
            with open(f) as fh:
                fh.readline()
                cum_r[sN] = int(fh.readline().split()[1])
        if os.path.exists(os.path.join(ref_dir, f"status{sN}")):
            have_pos.append(sN)
    last = max(list(cum_r) + have_pos + [0])
    if last < a.steps:
        print(f"the reference stopped after {last} of {a.steps} steps; last lines of its log:")
        print("".join(open(os.path.join(ref_dir, "log.txt")).readlines()[-6:]))
        a.steps = last
    pos_o, its_o = run_oracle(a.scene, a.steps)
    cum_o = np.cumsum(its_o)
    prev_r = prev_o = 0
    for sN in range(1, a.steps + 1):
        msg = f"step {sN}:"
        if sN in cum_r:
            msg += f" Newton iterations reference {cum_r[sN] - prev_r:3d}  oracle {int(cum_o[sN - 1]) - prev_o:3d}"
            prev_r, prev_o = cum_r[sN], int(cum_o[sN - 1])
        if sN in have_pos:
            Pr = read_status_positions(os.path.join(ref_dir, f"status{sN}"))
            # kinematic mesh obstacles (`meshCO`) are separate objects in the reference and trailing surface-only nodes here
            dev = np.abs(Pr - pos_o[sN - 1][:Pr.shape[0]]).max() / np.abs(Pr).max()
            msg += f"   max |dx| / scale = {dev:.3e}"
        if sN in cum_r or sN in have_pos:
            print(msg)
