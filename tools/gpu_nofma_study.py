"""The six GPU-only scene budgets of tests/test_gpu_vs_reference.py, re-run on a library built WITHOUT multiply-add contraction
(python ipc_amd/build.py --nofma -> ipc_amd/libipcgpu_nofma.so, every file -ffp-contract=off; select with IPCGPU_LIB_VARIANT=nofma).

VERDICT round 4, weak #1: the explanation offered for the budgets (FMA contraction in the element kernels + a touch-down from exact rest)
had never been demonstrated.  This prints, for each scene and for the library that is loaded, the Newton counts beside the reference's, the
number of differing steps, the first differing step and the relative position deviation per step -- next to the CPU restatement's budgets.

    IPCGPU_LIB_VARIANT=nofma python tools/gpu_nofma_study.py > gpurun_out/nofma/nofma.txt
    python tools/gpu_nofma_study.py > gpurun_out/nofma/default.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from test_oracle_vs_reference import load_scene, run_scene  # noqa: E402

import ipc_amd.lib as gpu_lib  # noqa: E402

# (fixture, steps, the CPU restatement's mismatch budget, its end-position tolerance)
SCENES = [("dbc_time_range", 30, 3, 1e-2), ("aligned_cubes", 30, 3, 1e-2), ("aligned_cubes_fric", 30, 2, 1e-2), ("attach", None, 0, 5e-3),
          ("two_cubes_fall", None, 8, 1e-2), ("script_stamp_inv", None, 0, 1e-11)]


def main():
    only = sys.argv[1:]
    print("library:", gpu_lib.lib_path())
    for name, steps, mism, tol in SCENES:
        if only and name not in only:
            continue
        S, meshes = load_scene(name)
        n_steps = min(int(S["steps"]), steps) if steps else int(S["steps"])
        c = gpu_lib.Context(0)
        pos, its = run_scene(S, meshes, c, n_steps)
        c.close()
        ref_its = S["iters"][:len(its)]
        ref = S["positions"]
        n = min(pos.shape[1], ref.shape[1])
        dev = [float(np.abs(pos[s][:n] - ref[s][:n]).max() / np.abs(ref[s]).max()) for s in range(len(its))]
        differ = np.nonzero(its != ref_its)[0]
        print(f"== {name}: {len(differ)} differing steps (restatement's budget {mism}), first at step {int(differ[0]) + 1 if len(differ) else '-'}, "
              f"sum {int(its.sum())} vs {int(ref_its.sum())}, end deviation {dev[-1]:.3e} (restatement's tolerance {tol:g})")
        print("   gpu", its.tolist())
        print("   ref", ref_its.tolist())
        print("   dev", " ".join("%.1e" % d for d in dev), flush=True)


if __name__ == "__main__":
    main()
