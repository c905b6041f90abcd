"""Scratch diagnosis on the GPU box (not a test): Newton counts of the reference scenes on the HIP stepper."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ipc_amd
from test_oracle_vs_reference import load_scene, run_scene
ipc_amd.load_library()
for name in ("two_cubes_fall", "rotate_co", "rotate_co_surface", "dbc_time_range", "aligned_cubes", "aligned_cubes_fric", "bar_twist_damped", "two_cubes_nm_damped", "cubes_dhat_homotopy", "point_triangle_rotated"):
    S, meshes = load_scene(name)
    c = ipc_amd.Context(0)
    try:
        pos, its = run_scene(S, meshes, c, int(S["steps"]))
        n = min(pos.shape[1], S["positions"].shape[1])
        dev = [float(np.abs(pos[s][:n] - S["positions"][s][:n]).max()) for s in range(len(pos))]
        print(name, "\n  gpu", its.tolist(), "\n  ref", S["iters"].tolist(), "\n  max dev per step", " ".join(f"{d:.0e}" for d in dev), flush=True)
    except Exception as e:
        print(name, "FAILED:", str(e)[:300], flush=True)
    c.close()
