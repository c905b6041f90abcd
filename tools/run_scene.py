"""Run one of the reference's scene scripts (src/Config.cpp grammar) on the GPU through the C ABI.
usage: python tools/run_scene.py <scene.txt> [--root DIR] [--steps N] [--status-every K] [--out DIR]
`--root` is the directory the script's relative mesh paths are resolved against (the reference resolves them against its
repository root).  Prints one line per time step; writes `status<N>` checkpoints in the reference's format."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipc_amd import lib, scene_script as ss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene")
ap.add_argument("--root", default=None)
ap.add_argument("--steps", type=int, default=None)
ap.add_argument("--status-every", type=int, default=0)
ap.add_argument("--out", default=".")
args = ap.parse_args()

root = args.root or os.path.dirname(os.path.abspath(args.scene))
cfg = ss.SceneConfig.parse(open(args.scene).read(), root)
sc = ss.assemble(cfg, lib.read_tet_mesh)
print(f"{len(cfg.shapes)} shapes, {sc.V.shape[0]} nodes, {sc.T.shape[0]} tets, {sc.SF.shape[0]} surface triangles, dt = {cfg.dt}")
c = ss.apply(sc, lib.Context(0))
steps = args.steps if args.steps is not None else int(round(cfg.duration / cfg.dt))
for step in range(steps):
    t0 = time.time()
    sc.before_step(c, step * cfg.dt)  # state-dependent script decisions (AnimScripter::stepAnimScript)
    it = c.solve_timestep(1000)
    st = c.state()
    cs = c.contact_state() if cfg.self_collision or cfg.half_spaces else {}
    print(f"step {st['timestep']:5d}  {it:4d} Newton iterations  {1e3 * (time.time() - t0):8.1f} ms  E = {st['E']:.6e}  active = {cs.get('nActive', 0)}", flush=True)
    if args.status_every and st["timestep"] % args.status_every == 0:
        c.save_status(os.path.join(args.out, f"status{st['timestep']}"))
c.close()
