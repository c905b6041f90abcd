"""How many sweeps the 3 x 3 one-sided Jacobi SVD of the assembly kernel (svd3, ipc_amd/csrc/nh_device.h) takes on bench-like states of the mat150 mesh: a numpy
model of its sweep logic (same pair order, same 1e-30 skip rule; the count does not depend on last bits), per element and per wave of 64 consecutive elements
(a wave runs as many sweeps as its slowest lane).  Record: profiles/r05_svd_sweep_count_study.txt.   usage: python tools/svd_sweep_count.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ipc_amd import scene
def sweeps(Fs):
    # model of svd3 (ipc_amd/csrc/nh_device.h): cyclic one-sided Jacobi, pairs (0,1), (0,2), (1,2), rotation skipped below 1e-30 relative; counts sweeps incl. the last (checking) one
    G = Fs.copy()  # rows of G = columns of F as stored (G[3p+i])
    n = len(G); cnt = np.zeros(n, int); active = np.ones(n, bool)
    for sweep in range(30):
        rot = np.zeros(n, bool)
        for p, q in ((0, 1), (0, 2), (1, 2)):
            gp, gq = G[:, p, :], G[:, q, :]
            al = (gp * gp).sum(1); be = (gq * gq).sum(1); ga = (gp * gq).sum(1)
            do = active & (ga != 0) & (ga * ga > 1e-30 * al * be)
            rot |= do
            with np.errstate(all="ignore"):
                zeta = (be - al) / (2 * ga)
                t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta))
            t[zeta == 0] = 1.0
            c = 1 / np.sqrt(1 + t * t); s = c * t
            c = np.where(do, c, 1.0); s = np.where(do, s, 0.0)
            ngp = c[:, None] * gp - s[:, None] * gq; ngq = s[:, None] * gp + c[:, None] * gq
            G[:, p, :], G[:, q, :] = ngp, ngq
        cnt[active] += 1
        active &= rot
        if not active.any(): break
    return cnt
def grads(V, T, X):
    Dm = np.stack([V[T[:, i]] - V[T[:, 0]] for i in (1, 2, 3)], 2)
    Ds = np.stack([X[T[:, i]] - X[T[:, 0]] for i in (1, 2, 3)], 2)
    F = Ds @ np.linalg.inv(Dm)
    return np.transpose(F, (0, 2, 1)).copy()  # [t][column][row]
V, T = scene.make_mat(150)
for name, X in (("rest + 1e-3 jitter", scene.jitter(V, T, rel=1e-3)), ("twisted 0.15 rad / unit + 2e-2 jitter", scene.twist_state(scene.jitter(V, T, rel=2e-2), 0.15)), ("twisted 0.6 rad / unit", scene.twist_state(V, 0.6))):
    c = sweeps(grads(V, T, X))
    w = c[: len(c) // 64 * 64].reshape(-1, 64).max(1)
    print(f"{name:42s} sweeps per element: mean {c.mean():.2f}  histogram {np.bincount(c)[1:].tolist()}   per wave of 64 consecutive elements (max): mean {w.mean():.2f} histogram {np.bincount(w)[1:].tolist()}")
