"""ipc_amd/csrc/stencil_hessian_device.h -- the barrier Hessian of one contact stencil formed and PSD-projected in the complement of the rigid
translations -- compiled for the host and checked against the ORACLE's node-space derivatives (orc_contact.cpp: stencil_distance, barrier,
cross_sqnorm, mollifier) put together as the reference does (SelfCollisionHandler.cpp:418-561 active stencils, :3039-3201 mollified ones), with
numpy's eigh as IglUtils::makePD (IglUtils.hpp:119-137)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
K_PP, K_PE, K_PT, K_EE = 0, 1, 2, 3
NN = {K_PP: 2, K_PE: 3, K_PT: 4, K_EE: 4}


@pytest.fixture(scope="module")
def shl():
    out = os.path.join(HERE, "stencil_hessian", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libsh.so")
    src = os.path.join(HERE, "stencil_hessian", "sh_host.cpp")
    hdrs = [os.path.join(HERE, "..", "ipc_amd", "csrc", h) for h in ("stencil_hessian_device.h", "jacobi9_device.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.sh_active.restype = ctypes.c_double
    lib.sh_active.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    lib.sh_active9.restype = ctypes.c_double
    lib.sh_active9.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    lib.sh_para.restype = None
    lib.sh_para.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    return lib


def project(B):
    w, V = np.linalg.eigh(0.5 * (B + B.T))
    return (V * np.maximum(w, 0.0)) @ V.T


def stencil(rng, kind, gap):
    """node positions of a stencil of `kind` whose closest features are `gap` apart (roughly), unit-size primitives"""
    X = np.zeros((4, 3))
    if kind == K_PP:
        X[0] = rng.normal(size=3)
        X[1] = X[0] + gap * rng.normal(size=3)
    elif kind == K_PE:
        X[1], X[2] = rng.normal(size=3), rng.normal(size=3)
        t = rng.uniform(0.2, 0.8)
        n = np.cross(X[2] - X[1], rng.normal(size=3))
        X[0] = X[1] + t * (X[2] - X[1]) + gap * n / np.linalg.norm(n)
    elif kind == K_PT:
        X[1], X[2], X[3] = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
        w = rng.dirichlet([2, 2, 2])
        n = np.cross(X[2] - X[1], X[3] - X[1])
        X[0] = w @ X[1:4] + gap * n / np.linalg.norm(n) + 0.3 * rng.normal(size=3) * gap
    else:
        X[0], X[1] = rng.normal(size=3), rng.normal(size=3)
        d2 = rng.normal(size=3)
        n = np.cross(X[1] - X[0], d2)
        mid = X[0] + rng.uniform(0.2, 0.8) * (X[1] - X[0]) + gap * n / np.linalg.norm(n)
        s = rng.uniform(0.2, 0.8)
        X[2], X[3] = mid - s * d2, mid + (1 - s) * d2
    return X


@pytest.mark.parametrize("kind", [K_PP, K_PE, K_PT, K_EE])
def test_active_stencil_block_matches_the_oracle(shl, orc, kind):
    rng = np.random.default_rng(40 + kind)
    n3 = 3 * NN[kind]
    projected = 0
    for trial in range(60):
        gap = 10.0 ** rng.uniform(-4, -1)
        X = stencil(rng, kind, gap)
        d, g, H = orc.stencil_distance(kind, X)
        dHat = d * 10.0 ** rng.uniform(0.05, 2.0)
        kappa, mult = 10.0 ** rng.uniform(0, 4), float(rng.integers(1, 4))
        b, gb, Hb = orc.barrier(d, dHat)
        B = kappa * mult * (Hb * np.outer(g, g) + gb * H)
        B[n3:, :] = 0.0
        B[:, n3:] = 0.0
        scale = np.abs(B).max()
        A = np.zeros((12, 12), order="F")
        Xc = np.ascontiguousarray(X)
        dg = shl.sh_active(kind, Xc.ctypes.data, dHat, kappa * mult, 0, A.ctypes.data)
        assert abs(dg - d) <= 1e-13 * d
        assert np.abs(A - B).max() <= 1e-11 * scale, (kind, trial, np.abs(A - B).max() / scale)  # the unprojected block: same polynomial, other grouping
        shl.sh_active(kind, Xc.ctypes.data, dHat, kappa * mult, 1, A.ctypes.data)
        want = project(B)
        assert np.abs(A - want).max() <= 1e-11 * scale, (kind, trial, np.abs(A - want).max() / scale)
        assert np.abs(A - A.T).max() <= 1e-13 * scale
        A9 = np.zeros((12, 12), order="F")  # the same through the 9 x 9 frame the device kernel iterates on for every kind
        shl.sh_active9(kind, Xc.ctypes.data, dHat, kappa * mult, A9.ctypes.data)
        assert np.abs(A9 - A).max() <= 1e-13 * scale
        Ao = np.zeros((12, 12))
        Ao[:n3, :n3] = orc.make_pd(B[:n3, :n3])  # the oracle's own makePD (what the GPU path is compared with at full size)
        assert np.abs(A - Ao).max() <= 1e-9 * scale
        projected += np.linalg.eigvalsh(B).min() < -1e-9 * scale
    assert projected > 30  # the blocks are indefinite as a rule: the projection is exercised


@pytest.mark.parametrize("kind", [K_PP, K_PE, K_EE])
def test_mollified_stencil_block_matches_the_oracle(shl, orc, kind):
    """the edge pair's four nodes (a0, a1, b0, b1) nearly parallel; the distance stencil is a sub-stencil of them (dType_EE cases: PP a_i b_j, PE a_i (b0 b1) or
    b_j (a0 a1), EE all four)"""
    rng = np.random.default_rng(70 + kind)
    subs = {K_PP: [(0, 2), (0, 3), (1, 2), (1, 3)], K_PE: [(0, 2, 3), (1, 2, 3), (2, 0, 1), (3, 0, 1)], K_EE: [(0, 1, 2, 3)]}[kind]
    for trial in range(60):
        XE = np.zeros((4, 3))
        XE[0], dirn = rng.normal(size=3), rng.normal(size=3)
        XE[1] = XE[0] + dirn
        off = np.cross(dirn, rng.normal(size=3))
        off *= 10.0 ** rng.uniform(-3, -1) / np.linalg.norm(off)
        tilt = 10.0 ** rng.uniform(-3, -1) * rng.normal(size=3)
        XE[2] = XE[0] + rng.uniform(-0.3, 0.6) * dirn + off
        XE[3] = XE[2] + rng.uniform(0.5, 1.2) * dirn + tilt
        sub = subs[trial % len(subs)]
        Xs = np.zeros((4, 3))
        Xs[:len(sub)] = XE[list(sub)]
        d, gS, HS = orc.stencil_distance(kind, Xs)
        dHat = d * 10.0 ** rng.uniform(0.05, 2.0)
        kappa = 10.0 ** rng.uniform(0, 4)
        b, gb, Hb = orc.barrier(d, dHat)
        c, cg, Q = orc.cross_sqnorm(XE)
        eps_x = c * 10.0 ** rng.uniform(-0.5, 1.5)  # both sides of the threshold
        e, eg, eH = orc.mollifier(c, eps_x)
        # distance derivatives mapped onto the four edge nodes (SelfCollisionHandler.cpp:3105-3160)
        gd, W = np.zeros(12), np.zeros((12, 12))
        for k, q in enumerate(sub):
            gd[3 * q:3 * q + 3] = gS[3 * k:3 * k + 3]
            for l, r in enumerate(sub):
                W[3 * q:3 * q + 3, 3 * r:3 * r + 3] = HS[3 * k:3 * k + 3, 3 * l:3 * l + 3]
        B = kappa * (gb * eg * (np.outer(gd, cg) + np.outer(cg, gd)) + b * (eg * Q + eH * np.outer(cg, cg)) + e * Hb * np.outer(gd, gd) + e * gb * W)
        scale = np.abs(B).max()
        sel = np.zeros((4, 4))
        for k, q in enumerate(sub):
            sel[k, q] = 1.0
        A = np.zeros((12, 12), order="F")
        shl.sh_para(kind, np.ascontiguousarray(XE).ctypes.data, sel.ctypes.data, dHat, kappa, eps_x, 0, A.ctypes.data)
        assert np.abs(A - B).max() <= 1e-10 * scale, (kind, trial, np.abs(A - B).max() / scale)
        shl.sh_para(kind, np.ascontiguousarray(XE).ctypes.data, sel.ctypes.data, dHat, kappa, eps_x, 1, A.ctypes.data)
        assert np.abs(A - project(B)).max() <= 1e-10 * scale
        assert np.abs(A - orc.make_pd(B)).max() <= 1e-9 * scale
