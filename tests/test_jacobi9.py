"""ipc_amd/csrc/jacobi9_device.h -- the register-resident cyclic Jacobi behind the barrier Hessian's PSD projection -- compiled for the
host and checked against numpy: B+ = V max(lambda, 0) V^T of a symmetric block that annihilates the rigid translations
(IglUtils::makePD, IglUtils.hpp:119-137, on the blocks of SelfCollisionHandler.cpp:418-561)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def j9():
    out = os.path.join(HERE, "jacobi9", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libj9.so")
    src = os.path.join(HERE, "jacobi9", "j9_host.cpp")
    hdr = os.path.join(HERE, "..", "ipc_amd", "csrc", "jacobi9_device.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.j9_make_pd.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.j9_make_pd.restype = ctypes.c_int
    return lib


def stencil_block(rng, nn, indefinite=True):
    """a symmetric 3 nn x 3 nn block with B t = 0 for the three translations, embedded in 12 x 12"""
    n = 3 * nn
    X = rng.normal(size=(n, n))
    S = X + X.T if indefinite else X @ X.T
    T = np.kron(np.ones((nn, 1)) / np.sqrt(nn), np.eye(3))
    P = np.eye(n) - T @ T.T
    S = P @ S @ P
    S = 0.5 * (S + S.T)
    B = np.zeros((12, 12))
    B[:n, :n] = S
    return B


def project(B):
    w, V = np.linalg.eigh(B)
    return (V * np.maximum(w, 0.0)) @ V.T


@pytest.mark.parametrize("nn", [2, 3, 4])
def test_projection_matches_eigh(j9, nn):
    rng = np.random.default_rng(10 + nn)
    for trial in range(50):
        B = stencil_block(rng, nn)
        A = np.asfortranarray(B.copy())
        sweeps = j9.j9_make_pd(nn, A.ctypes.data)
        assert 1 <= sweeps < 20
        want = project(B)
        assert np.abs(A - want).max() <= 1e-12 * np.abs(B).max()
        assert np.abs(A - A.T).max() <= 1e-13 * np.abs(B).max()
        assert np.linalg.eigvalsh(A).min() >= -1e-12 * np.abs(B).max()


@pytest.mark.parametrize("nn", [2, 3, 4])
def test_semidefinite_blocks_stay_untouched(j9, nn):
    rng = np.random.default_rng(20 + nn)
    for trial in range(20):
        B = stencil_block(rng, nn, indefinite=False)
        A = np.asfortranarray(B.copy())
        j9.j9_make_pd(nn, A.ctypes.data)
        # eigenvalues of the reduced block are >= -round-off: either untouched bit for bit, or projected by a round-off amount
        assert np.abs(A - B).max() <= 1e-12 * np.abs(B).max()


def test_barrier_shaped_block(j9):
    """what the kernel feeds it: cf (Hb g g^T + gb H) with gb < 0 -- one large positive eigenvalue, a few negative ones"""
    rng = np.random.default_rng(5)
    for trial in range(50):
        nn = 4
        g = rng.normal(size=12)
        g -= np.kron(np.ones(4), g.reshape(4, 3).mean(axis=0))
        H = stencil_block(rng, nn)
        B = 1e6 * np.outer(g, g) - 1e2 * H
        B = 0.5 * (B + B.T)
        A = np.asfortranarray(B.copy())
        j9.j9_make_pd(nn, A.ctypes.data)
        assert np.abs(A - project(B)).max() <= 1e-11 * np.abs(B).max()


def test_zero_block(j9):
    A = np.zeros((12, 12), order="F")
    assert j9.j9_make_pd(4, A.ctypes.data) == 0
    assert not A.any()
