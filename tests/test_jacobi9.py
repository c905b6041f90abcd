"""ipc_amd/csrc/jacobi9_device.h -- the register-resident cyclic Jacobi behind the barrier Hessian's PSD projection -- compiled for the
host (through tests/stencil_hessian/sh_host.cpp) and checked against numpy: C+ = V max(lambda, 0) V^T of a symmetric M x M matrix, M = 3, 6, 9 =
the reduced blocks of point-point, point-edge and point-triangle / edge-edge stencils (IglUtils::makePD, IglUtils.hpp:119-137, on the
blocks of SelfCollisionHandler.cpp:418-561)."""
import ctypes

import numpy as np
import pytest

from test_stencil_hessian import shl  # noqa: F401  (the fixture that builds the host library)


def project(B):
    w, V = np.linalg.eigh(B)
    return (V * np.maximum(w, 0.0)) @ V.T


@pytest.mark.parametrize("M", [3, 6, 9])
def test_projection_matches_eigh(shl, M):  # noqa: F811
    shl.sh_project_psd.restype = ctypes.c_int
    shl.sh_project_psd.argtypes = [ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(10 + M)
    for trial in range(60):
        X = rng.normal(size=(M, M))
        B = X + X.T if trial % 3 else X @ X.T  # indefinite as a rule, positive definite every third time (then the block comes back as it was)
        if trial % 7 == 0:  # rank-deficient with a dominant direction, like b'' g g^T + b' H at a small distance
            g = rng.normal(size=M)
            B = 1e6 * np.outer(g, g) + B
        A = np.asfortranarray(B.copy())
        sweeps = shl.sh_project_psd(M, A.ctypes.data)
        assert 1 <= sweeps < 20
        assert np.abs(A - project(B)).max() <= 1e-12 * np.abs(B).max()
        assert np.abs(A - A.T).max() <= 1e-13 * np.abs(B).max()
        assert np.linalg.eigvalsh(A).min() >= -1e-12 * np.abs(B).max()


def test_diagonal_input_takes_no_sweep(shl):  # noqa: F811
    shl.sh_project_psd.restype = ctypes.c_int
    shl.sh_project_psd.argtypes = [ctypes.c_int, ctypes.c_void_p]
    A = np.asfortranarray(np.diag([3.0, -2.0, 1.0, 0.0, -5.0, 4.0, 2.0, -1.0, 6.0]))
    assert shl.sh_project_psd(9, A.ctypes.data) == 0
    assert np.array_equal(A, np.diag([3.0, 0.0, 1.0, 0.0, 0.0, 4.0, 2.0, 0.0, 6.0]))
