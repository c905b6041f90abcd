"""Bring-up tests of the two-panels-per-launch step kernel (`k_big_step2`, `IPCGPU_MF_STEP2=1`: written and emulated on the host at the end of
round 4, never run on a GPU; tools/emulation/step2_*.py).  Skipped unless IPCGPU_TEST_STEP2=1, so that the regular `-m gpu`
run does not depend on a kernel that is off by default; `tools/gpu_step2_bringup.sh` is the first GPU call of the next round.

The switch is read at every symbolic analysis (MfNumeric::setup), so one process factorises the same matrix both ways: the solutions must agree to
round-off and both must solve the system.  The sizes are chosen for their front shapes: widths of the top separators that are and are not multiples
of 64 (partial last pairs, single-panel fronts), fronts with and without an explicit inverse, and the contact pattern of a stack of sheets."""
import os

import numpy as np
import pytest

from ipc_amd import scene

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("IPCGPU_TEST_STEP2") != "1", reason="k_big_step2 is off by default; set IPCGPU_TEST_STEP2=1")]


def factor_and_solve(gpu_lib, V, F, Vt, dbc, step2, border):
    os.environ["IPCGPU_MF_STEP2"] = "1" if step2 else "0"
    os.environ["IPCGPU_MF_XINV_BORDER"] = "1" if border else "0"
    try:
        c = gpu_lib.Context(0)
        c.set_mesh(V, F, YM=2e4, PR=0.4, density=1000.0)
        c.opt_init(0.04, False)
        c.set_dbc(dbc, 2)
        c.set_positions(Vt)
        c.set_pattern()
        c.assemble_newton(0.04 ** 2, True, with_gradient=False)
        c.analyze_pattern()
        assert c.factorize()
        rows, _ = c.get_dims()
        out = []
        for seed in (1, 2):
            b = np.random.default_rng(seed).normal(size=rows)
            x = c.solve(b)
            out.append((x, np.linalg.norm(c.multiply(x) - b) / np.linalg.norm(b)))
        # not positive definite -> reported, in both modes
        a = c.get_a()
        ia, _ = c.get_pattern()
        k = ia[3 * (rows // 6)]
        c.set_coeff(3 * (rows // 6), 3 * (rows // 6), -abs(a[k]))
        bad = c.factorize()
        c.close()
        return out, bad
    finally:
        os.environ.pop("IPCGPU_MF_STEP2", None)
        os.environ.pop("IPCGPU_MF_XINV_BORDER", None)


@pytest.mark.parametrize("n", [24, 43, 64, 87, 150])
@pytest.mark.parametrize("border", [False, True])
def test_two_panel_steps_give_the_same_solution(gpu_lib, n, border):
    V, F = scene.make_mat(n)
    Vt = scene.twist_state(scene.jitter(V, F), 0.5)
    left, right = scene.border_verts(V, 0.01)
    dbc = np.concatenate([left, right])
    ref, bad0 = factor_and_solve(gpu_lib, V, F, Vt, dbc, step2=False, border=border)
    new, bad1 = factor_and_solve(gpu_lib, V, F, Vt, dbc, step2=True, border=border)
    assert not bad0 and not bad1, "a matrix with a negative diagonal entry must be reported as not positive definite"
    for (x0, r0), (x1, r1) in zip(ref, new):
        assert r0 < 1e-10 and r1 < 1e-10, (r0, r1)
        assert np.abs(x1 - x0).max() <= 1e-9 * np.abs(x0).max()


def test_two_panel_steps_on_a_block_shaped_mesh(gpu_lib):
    """a 3D block: wide separators (hundreds of columns) with long struct lists -- several role-B' workgroups per front and trailing tiles in both directions"""
    V, F = scene.make_box(14, 12, 10, size=(1.4, 1.2, 1.0), origin=(0, 0, 0))
    Vt = scene.jitter(V, F, rel=2e-2)
    dbc = np.where(V[:, 0] < 1e-9)[0].astype(np.int32)
    for border in (False, True):
        ref, _ = factor_and_solve(gpu_lib, V, F, Vt, dbc, step2=False, border=border)
        new, _ = factor_and_solve(gpu_lib, V, F, Vt, dbc, step2=True, border=border)
        for (x0, r0), (x1, r1) in zip(ref, new):
            assert r0 < 1e-10 and r1 < 1e-10, (r0, r1)
            assert np.abs(x1 - x0).max() <= 1e-9 * np.abs(x0).max()
