"""Pins the CPU oracle (oracle/) against everything the reference holds for the hot path
 and against independent numerics (LAPACK, finite differences, dense Cholesky).

Known answers from the reference:
  * psi(I) = 0                          NeoHookeanEnergy::checkEnergyVal, NeoHookeanEnergy.cpp:156-170
  * 10 isolated nodes, diag 10, rhs 1   Diagnostic.cpp:367-392  ->  x = 0.1
  * finite-difference recipes           Energy::unitTest_* , Energy.cpp:584-893 (h = 1e-6, YM = 100, nu = 0.4)
  * symmetric-upper CSR layout          LinSysSolver.hpp:46-150 (row lengths 3+3k, 2+3k, 1+3k)
"""
import numpy as np
import pytest

from ipc_amd import scene

YM, PR = 100.0, 0.4
MU = YM / 2 / (1 + PR)
LAM = YM * PR / (1 + PR) / (1 - 2 * PR)


def test_svd_convention_against_lapack(orc):
    rng = np.random.default_rng(1)
    for i in range(300):
        F = rng.normal(size=(3, 3))
        if i % 3 == 0:
            F = np.eye(3) + 1e-3 * F  # near-rest: clustered singular values
        U, S, V = orc.svd3(F)
        assert np.allclose(U @ np.diag(S) @ V.T, F, atol=1e-13)
        assert abs(np.linalg.det(U) - 1) < 1e-12 and abs(np.linalg.det(V) - 1) < 1e-12
        assert abs(S[0]) >= abs(S[1]) >= abs(S[2]) and S[0] >= 0 and S[1] >= 0
        assert (S[2] < 0) == (np.linalg.det(F) < 0)
        assert np.allclose(np.abs(S), np.linalg.svd(F, compute_uv=False), rtol=1e-12, atol=1e-14)


def test_psi_identity_is_zero(orc):
    # NeoHookeanEnergy.cpp:156-170
    assert orc.nh_energy_sigma(np.ones(3), MU, LAM) == 0.0


def _psi(orc, F):
    s = np.linalg.svd(F, compute_uv=False)
    return orc.nh_energy_sigma(s, MU, LAM)


def test_dE_div_dF_finite_difference(orc):
    # Energy::unitTest_dE_div_dF recipe (Energy.cpp:749-815): central differences instead of forward ones
    rng = np.random.default_rng(2)
    h = 1e-6
    for _ in range(10):
        F = np.eye(3) + 0.25 * rng.normal(size=(3, 3))
        if np.linalg.det(F) <= 0.1:
            continue
        P = orc.nh_P(F, MU, LAM)
        Pfd = np.zeros((3, 3))
        for i in range(3):
            for j in range(3):
                Fp, Fm = F.copy(), F.copy()
                Fp[i, j] += h
                Fm[i, j] -= h
                Pfd[i, j] = (_psi(orc, Fp) - _psi(orc, Fm)) / (2 * h)
        assert np.abs(P - Pfd).max() <= 1e-7 * max(1.0, np.abs(P).max())


def test_dP_div_dF_finite_difference_and_psd(orc):
    # Energy::unitTest_dP_div_dF recipe (Energy.cpp:817-893); index 3*i+j (Energy.cpp:535-561)
    rng = np.random.default_rng(3)
    h = 1e-6
    for _ in range(10):
        F = np.eye(3) + 0.25 * rng.normal(size=(3, 3))
        if np.linalg.det(F) <= 0.1:
            continue
        D = orc.nh_dPdF(F, MU, LAM, 1.0, False)
        Dfd = np.zeros((9, 9))
        for r in range(3):
            for s in range(3):
                Fp, Fm = F.copy(), F.copy()
                Fp[r, s] += h
                Fm[r, s] -= h
                Dfd[:, 3 * r + s] = ((orc.nh_P(Fp, MU, LAM) - orc.nh_P(Fm, MU, LAM)) / (2 * h)).reshape(-1)
        assert np.abs(D - Dfd).max() <= 1e-6 * np.abs(D).max()
        assert np.allclose(D, D.T, atol=1e-12 * np.abs(D).max())
        Dp = orc.nh_dPdF(F, MU, LAM, 1.0, True)
        w = np.linalg.eigvalsh(Dp)
        assert w.min() >= -1e-10 * max(1.0, w.max())


def test_make_pd_is_eigen_clamp(orc):
    rng = np.random.default_rng(4)
    for n in (3, 6, 9, 12):
        A = rng.normal(size=(n, n))
        A = A + A.T
        w, Q = np.linalg.eigh(A)
        ref = (Q * np.maximum(w, 0)) @ Q.T
        assert np.allclose(orc.make_pd(A), ref, atol=1e-11 * np.abs(A).max())
        B = A @ A.T + np.eye(n)  # already PD: returned untouched (IglUtils.hpp:122-124)
        assert np.array_equal(orc.make_pd(B), B)


def _bar(orc, twist=0.3):
    V, F = scene.make_bar(8, 2, 2, size=(4.0, 0.5, 1.0))
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000)
    Vt = scene.twist_state(scene.jitter(V, F), twist)
    m.set_V(Vt)
    return V, F, Vt, m


def test_mesh_features(orc):
    V, F, Vt, m = _bar(orc)
    f = m.features()
    assert np.all(f["triArea"] > 0)
    assert abs(f["triArea"].sum() - 4.0 * 0.5 * 1.0) < 1e-12  # rest volume
    assert abs(f["mass"].sum() - 1000 * 2.0) < 1e-9  # Mesh.cpp:255-266, 399
    # restTriInv really is the inverse of the rest edge matrix (Mesh.cpp:440-449)
    for t in (0, 7, F.shape[0] - 1):
        X0 = np.stack([V[F[t, 1]] - V[F[t, 0]], V[F[t, 2]] - V[F[t, 0]], V[F[t, 3]] - V[F[t, 0]]], axis=1)
        A = f["restTriInv"][t].reshape(3, 3, order="F")
        assert np.allclose(A @ X0, np.eye(3), atol=1e-12)
    m.set_V(V)
    assert m.elastic_energy() == 0.0  # rest state: F = I exactly


def test_gradient_matches_energy_fd(orc):
    V, F, Vt, m = _bar(orc)
    g = m.elastic_gradient(projectDBC=False)
    rng = np.random.default_rng(5)
    for k in rng.integers(0, 3 * V.shape[0], 12):
        v, c = divmod(int(k), 3)
        Vp = Vt.copy()
        Vp[v, c] += 1e-6
        m.set_V(Vp)
        Ep = m.elastic_energy()
        Vp[v, c] -= 2e-6
        m.set_V(Vp)
        Em = m.elastic_energy()
        assert abs((Ep - Em) / 2e-6 - g[k]) <= 1e-6 * max(1.0, abs(g[k]))
    m.set_V(Vt)


def test_unprojected_element_hessian_matches_gradient_fd(orc):
    V, F, Vt, m = _bar(orc, twist=0.05)
    t = 11
    H = m.elastic_hessian_elem(t, 1.0, projectSPD=False)
    assert np.allclose(H, H.T, atol=1e-9 * np.abs(H).max())
    # FD of the element's own gradient contribution: perturb each of its 12 dofs, look at the energy of that element
    eps = 1e-6
    Hfd = np.zeros((12, 12))
    for a in range(4):
        for c in range(3):
            def elem_grad(Vc):
                m.set_V(Vc)
                _, pe = m.elastic_energy(per_elem=True)
                return pe[t]
            # second differences of the element energy
            for b in range(4):
                for d in range(3):
                    Vpp, Vpm, Vmp, Vmm = (Vt.copy() for _ in range(4))
                    Vpp[F[t, a], c] += eps; Vpp[F[t, b], d] += eps
                    Vpm[F[t, a], c] += eps; Vpm[F[t, b], d] -= eps
                    Vmp[F[t, a], c] -= eps; Vmp[F[t, b], d] += eps
                    Vmm[F[t, a], c] -= eps; Vmm[F[t, b], d] -= eps
                    Hfd[3 * a + c, 3 * b + d] = (elem_grad(Vpp) - elem_grad(Vpm) - elem_grad(Vmp) + elem_grad(Vmm)) / (4 * eps * eps)
    m.set_V(Vt)
    assert np.abs(H - Hfd).max() <= 2e-3 * np.abs(H).max()


def test_csr_pattern_layout(orc):
    # LinSysSolver.hpp:46-150: symmetric upper, 3x3 node blocks, rows 3v,3v+1,3v+2 have 3+3k, 2+3k, 1+3k entries
    V, F, Vt, m = _bar(orc)
    ia, ja = m.pattern()
    nV = V.shape[0]
    nb = [set() for _ in range(nV)]
    for t in F:
        for a in t:
            for b in t:
                if a != b:
                    nb[a].add(int(b))
    assert ia[0] == 0 and len(ia) == 3 * nV + 1
    for v in range(nV):
        up = sorted(n for n in nb[v] if n > v)
        k = len(up)
        for r in range(3):
            row = ja[ia[3 * v + r]:ia[3 * v + r + 1]]
            assert len(row) == 3 - r + 3 * k
            expect = [3 * v + c for c in range(r, 3)] + [3 * n + c for n in up for c in range(3)]
            assert list(row) == expect
    nE = sum(len(s) for s in nb) // 2
    assert len(ja) == 6 * nV + 9 * nE


def test_assembled_hessian_is_spd_and_respects_dbc(orc):
    V, F, Vt, m = _bar(orc)
    left, right = scene.border_verts(V, 0.01)
    m.set_dbc(np.concatenate([left, right]), 2)
    ia, ja = m.pattern()
    a = m.assemble_hessian(len(ja), coef=1e-3, projectDBC=True)
    n = len(ia) - 1
    A = np.zeros((n, n))
    for r in range(n):
        for k in range(ia[r], ia[r + 1]):
            A[r, ja[k]] = a[k]
            A[ja[k], r] = a[k]
    assert np.linalg.eigvalsh(A).min() > 0
    for v in np.concatenate([left, right]):
        for d in range(3):
            r = 3 * v + d
            e = np.zeros(n)
            e[r] = 1.0
            assert np.array_equal(A[r], e)  # identity row / column (IglUtils.hpp:45-53, Optimizer.cpp:3654-3663)
    x = np.random.default_rng(6).normal(size=n)
    assert np.allclose(m.symv(a, x), A @ x, rtol=1e-12, atol=1e-9)


def test_diagnostic_linear_solve_known_answer(orc):
    # Diagnostic.cpp:367-392: 10 isolated nodes, every diagonal 10, rhs 1  =>  x = 0.1
    n = 30
    ia = np.arange(n + 1, dtype=np.int32)
    # isolated nodes still carry their 3x3 upper block in the reference layout
    rows = []
    ja = []
    ptr = [0]
    for v in range(10):
        for r in range(3):
            cols = [3 * v + c for c in range(r, 3)]
            ja += cols
            ptr.append(len(ja))
    ia = np.array(ptr, dtype=np.int32)
    ja = np.array(ja, dtype=np.int32)
    a = np.array([10.0 if ja[k] == r else 0.0 for r in range(n) for k in range(ia[r], ia[r + 1])])
    ch = orc.Chol(ia, ja, 2)
    assert ch.factorize(a)
    x = ch.solve(np.ones(n))
    assert np.allclose(x, 0.1, rtol=0, atol=1e-15)


def test_cholesky_against_dense_and_not_pd(orc):
    V, F, Vt, m = _bar(orc)
    ia, ja = m.pattern()
    a = m.assemble_hessian(len(ja), coef=1e-3, projectDBC=True)
    n = len(ia) - 1
    A = np.zeros((n, n))
    for r in range(n):
        for k in range(ia[r], ia[r + 1]):
            A[r, ja[k]] = a[k]
            A[ja[k], r] = a[k]
    b = np.random.default_rng(7).normal(size=n)
    ch = orc.Chol(ia, ja, 4)
    assert ch.factorize(a)
    x = ch.solve(b)
    assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # indefinite matrix -> factorize() == false (CHOLMODSolver.cpp:130-137)
    a2 = a.copy()
    a2[ia[3 * 5]] = -1.0
    assert not ch.factorize(a2)


def test_inversion_step_bound(orc):
    # get_feasible_steps.cpp:75-172: step at which an element reaches (1 - 0.8) of its volume... slackness 0.2
    V, F = scene.make_bar(2, 1, 1, size=(2.0, 1.0, 1.0))
    m = orc.Mesh(V, F)
    p = np.zeros((V.shape[0], 3))
    right = V[:, 0] > 0.99
    p[right, 0] = -1.0  # squash the right cube flat in one unit step
    out = m.inversion_step(p.reshape(-1), 0.2)
    # volume of right-cube tets scales as (1 - t); root of (1 - t) = 0.2  ->  t = 0.8
    touched = np.array([right[F[t]].any() and not right[F[t]].all() for t in range(F.shape[0])])
    assert np.allclose(out[touched], 0.8, rtol=1e-9)
    assert np.all(out[~touched] == 1e20)
    assert abs(m.filter_step_size(p.reshape(-1), 1.0) - 0.8) < 1e-9


def test_newton_twist_bar_converges_and_decreases_energy(orc):
    V, F = scene.make_bar(10, 2, 2, size=(5.0, 0.5, 1.0))
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000)
    left, right = scene.border_verts(V, 0.01)
    o = orc.Optimizer(m, dt=0.025, gravity=False, nthreads=4)
    o.set_twist(left, right, 0.4 * np.pi)
    o.precompute()
    for step in range(3):
        o.begin_timestep()
        E_prev = o.state()["E"]
        for it in range(50):
            if o.newton_iter():
                break
            s = o.state()
            assert s["E"] <= E_prev + 1e-12 * abs(E_prev)
            E_prev = s["E"]
        else:
            pytest.fail("Newton did not converge")
        o.end_timestep()
        assert m.check_inversion()
    s = o.state()
    assert s["timestep"] == 3 and s["innerIterAmt"] >= 3
    # handles followed the scripted rotation exactly: 3 steps of 0.4 pi * 0.025 rad
    ang = 3 * 0.4 * np.pi * 0.025
    Vn = s["V"]
    c = 0.5 * (V.min(0) + V.max(0))
    for v in right[:5]:
        y, z = V[v, 1] - c[1], V[v, 2] - c[2]
        assert abs(Vn[v, 1] - (c[1] + np.cos(ang) * y - np.sin(ang) * z)) < 1e-12
        assert abs(Vn[v, 2] - (c[2] + np.sin(ang) * y + np.cos(ang) * z)) < 1e-12


def test_config0_golden_fixture_is_what_the_oracle_computes(orc):
    """tests/golden/config0_bar2523.npz (tools/make_golden_config0.py): the reference's hello-world scene on its own mesh."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "config0_bar2523.npz"))
    m = orc.Mesh(g["V"], g["T"], YM=1e9, PR=0.4, density=1000.0)
    o = orc.Optimizer(m, dt=0.025, gravity=True, nthreads=4)
    orc.opt_add_dirichlet(o, g["left"])
    orc.opt_add_dirichlet(o, g["right"], ang_vel_deg=(270, 0, 0))
    o.set_rel_tol(float(g["rel_tol"]))
    o.precompute()
    for step in range(2):
        assert o.solve_timestep(100) == g["iters"][step]
        assert np.abs(o.state()["V"] - g["positions"][step]).max() <= 1e-10 * np.abs(g["positions"][step]).max()
    # Dirichlet groups with a time range: outside of it the nodes are free again (AnimScripter.cpp:98-107)
    V, F = scene.make_box(3, 1, 1, size=(3.0, 1.0, 1.0))
    SF = scene.surface_tris(F)
    m = orc.Mesh(V, F, YM=1e5, PR=0.3, density=1000.0)
    o = orc.Optimizer(m, dt=0.02, gravity=True, nthreads=2)
    sel = scene.select_dirichlet(V, SF, (0, 0, 0), (0.01, 1, 1))
    orc.opt_add_dirichlet(o, sel, lin_vel=(0.0, 0.5, 0.0), t0=0.0, t1=0.03)
    o.precompute()
    o.solve_timestep(60)
    assert np.allclose(o.state()["V"][sel, 1], V[sel, 1] + 0.5 * 0.02)
    o.solve_timestep(60)
    y2 = o.state()["V"][sel, 1].copy()
    assert np.allclose(y2, V[sel, 1] + 2 * 0.5 * 0.02)
    o.solve_timestep(60)  # stepStartTime = 0.04 >= t1: released, gravity takes over
    assert (o.state()["V"][sel, 1] < y2 + 0.5 * 0.02 - 1e-6).all()
