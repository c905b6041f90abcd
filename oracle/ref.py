"""TEST INFRASTRUCTURE -- ctypes binding of oracle/_ref/libipcref.so: the REFERENCE's own sources compiled from /root/reference
(oracle/Makefile.ref) behind the C entry points of oracle/ref_api.cpp.  Exists in the build container only; tests that use it
skip when the library is absent, and tools/make_golden_ref.py turns its outputs into committed fixtures (tests/golden/)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libipcref.so")


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        d, i, p = C.c_double, C.c_int, C.c_void_p
        sig = {
            "ipcref_mesh_create": (p, [i, p, i, p, i, p, i, p, p, d, d, d]),
            "ipcref_mesh_destroy": (None, [p]),
            "ipcref_mesh_set_positions": (None, [p, p]),
            "ipcref_mesh_set_dbc": (None, [p, i, p, p]),
            "ipcref_mesh_set_lame": (None, [p, d, d]),
            "ipcref_mesh_features": (None, [p, p, p, p, p, p]),
            "ipcref_mesh_scalars": (None, [p, p]),
            "ipcref_mesh_surface_counts": (None, [p, p]),
            "ipcref_mesh_get_surface": (None, [p, p, p]),
            "ipcref_elastic_energy": (d, [p, i, d]),
            "ipcref_elastic_gradient": (None, [p, i, d, i, p]),
            "ipcref_elastic_hessian": (i, [p, i, d, i, i, i, p, p, p, p, i]),
            "ipcref_filter_step_size": (d, [p, i, p, d]),
            "ipcref_svd3": (None, [p, p, p, p]),
            "ipcref_make_pd": (None, [i, p]),
            "ipcref_stencil_distance": (None, [i, p, p, p, p]),
            "ipcref_dtype_pt": (i, [p]),
            "ipcref_dtype_ee": (i, [p]),
            "ipcref_classified_distance": (d, [i, p]),
            "ipcref_barrier": (None, [d, d, p, p, p]),
            "ipcref_ee_mollifier": (None, [p, d, p, p, p, p, p, p]),
            "ipcref_seg_tri_intersect": (i, [p]),
            "ipcref_constraint_set": (None, [p, d, p]),
            "ipcref_constraint_get": (None, [p, p, p, p, p]),
            "ipcref_constraint_put": (None, [p, i, p, i, p, p]),
            "ipcref_barrier_energy": (d, [p, d, d]),
            "ipcref_barrier_gradient": (None, [p, d, d, p]),
            "ipcref_barrier_hessian": (i, [p, d, d, i, p, p, p, i]),
            "ipcref_full_ccd": (d, [p, p, d, d]),
            "ipcref_partial_ccd": (d, [p, p, d, d]),
            "ipcref_is_intersected": (i, [p]),
            "ipcref_halfspace_eval": (i, [p, p, p, d, d, i, p, p, p, p, p, p, i]),
            "ipcref_halfspace_step_bound": (d, [p, p, p, p, d, d]),
            "ipcref_halfspace_move": (d, [p, p, p, p, d, p]),
        }
        for name, (res, args) in sig.items():
            f = getattr(_lib, name)
            f.restype, f.argtypes = res, args
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Mesh:
    """IPC::Mesh<3> built by its own constructor (Mesh.cpp:40-110: features, barycentric lumped mass, surface bookkeeping)."""

    def __init__(self, V, T, SF, YM, PR, density, node_ranges=None, sf_ranges=None):
        self.V0, self.T, self.SF = _d(V), _i(T), _i(SF)
        self.nV, self.nT = self.V0.shape[0], self.T.shape[0]
        nr = _i(node_ranges if node_ranges is not None else [0, self.nV])
        sr = _i(sf_ranges if sf_ranges is not None else [0, self.SF.shape[0]])
        self.h = lib().ipcref_mesh_create(self.nV, _p(self.V0), self.nT, _p(self.T), self.SF.shape[0], _p(self.SF), len(nr) - 1, _p(nr), _p(sr), YM, PR, density)

    def close(self):
        if self.h:
            lib().ipcref_mesh_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_positions(self, V):
        V = _d(V)
        lib().ipcref_mesh_set_positions(self.h, _p(V))

    def set_dbc(self, ids, types):
        ids = _i(ids)
        types = _i(np.broadcast_to(types, ids.shape))
        lib().ipcref_mesh_set_dbc(self.h, len(ids), _p(ids), _p(types))

    def features(self):
        A, vol, m, mu, lam = np.zeros((self.nT, 3, 3)), np.zeros(self.nT), np.zeros(self.nV), np.zeros(self.nT), np.zeros(self.nT)
        lib().ipcref_mesh_features(self.h, _p(A), _p(vol), _p(m), _p(mu), _p(lam))
        s = np.zeros(3)
        lib().ipcref_mesh_scalars(self.h, _p(s))
        return dict(restTriInv=A, triArea=vol, mass=m, mu=mu, lam=lam, avgEdgeLen=s[0], bbox2=s[1], avgNodeMass=s[2])

    def surface(self):
        n = np.zeros(2, np.int32)
        lib().ipcref_mesh_surface_counts(self.h, _p(n))
        SVI, E = np.zeros(n[0], np.int32), np.zeros((n[1], 2), np.int32)
        lib().ipcref_mesh_get_surface(self.h, _p(SVI), _p(E))
        return SVI, E

    def elastic_energy(self, kind=0, coef=1.0):
        return lib().ipcref_elastic_energy(self.h, kind, coef)

    def elastic_gradient(self, kind=0, coef=1.0, project_dbc=True):
        g = np.zeros(3 * self.nV)
        lib().ipcref_elastic_gradient(self.h, kind, coef, int(project_dbc), _p(g))
        return g

    def _csr(self, call):
        cap = 9 * 40 * self.nV + 1024
        ia, ja, a = np.zeros(3 * self.nV + 1, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
        nnz = call(ia, ja, a, cap)
        if nnz < 0:
            cap = -nnz
            ja, a = np.zeros(cap, np.int32), np.zeros(cap)
            nnz = call(ia, ja, a, cap)
        return ia, ja[:nnz].copy(), a[:nnz].copy()

    def elastic_hessian(self, kind=0, coef=1.0, project_spd=True, project_dbc=True, extra_pairs=()):
        ep = _i(np.reshape(extra_pairs, (-1, 2)))
        return self._csr(lambda ia, ja, a, cap: lib().ipcref_elastic_hessian(self.h, kind, coef, int(project_spd), int(project_dbc), len(ep), _p(ep), _p(ia), _p(ja), _p(a), cap))

    def filter_step_size(self, p, step=1.0, kind=0):
        p = _d(p)
        return lib().ipcref_filter_step_size(self.h, kind, _p(p), step)

    def constraint_set(self, dHat):
        n = np.zeros(3, np.int32)
        lib().ipcref_constraint_set(self.h, dHat, _p(n))
        act, par, eiej, cs = np.zeros((n[0], 4), np.int32), np.zeros((n[1], 4), np.int32), np.zeros((n[1], 2), np.int32), np.zeros((n[2], 2), np.int32)
        lib().ipcref_constraint_get(self.h, _p(act), _p(par), _p(eiej), _p(cs))
        return act, par, eiej, cs

    def put_constraints(self, act, par, eiej):
        act, par, eiej = _i(np.reshape(act, (-1, 4))), _i(np.reshape(par, (-1, 4))), _i(np.reshape(eiej, (-1, 2)))
        lib().ipcref_constraint_put(self.h, len(act), _p(act), len(par), _p(par), _p(eiej))

    def barrier_energy(self, dHat, kappa):
        return lib().ipcref_barrier_energy(self.h, dHat, kappa)

    def barrier_gradient(self, dHat, kappa):
        g = np.zeros(3 * self.nV)
        lib().ipcref_barrier_gradient(self.h, dHat, kappa, _p(g))
        return g

    def barrier_hessian(self, dHat, kappa, project_dbc=True):
        return self._csr(lambda ia, ja, a, cap: lib().ipcref_barrier_hessian(self.h, dHat, kappa, int(project_dbc), _p(ia), _p(ja), _p(a), cap))

    def full_ccd(self, p, slackness=0.8, step=1.0):
        p = _d(p)
        return lib().ipcref_full_ccd(self.h, _p(p), slackness, step)

    def partial_ccd(self, p, slackness=0.8, step=1.0):
        p = _d(p)
        return lib().ipcref_partial_ccd(self.h, _p(p), slackness, step)

    def is_intersected(self):
        return bool(lib().ipcref_is_intersected(self.h))

    def halfspace(self, origin, normal, dHat, kappa, project_dbc=True):
        """HalfSpace<3>: active vertices, kappa * sum b, its gradient and PSD Hessian in the mesh pattern."""
        o, n = _d(origin), _d(normal)
        act, E, g = np.zeros(self.nV, np.int32), np.zeros(1), np.zeros(3 * self.nV)
        out = {}

        def call(ia, ja, a, cap):
            out["n"] = lib().ipcref_halfspace_eval(self.h, _p(o), _p(n), dHat, kappa, int(project_dbc), _p(act), _p(E), _p(g), _p(ia), _p(ja), _p(a), cap)
            return len(ja) if out["n"] >= 0 else out["n"]

        cap = 9 * 40 * self.nV + 1024
        ia, ja, a = np.zeros(3 * self.nV + 1, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
        nact = lib().ipcref_halfspace_eval(self.h, _p(o), _p(n), dHat, kappa, int(project_dbc), _p(act), _p(E), _p(g), _p(ia), _p(ja), _p(a), cap)
        nnz = ia[3 * self.nV]
        return act[:nact].copy(), E[0], g, (ia, ja[:nnz].copy(), a[:nnz].copy())

    def halfspace_step_bound(self, origin, normal, p, slackness=0.9, step=1.0):
        o, n, p = _d(origin), _d(normal), _d(p)
        return lib().ipcref_halfspace_step_bound(self.h, _p(o), _p(n), _p(p), slackness, step)


def halfspace_move(mesh, origin, normal, delta, slackness=0.5):
    """HalfSpace::move run by the reference: (origin after the move, fraction left)"""
    o, n, d, out = _d(origin), _d(normal), _d(delta), np.zeros(3)
    left = lib().ipcref_halfspace_move(mesh.h, _p(o), _p(n), _p(d), slackness, _p(out))
    return out, left


def stencil_distance(kind, X):
    n = (2, 3, 4, 4)[kind]
    X12 = np.zeros((4, 3))
    X12[:n] = _d(X)[:n]
    d, g, H = np.zeros(1), np.zeros(3 * n), np.zeros((3 * n, 3 * n))
    lib().ipcref_stencil_distance(kind, _p(X12), _p(d), _p(g), _p(H))
    return d[0], g, H


def dtype_pt(X):
    X = _d(X)
    return lib().ipcref_dtype_pt(_p(X))


def dtype_ee(X):
    X = _d(X)
    return lib().ipcref_dtype_ee(_p(X))


def classified_distance(kind, X):
    X = _d(X)
    return lib().ipcref_classified_distance(kind, _p(X))


def barrier(d, dHat):
    b, g, H = C.c_double(), C.c_double(), C.c_double()
    lib().ipcref_barrier(d, dHat, C.byref(b), C.byref(g), C.byref(H))
    return b.value, g.value, H.value


def ee_mollifier(X, eps_x):
    X = _d(X)
    c, e = np.zeros(1), np.zeros(1)
    cg, cH, eg, eH = np.zeros(12), np.zeros((12, 12)), np.zeros(12), np.zeros((12, 12))
    lib().ipcref_ee_mollifier(_p(X), eps_x, _p(c), _p(cg), _p(cH), _p(e), _p(eg), _p(eH))
    return c[0], cg, cH, e[0], eg, eH


def seg_tri_intersect(X5):
    X5 = _d(X5)
    return bool(lib().ipcref_seg_tri_intersect(_p(X5)))


def svd3(F):
    F = _d(F)
    U, s, V = np.zeros((3, 3)), np.zeros(3), np.zeros((3, 3))
    lib().ipcref_svd3(_p(F), _p(U), _p(s), _p(V))
    return U, s, V


def make_pd(A):
    A = _d(A).copy()
    lib().ipcref_make_pd(A.shape[0], _p(A))
    return A
