"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).

ORACLE -- TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never by ipc_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liborc.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def build(force: bool = False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        # on the GPU box only the prebuilt .so is expected; build when sources are newer / missing
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        L = _lib
        L.orc_mesh_create.restype = C.c_void_p
        L.orc_mesh_create.argtypes = [C.c_int, C.c_int, c_dp, c_ip, C.c_double, C.c_double, C.c_double]
        L.orc_opt_create.restype = C.c_void_p
        L.orc_opt_create.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
        L.orc_chol_create.restype = C.c_void_p
        L.orc_chol_create.argtypes = [C.c_int, c_ip, c_ip, C.c_int]
        L.orc_chol_nnzL.restype = C.c_longlong
        L.orc_chol_flops.restype = C.c_double
        L.orc_mesh_avg_edge_len.restype = C.c_double
        L.orc_mesh_bbox_diag2.restype = C.c_double
        L.orc_filter_step_size.restype = C.c_double
        L.orc_pattern_ia.restype = c_ip
        L.orc_pattern_ja.restype = c_ip
        for name in ("orc_mesh_destroy", "orc_mesh_set_surface", "orc_mesh_set_dbc", "orc_mesh_clear_dbc",
                     "orc_mesh_set_V", "orc_mesh_get_V", "orc_mesh_get_features", "orc_mesh_avg_edge_len",
                     "orc_mesh_bbox_diag2", "orc_mesh_check_inversion", "orc_elastic_energy",
                     "orc_elastic_gradient", "orc_elastic_hessian_elem", "orc_pattern_build",
                     "orc_pattern_add_edges", "orc_pattern_rows", "orc_pattern_ia", "orc_pattern_ja",
                     "orc_assemble_hessian", "orc_csr_symv", "orc_inversion_step", "orc_filter_step_size",
                     "orc_chol_destroy", "orc_chol_nnzL", "orc_chol_flops", "orc_chol_factorize",
                     "orc_chol_solve", "orc_opt_destroy", "orc_opt_set_twist", "orc_opt_set_rel_tol",
                     "orc_opt_precompute", "orc_opt_newton_iter", "orc_opt_begin_timestep",
                     "orc_opt_end_timestep", "orc_opt_solve_timestep", "orc_opt_get", "orc_opt_timers"):
            getattr(L, name).argtypes = None
    return _lib


def svd3(F):
    F = np.asfortranarray(F, dtype=np.float64)
    U = np.zeros((3, 3), order="F")
    V = np.zeros((3, 3), order="F")
    S = np.zeros(3)
    lib().orc_svd3(_dp(F), _dp(U), _dp(S), _dp(V))
    return U, S, V


def make_pd(A):
    A = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    lib().orc_make_pd(C.c_int(A.shape[0]), _dp(A))
    return A


def nh_energy_sigma(s, mu, lam):
    s = np.ascontiguousarray(s, dtype=np.float64)
    E = C.c_double()
    lib().orc_nh_energy_sigma(_dp(s), C.c_double(mu), C.c_double(lam), C.byref(E))
    return E.value


def nh_P(F, mu, lam):
    F = np.asfortranarray(F, dtype=np.float64)
    P = np.zeros((3, 3), order="F")
    lib().orc_nh_P(_dp(F), C.c_double(mu), C.c_double(lam), _dp(P))
    return P


def nh_dPdF(F, mu, lam, w=1.0, projectSPD=False):
    F = np.asfortranarray(F, dtype=np.float64)
    out = np.zeros((9, 9), order="F")
    lib().orc_nh_dPdF(_dp(F), C.c_double(mu), C.c_double(lam), C.c_double(w), C.c_int(int(projectSPD)), _dp(out))
    return out


class Mesh:
    def __init__(self, V, F, YM=2e4, PR=0.4, density=1000.0):
        self.nV, self.nT = V.shape[0], F.shape[0]
        self._Vr = np.asfortranarray(V, dtype=np.float64)
        self._F = np.asfortranarray(F, dtype=np.int32)
        self.h = C.c_void_p(lib().orc_mesh_create(self.nV, self.nT, _dp(self._Vr), _ip(self._F),
                                                  YM, PR, density))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mesh_destroy(self.h)
            self.h = None

    def set_exact_predicates(self, on=True):
        lib().orc_mesh_set_exact_predicates(self.h, C.c_int(int(on)))

    def set_surface(self, SF, codim_edges=None):
        """codim_edges: n x 2 node pairs of `.seg` shapes (Mesh::CE); nodes without any neighbour count as `.pt` points"""
        SF = np.asfortranarray(np.asarray(SF, dtype=np.int32).reshape(-1, 3))
        if codim_edges is None or len(codim_edges) == 0:
            lib().orc_mesh_set_surface_codim(self.h, C.c_int(SF.shape[0]), _ip(SF), C.c_int(0), None)
            return
        CE = np.ascontiguousarray(codim_edges, dtype=np.int32).reshape(-1, 2)
        lib().orc_mesh_set_surface_codim(self.h, C.c_int(SF.shape[0]), _ip(SF), C.c_int(CE.shape[0]), _ip(CE))

    def set_dbc(self, ids, typ):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lib().orc_mesh_set_dbc(self.h, C.c_int(len(ids)), _ip(ids), C.c_int(typ))

    def set_codim_nodes(self, ids, mass):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        m = np.ascontiguousarray(mass, dtype=np.float64)
        lib().orc_mesh_set_codim_nodes(self.h, C.c_int(len(ids)), _ip(ids), _dp(m))

    def set_obstacle(self, ids, obstacle_only=False):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lib().orc_mesh_set_obstacle(self.h, C.c_int(len(ids)), _ip(ids), C.c_int(int(obstacle_only)))

    def set_component_material(self, node_range, tet_range, density, YM, PR):
        lib().orc_mesh_set_component_material(self.h, C.c_int(node_range[0]), C.c_int(node_range[1]), C.c_int(tet_range[0]), C.c_int(tet_range[1]),
                                              C.c_double(density), C.c_double(YM), C.c_double(PR))

    def clear_dbc(self):
        lib().orc_mesh_clear_dbc(self.h)

    def set_energy_type(self, name):
        """Config `energy NH|FCR` (Config.cpp:23-24, 107-111)."""
        lib().orc_mesh_set_energy_type(self.h, C.c_int({"NH": 0, "FCR": 1}[name]))

    def set_V(self, V):
        V = np.asfortranarray(V, dtype=np.float64)
        lib().orc_mesh_set_V(self.h, _dp(V))

    def get_V(self):
        V = np.zeros((self.nV, 3), order="F")
        lib().orc_mesh_get_V(self.h, _dp(V))
        return V

    def features(self):
        A = np.zeros((self.nT, 9))
        vol = np.zeros(self.nT)
        mass = np.zeros(self.nV)
        mu = np.zeros(self.nT)
        lam = np.zeros(self.nT)
        lib().orc_mesh_get_features(self.h, _dp(A), _dp(vol), _dp(mass), _dp(mu), _dp(lam))
        return dict(restTriInv=A, triArea=vol, mass=mass, mu=mu, lam=lam,
                    avgEdgeLen=lib().orc_mesh_avg_edge_len(self.h),
                    bboxDiag2=lib().orc_mesh_bbox_diag2(self.h))

    def check_inversion(self):
        return bool(lib().orc_mesh_check_inversion(self.h))

    def elastic_energy(self, coef=1.0, per_elem=False):
        E = C.c_double()
        pe = np.zeros(self.nT) if per_elem else None
        lib().orc_elastic_energy(self.h, C.c_double(coef), C.byref(E), _dp(pe))
        return (E.value, pe) if per_elem else E.value

    def elastic_gradient(self, coef=1.0, projectDBC=True):
        g = np.zeros(3 * self.nV)
        lib().orc_elastic_gradient(self.h, C.c_double(coef), C.c_int(int(projectDBC)), _dp(g))
        return g

    def elastic_hessian_elem(self, e, coef=1.0, projectSPD=True):
        H = np.zeros((12, 12), order="F")
        lib().orc_elastic_hessian_elem(self.h, C.c_int(e), C.c_double(coef), C.c_int(int(projectSPD)), _dp(H))
        return H

    def pattern(self, extra_edges=None):
        if extra_edges is not None and len(extra_edges):
            ee = np.ascontiguousarray(extra_edges, dtype=np.int32)
            lib().orc_pattern_add_edges(self.h, C.c_int(ee.shape[0]), _ip(ee))
        nnz = lib().orc_pattern_build(self.h)
        n = lib().orc_pattern_rows(self.h)
        ia = np.ctypeslib.as_array(lib().orc_pattern_ia(self.h), shape=(n + 1,)).copy()
        ja = np.ctypeslib.as_array(lib().orc_pattern_ja(self.h), shape=(nnz,)).copy()
        return ia, ja

    def assemble_hessian(self, nnz, coef=1.0, projectDBC=True):
        a = np.zeros(nnz)
        lib().orc_assemble_hessian(self.h, C.c_double(coef), C.c_int(int(projectDBC)), _dp(a))
        return a

    def symv(self, a, x):
        y = np.zeros_like(x)
        lib().orc_csr_symv(self.h, _dp(a), _dp(x), _dp(y))
        return y

    def inversion_step(self, p, slackness=0.2):
        out = np.zeros(self.nT)
        p = np.ascontiguousarray(p, dtype=np.float64)
        lib().orc_inversion_step(self.h, _dp(p), C.c_double(slackness), _dp(out))
        return out

    def filter_step_size(self, p, step=1.0):
        p = np.ascontiguousarray(p, dtype=np.float64)
        return lib().orc_filter_step_size(self.h, _dp(p), C.c_double(step))


class Chol:
    def __init__(self, ia, ja, nthreads=0):
        self.n = len(ia) - 1
        self._ia = np.ascontiguousarray(ia, dtype=np.int32)
        self._ja = np.ascontiguousarray(ja, dtype=np.int32)
        self.h = C.c_void_p(lib().orc_chol_create(self.n, _ip(self._ia), _ip(self._ja), nthreads or os.cpu_count()))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_chol_destroy(self.h)
            self.h = None

    @property
    def nnzL(self):
        return lib().orc_chol_nnzL(self.h)

    @property
    def flops(self):
        return lib().orc_chol_flops(self.h)

    def factorize(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return bool(lib().orc_chol_factorize(self.h, _dp(a)))

    def solve(self, rhs):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.zeros_like(rhs)
        lib().orc_chol_solve(self.h, _dp(rhs), _dp(x))
        return x


class Optimizer:
    def __init__(self, mesh: Mesh, dt=0.04, gravity=False, nthreads=0):
        self.mesh = mesh
        self.h = C.c_void_p(lib().orc_opt_create(mesh.h, dt, int(gravity), nthreads or os.cpu_count()))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_opt_destroy(self.h)
            self.h = None

    def set_twist(self, left, right, ang_vel=0.4 * np.pi):
        left = np.ascontiguousarray(left, dtype=np.int32)
        right = np.ascontiguousarray(right, dtype=np.int32)
        lib().orc_opt_set_twist(self.h, C.c_int(len(left)), _ip(left), C.c_int(len(right)), _ip(right),
                                C.c_double(ang_vel))

    def set_rel_tol(self, tol):
        lib().orc_opt_set_rel_tol(self.h, C.c_double(tol))

    def set_kappa(self, kappa):
        lib().orc_opt_set_kappa(self.h, C.c_double(kappa))

    def set_dhat_target(self, eps):
        lib().orc_opt_set_dhat_target(self.h, C.c_double(eps))

    def set_damping(self, damping_stiff):
        lib().orc_opt_set_damping(self.h, C.c_double(damping_stiff))

    def precompute(self):
        if lib().orc_opt_precompute(self.h) != 0:
            raise RuntimeError("intersection detected in initial configuration")

    def begin_timestep(self):
        lib().orc_opt_begin_timestep(self.h)

    def newton_iter(self):
        return int(lib().orc_opt_newton_iter(self.h))

    def end_timestep(self):
        lib().orc_opt_end_timestep(self.h)

    def solve_timestep(self, max_iter=100):
        return int(lib().orc_opt_solve_timestep(self.h, C.c_int(max_iter)))

    def state(self):
        nV = self.mesh.nV
        V = np.zeros((nV, 3), order="F")
        p = np.zeros(3 * nV)
        g = np.zeros(3 * nV)
        sc = np.zeros(8)
        lib().orc_opt_get(self.h, _dp(V), _dp(p), _dp(g), _dp(sc))
        return dict(V=V, searchDir=p, gradient=g, E=sc[0], stepSize=sc[1], targetGRes=sc[2],
                    innerIterAmt=int(sc[3]), timestep=int(sc[4]), alphaFeasible=sc[5], kappa=sc[6], dHat=sc[7])

    def timers(self):
        t = np.zeros(16)
        lib().orc_opt_timers(self.h, _dp(t))
        return t


def assemble_shard(mesh: "Mesh", nnz, coef, projectDBC, t0, t1, owner, xTilde):
    """Elasticity of the tets [t0, t1) (+ nodal terms when owner) -- one rank's share of a sharded assembly."""
    xt = np.asfortranarray(xTilde, dtype=np.float64)
    a = np.zeros(nnz)
    g = np.zeros(3 * mesh.nV)
    lib().orc_assemble_shard(mesh.h, C.c_double(coef), C.c_int(int(projectDBC)), C.c_int(t0), C.c_int(t1),
                             C.c_int(int(owner)), _dp(xt), _dp(a), _dp(g))
    return a, g


# ---- contact (orc_contact.cpp) --------------------------------------------------------------------------------
K_PP, K_PE, K_PT, K_EE = 0, 1, 2, 3


def stencil_distance(kind, X, derivs=True):
    """X: (4,3) node positions (unused rows ignored).  Returns d, g (12), H (12,12)."""
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    d = C.c_double()
    g = np.zeros(12)
    H = np.zeros((12, 12), order="F")
    lib().orc_stencil_distance(C.c_int(kind), _dp(X), C.byref(d), _dp(g) if derivs else None, _dp(H) if derivs else None)
    return d.value, g, H


def cross_sqnorm(X):
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    c = C.c_double()
    g = np.zeros(12)
    H = np.zeros((12, 12), order="F")
    lib().orc_cross_sqnorm(_dp(X), C.byref(c), _dp(g), _dp(H))
    return c.value, g, H


def barrier(d, dHat):
    b, gb, Hb = C.c_double(), C.c_double(), C.c_double()
    lib().orc_barrier(C.c_double(d), C.c_double(dHat), C.byref(b), C.byref(gb), C.byref(Hb))
    return b.value, gb.value, Hb.value


def mollifier(c, eps_x):
    e, eg, eH = C.c_double(), C.c_double(), C.c_double()
    lib().orc_mollifier(C.c_double(c), C.c_double(eps_x), C.byref(e), C.byref(eg), C.byref(eH))
    return e.value, eg.value, eH.value


def dtype_pt(X):
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    return int(lib().orc_dtype_pt(_dp(X)))


def dtype_ee(X):
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    return int(lib().orc_dtype_ee(_dp(X)))


class Contacts:
    """MMActiveSet / paraEEMMCVIDSet / paraEEeIeJSet of one self-collision handler."""

    def __init__(self):
        lib().orc_contacts_create.restype = C.c_void_p
        lib().orc_contact_energy.restype = C.c_double
        self.h = C.c_void_p(lib().orc_contacts_create())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_contacts_destroy(self.h)
            self.h = None

    def build(self, mesh: "Mesh", dHat, brute=False):
        lib().orc_contacts_build(self.h, mesh.h, C.c_double(dHat), C.c_int(int(brute)))
        return self.get()

    def set(self, active, para=None, para_eiej=None):
        a = np.ascontiguousarray(active, dtype=np.int32).reshape(-1, 4)
        p = np.ascontiguousarray(para if para is not None else np.zeros((0, 4)), dtype=np.int32).reshape(-1, 4)
        q = np.ascontiguousarray(para_eiej if para_eiej is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
        lib().orc_contacts_set(self.h, C.c_int(a.shape[0]), _ip(a), C.c_int(p.shape[0]), _ip(p), _ip(q))

    def get(self):
        n = np.zeros(3, dtype=np.int32)
        lib().orc_contacts_sizes(self.h, _ip(n))
        a = np.zeros((n[0], 4), dtype=np.int32)
        p = np.zeros((n[1], 4), dtype=np.int32)
        q = np.zeros((n[1], 2), dtype=np.int32)
        cs = np.zeros((n[2], 2), dtype=np.int32)
        lib().orc_contacts_get(self.h, _ip(a), _ip(p), _ip(q), _ip(cs))
        return dict(active=a, para=p, para_eiej=q, cs_ptee=cs)

    def energy(self, mesh, dHat, kappa):
        return lib().orc_contact_energy(self.h, mesh.h, C.c_double(dHat), C.c_double(kappa))

    def gradient(self, mesh, dHat, kappa, projectDBC=True):
        g = np.zeros(3 * mesh.nV)
        lib().orc_contact_gradient(self.h, mesh.h, C.c_double(dHat), C.c_double(kappa), C.c_int(int(projectDBC)), _dp(g))
        return g

    def hessian(self, mesh, nnz, dHat, kappa, projectDBC=True):
        a = np.zeros(nnz)
        lib().orc_contact_hessian(self.h, mesh.h, C.c_double(dHat), C.c_double(kappa), C.c_int(int(projectDBC)), _dp(a))
        return a

    def connectivity(self, mesh):
        cap = 1 << 20
        buf = np.zeros((cap, 2), dtype=np.int32)
        n = lib().orc_contact_connectivity(self.h, mesh.h, C.c_int(cap), _ip(buf))
        return buf[:n].copy()


def accd(kind, X, P, eta=0.2, tmax=1.0):
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(4, 3)
    lib().orc_accd.restype = C.c_double
    return lib().orc_accd(C.c_int(kind), _dp(X), _dp(P), C.c_double(eta), C.c_double(tmax))


def unclassified_d2(kind, X):
    """Squared distance of the stencil as a whole (point-triangle / segment-segment as closed sets): what accd advances on."""
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    lib().orc_unclassified_distance.restype = C.c_double
    d = lib().orc_unclassified_distance(C.c_int(kind), _dp(X))
    return d * d


def ccd_exact(kind, X, P, tmax=1.0):
    """First time in [0, tmax] at which the point touches the triangle / the edges cross (eta = 0, cubic coplanarity roots), inf if none."""
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3)
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(4, 3)
    lib().orc_ccd_exact.restype = C.c_double
    return lib().orc_ccd_exact(C.c_int(kind), _dp(X), _dp(P), C.c_double(tmax))


def ccd_partial(contacts: "Contacts", mesh: "Mesh", p, slackness=0.8, step=1.0):
    p = np.ascontiguousarray(p, dtype=np.float64)
    arg = C.c_int(-1)
    lib().orc_ccd_partial.restype = C.c_double
    s = lib().orc_ccd_partial(contacts.h, mesh.h, _dp(p), C.c_double(slackness), C.c_double(step), C.byref(arg))
    return s, arg.value


def ccd_full(mesh: "Mesh", p, slackness=0.8, step=1.0):
    p = np.ascontiguousarray(p, dtype=np.float64)
    pair = np.zeros(2, dtype=np.int32)
    n = C.c_int()
    lib().orc_ccd_full.restype = C.c_double
    s = lib().orc_ccd_full(mesh.h, _dp(p), C.c_double(slackness), C.c_double(step), _ip(pair), C.byref(n))
    return s, tuple(int(x) for x in pair), n.value


def ccd_full_reference(mesh: "Mesh", p, slackness=0.8, step=1.0):
    """The reference's full sweep (cap of the step by the hash, shared-cell candidates, PP / PE / PT / EE pairs).  Returns the bound,
    the step after the cap, the limiting pair (kind, i, j) and the number of pairs queried."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    arg = np.zeros(3, dtype=np.int32)
    n, cap = C.c_int(), C.c_double()
    lib().orc_ccd_full_reference.restype = C.c_double
    s = lib().orc_ccd_full_reference(mesh.h, _dp(p), C.c_double(slackness), C.c_double(step), C.byref(cap), _ip(arg), C.byref(n))
    return s, cap.value, tuple(int(x) for x in arg), n.value


def is_intersected(mesh: "Mesh"):
    return bool(lib().orc_is_intersected(mesh.h))


def opt_enable_self_collision(opt: "Optimizer", dHatEps=1e-3):
    lib().orc_opt_enable_self_collision(opt.h, C.c_double(dHatEps))


def opt_add_dirichlet(opt: "Optimizer", ids, lin_vel=(0, 0, 0), ang_vel_deg=(0, 0, 0), t0=0.0, t1=float("inf")):
    """One `DBC bboxMin bboxMax linVel angVel [t0 t1]` entry of a shape line (Config.cpp:246-263; degrees per second in the script)."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    lin = np.ascontiguousarray(lin_vel, dtype=np.float64)
    ang = np.ascontiguousarray(np.asarray(ang_vel_deg, dtype=np.float64) * np.pi / 180)
    lib().orc_opt_add_dirichlet(opt.h, C.c_int(len(ids)), _ip(ids), _dp(lin), _dp(ang), C.c_double(t0), C.c_double(t1))


def opt_end_dirichlet(opt: "Optimizer", group, t_end):
    """The handle of a state-dependent script is released: group `group` ends at t_end (AnimScripter.cpp:1619-1632)."""
    lib().orc_opt_end_dirichlet(opt.h, C.c_int(group), C.c_double(t_end))


def opt_set_dirichlet_targets(o, group, targets):
    """mesh-sequence motion (AnimScripter.cpp:1465-1528): where the group's nodes are to be after the next time step; None ends it"""
    if targets is None:
        lib().orc_opt_set_dirichlet_targets(o.h, C.c_int(group), C.c_int(0), None)
        return
    t = np.ascontiguousarray(np.asarray(targets, dtype=np.float64).reshape(-1, 3))
    lib().orc_opt_set_dirichlet_targets(o.h, C.c_int(group), C.c_int(t.shape[0]), _dp(t))


def opt_set_dirichlet_motion(opt: "Optimizer", group, lin_vel=(0, 0, 0), ang_vel_deg=(0, 0, 0), center=None, force_nonzero=True):
    """Motion of Dirichlet group `group` for the coming time steps (the rule-driven scripts of AnimScripter.cpp:1961-2135)."""
    lin = np.ascontiguousarray(lin_vel, dtype=np.float64)
    ang = np.ascontiguousarray(np.asarray(ang_vel_deg, dtype=np.float64) * np.pi / 180)
    ctr = None if center is None else np.ascontiguousarray(center, dtype=np.float64)
    lib().orc_opt_set_dirichlet_motion(opt.h, C.c_int(group), _dp(lin), _dp(ang), None if ctr is None else _dp(ctr), C.c_int(int(force_nonzero)))


def opt_add_neumann(opt: "Optimizer", ids, accel, t0=0.0, t1=float("inf")):
    """One `NBC bboxMin bboxMax force [t0 t1]` entry of a shape line (Config.cpp:264-280)."""
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    a = np.ascontiguousarray(accel, dtype=np.float64)
    lib().orc_opt_add_neumann(opt.h, C.c_int(len(ids)), _ip(ids), _dp(a), C.c_double(t0), C.c_double(t1))


def opt_dbc_state(opt: "Optimizer"):
    out = np.zeros(4)
    lib().orc_opt_get_dbc_state(opt.h, _dp(out))
    return dict(completed=out[0], rho=out[1], projectDBC=bool(out[2]), n_targets=int(out[3]))


def opt_kinematics(opt: "Optimizer"):
    n3 = 3 * opt.mesh.nV
    vel, acc, dx = np.zeros(n3), np.zeros(n3), np.zeros(n3)
    lib().orc_opt_get_kinematics(opt.h, _dp(vel), _dp(acc), _dp(dx))
    return dict(velocity=vel, acceleration=acc, dx_Elastic=dx)


def opt_save_status(opt: "Optimizer", path):
    """Optimizer::saveStatus (Optimizer.cpp:2964-3011): the reference's text checkpoint."""
    st, k = opt.state(), opt_kinematics(opt)
    nV = opt.mesh.nV
    f = lambda x: format(float(x), ".20g")  # setprecision(digits10 of long double + 2)
    with open(path, "w") as out:
        out.write(f"timestep {st['timestep']}\n\n")
        out.write(f"position {nV} 3\n")
        for r in st["V"]:
            out.write(" ".join(f(x) for x in r) + "\n")
        out.write(f"\nvelocity {3 * nV}\n")
        for x in k["velocity"]:
            out.write(f(x) + "\n")
        for name in ("acceleration", "dx_Elastic"):
            out.write(f"\n{name} {nV} 3\n")
            for r in k[name].reshape(nV, 3):
                out.write(" ".join(f(x) for x in r) + "\n")


def opt_load_status(opt: "Optimizer", path):
    """`restart` branch of the Optimizer constructor (Optimizer.cpp:179-248): token grammar of the status file."""
    nV = opt.mesh.nV
    toks = open(path).read().split()
    V = opt.state()["V"].copy()
    vel, acc, dx = np.zeros(3 * nV), np.zeros((nV, 3)), np.zeros((nV, 3))
    timestep, i = 0, 0
    while i < len(toks):
        t = toks[i]
        if t == "timestep":
            timestep = int(toks[i + 1]); i += 2
        elif t == "velocity":
            n = int(toks[i + 1]); vel[:n] = [float(x) for x in toks[i + 2:i + 2 + n]]; i += 2 + n
        elif t in ("position", "acceleration", "dx_Elastic"):
            rows, dim = int(toks[i + 1]), int(toks[i + 2])
            assert rows <= nV and dim == 3
            vals = np.array([float(x) for x in toks[i + 3:i + 3 + rows * dim]]).reshape(rows, dim)
            {"position": V, "acceleration": acc, "dx_Elastic": dx}[t][:rows] = vals
            i += 3 + rows * dim
        else:
            i += 1
    opt.mesh.set_V(V)
    lib().orc_opt_restart(opt.h, C.c_int(timestep), _dp(vel), _dp(np.ascontiguousarray(acc).reshape(-1)), _dp(np.ascontiguousarray(dx).reshape(-1)))


def opt_set_time_integration(opt: "Optimizer", name, beta=0.25, gamma=0.5):
    """Config `timeIntegration BE | NM beta gamma` (defaults Config.hpp:96)."""
    lib().orc_opt_set_time_integration(opt.h, C.c_int({"BE": 0, "NM": 1}[name]), C.c_double(beta), C.c_double(gamma))


def opt_set_warm_start(opt: "Optimizer", option):
    lib().orc_opt_set_warm_start(opt.h, C.c_int(int(option)))


def opt_warm_step(opt: "Optimizer"):
    lib().orc_opt_warm_step.restype = C.c_double
    return lib().orc_opt_warm_step(opt.h)


def opt_set_velocity(opt: "Optimizer", vel):
    v = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1)
    lib().orc_opt_set_velocity(opt.h, _dp(v))


class HalfSpace:
    """Analytic half-space obstacle (HalfSpace.cpp) with its vertex constraint set."""

    def __init__(self, origin, normal):
        L = lib()
        L.orc_halfspace_create.restype = C.c_void_p
        L.orc_halfspace_energy.restype = C.c_double
        L.orc_halfspace_step_bound.restype = C.c_double
        o = np.ascontiguousarray(origin, dtype=np.float64)
        n = np.ascontiguousarray(normal, dtype=np.float64)
        self.h = C.c_void_p(L.orc_halfspace_create(_dp(o), _dp(n)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_halfspace_destroy(self.h)
            self.h = None

    def build(self, mesh, dHat):
        n = lib().orc_halfspace_build(self.h, mesh.h, C.c_double(dHat))
        v = np.zeros(n, dtype=np.int32)
        lib().orc_halfspace_get(self.h, _ip(v))
        return v

    def energy(self, mesh, dHat, kappa):
        return lib().orc_halfspace_energy(self.h, mesh.h, C.c_double(dHat), C.c_double(kappa))

    def gradient(self, mesh, dHat, kappa):
        g = np.zeros(3 * mesh.nV)
        lib().orc_halfspace_gradient(self.h, mesh.h, C.c_double(dHat), C.c_double(kappa), _dp(g))
        return g

    def hessian(self, mesh, nnz, dHat, kappa, projectDBC=True):
        a = np.zeros(nnz)
        lib().orc_halfspace_hessian(self.h, mesh.h, C.c_double(dHat), C.c_double(kappa), C.c_int(int(projectDBC)), _dp(a))
        return a

    def move(self, mesh, delta, slackness=0.5):
        """HalfSpace::move (HalfSpace.cpp:389-416): (origin after the move, fraction of delta that is left)"""
        lib().orc_halfspace_move.restype = C.c_double
        d, out = np.ascontiguousarray(delta, dtype=np.float64), np.zeros(3)
        left = lib().orc_halfspace_move(self.h, mesh.h, _dp(d), C.c_double(slackness), _dp(out))
        return out, left

    def step_bound(self, mesh, p, slackness=0.9, step=1.0):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        return lib().orc_halfspace_step_bound(self.h, mesh.h, _dp(p), C.c_double(slackness), C.c_double(step))


def opt_add_half_space(opt: "Optimizer", origin, normal, dHatEps=1e-3):
    o = np.ascontiguousarray(origin, dtype=np.float64)
    n = np.ascontiguousarray(normal, dtype=np.float64)
    return lib().orc_opt_add_half_space(opt.h, _dp(o), _dp(n), C.c_double(dHatEps))


def opt_half_space_set(opt: "Optimizer", idx=0):
    n = lib().orc_opt_get_half_space_set(opt.h, C.c_int(idx), None)
    v = np.zeros(n, dtype=np.int32)
    lib().orc_opt_get_half_space_set(opt.h, C.c_int(idx), _ip(v))
    return v


class Friction:
    """Lagged friction data of one active set (FrictionUtils.hpp / SelfCollisionHandler.cpp:2481-2988)."""

    def __init__(self):
        L = lib()
        L.orc_friction_create.restype = C.c_void_p
        L.orc_friction_energy.restype = C.c_double
        self.h = C.c_void_p(L.orc_friction_create())
        self.n = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_friction_destroy(self.h)
            self.h = None

    def update(self, mesh, active, dHat, kappa):
        a = np.ascontiguousarray(active, dtype=np.int32).reshape(-1, 4)
        self.n = a.shape[0]
        lib().orc_friction_update(self.h, mesh.h, C.c_int(self.n), _ip(a), C.c_double(dHat), C.c_double(kappa))
        lam, co, ba = np.zeros(self.n), np.zeros((self.n, 2)), np.zeros((self.n, 6))
        lib().orc_friction_get(self.h, _dp(lam), _dp(co), _dp(ba))
        return dict(lam=lam, coord=co, basis=ba)

    def energy(self, mesh, Vt, eps2, coef):
        Vt = np.asfortranarray(Vt, dtype=np.float64)
        return lib().orc_friction_energy(self.h, mesh.h, _dp(Vt), C.c_double(eps2), C.c_double(coef))

    def gradient(self, mesh, Vt, eps2, coef):
        Vt = np.asfortranarray(Vt, dtype=np.float64)
        g = np.zeros(3 * mesh.nV)
        lib().orc_friction_gradient(self.h, mesh.h, _dp(Vt), C.c_double(eps2), C.c_double(coef), _dp(g))
        return g

    def hessian(self, mesh, Vt, nnz, eps2, coef, projectDBC=True):
        Vt = np.asfortranarray(Vt, dtype=np.float64)
        a = np.zeros(nnz)
        lib().orc_friction_hessian(self.h, mesh.h, _dp(Vt), C.c_double(eps2), C.c_double(coef), C.c_int(int(projectDBC)), _dp(a))
        return a


def opt_set_parameter_scaling(opt: "Optimizer", use_abs_parameters=False, dtol_rel=1e-9, kappa_min_multiplier=1e11):
    lib().orc_opt_set_parameter_scaling(opt.h, C.c_int(int(use_abs_parameters)), C.c_double(dtol_rel), C.c_double(kappa_min_multiplier))


def opt_set_friction_target(opt: "Optimizer", eps_v_target):
    lib().orc_opt_set_friction_target(opt.h, C.c_double(eps_v_target))


def opt_set_friction(opt: "Optimizer", self_fric=0.0, fric_iter_amt=1, eps_v=1e-3):
    lib().orc_opt_set_friction(opt.h, C.c_double(self_fric), C.c_int(fric_iter_amt), C.c_double(eps_v))


def opt_force_friction_loop(opt: "Optimizer", on=True):
    lib().orc_opt_force_friction_loop(opt.h, C.c_int(int(on)))


def opt_set_friction_scales(opt: "Optimizer", scale_self=1.0, scale_obstacle=1.0):
    lib().orc_opt_set_friction_scales(opt.h, C.c_double(scale_self), C.c_double(scale_obstacle))


def opt_half_space_move(opt: "Optimizer", idx, delta, slackness=0.5):
    lib().orc_opt_half_space_move.restype = C.c_double
    d = np.ascontiguousarray(delta, dtype=np.float64)
    return lib().orc_opt_half_space_move(opt.h, C.c_int(idx), _dp(d), C.c_double(slackness))


def opt_set_half_space_friction(opt: "Optimizer", idx, mu):
    lib().orc_opt_set_half_space_friction(opt.h, C.c_int(idx), C.c_double(mu))


def opt_next_subproblem(opt: "Optimizer"):
    return bool(lib().orc_opt_next_subproblem(opt.h))


def opt_friction_state(opt: "Optimizer"):
    sc = np.zeros(4)
    lib().orc_opt_get_friction(opt.h, _dp(sc), None)
    lam = np.zeros(int(sc[1]))
    lib().orc_opt_get_friction(opt.h, _dp(sc), _dp(lam))
    return dict(fricDHat=sc[0], n_lagged=int(sc[1]), fric_iter=int(sc[2]), n_half_space_lagged=int(sc[3]), lam=lam)


def opt_contact_state(opt: "Optimizer"):
    n = np.zeros(6, dtype=np.int32)
    lib().orc_opt_get_contact(opt.h, _ip(n), None, None)
    a = np.zeros((n[0], 4), dtype=np.int32)
    p = np.zeros((n[1], 4), dtype=np.int32)
    lib().orc_opt_get_contact(opt.h, _ip(n), _ip(a), _ip(p))
    return dict(active=a, para=p, n_candidates=int(n[2]), ccd_arg=int(n[3]), n_full_ccd=int(n[4]), n_pattern_changes=int(n[5]))


def mesh_surface(mesh: "Mesh"):
    n = np.zeros(3, dtype=np.int32)
    lib().orc_mesh_surface_counts(mesh.h, _ip(n))
    svi = np.zeros(n[0], dtype=np.int32)
    sfe = np.zeros((n[2], 2), dtype=np.int32)
    lib().orc_mesh_get_surface(mesh.h, _ip(svi), _ip(sfe))
    return svi, sfe
