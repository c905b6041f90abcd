"""ctypes binding of include/ipcgpu.h.

Method names follow the reference interfaces the C ABI replaces
(LinSysSolver.hpp:31-467, Energy.hpp:27-138, Optimizer.hpp:28-283) so that the
parity tests read like tests of the reference classes.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IPCGPU_LIB_VARIANT selects an experiment build made by tools/ (ipc_amd/build.py variant=...); unset = the product library
_LIB_PATH = os.path.join(_HERE, "libipcgpu_%s.so" % os.environ["IPCGPU_LIB_VARIANT"] if os.environ.get("IPCGPU_LIB_VARIANT") else "libipcgpu.so")
_HEADER = os.path.join(_HERE, "..", "include", "ipcgpu.h")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)

IPCGPU_OK = 0
IPCGPU_NOT_PD = 1
SOLVER_MULTIFRONTAL = 0
SOLVER_ROCSOLVER_CSRRF = 1

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int)


class P2POp(C.Structure):  # ipcgpu_p2p_op
    _fields_ = [("buf_dev", C.c_void_p), ("count", C.c_longlong), ("peer", C.c_int), ("send", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(P2POp))


class IpcGpuError(RuntimeError):
    pass


class NotPositiveDefinite(IpcGpuError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def declared_symbols():
    """Every function declared in include/ipcgpu.h (used by the CPU-side export test)."""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ipcgpu_[a-z0-9_]+)\s*\(", txt)) - {"ipcgpu_allreduce_fn", "ipcgpu_allreduce_stream_fn"})


_RCCL_PATH = os.path.join(os.path.dirname(_LIB_PATH), os.path.basename(_LIB_PATH).replace("libipcgpu", "libipcgpu_rccl", 1) if "libipcgpu_" not in os.path.basename(_LIB_PATH)
                          else os.path.basename(_LIB_PATH).replace(".so", "_rccl.so"))
_RCCL_HEADER = os.path.join(os.path.dirname(_HEADER), "ipcgpu_rccl.h")
_rccl = None


def rccl_declared_symbols():
    """Every function include/ipcgpu_rccl.h declares (the caller-side RCCL binding, libipcgpu_rccl.so)."""
    txt = re.sub(r"/\*.*?\*/", "", open(_RCCL_HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ipcgpu_rccl_[a-z0-9_]+)\s*\(", txt)))


def load_rccl():
    """dlopen libipcgpu_rccl.so (include/adapters/ipcgpu_rccl.cpp: RCCL bound to a context from C)."""
    global _rccl
    if _rccl is None:
        load_library()
        if not os.path.exists(_RCCL_PATH):
            raise IpcGpuError(f"{_RCCL_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _rccl = C.CDLL(_RCCL_PATH)
        _rccl.ipcgpu_rccl_last_error.restype = C.c_char_p
    return _rccl


_lib = None


def load_library():
    """dlopen libipcgpu.so.  Raises when it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise IpcGpuError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(_LIB_PATH)
        _lib.ipcgpu_last_error.restype = C.c_char_p
        _lib.ipcgpu_tet_mesh_free.restype = None
        _lib.ipcgpu_tet_mesh_free.argtypes = [C.c_void_p]
        _lib.ipcgpu_tet_mesh_get.argtypes = [C.c_void_p, c_dp, c_ip, c_ip]
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def read_tet_mesh(path):
    """IglUtils::readTetMesh through the C ABI: (V, T, SF) of a .msh file (MSH 4.1 / 2.2 ASCII or the reference's msh-4.0 dialect)."""
    L = load_library()
    h = C.c_void_p()
    nV, nT, nSF = C.c_int(), C.c_int(), C.c_int()
    rc = L.ipcgpu_read_tet_mesh(str(path).encode(), C.byref(h), C.byref(nV), C.byref(nT), C.byref(nSF))
    if rc != 0:
        raise RuntimeError(f"ipcgpu_read_tet_mesh failed ({rc}): {L.ipcgpu_last_error().decode()}")
    try:
        V = np.zeros((nV.value, 3), order="F")
        T = np.zeros((nT.value, 4), dtype=np.int32, order="F")
        SF = np.zeros((nSF.value, 3), dtype=np.int32, order="F")
        rc = L.ipcgpu_tet_mesh_get(h, _dp(V), _ip(T), _ip(SF))
        if rc != 0:
            raise RuntimeError(f"ipcgpu_tet_mesh_get failed ({rc})")
    finally:
        L.ipcgpu_tet_mesh_free(h)
    return V, T, SF


def save_tet_mesh(path, V, T):
    """IglUtils::saveTetMesh through the C ABI (MSH 4.1 ASCII + $Surface)."""
    L = load_library()
    V = np.asfortranarray(V, dtype=np.float64)
    T = np.asfortranarray(T, dtype=np.int32)
    rc = L.ipcgpu_save_tet_mesh(str(path).encode(), C.c_int(V.shape[0]), C.c_int(T.shape[0]), _dp(V), _ip(T))
    if rc != 0:
        raise RuntimeError(f"ipcgpu_save_tet_mesh failed ({rc}): {L.ipcgpu_last_error().decode()}")


class Context:
    """One `ipcgpu_ctx` == one Optimizer with its Mesh, Energy and LinSysSolver on one GPU."""

    def __init__(self, device: int = 0, solver: int = SOLVER_MULTIFRONTAL):
        self._L = load_library()
        h = C.c_void_p()
        self._chk(self._L.ipcgpu_ctx_create(C.c_int(device), C.byref(h)))
        self.h = h
        self.nV = self.nT = 0
        self._cb = None
        if solver != SOLVER_MULTIFRONTAL:
            self.set_solver(solver)

    def _chk(self, rc, allow_not_pd=False):
        if rc == IPCGPU_OK:
            return rc
        if rc == IPCGPU_NOT_PD and allow_not_pd:
            return rc
        msg = self._L.ipcgpu_last_error().decode(errors="replace")
        if rc == IPCGPU_NOT_PD:
            raise NotPositiveDefinite(msg or "matrix not positive definite")
        raise IpcGpuError(f"ipcgpu error {rc}: {msg}")

    def close(self):
        if getattr(self, "h", None):
            self.rccl_detach()
            self._L.ipcgpu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- context
    def set_solver(self, solver):
        self._chk(self._L.ipcgpu_ctx_set_solver(self.h, C.c_int(solver)))

    def set_shard(self, rank, world):
        self._chk(self._L.ipcgpu_ctx_set_shard(self.h, C.c_int(rank), C.c_int(world)))

    def set_solver_shard(self, rank, world):
        """subtree-sharded factorisation / solves (ipcgpu_linsys_set_shard); call after set_allreduce"""
        self._chk(self._L.ipcgpu_linsys_set_shard(self.h, C.c_int(rank), C.c_int(world)))

    def set_exchange(self, pyfunc):
        """the point-to-point hook of the sharded solver (ipcgpu_opt_set_exchange, host-ordered): pyfunc(ops) with ops = [(dev_ptr, count, peer, send), ...],
        ONE group -- no order between the operations may be assumed; returns 0."""
        def tramp(user, n, ops):
            try:
                return int(pyfunc([(int(ops[i].buf_dev), int(ops[i].count), int(ops[i].peer), int(ops[i].send)) for i in range(n)]))
            except Exception as e:  # noqa: BLE001
                print("exchange hook failed:", e, flush=True)
                return 1
        self._xcb = EXCHANGE_FN(tramp)
        self._chk(self._L.ipcgpu_opt_set_exchange(self.h, self._xcb, None))

    def solver_exchange_stats(self):
        """bytes this rank sent / received point to point through the solver so far (ipcgpu_linsys_exchange_stats)"""
        o = np.zeros(4)
        self._chk(self._L.ipcgpu_linsys_exchange_stats(self.h, _dp(o)))
        return dict(sent_bytes=int(o[0]), received_bytes=int(o[1]), calls=int(o[2]), wait_ms=float(o[3]))

    def solver_critical_path(self):
        """inputs of the strong-scaling model (ipcgpu_linsys_critical_path): dependent pivot steps of the assembly tree, those above the cut, the largest rank's share
        of the flops below the cut, levels, levels with a front above the cut"""
        o = np.zeros(5)
        self._chk(self._L.ipcgpu_linsys_critical_path(self.h, _dp(o)))
        return dict(steps=o[0], steps_above_cut=o[1], max_rank_share_below=o[2], levels=int(o[3]), levels_above_cut=int(o[4]))

    def entry_destinations(self):
        """per CSR entry its slot in the front buffer as the set-up's device kernel computed it (ipcgpu_linsys_entry_destinations)"""
        rows, nnz = self.get_dims()
        out = np.zeros(nnz, dtype=np.int64)
        self._chk(self._L.ipcgpu_linsys_entry_destinations(self.h, out.ctypes.data_as(C.POINTER(C.c_longlong))))
        return out

    def solver_shard_stats(self):
        o = np.zeros(2)
        self._chk(self._L.ipcgpu_linsys_shard_stats(self.h, _dp(o)))
        return dict(world=int(o[0]), shared_flop_fraction=float(o[1]))

    def comm_stats(self):
        """what crossed ranks so far (bytes, calls) through the time stepper's and through the solver's all-reduces; nodes whose rows this rank assembles"""
        o = np.zeros(6)
        self._chk(self._L.ipcgpu_opt_comm_stats(self.h, _dp(o)))
        return dict(stepper_bytes=int(o[0]), stepper_calls=int(o[1]), solver_bytes=int(o[2]), solver_calls=int(o[3]), rows_assembled_nodes=int(o[4]), nodes=int(o[5]))

    def complete_matrix(self):
        self._chk(self._L.ipcgpu_opt_complete_matrix(self.h))

    # ---- RCCL from C (include/ipcgpu_rccl.h): every exchange is then one ncclAllReduce on the context's own stream
    @staticmethod
    def rccl_unique_id():
        R = load_rccl()
        buf = C.create_string_buffer(128)
        if R.ipcgpu_rccl_unique_id(buf) != 0:
            raise IpcGpuError("ipcgpu_rccl_unique_id: " + R.ipcgpu_rccl_last_error().decode())
        return buf.raw

    def rccl_attach(self, rank, world, id128):
        R = load_rccl()
        if R.ipcgpu_rccl_attach(self.h, C.c_int(rank), C.c_int(world), C.c_char_p(bytes(id128))) != 0:
            raise IpcGpuError("ipcgpu_rccl_attach: " + R.ipcgpu_rccl_last_error().decode())
        self._rccl_attached = True

    def rccl_detach(self):
        if getattr(self, "_rccl_attached", False):
            load_rccl().ipcgpu_rccl_detach(self.h)
            self._rccl_attached = False

    def rccl_selftest_p2p(self, rank, world, count=1024):
        R = load_rccl()
        out = C.c_double()
        if R.ipcgpu_rccl_selftest_p2p(self.h, C.c_int(rank), C.c_int(world), C.c_longlong(count), C.byref(out)) != 0:
            raise IpcGpuError("ipcgpu_rccl_selftest_p2p: " + R.ipcgpu_rccl_last_error().decode())
        return out.value

    def rccl_selftest(self, rank, count=1024, op=0):
        R = load_rccl()
        out = C.c_double()
        if R.ipcgpu_rccl_selftest(self.h, C.c_int(rank), C.c_longlong(count), C.c_int(op), C.byref(out)) != 0:
            raise IpcGpuError("ipcgpu_rccl_selftest: " + R.ipcgpu_rccl_last_error().decode())
        return out.value

    def set_allreduce(self, pyfunc):
        """pyfunc(dev_ptr:int, count:int, op:int) -> int (0 ok)."""
        def tramp(user, buf, count, op):
            try:
                return int(pyfunc(int(buf), int(count), int(op)))
            except Exception as e:  # noqa: BLE001
                print("allreduce hook failed:", e, flush=True)
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        self._chk(self._L.ipcgpu_opt_set_allreduce(self.h, self._cb, None))

    # ---- Mesh
    def set_mesh(self, V, F, YM=2e4, PR=0.4, density=1000.0):
        V = np.asfortranarray(V, dtype=np.float64)
        F = np.asfortranarray(F, dtype=np.int32)
        self.nV, self.nT = V.shape[0], F.shape[0]
        self._chk(self._L.ipcgpu_set_mesh(self.h, C.c_int(self.nV), C.c_int(self.nT), _dp(V), _ip(F),
                                          C.c_double(YM), C.c_double(PR), C.c_double(density)))

    def set_dbc(self, ids, typ):
        ids = _i32(ids)
        self._chk(self._L.ipcgpu_set_dbc(self.h, C.c_int(len(ids)), _ip(ids), C.c_int(typ)))

    def set_component_material(self, node_range, tet_range, density, YM, PR):
        self._chk(self._L.ipcgpu_set_component_material(self.h, C.c_int(node_range[0]), C.c_int(node_range[1]), C.c_int(tet_range[0]),
                                                        C.c_int(tet_range[1]), C.c_double(density), C.c_double(YM), C.c_double(PR)))

    def set_energy_type(self, name):
        """Scene-script `energy NH|FCR`."""
        self._chk(self._L.ipcgpu_set_energy_type(self.h, C.c_int({"NH": 0, "FCR": 1}[name])))

    def clear_dbc(self):
        self._chk(self._L.ipcgpu_clear_dbc(self.h))

    def set_positions(self, V):
        V = np.asfortranarray(V, dtype=np.float64)
        assert V.shape == (self.nV, 3)
        self._chk(self._L.ipcgpu_set_positions(self.h, _dp(V)))

    def get_positions(self):
        V = np.zeros((self.nV, 3), order="F")
        self._chk(self._L.ipcgpu_get_positions(self.h, _dp(V)))
        return V

    def set_xtilde(self, V):
        V = np.asfortranarray(V, dtype=np.float64)
        self._chk(self._L.ipcgpu_set_xtilde(self.h, _dp(V)))

    def features(self):
        A = np.zeros((self.nT, 9))
        vol = np.zeros(self.nT)
        mass = np.zeros(self.nV)
        mu = np.zeros(self.nT)
        lam = np.zeros(self.nT)
        self._chk(self._L.ipcgpu_get_features(self.h, _dp(A), _dp(vol), _dp(mass), _dp(mu), _dp(lam)))
        return dict(restTriInv=A, triArea=vol, mass=mass, mu=mu, lam=lam)

    def check_inversion(self):
        ok = C.c_int()
        self._chk(self._L.ipcgpu_check_inversion(self.h, C.byref(ok)))
        return bool(ok.value)

    # ---- Energy
    def elastic_energy(self, coef=1.0):
        E = C.c_double()
        self._chk(self._L.ipcgpu_elastic_energy(self.h, C.c_double(coef), C.byref(E)))
        return E.value

    def elastic_energy_per_elem(self):
        out = np.zeros(self.nT)
        self._chk(self._L.ipcgpu_elastic_energy_per_elem(self.h, _dp(out)))
        return out

    def elastic_gradient(self, coef=1.0, projectDBC=True):
        g = np.zeros(3 * self.nV)
        self._chk(self._L.ipcgpu_elastic_gradient(self.h, C.c_double(coef), C.c_int(int(projectDBC)), _dp(g)))
        return g

    def elastic_hessian_add(self, coef=1.0, projectDBC=True):
        self._chk(self._L.ipcgpu_elastic_hessian_add(self.h, C.c_double(coef), C.c_int(int(projectDBC))))

    def filter_step_size(self, p, step=1.0):
        p = _f64(p)
        s = C.c_double(step)
        self._chk(self._L.ipcgpu_filter_step_size(self.h, _dp(p), C.byref(s)))
        return s.value

    # ---- LinSysSolver
    def set_pattern(self, extra_pairs=None):
        if extra_pairs is None or len(extra_pairs) == 0:
            self._chk(self._L.ipcgpu_linsys_set_pattern(self.h, C.c_int(0), None))
        else:
            e = _i32(extra_pairs)
            self._chk(self._L.ipcgpu_linsys_set_pattern(self.h, C.c_int(e.shape[0]), _ip(e)))

    def set_pattern_csr(self, ia, ja):
        ia, ja = _i32(ia), _i32(ja)
        self._chk(self._L.ipcgpu_linsys_set_pattern_csr(self.h, C.c_int(len(ia) - 1), _ip(ia), _ip(ja)))

    def get_dims(self):
        n, nnz = C.c_int(), C.c_int()
        self._chk(self._L.ipcgpu_linsys_get_dims(self.h, C.byref(n), C.byref(nnz)))
        return n.value, nnz.value

    def get_pattern(self):
        n, nnz = self.get_dims()
        ia = np.zeros(n + 1, dtype=np.int32)
        ja = np.zeros(nnz, dtype=np.int32)
        self._chk(self._L.ipcgpu_linsys_get_pattern(self.h, _ip(ia), _ip(ja)))
        return ia, ja

    def set_zero(self):
        self._chk(self._L.ipcgpu_linsys_set_zero(self.h))

    def get_a(self):
        _, nnz = self.get_dims()
        a = np.zeros(nnz)
        self._chk(self._L.ipcgpu_linsys_get_values(self.h, _dp(a)))
        return a

    def set_a(self, a):
        a = _f64(a)
        self._chk(self._L.ipcgpu_linsys_set_values(self.h, _dp(a)))

    def add_coeff(self, r, c, v):
        self._chk(self._L.ipcgpu_linsys_add_coeff(self.h, C.c_int(r), C.c_int(c), C.c_double(v)))

    def set_coeff(self, r, c, v):
        self._chk(self._L.ipcgpu_linsys_set_coeff(self.h, C.c_int(r), C.c_int(c), C.c_double(v)))

    def multiply(self, x):
        x = _f64(x)
        y = np.zeros_like(x)
        self._chk(self._L.ipcgpu_linsys_multiply(self.h, _dp(x), _dp(y)))
        return y

    def analyze_pattern(self):
        self._chk(self._L.ipcgpu_linsys_analyze_pattern(self.h))

    def factorize(self):
        """Returns True when the matrix is positive definite (LinSysSolver::factorize)."""
        return self._chk(self._L.ipcgpu_linsys_factorize(self.h), allow_not_pd=True) == IPCGPU_OK

    def solve(self, rhs):
        rhs = _f64(rhs)
        x = np.zeros_like(rhs)
        self._chk(self._L.ipcgpu_linsys_solve(self.h, _dp(rhs), _dp(x)))
        return x

    def precondition_diag(self, v):
        v = _f64(v)
        out = np.zeros_like(v)
        self._chk(self._L.ipcgpu_linsys_precondition_diag(self.h, _dp(v), _dp(out)))
        return out

    def set_solver_tuning(self, bulk_min_mb=48.0, bulk_block=256):
        self._chk(self._L.ipcgpu_linsys_set_tuning(self.h, C.c_double(bulk_min_mb), C.c_int(bulk_block)))

    def linsys_stats(self):
        st = np.zeros(4)
        self._chk(self._L.ipcgpu_linsys_stats(self.h, _dp(st)))
        return dict(nnzL=st[0], flops=st[1], fronts=int(st[2]), levels=int(st[3]))

    # ---- SelfCollisionHandler
    def set_surface(self, SF, codim_edges=None):
        """codim_edges: n x 2 node pairs of `.seg` shapes (Mesh::CE); nodes without any neighbour count as `.pt` points"""
        SF = np.asfortranarray(np.asarray(SF, dtype=np.int32).reshape(-1, 3))
        if codim_edges is None or len(codim_edges) == 0:
            self._chk(self._L.ipcgpu_set_surface(self.h, C.c_int(SF.shape[0]), _ip(SF)))
            return
        CE = np.ascontiguousarray(codim_edges, dtype=np.int32).reshape(-1, 2)
        self._chk(self._L.ipcgpu_set_surface_codim(self.h, C.c_int(SF.shape[0]), _ip(SF), C.c_int(CE.shape[0]), _ip(CE)))

    def set_exact_predicates(self, on=True):
        """intersection checks as a USE_PREDICATES build of the reference makes them (exact orient3d)"""
        self._chk(self._L.ipcgpu_set_exact_predicates(self.h, C.c_int(int(on))))

    def set_codim_nodes(self, ids, mass):
        """Surface-only nodes that belong to the mesh (triangle meshes under `shapes`) with their lumped area masses; like a MeshCO they
        stay out of the bounding box behind dHat and of the mean nodal mass behind kappa (matSpaceBBoxSize2(dim) / avgNodeMass(dim))."""
        ids = _i32(ids)
        m = _f64(np.asarray(mass, dtype=np.float64))
        self._chk(self._L.ipcgpu_set_codim_nodes(self.h, C.c_int(len(ids)), _ip(ids), _dp(m)))

    def set_obstacle(self, ids, obstacle_only=False):
        ids = _i32(ids)
        self._chk(self._L.ipcgpu_set_obstacle_nodes(self.h, C.c_int(len(ids)), _ip(ids), C.c_int(int(obstacle_only))))

    def get_surface(self):
        n = np.zeros(3, dtype=np.int32)
        self._chk(self._L.ipcgpu_get_surface(self.h, _ip(n), None, None))
        svi = np.zeros(n[0], dtype=np.int32)
        sfe = np.zeros((n[2], 2), dtype=np.int32)
        self._chk(self._L.ipcgpu_get_surface(self.h, _ip(n), _ip(svi), _ip(sfe)))
        return svi, sfe

    def contact_build(self, dHat):
        n = np.zeros(3, dtype=np.int32)
        self._chk(self._L.ipcgpu_contact_build(self.h, C.c_double(dHat), _ip(n)))
        a = np.zeros((n[0], 4), dtype=np.int32)
        p = np.zeros((n[1], 4), dtype=np.int32)
        q = np.zeros((n[1], 2), dtype=np.int32)
        cs = np.zeros((n[2], 2), dtype=np.int32)
        self._chk(self._L.ipcgpu_contact_get(self.h, _ip(a), _ip(p), _ip(q), _ip(cs)))
        return dict(active=a, para=p, para_eiej=q, cs_ptee=cs)

    def contact_set(self, active, para=None, para_eiej=None):
        a = np.ascontiguousarray(active, dtype=np.int32).reshape(-1, 4)
        p = np.ascontiguousarray(para if para is not None else np.zeros((0, 4)), dtype=np.int32).reshape(-1, 4)
        q = np.ascontiguousarray(para_eiej if para_eiej is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
        self._chk(self._L.ipcgpu_contact_set(self.h, C.c_int(a.shape[0]), _ip(a), C.c_int(p.shape[0]), _ip(p), _ip(q)))

    def contact_energy(self, dHat, kappa):
        E = C.c_double()
        self._chk(self._L.ipcgpu_contact_energy(self.h, C.c_double(dHat), C.c_double(kappa), C.byref(E)))
        return E.value

    def contact_gradient_add(self, dHat, kappa, projectDBC=True, grad=None):
        g = np.zeros(3 * self.nV) if grad is None else _f64(grad).copy()
        self._chk(self._L.ipcgpu_contact_gradient_add(self.h, C.c_double(dHat), C.c_double(kappa), C.c_int(int(projectDBC)), _dp(g)))
        return g

    def contact_hessian_add(self, dHat, kappa, projectDBC=True):
        self._chk(self._L.ipcgpu_contact_hessian_add(self.h, C.c_double(dHat), C.c_double(kappa), C.c_int(int(projectDBC))))

    def contact_connectivity(self):
        n = C.c_int()
        self._chk(self._L.ipcgpu_contact_connectivity(self.h, C.c_int(0), None, C.byref(n)))
        buf = np.zeros((max(n.value, 1), 2), dtype=np.int32)
        self._chk(self._L.ipcgpu_contact_connectivity(self.h, C.c_int(n.value), _ip(buf), C.byref(n)))
        return buf[:n.value].copy()

    def ccd_partial(self, p, slackness=0.8, step=1.0):
        p = _f64(p)
        s = C.c_double(step)
        pair = np.zeros(2, dtype=np.int32)
        self._chk(self._L.ipcgpu_ccd_partial(self.h, _dp(p), C.c_double(slackness), C.byref(s), _ip(pair)))
        return s.value, (int(pair[0]), int(pair[1]))

    def ccd_full(self, p, slackness=0.8, step=1.0):
        p = _f64(p)
        s = C.c_double(step)
        pair = np.zeros(2, dtype=np.int32)
        n = C.c_int()
        self._chk(self._L.ipcgpu_ccd_full(self.h, _dp(p), C.c_double(slackness), C.byref(s), _ip(pair), C.byref(n)))
        return s.value, (int(pair[0]), int(pair[1])), n.value

    def ccd_full_reference(self, p, slackness=0.8, step=1.0):
        """The reference's full sweep: returns the bound, the step after the hash's cap, the limiting pair (kind, i, j), #pairs queried."""
        p = _f64(p)
        s, cap = C.c_double(step), C.c_double()
        arg = np.zeros(3, dtype=np.int32)
        n = C.c_int()
        self._chk(self._L.ipcgpu_ccd_full_reference(self.h, _dp(p), C.c_double(slackness), C.byref(s), C.byref(cap), _ip(arg), C.byref(n)))
        return s.value, cap.value, tuple(int(x) for x in arg), n.value

    def set_ccd_mode(self, mode):
        self._chk(self._L.ipcgpu_set_ccd_mode(self.h, C.c_int(mode)))

    def is_intersected(self):
        f = C.c_int()
        self._chk(self._L.ipcgpu_is_intersected(self.h, C.byref(f)))
        return bool(f.value)

    # ---- Optimizer building blocks
    def assemble_newton(self, dtSq, projectDBC=True, with_gradient=True):
        g = np.zeros(3 * self.nV) if with_gradient else None
        self._chk(self._L.ipcgpu_assemble_newton(self.h, C.c_double(dtSq), C.c_int(int(projectDBC)), _dp(g)))
        return g

    def incremental_potential(self, dtSq):
        E = C.c_double()
        self._chk(self._L.ipcgpu_incremental_potential(self.h, C.c_double(dtSq), C.byref(E)))
        return E.value

    def gradient(self, dtSq, projectDBC=True):
        g = np.zeros(3 * self.nV)
        self._chk(self._L.ipcgpu_gradient(self.h, C.c_double(dtSq), C.c_int(int(projectDBC)), _dp(g)))
        return g

    # ---- Optimizer
    def opt_init(self, dt=0.04, gravity=False):
        self._chk(self._L.ipcgpu_opt_init(self.h, C.c_double(dt), C.c_int(int(gravity))))

    def set_rel_tol(self, tol):
        self._chk(self._L.ipcgpu_opt_set_rel_tol(self.h, C.c_double(tol)))

    def set_twist(self, left, right, ang_vel=0.4 * np.pi):
        left, right = _i32(left), _i32(right)
        self._chk(self._L.ipcgpu_opt_set_twist(self.h, C.c_int(len(left)), _ip(left), C.c_int(len(right)),
                                               _ip(right), C.c_double(ang_vel)))

    def precompute(self):
        self._chk(self._L.ipcgpu_opt_precompute(self.h))

    def begin_timestep(self):
        self._chk(self._L.ipcgpu_opt_begin_timestep(self.h))

    def newton_iter(self):
        cv = C.c_int()
        self._chk(self._L.ipcgpu_opt_newton_iter(self.h, C.byref(cv)))
        return bool(cv.value)

    def end_timestep(self):
        self._chk(self._L.ipcgpu_opt_end_timestep(self.h))

    def solve_timestep(self, max_iter=100):
        n = C.c_int()
        self._chk(self._L.ipcgpu_opt_solve_timestep(self.h, C.c_int(max_iter), C.byref(n)))
        return n.value

    def state(self):
        V = np.zeros((self.nV, 3), order="F")
        p = np.zeros(3 * self.nV)
        g = np.zeros(3 * self.nV)
        sc = np.zeros(8)
        self._chk(self._L.ipcgpu_opt_get_state(self.h, _dp(V), _dp(p), _dp(g), _dp(sc)))
        return dict(V=V, searchDir=p, gradient=g, E=sc[0], stepSize=sc[1], targetGRes=sc[2],
                    innerIterAmt=int(sc[3]), timestep=int(sc[4]), alphaFeasible=sc[5], kappa=sc[6], dHat=sc[7])

    def enable_self_collision(self, dHatEps=1e-3):
        self._chk(self._L.ipcgpu_opt_enable_self_collision(self.h, C.c_double(dHatEps)))

    def set_pattern_lookahead(self, pad=4.0):
        self._chk(self._L.ipcgpu_opt_set_pattern_lookahead(self.h, C.c_double(pad)))

    def add_half_space(self, origin, normal, dHatEps=1e-3):
        o, n = _f64(np.asarray(origin)), _f64(np.asarray(normal))
        idx = C.c_int()
        self._chk(self._L.ipcgpu_opt_add_half_space(self.h, _dp(o), _dp(n), C.c_double(dHatEps), C.byref(idx)))
        return idx.value

    def halfspace_build(self, idx, dHat):
        verts = np.zeros(self.nV, dtype=np.int32)
        n = C.c_int()
        self._chk(self._L.ipcgpu_halfspace_build(self.h, C.c_int(idx), C.c_double(dHat), C.c_int(self.nV), _ip(verts), C.byref(n)))
        return verts[:n.value].copy()

    def halfspace_set(self, idx, verts):
        verts = _i32(verts)
        self._chk(self._L.ipcgpu_halfspace_set(self.h, C.c_int(idx), C.c_int(len(verts)), _ip(verts)))

    def halfspace_energy(self, idx, dHat, kappa):
        E = C.c_double()
        self._chk(self._L.ipcgpu_halfspace_energy(self.h, C.c_int(idx), C.c_double(dHat), C.c_double(kappa), C.byref(E)))
        return E.value

    def halfspace_gradient_add(self, idx, dHat, kappa, grad=None):
        g = np.zeros(3 * self.nV) if grad is None else _f64(grad).copy()
        self._chk(self._L.ipcgpu_halfspace_gradient_add(self.h, C.c_int(idx), C.c_double(dHat), C.c_double(kappa), _dp(g)))
        return g

    def halfspace_hessian_add(self, idx, dHat, kappa, projectDBC=True):
        self._chk(self._L.ipcgpu_halfspace_hessian_add(self.h, C.c_int(idx), C.c_double(dHat), C.c_double(kappa),
                                                       C.c_int(int(projectDBC))))

    def half_space_move(self, idx, delta, slackness=0.5):
        """HalfSpace::move: displace plane `idx` by (the feasible fraction of) delta; returns the fraction left"""
        d = _f64(np.asarray(delta, dtype=np.float64))
        left = C.c_double()
        self._chk(self._L.ipcgpu_halfspace_move(self.h, C.c_int(idx), _dp(d), C.c_double(slackness), C.byref(left)))
        return left.value

    def halfspace_step_bound(self, idx, p, slackness=0.9, step=1.0):
        p = _f64(np.asarray(p).reshape(-1))
        s = C.c_double(step)
        self._chk(self._L.ipcgpu_halfspace_step_bound(self.h, C.c_int(idx), _dp(p), C.c_double(slackness), C.byref(s)))
        return s.value

    def set_friction_target(self, eps_v_target):
        """eps_v homotopy target (`tuning`'s sixth entry); <= 0: the same as eps_v"""
        self._chk(self._L.ipcgpu_opt_set_friction_target(self.h, C.c_double(eps_v_target)))

    def set_constructor_dt(self, h):
        """the step size inside eps_v^2 h^2 and CN_MBC (0.025 in the reference whatever the scene's dt: Optimizer.cpp:116, 268, 290-303)"""
        self._chk(self._L.ipcgpu_opt_set_constructor_dt(self.h, C.c_double(h)))

    def set_parameter_scaling(self, use_abs_parameters=False, dtol_rel=1e-9, kappa_min_multiplier=1e11):
        """`useAbsParameters`, tuning[3] and `kappaMinMultiplier` of the scene file (Config.cpp:553-558)"""
        self._chk(self._L.ipcgpu_opt_set_parameter_scaling(self.h, C.c_int(int(use_abs_parameters)), C.c_double(dtol_rel),
                                                           C.c_double(kappa_min_multiplier)))

    def set_friction(self, self_fric=0.0, fric_iter_amt=1, eps_v=1e-3):
        self._chk(self._L.ipcgpu_opt_set_friction(self.h, C.c_double(self_fric), C.c_int(fric_iter_amt), C.c_double(eps_v)))

    def set_kappa(self, kappa):
        self._chk(self._L.ipcgpu_opt_set_kappa(self.h, C.c_double(kappa)))

    def set_dhat_target(self, eps):
        self._chk(self._L.ipcgpu_opt_set_dhat_target(self.h, C.c_double(eps)))

    def set_damping(self, damping_stiff):
        self._chk(self._L.ipcgpu_opt_set_damping(self.h, C.c_double(damping_stiff)))

    def set_friction_scales(self, scale_self=1.0, scale_obstacle=1.0):
        self._chk(self._L.ipcgpu_opt_set_friction_scales(self.h, C.c_double(scale_self), C.c_double(scale_obstacle)))

    def force_friction_loop(self, on=True):
        self._chk(self._L.ipcgpu_opt_force_friction_loop(self.h, C.c_int(int(on))))

    def set_half_space_friction(self, idx, mu):
        self._chk(self._L.ipcgpu_opt_set_half_space_friction(self.h, C.c_int(idx), C.c_double(mu)))

    def next_subproblem(self):
        more = C.c_int()
        self._chk(self._L.ipcgpu_opt_next_subproblem(self.h, C.byref(more)))
        return bool(more.value)

    def friction_state(self):
        sc = np.zeros(4)
        self._chk(self._L.ipcgpu_opt_get_friction_state(self.h, _dp(sc), None))
        lam = np.zeros(int(sc[1]))
        if lam.size:
            self._chk(self._L.ipcgpu_opt_get_friction_state(self.h, _dp(sc), _dp(lam)))
        return dict(fricDHat=sc[0], n_lagged=int(sc[1]), fric_iter=int(sc[2]), n_half_space_lagged=int(sc[3]), lam=lam)

    def friction_update(self, dHat, kappa):
        n = C.c_int()
        self._chk(self._L.ipcgpu_friction_update(self.h, C.c_double(dHat), C.c_double(kappa), C.byref(n)))
        lam, co, ba = np.zeros(n.value), np.zeros((n.value, 2)), np.zeros((n.value, 6))
        if n.value:
            self._chk(self._L.ipcgpu_friction_get(self.h, _dp(lam), _dp(co), _dp(ba)))
        return dict(lam=lam, coord=co, basis=ba)

    def friction_energy(self, Vt, eps2, coef):
        Vt = np.asfortranarray(Vt, dtype=np.float64)
        E = C.c_double()
        self._chk(self._L.ipcgpu_friction_energy(self.h, _dp(Vt), C.c_double(eps2), C.c_double(coef), C.byref(E)))
        return E.value

    def friction_gradient_add(self, Vt, eps2, coef, grad=None):
        Vt = np.asfortranarray(Vt, dtype=np.float64)
        g = np.zeros(3 * self.nV) if grad is None else _f64(grad).copy()
        self._chk(self._L.ipcgpu_friction_gradient_add(self.h, _dp(Vt), C.c_double(eps2), C.c_double(coef), _dp(g)))
        return g

    def friction_hessian_add(self, Vt, eps2, coef, projectDBC=True):
        Vt = np.asfortranarray(Vt, dtype=np.float64)
        self._chk(self._L.ipcgpu_friction_hessian_add(self.h, _dp(Vt), C.c_double(eps2), C.c_double(coef), C.c_int(int(projectDBC))))

    def set_time_integration(self, name, beta=0.25, gamma=0.5):
        """Scene-script `timeIntegration BE | NM beta gamma`."""
        self._chk(self._L.ipcgpu_opt_set_time_integration(self.h, C.c_int({"BE": 0, "NM": 1}[name]), C.c_double(beta), C.c_double(gamma)))

    def set_warm_start(self, option):
        self._chk(self._L.ipcgpu_opt_set_warm_start(self.h, C.c_int(int(option))))

    def warm_step(self):
        v = C.c_double()
        self._chk(self._L.ipcgpu_opt_get_warm_step(self.h, C.byref(v)))
        return v.value

    def add_dirichlet(self, ids, lin_vel=(0, 0, 0), ang_vel_deg=(0, 0, 0), t0=0.0, t1=float("inf")):
        """One `DBC bboxMin bboxMax linVel angVel [t0 t1]` entry of a shape line (degrees per second, as in the script)."""
        ids = _i32(ids)
        lin = _f64(np.asarray(lin_vel, dtype=np.float64))
        ang = _f64(np.asarray(ang_vel_deg, dtype=np.float64) * np.pi / 180)
        self._chk(self._L.ipcgpu_opt_add_dirichlet(self.h, C.c_int(len(ids)), _ip(ids), _dp(lin), _dp(ang), C.c_double(t0), C.c_double(t1)))

    def set_dirichlet_targets(self, group, targets):
        """mesh-sequence motion: the positions the group's nodes are to reach in the next time step (None ends the sequence)"""
        if targets is None:
            self._chk(self._L.ipcgpu_opt_set_dirichlet_targets(self.h, C.c_int(group), C.c_int(0), None))
            return
        t = _f64(np.ascontiguousarray(np.asarray(targets, dtype=np.float64).reshape(-1, 3)))
        self._chk(self._L.ipcgpu_opt_set_dirichlet_targets(self.h, C.c_int(group), C.c_int(t.shape[0]), _dp(t)))

    def set_dirichlet_motion(self, group, lin_vel=(0, 0, 0), ang_vel_deg=(0, 0, 0), center=None, force_nonzero=True):
        """Motion of Dirichlet group `group` for the coming time steps (the rule-driven scripts of AnimScripter.cpp:1961-2135)."""
        lin = _f64(np.asarray(lin_vel, dtype=np.float64))
        ang = _f64(np.asarray(ang_vel_deg, dtype=np.float64) * np.pi / 180)
        ctr = None if center is None else _f64(np.asarray(center, dtype=np.float64))
        self._chk(self._L.ipcgpu_opt_set_dirichlet_motion(self.h, C.c_int(group), _dp(lin), _dp(ang), None if ctr is None else _dp(ctr),
                                                          C.c_int(int(force_nonzero))))

    def end_dirichlet(self, group, t_end):
        """Free the vertices of Dirichlet group `group` from the time step starting at t_end on (scripts that let go of a handle)."""
        self._chk(self._L.ipcgpu_opt_end_dirichlet(self.h, C.c_int(group), C.c_double(t_end)))

    def add_neumann(self, ids, accel, t0=0.0, t1=float("inf")):
        """One `NBC bboxMin bboxMax force [t0 t1]` entry of a shape line."""
        ids = _i32(ids)
        a = _f64(np.asarray(accel, dtype=np.float64))
        self._chk(self._L.ipcgpu_opt_add_neumann(self.h, C.c_int(len(ids)), _ip(ids), _dp(a), C.c_double(t0), C.c_double(t1)))

    def dbc_state(self):
        out = np.zeros(4)
        self._chk(self._L.ipcgpu_opt_get_dbc_state(self.h, _dp(out)))
        return dict(completed=out[0], rho=out[1], projectDBC=bool(out[2]), n_targets=int(out[3]))

    def kinematics(self):
        vel, acc, dx = np.zeros(3 * self.nV), np.zeros(3 * self.nV), np.zeros(3 * self.nV)
        self._chk(self._L.ipcgpu_opt_get_kinematics(self.h, _dp(vel), _dp(acc), _dp(dx)))
        return dict(velocity=vel, acceleration=acc, dx_Elastic=dx)

    def save_status(self, path):
        self._chk(self._L.ipcgpu_opt_save_status(self.h, str(path).encode()))

    def load_status(self, path):
        self._chk(self._L.ipcgpu_opt_load_status(self.h, str(path).encode()))

    def set_velocity(self, vel):
        vel = _f64(np.asarray(vel).reshape(-1))
        assert vel.size == 3 * self.nV
        self._chk(self._L.ipcgpu_opt_set_velocity(self.h, _dp(vel)))

    def contact_state(self):
        cnt = np.zeros(6, dtype=np.int32)
        pr = np.zeros(2, dtype=np.int32)
        self._chk(self._L.ipcgpu_opt_get_contact_state(self.h, _ip(cnt), _ip(pr)))
        return dict(nActive=int(cnt[0]), nPara=int(cnt[1]), nCand=int(cnt[2]), nHalfSpace=int(cnt[3]), nFullCCD=int(cnt[4]),
                    nPatternChanges=int(cnt[5]), ccdPair=(int(pr[0]), int(pr[1])))

    def timers(self):
        t = np.zeros(16)
        self._chk(self._L.ipcgpu_opt_get_timers(self.h, _dp(t)))
        return t

    # ---- measurement
    def bench_assembly(self, dtSq, reps=20):
        ms, by = C.c_double(), C.c_double()
        self._chk(self._L.ipcgpu_bench_assembly(self.h, C.c_double(dtSq), C.c_int(reps), C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def bench_factor_solve(self, reps=3):
        f, s = C.c_double(), C.c_double()
        self._chk(self._L.ipcgpu_bench_factor_solve(self.h, C.c_int(reps), C.byref(f), C.byref(s)))
        return f.value, s.value

    def bench_stream(self, nbytes=1 << 30, reps=10):
        g = C.c_double()
        self._chk(self._L.ipcgpu_bench_stream(self.h, C.c_longlong(nbytes), C.c_int(reps), C.byref(g)))
        return g.value
