"""Host-only tool: flops of the multifrontal factorisation by kernel class (fused fronts / 32-column steps / Schur complements) and the chain of
dependent steps, for the headline workload (matTwist mat<n>).  usage: python tools/mf_flops.py [n=150]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from ipc_amd import scene  # noqa: E402
from oracle import orc  # noqa: E402  (pattern only)


def lib():
    so = os.path.join(HERE, "_build", "libmfflops.so")
    srcs = [os.path.join(HERE, "mf_flops.cpp"), os.path.join(ROOT, "ipc_amd", "csrc", "mf_symbolic.cpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread"] + srcs + ["-o", so])
    return C.CDLL(so)


def main(n=150):
    V, F = scene.make_mat(n)
    m = orc.Mesh(V, F, YM=1e5, PR=0.4, density=1000.0)
    ia, ja = m.pattern()
    ia, ja = np.ascontiguousarray(ia, np.int32), np.ascontiguousarray(ja, np.int32)
    Vr = np.ascontiguousarray(V, np.float64)
    out, chain = np.zeros(8), np.zeros(64, np.int32)
    L = lib()
    nl = L.mf_flops_report(C.c_int(len(ia) - 1), ia.ctypes.data_as(C.c_void_p), ja.ctypes.data_as(C.c_void_p), Vr.ctypes.data_as(C.c_void_p), C.c_int(12), C.c_int(8),
                           out.ctypes.data_as(C.c_void_p), chain.ctypes.data_as(C.c_void_p), C.c_int(64))
    tot = out[1] + out[2] + out[3]
    print(f"mat{n}: {V.shape[0]} nodes, nnz(L) {int(out[6])}, total {out[0] / 1e9:.2f} GFLOP (classes sum {tot / 1e9:.2f})")
    print(f"  fused fronts   {int(out[4]):6d}  {out[1] / 1e9:7.3f} GFLOP")
    print(f"  big fronts     {int(out[5]):6d}  steps {out[2] / 1e9:7.3f} GFLOP, Schur complements {out[3] / 1e9:7.3f} GFLOP")
    print("  dependent 32-column steps per level:", chain[:nl].tolist(), "sum", int(chain[:nl].sum()))
    if "--levels" in sys.argv:
        sys.stdout.flush()
        L.mf_level_report(C.c_int(len(ia) - 1), ia.ctypes.data_as(C.c_void_p), ja.ctypes.data_as(C.c_void_p), Vr.ctypes.data_as(C.c_void_p), C.c_int(12), C.c_int(8))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 150)
