"""Build libipcgpu.so (hipcc, gfx950 only) in-tree: ipc_amd/libipcgpu.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container; the resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libipcgpu.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

SOURCES = ["nh_kernels.hip", "patch_assembly.hip", "hip_contact.hip", "hip_halfspace.hip", "mf_numeric.hip", "mf_sweeps.hip", "mf_exchange.hip", "hip_linsys.hip", "mf_symbolic.cpp", "hip_mesh.cpp",
           "hip_optimizer.cpp", "msh_io.cpp", "capi.cpp"]
FMA_OK = {"nh_kernels.hip", "patch_assembly.hip", "mf_numeric.hip", "mf_sweeps.hip", "mf_exchange.hip"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", f"-I{ROCM}/include", f"-I{os.path.join(HERE, '..', 'include')}"]


def _needs(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=(), fma: bool = True) -> str:
    """variant / extra_flags: experiment builds (tools/): objects under _obj_<variant>, library libipcgpu_<variant>.so,
    selected at load time with IPCGPU_LIB_VARIANT=<variant>.  The product build is the default (no variant).
    fma=False compiles EVERY file with -ffp-contract=off (the parity study of tools/gpu_nofma_study.py)."""
    global OBJ, LIB
    obj0, lib0 = OBJ, LIB
    if variant:
        OBJ = os.path.join(HERE, "_obj_" + variant)
        LIB = os.path.join(HERE, f"libipcgpu_{variant}.so")
    try:
        return _build(force, verbose, list(extra_flags), fma)
    finally:
        OBJ, LIB = obj0, lib0


def _build(force, verbose, extra, fma=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "ipcgpu.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or _needs(o, [s] + headers):
            # FMA contraction: on for the fp64 throughput kernels (parity there is a 1e-10 tolerance), off for
            # files that hold exact-comparison predicates (contact typing) and for host code
            contract = "fast" if (fma and src in FMA_OK) else "off"
            cmd = [HIPCC] + FLAGS + extra + [f"-ffp-contract={contract}"] + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [
            f"-L{ROCM}/lib", "-lrocblas", "-lrocsolver", f"-Wl,-rpath,{ROCM}/lib"]
        run(link)
    # the caller-side RCCL binding (include/adapters/ipcgpu_rccl.cpp): its own small library, so that libipcgpu.so itself never links RCCL
    rsrc = os.path.join(HERE, "..", "include", "adapters", "ipcgpu_rccl.cpp")
    rlib = LIB.replace("libipcgpu", "libipcgpu_rccl", 1) if "libipcgpu_" not in os.path.basename(LIB) else LIB.replace(".so", "_rccl.so")
    if force or _needs(rlib, [rsrc, os.path.join(HERE, "..", "include", "ipcgpu_rccl.h"), LIB]):
        run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", f"-I{ROCM}/include", f"-I{os.path.join(HERE, '..', 'include')}",
             "-x", "hip", rsrc, "-o", rlib, f"-L{os.path.dirname(LIB)}", "-l" + os.path.basename(LIB)[3:-3], f"-L{ROCM}/lib", "-lrccl",
             "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM}/lib"])
    return LIB


if __name__ == "__main__":
    # python ipc_amd/build.py [--force] [--nofma] [--variant NAME -DMACRO=VALUE ...]: experiment builds land in libipcgpu_NAME.so (IPCGPU_LIB_VARIANT=NAME)
    if "--nofma" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, variant="nofma", fma=False))
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build(force="--force" in sys.argv, verbose=True, variant=sys.argv[i + 1], extra_flags=[a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
