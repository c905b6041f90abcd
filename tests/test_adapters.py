"""The drop-in boundary as C++ classes: include/adapters/*.hpp compiled against Eigen-free stand-ins of the reference's
LinSysSolver.hpp / Energy.hpp / Mesh.hpp (tests/mock_ipc/, test infrastructure) and, on a GPU, exercised: the
Diagnostic.cpp:367-392 known answer through the adapter class, and the composition the reference relies on --
Energy::computeHessian adding into the LinSysSolver it is handed (Energy.hpp:52-58) with host-side addCoeff / setCoeff on top."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "adapters", "_build")
EXE = os.path.join(BUILD, "test_adapters")


def build_exe():
    from ipc_amd import build as b
    b.build()
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "adapters", "test_adapters.cpp")
    deps = [src] + [os.path.join(ROOT, "include", "adapters", f) for f in os.listdir(os.path.join(ROOT, "include", "adapters"))]
    deps += [os.path.join(ROOT, "tests", "mock_ipc", f) for f in ("LinSysSolver.hpp", "Energy.hpp", "Mesh.hpp", "Types.hpp")]
    deps += [os.path.join(ROOT, "include", "ipcgpu.h"), os.path.join(ROOT, "ipc_amd", "libipcgpu.so")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "mock_ipc"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "adapters"), src, "-o", EXE, "-L" + os.path.join(ROOT, "ipc_amd"), "-lipcgpu",
           "-Wl,-rpath," + os.path.join(ROOT, "ipc_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EXE


REF_SRC = "/root/reference/src"
EXE_REF = os.path.join(BUILD, "test_adapters_ref")
LIB_REF = os.path.join(ROOT, "oracle", "_ref", "libipcref.so")


def build_exe_ref():
    """The same translation unit against the reference's OWN LinSysSolver.hpp / Energy.hpp / Mesh.hpp (the dense-matrix stand-in of
    oracle/refshim in place of Eigen), linked with the reference's compiled sources (oracle/_ref/libipcref.so: Mesh<3>,
    Energy<3>, LinSysSolver::create).  Build container only; the executable travels to the GPU box like the other built files.
    LinSysSolverType::HIP is the one enum value a maintainer adds: the unmodified tree is given CHOLMOD's in its place."""
    from ipc_amd import build as b
    b.build()
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "adapters", "test_adapters.cpp")
    deps = [src, LIB_REF, os.path.join(ROOT, "include", "ipcgpu.h")] + [os.path.join(ROOT, "include", "adapters", f) for f in os.listdir(os.path.join(ROOT, "include", "adapters"))]
    if os.path.exists(EXE_REF) and all(os.path.getmtime(EXE_REF) >= os.path.getmtime(d) for d in deps):
        return EXE_REF
    inc = ["-I" + os.path.join(ROOT, "oracle", "refshim")] + ["-I" + os.path.join(REF_SRC, d) for d in
           ("", "Utils", "CollisionObject", "Energy", "Energy/Physics_Elasticity", "LinSysSolver", "Utils/SVD", "TimeStepper")]
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-DDIM=3", "-DNDEBUG", "-DIPCGPU_LINSYSSOLVER_TYPE=LinSysSolverType::CHOLMOD"] + inc + [
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "adapters"), src, "-o", EXE_REF,
           "-L" + os.path.join(ROOT, "ipc_amd"), "-lipcgpu", "-L" + os.path.dirname(LIB_REF), "-lipcref",
           "-Wl,-rpath,$ORIGIN/../../../ipc_amd", "-Wl,-rpath,$ORIGIN/../../../oracle/_ref", "-Wl,-rpath,$ORIGIN/../../../oracle/_build", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EXE_REF


MAIN_HIP = os.path.join(BUILD, "ipc_main_hip")
REF_OBJ = os.path.join(ROOT, "oracle", "_ref", "obj")


def _ref_includes():
    return ["-I" + os.path.join(ROOT, "oracle", "refshim")] + ["-I" + os.path.join(REF_SRC, d) for d in
            ("", "Utils", "CollisionObject", "Energy", "Energy/Physics_Elasticity", "LinSysSolver", "Utils/SVD", "TimeStepper")]


def build_main_hip():
    """tests/adapters/_build/ipc_main_hip: the reference's OWN main.cpp (compiled where it lies, tests/adapters/main_hook.hpp
    pre-included: the one `new Optimizer` of main.cpp:1397 becomes `new HipOptimizer`) + include/adapters/HipOptimizer.hpp +
    the reference's other compiled sources (the objects oracle/Makefile.ref built) + the factory line of
    tests/adapters/main_hip_plug.cpp.  Round 5: the reference's Optimizer.cpp is compiled once more for this executable, UNCHANGED, with
    tests/adapters/optimizer_hook.hpp pre-included -- its 44 `SelfCollisionHandler<dim>::` call sites then reach the statics of
    include/adapters/HipSelfCollisionHandler.hpp (device when HipOptimizer switched them on, the reference's own code otherwise).
    Build container only; the executable travels to the GPU box."""
    from ipc_amd import build as b
    b.build()
    os.makedirs(BUILD, exist_ok=True)
    hook = os.path.join(ROOT, "tests", "adapters", "main_hook.hpp")
    plug = os.path.join(ROOT, "tests", "adapters", "main_hip_plug.cpp")
    ctcd = os.path.join(ROOT, "oracle", "ref_plug.cpp")
    opt_hook = os.path.join(ROOT, "tests", "adapters", "optimizer_hook.hpp")
    deps = [hook, opt_hook, plug, ctcd, LIB_REF, os.path.join(ROOT, "include", "ipcgpu.h"), os.path.join(REF_SRC, "main.cpp"), os.path.join(REF_SRC, "TimeStepper", "Optimizer.cpp")]
    deps += [os.path.join(ROOT, "include", "adapters", f) for f in os.listdir(os.path.join(ROOT, "include", "adapters"))]
    if os.path.exists(MAIN_HIP) and all(os.path.getmtime(MAIN_HIP) >= os.path.getmtime(d) for d in deps):
        return MAIN_HIP
    # the flags of oracle/Makefile.ref (no FMA contraction: what the reference computes on a machine without FMA)
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-w", "-DDIM=3", "-DNDEBUG", "-DIPC_DEFAULT_LINSYSSOLVER=LinSysSolverType::CHOLMOD",
             "-DIPC_WITH_CHOLMOD", "-DIPCGPU_LINSYSSOLVER_TYPE=LinSysSolverType::CHOLMOD"] + _ref_includes() + [
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "adapters")]
    objs = []
    jobs = [(os.path.join(REF_SRC, "main.cpp"), "main_hip.o", ["-Dmain=ipc_reference_main", "-include", hook]),
            (os.path.join(REF_SRC, "TimeStepper", "Optimizer.cpp"), "optimizer_hip.o", ["-include", opt_hook]),
            (plug, "main_hip_plug.o", []), (ctcd, "ctcd_plug.o", ["-DIPCREF_PLUG_CTCD_ONLY"])]
    procs = []
    for src, name, extra in jobs:
        o = os.path.join(BUILD, name)
        objs.append(o)
        procs.append(subprocess.Popen(["g++"] + flags + extra + ["-c", src, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for pr in procs:
        out = pr.communicate()[0]
        assert pr.returncode == 0, out[-6000:]
    for dirpath, _d, files in os.walk(REF_OBJ):  # the reference's other translation units, as compiled for libipcref.so
        objs += [os.path.join(dirpath, f) for f in files if f.endswith(".o") and f not in ("main.o", "ref_plug.o", "ref_api.o", "Optimizer.o")]
    cmd = ["g++", "-o", MAIN_HIP] + objs + ["-L" + os.path.join(ROOT, "ipc_amd"), "-lipcgpu", "-L" + os.path.join(ROOT, "oracle", "_build"), "-lorc",
           "-lstdc++fs", "-Wl,-rpath,$ORIGIN/../../../ipc_amd", "-Wl,-rpath,$ORIGIN/../../../oracle/_build", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    return MAIN_HIP


@pytest.mark.skipif(not (os.path.isdir(REF_SRC) and os.path.exists(LIB_REF)), reason="the reference's headers exist in the build container only")
def test_adapters_compile_and_link_against_the_reference_headers():
    exe = build_exe_ref()
    r = subprocess.run([exe, "compile-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_adapters_compile_and_link_against_the_interface_stand_ins():
    exe = build_exe()
    r = subprocess.run([exe, "compile-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_adapter_headers_use_only_the_public_c_abi():
    """The adapters are what a maintainer compiles inside the reference tree: nothing but ipcgpu.h and the reference's own headers."""
    for f in os.listdir(os.path.join(ROOT, "include", "adapters")):
        txt = open(os.path.join(ROOT, "include", "adapters", f)).read()
        if f == "ipcgpu_rccl.cpp":  # the caller-side RCCL binding: the C ABI, HIP runtime and RCCL, nothing of the library's insides
            incs = [ln.split()[1].strip('<>"') for ln in txt.splitlines() if ln.startswith("#include")]
            assert set(incs) <= {"ipcgpu_rccl.h", "hip/hip_runtime.h", "rccl/rccl.h", "cstring", "map", "mutex", "string", "vector"}, incs
            assert "csrc" not in txt and "oracle" not in txt
            continue
        for line in txt.splitlines():
            if line.startswith("#include"):
                inc = line.split()[1].strip('<>"')
                assert inc in ("LinSysSolver.hpp", "Energy.hpp", "Optimizer.hpp", "HalfSpace.hpp", "SelfCollisionHandler.hpp", "HipLinSysSolver.hpp", "HipElasticEnergy.hpp",
                               "HipSelfCollisionHandler.hpp", "ipcgpu.h",
                               "algorithm", "array", "cmath", "cstdio", "cstdlib", "cstring", "limits", "memory", "stdexcept", "string", "vector"), (f, inc)
        assert "oracle" not in txt and "hip/hip_runtime" not in txt


@pytest.mark.gpu
def test_adapters_run_on_the_gpu():
    exe = build_exe()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adapters ok" in r.stdout


@pytest.mark.gpu
def test_adapters_built_on_the_reference_headers_run_on_the_gpu():
    """tests/adapters/_build/test_adapters_ref: HipLinSysSolver / HipElasticEnergy deriving from the reference's own LinSysSolver /
    Energy<3>, on a Mesh<3> of the reference's own Mesh.cpp -- built where /root/reference exists (__graft_entry__.build)."""
    if not (os.path.exists(EXE_REF) and os.path.exists(LIB_REF)):
        pytest.skip("built in the container that holds /root/reference")
    r = subprocess.run([EXE_REF], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adapters ok" in r.stdout
