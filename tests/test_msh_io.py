"""Tet-mesh files through the C ABI (IglUtils::readTetMesh / saveTetMesh, IglUtils.cpp:300-361, 451-584).
Host-only code: runs without a GPU."""
import os

import numpy as np
import pytest

from ipc_amd import lib as gl
from ipc_amd import scene

REF_MESHES = "/root/reference/input/tetMeshes"


def _mesh():
    V, F = scene.make_box(3, 2, 2, size=(1.5, 1.0, 0.7), origin=(0.1, -0.2, 0.3))
    V = V + 1e-3 * np.random.default_rng(1).normal(size=V.shape)
    return V, F.astype(np.int32)


def _write_41(path, V, F, blocks=2):
    nV, nT = V.shape[0], F.shape[0]
    cut = [0, nV // 3, nV] if blocks == 2 else [0, nV]
    with open(path, "w") as f:
        f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
        f.write(f"$Nodes\n{len(cut) - 1} {nV} 1 {nV}\n")
        for b in range(len(cut) - 1):
            f.write(f"3 {b + 1} 0 {cut[b + 1] - cut[b]}\n")
            for v in range(cut[b], cut[b + 1]):
                f.write(f"{v + 1}\n")
            for v in range(cut[b], cut[b + 1]):
                f.write(" ".join(repr(float(x)) for x in V[v]) + "\n")
        f.write("$EndNodes\n")
        tcut = [0, nT // 2, nT]
        f.write(f"$Elements\n2 {nT} 1 {nT}\n")
        for b in range(2):
            f.write(f"3 {b + 1} 4 {tcut[b + 1] - tcut[b]}\n")
            for t in range(tcut[b], tcut[b + 1]):
                f.write(f"{t + 1} " + " ".join(str(int(x) + 1) for x in F[t]) + "\n")
        f.write("$EndElements\n")


def _write_22(path, V, F):
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % V.shape[0])
        for i, v in enumerate(V):
            f.write(f"{i + 1} " + " ".join(repr(float(x)) for x in v) + "\n")
        f.write("$EndNodes\n$Elements\n%d\n" % (F.shape[0] + 2))
        f.write("1 15 2 0 1 1\n")  # a point and a boundary triangle: lower-dimensional entities are skipped
        f.write("2 2 2 0 1 1 2 3\n")
        for t, e in enumerate(F):
            f.write(f"{t + 3} 4 2 0 1 " + " ".join(str(int(x) + 1) for x in e) + "\n")
        f.write("$EndElements\n")


def _write_40(path, V, F, SF=None):
    """The reference's own dialect (IglUtils.cpp:514-584)."""
    with open(path, "w") as f:
        f.write("$MeshFormat\n4 0 8\n$EndMeshFormat\n$Entities\n0 0 0 1\n$EndEntities\n")
        f.write(f"$Nodes\n1 {V.shape[0]}\n0 3 0 {V.shape[0]}\n")
        for i, v in enumerate(V):
            f.write(f"{i + 1} " + " ".join(repr(float(x)) for x in v) + "\n")
        f.write(f"$EndNodes\n$Elements\n1 {F.shape[0]}\n0 3 4 {F.shape[0]}\n")
        for t, e in enumerate(F):
            f.write(f"{t + 1} " + " ".join(str(int(x) + 1) for x in e) + "\n")
        f.write("$EndElements\n")
        if SF is not None:
            f.write(f"$Surface\n{SF.shape[0]}\n")
            for t in SF:
                f.write(" ".join(str(int(x) + 1) for x in t) + "\n")
            f.write("$EndSurface\n")


@pytest.mark.parametrize("dialect", ["4.1", "4.1-one-block", "2.2", "4.0", "4.0+surface"])
def test_dialects_read_back_bit_exact(tmp_path, dialect):
    V, F = _mesh()
    SF = scene.surface_tris(F)
    p = tmp_path / "m.msh"
    if dialect == "4.1":
        _write_41(p, V, F)
    elif dialect == "4.1-one-block":
        _write_41(p, V, F, blocks=1)
    elif dialect == "2.2":
        _write_22(p, V, F)
    elif dialect == "4.0":
        _write_40(p, V, F)
    else:
        _write_40(p, V, F, SF[::-1])
    V2, F2, SF2 = gl.read_tet_mesh(p)
    assert np.array_equal(V2, V) and np.array_equal(F2, F)
    assert np.array_equal(SF2, SF[::-1] if dialect == "4.0+surface" else SF)  # file's surface wins, else (tet, face) order


def test_save_then_read(tmp_path):
    V, F = _mesh()
    p = tmp_path / "out.msh"
    gl.save_tet_mesh(p, V, F)
    txt = open(p).read()
    assert txt.startswith("$MeshFormat\n4.1 0 8\n") and "$Surface" in txt
    V2, F2, SF2 = gl.read_tet_mesh(p)
    assert np.array_equal(V2, V) and np.array_equal(F2, F) and np.array_equal(SF2, scene.surface_tris(F))


def test_errors_are_reported_not_thrown(tmp_path):
    with pytest.raises(RuntimeError, match="cannot open"):
        gl.read_tet_mesh(tmp_path / "missing.msh")
    p = tmp_path / "bad.msh"
    open(p, "w").write("$MeshFormat\n4.1 1 8\n$EndMeshFormat\n")
    with pytest.raises(RuntimeError, match="binary"):
        gl.read_tet_mesh(p)
    V, F = _mesh()
    Fb = F.copy()
    Fb[3, 2] = V.shape[0] + 5
    _write_41(p, V, Fb)
    with pytest.raises(RuntimeError, match="does not exist"):
        gl.read_tet_mesh(p)


def _independent_parse_41(path):
    toks = open(path).read().split()
    i = toks.index("$Nodes") + 1
    nb, n = int(toks[i]), int(toks[i + 1])
    i += 4
    V = []
    for _ in range(nb):
        k = int(toks[i + 3])
        i += 4 + k
        V += [float(x) for x in toks[i:i + 3 * k]]
        i += 3 * k
    i = toks.index("$Elements") + 1
    nb = int(toks[i])
    i += 4
    T = []
    for _ in range(nb):
        k = int(toks[i + 3])
        i += 4
        for _ in range(k):
            T.append([int(x) - 1 for x in toks[i + 1:i + 5]])
            i += 5
    return np.array(V).reshape(-1, 3), np.array(T, dtype=np.int32)


@pytest.mark.skipif(not os.path.isdir(REF_MESHES), reason="the reference's meshes are only present in the build container")
@pytest.mark.parametrize("name,nV,nT", [("bar-2523", 886, 2523), ("mat20x20", 800, None), ("cube", 8, None), ("sphere1K", None, 6851), ("mat40x40", None, 9126)])
def test_reference_meshes(name, nV, nT):
    """Real paper-scene meshes (bar-2523 = 886 nodes / 2 523 tets, sphere1K = 6 851 tets, mat40x40 = 9 126 tets)."""
    p = os.path.join(REF_MESHES, name + ".msh")
    V, T, SF = gl.read_tet_mesh(p)
    assert nV is None or V.shape[0] == nV
    assert nT is None or T.shape[0] == nT
    Vi, Ti = _independent_parse_41(p)
    assert np.array_equal(V, Vi) and np.array_equal(T, Ti)
    # every element positively oriented (the simulation's det F > 0 requirement), closed surface
    e = V[T[:, 1:]] - V[T[:, :1]]
    assert (np.einsum("ij,ij->i", np.cross(e[:, 0], e[:, 1]), e[:, 2]) > 0).all()
    edges = np.concatenate([SF[:, [0, 1]], SF[:, [1, 2]], SF[:, [2, 0]]])
    fwd = set(map(tuple, edges.tolist()))
    assert all((b, a) in fwd for a, b in fwd)
