"""Synthetic regular-tet scenes (test / bench inputs, host side only).

The reference sizes its paper scenes from Gmsh files that are partly missing
(`input/tetMeshes/mat150x150t40.msh`).  These
generators build the stand-ins named in SURVEY.md 8(d): Kuhn-split regular
boxes (6 tets per cube, all sharing the cube's (0,0,0)-(1,1,1) diagonal, the
same split as `input/tetMeshes/cube.msh`), the x-extent Dirichlet handles of
`IglUtils::findBorderVerts` (`src/Utils/IglUtils.cpp:671-689`) and the `twist`
script kinematics (`src/AnimScripter.cpp:555-572,1674-1684`).

Everything here is numpy; nothing touches the GPU or the oracle.
"""
from __future__ import annotations

import itertools

import numpy as np

SEED = 20200707


def make_box(ncx: int, ncy: int, ncz: int, size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0)):
    """Kuhn-split box with ncx*ncy*ncz cubes.

    Returns V (nV,3) float64, F (nT,4) int32 with positive rest volume
    det[X1-X0, X2-X0, X3-X0] > 0 (`Mesh.cpp:440-455` needs triArea > 0).
    Node id = ix + (ncx+1)*(iy + (ncy+1)*iz).
    """
    nx, ny, nz = ncx + 1, ncy + 1, ncz + 1
    xs = origin[0] + size[0] * np.arange(nx) / max(ncx, 1)
    ys = origin[1] + size[1] * np.arange(ny) / max(ncy, 1)
    zs = origin[2] + size[2] * np.arange(nz) / max(ncz, 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    V = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float64)

    def nid(ix, iy, iz):
        return ix + nx * (iy + ny * iz)

    cz, cy, cx = np.meshgrid(np.arange(ncz), np.arange(ncy), np.arange(ncx), indexing="ij")
    cx, cy, cz = cx.ravel(), cy.ravel(), cz.ravel()
    tets = []
    eye = np.eye(3, dtype=np.int64)
    for perm in itertools.permutations(range(3)):
        o = np.zeros((3,), dtype=np.int64)
        corners = [o.copy()]
        for ax in perm:
            o = o + eye[ax]
            corners.append(o.copy())
        # parity of the permutation decides orientation
        sign = np.linalg.det(np.stack([corners[1] - corners[0], corners[2] - corners[0],
                                       corners[3] - corners[0]], axis=1).astype(float))
        if sign < 0:
            corners[1], corners[2] = corners[2], corners[1]
        cols = [nid(cx + c[0], cy + c[1], cz + c[2]) for c in corners]
        tets.append(np.stack(cols, axis=1))
    F = np.stack(tets, axis=1).reshape(-1, 4).astype(np.int32)  # 6 tets of a cube adjacent
    return V, F


def make_mat(n: int, thickness_ratio: float = 0.013):
    """`matN`: N x N x 2 node sheet on [-0.5,0.5] x [-t/2,t/2] x [-0.5,0.5]
    (spacing 1/(N-1), thickness 0.013), 6(N-1)^2 tets."""
    V, F = make_box(n - 1, 1, n - 1, size=(1.0, thickness_ratio, 1.0),
                    origin=(-0.5, -thickness_ratio / 2, -0.5))
    return V, F


def make_mat_stack(n: int, layers: int = 2, gap: float = 1.0e-3, thickness_ratio: float = 0.013, shift: float = 0.37):
    """`layers` matN sheets stacked along y with `gap` between them, each shifted sideways by a fraction of the grid
    spacing ("stack of k mats with gaps < sqrt(dHat) for a large active set").
    Returns V, F and the node count of one sheet."""
    Vs, Fs = [], []
    h = 1.0 / (n - 1)
    for k in range(layers):
        V, F = make_mat(n, thickness_ratio)
        V = V + np.array([shift * h * k, k * (thickness_ratio + gap), 0.61 * shift * h * k])
        Fs.append(F + sum(v.shape[0] for v in Vs))
        Vs.append(V)
    return np.vstack(Vs), np.vstack(Fs).astype(np.int32), Vs[0].shape[0]


def make_bar(ncx=20, ncy=2, ncz=2, size=(10.0, 0.5, 1.0)):
    """Hello-world bar of SURVEY.md 8(d) config 1 (480 tets by default)."""
    return make_box(ncx, ncy, ncz, size=size, origin=(-size[0] / 2, -size[1] / 2, -size[2] / 2))


def jitter(V, F, rel=1e-3, seed=SEED):
    """Seeded uniform jitter of +-rel*h so that no SVD is degenerate."""
    rng = np.random.default_rng(seed)
    e = V[F[:, 1]] - V[F[:, 0]]
    h = np.sqrt((e * e).sum(1)).min()
    return V + rng.uniform(-rel * h, rel * h, size=V.shape)


def border_verts(V, ratio=0.01):
    """`IglUtils::findBorderVerts` (`IglUtils.cpp:671-689`): two handle sets at the x extremes."""
    lo, hi = V[:, 0].min(), V[:, 0].max()
    rng = hi - lo
    left = np.nonzero(V[:, 0] < lo + rng * ratio)[0]
    right = np.nonzero(V[:, 0] > hi - rng * ratio)[0]
    return left.astype(np.int32), right.astype(np.int32)


def rot_x(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def twist_state(V, angle_per_unit_x, center=None):
    """Analytic twisted configuration: rotate the cross-section at x by
    theta(x) = angle_per_unit_x * (x - xc) about the x axis through `center`."""
    if center is None:
        center = 0.5 * (V.min(0) + V.max(0))
    th = angle_per_unit_x * (V[:, 0] - center[0])
    c, s = np.cos(th), np.sin(th)
    y = V[:, 1] - center[1]
    z = V[:, 2] - center[2]
    out = V.copy()
    out[:, 1] = center[1] + c * y - s * z
    out[:, 2] = center[2] + s * y + c * z
    return out


def surface_tris(F):
    """Boundary faces of a tet mesh with outward orientation (each face that
    belongs to exactly one tet).  Mirrors what `igl::boundary_facets` gives the
    reference (`IglUtils.cpp:206-227`); order is by (tet, local face)."""
    # local faces opposite to vertex k, oriented outward for positive tets
    loc = np.array([[1, 2, 3], [0, 3, 2], [0, 1, 3], [0, 2, 1]])
    # for a positive tet (det[x1-x0,x2-x0,x3-x0]>0) the outward face opposite v0 is (1,2,3)? check sign numerically later
    faces = F[:, loc].reshape(-1, 3)
    key = np.sort(faces, axis=1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    keep = cnt[inv.ravel()] == 1
    return faces[keep].astype(np.int32)


def surface_edges(SF):
    """Unique surface edges following `Mesh::computeFeatures` (`Mesh.cpp:495-511`):
    an edge (a,b) of a triangle is kept unless (b,a) was already inserted; the
    result is the sorted content of the std::set."""
    s = set()
    for t in SF:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            if (int(b), int(a)) not in s:
                s.add((int(a), int(b)))
    return np.array(sorted(s), dtype=np.int32).reshape(-1, 2)


def select_dirichlet(V, SF, rel_min, rel_max):
    """IglUtils::Init_Dirichlet (IglUtils.cpp:691-718): surface nodes inside the box given relative to the shape's bounding box."""
    lo, hi = V.min(0), V.max(0)
    rmin = (hi - lo) * np.asarray(rel_min, dtype=float) + lo
    rmax = (hi - lo) * np.asarray(rel_max, dtype=float) + lo
    on_boundary = np.zeros(V.shape[0], dtype=bool)
    on_boundary[np.asarray(SF).reshape(-1)] = True
    sel = on_boundary & (V >= rmin).all(1) & (V <= rmax).all(1)
    return np.nonzero(sel)[0].astype(np.int32)
