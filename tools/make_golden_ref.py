"""TEST INFRASTRUCTURE -- golden vectors taken from the REFERENCE ITSELF.

oracle/_ref/libipcref.so is the reference's own sources compiled from /root/reference (oracle/Makefile.ref).  This script
evaluates them on seeded inputs and writes what they return to tests/golden/ref_functions.npz (single functions and
mesh-level pieces of the Newton path) and tests/golden/ref_scene_<name>.npz (whole scene scripts run through the reference's
main.cpp / Optimizer.cpp in offline mode: positions after every time step, Newton iterations per step).  The fixtures carry
their inputs (meshes included), so tests/ can check the oracle on the CPU and the HIP library on the GPU box, where neither
/root/reference nor libipcref.so exists.

    make -C oracle -f Makefile.ref && python tools/make_golden_ref.py [functions] [scenes]

Build container only."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from ipc_amd import lib as gl  # noqa: E402  (the msh reader only; no GPU is touched)
from ipc_amd import scene_script as ss  # noqa: E402
from oracle import ref  # noqa: E402
import ref_compare as rc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF_ROOT = "/root/reference"


def stencil_inputs(rng, n):
    """Generic stencils plus the configurations the classification has to get right: feet on vertices / edges, parallel edges."""
    X = rng.standard_normal((n, 4, 3))
    X[n // 2:] *= rng.uniform(1e-3, 1e2, size=(n - n // 2, 1, 1))
    return X


def functions():
    rng = np.random.default_rng(20240925)
    out = {}
    # --- distances with gradient and Hessian (MeshCollisionUtils.hpp:156-2004) -------------------------------------------------
    X = stencil_inputs(rng, 64)
    for kind, name in enumerate(("PP", "PE", "PT", "EE")):
        n = (2, 3, 4, 4)[kind]
        d, g, H = np.zeros(len(X)), np.zeros((len(X), 3 * n)), np.zeros((len(X), 3 * n, 3 * n))
        for i, x in enumerate(X):
            d[i], g[i], H[i] = ref.stencil_distance(kind, x)
        out[f"dist_{name}_d"], out[f"dist_{name}_g"], out[f"dist_{name}_H"] = d, g, H
    out["dist_X"] = X
    # --- closest-feature classification (MeshCollisionUtils.hpp:2073-2210) and the classified distances ---------------------
    Xc = [rng.standard_normal((4, 3)) for _ in range(300)]
    for _ in range(100):  # point over a vertex / an edge / the interior of a unit triangle, edges in special positions
        t = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
        p = np.array([rng.choice([-0.5, 0.0, 0.25, 0.5, 1.0, 1.5]), rng.choice([-0.5, 0.0, 0.25, 0.5, 1.0, 1.5]), rng.choice([0.3, 1.0])])
        Xc.append(np.vstack([p, t]))
        e = np.array([[0, 0, 0], [1, 0, 0], [rng.choice([-1, 0, 0.5, 1, 2]), rng.choice([-1, 0.5, 1]), 0.7], [rng.choice([-1, 0, 0.5, 1, 2]), rng.choice([-1, 0.5, 1]), 0.7]], float)
        Xc.append(e)
    Xc = np.array(Xc)
    out["cls_X"] = Xc
    out["cls_pt"] = np.array([ref.dtype_pt(x) for x in Xc], np.int32)
    out["cls_ee"] = np.array([ref.dtype_ee(x) for x in Xc], np.int32)
    out["cls_dpt"] = np.array([ref.classified_distance(2, x) for x in Xc])
    out["cls_dee"] = np.array([ref.classified_distance(3, x) for x in Xc])
    # --- barrier (BarrierFunctions.hpp) and the edge-edge mollifier (MeshCollisionUtils.hpp:2409-2912) -------------------------
    dHat = 1.3e-3
    ds = dHat * np.concatenate([rng.uniform(1e-6, 1.0, 40), [0.5, 0.999999]])
    out["bar_dHat"], out["bar_d"] = dHat, ds
    out["bar_bgH"] = np.array([ref.barrier(d, dHat) for d in ds])
    Xm = rng.standard_normal((48, 4, 3))
    Xm[24:, 3] = Xm[24:, 2] + (Xm[24:, 1] - Xm[24:, 0]) * rng.uniform(0.5, 2.0, (24, 1)) + 1e-3 * rng.standard_normal((24, 3))  # nearly parallel
    eps_x = rng.uniform(1e-3, 1.0, 48)
    mo = [ref.ee_mollifier(x, e) for x, e in zip(Xm, eps_x)]
    out["mol_X"], out["mol_eps"] = Xm, eps_x
    for k, name in enumerate(("c", "cg", "cH", "e", "eg", "eH")):
        out[f"mol_{name}"] = np.array([m[k] for m in mo])
    # --- segment / triangle intersection (IglUtils.hpp:214-265, the branch the default build compiles) ----------------------
    Xs = [rng.standard_normal((5, 3)) for _ in range(400)]
    for _ in range(200):  # a segment through / beside / in the plane of the unit triangle
        a = np.array([rng.uniform(-0.2, 1.2), rng.uniform(-0.2, 1.2), rng.choice([0.5, 1e-9, 0.0])])
        b = np.array([rng.uniform(-0.2, 1.2), rng.uniform(-0.2, 1.2), rng.choice([-0.5, -1e-9, 0.0, 0.5])])
        Xs.append(np.vstack([a, b, [0, 0, 0], [1, 0, 0], [0, 1, 0]]))
    Xs = np.array(Xs)
    out["seg_X"], out["seg_hit"] = Xs, np.array([ref.seg_tri_intersect(x) for x in Xs], np.int8)
    # --- 3x3 SVD (AutoFlipSVD over ImplicitQRSVD.h:687-850) and makePD (IglUtils.hpp) -----------------------------------------
    Fs = [np.eye(3) + 0.3 * rng.standard_normal((3, 3)) for _ in range(60)]
    Fs += [np.diag([1.0, 1.0, 1.0]), np.diag([2.0, 1.0, 0.5]), -np.eye(3), np.diag([1.0, 1.0, -0.3]), np.zeros((3, 3)), np.outer([1, 2, 3], [0.5, -1, 2.0])]
    Fs = np.array(Fs)
    sv = [ref.svd3(F) for F in Fs]
    out["svd_F"], out["svd_U"], out["svd_s"], out["svd_V"] = Fs, np.array([s[0] for s in sv]), np.array([s[1] for s in sv]), np.array([s[2] for s in sv])
    for n in (6, 9, 12):
        A = rng.standard_normal((12, n, n))
        A = A + A.transpose(0, 2, 1)
        out[f"pd{n}_A"], out[f"pd{n}_P"] = A, np.array([ref.make_pd(a) for a in A])

    # --- mesh level: bar-186 of the reference, jittered and pre-strained ------------------------------------------------------------
    V, T, SF = gl.read_tet_mesh(os.path.join(REF_ROOT, "input/tetMeshes/bar-186.msh"))
    YM, PR, rho = 1e5, 0.4, 1000.0
    Vx = V * [1.05, 0.97, 1.02] + 5e-3 * rng.standard_normal(V.shape)
    m = ref.Mesh(V, T, SF, YM, PR, rho)
    f = m.features()
    SVI, SFE = m.surface()
    out.update(bar_V=V, bar_T=T, bar_SF=SF, bar_Vx=Vx, bar_YM=YM, bar_PR=PR, bar_rho=rho, bar_restTriInv=f["restTriInv"], bar_triArea=f["triArea"], bar_mass=f["mass"],
               bar_mu=f["mu"], bar_lam=f["lam"], bar_bbox2=f["bbox2"], bar_avgNodeMass=f["avgNodeMass"], bar_SVI=SVI, bar_SFEdges=SFE)
    dbc = np.where(V[:, 0] < V[:, 0].min() + 1e-9)[0].astype(np.int32)
    out["bar_dbc"] = dbc
    m.set_positions(Vx)
    m.set_dbc(dbc, 1)
    for kind, name in ((0, "NH"), (1, "FCR")):
        out[f"bar_E_{name}"] = m.elastic_energy(kind, 0.7)
        out[f"bar_g_{name}"] = m.elastic_gradient(kind, 0.7)
        ia, ja, a = m.elastic_hessian(kind, 0.7)
        out[f"bar_ia_{name}"], out[f"bar_ja_{name}"], out[f"bar_a_{name}"] = ia, ja, a
    ps = 0.3 * rng.standard_normal((6, 3 * V.shape[0]))
    out["bar_p"], out["bar_filter"] = ps, np.array([m.filter_step_size(p, 1.0, 0) for p in ps])

    # --- contact: three mat20x20 sheets of the reference -- a base, one standing on its rim above it (point-edge, point-triangle,
    # edge-edge pairs, merged duplicates, nearly parallel edges), one hanging corner-down beside the base's corner (point-point) ------
    Vm, Tm, SFm = gl.read_tet_mesh(os.path.join(REF_ROOT, "input/tetMeshes/mat20x20.msh"))
    thick = Vm[:, 1].max() - Vm[:, 1].min()
    gap = 2e-3

    def rot(axis, th):
        c, s_ = np.cos(th), np.sin(th)
        R = np.eye(3)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        R[i, i], R[i, j], R[j, i], R[j, j] = c, -s_, s_, c
        return R

    A = Vm @ rot(2, np.pi / 2).T + [0.3, 0.5 + thick * 0.5 + gap, 0.0513]
    B = Vm @ (rot(0, 0.6) @ rot(2, -0.7)).T
    k = B[:, 1].argmin()
    B += [-0.5 - 8e-4 - B[k, 0], thick * 0.5 + 8e-4 - B[k, 1], -0.5 - 8e-4 - B[k, 2]]
    nm, nf = len(Vm), len(SFm)
    Vc = np.vstack([Vm, A, B])
    Tc, SFc = np.vstack([Tm, Tm + nm, Tm + 2 * nm]), np.vstack([SFm, SFm + nm, SFm + 2 * nm])
    mc = ref.Mesh(Vc, Tc, SFc, 2e4, 0.4, 1000.0, node_ranges=[0, nm, 2 * nm, 3 * nm], sf_ranges=[0, nf, 2 * nf, 3 * nf])
    fc = mc.features()
    dHat = (2.5e-3) ** 2
    kappa = 1e4
    act, par, eiej, cs = mc.constraint_set(dHat)
    print("contact fixture:", len(act), "active,", len(par), "mollified,", len(cs), "PT/EE candidates")
    out.update(con_V=Vc, con_T=Tc, con_SF=SFc, con_nodeRanges=np.array([0, nm, 2 * nm, 3 * nm], np.int32), con_sfRanges=np.array([0, nf, 2 * nf, 3 * nf], np.int32),
               con_dHat=dHat, con_kappa=kappa, con_active=act, con_para=par, con_eiej=eiej, con_csPTEE=cs, con_bbox2=fc["bbox2"])
    out["con_E"] = mc.barrier_energy(dHat, kappa)
    out["con_g"] = mc.barrier_gradient(dHat, kappa)
    out["con_ia"], out["con_ja"], out["con_a"] = mc.barrier_hessian(dHat, kappa)
    pc = 1e-3 * rng.standard_normal((4, 3 * len(Vc)))
    pc[:, 3 * nm + 1::3] -= 2e-3  # the two upper sheets move into the base
    out["con_p"] = pc
    out["con_ccd_partial"] = np.array([mc.partial_ccd(p, 0.8, 1.0) for p in pc])
    out["con_ccd_full"] = np.array([mc.full_ccd(p, 0.8, 1.0) for p in pc])
    out["con_intersected"] = np.array([mc.is_intersected()], np.int8)
    Vi = Vc.copy()
    Vi[nm:2 * nm, 1] -= gap + 0.3 * thick  # the standing sheet pushed through the base
    mc.set_positions(Vi)
    out["con_Vi"], out["con_intersected_i"] = Vi, np.array([mc.is_intersected()], np.int8)
    mc.set_positions(Vc)
    # half-space under the lower sheet
    o, n = np.array([0.0, Vm[:, 1].min() - 1e-3, 0.0]), np.array([0.0, 1.0, 0.0])
    hact, hE, hg, (hia, hja, ha) = mc.halfspace(o, n, dHat, kappa)
    out.update(hs_o=o, hs_n=n, hs_active=hact, hs_E=hE, hs_g=hg, hs_a=ha, hs_step=np.array([mc.halfspace_step_bound(o, n, p, 0.9, 1.0) for p in pc]))
    # HalfSpace::move: the plane displaced towards the sheets (limited by the nearest surface node), along them, and away
    deltas = np.array([[0.0, 0.01, 0.0], [0.0, 2e-4, 0.0], [0.3, 0.004, -0.2], [0.0, -0.5, 0.1], [0.02, 0.0, 0.0]])
    moved = [ref.halfspace_move(mc, o, n, d, 0.5) for d in deltas]
    out.update(hs_move_delta=deltas, hs_move_origin=np.array([m_[0] for m_ in moved]), hs_move_left=np.array([m_[1] for m_ in moved]))
    np.savez_compressed(os.path.join(GOLD, "ref_functions.npz"), **out)
    print("wrote ref_functions.npz", os.path.getsize(os.path.join(GOLD, "ref_functions.npz")) >> 10, "KiB")


# scene scripts run through the reference's own main(): (name, path under the reference's input/, appended text, steps)
SCENES = [
    ("bar_twist", "otherExamples/barTwist_noCollisions.txt", "", 4),
    ("bar_twist_tight", "otherExamples/barTwist_noCollisions.txt", "\ntol 1\n1e-6\n", 3),
    ("two_cubes_fall", "tutorialExamples/2cubesFall.txt", "", 40),
    # a cube falling on a rotating kinematic cube given as a tetrahedral mesh / as a closed triangle surface (codimension 2)
    ("rotate_co", "tutorialExamples/MCO/2cubesFall_rotateCO.txt", "", 30),
    ("rotate_co_surface", "tutorialExamples/MCO/2cubesFall_rotateCO_closedSurface.txt", "", 30),
    # ... on the three EDGES of a rotating triangle (`.seg` shape, codimension 1) and on its three CORNERS (`.pt` shape, codimension 0)
    ("rotate_co_edges", "tutorialExamples/MCO/2cubesFall_rotateCO_edges.txt", "", 44),
    ("rotate_co_points", "tutorialExamples/MCO/2cubesFall_rotateCO_points.txt", "", 44),
    # ... and on the segments of a triangle that follows a MESH SEQUENCE (`meshSeq input/segMeshes/sequence`: one .seg file per time step)
    ("rotate_co_mesh_seq", "tutorialExamples/advanced/2cubesFall_rotateCO_meshSeq.txt", "", 25),  # (in step 26 the reference as compiled here runs into its iteration cap)
    # Dirichlet groups with time ranges
    ("dbc_time_range", "tutorialExamples/BC/2cubesFall_DBC_timeRange.txt", "", 30),
    # fixed-corotated energy, `size`, `script fall`, a kinematic mesh obstacle (meshCO plane.obj), self-collision
    ("aligned_cubes", "paperExamples/supplementB/SQPBenchmark/12_alignedCubes.txt", "", 30),
    # the same with friction between the cubes: the pairs that involve the mesh collision object carry none (its coefficient only
    # switches the lagging loop on in the reference), the cube-cube pairs carry selfFric
    ("aligned_cubes_fric", "paperExamples/supplementB/SQPBenchmark/12_alignedCubes.txt", "\nselfFric 0.3\n", 40),
    # lagged stiffness-proportional damping (dampingRatio): the twisting bar solved to its minimisers, and the reference's own
    # Newmark + damping tutorial scene
    ("bar_twist_damped", "otherExamples/barTwist_noCollisions.txt", "\ndampingRatio 0.5\ntol 1\n1e-6\n", 6),
    # `tuning 2`: a start value for kappa and the dHat homotopy of fullyImplicit_IP (dHat 0.5 of the diagonal halved down to 1e-3 in
    # every time step), FCR, `size`, `script fall`, a mesh collision object
    ("cubes_dhat_homotopy", "paperExamples/supplementB/SQPBenchmark/11_cubes.txt", "", 20),
    # `rotateModel` (start positions turned against the rest shape), `tuning 2` homotopy, warm start 1, point-triangle impact
    ("point_triangle_rotated", "paperExamples/supplementB/SQPBenchmark/04_pointTriangle.txt", "", 45),
    # `useAbsParameters` (dHat, its target, dTol and the Newton tolerance as ABSOLUTE lengths), `kappaMinMultiplier`, a fourth `tuning` entry
    ("point_triangle_abs_parameters", "paperExamples/supplementB/SQPBenchmark/04_pointTriangle.txt",
     "\nuseAbsParameters\nkappaMinMultiplier 3e10\ntuning 6\n0\n8e-2\n2e-3\n1e-8\n1e-3\n1e-3\ntol 1\n2e-2\n", 45),
    # the scene-wide `DBCTimeRange` / `NBCTimeRange` (Config.cpp:175-180): groups act where their own range and the scene's overlap
    ("dbc_global_time_range", "tutorialExamples/BC/2cubesFall_DBC_timeRange.txt", "\nDBCTimeRange 0.1 0.25\n", 14),
    ("nbc_global_time_range", "tutorialExamples/BC/2cubesFall_NBC.txt", "\nNBCTimeRange 0.1 0.2\ntol 1\n1e-5\n", 14),
    ("two_cubes_nm_damped", "tutorialExamples/advanced/2cubesFall_NM.txt", "\ntime 5 0.025\n", 36),  # every step written at this step size
]


# Restart fixtures: the reference run is continued from ITS OWN status file at step R (`restart <file>`, Optimizer.cpp:179-248) for K more
# steps, and that second run is what the fixture holds.  Both implementations then start from one and the same post-contact state --
# deformed, moving, constraints active -- so every step can be held to round-off instead of "the same minimiser within the Newton
# tolerance": the scenes above all touch down from exact rest (F = I), where IglUtils::makePD2d is decided by round-off.
RESTARTS = {"two_cubes_fall": (30, 8), "aligned_cubes": (20, 8), "aligned_cubes_fric": (24, 10), "cubes_dhat_homotopy": (14, 4),
            "two_cubes_nm_damped": (28, 8), "rotate_co": (20, 8), "sphere_rot_cylinders": (4, 4)}


# scene scripts written here (the reference's own input files do not exercise these scripts on a mesh small enough for a fixture)
_SQUASH6 = """energy FCR
warmStart 0
time 2 0.025
density 1000
stiffness 10000 0.4
script DCOSquash6
turnOffGravity

shapes input 7
input/triMeshes/plane.obj -1 1.5 -1.5  0 0 -90  3 3 3
input/triMeshes/plane.obj 1 1.5 -1.5  0 0 -90  3 3 3
input/triMeshes/plane.obj -1.5 -1 -1.5  0 0 0  3 3 3
input/triMeshes/plane.obj -1.5 1 -1.5  0 0 0  3 3 3
input/triMeshes/plane.obj -1.5 1.5 -1  90 0 0  3 3 3
input/triMeshes/plane.obj -1.5 1.5 1  90 0 0  3 3 3
input/tetMeshes/cube.msh %s

selfCollisionOn
constraintSolver interiorPoint
"""
_HANDLES = """energy NH
warmStart 0
time 1 0.025
density 1000
stiffness 1e5 0.4
script %s
%s
shapes input 1
input/tetMeshes/cube.msh 0 0 0  0 0 0  1 1 1
"""
_SEGBED = """energy NH
warmStart 0
time 1 0.025
density 1000
stiffness 1e5 0.4
script DCOSegBedSquash
turnOffGravity

shapes input 1
input/tetMeshes/cube.msh 0.07 0.09 0.07  0 0 0  0.14 0.14 0.14

shapeMatrix input 1 1 3  -0.08 0 0.03
input/segMeshes/edge.seg 1 1 0.04  0 0 0  0.3 0.3 0.3

shapeMatrix input 1 1 3  -0.08 0.3 0.03
input/segMeshes/edge.seg 1 1 0.04  0 0 0  0.3 0.3 0.3

selfCollisionOn
constraintSolver interiorPoint
"""
_BOXRULE = """energy NH
time 1 0.02
density 1000
stiffness 1e5 0.4
script %s
shapes input 1
%s
selfCollisionOff
tol 1
1e-4
%s
"""
_BAR, _MAT, _MAT_UPRIGHT = ("input/tetMeshes/bar-186.msh 0 0 0  0 0 0  1 1 1", "input/tetMeshes/mat20x20.msh 0 0 0  0 0 0  1 1 1",
                            "input/tetMeshes/mat20x20.msh 0 0 0  0 0 90  1 1 1")
_SEQ_FOLDER = "/tmp/ipc_amd_fixture_seq_from_file"  # written by write_obj_sequence() right before the reference runs


def write_obj_sequence(folder, n):
    """<folder>/1.obj .. n.obj: triangle.obj lifted to y = 1, turning about y through its centre by 4 degrees and rising by 0.004 per file"""
    os.makedirs(folder, exist_ok=True)
    Vt, Ft = ss.read_obj(os.path.join(REF_ROOT, "input/triMeshes/triangle.obj"))
    W0 = Vt + np.array([0.0, 1.0, 0.0])
    c = W0.mean(0)
    for k in range(1, n + 1):
        th = np.radians(4.0 * k)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        W = (W0 - c) @ R.T + c + np.array([0.0, 0.004 * k, 0.0])
        with open(os.path.join(folder, f"{k}.obj"), "w") as f:
            for v in W:
                f.write("v %.17g %.17g %.17g\n" % tuple(v))
            for t in Ft:
                f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))


INLINE = {
    # `script meshSeqFromFile <folder>` (AnimScripter.cpp:2126-2144): the tutorial's cube falls on a triangle (surface-only component) that follows
    # <folder>/<n>.obj, n = 1, 2, ...
    "inline:mesh_seq_from_file": "shapes input 2\ninput/tetMeshes/cube.msh 0 3 0  0 0 0  1 1 1\ninput/triMeshes/triangle.obj 0 1 0  0 0 0  1 1 1\n"
                                 "selfFric 0.1\nground 0.1 0\nscript meshSeqFromFile " + _SEQ_FOLDER + "\n",
    # more scripts that pick nodes by a box rule of the start positions: the top 1 % of an upright sheet held (ZERO); the x < 1 %, y < 1 % or z < 1 %
    # nodes of a bar held (NONZERO); the two ends of a bar pushed together at 0.03; the middle of the bottom tenth of a lifted sheet dragged down
    # at 1.5 through a ground plane's barrier; the left half of a bar starting at +1 in x
    "inline:script_hang2": _BOXRULE % ("hang2", _MAT_UPRIGHT, ""),
    "inline:script_corner": _BOXRULE % ("corner", _BAR, ""),
    "inline:script_squash": _BOXRULE % ("squash", _BAR, ""),
    "inline:script_dragdown": _BOXRULE % ("dragdown", _MAT, "ground 0.1 0"),
    "inline:script_left_hit_right": _BOXRULE % ("leftHitRight", _BAR, "turnOffGravity"),
    # scripts whose handles turn round or stop by a rule on one node: the ends of a bar going up / down at 1.8 and turning 0.6 from the start (step 8);
    # a twist at 0.4 pi with a pull at 0.9 that turns round; a twist at 0.1 pi with a steady pull; the top of a cube dragged at 5, turning 4 further left
    "inline:script_upndown": (_BOXRULE % ("upndown", _BAR, "")).replace("time 1 0.02", "time 1 0.05"),
    "inline:script_twistnsns_old": (_BOXRULE % ("twistnsns_old", _BAR, "")).replace("time 1 0.02", "time 1 0.05"),
    "inline:script_twistnstretch": (_BOXRULE % ("twistnstretch", _BAR, "")).replace("time 1 0.02", "time 1 0.1"),
    # the top of a bar drawn sideways and LET GO after 0.1 (step 5); two handle sets of an upright sheet dragged apart (12 to 37 Newton iterations per
    # step); both ends of a sheet pulled along x with its middle strip held; start positions times 1.5 (scaleF); the start turned inside out and
    # squeezed to a tenth about the held left end (stampInv, FCR: 548 Newton iterations in the first step)
    "inline:script_toggle_top": (_BOXRULE % ("toggleTop", _BAR, "")).replace("time 1 0.02", "time 1 0.05"),
    "inline:script_four_leg_pull": _BOXRULE % ("fourLegPull", _MAT_UPRIGHT, ""),
    "inline:script_head_tail_pull": _BOXRULE % ("headTailPull", _MAT, ""),
    "inline:script_scale_f": _BOXRULE % ("scaleF", _BAR, ""),
    "inline:script_stamp_inv": (_BOXRULE % ("stampInv", _BAR, "")).replace("energy NH", "energy FCR"),
    # `script DCOCut`: the second component (a triangle as the knife) moves at (0, -1, -1) while its lowest node is above 0.001, over a cube on the ground
    "inline:script_dco_cut": "energy NH\ntime 1 0.02\ndensity 1000\nstiffness 1e5 0.4\nscript DCOCut\nshapes input 2\ninput/tetMeshes/cube.msh 0 0.002 0  0 0 0  1 1 1\n"
                             "input/triMeshes/triangle.obj 0.2 1.6 0.9  0 0 0  1 1 1\nselfCollisionOn\nground 0.1 0\ntol 1\n1e-4\n",
    "inline:script_tear": (_BOXRULE % ("tear", "input/tetMeshes/cube.msh 0 0 0  0 0 0  1 1 1", "")).replace("time 1 0.02", "time 1 0.1"),
    # scripts that pick their handles from the bounding box of the mesh (AnimScripter::initAnimScript): the lower half of a cube held under
    # gravity; one corner node pushed in -x; the bottom held and the top pressed down by a Neumann acceleration
    "inline:fix_lower_half": _HANDLES % ("fixLowerHalf", ""),
    "inline:push_right_most": _HANDLES % ("pushRightMost1", "turnOffGravity"),
    "inline:utopia": _HANDLES % ("utopiaComparison", "turnOffGravity"),
    # `script DCOSegBedSquash` (17_pinCushionBall.txt's script): a small cube between two beds of three segments each, the upper bed coming
    # down at 1 until it is 0.1 above the lower one -- the cube is picked up, pressed onto the lower bed and squeezed by a quarter
    "inline:seg_bed_squash": _SEGBED,
    # 15_trashComp_shapes.txt's six closing plates (`script DCOSquash6`, FCR, no gravity) around one cube: a tiny one that is never
    # touched -- the plates close to 0.1, turn round (the sign flip of AnimScripter.cpp:2053-2074, once per step as written) and open --
    # and one that fills the box and is squeezed from step 16 on
    "inline:squash6_small": _SQUASH6 % "-0.025 -0.025 -0.025  0 0 0  0.05 0.05 0.05",
    "inline:squash6_contact": _SQUASH6 % "-0.6 -0.6 -0.6  0 0 0  1.2 1.2 1.2",
}
SCENES += [
    # `warmStart 5` (Optimizer::initX option 5, the Jacobi guess -g_i / H_ii): the twisting bar and the tutorial scene through its impacts
    ("bar_twist_warm5", "otherExamples/barTwist_noCollisions.txt", "\nwarmStart 5\n", 4),
    ("two_cubes_warm5", "tutorialExamples/2cubesFall.txt", "\nwarmStart 5\n", 24),
    ("fix_lower_half", "inline:fix_lower_half", "", 6),
    ("push_right_most", "inline:push_right_most", "", 6),
    ("utopia", "inline:utopia", "", 6),
    ("seg_bed_squash", "inline:seg_bed_squash", "", 16),
    # BASELINE configs[4]: 15_trashComp_shapes.txt as shipped -- six closing plates (`script DCOSquash6`, FCR, `size 1`) around a ball, a mat and a
    # bunny (46 K tets together) that touch the plates and each other from the first step on; 4 steps = 47 Newton iterations of the serial reference
    ("trash_compactor", "paperExamples/15_trashComp_shapes.txt", "", 4),
    # BASELINE configs[4], the other scene it names: videoExamples/chain10.txt as shipped -- ten interlocked tori (NH, E = 1e7) dropping onto a fixed
    # torus (meshCO), `script fallNoShift`: link after link is caught by the one above it
    ("chain10", "paperExamples/videoExamples/chain10.txt", "", 30),
    # more of the reference's shipped scenes, small ones: a block on a slope just below / at the friction angle (half-space friction, FCR, tol
    # 1e-4, fricIterAmt -1); the same with an eps_v homotopy (`tuning`'s sixth entry: 4e-3 halved down to 1e-3 between the friction-lag passes);
    # a stiff cube held by its lower half in a tight corner (fixLowerHalf, halfSpace, tuning); two mats dropped edge-on onto a board and a
    # half-space (flat sides oblique to the axes: the rank-revealing solve of segTriIntersect decides the intersection checks there)
    ("slope_049", "otherExamples/friction/slopeTest_highSchoolPhysics_0.49.txt", "", 16),
    ("slope_05", "otherExamples/friction/slopeTest_highSchoolPhysics_0.5.txt", "", 16),
    ("slope_epsv_homotopy", "otherExamples/friction/slopeTest_highSchoolPhysics_0.5.txt", "\ntuning 6\n0\n1e-3\n1e-3\n1e-9\n4e-3\n1e-3\n", 16),
    ("tight_fit_cube", "paperExamples/videoExamples/tightFitCube.txt", "", 6),
    ("mat_on_board", "paperExamples/12_matOnBoard.txt", "", 4),
    # a mat dropped on a bed of 210 segments / 420 points held by `script DCOFix` (coDimUnitTests); Neumann groups with time ranges; `attach`
    ("mat_on_segments", "otherExamples/coDimUnitTests/mat40x40_segPlaneDrop.txt", "", 4),
    ("mat_on_points", "otherExamples/coDimUnitTests/mat40x40_pointPlaneDrop.txt", "", 4),
    ("nbc_time_range", "tutorialExamples/BC/2cubesFall_NBC_timeRange.txt", "", 22),
    ("attach", "tutorialExamples/advanced/2cubesFall_attach.txt", "", 22),
    # 13_dolphinFunnel.txt as shipped: `script dragright` on a model that `rotateModel` turns at the start (lift and handle follow the START
    # positions), a funnel as mesh collision object, no gravity
    ("dolphin_funnel", "paperExamples/13_dolphinFunnel.txt", "", 3),
    ("script_hang2", "inline:script_hang2", "", 6),
    ("script_corner", "inline:script_corner", "", 6),
    ("script_squash", "inline:script_squash", "", 6),
    ("script_dragdown", "inline:script_dragdown", "", 6),
    ("script_left_hit_right", "inline:script_left_hit_right", "", 6),
    ("script_upndown", "inline:script_upndown", "", 12),
    ("script_twistnsns_old", "inline:script_twistnsns_old", "", 14),
    ("script_twistnstretch", "inline:script_twistnstretch", "", 8),
    ("script_tear", "inline:script_tear", "", 12),
    ("script_toggle_top", "inline:script_toggle_top", "", 10),
    ("script_four_leg_pull", "inline:script_four_leg_pull", "", 6),
    ("script_head_tail_pull", "inline:script_head_tail_pull", "", 6),
    ("script_scale_f", "inline:script_scale_f", "", 6),
    ("script_stamp_inv", "inline:script_stamp_inv", "", 6),
    ("mesh_seq_from_file", "inline:mesh_seq_from_file", "", 30),
    ("script_dco_cut", "inline:script_dco_cut", "", 20),
    # otherExamples/typical/sphere1K_DCORotCylinders.txt as shipped: a ball dropped between four turning cylinders (surface-only components, `script
    # DCORotCylinders`), selfFric 0.5; first contact in step 4 (19 Newton iterations), continued from the reference's own state after it
    ("sphere_rot_cylinders", "otherExamples/typical/sphere1K_DCORotCylinders.txt", "", 8),
    ("squash6_small", "inline:squash6_small", "", 44),
    ("squash6_contact", "inline:squash6_contact", "", 24),
    # BASELINE configs[1] on the reference's own mesh: 21_scalability/mat100x100_twist.txt (mat100x100t40.msh, 58 806 tets, `script twist`)
    ("mat100_twist", "paperExamples/21_scalability/mat100x100_twist.txt", "", 3),
    # BASELINE configs[3]: 4_rodsTwist.txt (4 x rod300x33.msh = 202 044 tets, `script twist`, selfCollisionOn)
    ("rods_twist", "paperExamples/4_rodsTwist.txt", "", 2),
    # BASELINE configs[2]: 12_sphereOnMat.txt (sphere1K 6 851 tets, E = 1e8, on mat40x40 9 126 tets, E = 1e6; `script stretchAndPause`, half-space,
    # selfCollisionOn): the mat is being stretched when the ball lands on it (step 29 on) -- contact in a generic, pre-strained state
    ("sphere_on_mat", "paperExamples/12_sphereOnMat.txt", "", 36),
]


def scenes(only=()):
    for name, rel, extra, steps in SCENES:
        if only and name not in only:
            continue
        text = (INLINE[rel] if rel in INLINE else open(os.path.join(REF_ROOT, "input", rel)).read()) + extra
        cfg = ss.SceneConfig.parse(text, REF_ROOT)
        if cfg.script_seq_folder:
            write_obj_sequence(cfg.script_seq_folder, steps + 1)
        with tempfile.TemporaryDirectory(prefix="ipcref_") as tmp:
            path = os.path.join(tmp, "scene.txt")
            lines = [ln for ln in text.splitlines() if not ln.strip().startswith("time ")]
            open(path, "w").write("\n".join(lines) + f"\ntime {steps * cfg.dt:.17g} {cfg.dt:.17g}\n")
            rcode, log = rc.run_reference(path, os.path.join(tmp, "ref"))
            assert rcode == 0, log[-2000:]
            its = rc.read_iter_counts(os.path.join(tmp, "ref"), steps)
            pos = np.array([rc.read_status_positions(os.path.join(tmp, "ref", f"status{s + 1}")) for s in range(steps)])
            extra = {}
            if name in RESTARTS:
                R, K = RESTARTS[name]
                status = open(os.path.join(tmp, "ref", f"status{R}")).read()
                spath = os.path.join(tmp, "restart_status")
                open(spath, "w").write(status)
                path2 = os.path.join(tmp, "scene_restart.txt")
                open(path2, "w").write("\n".join(lines) + f"\ntime {(R + K) * cfg.dt:.17g} {cfg.dt:.17g}\nrestart {spath}\n")
                rcode, log = rc.run_reference(path2, os.path.join(tmp, "ref2"))
                assert rcode == 0, log[-2000:]
                cum = []
                for sN in range(R + 1, R + K + 1):
                    with open(os.path.join(tmp, "ref2", f"info{sN}.txt")) as fh:
                        fh.readline()
                        cum.append(int(fh.readline().split()[1]))
                extra = dict(restart_step=R, restart_status=np.array(status), restart_iters=np.diff(np.array([0] + cum)),
                             restart_positions=np.array([rc.read_status_positions(os.path.join(tmp, "ref2", f"status{sN}")) for sN in range(R + 1, R + K + 1)]))
                print(f"  restart at step {R}: Newton iterations {extra['restart_iters'].tolist()}")
        out = dict(script=np.array(text), steps=steps, positions=pos, iters=its, **extra)
        keys = []
        for pth in [sh.path for sh in cfg.shapes] + [mc[0] for mc in cfg.mesh_cos]:  # every mesh file the script names travels with the fixture
            key = os.path.relpath(pth, REF_ROOT)
            if key in keys:
                continue
            i = len(keys)
            keys.append(key)
            if pth.lower().endswith(".obj"):
                out[f"mesh{i}_V"], out[f"mesh{i}_SF"] = ss.read_obj(pth)
                out[f"mesh{i}_T"] = np.zeros((0, 4), np.int32)
            elif pth.lower().endswith((".seg", ".pt")):
                if pth.lower().endswith(".seg"):
                    out[f"mesh{i}_V"], out[f"mesh{i}_E"] = ss.read_seg(pth)
                else:
                    out[f"mesh{i}_V"] = ss.read_pt(pth)
                out[f"mesh{i}_T"], out[f"mesh{i}_SF"] = np.zeros((0, 4), np.int32), np.zeros((0, 3), np.int32)
            else:
                out[f"mesh{i}_V"], out[f"mesh{i}_T"], out[f"mesh{i}_SF"] = gl.read_tet_mesh(pth)
        out["mesh_keys"] = np.array(keys)
        for sh in cfg.shapes:  # the files of a mesh sequence travel too: positions of file 0 .. steps - 1
            if sh.mesh_seq is not None:
                ext = os.path.splitext(sh.path.lower())[1]
                out["seq_" + os.path.relpath(sh.mesh_seq, REF_ROOT)] = np.array([ss.read_seq_file(sh.mesh_seq, i, ext) for i in range(steps)])
        if cfg.script_seq_folder:  # file n at index n (index 0 unused: the script counts from 1)
            files = [ss.read_seq_file(cfg.script_seq_folder, i, ".obj") for i in range(1, steps + 1)]
            out["seq_" + os.path.relpath(cfg.script_seq_folder, REF_ROOT)] = np.array([np.zeros_like(files[0])] + files)
        fn = os.path.join(GOLD, f"ref_scene_{name}.npz")
        np.savez_compressed(fn, **out)
        print(f"wrote {os.path.basename(fn)}: {steps} steps, Newton iterations per step {its.tolist()}, {os.path.getsize(fn) >> 10} KiB")


if __name__ == "__main__":
    what = sys.argv[1:] or ["functions", "scenes"]
    assert ref.available(), "build oracle/_ref first: make -C oracle -f Makefile.ref"
    if "functions" in what:
        functions()
    if "scenes" in what:
        scenes([w for w in what if w not in ("functions", "scenes")])
