"""The reference's scene-script grammar (src/Config.cpp:97-620) mapped onto the C ABI.

This is tooling, not product: a maintainer who links libipcgpu.so into IPC keeps IPC's own Config class; this module lets
the tests and tools/run_scene.py drive the library (or the CPU oracle, through the same `Backend` protocol) from the very
scene files the reference ships, so both start from identical inputs.  Only keywords whose feature exists here are accepted;
anything else raises `UnsupportedKeyword` instead of being ignored silently.

    cfg = SceneConfig.parse(open(path).read(), base_dir)       # grammar only
    scene = assemble(cfg, read_mesh)                          # shapes -> one mesh (main.cpp:880-1198)
    apply(scene, backend)                                     # C-ABI calls, in the order INTEGRATION.md gives
"""
from dataclasses import dataclass, field
import math
import os

import numpy as np

from . import scene as _scene


class UnsupportedKeyword(ValueError):
    pass


VIEWER_KEYWORDS = {"view", "zoom", "cameraTracking", "playBackSpeed", "appendStr", "disableCout", "section"}


def _rot(deg):
    """Rx Ry Rz of Euler angles in degrees (Config.cpp:222-226)."""
    x, y, z = (math.radians(a) for a in deg)
    cx, sx, cy, sy, cz, sz = math.cos(x), math.sin(x), math.cos(y), math.sin(y), math.cos(z), math.sin(z)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


@dataclass
class Shape:
    path: str
    translate: np.ndarray
    rotate_deg: np.ndarray
    scale: np.ndarray
    material: tuple = None  # (rho, E, nu)
    lin_vel: tuple = None  # scripted (kinematic component)
    ang_vel_deg: tuple = None
    init_vel: tuple = None  # (linear, angular deg/s)
    mesh_seq: str = None  # `meshSeq <folder>`: the component follows the positions of <folder>/<step>.{msh,obj,seg,pt} (Config.cpp:284-289)
    dbc: list = field(default_factory=list)  # (rel_min, rel_max, lin_vel, ang_vel_deg, t0, t1)
    nbc: list = field(default_factory=list)  # (rel_min, rel_max, acceleration, t0, t1)


@dataclass
class SceneConfig:
    energy: str = "NH"
    time_integration: str = "BE"
    beta: float = 0.25
    gamma: float = 0.5
    duration: float = 5.0
    dt: float = 0.025
    rho: float = 1000.0
    YM: float = 1e5
    PR: float = 0.4
    gravity: bool = True
    self_collision: bool = True  # Config.hpp:99
    self_fric: float = 0.0
    dHat_eps: float = 1e-3  # tuning[1]
    eps_v: float = 1e-3  # tuning[4]
    eps_v_target: float = -1.0  # tuning[5]; < 0: the same as eps_v (the `epsv` keyword sets both)
    handle_ratio: float = 0.01  # `handleRatio r` (Config.cpp:528-531): the width, in bounding-box extents, of the border slabs the handle scripts take (main.cpp:1180)
    dbc_time_range: tuple = (0.0, math.inf)  # `DBCTimeRange t0 t1` (Config.cpp:175-177): every Dirichlet group acts inside it only (AnimScripter.cpp:99, 1440)
    nbc_time_range: tuple = (0.0, math.inf)  # `NBCTimeRange t0 t1` (Config.cpp:178-180; AnimScripter.cpp:2372-2375)
    dtol_rel: float = 1e-9  # tuning[3] (Optimizer.cpp:102-106)
    use_abs_parameters: bool = False  # Config.cpp:553-555: the lengths of `tuning` and the tolerance are absolute
    kappa_min_multiplier: float = 1e11  # Config.cpp:556-558, Config.hpp:139
    fric_iter_amt: int = 1
    rot_axis: tuple = (0.0, 0.0, 0.0)  # rotateModel ax ay az deg (Config.cpp:523-526)
    rot_deg: float = 0.0
    kappa: float = 0.0  # tuning[0]; 0 = suggestKappa
    dHat_target: float = -1.0  # tuning[2]; < 0: the same as dHat (no homotopy)
    damping_stiff: float = 0.0  # Config.cpp:141-147; `dampingRatio r` becomes r * dt^3 * 3 / 4 once the file is read (:614-616)
    damping_ratio: float = 0.0
    tol: float = 1e-2
    script: str = "null"
    script_params: list = field(default_factory=list)
    script_seq_folder: str = None  # `script meshSeqFromFile <folder>`
    size: float = -1.0  # > 0: the assembled model is scaled so that its largest extent is `size` and moved to the origin (main.cpp:1140-1145)
    warm_start: int = 0  # initX option
    restart: str = None
    half_spaces: list = field(default_factory=list)  # (origin, normal, friction)
    mesh_cos: list = field(default_factory=list)  # (obj path, origin, scale, friction, rotate_deg) -- kinematic mesh obstacles (MeshCO)
    shapes: list = field(default_factory=list)

    @staticmethod
    def parse(text, base_dir="."):
        cfg = SceneConfig()
        lines = text.splitlines()
        i = 0

        def resolve(p):
            return p if os.path.isabs(p) else os.path.normpath(os.path.join(base_dir, p))

        while i < len(lines):
            tok = lines[i].split()
            i += 1
            if not tok or tok[0].startswith("#"):
                continue
            k, a = tok[0], tok[1:]
            if k == "energy":
                if a[0] not in ("NH", "FCR"):
                    raise UnsupportedKeyword(f"energy {a[0]}")
                cfg.energy = a[0]
            elif k == "timeIntegration":
                cfg.time_integration = a[0]
                if a[0] == "NM" and len(a) >= 3:
                    cfg.beta, cfg.gamma = float(a[1]), float(a[2])
                elif a[0] not in ("BE", "NM"):
                    raise UnsupportedKeyword(f"timeIntegration {a[0]}")
            elif k == "time":
                cfg.duration, cfg.dt = float(a[0]), float(a[1])
            elif k == "density":
                cfg.rho = float(a[0])
            elif k == "stiffness":
                cfg.YM, cfg.PR = float(a[0]), float(a[1])
            elif k == "turnOffGravity":
                cfg.gravity = False
            elif k == "script":
                if a[0] not in ("null", "twist", "fall", "fallNoShift", "dragright", "DCOFix", "DCOBallHitWall", "stretchAndPause", "meshSeqFromFile", "ACOSquash", "ACOSquash6", "DCOHammerWalnut", "DCOCut") + HANDLE_SCRIPTS + HOLD_SCRIPTS + PULL_SCRIPTS + INITVEL_SCRIPTS + RULE_SCRIPTS and a[0] not in DCO_SCRIPTS:
                    raise UnsupportedKeyword(f"script {a[0]}")
                cfg.script = a[0]
                if a[0] == "meshSeqFromFile":  # Config.cpp:161-164: the folder of <n>.obj files follows the name
                    cfg.script_seq_folder = resolve(a[1])
                    a = a[:1] + a[2:]
                if len(a) > 1 and int(a[1]) > 0:  # `script name n p1 .. pn` (Config.cpp:166-175): parameters of the script
                    cfg.script_params = [float(x) for x in a[2:2 + int(a[1])]]
            elif k == "warmStart":  # initX option (Optimizer.cpp:925-1110)
                if int(a[0]) not in (0, 1, 2, 3, 4, 5):
                    raise UnsupportedKeyword(f"warmStart {a[0]}")
                cfg.warm_start = int(a[0])
            elif k == "constraintSolver":
                if a[0] not in ("IP", "interiorPoint"):
                    raise UnsupportedKeyword(f"constraintSolver {a[0]}")
            elif k == "timeStepper":
                if a[0] != "Newton":
                    raise UnsupportedKeyword(f"timeStepper {a[0]}")
            elif k == "shape":  # single shape, identity transform (Config.cpp:188-203)
                if a[0] != "input":
                    raise UnsupportedKeyword(f"shape {a[0]}")
                cfg.shapes.append(Shape(resolve(a[1]), np.zeros(3), np.zeros(3), np.ones(3)))
            elif k == "shapes":
                if a[0] != "input":
                    raise UnsupportedKeyword(f"shapes {a[0]}")
                n = int(a[1])
                got = 0
                while got < n:
                    st = lines[i].split()
                    i += 1
                    if not st or st[0].startswith("#"):
                        continue
                    # a `\` token continues the shape on the next line; a `#` token drops the rest of the line up to a
                    # `\` (Config.cpp:290-305)
                    toks = []
                    while True:
                        cont = False
                        for j, t in enumerate(st):
                            if t == "\\":
                                cont = True
                                break
                            if t.startswith("#"):
                                cont = "\\" in st[j + 1:]
                                break
                            toks.append(t)
                        if not cont or i >= len(lines):
                            break
                        st = lines[i].split()
                        i += 1
                    cfg.shapes.append(_parse_shape(toks, resolve))
                    got += 1
            elif k == "shapeMatrix":  # Config.cpp:319-378: one shape replicated on an nx x ny x nz grid
                if a[0] != "input":
                    raise UnsupportedKeyword(f"shapeMatrix {a[0]}")
                cnt = [int(x) for x in a[1:4]]
                pos = [float(x) for x in a[4:7]] + [0.0] * (3 - len(a[4:7]))
                st = lines[i].split()
                i += 1
                step = np.array([float(x) for x in st[1:4]])
                rot, scl = np.array([float(x) for x in st[4:7]]), np.array([float(x) for x in st[7:10]])
                mat = tuple(float(x) for x in st[11:14]) if len(st) > 13 and st[10] == "material" else None
                for xi in range(cnt[0]):
                    for yi in range(cnt[1]):
                        for zi in range(cnt[2]):
                            cfg.shapes.append(Shape(resolve(st[0]), np.array(pos) + step * np.array([xi, yi, zi]), rot, scl, material=mat))
            elif k == "ground":  # friction, height (Config.cpp:425-430)
                cfg.half_spaces.append((np.array([0.0, float(a[1]), 0.0]), np.array([0.0, 1.0, 0.0]), float(a[0])))
            elif k == "halfSpace":  # origin, normal, stiffness (unused), friction (Config.cpp:431-447)
                v = [float(x) for x in a]
                n = np.array(v[3:6])
                cfg.half_spaces.append((np.array(v[0:3]), n / np.linalg.norm(n), v[7] if len(v) > 7 else 0.0))
            elif k == "meshCO":  # path, origin, scale, stiffness (unused), friction [rotate x y z] (Config.cpp:448-474)
                v = [float(x) for x in a[1:7]]
                rot = np.zeros(3)
                if len(a) > 10 and a[7] == "rotate":
                    rot = np.array([float(x) for x in a[8:11]])
                cfg.mesh_cos.append((resolve(a[0]), np.array(v[0:3]), v[3], v[5], rot))
            elif k == "selfCollisionOn":
                cfg.self_collision = True
            elif k == "selfCollisionOff":
                cfg.self_collision = False
            elif k == "selfFric":
                cfg.self_fric = float(a[0])
            elif k == "dHat":  # Config.cpp:542-545: entries 1 and 2 of `tuning`
                cfg.dHat_eps = cfg.dHat_target = float(a[0])
            elif k == "epsv":  # Config.cpp:546-549: entries 4 and 5 of `tuning`
                cfg.eps_v = float(a[0])
                cfg.eps_v_target = -1.0
            elif k == "fricIterAmt":
                cfg.fric_iter_amt = int(a[0])
            elif k == "tol":
                vals = []
                while len(vals) < int(a[0]):
                    vals += [float(x) for x in lines[i].split()]
                    i += 1
                if vals:
                    cfg.tol = vals[0]
            elif k == "tuning":  # Config.cpp:41-45, 533-541: [kappa (0 = automatic), dHat, dHat target, dTol, eps_v, eps_v target]
                vals = []
                while len(vals) < int(a[0]):
                    vals += [float(x) for x in lines[i].split()]
                    i += 1
                # `tuning.resize(amt)` (Config.cpp:533-541) DROPS the entries the line does not give: whatever an earlier `dHat` / `epsv`
                # keyword had set falls back to the Optimizer's defaults (Optimizer.cpp:276-304: 1e-3 each, kappa automatic)
                cfg.kappa = max(vals[0], 0.0) if len(vals) > 0 else 0.0  # the start value of every time step (Optimizer.cpp:1540-1547)
                cfg.dHat_eps = vals[1] if len(vals) > 1 else 1e-3
                cfg.dHat_target = vals[2] if len(vals) > 2 else 1e-3  # Optimizer.cpp:283-289: without a third entry the target is 1e-3 (relative)
                cfg.dtol_rel = vals[3] if len(vals) > 3 else 1e-9
                cfg.eps_v = vals[4] if len(vals) > 4 else 1e-3
                cfg.eps_v_target = vals[5] if len(vals) > 5 else 1e-3  # Optimizer.cpp:296-299: without a sixth entry the target is 1e-3
            elif k == "handleRatio":
                cfg.handle_ratio = float(a[0])
                if not 0 < cfg.handle_ratio < 0.5:
                    raise ValueError("handleRatio must lie in (0, 0.5) (Config.cpp:530)")
            elif k in ("linearSolver", "linSysSolver"):
                pass  # Config.cpp:120-124 picks CHOLMOD / AMGCL / Eigen inside the reference; the library brings its own factorisation
            elif k in ("CCDTolerance", "ccdTolerance"):
                pass  # Config.cpp:569-571: read by the TightInclusion back end only (Optimizer.cpp:1149, 1173); the default method takes none
            elif k == "DBCTimeRange":
                cfg.dbc_time_range = (float(a[0]), float(a[1]))
            elif k == "NBCTimeRange":
                cfg.nbc_time_range = (float(a[0]), float(a[1]))
            elif k == "useAbsParameters":  # Config.cpp:553-555
                cfg.use_abs_parameters = True
            elif k in ("kappaMinMultiplier", "minBarrierStiffnessScale"):  # Config.cpp:556-558
                cfg.kappa_min_multiplier = float(a[0])
            elif k == "section":  # Config.cpp:572-605: settings for one constraint solver; other solvers' sections are skipped
                names = ["interiorPoint" if x == "IP" else x for x in a]
                if "end" not in names and "interiorPoint" not in names:
                    while i < len(lines):
                        t2 = lines[i].split()
                        i += 1
                        if len(t2) >= 2 and t2[0] == "section" and t2[1] == "end":
                            break
            elif k in ("CCDMethod", "ccdMethod"):  # Config.cpp:564-568: only the default method is rebuilt here
                if a[0] != "FloatingPointRootFinder":
                    raise UnsupportedKeyword(f"CCDMethod {a[0]}")
            elif k == "restart":
                cfg.restart = resolve(a[0])
            elif k == "size":  # Config.cpp: `size s`; applied to the whole model after the shapes are assembled (main.cpp:1140-1145)
                cfg.size = float(a[0])
            elif k == "rotateModel":  # Config.cpp:523-526
                cfg.rot_axis, cfg.rot_deg = tuple(float(x) for x in a[:3]), float(a[3])
            elif k == "dampingStiff":  # Config.cpp:141-147
                cfg.damping_stiff = max(float(a[0]), 0.0)
            elif k == "dampingRatio":  # Config.cpp:148-157
                cfg.damping_ratio = min(max(float(a[0]), 0.0), 1.0)
            elif k in VIEWER_KEYWORDS:
                pass  # viewer / logging only
            elif k in ("resolution", "inexactSolve"):
                pass  # tokens Config::loadFromFile itself has no branch for (Config.cpp:97-617): the reference reads past them
            else:
                raise UnsupportedKeyword(k)
        if cfg.damping_ratio > 0:  # Config.cpp:614-616
            cfg.damping_stiff = cfg.damping_ratio * cfg.dt ** 3 * 3 / 4
        return cfg


def _parse_shape(st, resolve):
    sh = Shape(resolve(st[0]), np.array([float(x) for x in st[1:4]]), np.array([float(x) for x in st[4:7]]), np.array([float(x) for x in st[7:10]]))
    j = 10
    while j < len(st):
        e = st[j]
        j += 1
        if e == "material":
            sh.material = tuple(float(x) for x in st[j:j + 3])
            j += 3
        elif e == "linearVelocity":
            sh.lin_vel = tuple(float(x) for x in st[j:j + 3])
            j += 3
        elif e == "angularVelocity":
            sh.ang_vel_deg = tuple(float(x) for x in st[j:j + 3])
            j += 3
        elif e == "meshSeq":
            sh.mesh_seq = resolve(st[j])
            j += 1
        elif e == "initVel":
            v = [float(x) for x in st[j:j + 6]]
            sh.init_vel = (tuple(v[:3]), tuple(v[3:]))
            j += 6
        elif e == "DBC":
            v = [float(x) for x in st[j:j + 12]]
            j += 12
            t0, t1 = 0.0, float("inf")
            try:  # optional start / stop time (Config.cpp:251-254)
                t0 = float(st[j])
                j += 1
                t1 = float(st[j])
                j += 1
            except (IndexError, ValueError):
                pass
            sh.dbc.append((v[0:3], v[3:6], v[6:9], v[9:12], t0, t1))
        elif e == "NBC":
            v = [float(x) for x in st[j:j + 9]]
            j += 9
            t0, t1 = 0.0, float("inf")
            try:  # optional start / stop time (Config.cpp:269-272)
                t0 = float(st[j])
                j += 1
                t1 = float(st[j])
                j += 1
            except (IndexError, ValueError):
                pass
            sh.nbc.append((v[0:3], v[3:6], v[6:9], t0, t1))
        else:
            raise UnsupportedKeyword(f"shape keyword {e}")
    return sh


def read_obj(path):
    """Triangle mesh of a Wavefront .obj (igl::readOBJ as MeshCO uses it, MeshCO.cpp:49): `v x y z` and `f a b c` lines, 1-based
    indices, `a/b/c` index groups reduced to the vertex index, negative (relative) indices, polygons fan-triangulated."""
    V, F = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                V.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = []
                for g in t[1:]:
                    k = int(g.split("/")[0])
                    idx.append(k - 1 if k > 0 else len(V) + k)
                for q in range(1, len(idx) - 1):
                    F.append([idx[0], idx[q], idx[q + 1]])
    return np.array(V, dtype=np.float64).reshape(-1, 3), np.array(F, dtype=np.int32).reshape(-1, 3)


def read_seg(path):
    """Segment mesh of a `.seg` shape (main.cpp:957-990): IglUtils::readSEG (`v x y z` / `s a b`, 1-based; IglUtils.cpp:146-175) or, when that
    file does not exist, the edges of the triangles of the `.obj` beside it -- each undirected edge once, with the direction it is first
    seen in, in the order of a std::set of pairs."""
    if os.path.exists(path):
        V, E = [], []
        with open(path) as f:
            for line in f:
                t = line.split()
                if not t:
                    continue
                if t[0] == "v":
                    V.append([float(x) for x in t[1:4]])
                elif t[0] == "s":
                    E.append([int(t[1]) - 1, int(t[2]) - 1])
        return np.array(V, dtype=np.float64).reshape(-1, 3), np.array(E, dtype=np.int32).reshape(-1, 2)
    V, F = read_obj(os.path.splitext(path)[0] + ".obj")
    es = set()
    for a, b, c in F.tolist():
        for i, j in ((a, b), (b, c), (c, a)):
            if (j, i) not in es:
                es.add((i, j))
    return V, np.array(sorted(es), dtype=np.int32).reshape(-1, 2)


def read_pt(path):
    """Point cloud of a `.pt` shape (main.cpp:991-1005): the vertices of the file read as an .obj, or of the `.obj` beside it"""
    if os.path.exists(path):
        return read_obj(path)[0]
    return read_obj(os.path.splitext(path)[0] + ".obj")[0]


def read_seq_file(folder, index, ext):
    """positions of file `index` of a mesh sequence, by the kind of the component that follows it (AnimScripter.cpp:1468-1518)"""
    base = os.path.join(folder, str(index))
    if ext == ".obj":
        return read_obj(base + ".obj")[0]
    if ext == ".seg":
        return read_seg(base + ".seg")[0]
    return read_pt(base + ".pt")


# The hard-coded scripts of AnimScripter that move whole components (set-up AnimScripter.cpp:1060-1300, per step :1961-2135): how many
# leading components they move and with what.  Linear velocities in units / s, angular velocities in rad / s about x, y, z.
_S6 = [(1.0, 0, 0), (-1.0, 0, 0), (0, 1.0, 0), (0, -1.0, 0), (0, 0, 1.0), (0, 0, -1.0)]
_HP = math.pi / 2.0
DCO_SCRIPTS = {
    "DCOSquash": {"n": 2, "lin": _S6[:2]},  # two plates closing along x, turning round when 0.1 apart (:1172-1191, 2034-2051)
    "DCOSquash6": {"n": 6, "lin": _S6},  # six plates closing along x, y, z (the trash compactor; :1193-1221, 2053-2074)
    "DCORotCylinders": {"n": 4, "ang": [(_HP, 0, 0), (-_HP, 0, 0), (0, 0, -_HP), (0, 0, _HP)]},  # :1060-1086, 1961-1975
    "DCOVerschoorRoller": {"n": 6, "ang": [(0, 0, -4.0), (0, 0, -2.0), (0, 0, 2.0), (0, 0, 4.0), (2.0, 0, 0), (-2.0, 0, 0)]},  # :1088-1118, 1977-1991
    "DCOSqueezeOut": {"n": 0},  # every surface-only component is held (the rule of :2102-2124 that would move the first one never fires, see assemble)
}


# scripts that pick their Dirichlet / Neumann nodes from the bounding box of the assembled mesh (set-up in AnimScripter::initAnimScript)
HANDLE_SCRIPTS = ("fixLowerHalf", "pushRightMost1", "utopiaComparison", "DCOSegBedSquash", "hangLeft")
# nodes picked by a box rule of the start positions and HELD (ZERO: taken out of the system; NONZERO: a Dirichlet set that does not move), or moved
# at a constant velocity (AnimScripter.cpp:153-189, 224-247, 284-297, 318-373, 459-473, 502-516, 790-807, 860-893; per step :1536-1551, 1597-1603,
# 1817-1826); `drop`, `leftHitRight`, `XYRotate` only give start velocities (:853-858, 1336-1374)
HOLD_SCRIPTS = ("hang", "hang2", "hangTopLeft", "stamp", "stampTopLeft", "stampBoth", "stand", "topbottomfix", "corner", "fixRightMost1")
PULL_SCRIPTS = ("stretch", "squash", "dragdown", "curtain")
INITVEL_SCRIPTS = ("drop", "leftHitRight", "XYRotate")
# handle sets that move, turn round or stop by a rule on one "turning" node (AnimScripter.cpp:375-457, 518-553, 574-631; per step :1554-1595, 1648-1729)
RULE_SCRIPTS = ("push", "tear", "upndown", "undstamp", "stretchnsquash", "bend", "twistnstretch", "twistnsns", "twistnsns_old",
                # handle sets that are LET GO (the others stop) when the turning node has travelled far enough (:633-760, 827-851; per step :1731-1812)
                "rubberBandPull", "fourLegPull", "headTailPull", "toggleTop",
                # start positions changed and / or nodes held (:129-151, 206-222, 265-282, 299-316), Neumann pull on the top (:912-954)
                "scaleF", "swing", "stampInv", "standInv", "NMFixBottomDragLeft", "NMFixBottomDragForward")


@dataclass
class AssembledScene:
    cfg: SceneConfig
    V: np.ndarray
    T: np.ndarray
    SF: np.ndarray
    node_ranges: list
    tet_ranges: list
    dirichlet: list  # (ids, lin_vel, ang_vel_deg, t0, t1)
    velocity: np.ndarray
    neumann: list = field(default_factory=list)  # (ids, acceleration, t0, t1)
    obstacle_nodes: np.ndarray = None  # nodes of the kinematic mesh obstacles (surface-only components, no tetrahedra)
    release: dict = None  # state-dependent end of a scripted handle (`script dragright`)
    codim_nodes: np.ndarray = None  # surface-only nodes of the mesh itself (triangle meshes under `shapes`, componentCoDim 2) ...
    codim_mass: np.ndarray = None  # ... and their lumped masses (density x a third of the adjacent triangle areas, Mesh.cpp:310-345)
    codim_fixed: np.ndarray = None  # `script DCOFix`: those of them held as NONZERO Dirichlet nodes (AnimScripter.cpp:1222-1236)
    V0: np.ndarray = None  # start positions when they differ from the rest shape V (`rotateModel`, main.cpp:1115-1139)
    codim_edges: np.ndarray = None  # segments of the `.seg` shapes (Mesh::CE), global node pairs
    mesh_seqs: list = None  # `meshSeq` components: {ids, folder, ext, group}
    mesh_i: int = 0  # AnimScripter::meshI: index of the next file of the sequences
    read_seq: object = None  # (folder, index, ext) -> positions; None = the files themselves (tests hand in a fixture's)
    motions: list = None  # rule-driven scripts: per Dirichlet group (lin, ang in degrees, fixed rotation centre or None), nodes NONZERO throughout

    def before_step(self, be, t):
        """What AnimScripter::stepAnimScript decides from the state before a time step (call with the step's start time)."""
        if self.mesh_seqs:
            # AnimScripter.cpp:1465-1532: <folder>/<meshI>.{obj,seg,pt} read as main.cpp reads the shape itself (file positions as they are,
            # NOT moved by the shape's translation / rotation / scale); meshI counts the time steps
            for q in self.mesh_seqs:
                be.set_dirichlet_targets(q["group"], (self.read_seq or read_seq_file)(q["folder"], self.mesh_i, q["ext"]))
            self.mesh_i += 1
        r = self.release
        if r is None or r["done"]:
            return False
        x = np.asarray(be.state()["V"]).reshape(-1, 3)
        if r.get("kind") == "dco":
            nr = self.node_ranges
            if r["script"] in ("DCOSquash", "DCOSquash6"):
                # AnimScripter.cpp:2034-2074: while the first two plates are closer than 0.1 in x, EVERY velocity changes sign -- once per
                # time step, as written
                if x[nr[1]:nr[2], 0].min() - x[nr[0]:nr[1], 0].max() < 0.1:
                    r["lin"] = [tuple(-c for c in v) for v in r["lin"]]
                    for g, v in enumerate(r["lin"]):
                        be.set_dirichlet_motion(g, lin_vel=v, force_nonzero=True)
                    return True
            return False
        if r.get("kind") == "segbed":  # AnimScripter.cpp:2080-2100: the upper parts move only while they are more than 0.1 above the lower ones
            top_min = x[r["up"], 1].min()
            bottom_max = x[r["down"], 1].max() if len(r["down"]) else -np.inf
            moving = bool(top_min - bottom_max > 0.1)
            if moving != r["moving"]:
                r["moving"] = moving
                be.set_dirichlet_motion(r["group"], lin_vel=(0.0, -1.0, 0.0) if moving else (0.0, 0.0, 0.0), force_nonzero=True)
                return True
            return False
        if r.get("kind") == "turn":
            # one node decides (velocityTurningPoints): outside its window the listed velocity components change sign -- every step it is found
            # outside, as written (AnimScripter.cpp:1568-1595, 1648-1660, 1702-1729) -- or, `push`, the handles stop for good (:1554-1566)
            c = x[r["turn"], r["axis"]]
            if not (c <= r["lo"] or c >= r["hi"]):
                return False
            if r["stop"]:
                r["lin"] = [(0.0, 0.0, 0.0) if g in r["groups"] else v for g, v in enumerate(r["lin"])]
                r["done"] = True
            else:
                r["lin"] = [tuple(-c_ if k == r["comp"] else c_ for k, c_ in enumerate(v)) if g in r["groups"] else v for g, v in enumerate(r["lin"])]
            for g in r["groups"]:
                be.set_dirichlet_motion(g, lin_vel=r["lin"][g], ang_vel_deg=r["ang"][g], center=r["ctr"][g], force_nonzero=True)
            return True
        if r.get("kind") == "aco":
            # AnimScripter.cpp:1832-1871: the analytic planes close in at 1 along their axis, every velocity of a pair changing sign while the two
            # are closer than 0.1 (ACOSquash) / 0.2 (ACOSquash6); each plane then moves by the fraction of v dt that HalfSpace::move admits
            o, v = r["origins"], r["vel"]
            for a, gap in r["pairs"]:
                if o[a + 1][a // 2] - o[a][a // 2] < gap:
                    v[a][a // 2] *= -1.0
                    v[a + 1][a // 2] *= -1.0
            for i in range(len(v)):
                d = v[i] * self.cfg.dt
                o[i] = o[i] + (1.0 - be.half_space_move(i, d, 0.5)) * d
            return True
        if r.get("kind") == "while_above":  # AnimScripter.cpp:1996-2012, 2019-2030: the component moves in the steps that start with its lowest node above the mark
            moving = bool(x[r["ids"], 1].min() > r["y"])
            if moving != r["moving"]:
                r["moving"] = moving
                lin, ang, ctr = r["motion"] if moving else ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), None)
                be.set_dirichlet_motion(0, lin_vel=lin, ang_vel_deg=ang, center=ctr, force_nonzero=True)
                return True
            return False
        if r.get("kind") == "let_go":
            c = x[r["turn"], r["axis"]]
            if not (c <= r["limit"] if r["below"] else c >= r["limit"]):
                return False
            for g in r["free"]:
                be.end_dirichlet(g, t)
            for g in r["stop"]:
                be.set_dirichlet_motion(g, lin_vel=(0.0, 0.0, 0.0), force_nonzero=True)
            r["done"] = True
            return True
        if r.get("kind") == "pause":
            # `script stretchAndPause` (AnimScripter.cpp:1605-1616): the handles move while the turning vertex has not passed x = -0.28;
            # from then on every Dirichlet node is held (vertexDBCType ZERO)
            if x[r["turn"], 0] >= r["x_limit"]:
                return False
            for g in r["groups"]:
                be.end_dirichlet(g, t)
            for ids in r["ids"]:
                be.add_dirichlet(ids, lin_vel=(0.0, 0.0, 0.0), ang_vel_deg=(0.0, 0.0, 0.0), t0=t, t1=float("inf"))
            r["done"] = True
            return True
        if x[: r["nSim"], 0].min() > r["x_limit"]:
            be.end_dirichlet(r["group"], t)
            r["done"] = True
            return True
        return False


def assemble(cfg, read_mesh):
    """main.cpp:880-1198: select Dirichlet / Neumann nodes per shape on the mesh as read, transform the shape (R (p * scale) + translate), concatenate."""
    Vs, Ts, SFs, nr, tr, dirichlet, neumann = [], [], [], [0], [0], [], []
    codim = []  # (node ids, triangles, moved by its own keywords, segments) of the components of the mesh without tetrahedra
    CEs = []
    seqs = []  # mesh sequences: {ids, folder, ext}
    for sh in cfg.shapes:
        ext = os.path.splitext(sh.path.lower())[1]
        is_codim = ext in (".obj", ".seg", ".pt")  # main.cpp:948-1005: kinematic surface / segments / points (componentCoDim 2 / 1 / 0)
        E = np.zeros((0, 2), dtype=np.int32)
        if ext == ".obj":
            V, SF = read_obj(sh.path)
        elif ext == ".seg":
            V, E = read_seg(sh.path)
            SF = np.zeros((0, 3), dtype=np.int32)
        elif ext == ".pt":
            V = read_pt(sh.path)
            SF = np.zeros((0, 3), dtype=np.int32)
        else:
            V, T, SF = read_mesh(sh.path)
        if is_codim:
            T = np.zeros((0, 4), dtype=np.int32)
        off = nr[-1]
        # Dirichlet / Neumann nodes are picked on the mesh AS READ (IglUtils::Init_Dirichlet on newV, main.cpp:1045-1068); the
        # shape is scaled / rotated / translated only afterwards (main.cpp:1073-1077), so the relative box follows the shape
        for rel_min, rel_max, lin, ang, t0, t1 in sh.dbc:
            ids = _scene.select_dirichlet(V, SF, rel_min, rel_max)
            if len(ids):
                dirichlet.append((ids + off, lin, ang, t0, t1))
        for rel_min, rel_max, acc, t0, t1 in sh.nbc:  # same vertex selection (main.cpp:1057-1068)
            ids = _scene.select_dirichlet(V, SF, rel_min, rel_max)
            if len(ids):
                neumann.append((ids + off, acc, t0, t1))
        V = (V * sh.scale) @ _rot(sh.rotate_deg).T + sh.translate
        if sh.lin_vel is not None or sh.ang_vel_deg is not None:  # scripted component: every node moves (AnimScripter.cpp:1413-1435)
            ids = np.arange(V.shape[0], dtype=np.int32) + off
            dirichlet.append((ids, sh.lin_vel or (0, 0, 0), sh.ang_vel_deg or (0, 0, 0), 0.0, float("inf")))
        if sh.mesh_seq is not None:
            # AnimScripter.cpp:1465-1528: every node of the component is moved to the positions of the next file of the sequence before each
            # step.  The nodes carry the type their velocities give them: a codimensional component without velocities is held (ZERO, :62-70)
            if cfg.script != "null":
                raise UnsupportedKeyword("meshSeq under a script other than null")
            if not is_codim:
                raise UnsupportedKeyword("meshSeq on a tetrahedral component (its nodes are free: the sequence would only nudge them)")
            ids = np.arange(V.shape[0], dtype=np.int32) + off
            dirichlet.append((ids, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf")))
            seqs.append({"ids": ids, "folder": sh.mesh_seq, "ext": ext})
        if is_codim:
            codim.append((np.arange(off, off + V.shape[0], dtype=np.int32), SF + off,
                          sh.lin_vel is not None or sh.ang_vel_deg is not None or sh.mesh_seq is not None, E + off))
        CEs.append(E + off)
        Vs.append(V)
        Ts.append(T + off)
        SFs.append(SF + off)
        nr.append(off + V.shape[0])
        tr.append(tr[-1] + T.shape[0])
    # kinematic mesh obstacles (MeshCO::MeshCO, MeshCO.cpp:37-58): centred on the vertex mean, rotated, scaled so that the largest
    # bounding-box extent equals `scale`, moved to `origin`.  They ride along as surface-only components: nodes of no tetrahedron.
    obstacle = []
    for path, origin, scale, _mu, rot in cfg.mesh_cos:
        Vo, Fo = read_obj(path)
        Vo = Vo - Vo.mean(0)
        Vo = Vo @ _rot(rot).T
        Vo = Vo * (scale / (Vo.max(0) - Vo.min(0)).max()) + origin
        off = nr[-1]
        obstacle.append(np.arange(off, off + Vo.shape[0], dtype=np.int32))
        Vs.append(Vo)
        SFs.append(Fo + off)
        nr.append(off + Vo.shape[0])
        tr.append(tr[-1])
    V, T, SF = np.vstack(Vs), np.vstack(Ts).astype(np.int32), np.vstack(SFs).astype(np.int32)
    nSim = nr[len(cfg.shapes)]  # the simulated mesh: everything before the obstacle components
    V0 = None
    if cfg.rot_deg != 0.0 and nSim:
        # main.cpp:1115-1139: the START positions are the model turned about an axis (Eigen::AngleAxis::toRotationMatrix, the axis as
        # given) through `center` = HALF THE EXTENT of the bounding box (not its middle); the rest shape stays as assembled
        c, sn = math.cos(math.radians(cfg.rot_deg)), math.sin(math.radians(cfg.rot_deg))
        ax = np.array(cfg.rot_axis, float)
        R = (1 - c) * np.outer(ax, ax) + c * np.eye(3) + sn * np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        V0 = V.copy()
        center = (V[:nSim].max(0) - V[:nSim].min(0)) / 2.0
        V0[:nSim] = (V[:nSim] - center) @ R.T + center
    if cfg.size > 0 and nSim:
        # main.cpp:1140-1145: the assembled model (all shapes, not the collision objects) is scaled so that the largest
        # bounding-box extent of its start positions equals `size`, then moved so that the box's lower corner sits at the origin
        U = V if V0 is None else V0
        ext = (U[:nSim].max(0) - U[:nSim].min(0)).max()
        V[:nSim] *= cfg.size / ext
        if V0 is not None:
            V0[:nSim] *= cfg.size / ext
        lo = (V if V0 is None else V0)[:nSim].min(0).copy()
        V[:nSim] -= lo
        if V0 is not None:
            V0[:nSim] -= lo
    if cfg.script in ("fall", "fallNoShift"):  # AnimScripter.cpp:779-788: lifted by half the bounding-box diagonal, no Dirichlet nodes
        if cfg.script == "fall":  # Mesh<3> only: the obstacles are not part of it in the reference (dragdown lifts the same way, below)
            U = V if V0 is None else V0
            lift = 0.5 * np.linalg.norm(U[:nSim].max(0) - U[:nSim].min(0))
            V[:nSim, 1] += lift
            if V0 is not None:
                V0[:nSim, 1] += lift
        dirichlet = []
    if cfg.script in HOLD_SCRIPTS + PULL_SCRIPTS:
        if cfg.script == "dragdown":  # AnimScripter.cpp:790-792: lifted like `fall`
            U = V if V0 is None else V0
            lift = 0.5 * np.linalg.norm(U[:nSim].max(0) - U[:nSim].min(0))
            V[:nSim, 1] += lift
            if V0 is not None:
                V0[:nSim, 1] += lift
        U = (V if V0 is None else V0)[:nSim]
        lo, hi = U.min(0), U.max(0)  # mesh.V.colwise().minCoeff() / maxCoeff()
        rng = hi - lo
        left, right = _scene.border_verts(U, cfg.handle_ratio)  # mesh.borderVerts_primitive (main.cpp:1180)
        ids32 = lambda m: np.nonzero(m)[0].astype(np.int32)  # noqa: E731
        inf = float("inf")
        zero3 = (0.0, 0.0, 0.0)
        hold = None  # (node ids, ZERO?)
        if cfg.script == "hang":  # the LAST node of either border set, ZERO
            hold = (np.array([b[-1] for b in (left, right) if len(b)], dtype=np.int32), True)
        elif cfg.script == "hang2":  # the top 1 %, ZERO
            hold = (ids32(U[:, 1] > hi[1] - rng[1] * 0.01), True)
        elif cfg.script == "hangTopLeft":  # the nodes of the left border set in the top 1 % and within 1 % of either z end, ZERO
            L = np.asarray(left)
            m = (U[L, 1] > hi[1] - rng[1] * 0.01) & ((U[L, 2] > hi[2] - rng[2] * 0.01) | (U[L, 2] < lo[2] + rng[2] * 0.01))
            hold = (L[m].astype(np.int32), True)
        elif cfg.script == "stamp":
            hold = (np.asarray(left, dtype=np.int32), False)
        elif cfg.script == "stampTopLeft":
            L = np.asarray(left)
            hold = (L[U[L, 1] > hi[1] - rng[1] * 0.01].astype(np.int32), False)
        elif cfg.script == "stampBoth":
            hold = (np.concatenate([left, right]).astype(np.int32), False)
        elif cfg.script == "stand":  # the bottom 1 %
            hold = (ids32(U[:, 1] < lo[1] + rng[1] * 0.01), False)
        elif cfg.script == "topbottomfix":  # the bottom and the top 2 %
            hold = (ids32((U[:, 1] < lo[1] + rng[1] * 0.02) | (U[:, 1] > hi[1] - rng[1] * 0.02)), False)
        elif cfg.script == "corner":  # within 1 % of the x, y or z minimum
            hold = (ids32((U[:, 0] < lo[0] + rng[0] * 0.01) | (U[:, 1] < lo[1] + rng[1] * 0.01) | (U[:, 2] < lo[2] + rng[2] * 0.01)), False)
        elif cfg.script == "fixRightMost1":  # the FIRST node within 1e-3 of the right end
            hold = (ids32(U[:, 0] > hi[0] - 1.0e-3 * rng[0])[:1], False)
        if hold is not None:
            hp_dirichlet = [(hold[0], zero3, zero3, 0.0, inf)] if len(hold[0]) else []
            hp_motions = None if hold[1] or not hp_dirichlet else [(zero3, zero3, None)]
        else:
            groups = []
            if cfg.script in ("stretch", "squash"):  # the border sets pulled apart at 0.1 / pushed together at 0.03 along x
                v = -0.1 if cfg.script == "stretch" else 0.03
                groups = [(np.asarray(left, dtype=np.int32), (v, 0.0, 0.0)), (np.asarray(right, dtype=np.int32), (-v, 0.0, 0.0))]
            elif cfg.script == "dragdown":  # the middle of the bottom tenth pulled down at 1.5
                m = (U[:, 1] < lo[1] + rng[1] * 0.1) & (U[:, 0] < lo[0] + rng[0] * 0.52) & (U[:, 0] > lo[0] + rng[0] * 0.42)
                groups = [(ids32(m), (0.0, -1.5, 0.0))]
            else:  # curtain: eight pins along the top edge, pin i drawn in +x at 0.04 (7 - i) / 7; a node takes the first pin that fits
                taken = np.zeros(len(U), dtype=bool)
                for pin in range(8):
                    x0 = lo[0] + rng[0] / 7.0 * pin
                    m = (U[:, 0] > x0 - rng[0] * 0.0025) & (U[:, 0] < x0 + rng[0] * 0.0025) & (U[:, 1] > hi[1] - rng[1] * 0.005) & ~taken
                    taken |= m
                    groups.append((ids32(m), (0.04 * (7.0 - pin) / 7.0, 0.0, 0.0)))
            groups = [g for g in groups if len(g[0])]
            hp_dirichlet = [(ids, lin, zero3, 0.0, inf) for ids, lin in groups]
            hp_motions = [(lin, zero3, None) for _ids, lin in groups]
    rule_release = None
    if cfg.script in RULE_SCRIPTS:
        U = (V if V0 is None else V0)[:nSim]
        lo, hi = U.min(0), U.max(0)
        rng = hi - lo
        left, right = (np.asarray(b, dtype=np.int32) for b in _scene.border_verts(U, cfg.handle_ratio))
        ids32 = lambda m: np.nonzero(m)[0].astype(np.int32)  # noqa: E731
        zero3 = (0.0, 0.0, 0.0)
        ctr_rest = tuple(0.5 * (V[:nSim].min(0) + V[:nSim].max(0)))  # mesh.bbox.colwise().mean(): the rest shape's box
        groups = []  # (ids, lin, ang in degrees, centre)
        turn = None  # (node, axis, lo, hi, component that changes sign, groups it applies to, stop instead)
        let_go = None
        rule_zero = False  # the held set is typed ZERO
        if cfg.script in ("push", "tear"):  # bottom 1 % held; top 1 % pushed down at 1 until it is 0.5 lower / dragged in -x at 5, turning at 4 further left
            bottom = U[:, 1] < lo[1] + rng[1] * 0.01
            top = ids32(~bottom & (U[:, 1] > hi[1] - rng[1] * 0.01))
            groups = [(ids32(bottom), zero3, zero3, None), (top, (0.0, -1.0, 0.0) if cfg.script == "push" else (-5.0, 0.0, 0.0), zero3, None)]
            if len(top):
                turn = (int(top[0]), 1, U[top[0], 1] - 0.5, np.inf, 1, [1], True) if cfg.script == "push" else (int(top[0]), 0, U[top[0], 0] - 4.0, np.inf, 0, [1], False)
        elif cfg.script in ("upndown", "undstamp"):  # the border sets (undstamp: the left one) go up / down at 1.8, turning 0.6 above and below the start
            sets = [left, right] if cfg.script == "upndown" else [left]
            groups = [(b, (0.0, (-1.0) ** i * 1.8, 0.0), zero3, None) for i, b in enumerate(sets)]
            turn = (int(left[0]), 1, U[left[0], 1] - 0.6, U[left[0], 1] + 0.6, 1, list(range(len(sets))), False)
        elif cfg.script == "stretchnsquash":  # pulled apart at 0.9, turning when the first left node is 0.8 further out / 0.4 further in
            groups = [(b, ((-1.0) ** i * -0.9, 0.0, 0.0), zero3, None) for i, b in enumerate((left, right))]
            turn = (int(left[0]), 0, U[left[0], 0] - 0.8, U[left[0], 0] + 0.4, 0, [0, 1], False)
        elif cfg.script == "bend":  # every border node but the last of its set turns about z through that last one at 0.05 pi (9 degrees) per unit time
            for i, b in enumerate((left, right)):
                groups.append((b[:-1], zero3, (0.0, 0.0, (-1.0) ** i * -9.0), tuple(U[b[-1]])))
                groups.append((b[-1:], zero3, zero3, None))
        elif cfg.script in ("rubberBandPull", "fourLegPull", "headTailPull", "toggleTop"):
            y0, y1, x0, x1, z0, z1 = lo[1], hi[1], lo[0], hi[0], lo[2], hi[2]
            if cfg.script == "rubberBandPull":  # bottom / top 2 % drawn apart at 0.2, the waist pulled in -x at 2.5 and let go 5 further left
                a = U[:, 1] < y0 + rng[1] * 0.02
                b = ~a & (U[:, 1] > y1 - rng[1] * 0.02)
                w = ~a & ~b & (U[:, 1] < y1 - rng[1] * 0.48) & (U[:, 1] > y0 + rng[1] * 0.48)
                sets = [(a, (0.0, -0.2, 0.0), False), (b, (0.0, 0.2, 0.0), False), (w, (-2.5, 0.0, 0.0), True)]
                lim = (2, 0, -5.0, True)  # (set whose first node decides, axis, offset of the limit, "<=")
            elif cfg.script == "fourLegPull":
                a = (U[:, 1] > y1 - rng[1] * 0.129) & (U[:, 0] < x0 + rng[0] * 0.16)
                b = ~a & (U[:, 1] > y1 - rng[1] * 0.16) & (U[:, 0] > x1 - rng[0] * 0.16)
                c = ~a & ~b & (U[:, 1] < y0 + rng[1] * 0.02) & (U[:, 0] > x1 - rng[0] * 0.25)
                d = ~a & ~b & ~c & (U[:, 1] < y0 + rng[1] * 0.02) & (U[:, 0] < x0 + rng[0] * 0.25)
                sets = [(a, zero3, False), (b, (2.5, 0.0, 0.0), True), (c, (2.5, -3.5, 0.0), True), (d, (0.0, -3.5, 0.0), True)]
                lim = (3, 1, -5.0, True)
            elif cfg.script == "headTailPull":
                a = U[:, 2] < z0 + rng[2] * 0.02
                b = ~a & (U[:, 2] > z1 - rng[2] * 0.02)
                c = ~a & ~b & (U[:, 2] > z0 + rng[2] * 0.46) & (U[:, 2] < z0 + rng[2] * 0.54)
                sets = [(a, (3.5, 0.0, 0.0), True), (b, (3.5, 0.0, 0.0), True), (c, zero3, False)]
                lim = (0, 0, 4.5, False)
            else:  # toggleTop: the top 2 % drawn in -x at 0.5 and let go 0.1 further left
                sets = [(U[:, 1] > y1 - rng[1] * 0.02, (-0.5, 0.0, 0.0), True)]
                lim = (0, 0, -0.1, True)
            groups = [(ids32(m), lin, zero3, None) for m, lin, _f in sets]
            first = groups[lim[0]][0]
            if len(first):
                let_go = {"kind": "let_go", "turn": int(first[0]), "axis": lim[1], "limit": float(U[first[0], lim[1]] + lim[2]), "below": lim[3],
                          "free": [k for k, (_m, _l, f) in enumerate(sets) if f], "stop": [k for k, (_m, _l, f) in enumerate(sets) if not f], "done": False}
        elif cfg.script in ("scaleF", "swing", "stampInv", "standInv", "NMFixBottomDragLeft", "NMFixBottomDragForward"):
            W = V if V0 is None else V0  # the start positions themselves (the rest shape stays)
            if cfg.script in ("scaleF", "stampInv", "standInv", "swing") and V0 is None:
                V0 = V.copy()
                W = V0
            if cfg.script == "scaleF":  # every start position times 1.5 about the origin
                W[:nSim] *= 1.5
            elif cfg.script == "swing":  # lifted by 1.3 heights, the left 5 % held (ZERO)
                W[:nSim, 1] += 1.3 * rng[1]
                groups = [(ids32(U[:, 0] < lo[0] + rng[0] * 0.05), zero3, zero3, None)]
                rule_zero = True
            elif cfg.script in ("stampInv", "standInv"):  # the left / bottom 1 % held; the start turned inside out and squeezed to a tenth about it
                ax = 0 if cfg.script == "stampInv" else 1
                held = ids32(U[:, ax] < lo[ax] + rng[ax] * 0.01)
                groups = [(held, zero3, zero3, None)]
                off = 1.1 * U[held[0], ax]  # *mesh.DBCVertexIds.begin(): the smallest held index
                W[:nSim, ax] = -0.1 * W[:nSim, ax] + off
            else:  # the bottom 5 % held, the top 5 % pulled by a Neumann acceleration of 600 in -x / +x
                bottom = U[:, 1] < lo[1] + rng[1] * 0.05
                groups = [(ids32(bottom), zero3, zero3, None)]
                neumann = [(ids32(~bottom & (U[:, 1] > hi[1] - rng[1] * 0.05)), (-600.0 if cfg.script == "NMFixBottomDragLeft" else 600.0, 0.0, 0.0), 0.0, float("inf"))]
        else:  # twist about x through the rest box centre + a pull along x; twistnsns turns the pull round like stretchnsquash
            w, v, out = {"twistnstretch": (18.0, 0.1, None), "twistnsns": (72.0, 1.2, 1.2), "twistnsns_old": (72.0, 0.9, 0.8)}[cfg.script]
            groups = [(b, ((-1.0) ** i * -v, 0.0, 0.0), ((-1.0) ** i * -w, 0.0, 0.0), ctr_rest) for i, b in enumerate((left, right))]
            if out is not None:
                turn = (int(left[0]), 0, U[left[0], 0] - out, U[left[0], 0] + 0.4, 0, [0, 1], False)
        keep = [k for k, g in enumerate(groups) if len(g[0])]
        remap = {k: i for i, k in enumerate(keep)}
        groups = [groups[k] for k in keep]
        hp_dirichlet = [(ids, lin, ang, 0.0, float("inf")) for ids, lin, ang, _c in groups]
        hp_motions = None if rule_zero or not groups else [(lin, ang, c) for _ids, lin, ang, c in groups]
        if let_go is not None:
            let_go["free"] = [remap[g] for g in let_go["free"] if g in remap]
            let_go["stop"] = [remap[g] for g in let_go["stop"] if g in remap]
            rule_release = let_go
        if turn is not None:
            rule_release = {"kind": "turn", "turn": turn[0], "axis": turn[1], "lo": float(turn[2]), "hi": float(turn[3]), "comp": turn[4],
                            "groups": [remap[g] for g in turn[5] if g in remap], "stop": turn[6], "lin": [m[0] for m in hp_motions],
                            "ang": [m[1] for m in hp_motions], "ctr": [m[2] for m in hp_motions], "done": False}
    codim_nodes = codim_mass = codim_fixed = None
    codim_edges = np.vstack(CEs).astype(np.int32) if CEs else np.zeros((0, 2), dtype=np.int32)
    if codim:
        codim_nodes = np.concatenate([ids for ids, _f, _m, _e in codim])
        m = np.zeros(V.shape[0])
        for ids, tri, _m, seg in codim:
            if len(tri):  # barycentric lumping of the triangle areas times the density (Mesh.cpp:318-343, 399)
                a = 0.5 * np.linalg.norm(np.cross(V[tri[:, 1]] - V[tri[:, 0]], V[tri[:, 2]] - V[tri[:, 0]]), axis=1)
                for k in range(3):
                    np.add.at(m, tri[:, k], cfg.rho * a / 3.0)
            elif len(seg):  # both ends of a segment of length l get density * l^3 pi / 12 (Mesh.cpp:279-295, 399)
                l = np.linalg.norm(V[seg[:, 0]] - V[seg[:, 1]], axis=1)
                for k in range(2):
                    np.add.at(m, seg[:, k], cfg.rho * l ** 3 * math.pi / 12.0)
            else:
                # a point carries the mean nodal mass of the tetrahedral components (avgNodeMass(dim), Mesh.cpp:403-411, 583-607): lumped
                # masses = density * a quarter of the adjacent element volumes
                tm = np.zeros(V.shape[0])
                if T.shape[0]:
                    vol = np.abs(np.einsum("ij,ij->i", np.cross(V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]]), V[T[:, 3]] - V[T[:, 0]])) / 6.0
                    for k in range(4):
                        np.add.at(tm, T[:, k], cfg.rho * vol / 4.0)
                n_tet_nodes = sum(nr[c + 1] - nr[c] for c in range(len(cfg.shapes)) if tr[c + 1] > tr[c])
                m[ids] = tm.sum() / n_tet_nodes if n_tet_nodes else 0.0
        codim_mass = m[codim_nodes]
        if cfg.script in ("DCOFix", "DCOBallHitWall"):  # AnimScripter.cpp:1222-1236: every codimensional component is held (NONZERO, no motion)
            dirichlet = []
            codim_fixed = codim_nodes
        elif cfg.script not in DCO_SCRIPTS and cfg.script not in ("DCOSegBedSquash", "meshSeqFromFile", "DCOHammerWalnut", "DCOCut") and not all(moved for _i, _f, moved, _e in codim):
            raise UnsupportedKeyword("codimensional shape that no script fixes or moves")
    elif cfg.script in ("DCOFix", "DCOBallHitWall"):
        dirichlet = []  # mesh.resetDBCVertices(); nothing to hold
    release = None
    if cfg.script == "dragright":
        # AnimScripter.cpp:809-826: lifted like `fall`, the nodes within 4 % of the right end of the body become a NONZERO handle
        # pulled at 0.5 in +x; stepAnimScript lets go once the whole body is right of every mesh obstacle (:1619-1632)
        # (the script works on mesh.V, the START positions: with `rotateModel` those are not the rest shape -- 13_dolphinFunnel.txt; found by
        # running that scene through the reference: lift and handle were taken from the rest shape before)
        U = V if V0 is None else V0
        U[:nSim, 1] += 0.5 * np.linalg.norm(U[:nSim].max(0) - U[:nSim].min(0))
        lo, hi = U[:nSim].min(0), U[:nSim].max(0)
        ids = np.nonzero(U[:nSim, 0] > hi[0] - 0.04 * (hi[0] - lo[0]))[0].astype(np.int32)
        dirichlet = [(ids, (0.5, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf"))]
        limit = max((V[o, 0].max() for o in obstacle), default=-np.inf)
        release = {"group": 0, "x_limit": float(limit), "nSim": int(nSim), "done": False}
    motions = None  # per Dirichlet group: (lin, ang in degrees, fixed centre or None) with the nodes typed NONZERO throughout
    if cfg.script in ("ACOSquash", "ACOSquash6"):  # AnimScripter.cpp:956-985: no Dirichlet nodes; planes 2 i / 2 i + 1 move at +1 / -1 along axis i
        nP = 2 if cfg.script == "ACOSquash" else 6
        if len(cfg.half_spaces) < nP:
            raise UnsupportedKeyword(f"script {cfg.script} needs {nP} analytic planes (ground / halfSpace)")
        dirichlet = []
        vel_p = [np.eye(3)[i // 2] * (1.0 if i % 2 == 0 else -1.0) for i in range(nP)]
        release = {"kind": "aco", "vel": vel_p, "origins": [np.array(h[0], dtype=np.float64) for h in cfg.half_spaces[:nP]],
                   "pairs": [(a, 0.1 if nP == 2 else 0.2) for a in range(0, nP, 2)], "done": False}
    if cfg.script == "meshSeqFromFile":
        # AnimScripter.cpp:1222-1236, 2126-2144: every component without tetrahedra is a NONZERO Dirichlet set; before each step the FIRST of them is
        # moved onto the positions of <folder>/<meshI>.obj, meshI counted from 1
        parts = [ids for ids, _f, _m, _e in codim]
        if not parts:
            raise UnsupportedKeyword("script meshSeqFromFile without a surface-only component to move")
        dirichlet = [(ids, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf")) for ids in parts]
        motions = [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), None)] * len(parts)
        seqs = [{"ids": parts[0], "folder": cfg.script_seq_folder, "ext": ".obj"}]
        codim_fixed = None
    if cfg.script in HOLD_SCRIPTS + PULL_SCRIPTS + RULE_SCRIPTS:  # resetDBCVertices(): the scripts replace whatever the shapes' own DBC keywords selected
        dirichlet, motions = hp_dirichlet, hp_motions
        if cfg.script in RULE_SCRIPTS:
            release = rule_release
    if cfg.script in DCO_SCRIPTS:
        spec = DCO_SCRIPTS[cfg.script]
        U = V if V0 is None else V0
        dirichlet, motions = [], []
        codim_comp = {int(ids[0]): True for ids, _f, _m, _e in codim}  # first node of every component without tetrahedra
        if cfg.script == "DCOSqueezeOut":
            # Every surface-only component is a NONZERO Dirichlet node set (AnimScripter.cpp:1261-1280); the first one carries a velocity of
            # 0.3 downwards that is applied while `topMax > bottomMin + (bottomMax - bottomMin) / 3.8 * 0.9` (:2102-2124).  As shipped that
            # test is never true: bottomMin starts at -infinity and is updated with std::min, so it STAYS -infinity and the right-hand side
            # is -inf + inf = NaN.  Seen in a run of the reference-compiled code (the plane does not move); restated as what it does.
            comps = [c for c in range(len(cfg.shapes)) if nr[c] in codim_comp]  # mesh.componentCoDim[compI] < 3
            for c in comps:
                dirichlet.append((np.arange(nr[c], nr[c + 1], dtype=np.int32), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf")))
                motions.append(((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), None))
            release = None
        else:
            n = spec["n"]
            if len(cfg.shapes) < n + 1:
                raise UnsupportedKeyword(f"script {cfg.script} needs {n} scripted components and a body")
            for c in range(n):
                ids = np.arange(nr[c], nr[c + 1], dtype=np.int32)
                lin = tuple(float(x) for x in spec["lin"][c]) if "lin" in spec else (0.0, 0.0, 0.0)
                ang = tuple(math.degrees(x) for x in spec["ang"][c]) if "ang" in spec else (0.0, 0.0, 0.0)
                ctr = tuple(0.5 * (U[ids].max(0) + U[ids].min(0))) if "ang" in spec else None  # MCORotCenter: fixed at set-up
                dirichlet.append((ids, lin, ang, 0.0, float("inf")))
                motions.append((lin, ang, ctr))
            release = {"kind": "dco", "script": cfg.script, "done": False, "lin": [m[0] for m in motions]} if "lin" in spec else None
            comps = list(range(n))
        if any(nr[c] in codim_comp and c not in comps for c in range(len(cfg.shapes))):
            raise UnsupportedKeyword("surface-only component that the script neither moves nor holds")
        codim_fixed = None
    if cfg.script in ("DCOHammerWalnut", "DCOCut"):
        # AnimScripter.cpp:1120-1170, 1993-2032: the SECOND component (whatever its kind) is a NONZERO set; while its lowest node is above 0.05 it
        # turns about z at pi / 6 through (x max, y min, z middle) of its start box (the hammer), resp. above 0.001 it moves at (0, -1, -1) (the knife)
        if len(cfg.shapes) < 2:
            raise UnsupportedKeyword(f"script {cfg.script} needs a second component to move")
        U = V if V0 is None else V0
        ids = np.arange(nr[1], nr[2], dtype=np.int32)
        lo, hi = U[ids].min(0), U[ids].max(0)
        if cfg.script == "DCOHammerWalnut":
            mot = ((0.0, 0.0, 0.0), (0.0, 0.0, 30.0), (float(hi[0]), float(lo[1]), float(0.5 * (hi[2] + lo[2]))))
            y_stop = 0.05
        else:
            mot = ((0.0, -1.0, -1.0), (0.0, 0.0, 0.0), None)
            y_stop = 0.001
        dirichlet, motions = [(ids, mot[0], mot[1], 0.0, float("inf"))], [mot]
        release = {"kind": "while_above", "ids": ids, "y": y_stop, "motion": mot, "moving": True, "done": False}
        if any(c != 1 for c in range(len(cfg.shapes)) if not tr[c + 1] > tr[c]):
            raise UnsupportedKeyword("surface-only component that the script neither moves nor holds")
        codim_fixed = None
    if cfg.script == "stretchAndPause":
        # AnimScripter.cpp:475-500: the nodes within 1 % of the left / right end of the model are NONZERO handles pulled apart at 1 in
        # -x / +x; the turning vertex is the LAST left handle in index order (`turningPointAdded` is never set), the limit x = -0.28
        U = V if V0 is None else V0
        lo, hi = U[:nSim].min(0), U[:nSim].max(0)
        left = np.nonzero(U[:nSim, 0] < lo[0] + 0.01 * (hi[0] - lo[0]))[0].astype(np.int32)
        right = np.nonzero((U[:nSim, 0] > hi[0] - 0.01 * (hi[0] - lo[0])) & ~(U[:nSim, 0] < lo[0] + 0.01 * (hi[0] - lo[0])))[0].astype(np.int32)
        dirichlet = [(left, (-1.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf")), (right, (1.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf"))]
        release = {"kind": "pause", "turn": int(left[-1]), "x_limit": -0.28, "groups": [0, 1], "ids": [left, right], "done": False}
    if cfg.script in HANDLE_SCRIPTS:
        U = V if V0 is None else V0
        lo, hi = U[:nSim].min(0), U[:nSim].max(0)  # mesh.V.colwise().minCoeff() / maxCoeff(): every node of Mesh<3>
        rng = hi - lo
        still = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), None)
        if cfg.script == "hangLeft":  # AnimScripter.cpp:191-204: the left border nodes (IglUtils::findBorderVerts, handleRatio 0.01) are held (ZERO)
            left, _right = _scene.border_verts(U[:nSim], cfg.handle_ratio)
            dirichlet = [(np.asarray(left, dtype=np.int32), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf"))]
        elif cfg.script == "fixLowerHalf":  # AnimScripter.cpp:337-350: the lower half of the model is held (NONZERO, no motion)
            ids = np.nonzero(U[:nSim, 1] < lo[1] + rng[1] * 0.5)[0].astype(np.int32)
            dirichlet, motions = [(ids, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf"))], [still]
        elif cfg.script == "pushRightMost1":  # :895-910, 1820-1826: the FIRST node within 1e-3 of the right end is pushed at 0.15 in -x
            ids = np.nonzero(U[:nSim, 0] > hi[0] - 1.0e-3 * rng[0])[0][:1].astype(np.int32)
            dirichlet, motions = [(ids, (-0.15, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf"))], [((-0.15, 0.0, 0.0), (0.0, 0.0, 0.0), None)]
        elif cfg.script == "utopiaComparison":
            # :1283-1302, 1641-1646: nodes within 1e-4 of the model's WIDTH (range[0], as written) of the top carry a Neumann acceleration of
            # 1.5 downwards, those as close to the bottom are held
            top = U[:nSim, 1] > hi[1] - rng[0] * 1e-4
            bottom = ~top & (U[:nSim, 1] < lo[1] + rng[0] * 1e-4)
            dirichlet, motions = [(np.nonzero(bottom)[0].astype(np.int32), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, float("inf"))], [still]
            neumann = [(np.nonzero(top)[0].astype(np.int32), (0.0, -1.5, 0.0), 0.0, float("inf"))]
        else:
            # DCOSegBedSquash (:1239-1259, 2080-2100): every component without tetrahedra is a NONZERO Dirichlet set; those in the upper half of
            # the component list (index >= (#components + 1) / 2) move down at 1 while the gap between the lowest node of the upper ones and the
            # highest node of the lower ones exceeds 0.1 -- the two beds of `17_pinCushionBall.txt` closing on the ball
            ncomp = len(cfg.shapes)
            parts = [c for c in range(ncomp) if not tr[c + 1] > tr[c]]
            if not parts:
                raise UnsupportedKeyword("script DCOSegBedSquash without a component to move")
            up = [c for c in parts if c >= (ncomp + 1) // 2]
            down = [c for c in parts if c < (ncomp + 1) // 2]
            cat = lambda cs: np.concatenate([np.arange(nr[c], nr[c + 1], dtype=np.int32) for c in cs]) if cs else np.zeros(0, np.int32)  # noqa: E731
            dirichlet, motions = [], []
            for ids, lin in ((cat(down), (0.0, 0.0, 0.0)), (cat(up), (0.0, -1.0, 0.0))):
                if len(ids):
                    dirichlet.append((ids, lin, (0.0, 0.0, 0.0), 0.0, float("inf")))
                    motions.append((lin, (0.0, 0.0, 0.0), None))
            release = {"kind": "segbed", "up": cat(up), "down": cat(down), "group": len(dirichlet) - 1, "moving": True, "done": False} if len(up) else None
            codim_fixed = None
    vel = np.zeros_like(V)
    fixed = np.zeros(V.shape[0], dtype=bool)
    for ids, *_ in dirichlet:
        fixed[ids] = True
    if cfg.script == "DCOBallHitWall":  # AnimScripter.cpp:1376-1389: every node of a tetrahedral component starts at v_x (default 1000)
        vx = cfg.script_params[0] if cfg.script_params and cfg.script_params[0] == cfg.script_params[0] else 1000.0
        for s in range(len(cfg.shapes)):
            if tr[s + 1] > tr[s]:
                vel[nr[s]:nr[s + 1], 0] = vx
    if cfg.script in INITVEL_SCRIPTS:  # AnimScripter::initVelocity (AnimScripter.cpp:1336-1374), over every node of Mesh<3>
        U = (V if V0 is None else V0)[:nSim]
        lo, hi = U.min(0), U.max(0)
        rng = hi - lo
        if cfg.script == "drop":
            vel[:nSim, 1] = -1.0
        elif cfg.script == "leftHitRight":
            vel[:nSim][U[:, 0] < lo[0] + rng[0] / 2.0, 0] = 1.0
        else:
            bottom = U[:, 1] < lo[1] + rng[1] * 0.01
            vel[:nSim][bottom, 0] = 1.0
            vel[:nSim][~bottom & (U[:, 1] > hi[1] - rng[1] * 0.01), 0] = -1.0
    for s, sh in enumerate(cfg.shapes):  # AnimScripter::initVelocity (AnimScripter.cpp:1319-1333): `initVel` under `script null` only
        if sh.init_vel is None or cfg.script != "null" or not tr[s + 1] > tr[s]:
            continue
        a, b = nr[s], nr[s + 1]
        ctr = 0.5 * (V[a:b].max(0) + V[a:b].min(0))
        w = np.radians(np.array(sh.init_vel[1]))
        v = np.array(sh.init_vel[0]) + np.cross(w, V[a:b] - ctr)
        v[fixed[a:b]] = 0.0
        vel[a:b] = v
    obst = np.concatenate(obstacle) if obstacle else None
    sc = AssembledScene(cfg, V, T, SF, nr, tr, dirichlet, vel, neumann, obst, release, codim_nodes, codim_mass, codim_fixed, V0)
    sc.motions = motions
    sc.codim_edges = codim_edges
    for q in seqs:  # the group a sequence drives = the entry of `dirichlet` that holds its nodes
        q["group"] = next(g for g, d in enumerate(dirichlet) if d[0] is q["ids"])
    sc.mesh_seqs = seqs
    if cfg.script == "meshSeqFromFile":
        sc.mesh_i = 1
    return sc


def apply(sc, be):
    """Drive a backend (the ctypes `Context` of ipc_amd/lib.py, or an adapter over the oracle with the same method names)."""
    cfg = sc.cfg
    be.set_mesh(sc.V, sc.T, YM=cfg.YM, PR=cfg.PR, density=cfg.rho)
    if sc.codim_nodes is not None:
        be.set_codim_nodes(sc.codim_nodes, sc.codim_mass)
    be.set_energy_type(cfg.energy)
    for s, sh in enumerate(cfg.shapes):
        if sh.material is not None and all(np.isfinite(sh.material)):
            be.set_component_material((sc.node_ranges[s], sc.node_ranges[s + 1]), (sc.tet_ranges[s], sc.tet_ranges[s + 1]), *sh.material)
    if sc.V0 is not None:
        be.set_positions(sc.V0)
    be.opt_init(cfg.dt, cfg.gravity)
    if cfg.time_integration == "NM":
        be.set_time_integration("NM", cfg.beta, cfg.gamma)
    if sc.codim_edges is not None and len(sc.codim_edges):
        be.set_surface(sc.SF, sc.codim_edges)
    else:
        be.set_surface(sc.SF)
    if sc.codim_fixed is not None:
        be.set_dbc(sc.codim_fixed, 2)
    self_fric = cfg.self_fric
    fric_scales = None
    if sc.obstacle_nodes is not None:
        # static obstacle: all of its nodes are ZERO Dirichlet nodes; without `selfCollisionOn` only pairs that involve the
        # obstacle collide.
        be.set_dbc(sc.obstacle_nodes, 1)
        be.set_obstacle(sc.obstacle_nodes, obstacle_only=not cfg.self_collision)
        # MeshCO::friction is read (Config.cpp:459-474) and stored, but no friction term is ever evaluated for a mesh collision object:
        # Optimizer::computeEnergyVal / computeGradient / computePrecondMtr add friction for the analytic objects and for the
        # self-collision set only (Optimizer.cpp:3357-3376, 3473-3510, 3676-3705); the value merely switches the lagging loop on
        # (:156-161).  Seen in a run of the reference-compiled code (cubeCliffCO.txt with and without the coefficient).  So pairs that
        # involve an obstacle node carry no friction, the others selfFric.
        self_fric = cfg.self_fric if cfg.self_collision else 0.0
        if self_fric > 0:
            fric_scales = (1.0, 0.0)
    if cfg.self_collision or sc.obstacle_nodes is not None:
        be.enable_self_collision(cfg.dHat_eps)
    for origin, normal, mu in cfg.half_spaces:
        idx = be.add_half_space(origin, normal, cfg.dHat_eps)
        if mu > 0:
            be.set_half_space_friction(idx, mu)
    plane_fric = any(mu > 0 for *_, mu in cfg.half_spaces)
    # Optimizer.cpp:146-166: solveFric is true as soon as ANY collision object -- a mesh collision object included -- or selfFric carries a
    # coefficient: the lagging loop with its tangent-space convergence solve then runs even when no pair carries friction
    loop_only = any(mc[3] > 0 for mc in cfg.mesh_cos) and not (self_fric > 0 or plane_fric)
    if self_fric > 0 or plane_fric or loop_only:
        be.set_friction(self_fric, cfg.fric_iter_amt, cfg.eps_v)
        if cfg.eps_v_target > 0 and cfg.eps_v_target != cfg.eps_v:
            be.set_friction_target(cfg.eps_v_target)
        if sc.obstacle_nodes is not None and fric_scales is not None:
            be.set_friction_scales(*fric_scales)
        if loop_only:
            be.force_friction_loop(True)
    if cfg.use_abs_parameters or cfg.dtol_rel != 1e-9 or cfg.kappa_min_multiplier != 1e11:
        be.set_parameter_scaling(cfg.use_abs_parameters, cfg.dtol_rel, cfg.kappa_min_multiplier)
    if cfg.kappa > 0:
        be.set_kappa(cfg.kappa)
    if 0 < cfg.dHat_target < cfg.dHat_eps:
        be.set_dhat_target(cfg.dHat_target)
    if cfg.damping_stiff > 0:
        be.set_damping(cfg.damping_stiff)
    for g, (ids, lin, ang, t0, t1) in enumerate(sc.dirichlet):
        t0, t1 = max(t0, cfg.dbc_time_range[0]), min(t1, cfg.dbc_time_range[1])  # a group acts where its own range and the scene's overlap
        be.add_dirichlet(ids, lin_vel=lin, ang_vel_deg=ang, t0=t0, t1=t1)
        if sc.motions is not None:
            be.set_dirichlet_motion(g, lin_vel=sc.motions[g][0], ang_vel_deg=sc.motions[g][1], center=sc.motions[g][2], force_nonzero=True)
    for ids, acc, t0, t1 in sc.neumann:
        be.add_neumann(ids, acc, t0=max(t0, cfg.nbc_time_range[0]), t1=min(t1, cfg.nbc_time_range[1]))
    if cfg.script == "twist":
        left, right = _scene.border_verts(sc.V, cfg.handle_ratio)
        be.set_twist(left, right)
    if np.any(sc.velocity):
        be.set_velocity(sc.velocity)
    if cfg.warm_start:
        be.set_warm_start(cfg.warm_start)
    be.set_rel_tol(cfg.tol)
    if cfg.restart:
        be.load_status(cfg.restart)
    be.precompute()
    return be
