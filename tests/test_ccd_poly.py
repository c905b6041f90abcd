"""A second, independent statement of the per-pair continuous collision query (oracle/ccd_poly.py: the polynomial / interval form of
the published CTCD algorithm with thickness eta) against the conservative additive advancement the product and the oracle run
(oracle/orc_contact.cpp::accd).  CTCD itself is a binary dependency that is absent from /root/reference (CCD-Wrapper@23907da); the only
values the reference's own tests hold for it are twelve booleans at eta = 0 (tests/Collisions/CollisionConstraintTests.cpp:18-35,
83-99): both statements reproduce them, agree with closed-form times of impact, and agree with each other on random pairs."""
import numpy as np
import pytest

from oracle import ccd_poly

K_PT, K_EE = 2, 3


@pytest.mark.parametrize("u0y", [-1.1, 0.0, 1.1])
@pytest.mark.parametrize("u1y", [-1.1, 0.0, 1.1])
def test_reference_point_triangle_booleans(orc, u0y, u1y):
    v0 = np.array([0, 1, -0.5])
    tri = np.array([[-1, 0, 1], [0, 0, -1], [1, 0, 1.0]])  # the order the reference passes: v1, v3, v2
    X = np.vstack([v0, tri])
    V = np.vstack([[0, u0y, 0], [[0, u1y, 0]] * 3]).astype(float)
    hit = (-u0y + u1y >= 1)
    t = ccd_poly.toi(K_PT, X, V, 0.0)
    assert (t is not None) == hit
    if hit:  # the gap closes linearly: height 1, closing speed u1y - u0y
        assert abs(t - 1.0 / (u1y - u0y)) < 1e-12
        ta = orc.accd(K_PT, X, V, eta=1e-9, tmax=1.0)
        assert abs(ta - t) < 1e-7


@pytest.mark.parametrize("dy", [-2.0, 0.0, 2.0])
def test_reference_edge_edge_booleans(orc, dy):
    X = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, -1], [0, 1, 1.0]])
    V = np.array([[0, dy, 0], [0, dy, 0], [0, -dy, 0], [0, -dy, 0.0]])
    hit = dy >= 1.0
    t = ccd_poly.toi(K_EE, X, V, 0.0)
    assert (t is not None) == hit
    if hit:
        assert abs(t - 2.0 / (2 * dy)) < 1e-12  # distance 2, closing speed 2 dy


def test_thickened_queries_on_closed_forms(orc):
    # a point falling on the interior of a triangle: distance h - v t reaches eta at (h - eta) / v
    X = np.array([[0.2, 1.0, 0.1], [-1, 0, -1], [1, 0, -1], [0, 0, 1.5]])
    V = np.array([[0, -2.0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0.0]])
    for eta in (0.0, 0.05, 0.3):
        t = ccd_poly.toi(K_PT, X, V, eta)
        assert abs(t - (1.0 - eta) / 2.0) < 1e-12
        if eta > 0:
            assert abs(orc.accd(K_PT, X, V, eta=eta / 1.0, tmax=1.0) - t) < 1e-7
    # a point approaching a triangle VERTEX from outside: the rim's vertex-vertex test carries it
    X = np.array([[3.0, 0.0, -1.0], [-1, 0, -1], [1, 0, -1], [0, 0, 1.5]])
    V = np.array([[-4.0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0.0]])
    t = ccd_poly.toi(K_PT, X, V, 0.25)
    assert abs(t - (2.0 - 0.25) / 4.0) < 1e-12
    assert abs(orc.accd(K_PT, X, V, eta=0.25 / 2.0, tmax=1.0) - t) < 1e-7
    # crossing edges closing at speed 1 from distance 1
    X = np.array([[-1, 0, 0], [1, 0, 0], [0, 1, -1], [0, 1, 1.0]])
    V = np.array([[0, 0, 0], [0, 0, 0], [0, -1.0, 0], [0, -1.0, 0]])
    t = ccd_poly.toi(K_EE, X, V, 0.1)
    assert abs(t - 0.9) < 1e-12
    assert abs(orc.accd(K_EE, X, V, eta=0.1, tmax=1.0) - t) < 1e-7
    # parallel edges (the sextic degenerates: n = 0 throughout): the end-point tests decide
    X = np.array([[-1, 0, 0], [1, 0, 0], [-0.5, 1, 0], [0.5, 1, 0.0]])
    t = ccd_poly.toi(K_EE, X, V, 0.1)
    assert abs(t - 0.9) < 1e-12
    # no approach: no hit
    assert ccd_poly.toi(K_EE, X, -V, 0.1) is None


def test_polynomial_query_and_advancement_agree_on_random_pairs(orc):
    """600 random point-triangle / edge-edge pairs with random straight-line motions, thickness = 20 % of the start distance (what the
    reference passes: eta = (1 - slackness) * d0, slackness 0.8): the time the distance first reaches eta, by root isolation and by
    advancement.  The advancement stops within 1e-8 d0 of the gap from above, i.e. it is never later and at most 1e-8 d0 / (closing
    speed) earlier."""
    rng = np.random.default_rng(77)
    n_hit = 0
    worst = 0.0
    for i in range(600):
        kind = K_PT if i % 2 == 0 else K_EE
        X = rng.normal(size=(4, 3))
        V = rng.normal(size=(4, 3)) * rng.choice([0.5, 2.0, 6.0])
        d0 = np.sqrt(orc.unclassified_d2(kind, X))
        if d0 < 1e-3:
            continue
        eta = 0.2 * d0
        tp = ccd_poly.toi(kind, X, V, eta)
        ta = orc.accd(kind, X, V, eta=0.2, tmax=1.0)
        if tp is None:
            assert ta >= 1.0 - 1e-9, (i, ta)
            continue
        n_hit += 1
        assert ta <= tp + 1e-9, (i, ta, tp)  # conservative: never beyond the first contact with the eta-offset
        lp = np.linalg.norm(V - V.mean(0), axis=1).max() * 2 + 1e-300
        assert tp - ta <= 5e-8 * d0 / lp * 50 + 1e-7, (i, ta, tp)
        worst = max(worst, tp - ta)
    assert n_hit > 60
    print("polynomial vs advancement: hits", n_hit, "largest t_poly - t_advance", worst)
