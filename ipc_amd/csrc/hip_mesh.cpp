// HipMesh: the Mesh<3> data contract of the hot path, derived on the host once per mesh and uploaded.
#include "hip_ipc.h"
#include <algorithm>
#include <cmath>

namespace ipcgpu {

namespace {
inline double det3(const double X[9])
{
    return X[0] * (X[4] * X[8] - X[7] * X[5]) - X[3] * (X[1] * X[8] - X[7] * X[2]) + X[6] * (X[1] * X[5] - X[4] * X[2]);
}
// inverse of a column-major 3x3 through the adjugate
inline void inv3(const double X[9], double R[9])
{
    const double d = det3(X);
    R[0] = (X[4] * X[8] - X[7] * X[5]) / d;
    R[1] = (X[7] * X[2] - X[1] * X[8]) / d;
    R[2] = (X[1] * X[5] - X[4] * X[2]) / d;
    R[3] = (X[6] * X[5] - X[3] * X[8]) / d;
    R[4] = (X[0] * X[8] - X[6] * X[2]) / d;
    R[5] = (X[3] * X[2] - X[0] * X[5]) / d;
    R[6] = (X[3] * X[7] - X[6] * X[4]) / d;
    R[7] = (X[6] * X[1] - X[0] * X[7]) / d;
    R[8] = (X[0] * X[4] - X[3] * X[1]) / d;
}
} // namespace

void HipMesh::computeFeatures(int nV_, int nT_, const double* Vr, const int* Fc, double YM, double PR, double density,
    hipStream_t s)
{
    if (nV_ <= 0 || nT_ < 0) throw ArgError("set_mesh: bad sizes");
    ++featuresVersion;
    nV = nV_;
    nT = nT_;
    V_rest.assign(Vr, Vr + 3 * (size_t)nV);
    F.assign(Fc, Fc + 4 * (size_t)nT);
    for (size_t i = 0; i < F.size(); ++i)
        if (F[i] < 0 || F[i] >= nV) throw ArgError("set_mesh: tetrahedron index out of range");
    dbcType.assign(nV, 0);
    restTriInv.assign(9 * (size_t)nT, 0.0);
    triArea.assign(nT, 0.0);
    mass.assign(nV, 0.0);
    std::vector<std::pair<int, int>> edges;
    edges.reserve(12 * (size_t)nT);
    double edgeSum = 0.0;
    auto X = [&](int v, int c) { return Vr[v + (size_t)nV * c]; };
    for (int t = 0; t < nT; ++t) {
        const int v[4] = { Fc[t], Fc[t + (size_t)nT], Fc[t + 2 * (size_t)nT], Fc[t + 3 * (size_t)nT] };
        double X0[9], A[9];
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i) X0[i + 3 * k] = X(v[k + 1], i) - X(v[0], i);
        inv3(X0, A); // restTriInv = X0^-1 (Mesh.cpp:449)
        for (int k = 0; k < 9; ++k) restTriInv[(size_t)k * nT + t] = A[k];
        const double d = det3(X0);
        triArea[t] = d / 3 / 2; // Mesh.cpp:455
        const double vol = std::fabs(d) / 6.0; // Mesh.cpp:255-266
        for (int k = 0; k < 4; ++k) mass[v[k]] += vol / 4.0;
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b) {
                edges.emplace_back(v[a], v[b]);
                edges.emplace_back(v[b], v[a]);
            }
        for (int a = 0; a < 4; ++a) { // igl::avg_edge_length walks the columns cyclically: edges (0,1) (1,2) (2,3) (3,0)
            const int b = (a + 1) % 4;
            double l2 = 0;
            for (int i = 0; i < 3; ++i) {
                const double dd = X(v[a], i) - X(v[b], i);
                l2 += dd * dd;
            }
            edgeSum += std::sqrt(l2);
        }
    }
    // Mesh.cpp:460: the mean over four of the six edges of every tetrahedron (what libigl's avg_edge_length computes for a
    // 4-column F); a third of it is the cell size of the reference's spatial hash, which caps the full-CCD step (SpatialHash.hpp:603-618)
    avgEdgeLen = nT ? edgeSum / (4.0 * nT) : 0.0;
    for (int v = 0; v < nV; ++v) mass[v] *= density; // Mesh.cpp:399
    this->density = density;
    mu.assign(nT, YM / 2.0 / (1.0 + PR)); // Mesh.cpp:663-664
    lam.assign(nT, YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR));
    // vNeighbor (Mesh.cpp:470-479); the surface triangles join in addSurfaceEdges
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    nbPtr.assign(nV + 1, 0);
    for (auto& e : edges) nbPtr[e.first + 1]++;
    for (int v = 0; v < nV; ++v) nbPtr[v + 1] += nbPtr[v];
    nb.resize(edges.size());
    for (size_t i = 0; i < edges.size(); ++i) nb[i] = edges[i].second;
    // bounding box of the simulated material (Mesh::matSpaceBBoxSize2): nodes of elements only -- a kinematic obstacle riding
    // along as a surface-only component must not change dHat = dHatEps^2 * diagonal^2.  (Surface-only components that DO belong
    // to the mesh are declared afterwards: setCodimNodes.)
    inMesh.assign(nV, 0);
    for (int t = 0; t < nT; ++t)
        for (int k = 0; k < 4; ++k) inMesh[F[t + (size_t)nT * k]] = 1;
    meshBBox();
    // upload
    std::vector<double> aos(3 * (size_t)nV);
    for (int v = 0; v < nV; ++v)
        for (int c = 0; c < 3; ++c) aos[3 * (size_t)v + c] = X(v, c);
    d_x.upload(aos, s);
    d_xTilde.upload(aos, s);
    d_mass.upload(mass, s);
    d_A.upload(restTriInv, s);
    d_vol.upload(triArea, s);
    d_mu.upload(mu, s);
    d_lam.upload(lam, s);
    std::vector<int4> tets(nT);
    for (int t = 0; t < nT; ++t) tets[t] = make_int4(Fc[t], Fc[t + (size_t)nT], Fc[t + 2 * (size_t)nT], Fc[t + 3 * (size_t)nT]);
    d_tet.upload(tets.data(), tets.size(), s);
    uploadDBC(s);
    HIP_CHECK(hipStreamSynchronize(s));
}

void HipMesh::meshBBox()
{
    nElemNodes = 0;
    for (int v = 0; v < nV; ++v) nElemNodes += inMesh[v];
    bboxDiag2 = 0;
    for (int c = 0; c < 3; ++c) {
        double lo = 1e300, hi = -1e300;
        for (int v = 0; v < nV; ++v) {
            if (nElemNodes && !inMesh[v]) continue;
            lo = std::min(lo, V_rest[v + (size_t)nV * c]);
            hi = std::max(hi, V_rest[v + (size_t)nV * c]);
        }
        bboxLo[c] = lo;
        bboxHi[c] = hi;
        bboxDiag2 += (hi - lo) * (hi - lo);
    }
}

void HipMesh::setCodimNodes(int n, const int* ids, const double* nodeMass, hipStream_t s)
{
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= nV) throw ArgError("set_codim_nodes: node id out of range");
        if (!(nodeMass[i] >= 0.0)) throw ArgError("set_codim_nodes: negative mass");
    }
    // Nodes of Mesh<3> with their lumped area masses; the Optimizer's bounding box and mean mass (matSpaceBBoxSize2(dim),
    // avgNodeMass(dim): Mesh.cpp:576-637, Optimizer.cpp:101, 2220, 2232) run over the components of codimension 3 only, so these
    // nodes stay out of both.
    for (int i = 0; i < n; ++i) mass[ids[i]] = nodeMass[i];
    meshBBox();
    d_mass.upload(mass, s);
    HIP_CHECK(hipStreamSynchronize(s));
}

void HipMesh::addSurfaceEdges(int nSF, const int* SF, int nCE, const int* CE)
{
    // Mesh.cpp:480-487: every surface triangle's edges enter vNeighbor too.  For a tet component they only repeat tet edges; for
    // a surface-only component (shell, scripted collision surface) they are the only adjacency its nodes have, and the barrier
    // blocks of a stencil whose nodes share such a triangle land on them once the nodes are free (penalty phase of a moving DBC).
    std::vector<std::pair<int, int>> edges;
    edges.reserve(nb.size() + 6 * (size_t)nSF);
    for (int v = 0; v < nV; ++v)
        for (int k = nbPtr[v]; k < nbPtr[v + 1]; ++k) edges.emplace_back(v, nb[k]);
    for (int t = 0; t < nSF; ++t)
        for (int a = 0; a < 3; ++a) {
            const int i = SF[t + (size_t)nSF * a], j = SF[t + (size_t)nSF * ((a + 1) % 3)];
            if (i < 0 || i >= nV || j < 0 || j >= nV) throw ArgError("set_surface: node id out of range");
            edges.emplace_back(i, j);
            edges.emplace_back(j, i);
        }
    for (int e = 0; e < nCE; ++e) { // codimensional segments (`.seg` shapes, Mesh.cpp:490-493)
        const int i = CE[2 * (size_t)e], j = CE[2 * (size_t)e + 1];
        if (i < 0 || i >= nV || j < 0 || j >= nV || i == j) throw ArgError("set_surface: bad codimensional segment");
        edges.emplace_back(i, j);
        edges.emplace_back(j, i);
    }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    if (edges.size() == nb.size()) return;
    nbPtr.assign(nV + 1, 0);
    for (auto& e : edges) nbPtr[e.first + 1]++;
    for (int v = 0; v < nV; ++v) nbPtr[v + 1] += nbPtr[v];
    nb.resize(edges.size());
    for (size_t i = 0; i < edges.size(); ++i) nb[i] = edges[i].second;
}

void HipMesh::uploadDBC(hipStream_t s)
{
    d_dbc.upload(dbcType, s);
    HIP_CHECK(hipStreamSynchronize(s));
}

void HipMesh::setComponentMaterial(int nodeBegin, int nodeEnd, int tetBegin, int tetEnd, double rho, double YM, double PR, hipStream_t s)
{
    // Mesh::setLameParam, componentMaterial branch (Mesh.cpp:665-671)
    for (int v = nodeBegin; v < nodeEnd; ++v) mass[v] *= rho / density;
    for (int t = tetBegin; t < tetEnd; ++t) {
        mu[t] = YM / 2.0 / (1.0 + PR);
        lam[t] = YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR);
    }
    d_mass.upload(mass, s);
    d_mu.upload(mu, s);
    d_lam.upload(lam, s);
    HIP_CHECK(hipStreamSynchronize(s));
}

} // namespace ipcgpu
