// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
#pragma once
#include "orc_math.h"
#include <set>
#include <utility>
#include <vector>

namespace orc {

// Restatement of the parts of Mesh<3> the hot path reads (Mesh.hpp:58-171).
struct Mesh {
    int nV, nT;
    double density;
    int energyType = 0; // Config energyType: 0 "NH" (needs the element-inversion safeguard), 1 "FCR" (Config.cpp:23-24)
    std::vector<double> V_rest, V; // column-major nV x 3
    std::vector<int> F; // column-major nT x 4
    std::vector<int> dbcType; // DirichletBCType: 0 NOT_DBC, 1 ZERO, 2 NONZERO (Mesh.hpp:41-45)
    // Kinematic obstacle meshes (the reference's MeshCO, src/CollisionObject/MeshCO.cpp: a triangle mesh outside the simulated
    // mesh that only collides) ride along as surface-only components: nodes that belong to no tetrahedron (zero mass, Dirichlet),
    // their triangles in SF.  obstacle[v] marks them; with obstacleOnly the contact sets keep only primitive pairs that involve an
    // obstacle (a scene with `meshCO` but `selfCollisionOff`: Optimizer.cpp:2448-2470 then only asks the collision objects).
    std::vector<char> obstacle;
    bool obstacleOnly = false;
    bool pairAllowed(int a, int b) const { return !obstacleOnly || (!obstacle.empty() && (obstacle[a] || obstacle[b])); }
    int nElemNodes = 0; // nodes referenced by at least one element (mean nodal mass and bounding box are taken over these)
    std::vector<char> inMesh; // referenced by an element: the components of codimension 3 that matSpaceBBoxSize2(dim) / avgNodeMass(dim) run over
    void meshBBox();
    std::vector<M3> restTriInv;
    std::vector<double> triArea, mass, mu, lam;
    std::vector<std::set<std::pair<int, int>>> vFLoc;
    std::vector<std::set<int>> vNeighbor;
    std::vector<std::pair<int, int>> extraEdges; // contact connectivity (SelfCollisionHandler.cpp:330-415)
    double avgEdgeLen, bboxDiag2, bboxLo[3], bboxHi[3];
    // surface
    int nSF = 0;
    std::vector<int> SF; // column-major nSF x 3
    std::vector<int> SVI;
    std::vector<std::pair<int, int>> SFEdges;
    // symmetric-upper CSR, 0-based (LinSysSolver.hpp:46-150)
    std::vector<int> ia, ja;

    Mesh(int nV, int nT, const double* Vrest, const int* F, double YM, double PR, double density);
    int Fi(int t, int k) const { return F[t + nT * k]; }
    double Vx(int v, int c) const { return V[v + nV * c]; }
    bool isDBC(int v) const { return dbcType[v] != 0; }
    bool isProjectDBC(int v, bool projectDBC) const { return dbcType[v] == 1 || (dbcType[v] == 2 && projectDBC); } // Mesh.hpp:135-144
    void setSurface(int nSF, const int* SF, int nCE = 0, const int* CE = nullptr); // CE: codimensional segments (node pairs); nodes without any neighbour are codimensional points
    std::vector<int> codimPoints; // nodes of no tetrahedron, triangle or segment (`.pt` shapes, Mesh.cpp:916-920)
    bool exactPredicates = false; // the intersection checks of a USE_PREDICATES build (IglUtils.hpp:222-233, 280-294)
    bool checkInversion() const;
    bool inversionFree() const { return energyType == 1 || checkInversion(); } // Optimizer.cpp:252,517,545,2710: getNeedElemInvSafeGuard()
    M3 defGrad(int t) const;
    void buildPattern();
    int findEntry(int row, int col) const;
};

double elasticEnergy(const Mesh& m, double coef, double* perElem);
void elasticGradient(const Mesh& m, double coef, bool projectDBC, double* grad);
void elemHessian(const Mesh& m, int t, double coef, bool projectSPD, double H[144]);
void assembleHessian(const Mesh& m, double coef, bool projectDBC, double* a);
void addBlockToMatrix(const Mesh& m, double* a, const double* H12 /*12x12 col-major*/, const int vInd[4], int rowIndI);
void inversionStep(const Mesh& m, const double* p, double slackness, double* out);
double filterStepSize(const Mesh& m, const double* p, double stepSize); // Energy.cpp:565-581

} // namespace orc

struct orc_mesh {
    orc::Mesh m;
    orc_mesh(int nV, int nT, const double* V, const int* F, double YM, double PR, double rho)
        : m(nV, nT, V, F, YM, PR, rho) {}
};
