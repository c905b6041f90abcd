// TEST INFRASTRUCTURE -- compiles the adapters of include/adapters/ against the Eigen-free stand-ins of the reference's
// interfaces (tests/mock_ipc/) -- and, where /root/reference exists, against the reference's own headers and compiled sources
// (tests/test_adapters.py::build_exe_ref) -- and, on a GPU, runs them:
//   1. Diagnostic.cpp:367-392 through the adapter CLASS: 10 isolated nodes, diagonal 10, rhs 1  =>  x = 0.1
//   2. composition on one shared context: HipElasticEnergy::computeHessian adds into the HipLinSysSolver it is handed
//      (values stay in HBM), host-side addCoeff / setCoeff of the reference's Optimizer land on top, factorize + solve;
//      the same energy adapter handed a plain host solver produces the same matrix (A/B inside one binary);
//      per-element arrays changed on the Mesh<3> side (a stiffer component) reach the device through hipUploadMesh.
// usage: test_adapters [compile-only]   (exit code 0 = pass; with "compile-only" nothing touches the GPU)
#include "HipElasticEnergy.hpp"
#include "HipLinSysSolver.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>

using namespace IPC;
typedef LinSysSolver<Eigen::VectorXi, Eigen::VectorXd> Solver;
typedef HipLinSysSolver<Eigen::VectorXi, Eigen::VectorXd> HipSolver;

// a host solver of the reference's kind (values in Base::a), standing in for CHOLMODSolver in the A/B run
class HostSolver : public Solver {
public:
    LinSysSolverType type() const override { return LinSysSolverType::EIGEN; }
    void analyze_pattern(void) override {}
    bool factorize(void) override { return true; }
    void solve(Eigen::VectorXd&, Eigen::VectorXd&) override {}
};

static int fails = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            ++fails;                                       \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                      \
            std::printf("\n");                             \
        }                                                  \
    } while (0)

// Kuhn-split bar nx x 1 x 1 cubes
static void make_bar(int nx, Mesh<3>& m)
{
    const int nvx = nx + 1, nV = nvx * 4, nT = nx * 6;
    m.V_rest.resize(nV, 3);
    auto nid = [&](int ix, int iy, int iz) { return ix + nvx * (iy + 2 * iz); };
    for (int iz = 0; iz < 2; ++iz)
        for (int iy = 0; iy < 2; ++iy)
            for (int ix = 0; ix < nvx; ++ix) {
                const int v = nid(ix, iy, iz);
                m.V_rest(v, 0) = 0.5 * ix;
                m.V_rest(v, 1) = 0.4 * iy;
                m.V_rest(v, 2) = 0.3 * iz;
            }
    m.V = m.V_rest;
    m.F.resize(nT, 4);
    const int perms[6][3] = { { 0, 1, 2 }, { 0, 2, 1 }, { 1, 0, 2 }, { 1, 2, 0 }, { 2, 0, 1 }, { 2, 1, 0 } };
    int t = 0;
    for (int cx = 0; cx < nx; ++cx)
        for (const auto& p : perms) {
            int o[3] = { 0, 0, 0 }, c[4];
            c[0] = nid(cx, 0, 0);
            for (int k = 0; k < 3; ++k) {
                o[p[k]] = 1;
                c[k + 1] = nid(cx + o[0], o[1], o[2]);
            }
            // orientation: positive rest volume
            double e[3][3];
            for (int k = 0; k < 3; ++k)
                for (int d = 0; d < 3; ++d) e[k][d] = m.V_rest(c[k + 1], d) - m.V_rest(c[0], d);
            const double det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0])
                + e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
            if (det < 0) std::swap(c[1], c[2]);
            for (int k = 0; k < 4; ++k) m.F(t, k) = c[k];
            ++t;
        }
    m.vNeighbor.assign(nV, std::set<int>());
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
                if (a != b) m.vNeighbor[m.F(e, a)].insert(m.F(e, b));
    m.vertexDBCType.assign(nV, DirichletBCType::NOT_DBC);
}

static int run_gpu()
{
    // ---- 1. Diagnostic.cpp:367-392 through the adapter class
    {
        HipSolver s;
        std::vector<std::set<int>> vNeighbor(10);
        s.set_pattern(vNeighbor, std::set<int>());
        s.setZero();
        for (int i = 0; i < 30; ++i) s.addCoeff(i, i, 10.0);
        s.addCoeff(5, 3, 99.0); // lower-triangle write: ignored (LinSysSolver.hpp:404)
        s.analyze_pattern();
        CHECK(s.factorize(), "factorize of 10 I reported not-PD");
        Eigen::VectorXd rhs(30), x;
        for (int i = 0; i < 30; ++i) rhs[i] = 1.0;
        s.solve(rhs, x);
        for (int i = 0; i < 30; ++i) CHECK(std::fabs(x[i] - 0.1) < 1e-15, "x[%d] = %.17g", i, x[i]);
        CHECK(s.coeffMtr(4, 4) == 10.0 && s.coeffMtr(3, 5) == 0.0, "coeffMtr through the device mirror");
        s.setCoeff(7, 7, -1.0);
        CHECK(!s.factorize(), "negative pivot not reported");
    }
    // ---- 2. composition on a shared context
    ipcgpu_ctx* ctx = nullptr;
    if (ipcgpu_ctx_create(0, &ctx) < 0) {
        std::printf("FAIL ctx: %s\n", ipcgpu_last_error());
        return 1;
    }
    {
        Mesh<3> mesh;
        make_bar(6, mesh);
        const int nV = (int)mesh.V_rest.rows(), nT = (int)mesh.F.rows();
        const double YM = 1e5, PR = 0.4, rho = 1000.0, dt = 0.025;
        // features as Mesh<3> would hold them: take the library's own computeFeatures, then change them on the Mesh side
        if (ipcgpu_set_mesh(ctx, nV, nT, mesh.V_rest.data(), mesh.F.data(), YM, PR, rho) < 0) {
            std::printf("FAIL set_mesh: %s\n", ipcgpu_last_error());
            return 1;
        }
        std::vector<double> A(9 * (size_t)nT), mass(nV);
        mesh.triArea.resize(nT);
        mesh.u.resize(nT);
        mesh.lambda.resize(nT);
        ipcgpu_get_features(ctx, A.data(), mesh.triArea.data(), mass.data(), mesh.u.data(), mesh.lambda.data());
        mesh.restTriInv.resize(nT);
        for (int t = 0; t < nT; ++t)
            for (int k = 0; k < 9; ++k) mesh.restTriInv[t].data()[k] = A[9 * (size_t)t + k];
        mesh.massMatrix.resize(nV, nV);
        for (int v = 0; v < nV; ++v) mesh.massMatrix.coeffRef(v, v) = mass[v];
        for (int t = nT / 2; t < nT; ++t) { // a stiffer second half, set on the Mesh<3> side only
            mesh.u[t] *= 3.0;
            mesh.lambda[t] *= 3.0;
        }
        mesh.vertexDBCType[0] = mesh.vertexDBCType[7] = DirichletBCType::ZERO;
        mesh.DBCVertexIds = { 0, 7 };
        for (int v = 0; v < nV; ++v) { // a deformed state
            mesh.V(v, 0) = mesh.V_rest(v, 0) * 1.05 + 0.01 * mesh.V_rest(v, 1);
            mesh.V(v, 1) = mesh.V_rest(v, 1) * 0.97;
            mesh.V(v, 2) = mesh.V_rest(v, 2) + 0.02 * mesh.V_rest(v, 0) * mesh.V_rest(v, 0);
        }
        hipUploadMesh(ctx, mesh, YM, PR, rho);
        ipcgpu_opt_init(ctx, dt, 0);
        std::vector<double> mu2(nT);
        ipcgpu_get_features(ctx, nullptr, nullptr, nullptr, mu2.data(), nullptr);
        CHECK(mu2[nT - 1] == mesh.u[nT - 1] && mu2[0] == mesh.u[0], "per-element Lame parameters did not reach the device");

        HipSolver hip(ctx);
        HostSolver host;
        HipElasticEnergy energy(ctx, 0);
        hip.set_pattern(mesh.vNeighbor, mesh.DBCVertexIds);
        host.set_pattern(mesh.vNeighbor, mesh.DBCVertexIds);
        CHECK(hip.getNumNonzeros() == host.getNumNonzeros(), "patterns differ");
        std::vector<AutoFlipSVD<Eigen::Matrix<double, 3, 3>>> svd;
        std::vector<Eigen::Matrix<double, 3, 3>> F;
        const double coef = dt * dt;
        // device path: values never leave HBM
        hip.setZero();
        energy.computeHessian(mesh, true, svd, F, coef, &hip, true, true);
        // host path (A/B): the same adapter into a plain host solver
        host.setZero();
        energy.computeHessian(mesh, true, svd, F, coef, &host, true, true);
        // what Optimizer::computePrecondMtr does next on the host (Optimizer.cpp:3638-3668): mass on the free diagonals,
        // unit diagonal on the projected Dirichlet rows
        for (Solver* s : { static_cast<Solver*>(&hip), static_cast<Solver*>(&host) })
            for (int v = 0; v < nV; ++v)
                for (int d = 0; d < 3; ++d) {
                    if (mesh.isProjectDBCVertex(v, true)) s->setCoeff(3 * v + d, 3 * v + d, 1.0);
                    else s->addCoeff(3 * v + d, 3 * v + d, mesh.massMatrix.coeff(v, v));
                }
        const Eigen::VectorXd& ah = host.get_a();
        const Eigen::VectorXd& ad = hip.get_a();
        double worst = 0, big = 0;
        for (long k = 0; k < ah.size(); ++k) {
            worst = std::max(worst, std::fabs(ah[k] - ad[k]));
            big = std::max(big, std::fabs(ah[k]));
        }
        CHECK(big > 0 && worst <= 1e-13 * big, "device-resident and host-assembled matrices differ: %.3e of %.3e", worst, big);
        CHECK(hip.coeffMtr(0, 0) == 1.0 && hip.coeffMtr(0, 3) == 0.0, "Dirichlet row: diag %.17g", hip.coeffMtr(0, 0));
        // the stiffer half must show in the matrix: compare against an assembly with uniform material
        hip.analyze_pattern();
        CHECK(hip.factorize(), "Newton matrix reported not-PD");
        Eigen::VectorXd rhs(3 * nV), x, Ax;
        for (int i = 0; i < 3 * nV; ++i) rhs[i] = std::sin(0.37 * i);
        hip.solve(rhs, x);
        hip.multiply(x, Ax);
        double rn = 0, bn = 0;
        for (int i = 0; i < 3 * nV; ++i) {
            rn += (Ax[i] - rhs[i]) * (Ax[i] - rhs[i]);
            bn += rhs[i] * rhs[i];
        }
        CHECK(std::sqrt(rn / bn) < 1e-11, "residual %.3e", std::sqrt(rn / bn));
        // energy / gradient / step filter through the adapter
        double E = -1;
        energy.computeEnergyVal(mesh, 1, svd, F, coef, E);
        CHECK(E > 0, "energy %.17g", E);
        Eigen::VectorXd g;
        energy.computeGradient(mesh, true, svd, F, coef, g, true);
        CHECK(g.size() == 3 * nV && g[0] == 0.0 && g[3 * 7 + 1] == 0.0, "projected gradient rows");
        double step = 1.0;
        Eigen::VectorXd p(3 * nV);
        for (int i = 0; i < 3 * nV; ++i) p[i] = 0.3 * std::cos(1.1 * i);
        energy.filterStepSize(mesh, p, step);
        CHECK(step > 0 && step <= 1.0, "step %.17g", step);
    }
    ipcgpu_ctx_destroy(ctx);
    return fails;
}

int main(int argc, char** argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "compile-only")) {
        std::printf("adapters compiled and linked\n");
        return 0;
    }
    const int f = run_gpu();
    std::printf(f ? "%d adapter check(s) failed\n" : "adapters ok (%d failures)\n", f);
    return f ? 1 : 0;
}
