// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
//
// CPU restatement of the time-step driver around the Newton loop, following
// Optimizer.cpp: solve() 510-602, fullyImplicit_IP() 1518-1819, solveSub_IP() 1822-2213,
// computeSearchDir() 2324-2355, postLineSearch() 2357-2445, computeConstraintSets() 2448-2470,
// lineSearch() 2662-2916, stepForward() 2919-2938, suggest/upperBound/initKappa 2216-2313,
// computeEnergyVal() 3199-3405, computeGradient() 3409-3545, computePrecondMtr() 3549-3720,
// computeXTilta() 1236-1257, and the `twist` script of AnimScripter.cpp:555-572,1674-1684.
#include "orc_api.h"
#include "orc_contact.h"
#include "orc_core.h"
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <map>
#include <omp.h>

using namespace orc;

// The reference's Optimizer constructor calls setTime(10.0, 0.025) (Optimizer.cpp:116) and derives fricDHat0 / fricDHatTarget (:290-303) and CN_MBC (:268)
// from that step size; the scene's dt arrives later (main.cpp:1398) through setTime (:421-429), which recomputes neither: they carry h = 0.025 always.
static constexpr double kCtorDtSq = 0.025 * 0.025;

struct orc_opt {
    Mesh* m;
    double dt, dtSq, gravity[3] = { 0, 0, 0 };
    // time integration (Config timeIntegration BE | NM beta gamma, Config.hpp:96, Config.cpp:112-118)
    int tit = 0;
    int warmStart = 0; // Config `warmStart` (initX option, Optimizer.cpp:925-1080)
    double warmStepSize = 0.0;
    double betaNM = 0.25, gammaNM = 0.5;
    std::vector<double> acceleration, dxElastic; // 3 v + c
    double elCoef() const { return tit == 1 ? dtSq * betaNM : dtSq; } // Optimizer.cpp:3205-3224, 3416-3434, 3618-3632
    int nthreads;
    double relGL2Tol = 1.0e-8, targetGRes = 0;
    bool absParameters = false; // useAbsParameters (Config.cpp:553-555)
    double dTolRel = 1.0e-9, kappaMinMultiplier = 1.0e11; // tuning[3] (Optimizer.cpp:102-106), Config.hpp:139
    double lenScale2() const { return absParameters ? 1.0 : m->bboxDiag2; }
    std::vector<double> velocity, xTilta, V_prev, searchDir, gradient, a;
    // lagged stiffness-proportional damping (Optimizer.cpp:3723-3735): the projected element Hessians at the state the last time
    // step ended in, times dampingStiff / dt, rows and columns of Dirichlet nodes dropped
    double dampingStiff = 0.0;
    double dHatTargetEps = -1.0; // tuning[2] (Optimizer.cpp:283-289): every time step starts at dHat and halves it down to this; < 0: no homotopy
    double kappaConfig = 0.0; // tuning[0] (Config.cpp:41-45): the stiffness a time step starts from, 0 = suggestKappa
    std::vector<double> dampH;
    std::vector<int> dampInd;
    std::map<int, double> angVel; // twist handles
    // Mesh::DirichletBCs (Mesh.hpp:23-39): `DBC bboxMin bboxMax linVel angVel [t0 t1]` of a shape line (Config.cpp:246-263), and
    // the scripted linear / angular velocity of whole components (componentLVels / componentAVels, AnimScripter.cpp:1413-1435)
    struct DBCGroup {
        std::vector<int> ids;
        double lin[3], ang[3], t0, t1;
        bool forceNonzero = false, hasCenter = false; // the hard-coded DCO scripts (AnimScripter.cpp:1060-1300, 1961-2135)
        double center[3] = { 0, 0, 0 };
        bool isZero() const { return !forceNonzero && lin[0] == 0 && lin[1] == 0 && lin[2] == 0 && ang[0] == 0 && ang[1] == 0 && ang[2] == 0; }
        std::vector<double> targets; // mesh-sequence motion (AnimScripter.cpp:1465-1528): where the nodes are to be after the next step (3 per node)
    };
    std::vector<DBCGroup> dbcGroups;
    // Mesh::NeumannBCs (Mesh.hpp:47-56): `NBC bboxMin bboxMax force [t0 t1]` of a shape line (Config.cpp:264-280); `force` is an
    // acceleration: the terms carry the nodal mass (Optimizer.cpp:3241-3250, 3452-3461)
    struct NBCGroup {
        std::vector<int> ids;
        double a[3], t0, t1;
    };
    std::vector<NBCGroup> nbcGroups;
    std::vector<int> baseDbcType; // types that do not come from a group (set_dbc, twist handles)
    double stepStartTime = 0, stepEndTime = 0; // AnimScripter.cpp:1406-1407
    // augmented-Lagrangian Dirichlet fallback (AnimScripter.cpp:2150-2157, 2280-2350; Optimizer.cpp:1826-1828, 2168-2203)
    std::vector<int> tpIds; // targetPos keys, ascending (std::map order)
    std::vector<double> tpPos, tpLam; // 3 per key
    double dist2Tol = 0, completedStep = 1.0, lastMove = 1.0, rhoDBC = 0.0, CN_MBC = 0.0;
    bool projDBC = true; // m_projectDBC
    double rotCenter[3];
    orc_chol* chol = nullptr;
    int innerIterAmt = 0, globalIterNum = 0, k = 0;
    double lastEnergyVal = 0, lastStepSize = 0, lastAlphaFeasible = 0;
    double timers[16] = { 0 };
    bool patternDirty = true;
    // self-contact (interior point)
    bool selfCollision = false;
    ContactSets cs;
    double dHatEps = 1.0e-3, dHat = 0, kappa = 0, dTol = 0;
    std::vector<std::pair<int, int>> curExtra; // contact connectivity inside the current pattern (vNeighbor_IP)
    std::vector<MMCVID> closeID; // closeMConstraintID / closeMConstraintVal (Optimizer.cpp:2396-2440)
    std::vector<double> closeVal;
    int lastCCDArg = -1, nFullCCD = 0, nPatternChanges = 0, dbcIncomplete = 0;
    int ccdMode = std::getenv("IPCGPU_CCD_MODE") ? (std::atoi(std::getenv("IPCGPU_CCD_MODE")) != 0) : 1; // 1: the reference's full sweep, 0: swept boxes over PT / EE
    // analytic half-space obstacles (animConfig.collisionObjects) with their active sets (activeSet[coI])
    std::vector<HalfSpace> planes;
    std::vector<std::vector<int>> hsSet;
    std::vector<std::pair<int, int>> closeHS; // closeConstraintID / Val (Optimizer.cpp:2364-2374, 2403-2415)
    std::vector<double> closeHSVal;
    // lagged friction: Optimizer.cpp:286-304, 1525-1600, 1615-1790
    double selfFric = 0.0, epsV = 1.0e-3, fricDHat0 = 0, fricDHat = -1.0;
    double epsVTarget = -1.0, fricDHatTarget = 0; // eps_v homotopy (tuning[5], Optimizer.cpp:296-303, 1717, 1776-1781); < 0: no homotopy
    // a kinematic mesh obstacle carries its own friction coefficient (MeshCO::friction, Config.cpp:459-474): selfFric holds the larger of the
    // two, the lagged normal forces of the stencils with / without an obstacle node are scaled by these factors (1 = the same coefficient)
    double fricScaleSelf = 1.0, fricScaleObst = 1.0;
    int fricIterAmt = 1, fricIterI = 0;
    std::vector<double> hsFric; // per half-space friction coefficient
    FrictionLag lag;
    std::vector<std::vector<int>> hsLagSet;
    std::vector<std::vector<double>> hsLambda;
    bool fricLoopForced = false; // a mesh collision object with a friction coefficient switches the lagging loop on (Optimizer.cpp:156-161)
    bool solveFric() const
    {
        if (fricLoopForced) return true;
        if (selfCollision && selfFric > 0.0) return true;
        for (double mu : hsFric)
            if (mu > 0.0) return true;
        return false;
    }
    bool ipOn() const { return selfCollision || !planes.empty(); }
    size_t nConstraints() const
    {
        size_t n = selfCollision ? cs.active.size() : 0;
        for (const auto& s : hsSet) n += s.size();
        return n;
    }
};

// the full CCD sweep of the search direction (Optimizer.cpp:1961-2021): as the reference runs it, or (mode 0) over the PT / EE pairs
// with overlapping swept boxes
static double fullCcd(orc_opt* o, double slackness, double stepSize)
{
    const Mesh& m = *o->m;
    if (o->ccdMode == 1) {
        int arg3[3];
        const double a = fullCcdReference(m, o->searchDir.data(), slackness, stepSize, nullptr, arg3, nullptr);
        o->lastCCDArg = arg3[1];
        return a;
    }
    std::vector<std::array<int, 2>> cand;
    sweptCandidates(m, o->searchDir.data(), stepSize, cand);
    return ccdStepBound(m, cand, o->searchDir.data(), slackness, stepSize, &o->lastCCDArg);
}


namespace {
struct Tic {
    double& acc;
    std::chrono::high_resolution_clock::time_point t0;
    explicit Tic(double& a) : acc(a), t0(std::chrono::high_resolution_clock::now()) {}
    ~Tic() { acc += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); }
};

void computeXTilta(orc_opt* o)
{
    Mesh& m = *o->m;
    for (int v = 0; v < m.nV; ++v)
        for (int c = 0; c < 3; ++c) {
            if (m.isDBC(v)) o->xTilta[v + m.nV * c] = o->V_prev[v + m.nV * c];
            else if (o->tit == 1) // Optimizer.cpp:1259-1277
                o->xTilta[v + m.nV * c] = o->V_prev[v + m.nV * c]
                    + (o->velocity[3 * v + c] * o->dt + o->betaNM * (o->dtSq * o->gravity[c]) + (0.5 - o->betaNM) * (o->dtSq * o->acceleration[3 * v + c]));
            else o->xTilta[v + m.nV * c] = o->V_prev[v + m.nV * c] + (o->velocity[3 * v + c] * o->dt + o->dtSq * o->gravity[c]);
        }
}

// Optimizer.cpp:3199-3353 (elasticity + inertia + barrier)
// computeDampingMtr (Optimizer.cpp:3723-3735): kept per element here (the reference keeps the same numbers summed into a second
// LinSysSolver; the products below differ from its CSR product by the order of the additions only)
void computeDampingMtr(orc_opt* o)
{
    if (!(o->dampingStiff > 0.0)) return;
    const Mesh& m = *o->m;
    o->dampH.assign(144 * (size_t)m.nT, 0.0);
    o->dampInd.assign(4 * (size_t)m.nT, 0);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < m.nT; ++t) {
        elemHessian(m, t, o->dampingStiff / o->dt, true, &o->dampH[144 * (size_t)t]);
        for (int k = 0; k < 4; ++k) {
            const int v = m.Fi(t, k);
            o->dampInd[4 * (size_t)t + k] = m.isProjectDBC(v, true) ? (-v - 1) : v; // projectDBC = true (Optimizer.hpp:273)
        }
    }
}
// y_e = D_e dx_e over the nodes that took part when D was built; dx = V - V_prev, zero where `skip` says so
template <class Skip>
static void dampingProduct(const orc_opt* o, Skip skip, double* y /* 3 nV, added to */, double* quad)
{
    const Mesh& m = *o->m;
    double q = 0.0;
    for (int t = 0; t < m.nT; ++t) {
        const double* H = &o->dampH[144 * (size_t)t];
        const int* ind = &o->dampInd[4 * (size_t)t];
        double dx[12];
        for (int k = 0; k < 4; ++k) {
            const int v = m.Fi(t, k);
            for (int c = 0; c < 3; ++c) dx[3 * k + c] = (ind[k] < 0 || skip(v)) ? 0.0 : m.V[v + m.nV * c] - o->V_prev[v + m.nV * c];
        }
        for (int k = 0; k < 4; ++k) {
            if (ind[k] < 0) continue;
            for (int i = 0; i < 3; ++i) {
                double s = 0.0;
                for (int j = 0; j < 12; ++j) s += H[(3 * k + i) + 12 * j] * dx[j];
                if (y) y[3 * ind[k] + i] += s;
                q += dx[3 * k + i] * s;
            }
        }
    }
    if (quad) *quad = q;
}

double computeEnergyVal(orc_opt* o)
{
    Mesh& m = *o->m;
    double E = elasticEnergy(m, o->elCoef(), nullptr);
    std::vector<double> ev(m.nV);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < m.nV; ++v) {
        double s = 0;
        for (int c = 0; c < 3; ++c) {
            double d = m.V[v + m.nV * c] - o->xTilta[v + m.nV * c];
            s += d * d;
        }
        ev[v] = s * m.mass[v] / 2.0;
    }
    double sum = 0;
    for (int v = 0; v < m.nV; ++v) sum += ev[v];
    E += sum;
    for (const auto& g : o->nbcGroups) { // Optimizer.cpp:3241-3250
        if (o->stepStartTime < g.t0 || o->stepStartTime >= g.t1) continue;
        for (int v : g.ids)
            if (!m.isDBC(v)) E -= o->dtSq * m.mass[v] * (m.V[v] * g.a[0] + m.V[v + m.nV] * g.a[1] + m.V[v + 2 * m.nV] * g.a[2]);
    }
    for (size_t i = 0; i < o->planes.size(); ++i) E += hsEnergy(m, o->planes[i], o->hsSet[i], o->dHat, o->kappa);
    if (o->selfCollision) E += contactEnergy(m, o->cs, o->dHat, o->kappa);
    if (o->fricDHat > 0.0) { // Optimizer.cpp:3357-3377
        for (size_t i = 0; i < o->planes.size(); ++i)
            if (o->hsFric[i] > 0.0 && !o->hsLagSet[i].empty())
                E += hsFrictionEnergy(m, o->V_prev.data(), o->planes[i], o->hsLagSet[i], o->hsLambda[i], o->hsFric[i], o->fricDHat);
        if (o->selfCollision && o->selfFric > 0.0 && !o->lag.set.empty())
            E += frictionEnergy(m, o->V_prev.data(), o->lag, o->fricDHat, o->selfFric);
    }
    if (o->dampingStiff > 0.0) { // Optimizer.cpp:3381-3400: displacement of the step, zero on every Dirichlet node
        double q = 0.0;
        dampingProduct(o, [&](int v) { return m.isDBC(v); }, nullptr, &q);
        E += 0.5 * q;
    }
    if (o->rhoDBC) { // augmentMDBCEnergy (AnimScripter.cpp:2303-2311; Optimizer.cpp:3402-3404)
        for (size_t t = 0; t < o->tpIds.size(); ++t) {
            const int v = o->tpIds[t];
            double dot = 0, sq = 0;
            for (int c = 0; c < 3; ++c) {
                const double d = m.V[v + m.nV * c] - o->tpPos[3 * t + c];
                dot += o->tpLam[3 * t + c] * d;
                sq += d * d;
            }
            E -= std::sqrt(m.mass[v]) * dot;
            E += o->rhoDBC / 2.0 * m.mass[v] * sq;
        }
    }
    return E;
}

// elasticity + inertia only (computeGradient with solveIP == false, as initKappa calls it, Optimizer.cpp:2243-2245)
void elasticInertiaGradient(orc_opt* o, bool projectDBC, double* g)
{
    Mesh& m = *o->m;
    elasticGradient(m, o->elCoef(), projectDBC, g);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < m.nV; ++v)
        if (!m.isProjectDBC(v, projectDBC))
            for (int c = 0; c < 3; ++c) g[3 * v + c] += m.mass[v] * (m.V[v + m.nV * c] - o->xTilta[v + m.nV * c]);
    for (const auto& nb : o->nbcGroups) { // Neumann terms, Optimizer.cpp:3452-3461
        if (o->stepStartTime < nb.t0 || o->stepStartTime >= nb.t1) continue;
        for (int v : nb.ids)
            if (!m.isDBC(v))
                for (int c = 0; c < 3; ++c) g[3 * v + c] -= o->dtSq * m.mass[v] * nb.a[c];
    }
}

// Optimizer.cpp:3409-3517
void computeGradient(orc_opt* o, bool projectDBC)
{
    Mesh& m = *o->m;
    elasticInertiaGradient(o, projectDBC, o->gradient.data());
    for (size_t i = 0; i < o->planes.size(); ++i) hsGradient(m, o->planes[i], o->hsSet[i], o->dHat, o->kappa, o->gradient.data());
    if (o->selfCollision) contactGradient(m, o->cs, o->dHat, o->kappa, projectDBC, o->gradient.data());
    if (o->fricDHat > 0.0) { // Optimizer.cpp:3474-3478, 3504-3506
        for (size_t i = 0; i < o->planes.size(); ++i)
            if (o->hsFric[i] > 0.0 && !o->hsLagSet[i].empty())
                hsFrictionGradient(m, o->V_prev.data(), o->planes[i], o->hsLagSet[i], o->hsLambda[i], o->hsFric[i], o->fricDHat, o->gradient.data());
        if (o->selfCollision && o->selfFric > 0.0 && !o->lag.set.empty())
            frictionGradient(m, o->V_prev.data(), o->lag, o->fricDHat, o->selfFric, o->gradient.data());
    }
    for (int v = 0; v < m.nV; ++v)
        if (m.isDBC(v) && m.isProjectDBC(v, projectDBC))
            for (int c = 0; c < 3; ++c) o->gradient[3 * v + c] = 0; // :3512-3516
    if (o->dampingStiff > 0.0) // :3519-3540
        dampingProduct(o, [&](int v) { return m.isDBC(v) && m.isProjectDBC(v, projectDBC); }, o->gradient.data(), nullptr);
    if (!projectDBC && o->rhoDBC) // augmentMDBCGradient (AnimScripter.cpp:2313-2321; Optimizer.cpp:3542-3544)
        for (size_t t = 0; t < o->tpIds.size(); ++t) {
            const int v = o->tpIds[t];
            for (int c = 0; c < 3; ++c) {
                o->gradient[3 * v + c] -= std::sqrt(m.mass[v]) * o->tpLam[3 * t + c];
                o->gradient[3 * v + c] += o->rhoDBC * m.mass[v] * (m.V[v + m.nV * c] - o->tpPos[3 * t + c]);
            }
        }
}

void rebuildPattern(orc_opt* o)
{
    Mesh& m = *o->m;
    {
        Tic t(o->timers[1]);
        m.extraEdges = o->curExtra;
        m.buildPattern();
        o->a.assign(m.ja.size(), 0.0);
    }
    {
        Tic t(o->timers[2]);
        if (o->chol) orc_chol_destroy(o->chol);
        o->chol = orc_chol_create(3 * m.nV, m.ia.data(), m.ja.data(), o->nthreads);
    }
    o->patternDirty = false;
}

// computePrecondMtr (Optimizer.cpp:3549-3720): pattern follows the contact connectivity, then elastic + mass + barrier
void computePrecondMtr(orc_opt* o, bool projectDBC)
{
    Mesh& m = *o->m;
    std::vector<std::pair<int, int>> extra;
    if (o->selfCollision && (o->cs.active.size() + o->cs.paraEE.size())) contactConnectivity(m, o->cs, extra);
    if (o->selfCollision && o->fricDHat > 0.0 && o->selfFric > 0.0) frictionConnectivity(o->lag, extra); // :3565-3566
    // only pairs that are not mesh edges change vNeighbor
    std::vector<std::pair<int, int>> fresh;
    for (const auto& e : extra)
        if (!m.vNeighbor[e.first].count(e.second)) fresh.push_back(e);
    if (o->patternDirty || fresh != o->curExtra) {
        if (!o->patternDirty) o->nPatternChanges++;
        o->curExtra = fresh;
        rebuildPattern(o);
    }
    Tic t(o->timers[0]);
    assembleHessian(m, o->elCoef(), projectDBC, o->a.data());
    for (size_t i = 0; i < o->planes.size(); ++i) hsHessian(m, o->planes[i], o->hsSet[i], o->dHat, o->kappa, projectDBC, o->a.data());
    if (o->selfCollision) contactHessian(m, o->cs, o->dHat, o->kappa, projectDBC, o->a.data());
    if (o->fricDHat > 0.0) { // Optimizer.cpp:3677-3702
        for (size_t i = 0; i < o->planes.size(); ++i)
            if (o->hsFric[i] > 0.0 && !o->hsLagSet[i].empty())
                hsFrictionHessian(m, o->V_prev.data(), o->planes[i], o->hsLagSet[i], o->hsLambda[i], o->hsFric[i], o->fricDHat, projectDBC, o->a.data());
        if (o->selfCollision && o->selfFric > 0.0 && !o->lag.set.empty())
            frictionHessian(m, o->V_prev.data(), o->lag, o->fricDHat, o->selfFric, projectDBC, o->a.data());
    }
    if (o->dampingStiff > 0.0) { // addCoeff(dampingMtr, 1.0), Optimizer.cpp:3707-3709
        std::vector<int> ind(4);
        for (int t = 0; t < m.nT; ++t) {
            for (int k = 0; k < 4; ++k) {
                const int v = m.Fi(t, k);
                ind[k] = (o->dampInd[4 * (size_t)t + k] < 0 || m.isProjectDBC(v, projectDBC)) ? (-v - 1) : v;
            }
            for (int k = 0; k < 4; ++k)
                if (ind[k] >= 0) addBlockToMatrix(m, o->a.data(), &o->dampH[144 * (size_t)t], ind.data(), k);
        }
    }
    if (!projectDBC && o->rhoDBC) // augmentMDBCHessian (AnimScripter.cpp:2323-2337; Optimizer.cpp:3711-3713)
        for (int v : o->tpIds)
            for (int c = 0; c < 3; ++c) o->a[m.ia[3 * v + c]] += o->rhoDBC * m.mass[v]; // the diagonal leads every upper-CSR row
}

void stepForward(orc_opt* o, const std::vector<double>& V0, double alpha)
{
    Mesh& m = *o->m;
    for (int v = 0; v < m.nV; ++v)
        for (int c = 0; c < 3; ++c) m.V[v + m.nV * c] = V0[v + m.nV * c] + alpha * o->searchDir[3 * v + c];
}

void computeConstraintSets(orc_opt* o)
{
    if (!o->ipOn()) return;
    Tic t(o->timers[14]);
    for (size_t i = 0; i < o->planes.size(); ++i) hsConstraintSet(*o->m, o->planes[i], o->dHat, o->hsSet[i]); // Optimizer.cpp:2460-2462
    if (o->selfCollision) computeConstraintSet(*o->m, o->dHat, false, o->cs);
}

// multipliers, closest points and tangent bases of the current constraint sets (Optimizer.cpp:1553-1600 / 1620-1675)
void updateFrictionLag(orc_opt* o)
{
    if (!o->solveFric()) return;
    Mesh& m = *o->m;
    for (size_t i = 0; i < o->planes.size(); ++i)
        if (o->hsFric[i] > 0.0) {
            hsFrictionLagUpdate(m, o->planes[i], o->hsSet[i], o->dHat, o->kappa, o->hsLambda[i]);
            o->hsLagSet[i] = o->hsSet[i];
        }
    if (o->selfCollision && o->selfFric > 0.0) {
        frictionLagUpdate(m, o->cs.active, o->dHat, o->kappa, o->lag);
        if ((o->fricScaleSelf != 1.0 || o->fricScaleObst != 1.0) && !m.obstacle.empty())
            for (size_t i = 0; i < o->lag.set.size(); ++i) {
                bool obst = false;
                for (int k = 0; k < 4; ++k) {
                    const int e = o->lag.set[i][k];
                    const int v = e >= 0 ? e : -e - 1;
                    // entry 0 is the point / first edge node (negative = -v-1 for point stencils); negative entries behind it are
                    // "absent" markers or multiplicities, not nodes
                    if (k == 0 || e >= 0) obst = obst || m.obstacle[v];
                }
                o->lag.lambda[i] *= obst ? o->fricScaleObst : o->fricScaleSelf;
            }
    }
}

bool anyIntersection(orc_opt* o)
{
    // isIntersected (Optimizer.cpp:2626-2659): analytic objects first, then the mesh against itself
    for (const auto& h : o->planes)
        if (hsIntersected(*o->m, h)) return true;
    return o->selfCollision && isIntersected(*o->m);
}

double kappaFloor(orc_opt* o)
{
    // suggestKappa (Optimizer.cpp:2228-2233): kappaMinMultiplier (1e11, Config.hpp:139) * mean nodal mass / (4e-16 L^2 b''(1e-16 L^2))
    const Mesh& m = *o->m;
    double Hb;
    barrier(1.0e-16 * m.bboxDiag2, o->dHat, nullptr, nullptr, &Hb);
    double avgMass = 0; // Mesh::avgNodeMass(dim): over the nodes of the tetrahedral components (Mesh.cpp:576-609)
    for (int v = 0; v < m.nV; ++v)
        if (!m.nElemNodes || m.inMesh[v]) avgMass += m.mass[v];
    avgMass /= std::max(m.nElemNodes, 1);
    return o->kappaMinMultiplier * avgMass / (4.0e-16 * m.bboxDiag2 * Hb);
}

void initKappa(orc_opt* o)
{
    // Optimizer.cpp:2236-2313
    Mesh& m = *o->m;
    if (!o->nConstraints()) return;
    std::vector<double> gE(3 * m.nV), gc(3 * m.nV, 0.0);
    elasticInertiaGradient(o, true, gE.data());
    if (o->dampingStiff > 0.0) // computeGradient with solveIP off still adds the damping force (Optimizer.cpp:3519-3540)
        dampingProduct(o, [&](int v) { return o->m->isDBC(v); }, gE.data(), nullptr);
    for (size_t i = 0; i < o->planes.size(); ++i) hsGradient(m, o->planes[i], o->hsSet[i], o->dHat, 1.0, gc.data());
    if (o->selfCollision) {
        ContactSets only;
        only.active = o->cs.active;
        contactGradient(m, only, o->dHat, 1.0, true, gc.data());
    }
    for (int v = 0; v < m.nV; ++v)
        if (m.isDBC(v)) gc[3 * v] = gc[3 * v + 1] = gc[3 * v + 2] = 0.0; // :2275-2277
    double num = 0, den = 0;
    for (int i = 0; i < 3 * m.nV; ++i) {
        num += gc[i] * gE[i];
        den += gc[i] * gc[i];
    }
    double minKappa = -num / den;
    if (minKappa > 0.0) o->kappa = minKappa;
    minKappa = kappaFloor(o);
    if (o->kappa < minKappa) o->kappa = minKappa;
    const double kappaMax = 100 * kappaFloor(o); // upperBoundKappa, :2216-2225
    if (o->kappa > kappaMax) o->kappa = kappaMax;
}

double evalMMCVID(const Mesh& m, const MMCVID& c)
{
    ContactSets one;
    one.active.push_back(c);
    // distance of a single stencil: reuse the energy path would add the barrier; do it directly
    int node[4], n, kind;
    if (c[0] >= 0) {
        kind = K_EE;
        n = 4;
        for (int k = 0; k < 4; ++k) node[k] = c[k];
    }
    else {
        node[0] = -c[0] - 1;
        node[1] = c[1];
        if (c[2] < 0) {
            kind = K_PP;
            n = 2;
        }
        else if (c[3] < 0) {
            kind = K_PE;
            n = 3;
            node[2] = c[2];
        }
        else {
            kind = K_PT;
            n = 4;
            node[2] = c[2];
            node[3] = c[3];
        }
    }
    double X[4][3] = { { 0 } }, d;
    for (int k = 0; k < n; ++k)
        for (int cc = 0; cc < 3; ++cc) X[k][cc] = m.Vx(node[k], cc);
    stencil_distance(kind, X, &d, nullptr, nullptr);
    return d;
}

void postLineSearch(orc_opt* o)
{
    // Optimizer.cpp:2357-2445 (ADAPTIVE_KAPPA)
    if (!o->ipOn()) return;
    Mesh& m = *o->m;
    if (o->kappa == 0.0) {
        initKappa(o);
        return;
    }
    bool updateKappa = false;
    for (size_t i = 0; i < o->closeHS.size() && !updateKappa; ++i) {
        const double dist = o->planes[o->closeHS[i].first].dist(m, o->closeHS[i].second);
        if (dist * dist <= o->closeHSVal[i]) updateKappa = true;
    }
    for (size_t i = 0; i < o->closeID.size() && !updateKappa; ++i)
        if (evalMMCVID(m, o->closeID[i]) <= o->closeVal[i]) {
            updateKappa = true;
            break;
        }
    if (updateKappa) {
        o->kappa *= 2.0;
        const double kappaMax = 100 * kappaFloor(o);
        if (o->kappa > kappaMax) o->kappa = kappaMax;
    }
    o->closeID.clear();
    o->closeVal.clear();
    o->closeHS.clear();
    o->closeHSVal.clear();
    for (size_t i = 0; i < o->planes.size(); ++i)
        for (int v : o->hsSet[i]) {
            const double dist = o->planes[i].dist(m, v);
            if (dist * dist < o->dTol) {
                o->closeHS.push_back({ (int)i, v });
                o->closeHSVal.push_back(dist * dist);
            }
        }
    if (o->selfCollision)
        for (const auto& c : o->cs.active) {
            const double d = evalMMCVID(m, c);
            if (d < o->dTol) {
                o->closeID.push_back(c);
                o->closeVal.push_back(d);
            }
        }
}

// Optimizer.cpp:2662-2916 with armijoParam = 0, lowerBound = 0 (the IP call site, :2059)
void lineSearch(orc_opt* o, double& stepSize)
{
    Mesh& m = *o->m;
    {
        Tic t(o->timers[9]);
        o->lastEnergyVal = computeEnergyVal(o); // :2681, with the current constraint set and kappa
    }
    std::vector<double> V0 = m.V;
    {
        Tic t(o->timers[5]);
        stepForward(o, V0, stepSize);
        while (!m.inversionFree()) {
            stepSize /= 2.0;
            stepForward(o, V0, stepSize);
        }
        if (o->ipOn())
            while (anyIntersection(o)) { // :2719-2736
                stepSize /= 2.0;
                stepForward(o, V0, stepSize);
            }
    }
    computeConstraintSets(o);
    double testingE;
    {
        Tic t(o->timers[9]);
        testingE = computeEnergyVal(o);
    }
    const double LFStepSize = stepSize;
    while (testingE > o->lastEnergyVal && stepSize > 0.0) {
        stepSize /= 2.0;
        if (stepSize == 0.0) break;
        {
            Tic t(o->timers[5]);
            stepForward(o, V0, stepSize);
        }
        computeConstraintSets(o);
        Tic t(o->timers[9]);
        testingE = computeEnergyVal(o);
    }
    if (stepSize < LFStepSize && o->ipOn()) { // :2799-2811
        bool needRecomputeCS = false;
        while (anyIntersection(o)) {
            stepSize /= 2.0;
            stepForward(o, V0, stepSize);
            needRecomputeCS = true;
        }
        if (needRecomputeCS) computeConstraintSets(o); // lastEnergyVal keeps the pre-halving value, as in the reference
    }
    o->lastEnergyVal = testingE;
}
} // namespace

extern "C" {

orc_opt* orc_opt_create(orc_mesh* mh, double dt, int withGravity, int nthreads)
{
    orc_opt* o = new orc_opt;
    o->m = &mh->m;
    Mesh& m = *o->m;
    o->dt = dt;
    o->dtSq = dt * dt;
    if (withGravity) o->gravity[1] = -9.80665; // Optimizer.cpp:112-115
    o->CN_MBC = std::sqrt(1.0e-4 * m.bboxDiag2 * kCtorDtSq); // Optimizer.cpp:268 -- evaluated in the constructor, see kCtorDtSq
    o->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
    omp_set_num_threads(o->nthreads);
    o->velocity.assign(3 * m.nV, 0.0);
    o->searchDir.assign(3 * m.nV, 0.0);
    o->gradient.assign(3 * m.nV, 0.0);
    o->V_prev = m.V;
    o->xTilta = m.V;
    for (int c = 0; c < 3; ++c) o->rotCenter[c] = 0.5 * (m.bboxLo[c] + m.bboxHi[c]);
    orc_opt_set_rel_tol(o, 1.0e-2); // main.cpp:159 -> setRelGL2Tol() default, Optimizer.hpp:148
    computeXTilta(o);
    return o;
}
void orc_opt_destroy(orc_opt* o)
{
    if (o->chol) orc_chol_destroy(o->chol);
    delete o;
}
void orc_opt_set_rel_tol(orc_opt* o, double relTol)
{
    o->relGL2Tol = relTol * relTol;
    o->targetGRes = std::sqrt(o->relGL2Tol * (o->absParameters ? 1.0 : o->m->bboxDiag2 * o->dtSq)); // Optimizer.cpp:2941-2945
}
void orc_opt_set_velocity(orc_opt* o, const double* vel3nV)
{
    o->velocity.assign(vel3nV, vel3nV + 3 * o->m->nV);
    computeXTilta(o);
}
void orc_opt_enable_self_collision(orc_opt* o, double dHatEps)
{
    // `selfCollisionOn` + interior point; dHat = dHatEps^2 * bbox diagonal^2 (Optimizer.cpp:1534-1537, Config.cpp:41-45)
    o->selfCollision = true;
    o->dHatEps = dHatEps;
    o->dHat = dHatEps * dHatEps * o->lenScale2();
    o->dTol = o->dTolRel * o->dTolRel * o->lenScale2(); // dTolRel = tuning[3], 1e-9 by default (Optimizer.cpp:102-109)
}
int orc_opt_add_half_space(orc_opt* o, const double* origin3, const double* normal3, double dHatEps)
{
    // `ground` / `halfSpace` script keywords (Config.cpp:306-345) -> animConfig.collisionObjects; friction is a 8f row
    HalfSpace h;
    h.init(origin3, normal3);
    o->planes.push_back(h);
    o->hsSet.emplace_back();
    o->hsFric.push_back(0.0);
    o->hsLagSet.emplace_back();
    o->hsLambda.emplace_back();
    o->dHatEps = dHatEps;
    o->dHat = dHatEps * dHatEps * o->lenScale2();
    o->dTol = o->dTolRel * o->dTolRel * o->lenScale2();
    return (int)o->planes.size() - 1;
}
int orc_opt_get_half_space_set(const orc_opt* o, int id, int* verts)
{
    if (verts)
        for (size_t i = 0; i < o->hsSet[id].size(); ++i) verts[i] = o->hsSet[id][i];
    return (int)o->hsSet[id].size();
}
void orc_opt_set_twist(orc_opt* o, int nL, const int* left, int nR, const int* right, double angVel)
{
    // AnimScripter.cpp:555-572: handle set bI gets (-1)^bI * -angVel (angVel = 0.4 pi in the reference), DBC type NONZERO
    Mesh& m = *o->m;
    for (int i = 0; i < nL; ++i) {
        m.dbcType[left[i]] = 2;
        o->angVel[left[i]] = -angVel;
    }
    for (int i = 0; i < nR; ++i) {
        m.dbcType[right[i]] = 2;
        o->angVel[right[i]] = angVel;
    }
    computeXTilta(o);
}

// AnimScripter::setDBCVertices (AnimScripter.cpp:58-110): types of the groups that are active at stepStartTime on top of the
// static ones; NONZERO overrides ZERO
static void setDBCVertices(orc_opt* o)
{
    Mesh& m = *o->m;
    if (o->dbcGroups.empty()) return;
    if (o->baseDbcType.empty()) o->baseDbcType = m.dbcType;
    std::vector<int> t = o->baseDbcType;
    for (const auto& g : o->dbcGroups) {
        if (o->stepStartTime < g.t0 || o->stepStartTime >= g.t1) continue;
        const int type = g.isZero() ? 1 : 2;
        for (int v : g.ids) t[v] = std::max(t[v], type);
    }
    m.dbcType = t;
}

// searchDir += R (x - c) + c + linVel dt - x, R = Rx Ry Rz of angVel dt, c = centre of the group's current bounding box
// (AnimScripter.cpp:1440-1462)
static void dbcGroupMotion(orc_opt* o, const orc_opt::DBCGroup& g)
{
    Mesh& m = *o->m;
    const double ax = g.ang[0] * o->dt, ay = g.ang[1] * o->dt, az = g.ang[2] * o->dt;
    const double cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
    const double Rx[9] = { 1, 0, 0, 0, cx, -sx, 0, sx, cx }, Ry[9] = { cy, 0, sy, 0, 1, 0, -sy, 0, cy }, Rz[9] = { cz, -sz, 0, sz, cz, 0, 0, 0, 1 };
    double T[9], R[9]; // row-major
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = Rx[3 * i] * Ry[j] + Rx[3 * i + 1] * Ry[3 + j] + Rx[3 * i + 2] * Ry[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = T[3 * i] * Rz[j] + T[3 * i + 1] * Rz[3 + j] + T[3 * i + 2] * Rz[6 + j];
    double lo[3], hi[3];
    for (int c = 0; c < 3; ++c) lo[c] = hi[c] = m.V[g.ids[0] + m.nV * c];
    for (int v : g.ids)
        for (int c = 0; c < 3; ++c) {
            lo[c] = std::min(lo[c], m.V[v + m.nV * c]);
            hi[c] = std::max(hi[c], m.V[v + m.nV * c]);
        }
    double ctr[3];
    for (int c = 0; c < 3; ++c) ctr[c] = g.hasCenter ? g.center[c] : (lo[c] + hi[c]) / 2;
    for (int v : g.ids) {
        const double d[3] = { m.V[v] - ctr[0], m.V[v + m.nV] - ctr[1], m.V[v + 2 * m.nV] - ctr[2] };
        for (int c = 0; c < 3; ++c)
            o->searchDir[3 * v + c] += (R[3 * c] * d[0] + R[3 * c + 1] * d[1] + R[3 * c + 2] * d[2]) + ctr[c] + g.lin[c] * o->dt - m.V[v + m.nV * c];
    }
}

void orc_opt_set_dirichlet_motion(orc_opt* o, int group, const double* lin3, const double* ang3, const double* center3, int forceNonzero)
{
    if (group < 0 || group >= (int)o->dbcGroups.size()) return;
    orc_opt::DBCGroup& g = o->dbcGroups[group];
    for (int c = 0; c < 3; ++c) {
        g.lin[c] = lin3[c];
        g.ang[c] = ang3[c];
        if (center3) g.center[c] = center3[c];
    }
    g.hasCenter = center3 != nullptr;
    g.forceNonzero = forceNonzero != 0;
    setDBCVertices(o);
}
void orc_opt_set_dirichlet_targets(orc_opt* o, int group, int n, const double* targets)
{
    if (group < 0 || group >= (int)o->dbcGroups.size()) return;
    orc_opt::DBCGroup& g = o->dbcGroups[group];
    if (!targets || n != (int)g.ids.size()) g.targets.clear();
    else g.targets.assign(targets, targets + 3 * (size_t)n);
}
void orc_opt_add_neumann(orc_opt* o, int n, const int* ids, const double* accel3, double t0, double t1)
{
    orc_opt::NBCGroup g;
    g.ids.assign(ids, ids + n);
    for (int c = 0; c < 3; ++c) g.a[c] = accel3[c];
    g.t0 = t0;
    g.t1 = t1;
    o->nbcGroups.push_back(g);
}

void orc_opt_end_dirichlet(orc_opt* o, int group, double t_end)
{
    if (group >= 0 && group < (int)o->dbcGroups.size()) o->dbcGroups[group].t1 = std::min(o->dbcGroups[group].t1, t_end);
}
void orc_opt_add_dirichlet(orc_opt* o, int n, const int* ids, const double* lin3, const double* angRad3, double t0, double t1)
{
    orc_opt::DBCGroup g;
    g.ids.assign(ids, ids + n);
    for (int c = 0; c < 3; ++c) {
        g.lin[c] = lin3[c];
        g.ang[c] = angRad3[c];
    }
    g.t0 = t0;
    g.t1 = t1;
    if (o->baseDbcType.empty()) o->baseDbcType = o->m->dbcType;
    o->dbcGroups.push_back(g);
    setDBCVertices(o); // initAnimScript (AnimScripter.cpp:122-124)
    computeXTilta(o);
}

int orc_opt_precompute(orc_opt* o)
{
    // Optimizer.cpp:258-263: "intersection detected in initial configuration!" ends the reference's process
    if (anyIntersection(o)) return -1;
    // Optimizer.cpp:457-507: set_pattern, constraint sets, computePrecondMtr(redoSVD), analyze_pattern, initial energy
    computeConstraintSets(o);
    computeDampingMtr(o); // computePrecondMtr(..., updateDamping = dampingStiff): Optimizer.cpp:470, 3598-3612
    computePrecondMtr(o, true);
    o->lastEnergyVal = computeEnergyVal(o);
    return 0;
}

// head of solveSub_IP (Optimizer.cpp:1826-1828)
static void initSubProblem(orc_opt* o)
{
    o->projDBC = true;
    o->rhoDBC = 0.0;
    o->lastMove = o->completedStep;
}

// AnimScripter::computeCompletedStepSize (AnimScripter.cpp:2286-2300)
static double computeCompletedStepSize(orc_opt* o)
{
    const Mesh& m = *o->m;
    if (o->dist2Tol == 0.0) return o->completedStep = 1.0;
    double sqNorm = 0.0;
    for (size_t t = 0; t < o->tpIds.size(); ++t)
        for (int c = 0; c < 3; ++c) {
            const double d = m.V[o->tpIds[t] + m.nV * c] - o->tpPos[3 * t + c];
            sqNorm += d * d;
        }
    return o->completedStep = 1.0 - std::sqrt(sqNorm / (o->dist2Tol * 1.0e6));
}

void orc_opt_begin_timestep(orc_opt* o)
{
    Mesh& m = *o->m;
    Tic t(o->timers[11]);
    // stepAnimScript, AST_TWIST (AnimScripter.cpp:1674-1684, 2140-2275)
    std::fill(o->searchDir.begin(), o->searchDir.end(), 0.0);
    for (const auto& h : o->angVel) {
        int v = h.first;
        double th = h.second * o->dt, cs = std::cos(th), sn = std::sin(th);
        double y = m.V[v + m.nV] - o->rotCenter[1], z = m.V[v + 2 * m.nV] - o->rotCenter[2];
        double ny = cs * y - sn * z + o->rotCenter[1], nz = sn * y + cs * z + o->rotCenter[2];
        o->searchDir[3 * v + 1] = ny - m.V[v + m.nV];
        o->searchDir[3 * v + 2] = nz - m.V[v + 2 * m.nV];
    }
    o->stepStartTime = o->stepEndTime; // AnimScripter.cpp:1406-1407
    o->stepEndTime += o->dt;
    setDBCVertices(o);
    bool scripted = !o->angVel.empty();
    for (const auto& g : o->dbcGroups)
        if (o->stepStartTime >= g.t0 && o->stepStartTime < g.t1 && !g.ids.empty() && g.targets.empty()) {
            dbcGroupMotion(o, g);
            scripted = true;
        }
    for (const auto& g : o->dbcGroups) // the sequence SETS the move of its nodes, behind the velocities (AnimScripter.cpp:1465-1528)
        if (o->stepStartTime >= g.t0 && o->stepStartTime < g.t1 && !g.targets.empty()) {
            for (size_t i = 0; i < g.ids.size(); ++i)
                for (int c = 0; c < 3; ++c) o->searchDir[3 * g.ids[i] + c] = g.targets[3 * i + c] - m.V[g.ids[i] + m.nV * c];
            scripted = true;
        }
    // targetPos / dist2Tol (AnimScripter.cpp:2150-2157)
    o->tpIds.clear();
    o->tpPos.clear();
    double sq = 0;
    for (int v = 0; v < m.nV; ++v) {
        const double* p = &o->searchDir[3 * v];
        sq += p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        if (m.isDBC(v) || p[0] != 0 || p[1] != 0 || p[2] != 0) {
            o->tpIds.push_back(v);
            for (int c = 0; c < 3; ++c) o->tpPos.push_back(m.V[v + m.nV * c] + p[c]);
        }
    }
    o->tpLam.assign(o->tpPos.size(), 0.0);
    o->dist2Tol = sq * 1.0e-6;
    o->completedStep = 1.0;
    if (scripted) {
        double stepSize = filterStepSize(m, o->searchDir.data(), 1.0);
        if (o->selfCollision) { // :2158-2171: CCD of the scripted motion with slackness 0.5
            stepSize = fullCcd(o, 0.5, stepSize);
        }
        std::vector<double> V0 = m.V;
        stepForward(o, V0, stepSize);
        while (!m.inversionFree()) {
            stepSize /= 2.0;
            stepForward(o, V0, stepSize);
        }
        if (o->ipOn())
            while (anyIntersection(o)) {
                stepSize /= 2.0;
                stepForward(o, V0, stepSize);
            }
        if (stepSize < 1.0) o->dbcIncomplete++;
        o->completedStep = stepSize; // AnimScripter::getCompletedStepSize
    }
    // fullyImplicit_IP head (1518-1613): initX(warmStart); dHat; constraint sets; kappa; initial energy
    std::fill(o->searchDir.begin(), o->searchDir.end(), 0.0);
    if (o->warmStart >= 1 && o->warmStart <= 5) {
        // initX options 1-4 (Optimizer.cpp:936-1080): explicit Euler / xHat / symplectic Euler / uniformly accelerated motion as the
        // first iterate, then the same feasibility filters as a Newton step with "always full CCD" (:1117-1215)
        if (o->warmStart == 5) {
            // option 5 (:1082-1110), "Jacobi": -g_i / H_ii with the gradient of the projected and the matrix of the unprojected Dirichlet
            // rows, zero on every Dirichlet node; sets, kappa and dHat are whatever the previous time step left
            computeGradient(o, true);
            computePrecondMtr(o, false);
            for (int v = 0; v < m.nV; ++v)
                for (int c = 0; c < 3; ++c) o->searchDir[3 * v + c] = m.isDBC(v) ? 0.0 : -o->gradient[3 * v + c] / o->a[m.ia[3 * v + c]];
        }
        else {
        static const double CG[2][5] = { { 0, 0, 1, 1, 1 }, { 0, 0, 0.5, 0.5, 0.5 } }, CE[2][5] = { { 0, 0, 0, 1, 0.5 }, { 0, 0, 0, 2, 1 } };
        const double cg = CG[o->tit][o->warmStart], ce = CE[o->tit][o->warmStart];
        for (int v = 0; v < m.nV; ++v)
            for (int c = 0; c < 3; ++c)
                o->searchDir[3 * v + c] = m.isDBC(v) ? 0.0
                                                     : o->dt * o->velocity[3 * v + c] + cg * (o->dtSq * o->gravity[c])
                        + ce * (o->dxElastic.empty() ? 0.0 : o->dxElastic[3 * v + c]);
        }
        double stepSize = filterStepSize(m, o->searchDir.data(), 1.0);
        if (o->ipOn()) {
            for (const auto& h : o->planes) stepSize = hsStepBound(m, h, o->searchDir.data(), 0.9, stepSize);
            if (o->selfCollision) {
                stepSize = fullCcd(o, 0.8, stepSize);
            }
        }
        std::vector<double> V0 = m.V;
        stepForward(o, V0, stepSize);
        while (!m.inversionFree()) {
            stepSize /= 2.0;
            stepForward(o, V0, stepSize);
        }
        if (o->ipOn())
            while (anyIntersection(o)) {
                stepSize /= 2.0;
                stepForward(o, V0, stepSize);
            }
        o->warmStepSize = stepSize;
    }
    if (o->ipOn()) {
        o->dHat = o->dHatEps * o->dHatEps * o->lenScale2();
        computeConstraintSets(o);
        // tuning[0] when the script gives one, bounded from above; 0 -> suggestKappa (Optimizer.cpp:1540-1547)
        o->kappa = o->kappaConfig > 0.0 ? std::min(o->kappaConfig, 100 * kappaFloor(o)) : kappaFloor(o);
        initKappa(o); // ADAPTIVE_KAPPA (:1548-1550)
        o->closeID.clear(); // initSubProb_IP (:2316-2322)
        o->closeVal.clear();
        o->closeHS.clear();
        o->closeHSVal.clear();
        // friction: lagged sets reset, eps_v^2 h^2 (Optimizer.cpp:1525-1533, 286-304), then lagged at x^n (:1553-1600)
        o->lag = FrictionLag();
        for (auto& s : o->hsLagSet) s.clear();
        o->fricDHat0 = o->epsV * o->epsV * kCtorDtSq * o->lenScale2(); // set once in the reference's constructor, see kCtorDtSq
        o->fricDHatTarget = o->epsVTarget > 0.0 ? o->epsVTarget * o->epsVTarget * kCtorDtSq * o->lenScale2() : o->fricDHat0;
        o->fricDHat = o->solveFric() ? o->fricDHat0 : -1.0;
        o->fricIterI = 0;
        updateFrictionLag(o);
    }
    initSubProblem(o);
    if (o->patternDirty) computePrecondMtr(o, true);
    o->lastEnergyVal = computeEnergyVal(o);
    o->k = 0;
}

int orc_opt_newton_iter(orc_opt* o)
{
    Mesh& m = *o->m;
    {
        Tic t(o->timers[12]);
        computeGradient(o, o->projDBC);
    }
    // convergence test (1869-1879): uses the search direction of the previous pass
    double distToOpt_PN = 0;
    for (double v : o->searchDir) distToOpt_PN = std::max(distToOpt_PN, std::fabs(v));
    if (o->k && distToOpt_PN < o->targetGRes && o->completedStep > 1.0 - 1.0e-3) return 1;
    o->innerIterAmt++;
    // computeSearchDir (2324-2355)
    computePrecondMtr(o, o->projDBC);
    int ok;
    {
        Tic t(o->timers[3]);
        ok = orc_chol_factorize(o->chol, o->a.data());
    }
    std::vector<double> minusG(o->gradient.size());
    for (size_t i = 0; i < minusG.size(); ++i) minusG[i] = -o->gradient[i];
    {
        Tic t(o->timers[4]);
        if (!ok) {
            // precondition_diag (LinSysSolver.hpp:411-420)
            for (int r = 0; r < 3 * m.nV; ++r) o->searchDir[r] = minusG[r] / o->a[m.ia[r]];
        }
        else orc_chol_solve(o->chol, minusG.data(), o->searchDir.data());
    }
    // step-size pipeline (1884-2040)
    double alpha = 1.0;
    {
        Tic t(o->timers[13]);
        alpha = filterStepSize(m, o->searchDir.data(), alpha);
        for (const auto& h : o->planes) alpha = hsStepBound(m, h, o->searchDir.data(), 0.9, alpha); // slackness_a (:1886-1890)
        if (o->selfCollision) {
            const double slackness_m = 0.8;
            alpha = ccdStepBound(m, o->cs.csPTEE, o->searchDir.data(), slackness_m, alpha, &o->lastCCDArg); // partial CCD (:1923-1928)
            double pMax = 0; // CFL bound (:1947-1953)
            for (int v : m.SVI) {
                double s = 0;
                for (int c = 0; c < 3; ++c) s += o->searchDir[3 * v + c] * o->searchDir[3 * v + c];
                pMax = std::max(pMax, std::sqrt(s));
            }
            const double alpha_CFL = std::sqrt(o->dHat) / (pMax * 2.0);
            if ((!o->k && alpha > alpha_CFL) || alpha > 2.0 * alpha_CFL) {
                // full CCD (:1961-2021): the hash may cap alpha, then vertices against vertices / edges / triangles, edges against edges
                alpha = fullCcd(o, slackness_m, alpha);
                o->nFullCCD++;
                if (alpha < alpha_CFL) alpha = alpha_CFL;
            }
            else alpha = std::min(alpha, alpha_CFL);
        }
    }
    o->lastAlphaFeasible = alpha;
    lineSearch(o, alpha);
    o->lastStepSize = alpha;
    if (std::getenv("ORC_TRACE")) { // debugging aid: one line per Newton iteration, comparable with the reference's spdlog lines (IPCREF_LOG)
        double g2 = 0, pInf = 0;
        for (double v : o->gradient) g2 += v * v;
        for (double v : o->searchDir) pInf = std::max(pInf, std::fabs(v));
        std::fprintf(stderr, "[orc] step %d k %d kappa %.10g dHat %.10g fricDHat %.10g #c %zu #para %zu ||g||^2 %.10g |p|inf %.10g alphaFeasible %.10g alpha %.10g E %.12g\n",
            o->globalIterNum, o->k, o->kappa, o->dHat, o->fricDHat, o->cs.active.size(), o->cs.paraEE.size(), g2, pInf, o->lastAlphaFeasible, alpha, o->lastEnergyVal);
    }
    postLineSearch(o);
    // Dirichlet nodes that could not reach their scripted targets: augmented-Lagrangian pull (Optimizer.cpp:2168-2203)
    if (o->projDBC) {
        if (o->completedStep < 1.0 - 1.0e-3) {
            o->projDBC = false;
            o->rhoDBC = 1.0e6;
        }
    }
    else {
        const double completed = computeCompletedStepSize(o);
        if (completed > 1.0 - 1.0e-3) o->projDBC = true;
        else if (completed < o->lastMove && o->rhoDBC < 1.0e8) o->rhoDBC *= 2.0;
        else {
            double pInf = 0;
            for (double v : o->searchDir) pInf = std::max(pInf, std::fabs(v));
            if (pInf < o->CN_MBC) { // safeToPull
                if (completed < 0.99 && o->rhoDBC < 1.0e8) o->rhoDBC *= 2.0;
                else // updateLambda (AnimScripter.cpp:2339-2346)
                    for (size_t t = 0; t < o->tpIds.size(); ++t)
                        for (int c = 0; c < 3; ++c)
                            o->tpLam[3 * t + c] -= o->rhoDBC * std::sqrt(m.mass[o->tpIds[t]]) * (m.V[o->tpIds[t] + m.nV * c] - o->tpPos[3 * t + c]);
            }
        }
    }
    o->k++;
    return 0;
}

// Config `timeIntegration NM beta gamma` (Config.cpp:112-118); call before precompute
void orc_opt_set_time_integration(orc_opt* o, int type, double beta, double gamma)
{
    o->tit = type;
    o->betaNM = beta;
    o->gammaNM = gamma;
    o->acceleration.assign(3 * (size_t)o->m->nV, 0.0); // Optimizer.cpp:177
    computeXTilta(o);
}

// {completed step size of the scripted motion, rho_DBC, m_projectDBC, number of target positions}
void orc_opt_get_dbc_state(const orc_opt* o, double* out4)
{
    out4[0] = o->completedStep;
    out4[1] = o->rhoDBC;
    out4[2] = o->projDBC ? 1.0 : 0.0;
    out4[3] = (double)o->tpIds.size();
}
void orc_opt_get_kinematics(const orc_opt* o, double* vel, double* acc, double* dx)
{
    const size_t n3 = 3 * (size_t)o->m->nV;
    for (size_t i = 0; i < n3; ++i) {
        if (vel) vel[i] = o->velocity[i];
        if (acc) acc[i] = i < o->acceleration.size() ? o->acceleration[i] : 0.0;
        if (dx) dx[i] = i < o->dxElastic.size() ? o->dxElastic[i] : 0.0;
    }
}
// restart branch of the Optimizer constructor (Optimizer.cpp:179-248) after the status file has been parsed: positions are
// already in the mesh
void orc_opt_restart(orc_opt* o, int timestep, const double* vel, const double* acc, const double* dx)
{
    const size_t n3 = 3 * (size_t)o->m->nV;
    o->globalIterNum = timestep;
    o->velocity.assign(vel, vel + n3);
    o->acceleration.assign(acc, acc + n3);
    o->dxElastic.assign(dx, dx + n3);
    o->V_prev = o->m->V;
    computeXTilta(o);
}

void orc_opt_end_timestep(orc_opt* o)
{
    Mesh& m = *o->m;
    Tic t(o->timers[11]);
    if (o->acceleration.empty()) o->acceleration.assign(3 * (size_t)m.nV, 0.0);
    o->dxElastic.resize(3 * (size_t)m.nV);
    for (int v = 0; v < m.nV; ++v)
        for (int c = 0; c < 3; ++c) o->dxElastic[3 * v + c] = m.V[v + m.nV * c] - o->xTilta[v + m.nV * c]; // :574, :583
    if (o->tit == 1) { // TIT_NM (582-590)
        for (int v = 0; v < m.nV; ++v)
            for (int c = 0; c < 3; ++c) {
                double& vel = o->velocity[3 * v + c];
                double& acc = o->acceleration[3 * v + c];
                vel = vel + o->dt * (1 - o->gammaNM) * acc;
                acc = (m.V[v + m.nV * c] - o->xTilta[v + m.nV * c]) / (o->dtSq * o->betaNM);
                acc += o->gravity[c];
                vel += o->dt * o->gammaNM * acc;
            }
    }
    else // TIT_BE (570-580)
        for (int v = 0; v < m.nV; ++v)
            for (int c = 0; c < 3; ++c) {
                const double vNew = (m.V[v + m.nV * c] - o->V_prev[v + m.nV * c]) / o->dt;
                o->acceleration[3 * v + c] = (vNew - o->velocity[3 * v + c]) / o->dt; // :577
                o->velocity[3 * v + c] = vNew;
            }
    o->V_prev = m.V;
    computeXTilta(o);
    computeDampingMtr(o); // Optimizer.cpp:593-595
    o->globalIterNum++;
}

// After a sub-problem has converged: the tail of the fullyImplicit_IP loop body (Optimizer.cpp:1617-1790 with USE_DISCRETE_CMS,
// HOMOTOPY_VAR 1, dHat already at its target).  Returns 1 when another solveSub_IP pass has to run (friction lagging).
int orc_opt_next_subproblem(orc_opt* o)
{
    // tail of the fullyImplicit_IP loop body (Optimizer.cpp:1617-1790 with USE_DISCRETE_CMS, HOMOTOPY_VAR 1)
    if (!o->ipOn()) return 0;
    Mesh& m = *o->m;
    const double dHatTarget = o->dHatTargetEps > 0.0 ? o->dHatTargetEps * o->dHatTargetEps * o->lenScale2() : o->dHat;
    const bool fric = o->solveFric(), homotopy = o->dHat > dHatTarget;
    if (!fric && !homotopy) return 0; // every active distance is below dHat = dHatTarget: nothing left to update (:1706-1709, 1754-1757)
    o->fricIterI++;
    if (fric) updateFrictionLag(o);
    if (!o->nConstraints()) return 0; // "no collision in this time step"
    bool updateDHat = true;
    { // discrete complementarity slackness on the distances of the active sets (:1706-1713)
        double dMax = 0.0, dMin = 1.0e300;
        for (size_t i = 0; i < o->planes.size(); ++i)
            for (int v : o->hsSet[i]) {
                const double d = o->planes[i].dist(m, v);
                dMax = std::max(dMax, d * d);
                dMin = std::min(dMin, d * d);
            }
        if (o->selfCollision)
            for (const auto& c : o->cs.active) {
                const double d2 = evalMMCVID(m, c);
                dMax = std::max(dMax, d2);
                dMin = std::min(dMin, d2);
            }
        if (dMax < dHatTarget) updateDHat = false;
        else if (dMin < o->dTol) return 0; // "tiny distance fail-safe"
    }
    bool updateFricDHat = fric;
    if (fric && o->fricDHat <= o->fricDHatTarget) { // :1717 (the target equals the start value unless `tuning` gives a sixth entry)
        // tangent-space convergence test: one Newton direction with the refreshed lag (:1717-1731)
        computeGradient(o, true);
        computePrecondMtr(o, true);
        const int ok = orc_chol_factorize(o->chol, o->a.data());
        std::vector<double> minusG(o->gradient.size());
        for (size_t i = 0; i < minusG.size(); ++i) minusG[i] = -o->gradient[i];
        if (!ok)
            for (int r = 0; r < 3 * m.nV; ++r) o->searchDir[r] = minusG[r] / o->a[m.ia[r]];
        else orc_chol_solve(o->chol, minusG.data(), o->searchDir.data());
        double pMax = 0;
        for (double v : o->searchDir) pMax = std::max(pMax, std::fabs(v));
        if (pMax < o->targetGRes) updateFricDHat = false;
        if (o->fricIterAmt > 0 && o->fricIterI >= o->fricIterAmt) updateFricDHat = false;
    }
    if (!updateDHat && !updateFricDHat) return 0;
    if (updateDHat) { // :1763-1774
        o->dHat = std::max(0.5 * o->dHat, dHatTarget);
        computeConstraintSets(o);
        initKappa(o);
    }
    if (updateFricDHat && o->fricDHat > 0.0) o->fricDHat = std::max(0.5 * o->fricDHat, o->fricDHatTarget); // :1776-1781
    o->closeID.clear(); // initSubProb_IP
    o->closeVal.clear();
    o->closeHS.clear();
    o->closeHSVal.clear();
    o->k = 0;
    initSubProblem(o); // the next solveSub_IP starts with m_projectDBC = true, rho_DBC = 0 (Optimizer.cpp:1826-1828)
    return 1;
}

void orc_opt_set_dhat_target(orc_opt* o, double dHatTargetEps) { o->dHatTargetEps = dHatTargetEps; }
void orc_opt_set_kappa(orc_opt* o, double kappa) { o->kappaConfig = kappa > 0.0 ? kappa : 0.0; }
void orc_opt_set_damping(orc_opt* o, double dampingStiff) { o->dampingStiff = dampingStiff > 0.0 ? dampingStiff : 0.0; } // Config.cpp:141-147
void orc_opt_force_friction_loop(orc_opt* o, int on) { o->fricLoopForced = on != 0; }
void orc_opt_set_friction_scales(orc_opt* o, double scaleSelf, double scaleObstacle)
{
    o->fricScaleSelf = scaleSelf;
    o->fricScaleObst = scaleObstacle;
}
void orc_opt_set_friction(orc_opt* o, double selfFric, int fricIterAmt, double epsV)
{
    // `selfFric mu`, `fricIterAmt n`, `tuning ... eps_v` (Config.cpp:482-488, 550-551, 45)
    o->selfFric = selfFric;
    o->fricIterAmt = fricIterAmt;
    o->epsV = epsV;
}
void orc_opt_set_half_space_friction(orc_opt* o, int id, double mu) { o->hsFric[id] = mu; }
void orc_opt_get_friction(const orc_opt* o, double* scalars4, double* lambda)
{
    scalars4[0] = o->fricDHat;
    scalars4[1] = (double)o->lag.set.size();
    scalars4[2] = (double)o->fricIterI;
    size_t nh = 0;
    for (const auto& s : o->hsLagSet) nh += s.size();
    scalars4[3] = (double)nh;
    if (lambda)
        for (size_t i = 0; i < o->lag.lambda.size(); ++i) lambda[i] = o->lag.lambda[i];
}

int orc_opt_solve_timestep(orc_opt* o, int maxIter)
{
    orc_opt_begin_timestep(o);
    int it = 0;
    for (;;) {
        while (it < maxIter) {
            if (orc_opt_newton_iter(o)) break;
            ++it;
        }
        if (it >= maxIter || !orc_opt_next_subproblem(o)) break;
    }
    orc_opt_end_timestep(o);
    return it;
}

void orc_opt_get(const orc_opt* o, double* V, double* searchDir, double* gradient, double* sc)
{
    const Mesh& m = *o->m;
    if (V) std::memcpy(V, m.V.data(), 8 * 3 * m.nV);
    if (searchDir) std::memcpy(searchDir, o->searchDir.data(), 8 * 3 * m.nV);
    if (gradient) std::memcpy(gradient, o->gradient.data(), 8 * 3 * m.nV);
    if (sc) {
        sc[0] = o->lastEnergyVal;
        sc[1] = o->lastStepSize;
        sc[2] = o->targetGRes;
        sc[3] = o->innerIterAmt;
        sc[4] = o->globalIterNum;
        sc[5] = o->lastAlphaFeasible;
        sc[6] = o->kappa;
        sc[7] = o->dHat;
    }
}
// contact-side state: counts6 = {nActive, nParaEE, nCandidates, lastCCDArg, nFullCCD, nPatternChanges}
void orc_opt_get_contact(const orc_opt* o, int* counts6, int* active4, int* para4)
{
    counts6[0] = (int)o->cs.active.size();
    counts6[1] = (int)o->cs.paraEE.size();
    counts6[2] = (int)o->cs.csPTEE.size();
    counts6[3] = o->lastCCDArg;
    counts6[4] = o->nFullCCD;
    counts6[5] = o->nPatternChanges;
    if (active4)
        for (size_t i = 0; i < o->cs.active.size(); ++i)
            for (int k = 0; k < 4; ++k) active4[4 * i + k] = o->cs.active[i][k];
    if (para4)
        for (size_t i = 0; i < o->cs.paraEE.size(); ++i)
            for (int k = 0; k < 4; ++k) para4[4 * i + k] = o->cs.paraEE[i][k];
}
// HalfSpace::move (HalfSpace.cpp:389-416): returns the fraction of delta that is left
double orc_opt_half_space_move(orc_opt* o, int id, const double* delta3, double slackness) { return hsMove(*o->m, o->planes.at(id), delta3, slackness); }
void orc_opt_set_parameter_scaling(orc_opt* o, int useAbs, double dTolRel, double kappaMinMultiplier)
{
    // useAbsParameters / tuning[3] / kappaMinMultiplier (Config.cpp:553-558; Optimizer.cpp:102-109, 279-302, 1535-1537, 2228-2233, 2941-2945)
    o->absParameters = useAbs != 0;
    o->dTolRel = dTolRel;
    o->kappaMinMultiplier = kappaMinMultiplier;
    o->targetGRes = std::sqrt(o->relGL2Tol * (o->absParameters ? 1.0 : o->m->bboxDiag2 * o->dtSq));
    if (o->selfCollision || !o->planes.empty()) {
        o->dHat = o->dHatEps * o->dHatEps * o->lenScale2();
        o->dTol = dTolRel * dTolRel * o->lenScale2();
    }
}
void orc_opt_set_friction_target(orc_opt* o, double epsVTarget) { o->epsVTarget = epsVTarget > 0.0 ? epsVTarget : -1.0; }
void orc_opt_set_warm_start(orc_opt* o, int option)
{
    if (option < 0 || option > 5) return;
    o->warmStart = option;
}
double orc_opt_warm_step(const orc_opt* o) { return o->warmStepSize; }
void orc_opt_timers(const orc_opt* o, double* t16) { std::memcpy(t16, o->timers, sizeof(o->timers)); }
}
