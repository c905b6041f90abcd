// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_api.h).
//
// CPU restatement of the time-step driver around the Newton loop, following
// Optimizer.cpp: solve() 510-602, fullyImplicit_IP() 1518-1819, solveSub_IP() 1822-2213,
// computeSearchDir() 2324-2355, lineSearch() 2662-2916, stepForward() 2919-2938,
// computeEnergyVal() 3199-3405, computeGradient() 3409-3545, computePrecondMtr() 3549-3720,
// computeXTilta() 1236-1257, and the `twist` script of AnimScripter.cpp:555-572,1674-1684.
// Contact terms are added by orc_contact.cpp when a surface is registered.
#include "orc_api.h"
#include "orc_core.h"
#include <chrono>
#include <cstdio>
#include <map>
#include <omp.h>

using namespace orc;

struct orc_opt {
    Mesh* m;
    double dt, dtSq, gravity[3] = { 0, 0, 0 };
    int nthreads;
    double relGL2Tol = 1.0e-8, targetGRes = 0;
    std::vector<double> velocity, xTilta, V_prev, searchDir, gradient, a;
    std::map<int, double> angVel; // twist handles
    double rotCenter[3];
    orc_chol* chol = nullptr;
    int innerIterAmt = 0, globalIterNum = 0, k = 0;
    double lastEnergyVal = 0, lastStepSize = 0, lastAlphaFeasible = 0;
    double timers[16] = { 0 };
    bool patternDirty = true;
};

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
            else o->xTilta[v + m.nV * c] = o->V_prev[v + m.nV * c] + (o->velocity[3 * v + c] * o->dt + o->dtSq * o->gravity[c]);
        }
}

// Optimizer.cpp:3199-3239 (elasticity + inertia)
double computeEnergyVal(orc_opt* o)
{
    Mesh& m = *o->m;
    double E = elasticEnergy(m, o->dtSq, nullptr);
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
    return E + sum;
}

// Optimizer.cpp:3409-3450
void computeGradient(orc_opt* o, bool projectDBC)
{
    Mesh& m = *o->m;
    elasticGradient(m, o->dtSq, projectDBC, o->gradient.data());
#pragma omp parallel for schedule(static)
    for (int v = 0; v < m.nV; ++v)
        if (!m.isProjectDBC(v, projectDBC))
            for (int c = 0; c < 3; ++c)
                o->gradient[3 * v + c] += m.mass[v] * (m.V[v + m.nV * c] - o->xTilta[v + m.nV * c]);
}

void ensurePattern(orc_opt* o)
{
    if (!o->patternDirty) return;
    Mesh& m = *o->m;
    {
        Tic t(o->timers[1]);
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

void stepForward(orc_opt* o, const std::vector<double>& V0, double alpha)
{
    Mesh& m = *o->m;
    for (int v = 0; v < m.nV; ++v)
        for (int c = 0; c < 3; ++c) m.V[v + m.nV * c] = V0[v + m.nV * c] + alpha * o->searchDir[3 * v + c];
}

// Optimizer.cpp:2662-2916 with armijoParam = 0, lowerBound = 0 (the IP call site, :2059)
void lineSearch(orc_opt* o, double& stepSize)
{
    Mesh& m = *o->m;
    {
        Tic t(o->timers[9]);
        o->lastEnergyVal = computeEnergyVal(o);
    }
    std::vector<double> V0 = m.V;
    {
        Tic t(o->timers[5]);
        stepForward(o, V0, stepSize);
        while (!m.checkInversion()) {
            stepSize /= 2.0;
            stepForward(o, V0, stepSize);
        }
    }
    double testingE;
    {
        Tic t(o->timers[9]);
        testingE = computeEnergyVal(o);
    }
    while (testingE > o->lastEnergyVal && stepSize > 0.0) {
        stepSize /= 2.0;
        if (stepSize == 0.0) break;
        {
            Tic t(o->timers[5]);
            stepForward(o, V0, stepSize);
        }
        Tic t(o->timers[9]);
        testingE = computeEnergyVal(o);
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
    o->targetGRes = std::sqrt(o->relGL2Tol * o->m->bboxDiag2 * o->dtSq); // Optimizer.cpp:2941-2945
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

void orc_opt_precompute(orc_opt* o)
{
    // Optimizer.cpp:457-507: set_pattern, computePrecondMtr(redoSVD), analyze_pattern, initial energy
    ensurePattern(o);
    assembleHessian(*o->m, o->dtSq, true, o->a.data());
    o->lastEnergyVal = computeEnergyVal(o);
}

void orc_opt_begin_timestep(orc_opt* o)
{
    Mesh& m = *o->m;
    Tic t(o->timers[11]);
    // stepAnimScript, AST_TWIST (AnimScripter.cpp:1674-1684): rotate handle vertices about the x axis
    // through the rest bbox centre by angVel*dt; move them fully (no contact => step size 1 unless inversion).
    std::fill(o->searchDir.begin(), o->searchDir.end(), 0.0);
    for (const auto& h : o->angVel) {
        int v = h.first;
        double th = h.second * o->dt, cs = std::cos(th), sn = std::sin(th);
        double y = m.V[v + m.nV] - o->rotCenter[1], z = m.V[v + 2 * m.nV] - o->rotCenter[2];
        double ny = cs * y - sn * z + o->rotCenter[1], nz = sn * y + cs * z + o->rotCenter[2];
        o->searchDir[3 * v + 1] = ny - m.V[v + m.nV];
        o->searchDir[3 * v + 2] = nz - m.V[v + 2 * m.nV];
    }
    if (!o->angVel.empty()) {
        double stepSize = filterStepSize(m, o->searchDir.data(), 1.0);
        std::vector<double> V0 = m.V;
        stepForward(o, V0, stepSize);
        while (!m.checkInversion()) {
            stepSize /= 2.0;
            stepForward(o, V0, stepSize);
        }
        if (stepSize < 1.0) std::fprintf(stderr, "[oracle] scripted DBC motion only completed %g (penalty path not restated)\n", stepSize);
    }
    // fullyImplicit_IP head (1518-1613): initX(0) -> searchDir = 0; initial energy with redoSVD
    std::fill(o->searchDir.begin(), o->searchDir.end(), 0.0);
    ensurePattern(o);
    o->lastEnergyVal = computeEnergyVal(o);
    o->k = 0;
}

int orc_opt_newton_iter(orc_opt* o)
{
    Mesh& m = *o->m;
    {
        Tic t(o->timers[12]);
        computeGradient(o, true);
    }
    // convergence test (1869-1879): uses the search direction of the previous pass
    double distToOpt_PN = 0;
    for (double v : o->searchDir) distToOpt_PN = std::max(distToOpt_PN, std::fabs(v));
    if (o->k && distToOpt_PN < o->targetGRes) return 1;
    o->innerIterAmt++;
    // computeSearchDir (2324-2355)
    {
        Tic t(o->timers[0]);
        assembleHessian(m, o->dtSq, true, o->a.data());
    }
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
    double alpha = 1.0;
    {
        Tic t(o->timers[13]);
        alpha = filterStepSize(m, o->searchDir.data(), alpha);
    }
    o->lastAlphaFeasible = alpha;
    lineSearch(o, alpha);
    o->lastStepSize = alpha;
    o->k++;
    return 0;
}

void orc_opt_end_timestep(orc_opt* o)
{
    Mesh& m = *o->m;
    Tic t(o->timers[11]);
    // TIT_BE (570-580)
    for (int v = 0; v < m.nV; ++v)
        for (int c = 0; c < 3; ++c) o->velocity[3 * v + c] = (m.V[v + m.nV * c] - o->V_prev[v + m.nV * c]) / o->dt;
    o->V_prev = m.V;
    computeXTilta(o);
    o->globalIterNum++;
}

int orc_opt_solve_timestep(orc_opt* o, int maxIter)
{
    orc_opt_begin_timestep(o);
    int it = 0;
    while (it < maxIter) {
        if (orc_opt_newton_iter(o)) break;
        ++it;
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
        sc[6] = sc[7] = 0;
    }
}
void orc_opt_timers(const orc_opt* o, double* t16) { std::memcpy(t16, o->timers, sizeof(o->timers)); }
}
