// HipOptimizer: the Newton time stepper of Optimizer<3> with all state in HBM.
//   solve()            Optimizer.cpp:510-602        beginTimestep / newtonIter* / endTimestep
//   fullyImplicit_IP() Optimizer.cpp:1518-1819      (contact-free branch: one solveSub_IP per time step)
//   solveSub_IP()      Optimizer.cpp:1822-2213      newtonIter()
//   computeSearchDir() Optimizer.cpp:2324-2355
//   lineSearch()       Optimizer.cpp:2662-2916      (armijoParam = 0 at the IP call site :2059)
//   computeEnergyVal / computeGradient / computePrecondMtr   Optimizer.cpp:3199-3239, 3409-3450, 3549-3668
// Per Newton iteration only a handful of scalars cross PCIe (energies, step bound, |p|_inf, not-PD flag).
#include "hip_ipc.h"
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <algorithm>
#include <chrono>
#include <thread>
#include <iterator>
#include <cmath>
#include <cstring>

#ifndef BEGIN_BATCHED
#define BEGIN_BATCHED 1 // 0: the time-step set-up of the contact-free twist question by question, as in rounds 1-5 (A/B: profiles/r06_begin_timestep_batch_ab.txt)
#endif

namespace ipcgpu {

// The reference's Optimizer constructor calls setTime(10.0, 0.025) (Optimizer.cpp:116) and derives eps_v^2 h^2 (fricDHat0 / fricDHatTarget, :290-303)
// and CN_MBC (:268) from THAT step size; main.cpp:1398 sets the scene's dt afterwards and setTime (:421-429) recomputes neither.  So in the reference
// these three carry h = 0.025 whatever `time` the scene file gives -- found on otherExamples/typical/sphere1K_DCORotCylinders.txt (dt 0.04, selfFric 0.5:
// 47 Newton iterations in the step after the first contact, 20 with eps_v scaled by the scene's own dt; same counts once this is followed).
// The value is an optimizer parameter since round 4 (HipOptimizer::ctorDt, ipcgpu_opt_set_constructor_dt): 0.025 = the reference's behaviour is the default,
// the scene's own dt gives the paper's eps_v h.

namespace {
struct Tic {
    double& acc;
    hipStream_t s;
    std::chrono::high_resolution_clock::time_point t0;
    bool nosync = false; // the caller has synchronised what it timed by an event: the stream may carry work enqueued AHEAD (speculativeAssembly) that must not be waited for
    Tic(double& a, hipStream_t st) : acc(a), s(st), t0(std::chrono::high_resolution_clock::now()) {}
    ~Tic()
    {
        if (!nosync) (void)hipStreamSynchronize(s);
        acc += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    }
};
} // namespace

HipOptimizer::HipOptimizer(HipMesh& m, HipLinSysSolver& l, hipStream_t s) : mesh(m), lin(l), stream(s) {}

ElemView HipOptimizer::view() const
{
    ElemView v;
    v.nV = mesh.nV;
    v.nT = mesh.nT;
    v.tetBegin = tetBegin;
    v.tetEnd = tetEnd;
    v.energyType = mesh.energyType;
    v.x = mesh.d_x.p;
    v.xTilde = mesh.d_xTilde.p;
    v.mass = mesh.d_mass.p;
    v.dbc = mesh.d_dbc.p;
    v.tet = mesh.d_tet.p;
    v.A = mesh.d_A.p;
    v.vol = mesh.d_vol.p;
    v.mu = mesh.d_mu.p;
    v.lam = mesh.d_lam.p;
    v.rowBase = lin.d_rowBase.p;
    v.rowLen = lin.d_rowLen.p;
    v.edgeP0 = lin.d_edgeP0.p;
    return v;
}

void HipOptimizer::init(double dt_, bool withGravity)
{
    dt = dt_;
    dtSq = dt * dt;
    gravity[0] = gravity[2] = 0.0;
    gravity[1] = withGravity ? -9.80665 : 0.0; // Optimizer.cpp:112-115
    const size_t n3 = 3 * (size_t)mesh.nV;
    d_vel.alloc(n3);
    d_vel.zero(stream);
    d_acc.alloc(n3); // Optimizer.cpp:176-177
    d_acc.zero(stream);
    d_dxElastic.alloc(n3);
    d_dxElastic.zero(stream);
    d_xPrev.alloc(n3);
    d_searchDir.alloc(n3);
    d_searchDir.zero(stream);
    d_gradient.alloc(n3);
    d_gradient.zero(stream);
    d_minusG.alloc(n3);
    d_x0.alloc(n3);
    d_partial.alloc((size_t)std::max(mesh.nT, mesh.nV) / 256 + 2);
    d_scalar.alloc(8);
    d_flag.alloc(2); // [0] the inversion flag of the stepper, [1] a second one for batches that test two states (beginTimestep)
    h_scalar.alloc(8);
    h_flag.alloc(2);
    HIP_CHECK(hipMemcpyAsync(d_xPrev.p, mesh.d_x.p, n3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(mesh.d_xTilde.p, mesh.d_x.p, n3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    for (int c = 0; c < 3; ++c) rotCenter[c] = 0.5 * (mesh.bboxLo[c] + mesh.bboxHi[c]);
    // element shard of this rank (contiguous blocks of the caller's tet order)
    tetBegin = (int)((long long)mesh.nT * rank / worldSize);
    tetEnd = (int)((long long)mesh.nT * (rank + 1) / worldSize);
    setRelGL2Tol(1.0e-2); // main.cpp:159 -> Optimizer.hpp:148 default
    innerIterAmt = globalIterNum = k = 0;
    std::memset(timers, 0, sizeof(timers));
    initialised = true;
    HIP_CHECK(hipStreamSynchronize(stream));
    computeXTilta(); // Optimizer.cpp:247-248: the first time step already sees gravity
}

void HipOptimizer::setRelGL2Tol(double relTol)
{
    relGL2Tol = relTol * relTol;
    targetGRes = std::sqrt(relGL2Tol * (absParameters ? 1.0 : mesh.bboxDiag2 * dtSq)); // Optimizer.cpp:2941-2945
    CN_MBC = std::sqrt(1.0e-4 * mesh.bboxDiag2 * (ctorDt * ctorDt)); // Optimizer.cpp:268 -- evaluated in the constructor, see (ctorDt * ctorDt)
}

void HipOptimizer::setParameterScaling(bool absolute, double dTolRel_, double kappaMinMultiplier_)
{
    // `useAbsParameters`, tuning[3], `kappaMinMultiplier` of the scene file (Config.cpp:553-558, Optimizer.cpp:102-109, 279-302, 1535-1537,
    // 2228-2233, 2941-2945): dHat, its target, dTol, eps_v and the Newton tolerance are absolute lengths instead of fractions of the
    // bounding-box diagonal; the distances of suggestKappa and CN_MBC stay relative there too
    if (!(dTolRel_ > 0.0) || !(kappaMinMultiplier_ > 0.0)) throw ArgError("set_parameter_scaling: dTolRel and kappaMinMultiplier must be positive");
    absParameters = absolute;
    dTolRel = dTolRel_;
    kappaMinMultiplier = kappaMinMultiplier_;
    targetGRes = std::sqrt(relGL2Tol * (absParameters ? 1.0 : mesh.bboxDiag2 * dtSq));
    if (ipOn()) {
        dHat = dHatEps * dHatEps * lenScale2();
        dTol = dTolRel * dTolRel * lenScale2();
    }
}

void HipOptimizer::setTwist(int nL, const int* left, int nR, const int* right, double angVel)
{
    // AnimScripter.cpp:555-572: handle set bI rotates with (-1)^bI * -0.4 pi about x through the rest bbox centre
    std::vector<int> ids;
    std::vector<double> ang;
    for (int i = 0; i < nL; ++i) {
        mesh.dbcType[left[i]] = 2;
        ids.push_back(left[i]);
        ang.push_back(-angVel * dt);
    }
    for (int i = 0; i < nR; ++i) {
        mesh.dbcType[right[i]] = 2;
        ids.push_back(right[i]);
        ang.push_back(angVel * dt);
    }
    nHandles = (int)ids.size();
    if (nHandles) {
        d_handleIds.upload(ids, stream);
        d_handleAng.upload(ang, stream);
    }
    mesh.uploadDBC(stream);
    if (initialised) computeXTilta(); // the handles are Dirichlet nodes now: xTilta = V_prev there (Optimizer.cpp:1244-1246)
}

void HipOptimizer::addNeumannBC(int n, const int* ids, const double* accel3, double t0, double t1)
{
    std::unique_ptr<NbcGroup> g(new NbcGroup);
    g->n = n;
    g->d_ids.upload(ids, (size_t)n, stream);
    for (int c = 0; c < 3; ++c) g->a[c] = accel3[c];
    g->t0 = t0;
    g->t1 = t1;
    HIP_CHECK(hipStreamSynchronize(stream));
    nbcGroups.push_back(std::move(g));
}

void HipOptimizer::neumannGradientAdd(double* grad_dev)
{
    for (const auto& g : nbcGroups) {
        if (stepStartTime < g->t0 || stepStartTime >= g->t1) continue;
        const double c[3] = { dtSq * g->a[0], dtSq * g->a[1], dtSq * g->a[2] };
        launch_nbc_gradient(g->n, g->d_ids.p, mesh.d_dbc.p, mesh.d_mass.p, c, grad_dev, stream);
    }
}

double HipOptimizer::neumannEnergy()
{
    double E = 0.0;
    for (const auto& g : nbcGroups) {
        if (stepStartTime < g->t0 || stepStartTime >= g->t1) continue;
        const double c[3] = { dtSq * g->a[0], dtSq * g->a[1], dtSq * g->a[2] };
        launch_nbc_energy(g->n, g->d_ids.p, mesh.d_dbc.p, mesh.d_mass.p, mesh.d_x.p, c, d_scalar.p + 5, stream);
        E -= readScalar(d_scalar.p + 5);
    }
    return E;
}

void HipOptimizer::addDirichletBC(int n, const int* ids, const double* lin3, const double* angRad3, double t0, double t1)
{
    std::unique_ptr<DbcGroup> g(new DbcGroup);
    g->ids.assign(ids, ids + n);
    for (int c = 0; c < 3; ++c) {
        g->lin[c] = lin3[c];
        g->ang[c] = angRad3[c];
    }
    g->t0 = t0;
    g->t1 = t1;
    g->d_ids.upload(g->ids, stream);
    g->d_pos.alloc(3 * (size_t)n);
    if (baseDbcType.empty()) baseDbcType = mesh.dbcType;
    dbcGroups.push_back(std::move(g));
    setDBCVertices(); // initAnimScript (AnimScripter.cpp:122-124)
    if (initialised) computeXTilta();
}

void HipOptimizer::setDBCVertices()
{
    if (dbcGroups.empty()) return;
    if (baseDbcType.empty()) baseDbcType = mesh.dbcType;
    std::vector<int> t = baseDbcType;
    for (const auto& g : dbcGroups) {
        if (stepStartTime < g->t0 || stepStartTime >= g->t1) continue;
        const int type = g->isZero() ? 1 : 2;
        for (int v : g->ids) t[v] = std::max(t[v], type); // NONZERO overrides ZERO
    }
    if (t != mesh.dbcType) {
        mesh.dbcType = t;
        mesh.uploadDBC(stream);
    }
}

bool HipOptimizer::dbcGroupMotion()
{
    bool any = false;
    std::vector<double> pos;
    for (const auto& g : dbcGroups) {
        if (stepStartTime < g->t0 || stepStartTime >= g->t1 || g->ids.empty() || g->hasTargets) continue;
        const int n = (int)g->ids.size();
        // centre of the group's current bounding box (AnimScripter.cpp:1450-1455): a few KB to the host, once per time step
        launch_gather3(n, g->d_ids.p, mesh.d_x.p, g->d_pos.p, stream);
        pos.resize(3 * (size_t)n);
        g->d_pos.download(pos.data(), pos.size(), stream);
        HIP_CHECK(hipStreamSynchronize(stream));
        DbcMotion m;
        double lo[3] = { pos[0], pos[1], pos[2] }, hi[3] = { pos[0], pos[1], pos[2] };
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) {
                lo[c] = std::min(lo[c], pos[3 * (size_t)i + c]);
                hi[c] = std::max(hi[c], pos[3 * (size_t)i + c]);
            }
        for (int c = 0; c < 3; ++c) {
            m.c[c] = g->hasCenter ? g->center[c] : (lo[c] + hi[c]) / 2;
            m.linDt[c] = g->lin[c] * dt;
        }
        const double ax = g->ang[0] * dt, ay = g->ang[1] * dt, az = g->ang[2] * dt;
        const double cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
        const double Rx[9] = { 1, 0, 0, 0, cx, -sx, 0, sx, cx }, Ry[9] = { cy, 0, sy, 0, 1, 0, -sy, 0, cy }, Rz[9] = { cz, -sz, 0, sz, cz, 0, 0, 0, 1 };
        double T[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T[3 * i + j] = Rx[3 * i] * Ry[j] + Rx[3 * i + 1] * Ry[3 + j] + Rx[3 * i + 2] * Ry[6 + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) m.R[3 * i + j] = T[3 * i] * Rz[j] + T[3 * i + 1] * Rz[3 + j] + T[3 * i + 2] * Rz[6 + j];
        launch_dbc_motion(n, g->d_ids.p, m, mesh.d_x.p, d_searchDir.p, stream);
        any = true;
    }
    for (const auto& g : dbcGroups) { // behind the velocities, as in AnimScripter.cpp:1465: the sequence SETS the move of its nodes
        if (!g->hasTargets || stepStartTime < g->t0 || stepStartTime >= g->t1 || g->ids.empty()) continue;
        launch_dbc_targets((int)g->ids.size(), g->d_ids.p, g->d_targets.p, mesh.d_x.p, d_searchDir.p, stream);
        any = true;
    }
    return any;
}

void HipOptimizer::reduceSum(double* dev, long long n)
{
    if (worldSize > 1) hookReduce(dev, n, 0);
}
void HipOptimizer::reduceMin(double* dev, long long n)
{
    if (worldSize > 1) hookReduce(dev, n, 1);
}
void HipOptimizer::hookReduce(double* dev, long long n, int op)
{
    commBytes += 8 * n;
    commCalls++;
    if (allreduceStream) { // RCCL from C, ordered on our stream
        if (allreduceStream(allreduceStreamUser, dev, n, op, (void*)stream) != 0) throw HipError("all-reduce hook failed");
        return;
    }
    if (!allreduce) throw StateError("sharded context without an all-reduce hook");
    HIP_CHECK(hipStreamSynchronize(stream));
    if (allreduce(allreduceUser, dev, n, op) != 0) throw HipError("all-reduce hook failed");
}
double HipOptimizer::readScalar(const double* dev)
{
    launch_publish(dev, h_scalar.dev, 2, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    return h_scalar.p[0];
}

double HipOptimizer::computeEnergyVal()
{
    launch_energy(view(), elasticCoef(), true, rank == 0, d_partial.p, (int)d_partial.n, d_scalar.p, stream);
    reduceSum(d_scalar.p, 1);
    // the barrier energy of the self-contact sets is enqueued behind the elastic one and read back with it: one synchronisation instead of two (round 6; the
    // reduction of the elastic partial sums is on the stream before the contact kernel reuses d_partial)
    const bool contactQueued = selfCollision && contact->energyEnqueue(mesh.d_x.p, dHat, kappa, d_partial, d_scalar.p + 4, d_scalar.p, h_scalar.dev, 10);
    if (!(contactQueued && contact->publishedByEnergy())) launch_publish(d_scalar.p, h_scalar.dev, 10, stream); // d_scalar[0 .. 4]
    HIP_CHECK(hipStreamSynchronize(stream));
    double E = h_scalar.p[0];
    const double Econtact = contactQueued ? h_scalar.p[4] : 0.0;
    if (!nbcGroups.empty()) E += neumannEnergy();
    // barrier terms over the current constraint sets (Optimizer.cpp:3252-3353); replicated on every rank
    for (auto& h : planes) E += h->energy(mesh.d_x.p, dHat, kappa);
    if (selfCollision) E += Econtact;
    if (fricDHat > 0.0) { // lagged friction (Optimizer.cpp:3357-3377)
        for (auto& h : planes)
            if (h->friction > 0.0) E += h->frictionEnergy(mesh.d_x.p, d_xPrev.p, fricDHat);
        if (selfCollision && selfFric > 0.0) E += contact->frictionEnergy(mesh.d_x.p, d_xPrev.p, fricDHat, selfFric, d_partial, d_scalar.p + 4);
    }
    if (dampingStiff > 0.0) E += dampingEnergy();
    if (rhoDBC && !tpIds.empty()) { // augmentMDBCEnergy (Optimizer.cpp:3402-3404)
        launch_mdbc_reduce(mdbc(), mesh.d_x.p, rhoDBC, 0, d_scalar.p + 5, stream);
        E += readScalar(d_scalar.p + 5);
    }
    return E;
}

// ---- lagged damping ----------------------------------------------------------------------------------------------
void HipOptimizer::setDamping(double stiff)
{
    dampingStiff = stiff > 0.0 ? stiff : 0.0; // Config.cpp:141-147
}
void HipOptimizer::computeDampingMtr()
{
    // computeDampingMtr (Optimizer.cpp:3723-3735) at the end of a time step (:593-595) and inside the first computePrecondMtr (:470,
    // 3598-3612).  The positions are kept: when the pattern grows the values are rebuilt from them on the new pattern.
    if (!(dampingStiff > 0.0)) return;
    d_xDamp.ensure(3 * (size_t)mesh.nV);
    HIP_CHECK(hipMemcpyAsync(d_xDamp.p, mesh.d_x.p, sizeof(double) * 3 * (size_t)mesh.nV, hipMemcpyDeviceToDevice, stream));
    assembleDampingMtr();
}
void HipOptimizer::assembleDampingMtr()
{
    const size_t nnz = lin.ja.size();
    d_damp.ensure(nnz);
    if (d_zeroMass.n < (size_t)mesh.nV) {
        d_zeroMass.alloc(mesh.nV);
        d_zeroMass.zero(stream);
    }
    ensurePatchPlan();
    int pb, pe;
    patchShard(pb, pe);
    ElemView v = view();
    v.x = d_xDamp.p;
    v.xTilde = d_xDamp.p;
    v.mass = d_zeroMass.p; // the patch pass owns the diagonal: no mass here, and the identity of the projected rows is cleared below
    if (worldSize > 1) HIP_CHECK(hipMemsetAsync(d_damp.p, 0, sizeof(double) * nnz, stream));
    launch_assemble_patches(v, patch, pb, pe, dampingStiff / dt, 1, nullptr, d_damp.p, stream);
    if (worldSize > 1) reduceSum(d_damp.p, (long long)nnz);
    launch_damp_clear_diag(mesh.nV, mesh.d_dbc.p, lin.d_ia.p, d_damp.p, stream);
    dampPatternVersion = lin.patternVersion;
}
double HipOptimizer::dampingEnergy()
{
    // Optimizer.cpp:3381-3400: 1/2 dx^T D dx with the displacement of the step, zero on every Dirichlet node
    if (dampPatternVersion != lin.patternVersion) assembleDampingMtr();
    const int n3 = 3 * mesh.nV;
    d_dampDx.ensure(n3);
    d_dampAdx.ensure(n3);
    launch_damp_dx(mesh.nV, mesh.d_dbc.p, 0, 1, mesh.d_x.p, d_xPrev.p, d_dampDx.p, stream);
    launch_csr_symv(n3, lin.d_ia.p, lin.d_ja.p, d_damp.p, d_dampDx.p, d_dampAdx.p, stream);
    launch_dot_scaled(n3, d_dampDx.p, d_dampAdx.p, 0.5, d_scalar.p + 6, stream);
    return readScalar(d_scalar.p + 6);
}
void HipOptimizer::dampingGradientAdd(bool projectDBC, double* grad_dev)
{
    // Optimizer.cpp:3519-3540: gradient += D dx, the displacement cleared on the projected Dirichlet nodes
    if (dampPatternVersion != lin.patternVersion) assembleDampingMtr();
    const int n3 = 3 * mesh.nV;
    d_dampDx.ensure(n3);
    d_dampAdx.ensure(n3);
    launch_damp_dx(mesh.nV, mesh.d_dbc.p, 1, projectDBC ? 1 : 0, mesh.d_x.p, d_xPrev.p, d_dampDx.p, stream);
    launch_csr_symv(n3, lin.d_ia.p, lin.d_ja.p, d_damp.p, d_dampDx.p, d_dampAdx.p, stream);
    launch_axpy(n3, 1.0, d_dampAdx.p, grad_dev, stream);
}

// ---- lagged friction ---------------------------------------------------------------------------------------------
bool HipOptimizer::solveFric() const
{
    if (fricLoopForced) return true;
    if (selfCollision && selfFric > 0.0) return true;
    for (const auto& h : planes)
        if (h->friction > 0.0) return true;
    return false;
}

void HipOptimizer::updateFrictionLag()
{
    // multipliers, closest points and tangent bases of the current constraint sets (Optimizer.cpp:1553-1600 / 1620-1675)
    if (!solveFric()) return;
    for (auto& h : planes)
        if (h->friction > 0.0) h->lagUpdate(mesh.d_x.p, dHat, kappa);
    if (selfCollision && selfFric > 0.0) contact->frictionLagUpdate(mesh.d_x.p, dHat, kappa);
}

bool HipOptimizer::nextSubproblem()
{
    // tail of the fullyImplicit_IP loop body (Optimizer.cpp:1617-1790 with USE_DISCRETE_CMS, HOMOTOPY_VAR 1)
    specAsmValid = false; // friction lag / dHat / kappa change under an assembly enqueued ahead
    if (!ipOn()) return false;
    const double dHatTarget = dHatTargetEps > 0.0 ? dHatTargetEps * dHatTargetEps * lenScale2() : dHat;
    const bool fric = solveFric(), homotopy = dHat > dHatTarget;
    if (!fric && !homotopy) return false; // every active distance is below dHat = dHatTarget: nothing left to update (:1706-1709, 1754-1757)
    fricIterI++;
    if (fric) updateFrictionLag();
    if (!nConstraints()) return false; // "no collision in this time step"
    bool updateDHat = true;
    { // discrete complementarity slackness on the distances of the active sets (:1706-1713); once per sub-problem, on the host
        double dMax = 0.0, dMin = 1.0e300;
        std::vector<double> d;
        for (auto& h : planes) {
            h->evalDist2(h->set, mesh.d_x.p, d);
            for (size_t i = 0; i < h->set.size(); ++i) {
                dMax = std::max(dMax, d[i]);
                dMin = std::min(dMin, d[i]);
            }
        }
        if (selfCollision) {
            std::vector<std::array<int, 4>> ids;
            contact->closeStencils(mesh.d_x.p, 1.0e300, ids, d);
            for (size_t i = 0; i < ids.size(); ++i) {
                dMax = std::max(dMax, d[i]);
                dMin = std::min(dMin, d[i]);
            }
        }
        if (dMax < dHatTarget) updateDHat = false;
        else if (dMin < dTol) return false; // "tiny distance fail-safe"
    }
    bool updateFricDHat = fric;
    if (fric && fricDHat <= fricDHatTarget) { // :1717 (the target equals the start value unless `tuning` gives a sixth entry)
        // tangent-space convergence test: one Newton direction with the refreshed lag (:1717-1731)
        computePrecondMtr(true, true);
        computeSearchDir(true);
        launch_fill(d_scalar.p + 3, 1, 0.0, stream);
        launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
        if (readScalar(d_scalar.p + 3) < targetGRes) updateFricDHat = false;
        if (fricIterAmt > 0 && fricIterI >= fricIterAmt) updateFricDHat = false;
    }
    if (!updateDHat && !updateFricDHat) return false;
    if (updateDHat) { // :1763-1774
        dHat = std::max(0.5 * dHat, dHatTarget);
        computeConstraintSets();
        initKappa();
    }
    if (updateFricDHat && fricDHat > 0.0) fricDHat = std::max(0.5 * fricDHat, fricDHatTarget); // :1776-1781
    initSubProblem(); // the next solveSub_IP starts with m_projectDBC = true, rho_DBC = 0 (Optimizer.cpp:1826-1828)
    closeID.clear(); // initSubProb_IP
    closeVal.clear();
    closeHS.clear();
    closeHSVal.clear();
    k = 0;
    return true;
}

size_t HipOptimizer::nConstraints() const
{
    size_t n = selfCollision ? (size_t)contact->nActive() : 0;
    for (const auto& h : planes) n += h->set.size();
    return n;
}

int HipOptimizer::addHalfSpace(HipContact* c, const double* origin3, const double* normal3, double eps)
{
    // `ground` / `halfSpace` script keywords (Config.cpp:306-345) -> animConfig.collisionObjects; friction is a SURVEY 8f row
    if (!c || !c->surfaceSet) throw StateError("opt_add_half_space before set_surface");
    // one dHat for the whole interior-point problem, as in the reference (a single `dHat` script keyword, Config.cpp:41-45):
    // a second object registered with another value would silently change the first one's activation distance
    if ((selfCollision || !planes.empty()) && eps != dHatEps) throw ArgError("all collision objects of a context share one dHat (Config.cpp:41-45): got a different dHatEps");
    contact = c;
    planes.emplace_back(new HipHalfSpace(stream, origin3, normal3));
    dHatEps = eps;
    dHat = eps * eps * lenScale2();
    dTol = dTolRel * dTolRel * lenScale2();
    return (int)planes.size() - 1;
}

bool HipOptimizer::anyIntersection()
{
    // isIntersected (Optimizer.cpp:2626-2659): analytic objects first, then the mesh against itself
    for (auto& h : planes)
        if (h->intersected(mesh.nV, mesh.d_x.p, mesh.d_dbc.p)) return true;
    return selfCollision && isIntersected();
}

// ---- self-contact ----------------------------------------------------------------------------------------------
void HipOptimizer::enableSelfCollision(HipContact* c, double eps)
{
    // `selfCollisionOn` + interior point; dHat = dHatEps^2 * bbox diagonal^2 (Optimizer.cpp:1534-1537, Config.cpp:41-45)
    if (!c || !c->surfaceSet) throw StateError("opt_enable_self_collision before set_surface");
    if (!planes.empty() && eps != dHatEps) throw ArgError("all collision objects of a context share one dHat (Config.cpp:41-45): got a different dHatEps");
    contact = c;
    selfCollision = true;
    dHatEps = eps;
    dHat = eps * eps * lenScale2();
    dTol = dTolRel * dTolRel * lenScale2(); // dTolRel = tuning[3], 1e-9 by default (Optimizer.cpp:102-109)
}

void HipOptimizer::computeXTilta()
{
    // Optimizer.cpp:1236-1257 on the host: only used when the caller overrides the velocity
    const size_t n3 = 3 * (size_t)mesh.nV;
    std::vector<double> xp(n3), vel(n3), xt(n3), acc(n3, 0.0);
    d_xPrev.download(xp.data(), n3, stream);
    d_vel.download(vel.data(), n3, stream);
    d_acc.download(acc.data(), n3, stream);
    for (int v = 0; v < mesh.nV; ++v)
        for (int c = 0; c < 3; ++c) {
            const size_t i = 3 * (size_t)v + c;
            if (mesh.isDBCVertex(v)) xt[i] = xp[i];
            else if (timeIntegration == 1) xt[i] = xp[i] + (vel[i] * dt + betaNM * (dtSq * gravity[c]) + (0.5 - betaNM) * (dtSq * acc[i])); // :1259-1277
            else xt[i] = xp[i] + (vel[i] * dt + dtSq * gravity[c]);
        }
    mesh.d_xTilde.upload(xt, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
}

void HipOptimizer::setTimeIntegration(int type, double beta, double gamma)
{
    timeIntegration = type;
    betaNM = beta;
    gammaNM = gamma;
    computeXTilta();
}

void HipOptimizer::getDbcState(double* out4) const
{
    out4[0] = completedStep;
    out4[1] = rhoDBC;
    out4[2] = projDBC ? 1.0 : 0.0;
    out4[3] = (double)tpIds.size();
}

void HipOptimizer::getKinematics(double* vel, double* acc, double* dxElastic)
{
    const size_t n3 = 3 * (size_t)mesh.nV;
    if (vel) d_vel.download(vel, n3, stream);
    if (acc) d_acc.download(acc, n3, stream);
    if (dxElastic) d_dxElastic.download(dxElastic, n3, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
}

// Optimizer::saveStatus (Optimizer.cpp:2964-3011): the reference's text checkpoint, readable by either implementation
void HipOptimizer::saveStatus(const std::string& path)
{
    const size_t n3 = 3 * (size_t)mesh.nV;
    std::vector<double> x(n3), vel(n3), acc(n3), dx(n3);
    mesh.d_x.download(x.data(), n3, stream);
    getKinematics(vel.data(), acc.data(), dx.data());
    std::ofstream out(path, std::ios::out);
    if (!out.is_open()) throw StateError("unable to create status file " + path);
    out << std::setprecision(std::numeric_limits<long double>::digits10 + 2);
    out << "timestep " << globalIterNum << "\n\n";
    auto rows = [&](const char* name, const std::vector<double>& a) {
        out << name << " " << mesh.nV << " 3\n";
        for (int v = 0; v < mesh.nV; ++v) out << a[3 * (size_t)v] << " " << a[3 * (size_t)v + 1] << " " << a[3 * (size_t)v + 2] << "\n";
    };
    rows("position", x);
    out << "\n";
    out << "velocity " << n3 << "\n";
    for (size_t i = 0; i < n3; ++i) out << vel[i] << "\n";
    out << "\n";
    rows("acceleration", acc);
    out << "\n";
    rows("dx_Elastic", dx);
    if (!out.good()) throw StateError("write error on status file " + path);
}

// restart branch of the Optimizer constructor (Optimizer.cpp:179-248): same token grammar, then V_prev = V and computeXTilta
void HipOptimizer::loadStatus(const std::string& path)
{
    specAsmValid = false;
    std::ifstream in(path);
    if (!in.is_open()) throw StateError("unable to open status file " + path);
    const size_t n3 = 3 * (size_t)mesh.nV;
    std::vector<double> x(n3), vel(n3, 0.0), acc(n3, 0.0), dx(n3, 0.0);
    mesh.d_x.download(x.data(), n3, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    auto readRows = [&](std::stringstream& ss, std::vector<double>& dst, bool zeroFirst) {
        int rowsIn = 0, dimIn = 0;
        ss >> rowsIn >> dimIn;
        if (rowsIn < 0 || rowsIn > mesh.nV || dimIn != 3) throw StateError("status file does not match the mesh");
        if (ss.fail()) throw StateError("malformed section header in status file " + path);
        if (zeroFirst) std::fill(dst.begin(), dst.end(), 0.0);
        for (int v = 0; v < rowsIn; ++v) in >> dst[3 * (size_t)v] >> dst[3 * (size_t)v + 1] >> dst[3 * (size_t)v + 2];
        // a truncated or malformed section leaves failbit set: a restart from a corrupt checkpoint must not go on with
        // partly read (or stale) arrays
        if (in.fail()) throw StateError("truncated or malformed section in status file " + path);
    };
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line);
        std::string token;
        ss >> token;
        if (token == "timestep") ss >> globalIterNum;
        else if (token == "position") readRows(ss, x, false);
        else if (token == "velocity") {
            long long n = 0;
            ss >> n;
            if (n < 0 || n > (long long)n3) throw StateError("status file does not match the mesh");
            std::fill(vel.begin(), vel.end(), 0.0);
            for (long long i = 0; i < n; ++i) in >> vel[i];
            if (in.fail()) throw StateError("truncated or malformed velocity section in status file " + path);
        }
        else if (token == "acceleration") readRows(ss, acc, true);
        else if (token == "dx_Elastic") readRows(ss, dx, true);
    }
    if (in.bad()) throw StateError("read error on status file " + path);
    mesh.d_x.upload(x, stream);
    d_xPrev.upload(x, stream);
    d_vel.upload(vel, stream);
    d_acc.upload(acc, stream);
    d_dxElastic.upload(dx, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    computeXTilta();
}

void HipOptimizer::setVelocity(const double* vel3nV)
{
    HIP_CHECK(hipMemcpyAsync(d_vel.p, vel3nV, 3 * (size_t)mesh.nV * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    computeXTilta();
}

void HipOptimizer::computeConstraintSets()
{
    if (!ipOn()) return;
    Tic t(timers[14], stream);
    for (auto& h : planes) h->build(contact->nSVI, contact->d_SVI.p, mesh.d_x.p, mesh.d_dbc.p, dHat); // Optimizer.cpp:2460-2462
    if (selfCollision) contact->buildConstraintSet(mesh, mesh.d_x.p, mesh.d_dbc.p, dHat); // Optimizer.cpp:2448-2470
}

bool HipOptimizer::isIntersected() { return contact->isIntersected(mesh, mesh.d_x.p, mesh.d_dbc.p); }

double HipOptimizer::kappaFloor() const
{
    // suggestKappa (Optimizer.cpp:2228-2233): kappaMinMultiplier (1e11, Config.hpp:139) * mean nodal mass / (4e-16 L^2 b''(1e-16 L^2))
    const double d = 1.0e-16 * mesh.bboxDiag2, t2 = d - dHat, lg = std::log(d / dHat);
    const double Hb = (lg * -2.0 - t2 * 4.0 / d) + 1.0 / (d * d) * (t2 * t2); // BarrierFunctions.hpp:76-83
    double avgMass = 0; // Mesh::avgNodeMass(dim): over the nodes of the tetrahedral components (Mesh.cpp:576-609)
    for (int v = 0; v < mesh.nV; ++v)
        if (!mesh.nElemNodes || mesh.inMesh[v]) avgMass += mesh.mass[v];
    avgMass /= std::max(mesh.nElemNodes, 1);
    return kappaMinMultiplier * avgMass / (4.0e-16 * mesh.bboxDiag2 * Hb);
}

void HipOptimizer::initKappa()
{
    // Optimizer.cpp:2236-2313: balance the barrier gradient of the active set against elasticity + inertia.  Once per time step; the two dot products are taken
    // on the device (round 6: both gradients used to travel to the host, 2 x 24 nV bytes per time step, to be dotted there)
    if (!nConstraints()) return;
    const int n3 = 3 * mesh.nV;
    elasticInertiaGradient(true); // computeGradient with solveIP == false (:2243-2245)
    if (dampingStiff > 0.0) dampingGradientAdd(true, d_gradient.p); // still part of it (:3519-3540)
    d_minusG.zero(stream);
    barrierGradientAdd(true, 1.0, true, d_minusG.p); // also clears the DBC rows (:2275-2277)
    launch_dot2(n3, d_minusG.p, d_gradient.p, d_partial.p, (int)d_partial.n, d_scalar.p + 6, stream);
    launch_publish(d_scalar.p + 6, h_scalar.dev, 4, stream); // two doubles = four words
    HIP_CHECK(hipStreamSynchronize(stream));
    const double num = h_scalar.p[0], den = h_scalar.p[1];
    double minKappa = -num / den;
    if (minKappa > 0.0) kappa = minKappa;
    minKappa = kappaFloor();
    if (kappa < minKappa) kappa = minKappa;
    const double kappaMax = 100 * kappaFloor(); // upperBoundKappa, :2216-2225
    if (kappa > kappaMax) kappa = kappaMax;
}

void HipOptimizer::postLineSearch()
{
    // Optimizer.cpp:2357-2445 (ADAPTIVE_KAPPA)
    if (!ipOn()) return;
    if (kappa == 0.0) {
        initKappa();
        return;
    }
    std::vector<double> d;
    bool updateKappa = false;
    for (size_t pi = 0; pi < planes.size() && !updateKappa; ++pi) {
        std::vector<int> verts;
        std::vector<double> was;
        for (size_t i = 0; i < closeHS.size(); ++i)
            if (closeHS[i].first == (int)pi) {
                verts.push_back(closeHS[i].second);
                was.push_back(closeHSVal[i]);
            }
        planes[pi]->evalDist2(verts, mesh.d_x.p, d);
        for (size_t i = 0; i < verts.size(); ++i)
            if (d[i] <= was[i]) updateKappa = true;
    }
    if (!updateKappa && selfCollision) {
        contact->evalStencils(closeID, mesh.d_x.p, d);
        for (size_t i = 0; i < closeID.size(); ++i)
            if (d[i] <= closeVal[i]) {
                updateKappa = true;
                break;
            }
    }
    if (updateKappa) {
        kappa *= 2.0;
        const double kappaMax = 100 * kappaFloor();
        if (kappa > kappaMax) kappa = kappaMax;
    }
    closeID.clear();
    closeVal.clear();
    closeHS.clear();
    closeHSVal.clear();
    for (size_t pi = 0; pi < planes.size(); ++pi) {
        planes[pi]->evalDist2(planes[pi]->set, mesh.d_x.p, d);
        for (size_t i = 0; i < planes[pi]->set.size(); ++i)
            if (d[i] < dTol) {
                closeHS.push_back({ (int)pi, planes[pi]->set[i] });
                closeHSVal.push_back(d[i]);
            }
    }
    if (!selfCollision) return;
    // ... and, in the same batch, whether the pattern holds every block of the sets the next assembly will use (computePrecondMtr asks first thing)
    int cov = 1;
    contact->closeStencils(mesh.d_x.p, dTol, closeID, closeVal, &lin, &cov);
    coverSets = contact->setsVersion;
    coverPattern = lin.patternVersion;
    coverAnswer = cov != 0;
}

void HipOptimizer::ensurePatchPlan()
{
    if (patchVersion == lin.patternVersion && patch.valid) return;
    patch.build(mesh, lin, stream);
    patchVersion = lin.patternVersion;
}
bool HipOptimizer::ownerMode() const
{
    // both halves sharded, and nothing in play that reads the WHOLE matrix every iteration (the lagged damping matrix is a second value array on the pattern
    // that enters energy and gradient through products with it: those runs keep the all-reduce of the values)
    return worldSize > 1 && lin.solverWorld() == worldSize && lin.solverType == 0 && lin.analyzed() && !(dampingStiff > 0.0); // (before the first analysis: the older scheme)
}

void HipOptimizer::ensureOwnerPlan()
{
    ensurePatchPlan();
    if (ownerPlanPatch == patchVersion && ownerPlanAnalysis == lin.analysisVersion) return;
    std::vector<int> owner;
    lin.nodeOwners(owner); // rank of the subtree that eliminates the node, -1 above the cut
    const int nV = mesh.nV;
    std::vector<unsigned char> need(nV, 0), mine(nV, 0);
    ownerNeededNodes = 0;
    for (int v = 0; v < nV; ++v) {
        const int o = (v < (int)owner.size()) ? owner[v] : -1;
        need[v] = (o < 0 || o == rank) ? 1 : 0;
        mine[v] = (o == rank || (o < 0 && rank == 0)) ? 1 : 0;
        ownerNeededNodes += need[v];
    }
    std::vector<int> list;
    const std::vector<int>&np = patch.topo.nodePtr, &nodes = patch.topo.nodes;
    for (int p = 0; p + 1 < (int)np.size(); ++p) {
        bool any = false;
        for (int j = np[p]; j < np[p + 1] && !any; ++j) any = need[nodes[j]] != 0;
        if (any) list.push_back(p);
    }
    nOwnerPatches = (int)list.size();
    if (list.empty()) list.push_back(0);
    d_ownerPatches.upload(list, stream);
    d_need.upload(need, stream);
    d_mine.upload(mine, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    ownerPlanPatch = patchVersion;
    ownerPlanAnalysis = lin.analysisVersion;
}

void HipOptimizer::maskAndReduceGradient(double* g)
{
    launch_keep_mine3(mesh.nV, d_mine.p, g, stream);
    reduceSum(g, 3LL * mesh.nV);
}

void HipOptimizer::completeMatrix()
{
    if (matrixComplete || worldSize <= 1) return;
    launch_keep_mine_rows(lin.numRows, d_mine.p, lin.d_ia.p, lin.d_a.p, stream);
    reduceSum(lin.d_a.p, (long long)lin.ja.size());
    matrixComplete = true;
}

void HipOptimizer::patchShard(int& pb, int& pe) const
{
    pb = (int)((long long)patch.nPatches * rank / worldSize);
    pe = (int)((long long)patch.nPatches * (rank + 1) / worldSize);
}

void HipOptimizer::barrierGradientAdd(bool projectDBC, double kappa_, bool activeOnly, double* grad_dev, int part)
{
    // barrier forces of the half-spaces and of the mesh against itself; the projected rows are cleared again at the end
    // (Optimizer.cpp:3452-3516).  activeOnly: initKappa leaves the mollified parallel-edge set out (:2262-2270)
    // part (owner-computes sharding): 0 = everything; 1 = only the self-contact stencils this rank evaluates, BEFORE the gradient exchange; 2 = the terms every
    // rank evaluates alike (half-spaces, lagged friction), after it
    if (part == 1) {
        if (contact && selfCollision)
            contact->gradientAdd(mesh.d_x.p, mesh.d_dbc.p, mesh.nV, dHat, kappa_, 0, grad_dev, true, !activeOnly, d_need.p);
        return;
    }
    for (auto& h : planes) h->gradientAdd(mesh.d_x.p, dHat, kappa_, grad_dev);
    if (!activeOnly && fricDHat > 0.0) { // Optimizer.cpp:3474-3478, 3504-3506
        for (auto& h : planes)
            if (h->friction > 0.0) h->frictionGradientAdd(mesh.d_x.p, d_xPrev.p, fricDHat, grad_dev);
        if (selfCollision && selfFric > 0.0) contact->frictionGradientAdd(mesh.d_x.p, d_xPrev.p, fricDHat, selfFric, grad_dev);
    }
    if (!contact) return;
    if (part == 2) { // the stencils went in before the exchange: only the projected rows are left to clear
        launch_clear_projected(mesh.nV, mesh.d_dbc.p, projectDBC ? 1 : 0, grad_dev, stream);
        return;
    }
    if (ownerMode() && !lin.rowBase.empty()) {
        // a caller outside computeGradient (initKappa's barrier-only gradient): this rank's stencils into a zeroed scratch vector, the designated entries
        // exchanged, the sum on top of the caller's vector
        ensureOwnerPlan();
        const size_t n3 = 3 * (size_t)mesh.nV;
        d_contactG.ensure(n3);
        d_contactG.zeroN(n3, stream);
        contact->gradientAdd(mesh.d_x.p, mesh.d_dbc.p, mesh.nV, dHat, kappa_, 0, d_contactG.p, selfCollision, selfCollision && !activeOnly, d_need.p);
        maskAndReduceGradient(d_contactG.p);
        launch_axpy((long long)n3, 1.0, d_contactG.p, grad_dev, stream);
        launch_clear_projected(mesh.nV, mesh.d_dbc.p, projectDBC ? 1 : 0, grad_dev, stream);
        return;
    }
    contact->gradientAdd(mesh.d_x.p, mesh.d_dbc.p, mesh.nV, dHat, kappa_, projectDBC, grad_dev, selfCollision, selfCollision && !activeOnly);
}

void HipOptimizer::elasticInertiaGradient(bool projectDBC, bool finish)
{
    if (lin.rowBase.empty()) { // no pattern yet: tet-parallel atomic path
        launch_node_init(view(), projectDBC, rank == 0, nullptr, d_gradient.p, stream);
        launch_assemble(view(), elasticCoef(), projectDBC, d_gradient.p, nullptr, stream);
        reduceSum(d_gradient.p, 3LL * mesh.nV);
        neumannGradientAdd(d_gradient.p);
        return;
    }
    ensurePatchPlan();
    if (ownerMode()) { // this rank's patches only; the caller (computeGradient) adds its share of the barrier forces and exchanges the sum once
        ensureOwnerPlan();
        d_gradient.zero(stream);
        launch_assemble_patches(view(), patch, 0, nOwnerPatches, elasticCoef(), projectDBC, d_gradient.p, nullptr, stream, d_ownerPatches.p);
        if (finish) { // a caller that wants the elastic + inertia gradient by itself (initKappa)
            maskAndReduceGradient(d_gradient.p);
            neumannGradientAdd(d_gradient.p);
        }
        return;
    }
    int pb, pe;
    patchShard(pb, pe);
    if (worldSize > 1) d_gradient.zero(stream);
    launch_assemble_patches(view(), patch, pb, pe, elasticCoef(), projectDBC, d_gradient.p, nullptr, stream);
    reduceSum(d_gradient.p, 3LL * mesh.nV);
    neumannGradientAdd(d_gradient.p);
}

void HipOptimizer::computeGradient(bool projectDBC)
{
    const bool owner = ownerMode() && !lin.rowBase.empty();
    elasticInertiaGradient(projectDBC, !owner);
    if (owner) {
        // owner-computes: elastic + inertia forces of this rank's patches, its share of the self-contact stencils (those that touch a node it owns rows
        // of), then ONE exchange in which every node is contributed by its designated rank; what is evaluated identically everywhere follows
        if (ipOn()) barrierGradientAdd(projectDBC, kappa, false, d_gradient.p, 1);
        maskAndReduceGradient(d_gradient.p);
        neumannGradientAdd(d_gradient.p);
        if (ipOn()) barrierGradientAdd(projectDBC, kappa, false, d_gradient.p, 2);
        penaltyGradientAdd(projectDBC);
        return;
    }
    if (ipOn()) barrierGradientAdd(projectDBC, kappa, false, d_gradient.p);
    penaltyGradientAdd(projectDBC);
}

// tail of Optimizer::computeGradient when the Dirichlet rows are not all projected: rows of the nodes that still are get
// cleared (Optimizer.cpp:3512-3516; with projectDBC the element pass never wrote them), then augmentMDBCGradient (:3542-3544)
void HipOptimizer::penaltyGradientAdd(bool projectDBC)
{
    if (dampingStiff > 0.0) dampingGradientAdd(projectDBC, d_gradient.p); // Optimizer.cpp:3519-3540 (rows of D of Dirichlet nodes are empty)
    if (projectDBC) return;
    launch_clear_projected(mesh.nV, mesh.d_dbc.p, 0, d_gradient.p, stream);
    if (rhoDBC && !tpIds.empty()) launch_mdbc_gradient(mdbc(), mesh.d_x.p, rhoDBC, d_gradient.p, stream);
}

void HipOptimizer::computePrecondMtr(bool projectDBC, bool withGradient)
{
    if (lin.rowBase.empty()) throw StateError("computePrecondMtr needs a pattern built by set_pattern");
    // The common case by far: the pattern already holds every block the current sets need (it only grows, and it is built with
    // look-ahead).  One small kernel answers that; the host-side connectivity (hundreds of thousands of pairs to build, filter
    // and sort) runs only when a pair is missing.  Lagged friction keeps its own set: always the host path there.
    const bool frictionPairs = fricDHat > 0.0 && selfFric > 0.0;
    static const bool timeIt = std::getenv("IPCGPU_PATTERN_TIMES") != nullptr; // stderr: where a pattern change spends its time
    auto tLap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timeIt) return;
        HIP_CHECK(hipStreamSynchronize(stream));
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "pattern change: %-34s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tLap).count());
        tLap = now;
    };
    // (the coverage answer usually arrived with the close-stencil bookkeeping at the end of the last iteration, postLineSearch: same sets, same pattern)
    const bool coverCached = selfCollision && coverSets == contact->setsVersion && coverPattern == lin.patternVersion;
    if (selfCollision && (frictionPairs || !(coverCached ? coverAnswer : contact->patternCovers(lin)))) {
        lap("coverage check");
        // the pattern follows the contact connectivity (augmentConnectivity into vNeighbor_IP, Optimizer.cpp:3560-3612);
        // only pairs that are not mesh edges change it
        std::vector<std::pair<int, int>> extra, fresh;
        // The connectivity of the live sets on the host (read-back of the tuples, node pairs, filter, sort: 1-2 ms at 40 K nodes) is needed only for the lagged
        // friction set and for a look-ahead below 1.  Otherwise (round 6) the device has already said that the pattern lacks a block of the live sets -- a new
        // analysis is certain --, and the look-ahead list below, the FULL stencils of every candidate within lookahead() x dHat >= dHat, contains every node pair
        // of the live sets: their stencils are sub-stencils of candidates, the mollified pairs' four nodes are their candidates' four nodes.
        const bool liveOnHost = frictionPairs || lookahead() < 1.0;
        if (liveOnHost) {
            if (contact->nActive() + contact->nPara()) contact->connectivity(extra);
            if (fricDHat > 0.0 && selfFric > 0.0) contact->frictionConnectivity(extra); // lagged set (:3565-3566)
            for (const auto& e : extra) {
                const int* b = mesh.nb.data() + mesh.nbPtr[e.first];
                const int* en = mesh.nb.data() + mesh.nbPtr[e.first + 1];
                if (!std::binary_search(b, en, e.second)) fresh.push_back(e);
            }
            std::sort(fresh.begin(), fresh.end());
            fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
        }
        lap("connectivity of the live sets");
        // The reference rebuilds pattern + symbolic analysis whenever the contact graph changes (:3570-3592).  Here the pattern
        // only ever GROWS inside the stepper: pairs that left the constraint set keep their (zero) slots, so a new analysis is
        // needed only when a pair shows up that no earlier iteration had.  Same matrix, fewer host-side analyses; the
        // union is dropped again once it has grown far beyond the live set.
        if (!liveOnHost || !std::includes(curExtra.begin(), curExtra.end(), fresh.begin(), fresh.end())) {
            // Look-ahead: contact spreads, so the next iterations bring pairs that are a little farther apart now.  The new
            // pattern is built from the constraint set at a larger distance (lookahead() * dHat, squared distances): its
            // blocks hold explicit zeros until the pairs become active, and pattern + symbolic analysis (tens of ms on the
            // host) are needed far less often.  Costs two extra constraint-set builds per analysis.
            std::vector<std::pair<int, int>> padded = fresh;
            const double pad = lookahead();
            if (pad >= 1.0) {
                std::vector<std::pair<int, int>> ahead;
                if (pad > 1.0) contact->buildConstraintSet(mesh, mesh.d_x.p, mesh.d_dbc.p, pad * dHat);
                contact->candidateConnectivitySorted(ahead); // full stencils: closest-feature changes need no new blocks; sorted, unique
                if (pad > 1.0) contact->buildConstraintSet(mesh, mesh.d_x.p, mesh.d_dbc.p, dHat); // back to the real sets
                std::vector<std::pair<int, int>> aheadNew;
                aheadNew.reserve(ahead.size());
                for (const auto& e : ahead) {
                    const int* b = mesh.nb.data() + mesh.nbPtr[e.first];
                    const int* en = mesh.nb.data() + mesh.nbPtr[e.first + 1];
                    if (!std::binary_search(b, en, e.second)) aheadNew.push_back(e);
                }
                padded.clear(); // fresh and aheadNew are both sorted and unique: their union is one linear merge
                std::set_union(fresh.begin(), fresh.end(), aheadNew.begin(), aheadNew.end(), std::back_inserter(padded));
            }
            lap("look-ahead sets + connectivity");
            std::vector<std::pair<int, int>> merged;
            std::set_union(curExtra.begin(), curExtra.end(), padded.begin(), padded.end(), std::back_inserter(merged));
            if (merged.size() > 3 * padded.size() + 4096) merged = padded;
            curExtra.swap(merged);
            nPatternChanges++;
            std::vector<int> flat;
            flat.reserve(2 * curExtra.size());
            for (const auto& e : curExtra) {
                flat.push_back(e.first);
                flat.push_back(e.second);
            }
            lap("union + flatten");
            {
                Tic t(timers[1], stream);
                lin.set_pattern(mesh, (int)curExtra.size(), flat.data());
            }
            lap("set_pattern");
            {
                // the assembly plan of the new pattern is built on a second host thread while this one runs the symbolic analysis:
                // both only read the pattern, and the GPU is idle either way
                int dev = 0;
                HIP_CHECK(hipGetDevice(&dev));
                std::exception_ptr planErr;
                std::thread planThread([&] {
                    try {
                        HIP_CHECK(hipSetDevice(dev));
                        ensurePatchPlan();
                    }
                    catch (...) {
                        planErr = std::current_exception();
                    }
                });
                std::exception_ptr anaErr;
                try {
                    Tic t(timers[2], stream);
                    lin.analyze_pattern(&mesh);
                }
                catch (...) {
                    anaErr = std::current_exception();
                }
                planThread.join();
                if (anaErr) std::rethrow_exception(anaErr);
                if (planErr) std::rethrow_exception(planErr);
            }
            lap("analyze_pattern + patch plan");
        }
    }
    // setZero (Optimizer.cpp:3616), elastic Hessian (:3619-3623) and the mass / DBC diagonal (:3638-3668) are one
    // pass: every owned CSR row is written exactly once by the patch that owns its node
    const bool planStale = !(patchVersion == lin.patternVersion && patch.valid);
    ensurePatchPlan();
    if (planStale) lap("patch plan");
    if (ownerMode()) {
        // Owner-computes (round 4): this rank's patches write the complete CSR rows of the nodes it owns or shares, its share of the contact stencils adds
        // their blocks to those rows -- and that is all its fronts ever read (an entry belongs to the front of whichever of its two nodes is eliminated
        // first; the other node then sits in the same subtree or above the cut).  No matrix value crosses ranks.  The rows of other ranks' nodes stay zero.
        ensureOwnerPlan();
        lin.setZero();
        if (withGradient) d_gradient.zero(stream);
        launch_assemble_patches(view(), patch, 0, nOwnerPatches, elasticCoef(), projectDBC, withGradient ? d_gradient.p : nullptr, lin.d_a.p, stream,
            d_ownerPatches.p);
        matrixComplete = false;
        if (ipOn() && selfCollision) contact->hessianAdd(mesh.d_x.p, mesh.d_dbc.p, lin, dHat, kappa, projectDBC, lin.d_a.p, d_need.p, /*deferCheck=*/true);
        if (withGradient) {
            if (ipOn()) barrierGradientAdd(projectDBC, kappa, false, d_gradient.p, 1);
            maskAndReduceGradient(d_gradient.p);
            neumannGradientAdd(d_gradient.p);
            if (ipOn()) barrierGradientAdd(projectDBC, kappa, false, d_gradient.p, 2);
        }
        if (ipOn()) { // evaluated alike on every rank (vertex-wise diagonal blocks / the lagged friction set): added to the rows each rank holds
            for (auto& h : planes)
                h->hessianAdd(mesh.d_x.p, mesh.d_dbc.p, lin.d_rowBase.p, lin.d_rowLen.p, dHat, kappa, projectDBC, lin.d_a.p);
            if (fricDHat > 0.0) {
                for (auto& h : planes)
                    if (h->friction > 0.0)
                        h->frictionHessianAdd(mesh.d_x.p, d_xPrev.p, mesh.d_dbc.p, lin.d_rowBase.p, lin.d_rowLen.p, fricDHat, projectDBC, lin.d_a.p);
                if (selfCollision && selfFric > 0.0)
                    contact->frictionHessianAdd(mesh.d_x.p, d_xPrev.p, mesh.d_dbc.p, lin, fricDHat, selfFric, projectDBC, lin.d_a.p);
            }
        }
        if (withGradient) penaltyGradientAdd(projectDBC);
        if (!projectDBC && rhoDBC && !tpIds.empty()) launch_mdbc_hessian(mdbc(), lin.d_ia.p, rhoDBC, lin.d_a.p, stream);
        return;
    }
    int pb, pe;
    patchShard(pb, pe);
    if (worldSize > 1) {
        lin.setZero();
        if (withGradient) d_gradient.zero(stream);
    }
    launch_assemble_patches(view(), patch, pb, pe, elasticCoef(), projectDBC, withGradient ? d_gradient.p : nullptr,
        lin.d_a.p, stream);
    matrixComplete = true;
    // (the older scheme, kept for contexts whose solver is not sharded: partial matrices summed by one all-reduce of the values; the contact stencils are
    // then evaluated alike on every rank, after the exchange)
    const bool contactSharded = false;
    if (worldSize > 1) {
        reduceSum(lin.d_a.p, (long long)lin.ja.size());
        if (withGradient) reduceSum(d_gradient.p, 3LL * mesh.nV);
    }
    if (withGradient) neumannGradientAdd(d_gradient.p);
    if (ipOn()) { // barrier blocks, PSD-projected per stencil (Optimizer.cpp:3625-3636, 3670-3676)
        if (withGradient) barrierGradientAdd(projectDBC, kappa, false, d_gradient.p);
        for (auto& h : planes)
            h->hessianAdd(mesh.d_x.p, mesh.d_dbc.p, lin.d_rowBase.p, lin.d_rowLen.p, dHat, kappa, projectDBC, lin.d_a.p);
        if (selfCollision && !contactSharded) contact->hessianAdd(mesh.d_x.p, mesh.d_dbc.p, lin, dHat, kappa, projectDBC, lin.d_a.p, nullptr, /*deferCheck=*/true);
        if (fricDHat > 0.0) { // Optimizer.cpp:3677-3702
            for (auto& h : planes)
                if (h->friction > 0.0)
                    h->frictionHessianAdd(mesh.d_x.p, d_xPrev.p, mesh.d_dbc.p, lin.d_rowBase.p, lin.d_rowLen.p, fricDHat, projectDBC, lin.d_a.p);
            if (selfCollision && selfFric > 0.0)
                contact->frictionHessianAdd(mesh.d_x.p, d_xPrev.p, mesh.d_dbc.p, lin, fricDHat, selfFric, projectDBC, lin.d_a.p);
        }
    }
    if (dampingStiff > 0.0) { // addCoeff(dampingMtr, 1.0), Optimizer.cpp:3707-3709
        if (dampPatternVersion != lin.patternVersion) assembleDampingMtr();
        launch_axpy((long long)lin.ja.size(), 1.0, d_damp.p, lin.d_a.p, stream);
    }
    if (withGradient) penaltyGradientAdd(projectDBC);
    if (!projectDBC && rhoDBC && !tpIds.empty()) launch_mdbc_hessian(mdbc(), lin.d_ia.p, rhoDBC, lin.d_a.p, stream); // :3711-3713
}

bool HipOptimizer::checkInversion()
{
    if (mesh.energyType == 1) return true; // Optimizer.cpp:252,517,545,2710: only under getNeedElemInvSafeGuard()
    d_flag.zero(stream);
    launch_check_inversion(view(), d_flag.p, stream);
    launch_publish(d_flag.p, h_flag.dev, 1, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    int f = h_flag.p[0];
    if (worldSize > 1) {
        h_scalar.p[1] = (double)f;
        HIP_CHECK(hipMemcpyAsync(d_scalar.p + 1, h_scalar.p + 1, sizeof(double), hipMemcpyHostToDevice, stream));
        reduceSum(d_scalar.p + 1, 1);
        f = readScalar(d_scalar.p + 1) != 0.0;
    }
    return f == 0;
}

double HipOptimizer::fullCcd(double slackness, double stepSize)
{
    if (contact->ccdMode == 1) {
        int arg3[3];
        const double a = contact->ccdFullReference(mesh, mesh.d_x.p, d_searchDir.p, mesh.d_dbc.p, slackness, stepSize, nullptr, arg3, nullptr);
        lastCCDPair[0] = arg3[1];
        lastCCDPair[1] = arg3[2];
        return a;
    }
    return contact->ccdFull(mesh, mesh.d_x.p, d_searchDir.p, mesh.d_dbc.p, slackness, stepSize, lastCCDPair, nullptr);
}

double HipOptimizer::filterStepSize(const double* p_dev, double stepSize)
{
    // Energy.cpp:565-581: min over elements of the root, applied only when 0 < min < stepSize
    if (mesh.energyType == 1) return stepSize; // :567 needElemInvSafeGuard
    launch_fill(d_scalar.p + 2, 1, 1e20, stream);
    launch_inversion_step(view(), p_dev, 0.2, stepSize, d_scalar.p + 2, stream);
    reduceMin(d_scalar.p + 2, 1);
    const double t = readScalar(d_scalar.p + 2);
    if (t > 0.0 && t < stepSize) stepSize = t;
    return stepSize;
}

void HipOptimizer::stepForward(const double* x0_dev, double alpha)
{
    launch_step_forward(3 * mesh.nV, x0_dev, d_searchDir.p, alpha, mesh.d_x.p, stream);
}

#ifndef ASM_TIME_EVERY
#define ASM_TIME_EVERY 8
#endif
void HipOptimizer::speculativeAssembly()
{
    specAsmValid = false;
    if (!specAsmOn || !fastPath() || !projDBC || lin.rowBase.empty()) return;
    ensurePatchPlan();
    d_aSpec.alloc(lin.d_a.n); // same capacity as the solver's value array: the two are swapped
    d_gradSpec.alloc(d_gradient.n);
    // The assembly bucket of the timers is fed by a pair of events around the launch -- on every ASM_TIME_EVERY-th launch only, counted ASM_TIME_EVERY times
    // (round 6): a timing event is a barrier packet that drains the stream in front of the assembly and behind it, every iteration, for a bucket that holds 2 % of it
    const bool timeIt = (asmLaunches++ % ASM_TIME_EVERY) == 0;
    if (timeIt) {
        resolveEventTimers(); // (the last timed assembly: finished long ago)
        HIP_CHECK(hipEventRecord(evAsm0, stream));
    }
    launch_assemble_patches(view(), patch, 0, patch.nPatches, elasticCoef(), 1, d_gradSpec.p, d_aSpec.p, stream); // what computePrecondMtr(true, true) launches on this path
    if (timeIt) {
        HIP_CHECK(hipEventRecord(evAsm1, stream));
        evAsmPending = true;
        evAsmWeight = ASM_TIME_EVERY; // (this pair stands for ASM_TIME_EVERY launches)
    }
    specAsmValid = true;
}

void HipOptimizer::computeSearchDir(bool projectDBC)
{
    (void)projectDBC;
    bool ok;
    const bool twoCalls = false; // (factorize(), then solve(): the A/B of profiles/r03h_bench_line_*forward_overlap.json)
    launch_negate(3 * mesh.nV, d_gradient.p, d_minusG.p, stream);
    const bool noSpec = false; // (synchronise after the solve, again after the trial step: profiles/r03t_trial_ahead_ab.txt)
    cachedTrialValid = false;
    cachedDistValid = false; // (|p|_inf of the direction this call replaces)
    if (fastPath() && !twoCalls && !noSpec) {
        // ONE synchronisation per Newton iteration.  Behind factorisation + sweeps, on the same stream and without the host in between: |p|_inf
        // (convergence test of the next pass, Optimizer.cpp:1869-1879), the inversion step filter (:1887), E at the iterate (:2681), then the
        // first trial of the line search -- step size decided on the device from the filter's result, the step, its inversion flag
        // (:2710-2717), E at the trial point (:2757).  The pivot flag of the factorisation arrives with them.  The host then only decides:
        // bad pivot -> back to the iterate and the diagonal fallback; trial inverted or uphill -> back to the iterate and the general loop.
        Tic t(timers[3], stream); // (its synchronisation is the one)
        const size_t bytes = 3 * (size_t)mesh.nV * sizeof(double);
        if (lin.factorizeSolve(d_minusG.p, d_searchDir.p, /*wait=*/false)) {
            // ten launches (fifteen before round 4: the resets, the copy + step-size + step and the two read-backs are one launch each now)
            launch_iter_reset(d_scalar.p, d_flag.p, stream);
            launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
            if (mesh.energyType != 1) launch_inversion_step(view(), d_searchDir.p, 0.2, 1.0, d_scalar.p + 2, stream); // (the trial step starts at 1)
            launch_energy(view(), elasticCoef(), true, true, d_partial.p, (int)d_partial.n, d_scalar.p, stream);
            launch_trial_step_fused(3 * mesh.nV, mesh.d_x.p, d_x0.p, d_searchDir.p, d_scalar.p + 2, mesh.energyType != 1, d_scalar.p + 6, stream);
            if (mesh.energyType != 1) launch_check_inversion(view(), d_flag.p, stream);
            launch_energy(view(), elasticCoef(), true, true, d_partial.p, (int)d_partial.n, d_scalar.p + 1, stream);
            launch_publish2(d_flag.p, h_flag.dev, 1, d_scalar.p, h_scalar.dev, 14, stream); // the inversion flag + 7 doubles
            // the ONE synchronisation waits for the read-back, not for the stream: behind it runs the next pass's assembly at the trial point, enqueued ahead
            if (!evTail) HIP_CHECK(hipEventCreateWithFlags(&evTail, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(evTail, stream));
            speculativeAssembly();
            HIP_CHECK(hipEventSynchronize(evTail));
            t.nosync = specAsmValid;
            if (lin.lastPivotsOk()) {
                cachedE0 = h_scalar.p[0];
                cachedTrialE = h_scalar.p[1];
                cachedFilter = h_scalar.p[2];
                cachedDist = h_scalar.p[3];
                cachedAlpha = h_scalar.p[6];
                cachedTrialInverted = mesh.energyType != 1 && h_flag.p[0] != 0;
                cachedDistValid = cachedE0Valid = cachedTrialValid = true;
                return;
            }
            HIP_CHECK(hipMemcpyAsync(mesh.d_x.p, d_x0.p, bytes, hipMemcpyDeviceToDevice, stream)); // the step was taken along garbage
            specAsmValid = false;
            t.nosync = false;
        }
        ok = false;
    }
    else {
        // the right-hand side is known before the factorisation starts: the forward sweep of each level runs beside the pivot chain of
        // the levels above (MfNumeric::factorizeSolve); this bucket then holds factorisation + both sweeps
        Tic t(timers[3], stream);
        if (twoCalls) ok = lin.factorize();
        else if (worldSize == 1) {
            // one process: nothing waits inside the solver; |p|_inf for the convergence test of the next pass (Optimizer.cpp:1869-1879) is enqueued behind
            // the sweeps and the pivot flag comes back with it -- one synchronisation (round 6; two before, and a third at the head of the next pass)
            ok = lin.factorizeSolve(d_minusG.p, d_searchDir.p, /*wait=*/false);
            if (ok) {
                launch_fill(d_scalar.p + 3, 1, 0.0, stream);
                launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
                launch_publish(d_scalar.p + 3, h_scalar.dev + 3, 2, stream);
                HIP_CHECK(hipStreamSynchronize(stream));
                ok = lin.lastPivotsOk();
                if (ok) {
                    cachedDist = h_scalar.p[3];
                    cachedDistValid = true;
                }
            }
        }
        else ok = lin.factorizeSolve(d_minusG.p, d_searchDir.p);
    }
    Tic t(timers[4], stream);
    if (!ok) {
        completeMatrix(); // (owner-computes sharding: the diagonal of the rows other ranks hold)
        lin.precondition_diag(d_minusG.p, d_searchDir.p); // Optimizer.cpp:2331-2348
    }
    else if (twoCalls) lin.solve(d_minusG.p, d_searchDir.p);
    if (fastPath()) {
        // everything the host needs next, behind the solve on the same stream, read back with ONE synchronisation (the Tic's): |p|_inf
        // (convergence test of the next pass, Optimizer.cpp:1869-1879), the inversion step filter (:1887) and E at the current iterate
        // (line search entry, :2681)
        launch_fill(d_scalar.p + 3, 1, 0.0, stream);
        launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
        launch_fill(d_scalar.p + 2, 1, 1e20, stream);
        if (mesh.energyType != 1) launch_inversion_step(view(), d_searchDir.p, 0.2, 1.0, d_scalar.p + 2, stream); // (the trial step starts at 1)
        launch_energy(view(), elasticCoef(), true, true, d_partial.p, (int)d_partial.n, d_scalar.p, stream);
        launch_publish(d_scalar.p, h_scalar.dev, 8, stream); // 4 doubles = 8 words
        HIP_CHECK(hipStreamSynchronize(stream));
        cachedE0 = h_scalar.p[0];
        cachedFilter = h_scalar.p[2];
        cachedDist = h_scalar.p[3];
        cachedDistValid = cachedE0Valid = true;
    }
    else {
        // the barrier Hessian's deferred "pair outside the pattern" flag has arrived with the solve's synchronisation; |p|_inf as above where it has not
        // come back yet (several processes, or the diagonal fallback just replaced the direction)
        if (!cachedDistValid) {
            launch_fill(d_scalar.p + 3, 1, 0.0, stream);
            launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
            launch_publish(d_scalar.p + 3, h_scalar.dev + 3, 2, stream);
            HIP_CHECK(hipStreamSynchronize(stream));
            cachedDist = h_scalar.p[3];
            cachedDistValid = true;
        }
        if (selfCollision) contact->takeHessianError();
    }
}

bool HipOptimizer::fastPath() const
{
    return worldSize == 1 && !ipOn() && nbcGroups.empty() && !(dampingStiff > 0.0) && !(rhoDBC && !tpIds.empty());
}

void HipOptimizer::resolveEventTimers()
{
    if (!evAsmPending) return;
    HIP_CHECK(hipEventSynchronize(evAsm1));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, evAsm0, evAsm1));
    timers[0] += 1.0e-3 * (double)ms * evAsmWeight;
    evAsmPending = false;
}

void HipOptimizer::lineSearch(double& stepSize)
{
    const size_t bytes = 3 * (size_t)mesh.nV * sizeof(double);
    if (cachedTrialValid && fastPath()) {
        // the first trial came back with the solve (computeSearchDir): d_x0 holds the iterate, x the trial point
        cachedTrialValid = cachedE0Valid = false;
        lastEnergyVal = cachedE0; // Optimizer.cpp:2681
        if (!cachedTrialInverted && !(cachedTrialE > lastEnergyVal)) {
            lastEnergyVal = cachedTrialE;
            return;
        }
        HIP_CHECK(hipMemcpyAsync(mesh.d_x.p, d_x0.p, bytes, hipMemcpyDeviceToDevice, stream)); // back to the iterate: the general loop redoes the step
        specAsmValid = false;
    }
    else if (cachedE0Valid && fastPath()) {
        // E at the iterate came back with the solve; the trial step, its inversion flag and E at the trial point are enqueued together
        // and read with one synchronisation.  Anything but "not inverted and E decreased" falls through to the general loop below.
        cachedE0Valid = false;
        lastEnergyVal = cachedE0; // Optimizer.cpp:2681
        bool done = false;
        double testingE = 0.0;
        {
            Tic t(timers[5], stream);
            HIP_CHECK(hipMemcpyAsync(d_x0.p, mesh.d_x.p, bytes, hipMemcpyDeviceToDevice, stream));
            stepForward(d_x0.p, stepSize);
            int inverted = 0;
            if (mesh.energyType != 1) {
                d_flag.zero(stream);
                launch_check_inversion(view(), d_flag.p, stream);
                launch_publish(d_flag.p, h_flag.dev, 1, stream);
            }
            launch_energy(view(), elasticCoef(), true, true, d_partial.p, (int)d_partial.n, d_scalar.p, stream);
            launch_publish(d_scalar.p, h_scalar.dev, 2, stream);
            HIP_CHECK(hipStreamSynchronize(stream));
            if (mesh.energyType != 1) inverted = h_flag.p[0];
            testingE = h_scalar.p[0];
            done = !inverted && !(testingE > lastEnergyVal);
        }
        if (done) {
            lastEnergyVal = testingE;
            return;
        }
        HIP_CHECK(hipMemcpyAsync(mesh.d_x.p, d_x0.p, bytes, hipMemcpyDeviceToDevice, stream)); // back to the iterate: the general loop redoes the step
    }
    {
        Tic t(timers[9], stream);
        lastEnergyVal = computeEnergyVal(); // Optimizer.cpp:2681
    }
    {
        Tic t(timers[5], stream);
        HIP_CHECK(hipMemcpyAsync(d_x0.p, mesh.d_x.p, bytes, hipMemcpyDeviceToDevice, stream));
        stepForward(d_x0.p, stepSize);
        while (!checkInversion()) { // Optimizer.cpp:2710-2717
            stepSize /= 2.0;
            if (stepSize == 0.0) break;
            stepForward(d_x0.p, stepSize);
        }
        if (ipOn())
            while (anyIntersection()) { // Optimizer.cpp:2719-2736
                stepSize /= 2.0;
                stepForward(d_x0.p, stepSize);
            }
    }
    computeConstraintSets();
    double testingE;
    {
        Tic t(timers[9], stream);
        testingE = computeEnergyVal();
    }
    const double LFStepSize = stepSize;
    while (testingE > lastEnergyVal && stepSize > 0.0) { // Optimizer.cpp:2761-2797
        stepSize /= 2.0;
        if (stepSize == 0.0) break;
        {
            Tic t(timers[5], stream);
            stepForward(d_x0.p, stepSize);
        }
        computeConstraintSets();
        Tic t(timers[9], stream);
        testingE = computeEnergyVal();
    }
    if (stepSize < LFStepSize && ipOn()) { // Optimizer.cpp:2799-2811
        bool needRecomputeCS = false;
        while (anyIntersection()) {
            stepSize /= 2.0;
            stepForward(d_x0.p, stepSize);
            needRecomputeCS = true;
        }
        if (needRecomputeCS) computeConstraintSets(); // lastEnergyVal keeps the pre-halving value, as in the reference
    }
    lastEnergyVal = testingE;
}

void HipOptimizer::precompute()
{
    specAsmValid = false;
    // Optimizer.cpp:457-507
    if (!initialised) throw StateError("opt_precompute before opt_init");
    // Optimizer.cpp:258-263: the reference ends its process on an intersecting start (every line search after it would halve forever)
    if (anyIntersection()) throw StateError("intersection detected in initial configuration");
    {
        Tic t(timers[1], stream);
        lin.set_pattern(mesh, 0, nullptr);
        curExtra.clear();
    }
    computeConstraintSets();
    computeDampingMtr(); // computePrecondMtr(..., updateDamping): Optimizer.cpp:470, 3598-3612
    {
        Tic t(timers[0], stream);
        computePrecondMtr(true, false); // re-patterns + analyses when contact pairs are already active
    }
    if (!lin.analyzed()) {
        Tic t(timers[2], stream);
        lin.analyze_pattern(&mesh);
    }
    lastEnergyVal = computeEnergyVal();
}

void HipOptimizer::beginTimestep()
{
    specAsmValid = false;
    if (!initialised) throw StateError("opt_begin_timestep before opt_init");
    if (!lin.analyzed()) throw StateError("opt_begin_timestep before opt_precompute");
    Tic t(timers[11], stream);
    // The contact-free twist (BASELINE configs[1]; round 6): everything the set-up of a time step asks the device -- inverted before the motion? the filter's bound
    // on the scripted step, the step, inverted after it? the energy there -- is enqueued as ONE batch and read back with one synchronisation (seven before: each
    // question waited for its answer).  Any answer but the usual one ("no", "no") falls back to the general sequence below from the state it left off.
    const bool batched = BEGIN_BATCHED && fastPath() && nHandles && dbcGroups.empty() && !selfCollision && planes.empty() && !(warmStart >= 1 && warmStart <= 5);
    if (batched) {
        const bool guard = mesh.energyType != 1; // Optimizer.cpp:252, 517: only under getNeedElemInvSafeGuard()
        d_flag.zero(stream);
        if (guard) launch_check_inversion(view(), d_flag.p, stream);
        d_searchDir.zero(stream);
        stepStartTime = stepEndTime; // AnimScripter.cpp:1406-1407
        stepEndTime += dt;
        setDBCVertices();
        launch_twist_dir(nHandles, d_handleIds.p, d_handleAng.p, rotCenter[1], rotCenter[2], mesh.d_x.p, d_searchDir.p, stream);
        buildTargetPositions(/*deferTolerance=*/true);
        launch_fill(d_scalar.p + 2, 1, 1e20, stream);
        if (guard) launch_inversion_step(view(), d_searchDir.p, 0.2, 1.0, d_scalar.p + 2, stream); // filterStepSize(p, 1.0) ...
        launch_trial_step_fused(3 * mesh.nV, mesh.d_x.p, d_x0.p, d_searchDir.p, d_scalar.p + 2, guard, d_scalar.p + 6, stream); // ... applied by its rule; x0 = x, x += step p
        if (guard) launch_check_inversion(view(), d_flag.p + 1, stream);
        launch_energy(view(), elasticCoef(), true, true, d_partial.p, (int)d_partial.n, d_scalar.p, stream); // computeEnergyVal() of this configuration: elasticity + inertia
        launch_publish2(d_flag.p, h_flag.dev, 2, d_scalar.p, h_scalar.dev, 14, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
        if (h_flag.p[0]) throw StateError("element inversion before scripted motion (Optimizer.cpp:517-522)");
        finishTolerance();
        double stepSize = h_scalar.p[6];
        bool energyKnown = true;
        if (h_flag.p[1]) { // inverted behind the filtered step: the halving loop (AnimScripter.cpp:2172-2180) from here
            energyKnown = false;
            do {
                stepSize /= 2.0;
                stepForward(d_x0.p, stepSize);
            } while (!checkInversion());
        }
        if (stepSize < 1.0) dbcIncomplete++;
        completedStep = stepSize;
        d_searchDir.zero(stream);
        initSubProblem();
        lastEnergyVal = energyKnown ? h_scalar.p[0] : computeEnergyVal(); // Optimizer.cpp:1609
        k = 0;
        return;
    }
    if (!checkInversion()) throw StateError("element inversion before scripted motion (Optimizer.cpp:517-522)");
    d_searchDir.zero(stream);
    stepStartTime = stepEndTime; // AnimScripter.cpp:1406-1407
    stepEndTime += dt;
    setDBCVertices();
    // stepAnimScript, AST_TWIST (AnimScripter.cpp:1674-1684, 2140-2215)
    if (nHandles) launch_twist_dir(nHandles, d_handleIds.p, d_handleAng.p, rotCenter[1], rotCenter[2], mesh.d_x.p, d_searchDir.p, stream);
    const bool groupMotion = dbcGroupMotion(); // AST_NULL: Dirichlet groups / scripted components (:1413-1462)
    buildTargetPositions();
    completedStep = 1.0;
    if (nHandles || groupMotion) {
        double stepSize = filterStepSize(d_searchDir.p, 1.0);
        if (selfCollision) // CCD of the scripted motion with slackness 0.5 (AnimScripter.cpp:2158-2171)
            stepSize = fullCcd(0.5, stepSize);
        HIP_CHECK(hipMemcpyAsync(d_x0.p, mesh.d_x.p, 3 * (size_t)mesh.nV * sizeof(double), hipMemcpyDeviceToDevice, stream));
        stepForward(d_x0.p, stepSize);
        while (!checkInversion()) {
            stepSize /= 2.0;
            stepForward(d_x0.p, stepSize);
        }
        if (ipOn())
            while (anyIntersection()) {
                stepSize /= 2.0;
                stepForward(d_x0.p, stepSize);
            }
        if (stepSize < 1.0) dbcIncomplete++; // the penalty solve of newtonIter() takes the nodes the rest of the way
        completedStep = stepSize; // AnimScripter::getCompletedStepSize
        d_searchDir.zero(stream); // initX(0), Optimizer.cpp:930-934
    }
    if (warmStart >= 1 && warmStart <= 5) {
        // initX options 1-4 (Optimizer.cpp:936-1080): explicit Euler / xHat / symplectic Euler / uniformly accelerated motion as the
        // first iterate, then the feasibility filters of a Newton step with "always full CCD" (:1117-1215)
        if (warmStart == 5) {
            // option 5 (:1082-1110), "Jacobi": -g_i / H_ii with the gradient of the projected and the matrix of the UNprojected Dirichlet rows,
            // zero on every Dirichlet node; sets, kappa and dHat are whatever the previous time step left
            computeGradient(true);
            computePrecondMtr(false, false);
            launch_negate(3 * mesh.nV, d_gradient.p, d_minusG.p, stream);
            completeMatrix();
            lin.precondition_diag(d_minusG.p, d_searchDir.p);
            launch_clear_projected(mesh.nV, mesh.d_dbc.p, 1, d_searchDir.p, stream); // isDBCVertex: ZERO and NONZERO alike
        }
        else {
        static const double CG[2][5] = { { 0, 0, 1, 1, 1 }, { 0, 0, 0.5, 0.5, 0.5 } }, CE[2][5] = { { 0, 0, 0, 1, 0.5 }, { 0, 0, 0, 2, 1 } };
        const double cg = CG[timeIntegration][warmStart], ce = CE[timeIntegration][warmStart];
        const double g3[3] = { cg * (dtSq * gravity[0]), cg * (dtSq * gravity[1]), cg * (dtSq * gravity[2]) };
        launch_warm_dir(mesh.nV, mesh.d_dbc.p, d_vel.p, d_dxElastic.p, dt, g3, ce, d_searchDir.p, stream);
        }
        double stepSize = filterStepSize(d_searchDir.p, 1.0);
        if (ipOn()) {
            for (auto& h : planes) stepSize = h->stepBound(contact->nSVI, contact->d_SVI.p, mesh.d_x.p, mesh.d_dbc.p, d_searchDir.p, 0.9, stepSize);
            if (selfCollision) stepSize = fullCcd(0.8, stepSize);
        }
        HIP_CHECK(hipMemcpyAsync(d_x0.p, mesh.d_x.p, 3 * (size_t)mesh.nV * sizeof(double), hipMemcpyDeviceToDevice, stream));
        stepForward(d_x0.p, stepSize);
        while (!checkInversion()) {
            stepSize /= 2.0;
            stepForward(d_x0.p, stepSize);
        }
        if (ipOn())
            while (anyIntersection()) {
                stepSize /= 2.0;
                stepForward(d_x0.p, stepSize);
            }
        warmStepSize = stepSize;
    }
    if (ipOn()) {
        // fullyImplicit_IP head (Optimizer.cpp:1534-1550, 2316-2322): dHat, constraint sets, kappa, empty close-pair list
        dHat = dHatEps * dHatEps * lenScale2();
        computeConstraintSets();
        // tuning[0] when the script gives one, bounded from above; 0 -> suggestKappa (Optimizer.cpp:1540-1547)
        kappa = kappaConfig > 0.0 ? std::min(kappaConfig, 100 * kappaFloor()) : kappaFloor();
        initKappa();
        closeHS.clear();
        closeHSVal.clear();
        closeID.clear();
        closeVal.clear();
        // friction: lagged sets reset, eps_v^2 h^2 (Optimizer.cpp:1525-1533, 286-304), then lagged at x^n (:1553-1600)
        if (contact) contact->frictionLagClear();
        for (auto& h : planes) h->lagClear();
        fricDHat0 = epsV * epsV * (ctorDt * ctorDt) * lenScale2(); // Optimizer.cpp:290-303: set once in the constructor, see (ctorDt * ctorDt)
        fricDHatTarget = epsVTarget > 0.0 ? epsVTarget * epsVTarget * (ctorDt * ctorDt) * lenScale2() : fricDHat0;
        fricDHat = solveFric() ? fricDHat0 : -1.0;
        fricIterI = 0;
        updateFrictionLag();
    }
    initSubProblem();
    lastEnergyVal = computeEnergyVal(); // Optimizer.cpp:1609
    k = 0;
}

void HipOptimizer::initSubProblem()
{
    specAsmValid = false;
    projDBC = true;
    rhoDBC = 0.0;
    lastMove = completedStep;
}

// targetPos / dist2Tol of stepAnimScript (AnimScripter.cpp:2150-2157).  Every scripted node is a Dirichlet node here (twist
// handles, Dirichlet groups, scripted components), so the keys are the Dirichlet nodes.  Per time step, a few KB.
void HipOptimizer::buildTargetPositions(bool deferTolerance)
{
    tpIds.clear();
    for (int v = 0; v < mesh.nV; ++v)
        if (mesh.dbcType[v] != 0) tpIds.push_back(v);
    dist2Tol = 0.0;
    tolPending = false;
    if (tpIds.empty()) return;
    const int n = (int)tpIds.size();
    if (tpIds != tpIdsOnDevice) { // (the same nodes step after step unless a Dirichlet group starts or ends)
        d_tpIds.uploadGrow(tpIds, stream);
        tpIdsOnDevice = tpIds;
    }
    d_tpPos.ensure(3 * (size_t)n);
    d_tpLam.ensure(3 * (size_t)n);
    d_tpLam.zeroN(3 * (size_t)n, stream);
    // targets x + p formed on the device (round 6: x and p of the scripted nodes used to travel to the host, their sum back); the host only needs p for the tolerance,
    // summed there in index order as before.  deferTolerance: p goes to pinned host memory and finishTolerance() sums it behind the caller's next synchronisation
    // (through a device buffer and ONE copy into pinned memory: thousands of 8-byte stores of a kernel into mapped host memory are thousands of bus transactions --
    // 7 ms for the Dirichlet nodes of 4_rodsTwist when this was first written that way)
    if (h_tpStage.n < 3 * (size_t)n) h_tpStage.alloc(3 * (size_t)n + 3 * (size_t)n / 2 + 16);
    d_tpStage.ensure(3 * (size_t)n);
    launch_target_positions(n, d_tpIds.p, mesh.d_x.p, d_searchDir.p, d_tpPos.p, d_tpStage.p, stream);
    HIP_CHECK(hipMemcpyAsync(h_tpStage.p, d_tpStage.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    tolPending = true;
    if (!deferTolerance) {
        HIP_CHECK(hipStreamSynchronize(stream));
        finishTolerance();
    }
}

void HipOptimizer::finishTolerance()
{
    if (!tolPending) return;
    tolPending = false;
    double sq = 0.0;
    for (size_t i = 0, m = 3 * tpIds.size(); i < m; ++i) sq += h_tpStage.p[i] * h_tpStage.p[i];
    dist2Tol = sq * 1.0e-6;
}

double HipOptimizer::computeCompletedStepSize()
{
    if (dist2Tol == 0.0 || tpIds.empty()) return completedStep = 1.0;
    launch_mdbc_reduce(mdbc(), mesh.d_x.p, 0.0, 1, d_scalar.p + 5, stream);
    const double sqNorm = readScalar(d_scalar.p + 5);
    return completedStep = 1.0 - std::sqrt(sqNorm / (dist2Tol * 1.0e6));
}

// after postLineSearch (Optimizer.cpp:2168-2203)
void HipOptimizer::dirichletPenaltyUpdate()
{
    if (projDBC) {
        if (completedStep < 1.0 - 1.0e-3) { // setup penalty solve
            projDBC = false;
            rhoDBC = 1.0e6;
        }
        return;
    }
    const double completed = computeCompletedStepSize();
    if (completed > 1.0 - 1.0e-3) projDBC = true; // penalty solve finished
    else if (completed < lastMove && rhoDBC < 1.0e8) rhoDBC *= 2.0;
    else {
        launch_fill(d_scalar.p + 3, 1, 0.0, stream);
        launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
        if (readScalar(d_scalar.p + 3) < CN_MBC) { // safeToPull
            if (completed < 0.99 && rhoDBC < 1.0e8) rhoDBC *= 2.0;
            else launch_mdbc_lambda(mdbc(), mesh.d_x.p, rhoDBC, stream); // updateLambda (AnimScripter.cpp:2339-2346)
        }
    }
}

bool HipOptimizer::newtonIter()
{
    // convergence test (Optimizer.cpp:1869-1879) looks at the search direction of the previous pass
    double distToOpt_PN;
    if (k && cachedDistValid) distToOpt_PN = cachedDist; // read back with the last solve
    else if (k) {
        launch_fill(d_scalar.p + 3, 1, 0.0, stream);
        launch_max_abs(3 * mesh.nV, d_searchDir.p, d_scalar.p + 3, stream);
        distToOpt_PN = readScalar(d_scalar.p + 3);
    }
    else distToOpt_PN = 0.0; // (the test needs k > 0: nothing to measure in the first pass of a time step)
    cachedDistValid = false;
    if (k && distToOpt_PN < targetGRes && completedStep > 1.0 - 1.0e-3) { // :1874-1879
        specAsmValid = false; // (the one assembly per time step that goes unused)
        Tic t(timers[12], stream);
        t.nosync = fastPath(); // (nobody on the host waits for this gradient: whoever reads it synchronises)
        computeGradient(projDBC); // the reference leaves the gradient of the converged state behind (:1861)
        return true;
    }
    innerIterAmt++;
    if (fastPath()) {
        // the assembly is timed with events: no host synchronisation between it and the factorisation
        if (!evAsm0) {
            HIP_CHECK(hipEventCreate(&evAsm0));
            HIP_CHECK(hipEventCreate(&evAsm1));
        }
        if (specAsmValid && projDBC && d_aSpec.n == lin.d_a.n && d_gradSpec.n == d_gradient.n) {
            // the assembly of exactly this state is already there (enqueued behind the last pass's trial, timed by its own events): swap it in
            std::swap(lin.d_a.p, d_aSpec.p);
            std::swap(d_gradient.p, d_gradSpec.p);
            matrixComplete = true;
        }
        else {
            resolveEventTimers();
            HIP_CHECK(hipEventRecord(evAsm0, stream));
            computePrecondMtr(projDBC, true);
            HIP_CHECK(hipEventRecord(evAsm1, stream));
            evAsmPending = true;
            evAsmWeight = 1; // (an assembly of its own: the first iteration of a time step, a rejected trial)
        }
        specAsmValid = false;
    }
    else {
        // gradient (:1861) and Hessian (:2327) come out of one fused element pass
        specAsmValid = false; // an assembly enqueued ahead by an earlier fast-path pass describes a state this branch has left behind
        Tic t(timers[0], stream);
        computePrecondMtr(projDBC, true);
    }
    cachedE0Valid = false;
    computeSearchDir(projDBC);
    double alpha = 1.0;
    {
        Tic t(timers[13], stream);
        t.nosync = specAsmValid && cachedTrialValid; // nothing is enqueued in here on that path, and the stream carries the assembly enqueued ahead
        // (one process, self-contact, no half-spaces between the filter and the CCD: the three bounds come back with one synchronisation, HipContact::stepBounds)
        const bool batchedBounds = !cachedTrialValid && !cachedE0Valid && selfCollision && planes.empty() && worldSize == 1;
        if (cachedTrialValid) alpha = cachedAlpha; // decided on the device by the same rule, the trial step is already taken with it
        else if (cachedE0Valid) { // Optimizer.cpp:1887 with the value the solve's batch brought back
            if (mesh.energyType != 1 && cachedFilter > 0.0 && cachedFilter < alpha) alpha = cachedFilter;
        }
        else if (!batchedBounds)
        alpha = filterStepSize(d_searchDir.p, alpha); // Optimizer.cpp:1887
        for (auto& h : planes) // slackness_a = 0.9 (:1886-1890)
            alpha = h->stepBound(contact->nSVI, contact->d_SVI.p, mesh.d_x.p, mesh.d_dbc.p, d_searchDir.p, 0.9, alpha);
        if (selfCollision) {
            // step-size pipeline of Optimizer.cpp:1884-2040: partial CCD over the candidates of the current
            // constraint set, CFL bound, full CCD only when the step leaves the CFL ball
            const double slackness_m = 0.8;
            double pMax;
            if (batchedBounds) {
                // inversion filter, partial CCD and surface speed in one batch: the filter's root stays on the device and bounds the CCD there
                const bool filter = mesh.energyType != 1; // Energy.cpp:567 needElemInvSafeGuard
                if (filter) {
                    launch_fill(d_scalar.p + 2, 1, 1e20, stream);
                    launch_inversion_step(view(), d_searchDir.p, 0.2, alpha, d_scalar.p + 2, stream);
                }
                contact->stepBounds(mesh.d_x.p, d_searchDir.p, slackness_m, alpha, filter ? d_scalar.p + 2 : nullptr, &alpha, lastCCDPair, &pMax);
            }
            else {
                alpha = contact->ccdPartial(mesh.d_x.p, d_searchDir.p, slackness_m, alpha, lastCCDPair);
                pMax = contact->maxSurfaceSpeed(d_searchDir.p);
            }
            const double alpha_CFL = std::sqrt(dHat) / (pMax * 2.0);
            if ((!k && alpha > alpha_CFL) || alpha > 2.0 * alpha_CFL) {
                alpha = fullCcd(slackness_m, alpha);
                nFullCCD++;
                if (alpha < alpha_CFL) alpha = alpha_CFL;
            }
            else alpha = std::min(alpha, alpha_CFL);
        }
    }
    lastAlphaFeasible = alpha;
    lineSearch(alpha);
    lastStepSize = alpha;
    postLineSearch();
    dirichletPenaltyUpdate();
    ++k;
    return false;
}

void HipOptimizer::endTimestep()
{
    specAsmValid = false;
    Tic t(timers[11], stream);
    t.nosync = fastPath(); // (the update is one kernel on the stream; the next time step's batch synchronises)
    if (timeIntegration == 1)
        launch_nm_update(mesh.nV, mesh.d_dbc.p, mesh.d_x.p, d_xPrev.p, d_vel.p, d_acc.p, d_dxElastic.p, mesh.d_xTilde.p, dt, betaNM, gammaNM,
            gravity[0], gravity[1], gravity[2], stream);
    else
        launch_be_update(mesh.nV, mesh.d_dbc.p, mesh.d_x.p, d_xPrev.p, d_vel.p, d_acc.p, d_dxElastic.p, mesh.d_xTilde.p, dt, gravity[0],
            gravity[1], gravity[2], stream);
    computeDampingMtr(); // Optimizer.cpp:593-595
    globalIterNum++;
}

int HipOptimizer::solveTimestep(int maxIter)
{
    beginTimestep();
    int it = 0;
    for (;;) {
        while (it < maxIter) {
            if (newtonIter()) break;
            ++it;
        }
        if (it >= maxIter || !nextSubproblem()) break; // friction lagging iterations (fricIterAmt)
    }
    endTimestep();
    return it;
}

} // namespace ipcgpu
