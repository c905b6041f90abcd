// HipOptimizer -- the Optimizer<3> subclass a maintainer drops into src/TimeStepper/ to run the Newton time-step loop of
// ipc-sim/IPC on an MI355X.  Optimizer<dim> is all-virtual (src/TimeStepper/Optimizer.hpp:131-283) and constructed at exactly
// one site, src/main.cpp:1397; the one-line change there is `new IPC::HipOptimizer<DIM>(...)` with the same arguments.
//
// Two ways to run, chosen with the environment variable IPCGPU_OPTIMIZER_MODE (or the last constructor argument):
//
//   resident (default)   precompute() hands the state the base-class constructor assembled -- Mesh<3> with the arrays it already
//       holds, Config, the Dirichlet / Neumann groups, half-spaces, mesh collision objects, start velocity -- to the library
//       once; solve() then runs whole time steps on the device (ipcgpu_opt_solve_timestep: scripted Dirichlet motion,
//       constraint sets, barrier + friction terms, assembly, Cholesky, CCD line search, velocity update) and mirrors positions,
//       velocity, acceleration, dx_Elastic and the counters back into the base-class members, so that getResult(),
//       getIterNum(), getInnerIterAmt() and saveStatus() (Optimizer.cpp:2964-3011) keep working unchanged.  Per iteration a
//       handful of scalars cross PCIe.
//   percall              the reference's own solve() / fullyImplicit_IP() / solveSub_IP() / lineSearch() control flow stays in
//       charge; the elasticity term it evaluates is a HipElasticEnergy and the solver LinSysSolver::create hands it is a
//       HipLinSysSolver on the same context (values in HBM).  Contact terms, CCD and the scripts run as the reference's host
//       code.  This is the A/B configuration inside one binary, and it supports every script / collision object the reference has.
//
// Scripts (AnimScripter.hpp:22-95) that run resident are the ones residentScript() below accepts: null (Dirichlet / Neumann groups, scripted component
// velocities), twist, fall, fallNoShift, dragright, DCOFix, stretchAndPause, DCOSquash, DCOSquash6, DCORotCylinders, DCOVerschoorRoller, DCOSqueezeOut, the
// handle sets that are held for good (staticScript) and the ones pulled at a constant velocity (pulledScript).  Any other one falls back to percall with a note
// on stderr.  Not reproduced in resident mode (side effects of Optimizer::solve, Optimizer.cpp:517-530): the pre-step inversion check that ends the process,
// and saveBCNodes.
// Needs: the reference's Optimizer.hpp, include/ipcgpu.h, -lipcgpu.
#pragma once
#include "Optimizer.hpp"
#include "HalfSpace.hpp"
#include "HipElasticEnergy.hpp"
#include "HipSelfCollisionHandler.hpp" // the registry and the class; the NAME is redirected where Optimizer.cpp is compiled (HipSelfCollisionHandlerRedirect.hpp), not here
#include <ipcgpu.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace IPC {

enum HipOptimizerMode {
    HIP_OPT_RESIDENT = 0,
    HIP_OPT_PERCALL = 1,
    HIP_OPT_FROM_ENV = -1
};

// LinSysSolver::create is a static factory without arguments (LinSysSolver.cpp:13-27): the optimizer adapter leaves its
// context here right before the base-class constructor asks for `linSysSolver`; the `case LinSysSolverType::HIP:` a maintainer
// adds to the factory is `return hipCreateLinSysSolver<vectorTypeI, vectorTypeS>();`.  The first solver created after an
// optimizer adapter shares its context (the Hessian), every other one owns a private context (the damping matrix, Diagnostic.cpp).
struct HipAdapterRegistry {
    ipcgpu_ctx* pending = nullptr;
};
inline HipAdapterRegistry& hipAdapterRegistry()
{
    static HipAdapterRegistry r;
    return r;
}
template <typename vectorTypeI, typename vectorTypeS>
inline LinSysSolver<vectorTypeI, vectorTypeS>* hipCreateLinSysSolver()
{
    ipcgpu_ctx* shared = hipAdapterRegistry().pending;
    hipAdapterRegistry().pending = nullptr;
    return new HipLinSysSolver<vectorTypeI, vectorTypeS>(shared);
}

// HalfSpace<3>::normal is protected and has no getter (HalfSpace.hpp:21-22); a pointer to member named through a derived class
// is the access the language allows without touching the reference (a maintainer adds `getNormal()` instead).
struct HipHalfSpaceAccess : HalfSpace<3> {
    static const Eigen::Matrix<double, 3, 1>& normalOf(const HalfSpace<3>& h) { return h.*(&HipHalfSpaceAccess::normal); }
};

// members that must exist before the Optimizer<3> base is constructed (it keeps a reference to the energy terms and asks
// LinSysSolver::create for its solver inside its constructor)
struct HipOptimizerParts {
    ipcgpu_ctx* ctx = nullptr;
    int mode = HIP_OPT_RESIDENT;
    std::vector<Energy<3>*> terms;
    std::unique_ptr<HipElasticEnergy> hipEnergy;

    static void chk(int rc)
    {
        if (rc < 0) throw std::runtime_error(ipcgpu_last_error());
    }
    static bool residentScript(AnimScriptType t)
    {
        return t == AST_NULL || t == AST_TWIST || t == AST_FALL || t == AST_FALL_NOSHIFT || t == AST_DRAGRIGHT || t == AST_DCOFIX || t == AST_STRETCHNPAUSE
            || t == AST_DCOSQUASH || t == AST_DCOSQUASH6 || t == AST_DCOROTCYLINDERS || t == AST_DCOVERSCHOORROLLER || t == AST_DCOSQUEEZEOUT
            || staticScript(t) || pulledScript(t);
    }
    // handle sets that move at a constant velocity for good (AnimScripter.cpp:459-473, 502-516, 790-807, 860-878, 895-910; per step :1597-1603, 1820-1826)
    static bool pulledScript(AnimScriptType t) { return t == AST_STRETCH || t == AST_SQUASH || t == AST_DRAGDOWN || t == AST_CURTAIN || t == AST_PUSHRIGHTMOST1; }
    // scripts whose whole effect is decided in AnimScripter::initAnimScript / initVelocity (run by the base-class constructor): node sets that are
    // held for good (ZERO or NONZERO without a velocity), changed start positions, a Neumann group, start velocities -- stepAnimScript does nothing
    // for them (AnimScripter.cpp:1536-1551, 1814-1817, 1828-1830)
    static bool staticScript(AnimScriptType t)
    {
        return t == AST_HANG || t == AST_HANG2 || t == AST_HANGTOPLEFT || t == AST_HANGLEFT || t == AST_SWING || t == AST_STAMP || t == AST_STAMPTOPLEFT
            || t == AST_STAMPBOTH || t == AST_STAMPINV || t == AST_STAND || t == AST_STANDINV || t == AST_TOPBOTTOMFIX || t == AST_FIXLOWERHALF
            || t == AST_CORNER || t == AST_SCALEF || t == AST_FIXRIGHTMOST1 || t == AST_LEFTHITRIGHT || t == AST_DROP || t == AST_XYROTATE
            || t == AST_NMFIXBOTTOMDRAGLEFT || t == AST_NMFIXBOTTOMDRAGFORWARD;
    }
    HipOptimizerParts(const std::vector<Energy<3>*>& given, const Config& cfg, int requested, int device)
    {
        mode = requested;
        if (mode == HIP_OPT_FROM_ENV) {
            const char* e = std::getenv("IPCGPU_OPTIMIZER_MODE");
            mode = (e && std::strcmp(e, "percall") == 0) ? HIP_OPT_PERCALL : HIP_OPT_RESIDENT;
        }
        if (mode == HIP_OPT_RESIDENT) {
            const char* why = nullptr;
            if (!residentScript(cfg.animScriptType)) why = "this script is not built into the resident stepper";
            else if (cfg.ccdMethod != ccd::CCDMethod::FLOATING_POINT_ROOT_FINDER) why = "a CCD back end other than the default (the library builds the step bound of FloatingPointRootFinder only)";
            else if (!cfg.inputShapeMeshSeqFolderPath.empty()) why = "a mesh sequence (ipcgpu_opt_set_dirichlet_targets is driven by the scene tooling only)";
            else if (cfg.isConstrained && cfg.constraintSolverType != CST_IP) why = "a constraint solver other than interiorPoint";
            else if (!cfg.isConstrained && (cfg.collisionObjects.size() || cfg.meshCollisionObjects.size())) why = "unconstrained run with collision objects";
            if (why) {
                std::fprintf(stderr, "HipOptimizer: %s -> percall mode (elasticity + Cholesky on the device, the rest as the reference's host code)\n", why);
                mode = HIP_OPT_PERCALL;
            }
        }
        chk(ipcgpu_ctx_create(device, &ctx));
        if (mode == HIP_OPT_PERCALL) {
            hipEnergy.reset(new HipElasticEnergy(ctx, cfg.energyType == ET_FCR ? 1 : 0));
            terms.assign(1, hipEnergy.get());
            hipAdapterRegistry().pending = ctx;
        }
        else {
            terms = given;
            hipAdapterRegistry().pending = nullptr;
        }
    }
    ~HipOptimizerParts()
    {
        hipEnergy.reset();
        if (ctx) ipcgpu_ctx_destroy(ctx);
    }
    HipOptimizerParts(const HipOptimizerParts&) = delete;
    HipOptimizerParts& operator=(const HipOptimizerParts&) = delete;
};

template <int dim>
class HipOptimizer : private HipOptimizerParts, public Optimizer<dim> {
    static_assert(dim == 3, "the device path is three-dimensional");
    typedef Optimizer<dim> Base;
    typedef HipOptimizerParts Parts;

protected:
    int nSim = 0, nObst = 0; // nodes of Mesh<3>; nodes of the mesh collision objects riding along behind them
    bool uploaded = false, dragReleased = false;
    bool contactOnDevice = false; // percall mode: the SelfCollisionHandler statics forward to this context (HipSelfCollisionHandler.hpp)
    int dragGroup = -1, pauseTurn = -1, pauseGroup[2] = { -1, -1 };
    std::vector<int> pauseIds[2];
    std::vector<std::array<double, 3>> plateVel; // DCOSquash / DCOSquash6: current velocity of every plate
    std::vector<unsigned char> dbcMirror_; // vertexDBCType as last handed to the device (percall)
    std::vector<double> bufV_, bufA_, bufB_, bufC_;

    using Parts::chk;

public:
    HipOptimizer(const Mesh<dim>& p_data0, const std::vector<Energy<dim>*>& p_energyTerms, const std::vector<double>& p_energyParams,
        bool p_mute = false, const Eigen::MatrixXd& UV_bnds = Eigen::MatrixXd(), const Eigen::MatrixXi& E = Eigen::MatrixXi(),
        const Eigen::VectorXi& bnd = Eigen::VectorXi(), const Config& p_animConfig = Config(), int p_mode = HIP_OPT_FROM_ENV, int device = 0)
        : Parts(p_energyTerms, p_animConfig, p_mode, device), Base(p_data0, Parts::terms, p_energyParams, p_mute, UV_bnds, E, bnd, p_animConfig)
    {
        hipAdapterRegistry().pending = nullptr;
        nSim = (int)Base::result.V.rows();
        if (Parts::mode == HIP_OPT_PERCALL) {
            // the energy adapter evaluates on the mesh of the shared context: Mesh<3> as the base-class constructor left it
            hipUploadMesh(Parts::ctx, Base::result, Base::result.m_YM, Base::result.m_PR, Base::result.density);
            chk(ipcgpu_opt_init(Parts::ctx, Base::dt, p_animConfig.withGravity ? 1 : 0)); // work buffers of the element kernels; the time step itself stays the base class's
            mirrorDBC();
            // self-contact on the device in this mode too (round 5): the statics of HipSelfCollisionHandler.hpp forward the constraint sets, the
            // per-constraint distances / Jacobian products, the barrier Hessian, both CCD step bounds and the intersection test of the reference's own
            // control flow to this context -- once it has the surface.  The mesh collision objects, the analytic planes, friction and the mollified
            // pairs stay the reference's host code.  IPCGPU_PERCALL_CONTACT=host: everything on the host, as before round 5 (A/B).
            const char* pc = std::getenv("IPCGPU_PERCALL_CONTACT");
            const bool hostContact = pc && std::strcmp(pc, "host") == 0;
            if (Base::solveIP && p_animConfig.isSelfCollision && !hostContact && p_animConfig.ccdMethod == ccd::CCDMethod::FLOATING_POINT_ROOT_FINDER) {
                const Mesh<dim>& m = Base::result;
                std::vector<int> ce(2 * (size_t)m.CE.rows());
                for (int e = 0; e < (int)m.CE.rows(); ++e) {
                    ce[2 * (size_t)e] = m.CE(e, 0);
                    ce[2 * (size_t)e + 1] = m.CE(e, 1);
                }
                chk(ipcgpu_set_surface_codim(Parts::ctx, (int)m.SF.rows(), m.SF.data(), (int)m.CE.rows(), ce.empty() ? nullptr : ce.data()));
#ifdef USE_PREDICATES
                chk(ipcgpu_set_exact_predicates(Parts::ctx, 1));
#endif
                hipCollisionRegistry().ctx = Parts::ctx;
                contactOnDevice = true;
                static bool registered = false; // the reference's main() leaves through exit() without deleting its optimizer: the summary is printed from there
                if (!registered) {
                    std::atexit(reportForwardedCalls);
                    registered = true;
                }
            }
        }
        std::fprintf(stderr, "HipOptimizer: %s mode%s\n", resident() ? "resident (whole time steps on the device)" : "percall (the reference's control flow)",
            resident() ? "" : (contactOnDevice ? "; elasticity, Cholesky and self-contact on the device" : "; elasticity and Cholesky on the device, contact on the host"));
    }
    ~HipOptimizer() override
    {
        if (contactOnDevice && hipCollisionRegistry().ctx == Parts::ctx) {
            reportForwardedCalls();
            hipCollisionRegistry().ctx = nullptr; // the context goes away with this object: the handler's statics are the reference's again
        }
    }
    // where the self-contact of a percall run was evaluated (stderr, once: from the destructor or, when the program leaves through exit(), from atexit)
    static void reportForwardedCalls()
    {
        static bool done = false;
        if (done) return;
        done = true;
        const HipCollisionRegistry& r = hipCollisionRegistry();
        std::fprintf(stderr, "HipOptimizer: self-contact calls forwarded to the device: %lld constraint sets, %lld evaluations, %lld Jacobian products, %lld barrier "
                             "Hessians, %lld + %lld step bounds, %lld intersection tests (%lld Hessians fell back to the host)\n",
            r.calls[0], r.calls[1], r.calls[2], r.calls[3], r.calls[4], r.calls[5], r.calls[6], r.hostFallbacks);
    }

    ipcgpu_ctx* context() const { return Parts::ctx; }
    bool resident() const { return Parts::mode == HIP_OPT_RESIDENT; }

    // ---- Optimizer API (Optimizer.hpp:131-161) --------------------------------------------------------------------------
    void setRelGL2Tol(double p_relTol = 1.0e-2) override
    {
        Base::setRelGL2Tol(p_relTol);
        if (resident() && uploaded) chk(ipcgpu_opt_set_rel_tol(Parts::ctx, p_relTol));
    }

    void precompute(void) override
    {
        if (!resident()) {
            Base::precompute();
            return;
        }
        uploadScene(); // after setTime (main.cpp:1398): dt is final
        chk(ipcgpu_opt_set_rel_tol(Parts::ctx, std::sqrt(Base::relGL2Tol)));
        chk(ipcgpu_opt_precompute(Parts::ctx));
        double sc[8];
        chk(ipcgpu_opt_get_state(Parts::ctx, nullptr, nullptr, nullptr, sc));
        Base::lastEnergyVal = sc[0];
    }

    // Optimizer::solve (Optimizer.cpp:509-602): maxIter time steps; 1 = the animation is over, 0 otherwise
    int solve(int maxIter = 100) override
    {
        if (!resident()) return Base::solve(maxIter);
        for (int iterI = 0; iterI < maxIter; ++iterI) {
            if (Base::globalIterNum >= Base::frameAmt) return 1;
            if (Base::animConfig.animScriptType == AST_DRAGRIGHT) dragRightRule();
            if (Base::animConfig.animScriptType == AST_STRETCHNPAUSE) stretchPauseRule();
            if (Base::animConfig.animScriptType == AST_DCOSQUASH || Base::animConfig.animScriptType == AST_DCOSQUASH6) squashRule();
            int n = 0;
            chk(ipcgpu_opt_solve_timestep(Parts::ctx, 1 << 30, &n));
            Base::innerIterAmt += n;
            Base::globalIterNum++;
            mirrorState();
        }
        return 0;
    }

    // ---- percall: the reference's control flow, device elasticity + solver ---------------------------------------------------
    // the Dirichlet types change under stepAnimScript (time ranges, released handles): hand them over before the next evaluation
    void computeEnergyVal(const Mesh<dim>& data, int redoSVD, double& energyVal) override
    {
        if (!resident()) mirrorDBC(&data);
        Base::computeEnergyVal(data, redoSVD, energyVal);
    }
    void computeGradient(const Mesh<dim>& data, bool redoSVD, Eigen::VectorXd& gradient, bool projectDBC = true) override
    {
        if (!resident()) mirrorDBC(&data);
        Base::computeGradient(data, redoSVD, gradient, projectDBC);
    }
    void computePrecondMtr(const Mesh<dim>& data, bool redoSVD, LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>* p_linSysSolver,
        bool updateDamping = false, bool projectDBC = true) override
    {
        if (!resident()) mirrorDBC(&data);
        Base::computePrecondMtr(data, redoSVD, p_linSysSolver, updateDamping, projectDBC);
    }

protected:
    void mirrorDBC(const Mesh<dim>* data = nullptr)
    {
        const Mesh<dim>& m = data ? *data : Base::result;
        const size_t n = m.vertexDBCType.size();
        bool same = dbcMirror_.size() == n;
        for (size_t v = 0; same && v < n; ++v) same = dbcMirror_[v] == (unsigned char)m.vertexDBCType[v];
        if (same) return;
        dbcMirror_.resize(n);
        std::vector<int> ids[3];
        for (size_t v = 0; v < n; ++v) {
            dbcMirror_[v] = (unsigned char)m.vertexDBCType[v];
            ids[dbcMirror_[v]].push_back((int)v);
        }
        chk(ipcgpu_clear_dbc(Parts::ctx));
        for (int type = 1; type <= 2; ++type)
            if (!ids[type].empty()) chk(ipcgpu_set_dbc(Parts::ctx, (int)ids[type].size(), ids[type].data(), type));
    }

    // `script dragright` (AnimScripter.cpp:1619-1632): the handle is let go once the whole body is right of every mesh obstacle
    void dragRightRule()
    {
        if (dragReleased || dragGroup < 0) return;
        double rightMost = -1.0e300;
        for (const auto& mco : Base::animConfig.meshCollisionObjects)
            for (int i = 0; i < mco->V.rows(); ++i) rightMost = std::max(rightMost, mco->V(i, 0));
        double left = 1.0e300;
        for (int v = 0; v < nSim; ++v) left = std::min(left, Base::result.V(v, 0));
        if (left > rightMost) {
            chk(ipcgpu_opt_end_dirichlet(Parts::ctx, dragGroup, Base::globalIterNum * Base::dt));
            dragReleased = true;
        }
    }

    // `script DCOSquash / DCOSquash6` (AnimScripter.cpp:2034-2074): while the first two plates are closer than 0.1 in x, every plate
    // velocity changes sign -- once per time step, as written
    void squashRule()
    {
        const Mesh<dim>& m = Base::result;
        double co0right = -1.0e300, co1left = 1.0e300;
        for (int v = m.componentNodeRange[0]; v < m.componentNodeRange[1]; ++v) co0right = std::max(co0right, m.V(v, 0));
        for (int v = m.componentNodeRange[1]; v < m.componentNodeRange[2]; ++v) co1left = std::min(co1left, m.V(v, 0));
        if (!(co1left - co0right < 0.1)) return;
        const double zero3[3] = { 0, 0, 0 };
        for (size_t g = 0; g < plateVel.size(); ++g) {
            for (int c = 0; c < 3; ++c) plateVel[g][c] = -plateVel[g][c];
            chk(ipcgpu_opt_set_dirichlet_motion(Parts::ctx, (int)g, plateVel[g].data(), zero3, nullptr, 1));
        }
    }

    // `script stretchAndPause` (AnimScripter.cpp:1605-1616): the handles are pulled apart while the turning vertex has not passed
    // x = -0.28; from then on every Dirichlet node is held (ZERO)
    void stretchPauseRule()
    {
        if (dragReleased || pauseTurn < 0) return;
        if (Base::result.V(pauseTurn, 0) >= -0.28) return;
        const double t = Base::globalIterNum * Base::dt, zero3[3] = { 0, 0, 0 };
        for (int g = 0; g < 2; ++g) {
            if (pauseIds[g].empty()) continue;
            chk(ipcgpu_opt_end_dirichlet(Parts::ctx, pauseGroup[g], t));
            chk(ipcgpu_opt_add_dirichlet(Parts::ctx, (int)pauseIds[g].size(), pauseIds[g].data(), zero3, zero3, t, std::numeric_limits<double>::infinity()));
        }
        dragReleased = true;
    }

    // device -> result.V / V_prev, velocity, acceleration, dx_Elastic (what saveStatus writes and main.cpp draws)
    void mirrorState()
    {
        const size_t nAll = (size_t)(nSim + nObst);
        bufV_.resize(3 * nAll);
        bufA_.resize(3 * nAll);
        bufB_.resize(3 * nAll);
        bufC_.resize(3 * nAll);
        double sc[8];
        chk(ipcgpu_opt_get_state(Parts::ctx, bufV_.data(), nullptr, nullptr, sc));
        chk(ipcgpu_opt_get_kinematics(Parts::ctx, bufA_.data(), bufB_.data(), bufC_.data()));
        for (int v = 0; v < nSim; ++v)
            for (int c = 0; c < 3; ++c) {
                Base::result.V(v, c) = bufV_[(size_t)c * nAll + v];
                Base::velocity[3 * v + c] = bufA_[3 * (size_t)v + c];
                Base::acceleration(v, c) = bufB_[3 * (size_t)v + c];
                Base::dx_Elastic(v, c) = bufC_[3 * (size_t)v + c];
            }
        Base::result.V_prev = Base::result.V;
        Base::lastEnergyVal = sc[0];
        Base::kappa = sc[6];
        Base::dHat = sc[7];
    }

    // the scene as the base-class constructor assembled it -> one context (what ipc_amd/scene_script.py::apply does from a script)
    void uploadScene()
    {
        ipcgpu_ctx* ctx = Parts::ctx;
        const Config& cfg = Base::animConfig;
        const Mesh<dim>& m = Base::result;
        const int nT = (int)m.F.rows();
        // mesh collision objects ride along as surface-only components behind the nodes of Mesh<3> (MeshCO.cpp:37-80)
        nObst = 0;
        int nSFObst = 0;
        for (const auto& mco : cfg.meshCollisionObjects) {
            nObst += (int)mco->V.rows();
            nSFObst += (int)mco->F.rows();
        }
        const int nAll = nSim + nObst, nSF = (int)m.SF.rows(), nSFAll = nSF + nSFObst;
        std::vector<double> Vrest(3 * (size_t)nAll), Vcur(3 * (size_t)nAll), mass((size_t)nAll, 0.0);
        std::vector<int> SF(3 * (size_t)nSFAll), obst;
        for (int v = 0; v < nSim; ++v)
            for (int c = 0; c < 3; ++c) {
                Vrest[(size_t)c * nAll + v] = m.V_rest(v, c);
                Vcur[(size_t)c * nAll + v] = m.V(v, c);
            }
        for (int f = 0; f < nSF; ++f)
            for (int c = 0; c < 3; ++c) SF[(size_t)c * nSFAll + f] = m.SF(f, c);
        int vOff = nSim, fOff = nSF;
        for (const auto& mco : cfg.meshCollisionObjects) {
            for (int v = 0; v < mco->V.rows(); ++v) {
                for (int c = 0; c < 3; ++c) Vrest[(size_t)c * nAll + vOff + v] = Vcur[(size_t)c * nAll + vOff + v] = mco->V(v, c);
                obst.push_back(vOff + v);
            }
            for (int f = 0; f < mco->F.rows(); ++f)
                for (int c = 0; c < 3; ++c) SF[(size_t)c * nSFAll + fOff + f] = mco->F(f, c) + vOff;
            vOff += (int)mco->V.rows();
            fOff += (int)mco->F.rows();
        }
        chk(ipcgpu_set_mesh(ctx, nAll, nT, Vrest.data(), m.F.data(), m.m_YM, m.m_PR, m.density));
        // components of codimension 2 / 1 / 0 (triangle meshes, `.seg` segments, `.pt` points under `shapes`, main.cpp:948-1005): nodes of no
        // tetrahedron with the masses Mesh<3> gave them (Mesh.cpp:279-345, 405-411)
        std::vector<int> codim;
        std::vector<double> codimMass;
        for (size_t compI = 0; compI < m.componentCoDim.size(); ++compI) {
            if (m.componentCoDim[compI] == 3) continue;
            for (int v = m.componentNodeRange[compI]; v < m.componentNodeRange[compI + 1]; ++v) {
                codim.push_back(v);
                codimMass.push_back(m.massMatrix.coeff(v, v));
            }
        }
        if (!codim.empty()) chk(ipcgpu_set_codim_nodes(ctx, (int)codim.size(), codim.data(), codimMass.data()));
        // the arrays Mesh<3> already holds (component materials and densities included, Mesh.cpp:660-671)
        {
            std::vector<double> A(9 * (size_t)nT);
            for (int t = 0; t < nT; ++t)
                for (int k = 0; k < 9; ++k) A[9 * (size_t)t + k] = m.restTriInv[t].data()[k];
            for (int v = 0; v < nSim; ++v) mass[v] = m.massMatrix.coeff(v, v);
            chk(ipcgpu_set_mesh_features(ctx, A.data(), m.triArea.data(), mass.data(), m.u.data(), m.lambda.data()));
        }
        chk(ipcgpu_set_energy_type(ctx, cfg.energyType == ET_FCR ? 1 : 0));
        chk(ipcgpu_set_positions(ctx, Vcur.data()));
        chk(ipcgpu_opt_init(ctx, Base::dt, cfg.withGravity ? 1 : 0));
        if (cfg.timeIntegrationType == TIT_NM) chk(ipcgpu_opt_set_time_integration(ctx, 1, Base::beta_NM, Base::gamma_NM));
        {
            // Mesh<3>::CE, the segments of `.seg` shapes; `.pt` points need nothing: nodes without a neighbour are found by the library
            std::vector<int> ce(2 * (size_t)m.CE.rows());
            for (int e = 0; e < (int)m.CE.rows(); ++e) {
                ce[2 * (size_t)e] = m.CE(e, 0);
                ce[2 * (size_t)e + 1] = m.CE(e, 1);
            }
            chk(ipcgpu_set_surface_codim(ctx, nSFAll, SF.data(), (int)m.CE.rows(), ce.empty() ? nullptr : ce.data()));
#ifdef USE_PREDICATES
            chk(ipcgpu_set_exact_predicates(ctx, 1)); // this build of the reference tests plane sides with igl::predicates::orient3d
#endif
        }
        // static Dirichlet types: held surfaces; the obstacle
        chk(ipcgpu_clear_dbc(ctx));
        if (!codim.empty() && (cfg.animScriptType == AST_NULL || cfg.animScriptType == AST_DCOFIX))
            chk(ipcgpu_set_dbc(ctx, (int)codim.size(), codim.data(), cfg.animScriptType == AST_DCOFIX ? IPCGPU_DBC_NONZERO : IPCGPU_DBC_ZERO));
        const bool selfCollision = Base::solveIP && cfg.isSelfCollision;
        if (!obst.empty()) {
            chk(ipcgpu_set_dbc(ctx, (int)obst.size(), obst.data(), IPCGPU_DBC_ZERO));
            chk(ipcgpu_set_obstacle_nodes(ctx, (int)obst.size(), obst.data(), selfCollision ? 0 : 1));
        }
        if (Base::solveIP && (selfCollision || !obst.empty())) chk(ipcgpu_opt_enable_self_collision(ctx, Base::dHatEps));
        // analytic half-spaces (`ground`, `halfSpace`: Config.cpp:306-345)
        bool planeFric = false;
        if (Base::solveIP)
            for (const auto& co : cfg.collisionObjects) {
                const HalfSpace<3>* hs = dynamic_cast<const HalfSpace<3>*>(co.get());
                if (!hs) throw std::runtime_error("HipOptimizer: an analytic collision object that is not a half-space");
                const Eigen::Matrix<double, 3, 1>& nrm = HipHalfSpaceAccess::normalOf(*hs);
                const double o[3] = { hs->origin[0], hs->origin[1], hs->origin[2] }, n[3] = { nrm[0], nrm[1], nrm[2] };
                int id = -1;
                chk(ipcgpu_opt_add_half_space(ctx, o, n, Base::dHatEps, &id));
                if (hs->friction > 0.0) {
                    chk(ipcgpu_opt_set_half_space_friction(ctx, id, hs->friction));
                    planeFric = true;
                }
            }
        // friction (Optimizer.cpp:146-166): mesh collision objects only switch the lagging loop on, their pairs carry none
        // (:3357-3376, 3473-3510, 3676-3705 evaluate friction for the analytic objects and the self-collision set only)
        const double epsV = cfg.tuning.size() > 4 ? cfg.tuning[4] : 1.0e-3;
        const double selfFric = selfCollision ? cfg.selfFric : 0.0;
        if (Base::solveFric) {
            chk(ipcgpu_opt_set_friction(ctx, selfFric, cfg.fricIterAmt, epsV));
            const double epsVTarget = cfg.tuning.size() > 5 ? cfg.tuning[5] : 1.0e-3; // Optimizer.cpp:296-299
            if (epsVTarget > 0.0 && epsVTarget != epsV) chk(ipcgpu_opt_set_friction_target(ctx, epsVTarget));
            if (selfFric > 0.0 && !obst.empty()) chk(ipcgpu_opt_set_friction_scales(ctx, 1.0, 0.0));
            if (!(selfFric > 0.0) && !planeFric) chk(ipcgpu_opt_force_friction_loop(ctx, 1));
        }
        // useAbsParameters, tuning[3], kappaMinMultiplier (Config.cpp:553-558; Optimizer.cpp:102-109, 279-302, 2228-2233, 2941-2945)
        chk(ipcgpu_opt_set_parameter_scaling(ctx, cfg.useAbsParameters ? 1 : 0, cfg.tuning.size() > 3 ? cfg.tuning[3] : 1.0e-9, cfg.kappaMinMultiplier));
        if (cfg.tuning.size() > 0 && cfg.tuning[0] > 0.0) chk(ipcgpu_opt_set_kappa(ctx, cfg.tuning[0]));
        {
            const double target = cfg.tuning.size() > 2 ? cfg.tuning[2] : 1.0e-3; // Optimizer.cpp:283-289
            if (target > 0.0 && target < Base::dHatEps) chk(ipcgpu_opt_set_dhat_target(ctx, target));
        }
        if (cfg.dampingStiff > 0.0) chk(ipcgpu_opt_set_damping(ctx, cfg.dampingStiff));
        // scripted motion
        const double zero3[3] = { 0, 0, 0 };
        const double inf = std::numeric_limits<double>::infinity();
        if (cfg.animScriptType == AST_NULL) {
            // whole components with a scripted velocity (AnimScripter.cpp:1413-1435), then the Dirichlet groups (:1437-1462)
            std::vector<std::array<double, 6>> comp(m.componentNodeRange.size());
            std::vector<char> has(m.componentNodeRange.size(), 0);
            for (const auto& lv : m.componentLVels) {
                has[lv.first[0]] = 1;
                for (int c = 0; c < 3; ++c) comp[lv.first[0]][c] = lv.second[c];
            }
            for (const auto& av : m.componentAVels) {
                if (!has[av.first[0]]) comp[av.first[0]] = { 0, 0, 0, 0, 0, 0 };
                has[av.first[0]] = 1;
                for (int c = 0; c < 3; ++c) comp[av.first[0]][3 + c] = av.second[c];
            }
            for (size_t compI = 0; compI + 1 < m.componentNodeRange.size(); ++compI) {
                if (!has[compI]) continue;
                std::vector<int> ids;
                for (int v = m.componentNodeRange[compI]; v < m.componentNodeRange[compI + 1]; ++v) ids.push_back(v);
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)ids.size(), ids.data(), comp[compI].data(), comp[compI].data() + 3, 0.0, inf));
            }
            for (const auto& dbc : m.DirichletBCs) {
                const double lin[3] = { dbc.linearVelocity[0], dbc.linearVelocity[1], dbc.linearVelocity[2] };
                const double ang[3] = { dbc.angularVelocity[0], dbc.angularVelocity[1], dbc.angularVelocity[2] };
                const double t0 = std::max(dbc.timeRange[0], cfg.DBCTimeRange[0]), t1 = std::min(dbc.timeRange[1], cfg.DBCTimeRange[1]);
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)dbc.vertIds.size(), dbc.vertIds.data(), lin, ang, t0, t1));
            }
        }
        else if (cfg.animScriptType == AST_TWIST) {
            // AnimScripter.cpp:555-572: the two border vertex sets turn about x with -/+ 0.4 pi
            if (m.borderVerts_primitive.size() != 2) throw std::runtime_error("HipOptimizer: script twist needs two border vertex sets");
            const std::vector<int>&l = m.borderVerts_primitive[0], &r = m.borderVerts_primitive[1];
            chk(ipcgpu_opt_set_twist(ctx, (int)l.size(), l.data(), (int)r.size(), r.data(), 0.4 * M_PI));
        }
        else if (cfg.animScriptType == AST_DRAGRIGHT) {
            // AnimScripter.cpp:809-826: the NONZERO handle the constructor picked, pulled at 0.5 in +x until the rule lets go
            std::vector<int> ids;
            for (int v = 0; v < nSim; ++v)
                if (m.vertexDBCType[v] == DirichletBCType::NONZERO) ids.push_back(v);
            const double lin[3] = { 0.5, 0.0, 0.0 };
            if (!ids.empty()) {
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)ids.size(), ids.data(), lin, zero3, 0.0, inf));
                dragGroup = 0;
            }
        }
        else if (cfg.animScriptType == AST_STRETCHNPAUSE) {
            // AnimScripter.cpp:475-500: the nodes within 1 % of the left / right end (the NONZERO nodes the constructor picked) move at
            // 1 in -x / +x; the turning vertex is the last left handle in index order
            double lo = 1.0e300, hi = -1.0e300;
            for (int v = 0; v < nSim; ++v) {
                lo = std::min(lo, m.V(v, 0));
                hi = std::max(hi, m.V(v, 0));
            }
            for (int v = 0; v < nSim; ++v) {
                if (m.V(v, 0) < lo + (hi - lo) * 0.01) {
                    pauseIds[0].push_back(v);
                    pauseTurn = v;
                }
                else if (m.V(v, 0) > hi - (hi - lo) * 0.01) pauseIds[1].push_back(v);
            }
            int group = 0;
            for (int g = 0; g < 2; ++g) {
                if (pauseIds[g].empty()) continue;
                const double lin[3] = { g == 0 ? -1.0 : 1.0, 0.0, 0.0 };
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)pauseIds[g].size(), pauseIds[g].data(), lin, zero3, 0.0, inf));
                pauseGroup[g] = group++;
            }
        }
        else if (cfg.animScriptType == AST_DCOSQUASH || cfg.animScriptType == AST_DCOSQUASH6 || cfg.animScriptType == AST_DCOROTCYLINDERS
            || cfg.animScriptType == AST_DCOVERSCHOORROLLER) {
            // whole leading components moved by the script, all their nodes NONZERO (AnimScripter.cpp:1060-1221 set-up, :1961-2074 per step)
            static const double S6[6][3] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 0, 0, -1 } };
            static const double RC[4][3] = { { M_PI / 2, 0, 0 }, { -M_PI / 2, 0, 0 }, { 0, 0, -M_PI / 2 }, { 0, 0, M_PI / 2 } };
            static const double VR[6][3] = { { 0, 0, -4 }, { 0, 0, -2 }, { 0, 0, 2 }, { 0, 0, 4 }, { 2, 0, 0 }, { -2, 0, 0 } };
            const bool rot = cfg.animScriptType == AST_DCOROTCYLINDERS || cfg.animScriptType == AST_DCOVERSCHOORROLLER;
            const int n = cfg.animScriptType == AST_DCOSQUASH ? 2 : (cfg.animScriptType == AST_DCOROTCYLINDERS ? 4 : 6);
            if ((int)m.componentNodeRange.size() < n + 2) throw std::runtime_error("HipOptimizer: the script needs more components");
            for (int compI = 0; compI < n; ++compI) {
                std::vector<int> ids;
                double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
                for (int v = m.componentNodeRange[compI]; v < m.componentNodeRange[compI + 1]; ++v) {
                    ids.push_back(v);
                    for (int c = 0; c < 3; ++c) {
                        lo[c] = std::min(lo[c], m.V(v, c));
                        hi[c] = std::max(hi[c], m.V(v, c));
                    }
                }
                const double* lin = rot ? zero3 : S6[compI];
                const double* ang = !rot ? zero3 : (cfg.animScriptType == AST_DCOROTCYLINDERS ? RC[compI] : VR[compI]);
                const double ctr[3] = { 0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]) }; // MCORotCenter: fixed at set-up
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)ids.size(), ids.data(), lin, ang, 0.0, inf));
                chk(ipcgpu_opt_set_dirichlet_motion(ctx, compI, lin, ang, rot ? ctr : nullptr, 1));
                if (!rot) plateVel.push_back({ lin[0], lin[1], lin[2] });
            }
        }
        else if (Parts::staticScript(cfg.animScriptType)) {
            // the node sets the constructor picked: ZERO nodes leave the system, NONZERO ones form a Dirichlet group that never moves
            std::vector<int> z, nz;
            for (int v = 0; v < nSim; ++v)
                if (m.isDBCVertex(v)) (m.vertexDBCType[v] == DirichletBCType::ZERO ? z : nz).push_back(v);
            int group = 0;
            if (!z.empty()) {
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)z.size(), z.data(), zero3, zero3, 0.0, inf));
                ++group;
            }
            if (!nz.empty()) {
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)nz.size(), nz.data(), zero3, zero3, 0.0, inf));
                chk(ipcgpu_opt_set_dirichlet_motion(ctx, group, zero3, zero3, nullptr, 1));
            }
        }
        else if (Parts::pulledScript(cfg.animScriptType)) {
            // the NONZERO sets the constructor picked, with the velocities initAnimScript gives them (they are private to AnimScripter: restated)
            std::vector<std::pair<std::vector<int>, std::array<double, 3>>> groups;
            if (cfg.animScriptType == AST_STRETCH || cfg.animScriptType == AST_SQUASH) {
                if (m.borderVerts_primitive.size() != 2) throw std::runtime_error("HipOptimizer: the script needs two border vertex sets");
                const double v = cfg.animScriptType == AST_STRETCH ? -0.1 : 0.03; // (-1)^bI * v along x
                groups.push_back({ m.borderVerts_primitive[0], { v, 0.0, 0.0 } });
                groups.push_back({ m.borderVerts_primitive[1], { -v, 0.0, 0.0 } });
            }
            else if (cfg.animScriptType == AST_CURTAIN) {
                // eight pins along the top edge, pin i drawn in +x at 0.04 (7 - i) / 7; a node takes the first pin whose window holds it
                double lo = 1.0e300, hi = -1.0e300;
                for (int v = 0; v < nSim; ++v) {
                    lo = std::min(lo, m.V(v, 0));
                    hi = std::max(hi, m.V(v, 0));
                }
                groups.resize(8);
                for (int pin = 0; pin < 8; ++pin) groups[pin].second = { 0.04 * (7.0 - pin) / 7.0, 0.0, 0.0 };
                for (int v = 0; v < nSim; ++v) {
                    if (m.vertexDBCType[v] != DirichletBCType::NONZERO) continue;
                    for (int pin = 0; pin < 8; ++pin) {
                        const double x0 = lo + (hi - lo) / 7.0 * pin;
                        if (m.V(v, 0) > x0 - (hi - lo) * 0.0025 && m.V(v, 0) < x0 + (hi - lo) * 0.0025) {
                            groups[pin].first.push_back(v);
                            break;
                        }
                    }
                }
            }
            else {
                std::vector<int> ids;
                for (int v = 0; v < nSim; ++v)
                    if (m.vertexDBCType[v] == DirichletBCType::NONZERO) ids.push_back(v);
                if (cfg.animScriptType == AST_DRAGDOWN) groups.push_back({ ids, { 0.0, -1.5, 0.0 } });
                else groups.push_back({ ids, { -0.15, 0.0, 0.0 } });
            }
            int group = 0;
            for (const auto& g : groups) {
                if (g.first.empty()) continue;
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)g.first.size(), g.first.data(), g.second.data(), zero3, 0.0, inf));
                chk(ipcgpu_opt_set_dirichlet_motion(ctx, group++, g.second.data(), zero3, nullptr, 1));
            }
        }
        else if (cfg.animScriptType == AST_DCOSQUEEZEOUT) {
            // every surface-only component is a held NONZERO node set (AnimScripter.cpp:1261-1280).  The rule that would move the first one
            // down (:2102-2124) never fires as shipped: bottomMin starts at -infinity and is updated with std::min, the test compares with NaN
            int group = 0;
            for (size_t compI = 0; compI < m.componentCoDim.size(); ++compI) {
                if (m.componentCoDim[compI] >= 3) continue;
                std::vector<int> ids;
                for (int v = m.componentNodeRange[compI]; v < m.componentNodeRange[compI + 1]; ++v) ids.push_back(v);
                chk(ipcgpu_opt_add_dirichlet(ctx, (int)ids.size(), ids.data(), zero3, zero3, 0.0, inf));
                chk(ipcgpu_opt_set_dirichlet_motion(ctx, group++, zero3, zero3, nullptr, 1));
            }
        }
        // Neumann groups (Optimizer.cpp:3241-3250; AnimScripter::isNBCActive)
        for (const auto& nbc : m.NeumannBCs) {
            const double f[3] = { nbc.force[0], nbc.force[1], nbc.force[2] };
            const double t0 = std::max(nbc.timeRange[0], cfg.NBCTimeRange[0]), t1 = std::min(nbc.timeRange[1], cfg.NBCTimeRange[1]);
            chk(ipcgpu_opt_add_neumann(ctx, (int)nbc.vertIds.size(), nbc.vertIds.data(), f, t0, t1));
        }
        // start velocity (AnimScripter::initVelocity ran in the base-class constructor)
        {
            bool any = false;
            std::vector<double> vel(3 * (size_t)nAll, 0.0);
            for (int i = 0; i < 3 * nSim; ++i) {
                vel[i] = Base::velocity[i];
                any = any || vel[i] != 0.0;
            }
            if (any) chk(ipcgpu_opt_set_velocity(ctx, vel.data()));
        }
        if (cfg.warmStart) chk(ipcgpu_opt_set_warm_start(ctx, cfg.warmStart));
        if (cfg.restart) chk(ipcgpu_opt_load_status(ctx, cfg.statusPath.c_str()));
        uploaded = true;
    }
};

} // namespace IPC
