// Host-side mirror of the three reference interfaces on the hot path, with their state resident in HBM:
//   HipMesh          <- Mesh<3>                       (src/Mesh.hpp:58-171)
//   HipLinSysSolver  <- LinSysSolver<VectorXi,VectorXd> (src/LinSysSolver/LinSysSolver.hpp:31-467)
//   HipNeoHookean    <- Energy<3>/NeoHookeanEnergy<3>  (src/Energy/Energy.hpp:27-138)
//   HipOptimizer     <- Optimizer<3>                   (src/TimeStepper/Optimizer.hpp:28-283)
// Method names and argument meaning follow the reference so that the adapters in INTEGRATION.md are
// one-line forwards.  Everything here is reached through the C ABI of include/ipcgpu.h.
#pragma once
#include "common.h"
#include "mf_numeric.h"
#include "mf_symbolic.h"
#include "nh_kernels.h"
#include "patch_assembly.h"
#include "hip_contact.h"
#include "hip_halfspace.h"
#include <map>
#include <memory>
#include <string>

struct ipcgpu_ctx; // opaque C handle
typedef int (*ipcgpu_allreduce_fn_t)(void* user, void* buf_dev, long long count, int op);
typedef int (*ipcgpu_allreduce_stream_fn_t)(void* user, void* buf_dev, long long count, int op, void* hipStream);

namespace ipcgpu {

class HipMesh {
public:
    int nV = 0, nT = 0;
    int nElemNodes = 0; // nodes referenced by at least one element (mean nodal mass, bounding box)
    std::vector<double> V_rest; // column-major nV x 3
    std::vector<int> F; // column-major nT x 4
    std::vector<double> restTriInv; // SoA [9][nT]
    std::vector<double> triArea, mass, mu, lam;
    std::vector<int> dbcType;
    std::vector<int> nbPtr, nb; // vNeighbor as sorted CSR adjacency (Mesh.cpp:470-493)
    double avgEdgeLen = 0, bboxDiag2 = 0, bboxLo[3] = { 0, 0, 0 }, bboxHi[3] = { 0, 0, 0 };
    // device
    DevBuf<double> d_x, d_xTilde, d_mass, d_A, d_vol, d_mu, d_lam;
    DevBuf<int> d_dbc;
    DevBuf<int4> d_tet;

    // Mesh::computeFeatures + computeMassMatrix + setLameParam (Mesh.cpp:414-527, 246-266, 399-401, 660-671)
    unsigned featuresVersion = 0; // bumped by every computeFeatures: host-side caches of mesh topology key on it
    void computeFeatures(int nV, int nT, const double* Vrest, const int* F, double YM, double PR, double density, hipStream_t s);
    // surface-only nodes that belong to the mesh (triangle meshes under `shapes`, componentCoDim 2): they count in the bounding box
    // and the mean nodal mass and carry the given lumped masses (Mesh.cpp:310-345)
    std::vector<char> inMesh; // per node: referenced by an element (the components of codimension 3)
    void addSurfaceEdges(int nSF, const int* SF_colmajor, int nCE = 0, const int* CE_pairs = nullptr);
    void setCodimNodes(int n, const int* ids, const double* nodeMass, hipStream_t s);
    void meshBBox(); // bounding box and node count over inMesh (Mesh::matSpaceBBoxSize2(dim) / avgNodeMass(dim): tetrahedral components only)
    void uploadDBC(hipStream_t s);
    int energyType = 0; // Config `energy NH|FCR` (Config.cpp:23-24): 0 neo-Hookean, 1 fixed corotated
    double density = 0; // global density handed to computeFeatures (component overrides rescale the nodal mass by rho / density)
    void setComponentMaterial(int nodeBegin, int nodeEnd, int tetBegin, int tetEnd, double rho, double YM, double PR, hipStream_t s);
    bool isDBCVertex(int v) const { return dbcType[v] != 0; }
    bool isProjectDBCVertex(int v, bool projectDBC) const { return dbcType[v] == 1 || (dbcType[v] == 2 && projectDBC); }
};

class HipLinSysSolver {
public:
    explicit HipLinSysSolver(hipStream_t s);
    ~HipLinSysSolver();
    int numRows = 0;
    std::vector<int> ia, ja; // 0-based symmetric-upper CSR (host copy, get_ia / get_ja)
    DevBuf<int> d_ia, d_ja;
    DevBuf<double> d_a;
    DevBuf<double> hostDelta, hostSetVal; // staging of ipcgpu_linsys_apply_host_updates
    DevBuf<unsigned char> hostSetMask;
    // per-node row geometry + per-tet edge slots used by the element kernels
    std::vector<int> rowBase, rowLen;
    DevBuf<int> d_rowBase, d_rowLen, d_edgeP0;
    int solverType = 0;
    int patternVersion = 0; // bumped by every set_pattern*: consumers (patch plan) rebuild lazily
    hipStream_t stream;

    // set_pattern(vNeighbor, fixedVert) (LinSysSolver.hpp:46-150); extra = contact connectivity
    void set_pattern(const HipMesh& mesh, int nExtra, const int* extraPairs);
    void set_pattern_csr(int nRows, const int* ia, const int* ja);
    void buildElementMap(const HipMesh& mesh); // tet edge -> CSR slot
    void setZero();
    int findEntry(int row, int col) const;
    void analyze_pattern(const HipMesh* meshForCoords);
    bool factorize();
    void solve(const double* rhs_dev, double* x_dev);
    bool factorizeSolve(const double* rhs_dev, double* x_dev, bool wait = true); // factorize + solve, forward sweep overlapped with the factorisation
    bool lastPivotsOk() const; // after factorizeSolve(..., wait = false) and a synchronisation of the stream
    bool lastSyncOk_ = true;
    void multiply(const double* x_dev, double* y_dev);
    void precondition_diag(const double* in_dev, double* out_dev);
    int getNumRows() const { return numRows; }
    int getNumNonzeros() const { return (int)ja.size(); }
    const MfSymbolic& symbolic() const { return sym_; }
    // subtree-sharded factorisation / solves over `world` ranks (MfNumeric::setShard); takes effect at the next analyze_pattern
    void setShard(int rank, int world, ipcgpu_allreduce_fn_t fn, void* user, ipcgpu_allreduce_stream_fn_t sfn = nullptr, void* streamUser = nullptr)
    {
        num_.setShard(rank, world, fn, user, sfn, streamUser);
        analyzed_ = false;
    }
    void setHooks(ipcgpu_allreduce_fn_t fn, void* user, ipcgpu_allreduce_stream_fn_t sfn, void* streamUser) { num_.setHooks(fn, user, sfn, streamUser); }
    void setExchangeHooks(MfNumeric::ExchangeFn fn, void* user, MfNumeric::ExchangeStreamFn sfn, void* streamUser) { num_.setExchangeHooks(fn, user, sfn, streamUser); }
    bool hasExchangeHook() const { return num_.hasExchangeHook(); }
    int solverWorld() const { return num_.world(); }
    void setBulkTuning(double minMB, int block) { num_.setBulkTuning(minMB, block); }
    void entryDestinations(long long* out) const { num_.entryDestinations(out, ja.size()); }
    void nodeOwners(std::vector<int>& o) const { num_.nodeOwners(o); }
    long long exchangedBytes() const { return num_.exchangedBytes(); }
    long long exchangeCalls() const { return num_.exchangeCalls(); }
    long long sentBytes() const { return num_.sentBytes(); }
    long long receivedBytes() const { return num_.receivedBytes(); }
    double exchangeWaitMs() { return num_.exchangeWaitMs(); }
    void criticalPath(double* out5) const { num_.criticalPath(out5); }
    int analysisVersion = 0; // bumped by every analyze_pattern (the owner-computes plan of the assembly follows the solver's cut)
    double sharedFlopFraction() const { return num_.sharedFlopFraction(); }
    bool analyzed() const { return analyzed_; }

private:
    MfSymbolic sym_;
    MfNumeric num_;
    bool analyzed_ = false;
    std::unique_ptr<struct RocsolverCsrrf> rs_;
    friend struct RocsolverCsrrf;
};

class HipOptimizer {
public:
    HipOptimizer(HipMesh& mesh, HipLinSysSolver& lin, hipStream_t s);
    void init(double dt, bool withGravity);
    void setRelGL2Tol(double relTol);
    void setParameterScaling(bool absolute, double dTolRel, double kappaMinMultiplier); // useAbsParameters / tuning[3] / kappaMinMultiplier
    void setTwist(int nL, const int* left, int nR, const int* right, double angVel);
    void precompute();
    void beginTimestep();
    bool newtonIter(); // true = converged before doing work
    void endTimestep();
    int solveTimestep(int maxIter);

    // building blocks (virtuals of Optimizer.hpp:229-283)
    double computeEnergyVal();
    void computeGradient(bool projectDBC);
    void penaltyGradientAdd(bool projectDBC);
    void computePrecondMtr(bool projectDBC, bool withGradient);
    // lagged stiffness-proportional damping (Optimizer.cpp:3381-3400, 3519-3540, 3707-3709, 3723-3735; Config.cpp:141-157, 614-616):
    // D = projected elastic Hessian at the state the last time step ended in, times dampingStiff / dt, on the solver's pattern
    double dampingStiff = 0.0;
    double ctorDt = 0.025; // the step size inside eps_v^2 h^2 and CN_MBC: the reference's constructor leaves its setTime(10, 0.025) in them (hip_optimizer.cpp, top)
    double dHatTargetEps = -1.0; // tuning[2] (Optimizer.cpp:283-289): every time step starts at dHat and halves it down to this; < 0: no homotopy
    double kappaConfig = 0.0; // tuning[0] (Config.cpp:41-45): the barrier stiffness a time step starts from, 0 = suggestKappa
    void setDamping(double stiff);
    void computeDampingMtr(); // at the current positions (precompute, end of a time step)
    void assembleDampingMtr(); // (re)builds the values at the remembered positions on the current pattern
    double dampingEnergy();
    void dampingGradientAdd(bool projectDBC, double* grad_dev);
    void computeSearchDir(bool projectDBC);
    void lineSearch(double& stepSize);
    void stepForward(const double* x0_dev, double alpha);
    double filterStepSize(const double* p_dev, double stepSize);
    double fullCcd(double slackness, double stepSize); // the full sweep of the search direction in the mode of the contact handler
    bool checkInversion();
    ElemView view() const;

    HipMesh& mesh;
    HipLinSysSolver& lin;
    hipStream_t stream;
    double dt = 0.025, dtSq = 0, gravity[3] = { 0, 0, 0 };
    double relGL2Tol = 1e-8, targetGRes = 0;
    // Config `timeIntegration BE | NM beta gamma` (Config.hpp:96, Config.cpp:112-118): 0 backward Euler, 1 Newmark
    int timeIntegration = 0;
    int warmStart = 0; // Config `warmStart`: initX option 0..4 (Optimizer.cpp:925-1080)
    double warmStepSize = 0.0; // step the last warm start could take
    double betaNM = 0.25, gammaNM = 0.5;
    DevBuf<double> d_acc, d_dxElastic; // acceleration, dx_Elastic = x - xTilta of the finished step (Optimizer.cpp:176-177, 574-586)
    double elasticCoef() const { return timeIntegration == 1 ? dtSq * betaNM : dtSq; } // Optimizer.cpp:3205-3224, 3416-3434, 3618-3632
    void setTimeIntegration(int type, double beta, double gamma);
    void getKinematics(double* vel, double* acc, double* dxElastic);
    void getDbcState(double* out4) const;
    void saveStatus(const std::string& path); // Optimizer::saveStatus, Optimizer.cpp:2964-3011
    void loadStatus(const std::string& path); // restart, Optimizer.cpp:179-248
    DevBuf<double> d_vel, d_xPrev, d_searchDir, d_gradient, d_minusG, d_x0, d_partial, d_scalar;
    DevBuf<double> d_damp, d_xDamp, d_zeroMass, d_dampDx, d_dampAdx;
    unsigned long long dampPatternVersion = ~0ULL;
    DevBuf<int> d_flag, d_handleIds;
    DevBuf<double> d_handleAng;
    int nHandles = 0;
    // Mesh::DirichletBCs (Mesh.hpp:23-39) and scripted component velocities (AnimScripter.cpp:1413-1435)
    struct DbcGroup {
        std::vector<int> ids;
        DevBuf<int> d_ids;
        DevBuf<double> d_pos;
        double lin[3], ang[3], t0, t1;
        // the hard-coded scripts of AnimScripter.cpp:1961-2135 (DCOSquash6, DCOSqueezeOut, DCORotCylinders, ...): every node of the group is a
        // NONZERO Dirichlet node whatever its velocity currently is, and rotations turn about a centre fixed at set-up time
        bool forceNonzero = false, hasCenter = false;
        double center[3] = { 0, 0, 0 };
        // mesh-sequence motion (`meshSeq <folder>`, AnimScripter.cpp:1465-1528): the positions the nodes are to reach in the next time step,
        // handed over before every step; the velocities above are then ignored
        bool hasTargets = false;
        DevBuf<double> d_targets;
        bool isZero() const { return !forceNonzero && lin[0] == 0 && lin[1] == 0 && lin[2] == 0 && ang[0] == 0 && ang[1] == 0 && ang[2] == 0; }
    };
    std::vector<std::unique_ptr<DbcGroup>> dbcGroups;
    // Mesh::NeumannBCs (Mesh.hpp:47-56): a mass-weighted acceleration on a vertex set while t0 <= stepStartTime < t1
    struct NbcGroup {
        int n = 0;
        DevBuf<int> d_ids;
        double a[3], t0, t1;
    };
    std::vector<std::unique_ptr<NbcGroup>> nbcGroups;
    void addNeumannBC(int n, const int* ids, const double* accel3, double t0, double t1);
    void neumannGradientAdd(double* grad_dev); // Optimizer.cpp:3452-3461
    double neumannEnergy(); // :3241-3250
    std::vector<int> baseDbcType;
    double stepStartTime = 0, stepEndTime = 0; // AnimScripter.cpp:1406-1407
    // augmented-Lagrangian Dirichlet fallback (AnimScripter.cpp:2150-2157, 2280-2350; Optimizer.cpp:1826-1828, 2168-2203)
    std::vector<int> tpIds, tpIdsOnDevice; // targetPos keys = the Dirichlet nodes, ascending (and the list d_tpIds currently holds)
    DevBuf<int> d_tpIds;
    DevBuf<double> d_tpPos, d_tpLam, d_tpStage;
    double dist2Tol = 0, completedStep = 1.0, lastMove = 1.0, rhoDBC = 0.0, CN_MBC = 0.0;
    bool projDBC = true; // m_projectDBC
    MdbcView mdbc() const { return MdbcView{ (int)tpIds.size(), d_tpIds.p, d_tpPos.p, d_tpLam.p, mesh.d_mass.p }; }
    void buildTargetPositions(bool deferTolerance = false); // after the scripted search direction is known, before it is applied
    void finishTolerance(); // dist2Tol from the scripted directions buildTargetPositions(true) left in pinned host memory (behind a synchronisation)
    PinnedBuf<double> h_tpStage;
    bool tolPending = false;
    void initSubProblem(); // head of solveSub_IP (Optimizer.cpp:1826-1828)
    double computeCompletedStepSize(); // AnimScripter.cpp:2286-2300
    void dirichletPenaltyUpdate(); // Optimizer.cpp:2168-2203
    void addDirichletBC(int n, const int* ids, const double* lin3, const double* angRad3, double t0, double t1);
    void setDBCVertices(); // AnimScripter::setDBCVertices, AnimScripter.cpp:58-110
    bool dbcGroupMotion(); // adds the active groups' motion to d_searchDir; true if any
    double rotCenter[3] = { 0, 0, 0 };
    PinnedBuf<double> h_scalar;
    PinnedBuf<int> h_flag;
    int innerIterAmt = 0, globalIterNum = 0, k = 0;
    double lastEnergyVal = 0, lastStepSize = 0, lastAlphaFeasible = 0;
    double timers[16] = { 0 };
    bool initialised = false;
    int rank = 0, worldSize = 1;
    int tetBegin = 0, tetEnd = 0;
    ipcgpu_allreduce_fn_t allreduce = nullptr;
    ipcgpu_allreduce_stream_fn_t allreduceStream = nullptr; // takes precedence: enqueued on `stream`, no host synchronisation
    void hookReduce(double* dev, long long n, int op);
    // Contact-free single-rank iterations (the matTwist bench): the scalars the host branches on come back in batches -- one read after the
    // solve (|p|_inf for the next convergence test, the inversion step filter, E at the current iterate), one after the trial step (inversion
    // flag + E at the trial point) -- instead of one host synchronisation per scalar; the assembly bucket is timed with HIP events.
    bool fastPath() const;
    bool cachedDistValid = false, cachedE0Valid = false;
    double cachedDist = 0, cachedE0 = 0, cachedFilter = 0;
    // ... and, second step (IPCGPU_NO_TRIAL_AHEAD restores the above): the first trial of the line search is taken on the device behind the solve
    // as well -- its step size, inversion flag and energy arrive with the same synchronisation, ONE per Newton iteration
    bool cachedTrialValid = false, cachedTrialInverted = false;
    double cachedTrialE = 0, cachedAlpha = 1.0;
    // ... and, round 4: the NEXT iteration's fused assembly (gradient + Hessian at the trial point) is enqueued behind the trial kernels, before that one
    // synchronisation -- into buffers of its own, swapped in when the trial is accepted and the next pass asks for exactly this assembly.  The host's
    // read-back, its decisions and the way back through the caller's loop (45 us per iteration on the bench) then run beside a kernel instead of in front of it.
    // Nothing is skipped and nothing observable moves: d_gradient and the solver's values keep describing the iterate the search direction came from
    // until the swap; a rejected trial, a converged pass (one assembly per time step goes unused), a bad pivot or a time-step call drop the buffers' content.
    DevBuf<double> d_aSpec, d_gradSpec;
    bool specAsmValid = false, specAsmOn = true; // IPCGPU_NO_SPEC_ASSEMBLY=1
    void speculativeAssembly();
    hipEvent_t evAsm0 = nullptr, evAsm1 = nullptr, evTail = nullptr;
    bool evAsmPending = false;
    int evAsmWeight = 1;
    unsigned asmLaunches = 0;
    void resolveEventTimers();
    DevBuf<double> d_contactG; // this rank's share of the barrier forces before their all-reduce (contact-pair lists sharded)
    void* allreduceUser = nullptr;
    void* allreduceStreamUser = nullptr; // separate slots: the two hooks can be set in any order
    void reduceSum(double* dev, long long n);
    void reduceMin(double* dev, long long n);
    double readScalar(const double* dev);
    PatchPlan patch; // atomic-free assembly plan for the current pattern
    int patchVersion = -1;
    void ensurePatchPlan();
    void patchShard(int& pb, int& pe) const;
    // Owner-computes sharding (round 4; SURVEY.md 8e as north_star states it): with the assembly AND the solver sharded, a rank assembles exactly the CSR rows
    // its fronts read -- the rows of the nodes its subtrees eliminate plus the rows of the separator nodes above the cut, which every rank repeats -- by running
    // the patches that hold such a node (elements and contact stencils on a cut are evaluated by both sides, like the halo of a patch).  NO matrix value crosses
    // ranks; the gradient does (one all-reduce of 3 nV doubles in which every node is contributed by one designated rank), scalars, and what the solver
    // exchanges above its cut (update matrices / vectors of the subtree roots, the solution).
    bool ownerMode() const;
    void ensureOwnerPlan();
    int ownerPlanPatch = -1, ownerPlanAnalysis = -1;
    int nOwnerPatches = 0;
    DevBuf<int> d_ownerPatches; // the patches this rank runs
    DevBuf<unsigned char> d_need, d_mine; // per node: rows needed on this rank / this rank is the node's designated contributor to all-reduced nodal vectors
    long long ownerNeededNodes = 0;
    bool matrixComplete = true; // false after an owner-mode assembly: a[] holds this rank's rows only
    void maskAndReduceGradient(double* g);
    void completeMatrix(); // sum of the designated rows over the ranks: for the consumers of the WHOLE matrix on a sharded context (diagonal fallback, get_a)
    long long commBytes = 0, commCalls = 0; // all-reduced through the hook by the optimizer itself (the solver counts its own: HipLinSysSolver::exchangedBytes)
    // self-contact, interior point (fullyImplicit_IP with isSelfCollision, Optimizer.cpp:1518-1819)
    HipContact* contact = nullptr;
    bool selfCollision = false;
    double dHatEps = 1.0e-3, dHat = 0, kappa = 0, dTol = 0;
    bool absParameters = false; // useAbsParameters: lengths of `tuning` and the tolerance are absolute (Config.cpp:553-555)
    double dTolRel = 1.0e-9, kappaMinMultiplier = 1.0e11; // tuning[3] (Optimizer.cpp:102-106), Config.hpp:139
    double lenScale2() const { return absParameters ? 1.0 : mesh.bboxDiag2; }
    std::vector<std::pair<int, int>> curExtra; // contact connectivity inside the current pattern (vNeighbor_IP)
    std::vector<std::array<int, 4>> closeID; // closeMConstraintID / Val (Optimizer.cpp:2396-2440)
    std::vector<double> closeVal;
    int lastCCDPair[2] = { 0, 0 }, nFullCCD = 0, nPatternChanges = 0, dbcIncomplete = 0;
    // "does the pattern cover the contact sets" as answered at the end of the last iteration, stamped with the versions of sets and pattern it was asked for
    unsigned long long coverSets = ~0ull;
    int coverPattern = -1;
    bool coverAnswer = false;
    // look-ahead of the contact pattern in units of dHat (ipcgpu_opt_set_pattern_lookahead; < 1 = exact pattern).  Negative = by mesh size (lookahead()): what it
    // trades is host-side analyses against fill in the factor, and the two scale differently -- measured (profiles/r06_pattern_lookahead_ab.txt): 40 K nodes, ms per
    // Newton iteration at pad 1 / 2.25 / 4: 22.6 / 11.4 / 9.9 (27 / 7 / 4 analyses of ~14 ms); 375 K nodes: 461 / 258 / 357 (an analysis costs ~130 ms, but the
    // factorisation 139 / 243 / 341 ms -- 3.7 / 7.2 / 10.7 TFLOP)
    double patternPad = -1.0;
    double lookahead() const { return patternPad >= 0.0 ? patternPad : (mesh.nV > 150000 ? 2.25 : 4.0); }
    // analytic half-space obstacles (animConfig.collisionObjects) and their close-constraint list (Optimizer.cpp:2364-2374)
    std::vector<std::unique_ptr<HipHalfSpace>> planes;
    std::vector<std::pair<int, int>> closeHS;
    std::vector<double> closeHSVal;
    // lagged friction: Optimizer.cpp:286-304, 1525-1600, 1615-1790; the lagged sets live in HipContact /
    // HipHalfSpace, x^t is d_xPrev
    double selfFric = 0.0, epsV = 1.0e-3, fricDHat0 = 0, fricDHat = -1.0;
    // eps_v homotopy (tuning[5], Optimizer.cpp:296-303, 1717, 1776-1781): fricDHat is halved down (or clamped UP) to this target between the
    // friction-lag passes, the tangent-space convergence test runs only once it is there; < 0: the same as the start value (no homotopy)
    double epsVTarget = -1.0, fricDHatTarget = 0;
    int fricIterAmt = 1, fricIterI = 0;
    bool fricLoopForced = false; // a mesh collision object with a friction coefficient: the lagging loop runs, no pair carries friction (Optimizer.cpp:156-161)
    bool solveFric() const;
    void updateFrictionLag();
    bool nextSubproblem(); // tail of the fullyImplicit_IP loop body after a converged solveSub_IP; true = run another one
    bool ipOn() const { return selfCollision || !planes.empty(); }
    size_t nConstraints() const;
    int addHalfSpace(HipContact* c, const double* origin3, const double* normal3, double dHatEps);
    bool anyIntersection();
    void elasticInertiaGradient(bool projectDBC, bool finish = true); // finish = false: owner-computes callers that exchange the sum themselves
    void barrierGradientAdd(bool projectDBC, double kappa, bool activeOnly, double* grad_dev, int part = 0);
    void enableSelfCollision(HipContact* c, double dHatEps);
    void setVelocity(const double* vel3nV);
    void computeConstraintSets();
    double kappaFloor() const;
    void initKappa();
    void postLineSearch();
    bool isIntersected();
    void computeXTilta();
};

} // namespace ipcgpu

struct ipcgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::unique_ptr<ipcgpu::HipMesh> mesh;
    std::unique_ptr<ipcgpu::HipLinSysSolver> lin;
    std::unique_ptr<ipcgpu::HipOptimizer> opt;
    std::unique_ptr<ipcgpu::HipContact> contact;
    int rank = 0, worldSize = 1;
    // point-to-point exchange hooks of the sharded solver (ipcgpu_opt_set_exchange[_stream]); stored here as opaque pointers of the C ABI's types
    void* exchange = nullptr;
    void* exchangeUser = nullptr;
    void* exchangeStream = nullptr;
    void* exchangeStreamUser = nullptr;
};
