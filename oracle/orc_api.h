// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's Newton time-step hot path,
// used as the parity checker by tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py.  The product (ipc_amd/, libipcgpu.so) never includes, links or
// calls anything in this directory.
//
// Parity status: PINNED AGAINST THE REFERENCE ITSELF for everything but two third-party pieces.  The reference's own
// sources (Mesh.cpp, Energy.cpp, NeoHookeanEnergy.cpp, FixedCoRotEnergy.cpp, ImplicitQRSVD.h, MeshCollisionUtils.hpp,
// SelfCollisionHandler.cpp, SpatialHash.hpp, HalfSpace.cpp, LinSysSolver.hpp, get_feasible_steps.cpp, Optimizer.cpp, main.cpp ...)
// compile where they lie under /root/reference into oracle/_ref/libipcref.so (Makefile.ref: Eigen, oneTBB, spdlog, libigl, CLI11,
// MshIO replaced by the small stand-in headers of refshim/).  tests/test_oracle_vs_reference.py compares this restatement with
// what that library returns -- single functions, mesh-level pieces, whole scene scripts run by the reference's main() -- through
// vectors committed under tests/golden/ (tools/make_golden_ref.py), so the check also runs where /root/reference is absent.
// Still "parity unpinned" (un-vendored third-party code, plugged into libipcref.so FROM this oracle, oracle/ref_plug.cpp):
// the time of impact of CTCD (CCD-Wrapper@23907da) -- defined by contract, conservative advancement -- and CHOLMOD's
// factorisation -- own multifrontal Cholesky, checked by residuals.  Dense linear algebra inside Eigen (LDLT, full-pivot LU,
// symmetric eigen-decomposition) is restated in refshim/mini_eigen.hpp from Eigen's documented algorithms.
//
// Layouts follow the reference: V is column-major nV x 3 (x[nV] y[nV] z[nV]), F is
// column-major nT x 4 int32, nodal vectors (gradient, searchDir) are xyzxyz... of
// length 3 nV, 3x3 matrices are column-major.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mesh orc_mesh;
typedef struct orc_opt orc_opt;

// ---- small math (for unit tests)
void orc_svd3(const double* F9, double* U9, double* S3, double* V9);
void orc_make_pd(int n, double* A);
void orc_nh_energy_sigma(const double* s3, double mu, double lam, double* E);
void orc_nh_dPdF(const double* F9, double mu, double lam, double w, int projectSPD, double* dPdF81);
void orc_nh_P(const double* F9, double mu, double lam, double* P9);

// ---- mesh = Mesh<3> data contract (Mesh.cpp:414-527, 246-266, 399-401, 660-671)
orc_mesh* orc_mesh_create(int nV, int nT, const double* Vrest_colmajor, const int* F_colmajor,
    double YM, double PR, double density);
void orc_mesh_destroy(orc_mesh*);
void orc_mesh_set_surface(orc_mesh*, int nSF, const int* SF_colmajor); // adds SF edges to vNeighbor, builds SVI/SFEdges
void orc_mesh_set_surface_codim(orc_mesh*, int nSF, const int* SF_colmajor, int nCE, const int* CE_pairs); // + `.seg` segments; isolated nodes = `.pt` points
void orc_mesh_set_exact_predicates(orc_mesh*, int on); // intersection checks as a USE_PREDICATES build of the reference makes them
int orc_seg_tri_intersect_exact(const double* X15);
void orc_mesh_set_dbc(orc_mesh*, int n, const int* vids, int type); // type: 1 ZERO, 2 NONZERO (Mesh.hpp:41-45)
void orc_mesh_set_obstacle(orc_mesh*, int n, const int* vids, int obstacleOnly);
void orc_mesh_set_codim_nodes(orc_mesh*, int n, const int* vids, const double* nodeMass); /* surface-only nodes of Mesh<3> (Mesh.cpp:310-345) */
void orc_mesh_clear_dbc(orc_mesh*);
void orc_mesh_set_energy_type(orc_mesh*, int type); // 0 NH, 1 FCR (Config.cpp:23-24)
// componentMaterial entry of Mesh::setLameParam (Mesh.cpp:661-671): node range gets density rho, tet range gets (YM, PR)
void orc_mesh_set_component_material(orc_mesh*, int nodeBegin, int nodeEnd, int tetBegin, int tetEnd, double rho, double YM, double PR);
void orc_mesh_set_V(orc_mesh*, const double* V_colmajor);
void orc_mesh_get_V(const orc_mesh*, double* V_colmajor);
void orc_mesh_get_features(const orc_mesh*, double* restTriInv9nT, double* triArea, double* mass, double* mu, double* lam);
double orc_mesh_avg_edge_len(const orc_mesh*);
double orc_mesh_bbox_diag2(const orc_mesh*);
int orc_mesh_check_inversion(const orc_mesh*); // 1 = no inverted element (Mesh.cpp:715-764)

// ---- elasticity (Energy.cpp:195-562, NeoHookeanEnergy.cpp:55-153)
void orc_elastic_energy(const orc_mesh*, double coef, double* E, double* perElem /*nT or NULL*/);
void orc_elastic_gradient(const orc_mesh*, double coef, int projectDBC, double* g3nV);
void orc_elastic_hessian_elem(const orc_mesh*, int e, double coef, int projectSPD, double* H144_colmajor);

// ---- CSR pattern + assembly (LinSysSolver.hpp:46-150, IglUtils.hpp:39-116, Optimizer.cpp:3549-3668)
// pattern of the symmetric-upper CSR, 0-based; returns nnz. Pointers stay valid until the next build.
int orc_pattern_build(orc_mesh*); // from mesh vNeighbor (+ extra connectivity set by orc_pattern_add_edges)
void orc_pattern_add_edges(orc_mesh*, int n, const int* pairs2n); // augmentConnectivity stand-in
int orc_pattern_rows(const orc_mesh*);
const int* orc_pattern_ia(const orc_mesh*);
const int* orc_pattern_ja(const orc_mesh*);
// setZero + elastic Hessian + mass / DBC diagonal == computePrecondMtr without contact
void orc_assemble_hessian(const orc_mesh*, double coef, int projectDBC, double* a_nnz);
// y = A x with the symmetric-upper CSR (LinSysSolver.hpp:238-253)
void orc_csr_symv(const orc_mesh*, const double* a, const double* x, double* y);

// ---- step bounds
void orc_inversion_step(const orc_mesh*, const double* p3nV, double slackness, double* perElem /*nT*/);
double orc_filter_step_size(const orc_mesh*, const double* p3nV, double stepSize); // Energy.cpp:565-581

// ---- sparse Cholesky on the symmetric-upper CSR (stand-in for CHOLMODSolver.cpp:123-154)
typedef struct orc_chol orc_chol;
orc_chol* orc_chol_create(int n, const int* ia, const int* ja, int nthreads);
void orc_chol_destroy(orc_chol*);
long long orc_chol_nnzL(const orc_chol*);
double orc_chol_flops(const orc_chol*);
int orc_chol_factorize(orc_chol*, const double* a); // 1 ok, 0 not positive definite
void orc_chol_solve(const orc_chol*, const double* rhs, double* x);

// ---- optimizer (Optimizer.cpp:457-627, 1518-1819, 1822-2213, 2324-2355, 2662-2945, 3199-3720), no contact yet
orc_opt* orc_opt_create(orc_mesh*, double dt, int withGravity, int nthreads);
void orc_opt_destroy(orc_opt*);
void orc_opt_set_twist(orc_opt*, int nL, const int* left, int nR, const int* right, double angVel); // AnimScripter.cpp:555-572
void orc_opt_set_dirichlet_motion(orc_opt*, int group, const double* lin3, const double* angRad3, const double* center3 /*nullable*/, int forceNonzero);
void orc_opt_set_dirichlet_targets(orc_opt*, int group, int n, const double* targets_3n); // mesh-sequence motion; NULL ends it
void orc_opt_force_friction_loop(orc_opt*, int on); // Optimizer.cpp:156-161: a MeshCO friction coefficient only switches the lagging loop on
void orc_opt_set_friction_scales(orc_opt*, double scaleSelf, double scaleObstacle); // MeshCO::friction beside selfFric
void orc_opt_set_rel_tol(orc_opt*, double relTol); // Optimizer.cpp:390-396
void orc_opt_set_kappa(orc_opt*, double kappa); /* tuning[0]: start value of the barrier stiffness in every time step (Optimizer.cpp:1540-1547) */
void orc_opt_set_dhat_target(orc_opt*, double dHatTargetEps); /* tuning[2]: the dHat homotopy of fullyImplicit_IP (Optimizer.cpp:283-289, 1706-1713, 1763-1774) */
void orc_opt_set_damping(orc_opt*, double dampingStiff); /* Config.cpp:141-147, 614-616; before precompute */
int orc_opt_precompute(orc_opt*); /* -1: the initial configuration intersects (the reference exits, Optimizer.cpp:258-263) */
// one pass of the solveSub_IP loop body; returns 1 if the time step converged before doing work
int orc_opt_newton_iter(orc_opt*);
void orc_opt_begin_timestep(orc_opt*); // stepAnimScript + initX(0) + energy, Optimizer.cpp:510-560,1518-1613
void orc_opt_end_timestep(orc_opt*); // BE / NM velocity, acceleration and xTilta update, Optimizer.cpp:570-590
// Mesh::DirichletBCs entry / scripted component velocity (Mesh.hpp:23-39, AnimScripter.cpp:58-110, 1413-1462); angular velocity in rad/s
void orc_opt_add_dirichlet(orc_opt*, int n, const int* ids, const double* lin3, const double* ang3, double t0, double t1);
void orc_opt_end_dirichlet(orc_opt*, int group, double t_end); /* mesh.resetDBCVertices() of a state-dependent script (AnimScripter.cpp:1619-1632) */
void orc_opt_add_neumann(orc_opt*, int n, const int* ids, const double* accel3, double t0, double t1); // Mesh::NeumannBCs, Optimizer.cpp:3241-3250, 3452-3461
void orc_opt_get_dbc_state(const orc_opt*, double* out4); // completed step, rho_DBC, m_projectDBC, #targets (Optimizer.cpp:2168-2203)
void orc_opt_get_kinematics(const orc_opt*, double* vel_3nV, double* acc_3nV, double* dx_3nV);
void orc_opt_restart(orc_opt*, int timestep, const double* vel, const double* acc, const double* dx); // Optimizer.cpp:179-248
void orc_opt_set_time_integration(orc_opt*, int type /*0 BE, 1 NM*/, double beta, double gamma); // Config.cpp:112-118
int orc_opt_solve_timestep(orc_opt*, int maxIter); // returns # Newton iterations
// state readers
void orc_opt_get(const orc_opt*, double* V_colmajor, double* searchDir, double* gradient, double* scalars8);
// scalars8 = {lastEnergyVal, lastStepSize, targetGRes, innerIterAmt, timestep, lastAlphaFeasible, 0, 0}
void orc_opt_set_parameter_scaling(orc_opt*, int useAbsParameters, double dTolRel, double kappaMinMultiplier); // Config.cpp:553-558
void orc_opt_set_friction_target(orc_opt*, double eps_v_target); // eps_v homotopy (tuning[5]); <= 0: none
void orc_opt_set_warm_start(orc_opt*, int option); // Config warmStart: initX option 0..4 (Optimizer.cpp:925-1080)
double orc_opt_warm_step(const orc_opt*);
void orc_opt_timers(const orc_opt*, double* t16); // timer_step buckets (main.cpp:1326-1340)

#ifdef __cplusplus
}
#endif
