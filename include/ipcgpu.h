/* ipcgpu.h -- C ABI of the MI355X-native IPC Newton hot path (libipcgpu.so).
 *
 * The reference has no FFI: its hot path sits behind three C++ class interfaces
 * (SURVEY.md 8b).  A maintainer drops this library in by adding three thin adapter
 * subclasses (shown in INTEGRATION.md) that forward to the entry points below:
 *
 *   LinSysSolver<VectorXi,VectorXd>   src/LinSysSolver/LinSysSolver.hpp:31-467
 *   Energy<3> / NeoHookeanEnergy<3>   src/Energy/Energy.hpp:27-138
 *   Optimizer<3>                      src/TimeStepper/Optimizer.hpp:28-283
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns an int status:
 *       IPCGPU_OK 0, IPCGPU_NOT_PD 1 (factorize: matrix not positive definite,
 *       the reference's `factorize() == false`, CHOLMODSolver.cpp:130-137),
 *       negative = error (message via ipcgpu_last_error()).  No exceptions cross
 *       the boundary.  One opaque context per Optimizer; a context is not thread-safe.
 *   - layouts are the reference's: V is column-major nV x 3 (Eigen::MatrixXd,
 *     Mesh.hpp:61), F is column-major nT x 4 int32 (Mesh.hpp:64), nodal vectors
 *     are xyzxyz... of length 3 nV (Optimizer.hpp:91-92), the matrix is the
 *     symmetric-UPPER CSR with 3x3 node blocks of LinSysSolver.hpp:46-150, 0-based.
 *   - host pointers unless the name ends in _dev.  Device state (positions, CSR
 *     values, factor) stays resident in HBM between calls.
 *   - all arithmetic is fp64, all indices int32, like the reference.
 */
#ifndef IPCGPU_H
#define IPCGPU_H

#ifdef __cplusplus
extern "C" {
#endif

#define IPCGPU_OK 0
#define IPCGPU_NOT_PD 1
#define IPCGPU_ERR_ARG -1
#define IPCGPU_ERR_HIP -2
#define IPCGPU_ERR_STATE -3
#define IPCGPU_ERR_UNSUPPORTED -4

typedef struct ipcgpu_ctx ipcgpu_ctx;

/* DirichletBCType, src/Mesh.hpp:41-45 */
enum { IPCGPU_NOT_DBC = 0, IPCGPU_DBC_ZERO = 1, IPCGPU_DBC_NONZERO = 2 };

/* linear-solver back ends selectable per context (the reference selects CHOLMOD / EIGEN /
 * AMGCL through LinSysSolver::create, LinSysSolver.cpp:10-28) */
enum { IPCGPU_SOLVER_MULTIFRONTAL = 0, IPCGPU_SOLVER_ROCSOLVER_CSRRF = 1 };

const char* ipcgpu_last_error(void);
int ipcgpu_version(void);

/* ---- context ------------------------------------------------------------------------ */
int ipcgpu_ctx_create(int device_id, ipcgpu_ctx** out);
int ipcgpu_ctx_destroy(ipcgpu_ctx*);
int ipcgpu_ctx_set_solver(ipcgpu_ctx*, int solver_type);
/* Sharding of the assembly for multi-GPU runs, one context per rank.  Together with ipcgpu_linsys_set_shard (same world) this is
 * OWNER-COMPUTES ROWS: the rank assembles exactly the CSR rows its fronts read (the rows of the nodes its subtrees eliminate plus the separator rows
 * above the cut; elements and contact stencils on a cut are evaluated by both sides) and NO matrix value crosses ranks -- see ipcgpu_opt_comm_stats
 * below.  Energies, step bounds and the CCD sweeps are split by index ranges (elements [nT rank / world, nT (rank + 1) / world) in the order given to
 * set_mesh) and combined by scalar all-reduces through the hook of ipcgpu_opt_set_allreduce[_stream].  Without a sharded solver (and for runs with
 * lagged damping) the element pass is split by that same element range and the partial gradient / matrix values are summed by the hook.
 * A consumer of the WHOLE matrix on an owner-computes context (ipcgpu_linsys_get_values / multiply / precondition_diag) gets IPCGPU_ERR_STATE until
 * ipcgpu_opt_complete_matrix has been called on every rank.  Default: rank 0 of 1. */
int ipcgpu_ctx_set_shard(ipcgpu_ctx*, int rank, int world_size);

/* ---- tet-mesh files (host only, no GPU) ------------------------------------------------ */
/* IglUtils::readTetMesh (src/Utils/IglUtils.cpp:451-584): Gmsh MSH 4.1 / 2.2 ASCII (the MshIO path) and the reference's own
 * "msh 4.0" dialect with its optional $Surface section; nodes in file order, elements by tag - 1.  The surface is taken from
 * $Surface when the dialect carries it, otherwise found as in IglUtils::findSurfaceTris (:203-233) in (tet, local face) order
 * (the reference's order is that of a std::unordered_map).  Column-major outputs like the rest of this header. */
typedef struct ipcgpu_tetmesh ipcgpu_tetmesh;
int ipcgpu_read_tet_mesh(const char* path, ipcgpu_tetmesh** mesh, int* nV, int* nT, int* nSF);
int ipcgpu_tet_mesh_get(const ipcgpu_tetmesh*, double* V_colmajor, int* T_colmajor, int* SF_colmajor);
void ipcgpu_tet_mesh_free(ipcgpu_tetmesh*);
/* IglUtils::saveTetMesh (src/Utils/IglUtils.cpp:300-361): MSH 4.1 ASCII + $Surface */
int ipcgpu_save_tet_mesh(const char* path, int nV, int nT, const double* V_colmajor, const int* T_colmajor);

/* ---- mesh = the Mesh<3> data contract ----------------------------------------------- */
/* Replaces Mesh::computeFeatures + computeMassMatrix + setLameParam
 * (src/Mesh.cpp:414-527, 246-266, 399-401, 660-671): restTriInv, triArea, lumped mass,
 * vNeighbor and the Lame parameters are derived here from the rest shape. */
int ipcgpu_set_mesh(ipcgpu_ctx*, int nV, int nT, const double* V_rest_colmajor,
    const int* F_colmajor, double youngs_modulus, double poisson_ratio, double density);
/* vertexDBCType (Mesh.hpp:131-144) */
int ipcgpu_set_dbc(ipcgpu_ctx*, int n, const int* vert_ids, int dbc_type);
int ipcgpu_clear_dbc(ipcgpu_ctx*);
/* one `componentMaterial` entry of Mesh::setLameParam (Mesh.cpp:661-671; `shapes ... material rho E nu` in the scene script):
 * nodes [nodeBegin, nodeEnd) get density rho, tets [tetBegin, tetEnd) get (YM, PR).  Call after ipcgpu_set_mesh. */
int ipcgpu_set_component_material(ipcgpu_ctx*, int nodeBegin, int nodeEnd, int tetBegin, int tetEnd, double rho, double YM, double PR);
/* Config `energy NH|FCR` (src/Config.cpp:23-24,107-111; selects NeoHookeanEnergy or FixedCoRotEnergy in main.cpp): 0 = NH,
 * 1 = FCR (src/Energy/Physics_Elasticity/FixedCoRotEnergy.cpp:62-153).  FCR has no element-inversion safeguard
 * (Energy<dim>(false)): inversion checks and the injective step filter are skipped as in Optimizer.cpp:252,517,545,2710. */
int ipcgpu_set_energy_type(ipcgpu_ctx*, int energy_type);
/* Mesh::V (current positions) */
int ipcgpu_set_positions(ipcgpu_ctx*, const double* V_colmajor);
int ipcgpu_get_positions(ipcgpu_ctx*, double* V_colmajor);
/* Optimizer::xTilta (Optimizer.cpp:1236-1257) */
int ipcgpu_set_xtilde(ipcgpu_ctx*, const double* xTilta_colmajor);
/* read back derived per-element / per-node data (any pointer may be NULL) */
int ipcgpu_get_features(ipcgpu_ctx*, double* restTriInv_9nT, double* triArea_nT, double* mass_nV,
    double* mu_nT, double* lambda_nT);
/* nV = nT = 0 until ipcgpu_set_mesh was called */
int ipcgpu_get_mesh_dims(ipcgpu_ctx*, int* nV, int* nT);
/* Hand over the arrays Mesh<3> already holds instead of having them re-derived from (V_rest, F, E, nu, rho): restTriInv
   (Mesh.hpp:163, one 3x3 per tet, column-major inside the 9), triArea (:151), the diagonal of massMatrix (:148, lumped),
   u / lambda (:150, per tet).  Any pointer may be NULL (that array keeps what ipcgpu_set_mesh computed).  Call after
   ipcgpu_set_mesh; lets per-element state the reference modified after construction (component materials, rescaled
   masses) reach the device unchanged. */
int ipcgpu_set_mesh_features(ipcgpu_ctx*, const double* restTriInv_9nT, const double* triArea_nT, const double* mass_nV,
    const double* mu_nT, const double* lambda_nT);
/* Mesh::checkInversion (Mesh.cpp:715-764): *ok = 1 when no element has det < 0 */
int ipcgpu_check_inversion(ipcgpu_ctx*, int* ok);

/* ---- Energy<3> plugin (NeoHookeanEnergy) -------------------------------------------- */
/* Energy::computeEnergyVal (Energy.hpp:42, Energy.cpp:195-242): E = coef * sum_t vol_t psi(F_t) */
int ipcgpu_elastic_energy(ipcgpu_ctx*, double coef, double* energy);
/* Energy::getEnergyValPerElemBySVD (Energy.hpp:66) */
int ipcgpu_elastic_energy_per_elem(ipcgpu_ctx*, double* perElem_nT);
/* Energy::computeGradient (Energy.hpp:47, Energy.cpp:245-289) */
int ipcgpu_elastic_gradient(ipcgpu_ctx*, double coef, int projectDBC, double* grad_3nV);
/* Energy::computeHessian (Energy.hpp:53, Energy.cpp:292-331) into the context's CSR values:
 * a += elastic Hessian (projectSPD, vInd sign convention of Energy.cpp:402-407) */
int ipcgpu_elastic_hessian_add(ipcgpu_ctx*, double coef, int projectDBC);
/* Energy::filterStepSize (Energy.hpp:129, Energy.cpp:565-581), slackness 0.2, tol 1e-6 */
int ipcgpu_filter_step_size(ipcgpu_ctx*, const double* searchDir_3nV, double* stepSize_inout);

/* ---- LinSysSolver plugin ------------------------------------------------------------ */
/* LinSysSolver::set_pattern(vNeighbor, fixedVert) (LinSysSolver.hpp:46-150).  extra_pairs adds
 * contact connectivity on top of the mesh's vNeighbor (augmentConnectivity,
 * SelfCollisionHandler.cpp:330-415); pass n_extra = 0 for the mesh pattern. */
int ipcgpu_linsys_set_pattern(ipcgpu_ctx*, int n_extra, const int* extra_pairs_2n);
/* LinSysSolver::set_pattern on a caller-provided 0-based symmetric-upper CSR (get_ia/get_ja) */
int ipcgpu_linsys_set_pattern_csr(ipcgpu_ctx*, int n_rows, const int* ia, const int* ja);
int ipcgpu_linsys_get_dims(ipcgpu_ctx*, int* n_rows, int* nnz); /* getNumRows / getNumNonzeros */
int ipcgpu_linsys_get_pattern(ipcgpu_ctx*, int* ia, int* ja); /* get_ia / get_ja */
int ipcgpu_linsys_set_zero(ipcgpu_ctx*); /* setZero, :348 */
int ipcgpu_linsys_get_values(ipcgpu_ctx*, double* a); /* get_a, :465 */
int ipcgpu_linsys_set_values(ipcgpu_ctx*, const double* a);
/* Batched form of the per-entry calls below for an adapter that keeps LinSysSolver's host-side addCoeff / setCoeff calls
   (LinSysSolver.hpp:331-348, 402-410) but owns the values in HBM: for every entry k of the CSR
   a[k] = (isSet && isSet[k]) ? setVal[k] + delta[k] : a[k] + delta[k].  isSet / setVal may be NULL (pure accumulate). */
int ipcgpu_linsys_apply_host_updates(ipcgpu_ctx*, const double* delta_nnz, const unsigned char* isSet_nnz, const double* setVal_nnz);
int ipcgpu_linsys_add_coeff(ipcgpu_ctx*, int row, int col, double v); /* addCoeff :402 (row>col ignored) */
int ipcgpu_linsys_set_coeff(ipcgpu_ctx*, int row, int col, double v); /* setCoeff :331 */
int ipcgpu_linsys_multiply(ipcgpu_ctx*, const double* x, double* Ax); /* multiply :238 */
int ipcgpu_linsys_analyze_pattern(ipcgpu_ctx*); /* analyze_pattern, CHOLMODSolver.cpp:123-128 */
int ipcgpu_linsys_factorize(ipcgpu_ctx*); /* factorize, :130-137; returns IPCGPU_NOT_PD */
int ipcgpu_linsys_solve(ipcgpu_ctx*, const double* rhs, double* result); /* solve, :139-154 */
int ipcgpu_linsys_precondition_diag(ipcgpu_ctx*, const double* in, double* out); /* :411-420 */
/* Multi-GPU direct solver (one process per GPU): the assembly tree is cut below its top separators; rank r factorises and solves
   the subtrees it owns; a front above the cut is executed by ONE rank (the one that holds its most expensive child), and the update
   matrices / vectors of children on other ranks and the solution entries of ancestors travel point to point through the exchange hook
   (ipcgpu_opt_set_exchange[_stream] below: set it first, together with an all-reduce hook for the pivot flag and the solution vector).
   A rank needs the matrix values of the fronts it executes: replicated assembly, or ipcgpu_ctx_set_shard (owner-computes rows, see
   ipcgpu_opt_comm_stats below).  Takes effect at the next analyze_pattern.  ipcgpu_linsys_shard_stats (after analyze_pattern):
   out2[0] = world size, out2[1] = the share of the factorisation flops that lies above the cut (executed once each, on the chain of the cut's levels). */
int ipcgpu_linsys_set_shard(ipcgpu_ctx*, int rank, int world_size);
int ipcgpu_linsys_shard_stats(ipcgpu_ctx*, double* out2);
/* out4 = bytes THIS rank sent, bytes it received point to point through the solver's exchange hook so far, number of collective / group calls, and the
 * milliseconds this rank's stream spent inside those groups (HIP events around each: a rank that executes nothing at a level of the cut waits in its receive) */
int ipcgpu_linsys_exchange_stats(ipcgpu_ctx*, double* out4);
/* inputs of the strong-scaling model that bench.py prints beside its measured N-GPU value, after analyze_pattern: out5 = { dependent
 * 32-column pivot steps on the critical path of the assembly tree (per level the widest front's steps, summed), the same over the fronts ABOVE the cut of
 * ipcgpu_linsys_set_shard only, the largest rank's share of the flops below the cut (1 / world when balanced), levels, levels holding a front above the cut }.
 * No counterpart in the reference: CHOLMOD (CHOLMODSolver.cpp:118-154) is single-process. */
int ipcgpu_linsys_critical_path(ipcgpu_ctx*, double* out5);
/* Owner-computes sharding (round 4; SURVEY.md 8e, the north star's "RCCL all-reduce of the shared-node gradient / Hessian rows"): a context that has BOTH
   ipcgpu_ctx_set_shard and ipcgpu_linsys_set_shard (same world) assembles, per rank, exactly the CSR rows its fronts read -- the rows of the nodes its
   subtrees eliminate plus the separator rows above the cut, which every rank repeats (elements and contact stencils on a cut are evaluated by both sides) --
   so NO matrix value crosses ranks: the nodal gradient (one all-reduce, every node contributed by one rank), scalars, and the solver's update matrices /
   vectors / solution do.  ipcgpu_opt_comm_stats: out6 = bytes and calls all-reduced by the time stepper, by the solver, then the number of nodes whose
   rows this rank assembles and the number of nodes; ipcgpu_opt_complete_matrix: the rare consumer of the WHOLE matrix (ipcgpu_linsys_get_a / multiply on
   such a context) sums the rows over the ranks first.  IPCGPU_NO_OWNER_COMPUTES=1 keeps the older all-reduce of the values. */
int ipcgpu_opt_comm_stats(ipcgpu_ctx*, double* out6);
int ipcgpu_opt_complete_matrix(ipcgpu_ctx*);
/* diagnosis / tests (multifrontal solver, after analyze_pattern): for every entry k of the CSR pattern (ipcgpu_linsys_get_pattern order) the offset of its slot in
 * the front buffer, as the device kernel of the set-up computed it -- the same numbers mf_entry_destinations (ipc_amd/csrc/mf_symbolic.cpp) computes on the host */
int ipcgpu_linsys_entry_destinations(ipcgpu_ctx*, long long* dst_nnz);
/* The two parameters of the multifrontal factorisation a caller may set (every other one is fixed; their sweeps are under profiles/): levels of the assembly
 * tree whose 32-column step launches move at least bulk_min_mb MB of own columns factor them in outer blocks of bulk_block columns (two-level blocking;
 * defaults 48 MB / 256, reached by meshes beyond ~200 K nodes).  Takes effect at the next ipcgpu_linsys_analyze_pattern.  No counterpart in the reference
 * (CHOLMOD's supernodal blocking is internal, CHOLMODSolver.cpp:118-131); exists so that tests can force the path at test sizes. */
int ipcgpu_linsys_set_tuning(ipcgpu_ctx*, double bulk_min_mb, int bulk_block);
/* factor statistics: nnz(L), factorisation flops, number of supernodes / levels */
int ipcgpu_linsys_stats(ipcgpu_ctx*, double* stats4);

/* ---- Optimizer<3> building blocks --------------------------------------------------- */
/* setZero + elastic Hessian + mass / DBC diagonal: computePrecondMtr without contact
 * (Optimizer.cpp:3549-3668).  Fused with the gradient when grad_3nV != NULL
 * (computeGradient, Optimizer.cpp:3409-3450: elasticity + m (x - xTilta)). */
int ipcgpu_assemble_newton(ipcgpu_ctx*, double dtSq, int projectDBC, double* grad_3nV /*nullable*/);
/* computeEnergyVal (Optimizer.cpp:3199-3239): dtSq * elastic + 1/2 m |x - xTilta|^2 */
int ipcgpu_incremental_potential(ipcgpu_ctx*, double dtSq, double* energy);
/* computeGradient alone */
int ipcgpu_gradient(ipcgpu_ctx*, double dtSq, int projectDBC, double* grad_3nV);

/* ---- SelfCollisionHandler<3> (barrier contact) ------------------------------------- */
/* Mesh::SF (column-major nSF x 3 surface triangles, outward).  SVI and SFEdges are derived exactly as
 * Mesh::computeFeatures / computeBoundaryVert do (src/Mesh.cpp:495-515, 890-930), so edge and vertex indices
 * agree with the reference's. */
int ipcgpu_set_surface(ipcgpu_ctx*, int nSF, const int* SF_colmajor);
/* The same with the codimensional parts of Mesh<3> (`.seg` / `.pt` shapes, main.cpp:957-1005): CE = Mesh::CE, nCE node pairs.  The
 * segments enter vNeighbor and -- behind the triangles' edges, in the order given -- SFEdges, their ends SVI (Mesh.cpp:490-493, 513-515,
 * 912-915); a node that has no neighbour at all (no tetrahedron, triangle or segment: a `.pt` point) is a surface vertex as well
 * (:916-920) and is tested against every tetrahedron by the intersection check (SelfCollisionHandler.cpp:3301-3338).  Their masses:
 * ipcgpu_set_codim_nodes (segments: density * l^3 pi / 12 per end, Mesh.cpp:279-295; points: the mean nodal mass, :405-411). */
int ipcgpu_set_surface_codim(ipcgpu_ctx*, int nSF, const int* SF_colmajor, int nCE, const int* CE_pairs);
/* A build of the reference with USE_PREDICATES (CMakeLists.txt:137-140; not what the shipped CMake configures) decides the plane-side tests
 * of IglUtils::segTriIntersect (IglUtils.hpp:222-233) and IglUtils::pointInsideTetrahedron (:280-294) with the exact orient3d of
 * igl::predicates instead of floating-point products.  on != 0: the intersection checks do the same (orient3d_exact.h: Shewchuk's predicate
 * restated -- floating-point filter, exact expansion arithmetic behind it).  Default off = the default build. */
int ipcgpu_set_exact_predicates(ipcgpu_ctx*, int on);
int ipcgpu_get_surface(ipcgpu_ctx*, int* counts3 /*nSVI,nSF,nSFEdges*/, int* SVI /*nullable*/, int* SFEdges_2n /*nullable*/);
/* Kinematic mesh obstacles (MeshCO, src/CollisionObject/MeshCO.cpp:37-80; `meshCO` script keyword, Config.cpp:448-474): the
   obstacle rides along as a surface-only component of the mesh handed to ipcgpu_set_mesh -- extra nodes that belong to no
   tetrahedron (no mass, Dirichlet: ipcgpu_set_dbc), its triangles in the surface of ipcgpu_set_surface.  This call marks those
   nodes.  obstacle_only != 0: the contact sets, CCD and intersection checks keep only primitive pairs that involve an obstacle
   (a scene with `meshCO` but `selfCollisionOff`: Optimizer.cpp:2448-2470 asks only the collision objects).  Bounding box and
   mean nodal mass (dHat, kappa) are taken over element nodes, so the obstacle does not change them.  Call after set_surface. */
int ipcgpu_set_obstacle_nodes(ipcgpu_ctx*, int n, const int* vert_ids, int obstacle_only);
/* Surface-only ("codimensional") components of the simulated mesh: nodes of no tetrahedron that nevertheless belong to Mesh<3> --
 * the triangle meshes listed under `shapes` (componentCoDim 2; src/main.cpp, src/Mesh.cpp:310-345).  Unlike the nodes of a MeshCO
 * (ipcgpu_set_obstacle_nodes), which the reference keeps outside the mesh, they carry lumped masses: density x a third of the areas
 * of the adjacent triangles.  Like a MeshCO they stay OUT of the bounding box behind dHat / eps_v / the Newton tolerance and out of
 * the mean nodal mass behind kappa: the Optimizer sizes those with matSpaceBBoxSize2(dim) / avgNodeMass(dim), i.e. over the
 * tetrahedral components only (Mesh.cpp:576-637).  The scripts
 * fix or move them as Dirichlet nodes (`script DCOFix`: ipcgpu_set_dbc(ids, NONZERO)).  Call after ipcgpu_set_mesh, before
 * ipcgpu_opt_init / ipcgpu_opt_enable_self_collision. */
int ipcgpu_set_codim_nodes(ipcgpu_ctx*, int n, const int* node_ids, const double* node_mass);
/* SelfCollisionHandler::computeConstraintSet (SelfCollisionHandler.cpp:2149-2478) at the current positions:
 * MMActiveSet (PP / PE duplicates merged, multiplicity in slot 3), paraEEMMCVIDSet + paraEEeIeJSet, and the
 * candidate list for the partial CCD.  counts3 = {nActive, nParaEE, nCandidates}. */
int ipcgpu_contact_build(ipcgpu_ctx*, double dHat, int* counts3);
/* the sizes of the sets the library holds since the last ipcgpu_contact_build / _set, counts3 as above (an adapter checks with it that the candidate list it is
 * handed -- MMActiveSet_CCD of Optimizer.cpp:1926 -- is the one the library will sweep in ipcgpu_ccd_partial) */
int ipcgpu_contact_counts(ipcgpu_ctx*, int* counts3);
int ipcgpu_contact_get(ipcgpu_ctx*, int* active_4n, int* paraEE_4n, int* paraEEeIeJ_2n, int* csPTEE_2n /*nullable*/);
/* hand an active set in (adapters that keep the reference's own constraint-set code; tests) */
int ipcgpu_contact_set(ipcgpu_ctx*, int nActive, const int* active_4n, int nParaEE, const int* paraEE_4n, const int* paraEEeIeJ_2n);
/* kappa * (sum mult b(d) + sum e(c) b(d))  -- the barrier part of computeEnergyVal (Optimizer.cpp:3252-3353) */
int ipcgpu_contact_energy(ipcgpu_ctx*, double dHat, double kappa, double* energy);
/* The reference's PER-CONSTRAINT interface, on caller-held MMCVID tuples at the positions of ipcgpu_set_positions (round 5: what the compiled collision-handler
 * adapter include/adapters/HipSelfCollisionHandler.hpp forwards to when the reference's own Optimizer.cpp drives the loop):
 * evaluateConstraints (SelfCollisionHandler.cpp:37-81): val[i] = squared distance of tuple i;
 * leftMultiplyConstraintJacobianT (:84-148): out += coef * multiplicity_i * input[i] * grad d_i (multiplicity = -MMCVID[3] of a PP / PE tuple, 1 otherwise). */
int ipcgpu_contact_evaluate(ipcgpu_ctx*, int n, const int* mmcvid_4n, double* val_n);
int ipcgpu_contact_jt_multiply(ipcgpu_ctx*, int n, const int* mmcvid_4n, const double* input_n, double coef, double* out_3nV_inout);
/* grad += kappa J^T b'  (leftMultiplyConstraintJacobianT + augmentParaEEGradient, SelfCollisionHandler.cpp:84-148,
 * 2990-3036), then rows of projected Dirichlet nodes zeroed (Optimizer.cpp:3512-3516) */
int ipcgpu_contact_gradient_add(ipcgpu_ctx*, double dHat, double kappa, int projectDBC, double* grad_3nV_inout);
/* a += PSD-projected barrier Hessians (augmentIPHessian + augmentParaEEHessian, SelfCollisionHandler.cpp:418-561,
 * 3039-3201).  The pattern must already contain the contact connectivity (ipcgpu_contact_connectivity ->
 * ipcgpu_linsys_set_pattern); otherwise IPCGPU_ERR_STATE. */
int ipcgpu_contact_hessian_add(ipcgpu_ctx*, double dHat, double kappa, int projectDBC);
/* augmentConnectivity (SelfCollisionHandler.cpp:330-415): node pairs (a < b) coupled by the barrier Hessians */
int ipcgpu_contact_connectivity(ipcgpu_ctx*, int capacity, int* pairs_2n, int* nPairs);
/* Conservative CCD step bounds (the role of SelfCollisionHandler::largestFeasibleStepSize, :564-686, on the candidate
 * list of the last contact_build, and of largestFeasibleStepSize_CCD, :982-1366, on every pair whose swept boxes
 * overlap).  Contract (CTCD itself is un-vendored): the returned step keeps every tested pair at
 * >= (1 - slackness) of its current distance.  pair2 = limiting pair, (-svI-1, sfI) or (eI, eJ); (0,0) if none. */
int ipcgpu_ccd_partial(ipcgpu_ctx*, const double* searchDir_3nV, double slackness, double* stepSize_inout, int* pair2);
int ipcgpu_ccd_full(ipcgpu_ctx*, const double* searchDir_3nV, double slackness, double* stepSize_inout, int* pair2, int* nCandidates);
/* The full sweep exactly as the reference runs it (largestFeasibleStepSize_CCD, SelfCollisionHandler.cpp:982-1366, over the swept
 * SpatialHash::build, SpatialHash.hpp:589-832, whose step-size argument is a reference): (1) the hash caps the step so that the mean
 * |component| of the search direction over the surface nodes, times the step, stays below its cell size avgEdgeLen / 3 -- returned in
 * alphaCapped; (2) a surface vertex is swept against the surface vertices (svJ > svI), the edges and the triangles that share a cell
 * with it, an edge against the edges (eJ > eI) that share a cell and whose swept boxes overlap; (3) each pair with the safety
 * distance (1 - slackness) * its own current distance, asked again without it when it reports t < 1e-6.  The per-pair query is the
 * conservative advancement of ipcgpu_ccd_full (CTCD is un-vendored).  arg3 = limiting pair (kind, i, j): 0 (svI, svJ), 1 (svI, eI),
 * 2 (svI, sfI), 3 (eI, eJ); (-1,-1,-1) if none.  This is what the time stepper uses (ipcgpu_set_ccd_mode 1, the default);
 * mode 0 keeps the swept-box sweep over point-triangle / edge-edge pairs of ipcgpu_ccd_full. */
int ipcgpu_ccd_full_reference(ipcgpu_ctx*, const double* searchDir_3nV, double slackness, double* stepSize_inout, double* alphaCapped, int* arg3,
    int* nCandidates);
int ipcgpu_set_ccd_mode(ipcgpu_ctx*, int mode);
/* isIntersected / checkEdgeTriIntersectionIfAny (Optimizer.cpp:2626-2659, SelfCollisionHandler.cpp:3255-3300) */
int ipcgpu_is_intersected(ipcgpu_ctx*, int* flag);

/* ---- Optimizer<3>: the time stepper itself, state resident on the GPU --------------- */
/* Optimizer ctor + setTime (Optimizer.cpp:97-115, 418-430).  Uses the mesh of this context. */
int ipcgpu_opt_init(ipcgpu_ctx*, double dt, int withGravity);
int ipcgpu_opt_set_rel_tol(ipcgpu_ctx*, double relTol); /* setRelGL2Tol, Optimizer.cpp:390-396 */
/* `script twist` (AnimScripter.cpp:555-572, 1674-1684): two handle sets rotating about x */
int ipcgpu_opt_set_twist(ipcgpu_ctx*, int nLeft, const int* left, int nRight, const int* right, double angVel);
/* `selfCollisionOn` with the interior-point solver (Config.cpp:41-45, Optimizer.cpp:1534-1550): needs ipcgpu_set_surface;
 * dHat = dHatEps^2 * bboxDiag^2 (`dHat` keyword, default 1e-3).  From then on the stepper builds constraint sets, adds the
 * barrier terms, adapts kappa and bounds every step by CCD exactly as fullyImplicit_IP / solveSub_IP do. */
int ipcgpu_opt_enable_self_collision(ipcgpu_ctx*, double dHatEps);
/* Look-ahead of the contact pattern (no counterpart in the reference, which rebuilds pattern + symbolic analysis whenever the contact graph changes,
 * Optimizer.cpp:3570-3592; here the pattern only grows and a new analysis is needed when a pair shows up that the pattern lacks): on such a change the new
 * pattern takes the full stencils of the candidate set at `pad` x dHat (squared distances), blocks of pairs that are not active yet hold explicit zeros.
 * pad < 1: exactly the live pairs; 1: full stencils of the current candidates; > 1: fewer analyses, more fill in the factor.  Default (never called): 4 (twice
 * the distance) for meshes up to 150 K nodes, 2.25 beyond -- measured on the bench's contact workloads, profiles/r06_pattern_lookahead_ab.txt.
 * The Newton iterates do not depend on it (explicit zeros); what it trades is host-side analyses against factorisation flops. */
int ipcgpu_opt_set_pattern_lookahead(ipcgpu_ctx*, double pad);
/* analytic half-space obstacle (`ground` / `halfSpace` keywords, Config.cpp:306-345; HalfSpace.cpp:41-85): points with
 * normal.(x - origin) > 0 are free.  Needs ipcgpu_set_surface.  Returns its index in *id.  The stepper then builds its
 * vertex constraint set (CollisionObject.h:323-351), adds barrier energy / gradient / diagonal Hessian blocks
 * (HalfSpace.cpp:106-214) and bounds each step by the ray test with slackness 0.9 (HalfSpace.cpp:242-269). */
int ipcgpu_opt_add_half_space(ipcgpu_ctx*, const double* origin3, const double* normal3, double dHatEps, int* id);
/* the same pieces one by one (adapter for a HalfSpace<3> subclass).  verts = activeSet[coI], ascending surface order. */
int ipcgpu_halfspace_build(ipcgpu_ctx*, int id, double dHat, int cap, int* verts, int* n);
int ipcgpu_halfspace_set(ipcgpu_ctx*, int id, int n, const int* verts);
int ipcgpu_halfspace_energy(ipcgpu_ctx*, int id, double dHat, double kappa, double* E);
int ipcgpu_halfspace_gradient_add(ipcgpu_ctx*, int id, double dHat, double kappa, double* grad_3nV_inout);
int ipcgpu_halfspace_hessian_add(ipcgpu_ctx*, int id, double dHat, double kappa, int projectDBC);
/* HalfSpace::move (HalfSpace.cpp:389-416), what the ACO* scripts of AnimScripter call before a time step (AnimScripter.cpp:1832-1890): the plane's
 * origin moves along delta by the largest fraction <= 1 that keeps `slackness` of the distance of every surface node (Dirichlet nodes included);
 * stepSizeLeft = 1 - that fraction. */
int ipcgpu_halfspace_move(ipcgpu_ctx*, int id, const double* delta3, double slackness, double* stepSizeLeft);
int ipcgpu_halfspace_step_bound(ipcgpu_ctx*, int id, const double* searchDir_3nV, double slackness, double* stepSize_inout);
/* ---- lagged smoothed Coulomb friction (FrictionUtils.hpp, SelfCollisionHandler.cpp:2481-2988, HalfSpace.cpp:272-381)
 * `selfFric mu` / `fricIterAmt n` / eps_v = tuning[4] (Config.cpp:482-488, 550-551, 45).  Self friction needs self collision. */
/* NOTE on eps_v: the smoothing distance the friction terms use is eps_v^2 * h^2 * (bounding-box diagonal)^2 with h = 0.025 -- the step size of the
 * setTime(10.0, 0.025) inside the reference's Optimizer constructor (Optimizer.cpp:116, 290-303) -- NOT the dt of ipcgpu_opt_init: the reference sets
 * the scene's dt afterwards (main.cpp:1398) without recomputing it, and this library follows the reference (CN_MBC of :268 likewise). */
int ipcgpu_opt_set_friction(ipcgpu_ctx*, double selfFric, int fricIterAmt, double epsV);
/* eps_v homotopy: `tuning`'s sixth entry (Config.cpp:41-45, Optimizer.cpp:296-303).  The smoothing distance of the friction terms starts every
   time step at eps_v (fifth entry) and is halved down -- or clamped up -- to eps_v_target between the friction-lag passes (:1776-1781); the
   tangent-space convergence test runs only once it has arrived (:1717).  <= 0: the target is eps_v itself (what the `epsv` keyword sets). */
int ipcgpu_opt_set_friction_target(ipcgpu_ctx*, double eps_v_target);
/* The step size h inside eps_v^2 h^2 (fricDHat0 / fricDHatTarget) and CN_MBC.  In the reference these are evaluated in the Optimizer constructor right after its
   setTime(10.0, 0.025) (Optimizer.cpp:116, 268, 290-303) and never again when main.cpp:1398 sets the scene's dt: they carry h = 0.025 whatever `time` says.  That
   is the default here; pass the scene's dt for the paper's semantics.  Call before ipcgpu_opt_init / ipcgpu_opt_precompute. */
int ipcgpu_opt_set_constructor_dt(ipcgpu_ctx*, double h);
/* The three remaining knobs of the scene file behind the interior-point lengths (Config.cpp:553-558, Config.hpp:138-139):
 *   useAbsParameters   dHat, its homotopy target, dTol, eps_v (and its target) and the Newton tolerance are ABSOLUTE lengths instead of fractions
 *                      of the rest-shape bounding-box diagonal (Optimizer.cpp:107-109, 279-302, 1535-1537, 2941-2945); suggestKappa's
 *                      distances and CN_MBC stay relative there too (:268, 2228-2233)
 *   dTolRel            tuning[3] (1e-9): below dTol = dTolRel^2 (x diagonal^2) a converged distance ends the homotopy / the close-pair
 *                      bookkeeping counts a stencil (:102-109, 1710, 1744, 2409-2434)
 *   kappaMinMultiplier 1e11 (`kappaMinMultiplier` / `minBarrierStiffnessScale`): numerator of suggestKappa and, x 100, of upperBoundKappa
 * Call it after the collision objects are registered or before -- the derived lengths are recomputed. */
int ipcgpu_opt_set_parameter_scaling(ipcgpu_ctx*, int useAbsParameters, double dTolRel, double kappaMinMultiplier);
/* MeshCO::friction (Config.cpp:459-474, MeshCO.cpp) beside Config::selfFric: a kinematic mesh obstacle carries its own friction
 * coefficient for the pairs that involve it.  Pass the larger coefficient to ipcgpu_opt_set_friction and the ratios here: the lagged
 * normal forces (MMLambda_lastH) of stencils without / with an obstacle node are multiplied by scaleSelf / scaleObstacle. */
int ipcgpu_opt_set_friction_scales(ipcgpu_ctx*, double scaleSelf, double scaleObstacle);
/* Optimizer.cpp:146-166: solveFric is also true when a mesh collision object carries a friction coefficient, although no friction
 * term is ever evaluated for its pairs (:3357-3376, 3473-3510, 3676-3705): the lagging loop with its tangent-space convergence solve
 * then runs after every converged sub-problem.  on != 0 switches that loop on without any frictional pair. */
int ipcgpu_opt_force_friction_loop(ipcgpu_ctx*, int on);
int ipcgpu_opt_set_half_space_friction(ipcgpu_ctx*, int id, double mu); /* CollisionObject::friction of half-space `id` */
/* Lagged stiffness-proportional damping: `dampingStiff s` (Config.cpp:141-147; `dampingRatio r` is s = r * dt^3 * 3 / 4, :148-157,
 * 614-616).  The damping matrix is the PSD-projected elastic Hessian at the end of the last time step times s / dt
 * (Optimizer.cpp:593-595, 3723-3735); 1/2 dx^T D dx joins the energy (:3381-3400), D dx the gradient (:3519-3540), D the Hessian
 * (:3707-3709).  Call before ipcgpu_opt_precompute; 0 switches it off. */
int ipcgpu_opt_set_damping(ipcgpu_ctx*, double dampingStiff);
/* `tuning` entry 0 (Config.cpp:41-45, 533-541): the barrier stiffness every time step starts from, bounded from above by
 * upperBoundKappa and still raised by initKappa / the adaptive updates (Optimizer.cpp:1540-1550, 2216-2225); 0 = suggestKappa. */
int ipcgpu_opt_set_kappa(ipcgpu_ctx*, double kappa);
/* `tuning` entry 2 (relative, like dHat): the dHat homotopy of fullyImplicit_IP.  Every time step starts at dHat; after a converged
 * sub-problem whose largest active distance is not below the target, ipcgpu_opt_next_subproblem halves dHat (not below the
 * target), rebuilds the constraint sets and re-initialises kappa (Optimizer.cpp:283-289, 1706-1713, 1763-1774).  <= 0: target =
 * dHat, no homotopy (the default, and what `dHat x` / a 6-entry `tuning` with equal entries 1 and 2 mean). */
int ipcgpu_opt_set_dhat_target(ipcgpu_ctx*, double dHatTargetEps);
/* After ipcgpu_opt_newton_iter reported convergence: the tail of the fullyImplicit_IP loop body (Optimizer.cpp:1617-1790) --
 * refresh the lagged multipliers / tangent bases, test tangent-space convergence.  *more = 1: another solveSub_IP pass has
 * started (keep calling newton_iter); 0: the time step is done.  Without friction it returns 0 and changes nothing. */
int ipcgpu_opt_next_subproblem(ipcgpu_ctx*, int* more);
/* scalars4 = {fricDHat (eps_v^2 h^2, < 0: off), #lagged self-contact constraints, friction iteration, #lagged half-space
 * vertices}; lambda (nullable) receives MMLambda_lastH */
int ipcgpu_opt_get_friction_state(ipcgpu_ctx*, double* scalars4, double* lambda);
/* building blocks over the lagged self-contact set.  friction_update lags the CURRENT constraint set (ipcgpu_contact_build /
 * _set) at the current positions: multipliers (Optimizer.cpp:1586-1591), computeDistCoordAndTanBasis (:2481-2527).
 * Vt_colmajor = positions at the beginning of the time step (result.V_prev). */
int ipcgpu_friction_update(ipcgpu_ctx*, double dHat, double kappa, int* nLagged);
int ipcgpu_friction_get(ipcgpu_ctx*, double* lambda_n, double* coord_2n, double* basis_6n);
int ipcgpu_friction_energy(ipcgpu_ctx*, const double* Vt_colmajor, double eps2, double coef, double* energy);
int ipcgpu_friction_gradient_add(ipcgpu_ctx*, const double* Vt_colmajor, double eps2, double coef, double* grad_3nV_inout);
int ipcgpu_friction_hessian_add(ipcgpu_ctx*, const double* Vt_colmajor, double eps2, double coef, int projectDBC);
/* overwrite Optimizer::velocity (xyz-interleaved) and recompute xTilta (computeXTilta, Optimizer.cpp:1236-1257): the
 * `initVel` script keyword (Config.cpp:247-262) */
int ipcgpu_opt_set_velocity(ipcgpu_ctx*, const double* vel_3nV);
/* Config `timeIntegration BE | NM beta gamma` (src/Config.cpp:112-118, defaults beta = 0.25, gamma = 0.5 src/Config.hpp:96):
 * type 0 = backward Euler, 1 = Newmark.  Newmark scales the elastic terms by dt^2 beta (Optimizer.cpp:3216-3224, 3427-3434,
 * 3627-3631), predicts xTilta with the stored acceleration (:1259-1277) and updates velocity / acceleration at the end of the
 * time step (:582-590).  Call after ipcgpu_opt_init, before ipcgpu_opt_precompute; the acceleration starts at zero (:177). */
int ipcgpu_opt_set_time_integration(ipcgpu_ctx*, int type, double beta, double gamma);
/* Config `warmStart n` -> Optimizer::initX(n) (Optimizer.cpp:925-1215): first iterate of every time step.  0 = the last
   configuration (default), 1 explicit Euler, 2 xHat, 3 symplectic Euler, 4 uniformly accelerated motion, 5 the Jacobi guess -g_i / H_ii
   (gradient with projected, matrix with unprojected Dirichlet rows, :1082-1110); the step is cut by the
   inversion filter, the half-space bounds and a full CCD pass, then halved while the mesh is inverted or intersecting.
   *last_step (nullable) receives the fraction of the predicted displacement the last begin_timestep could take. */
int ipcgpu_opt_set_warm_start(ipcgpu_ctx*, int option);
int ipcgpu_opt_get_warm_step(ipcgpu_ctx*, double* last_step);
/* One Mesh::DirichletBCs entry (src/Mesh.hpp:23-39; `DBC bboxMin bboxMax linVel angVel [t0 t1]` on a shape line,
 * src/Config.cpp:246-263, vertices picked by IglUtils::Init_Dirichlet) or the scripted linear / angular velocity of a whole
 * component (`linearVelocity` / `angularVelocity`, componentLVels / componentAVels -- how kinematic mesh obstacles move):
 * while t0 <= stepStartTime < t1 the vertices are Dirichlet nodes (ZERO if both velocities vanish, else NONZERO,
 * AnimScripter.cpp:58-110) and every time step moves them by R (x - c) + c + linVel dt - x with R = Rx Ry Rz of ang_vel dt and
 * c the centre of their current bounding box (AnimScripter.cpp:1413-1462).  ang_vel in rad/s (the script gives deg/s).
 * Call after ipcgpu_opt_init and after the static ipcgpu_set_dbc / ipcgpu_opt_set_twist calls. */
int ipcgpu_opt_add_dirichlet(ipcgpu_ctx*, int n, const int* vert_ids, const double* lin_vel3, const double* ang_vel3, double t0, double t1);
/* Ends group `group` (0-based, in the order of the ipcgpu_opt_add_dirichlet calls) at time t_end: from the time step that starts at
 * t_end on its vertices are free again.  This is what the state-dependent scripts do with mesh.resetDBCVertices() inside
 * AnimScripter::stepAnimScript -- e.g. `script dragright` lets go of the handle once the body has been pulled past the
 * obstacles (AnimScripter.cpp:1619-1632); the condition is the caller's. */
int ipcgpu_opt_end_dirichlet(ipcgpu_ctx*, int group, double t_end);
/* The hard-coded scripted motions of AnimScripter::stepAnimScript that move whole components by a rule evaluated on the current state
 * (`script DCOSquash / DCOSquash6 / DCOSqueezeOut / DCORotCylinders / DCOVerschoorRoller`, AnimScripter.cpp:1060-1300 set-up, :1961-2135
 * per step): the caller evaluates the rule before a time step and hands the group's motion for that step over.  center3 != NULL: the
 * rotation turns about this fixed point (MCORotCenter, the component's bounding-box centre at set-up) instead of the centre of the
 * group's current bounding box; force_nonzero: the nodes are NONZERO Dirichlet nodes even while their velocity is zero, as those
 * scripts type them (it decides whether they are released with the others when a scripted move is cut short, Optimizer.cpp:2168-2203). */
int ipcgpu_opt_set_dirichlet_motion(ipcgpu_ctx*, int group, const double* lin_vel3, const double* ang_vel3, const double* center3 /*nullable*/,
    int force_nonzero);
/* Mesh-sequence motion of a Dirichlet group (`meshSeq <folder>` behind a shape, Config.cpp:284-289; AnimScripter.cpp:1465-1532): before
   every time step the reference reads <folder>/<step>.{msh,obj,seg,pt} and makes `file position - current position` the move of the
   component's nodes, overriding its velocities; the move is then bounded by CCD and the intersection check like any scripted motion
   (:2162-2250).  targets_3n = the file's positions (node order of the group, xyz interleaved), handed over before each step; NULL ends
   the sequence.  The nodes keep the type their velocities give them (ZERO for a codimensional component that only follows a sequence). */
int ipcgpu_opt_set_dirichlet_targets(ipcgpu_ctx*, int group, int n, const double* targets_3n);
/* One Mesh::NeumannBCs entry (src/Mesh.hpp:47-56; `NBC bboxMin bboxMax force [t0 t1]` on a shape line, src/Config.cpp:264-280):
 * while t0 <= stepStartTime < t1 every listed vertex that is not a Dirichlet node feels the acceleration `accel3` -- the
 * incremental potential gets -dt^2 m_v accel . x_v, the gradient -dt^2 m_v accel (Optimizer.cpp:3241-3250, 3452-3461). */
int ipcgpu_opt_add_neumann(ipcgpu_ctx*, int n, const int* vert_ids, const double* accel3, double t0, double t1);
/* State of the augmented-Lagrangian Dirichlet fallback (Optimizer.cpp:1826-1828, 2168-2203; AnimScripter.cpp:2280-2350): when
 * the scripted motion of a time step is cut short (element inversion, CCD, intersection), the NONZERO Dirichlet nodes are
 * released and pulled to their targets by a penalty rho_DBC / 2 m |x - target|^2 with multipliers until the completed step
 * size exceeds 1 - 1e-3.  out4 = {completed step size, rho_DBC, m_projectDBC, number of target positions}. */
int ipcgpu_opt_get_dbc_state(ipcgpu_ctx*, double* out4);
/* Optimizer::velocity (xyz-interleaved), acceleration and dx_Elastic = V - xTilta of the last finished time step
 * (Optimizer.cpp:574-586); any pointer may be null */
int ipcgpu_opt_get_kinematics(ipcgpu_ctx*, double* vel_3nV, double* acc_3nV, double* dx_elastic_3nV);
/* Optimizer::saveStatus (Optimizer.cpp:2964-3011) and the `restart <status file>` branch of the Optimizer constructor
 * (Optimizer.cpp:179-248, Config.cpp:513-516): the reference's text checkpoint (timestep / position / velocity / acceleration /
 * dx_Elastic), written with 20 significant digits so that doubles round-trip; files are interchangeable with the reference's.
 * Load after ipcgpu_opt_init (and ipcgpu_opt_set_time_integration), before ipcgpu_opt_precompute. */
int ipcgpu_opt_save_status(ipcgpu_ctx*, const char* path);
int ipcgpu_opt_load_status(ipcgpu_ctx*, const char* path);
/* counts6 = {#active, #paraEE, #CCD candidates, #half-space constraints, #full CCD passes, #pattern changes}; pair2 = limiting CCD pair of the
 * last iteration ((-svI-1, sfI) or (eI, eJ), (0,0) if none) */
int ipcgpu_opt_get_contact_state(ipcgpu_ctx*, int* counts6, int* pair2);
int ipcgpu_opt_precompute(ipcgpu_ctx*); /* precompute, Optimizer.cpp:457-507 */
int ipcgpu_opt_begin_timestep(ipcgpu_ctx*); /* solve(): stepAnimScript + fullyImplicit_IP head */
/* one pass of the solveSub_IP loop (Optimizer.cpp:1829-2204) == one "Newton iteration" of the
 * metric; *converged = 1 when the convergence test fired before any work was done */
int ipcgpu_opt_newton_iter(ipcgpu_ctx*, int* converged);
int ipcgpu_opt_end_timestep(ipcgpu_ctx*); /* BE velocity + xTilta update, Optimizer.cpp:570-580 */
int ipcgpu_opt_solve_timestep(ipcgpu_ctx*, int maxIter, int* nIter); /* Optimizer::solve(1) */
/* state readers (any pointer may be NULL); scalars8 = {lastEnergyVal, lastStepSize, targetGRes,
 * innerIterAmt, timestep, alphaFeasible, kappa, dHat} */
int ipcgpu_opt_get_state(ipcgpu_ctx*, double* V_colmajor, double* searchDir_3nV, double* gradient_3nV, double* scalars8);
/* timer_step buckets in seconds (src/main.cpp:1326-1340): 0 matrixComputation .. 14 computeConstraintSets */
int ipcgpu_opt_get_timers(ipcgpu_ctx*, double* t16);
/* multi-GPU: the caller-supplied reduction hook used by the optimizer for sharded assembly.
 * fn(user, buf_dev, count, op) must all-reduce `count` doubles in device memory in place
 * (op 0 = sum, 1 = min).  bench.py binds it to torch.distributed (RCCL). */
typedef int (*ipcgpu_allreduce_fn)(void* user, void* buf_dev, long long count, int op);
int ipcgpu_opt_set_allreduce(ipcgpu_ctx*, ipcgpu_allreduce_fn fn, void* user);
/* Stream-ordered variant, and the one a C / C++ caller uses: fn enqueues the all-reduce on the HIP stream it is handed (the
 * context's own) and returns -- `ncclAllReduce(buf, buf, count, ncclDouble, op ? ncclMin : ncclSum, comm, stream)` is the whole body.
 * The library then never synchronises with the host around a collective.  include/adapters/ipcgpu_rccl.cpp is that hook on RCCL
 * (libipcgpu_rccl.so: ipcgpu_rccl_unique_id / ipcgpu_rccl_attach); takes precedence over ipcgpu_opt_set_allreduce. */
typedef int (*ipcgpu_allreduce_stream_fn)(void* user, void* buf_dev, long long count, int op, void* hip_stream);
int ipcgpu_opt_set_allreduce_stream(ipcgpu_ctx*, ipcgpu_allreduce_stream_fn fn, void* user);
/* Point-to-point exchange of the sharded direct solver (round 5; ipcgpu_linsys_set_shard).  A front above the cut of the assembly tree is executed by
 * ONE rank; the packed update matrix of a child that another rank computed (factorisation), its update vector (forward sweep) and the solution entries
 * of an ancestor (backward sweep) go from the rank that has them to exactly the ranks that need them.  One call = ONE group of operations between which
 * no order may be assumed (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd; torch.distributed.batch_isend_irecv): op i moves `count` doubles of the
 * device buffer `buf_dev` to (send = 1) or from (send = 0) rank `peer`; several operations between one pair of ranks match in the order given.
 * Host-ordered variant: the library drains its stream before the call and expects the data in place when it returns.  Stream variant: the hook
 * enqueues on the stream it is handed (include/adapters/ipcgpu_rccl.cpp); takes precedence.  Set one of them before ipcgpu_linsys_set_shard with
 * world_size > 1.  (The all-reduce hooks above stay in use for scalars, the nodal gradient, the pivot flag and the solution vector.) */
typedef struct ipcgpu_p2p_op {
    void* buf_dev;
    long long count;
    int peer;
    int send;
} ipcgpu_p2p_op;
typedef int (*ipcgpu_exchange_fn)(void* user, int n_ops, const ipcgpu_p2p_op* ops);
typedef int (*ipcgpu_exchange_stream_fn)(void* user, int n_ops, const ipcgpu_p2p_op* ops, void* hip_stream);
int ipcgpu_opt_set_exchange(ipcgpu_ctx*, ipcgpu_exchange_fn fn, void* user);
int ipcgpu_opt_set_exchange_stream(ipcgpu_ctx*, ipcgpu_exchange_stream_fn fn, void* user);
int ipcgpu_ctx_get_stream(ipcgpu_ctx*, void** hip_stream);

/* ---- measurement -------------------------------------------------------------------- */
/* Launch the fused element-assembly kernel `reps` times on the context stream and return the
 * average duration per launch measured with HIP events on that stream, plus the algorithmic
 * bytes of one launch (112 nT + 84 nV + 8 nnz). */
int ipcgpu_bench_assembly(ipcgpu_ctx*, double dtSq, int reps, double* avg_ms, double* algorithmic_bytes);
/* same for one numeric factorisation + solve */
int ipcgpu_bench_factor_solve(ipcgpu_ctx*, int reps, double* factor_ms, double* solve_ms);
/* device streaming copy bandwidth (GB/s) measured here, the "STREAM" figure quoted beside the
 * nominal 8 TB/s */
int ipcgpu_bench_stream(ipcgpu_ctx*, long long bytes, int reps, double* gbps);

#ifdef __cplusplus
}
#endif
#endif
