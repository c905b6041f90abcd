// HipContact: the self-collision handler of the hot path (SelfCollisionHandler<3> statics,
// src/CollisionObject/SelfCollisionHandler.hpp:21-250) with its state in HBM.
//   computeConstraintSet      SelfCollisionHandler.cpp:2149-2478  -> buildConstraintSet (grid broad phase + typing on the GPU,
//                                                                   duplicate merge on the host like the reference's std::map)
//   evaluateConstraints + b   :38-81, Optimizer.cpp:3252-3353     -> energy
//   leftMultiplyConstraintJacobianT / augmentParaEEGradient  :84-148, 2990-3036   -> gradientAdd
//   augmentIPHessian / augmentParaEEHessian   :418-561, 3039-3201 -> hessianAdd
//   augmentConnectivity       :330-415                            -> connectivity
//   largestFeasibleStepSize[_CCD]  :564-686, 982-1366             -> ccdStepBound (conservative additive CCD)
#pragma once
#include <cstdlib>
#include "common.h"
#include <array>
#include <utility>
#include <functional>
#include <vector>

namespace ipcgpu {

class HipMesh;
class HipLinSysSolver;

class HipContact {
public:
    explicit HipContact(hipStream_t s) : stream(s)
    {
    }
    hipStream_t stream;
    // surface (Mesh::SF, SVI, SFEdges; Mesh.cpp:495-515, 890-930)
    int nSF = 0, nSVI = 0, nSFE = 0;
    std::vector<int> SF; // column-major nSF x 3
    std::vector<int> SVI;
    std::vector<std::pair<int, int>> SFEdges;
    DevBuf<int> d_SF, d_SVI, d_SFE; // d_SF: int[3 nSF] (t0 t1 t2 per triangle), d_SFE: int[2 nSFE]
    DevBuf<double> d_xRest; // rest positions xyz-interleaved (eps_x needs rest edge lengths)
    // sets
    // The sets live in HBM (d_active ... d_csPTEE with the counts below); the host vectors are mirrors that syncHost() fills
    // when somebody needs the tuples on the host (tests, the connectivity of a pattern change, friction lagging).
    std::vector<std::array<int, 4>> active, para;
    std::vector<std::array<int, 2>> paraEIEJ, csPTEE;
    DevBuf<int> d_active, d_para, d_paraEIEJ, d_csPTEE;
    int nActive() const { return nActive_; }
    int nPara() const { return nPara_; }
    int nCand() const { return nCand_; }
    void syncHost() const;
    bool surfaceSet = false;

    // CE: codimensional segments (`.seg` shapes, Mesh::CE) as node pairs; nodes without any neighbour in the mesh are codimensional points
    // (`.pt` shapes): both join SVI, the segments SFEdges (Mesh.cpp:490-515, 912-927)
    void setSurface(const HipMesh& mesh, int nSF, const int* SF_colmajor, int nCE = 0, const int* CE_pairs = nullptr);
    std::vector<int> codimPoints;
    bool exactPredicates = false; // the intersection checks as a USE_PREDICATES build of the reference makes them (IglUtils.hpp:222-233, 280-294)
    // kinematic obstacle nodes (the reference's MeshCO riding along as a surface-only component, MeshCO.cpp) and, for scenes with
    // `selfCollisionOff`, the filter that keeps only primitive pairs involving an obstacle
    void setObstacle(int nV, int n, const int* ids, bool obstacleOnly);
    bool hasObstacle = false, obstacleOnly = false;
    DevBuf<int> d_obst, d_pairFlags;
    const int* pairFlags(int nV, const int* dbc_dev);
    void setSets(int nA, const int* a4, int nP, const int* p4, const int* pe2);
    void uploadSets();
    // returns #active
    int buildConstraintSet(const HipMesh& mesh, const double* x_dev, const int* dbc_dev, double dHat);
    // Multi-GPU ("contact pairs assigned to the owner of ..."): the constraint SETS are the same on every rank (integer outputs every rank
    // needs for the pattern), their EVALUATION is split.  energy(): rank r takes the stencils [n r / W, n (r + 1) / W) of the two lists and all-reduces its
    // scalar through `shardReduce`.  gradientAdd / hessianAdd (round 4): the CALLER hands in a node mask -- a stencil is evaluated where one of its nodes owns
    // rows (HipOptimizer's owner-computes plan); with a null mask they evaluate the whole lists, whatever the context's shard (the C entry points do).
    int shardRank = 0, shardWorld = 1;
    std::function<void(double*, long long)> shardReduce;
    void shardRange(int n, int& b, int& e) const
    {
        b = (int)((long long)n * shardRank / shardWorld);
        e = (int)((long long)n * (shardRank + 1) / shardWorld);
    }
    double energy(const double* x_dev, double dHat, double kappa, DevBuf<double>& partial, double* scalar_dev);
    // the same without the read-back: the value is left in *scalar_dev on the stream (false: empty sets, nothing enqueued, the value is 0) -- the time stepper
    // reads it together with the elastic energy, one synchronisation for both
    // pubWords > 0: the reduction's own launch also copies pubWords 32-bit words pubSrc -> pubDst (mapped host memory); publishedByEnergy() says whether it did
    bool energyEnqueue(const double* x_dev, double dHat, double kappa, DevBuf<double>& partial, double* scalar_dev, const void* pubSrc = nullptr,
        void* pubDst = nullptr, int pubWords = 0);
    bool publishedByEnergy() const { return publishedByEnergy_; }
    bool publishedByEnergy_ = false;
    // useActive / usePara: initKappa leaves the mollified set out (Optimizer.cpp:2262-2270)
    void gradientAdd(const double* x_dev, const int* dbc_dev, int nV, double dHat, double kappa, int projectDBC, double* grad_dev, bool useActive = true,
        bool usePara = true, const unsigned char* need_dev = nullptr);
    // deferCheck: the "pair outside the pattern" flag is not waited for; the caller calls takeHessianError() behind its next synchronisation of the stream
    void hessianAdd(const double* x_dev, const int* dbc_dev, const HipLinSysSolver& lin, double dHat, double kappa, int projectDBC,
        double* a_dev, const unsigned char* need_dev = nullptr, bool deferCheck = false);
    void takeHessianError(); // throws what hessianAdd(..., deferCheck = true) would have thrown
    // the reference's per-constraint interface on host arrays of MMCVID tuples (SelfCollisionHandler.cpp:37-148; ipcgpu_contact_evaluate / _jt_multiply):
    // val[i] = squared distance of tuple i;  out += coef * multiplicity_i * input[i] * grad d_i
    void evaluateTuples(const double* x_dev, int n, const int* tuples4, double* val);
    void jtMultiplyTuples(const double* x_dev, int nV, int n, const int* tuples4, const double* input, double coef, double* out_3nV);
    DevBuf<int> tupleBuf_;
    DevBuf<double> tupleVal_, tupleOut_;
    void connectivity(std::vector<std::pair<int, int>>& pairs) const;
    bool patternCovers(const HipLinSysSolver& lin); // every node pair of the current sets has its block in lin's pattern (one small kernel)
    void candidateConnectivity(std::vector<std::pair<int, int>>& pairs) const; // appends; all node pairs of the candidate list
    void candidateConnectivitySorted(std::vector<std::pair<int, int>>& pairs); // the same, sorted and unique, formed on the device
    // conservative CCD step bounds; pair2 receives the limiting pair ((-svI-1, sfI) or (eI, eJ)); returns the new bound
    double ccdPartial(const double* x_dev, const double* p_dev, double slackness, double stepSize, int* pair2);
    // inversion filter (root already on the device, may be null) -> ccdPartial -> maxSurfaceSpeed with one synchronisation (the time stepper's step-size pipeline)
    void stepBounds(const double* x_dev, const double* p_dev, double slackness, double stepSize, const double* filterDev, double* alphaOut, int* pair2,
        double* pMaxOut);
    double ccdFull(const HipMesh& mesh, const double* x_dev, const double* p_dev, const int* dbc_dev, double slackness, double stepSize, int* pair2,
        int* nCand);
    // the full sweep as the reference runs it (SelfCollisionHandler.cpp:982-1366 over SpatialHash.hpp:589-832): the hash first caps the step
    // (alphaCapped), a surface vertex is swept against vertices / edges / triangles that share a cell with it, an edge against edges;
    // arg3 = (kind, i, j) of the limiting pair: K_PP (svI, svJ), K_PE (svI, eI), K_PT (svI, sfI), K_EE (eI, eJ)
    double ccdFullReference(const HipMesh& mesh, const double* x_dev, const double* p_dev, const int* dbc_dev, double slackness, double stepSize,
        double* alphaCapped, int* arg3, int* nCand);
    int ccdMode = 1; // 1: ccdFullReference in the time stepper; 0: the swept-box sweep of PT / EE pairs (ccdFull)
    bool isIntersected(const HipMesh& mesh, const double* x_dev, const int* dbc_dev);
    void evalStencils(const std::vector<std::array<int, 4>>& ids, const double* x_dev, std::vector<double>& d2);
    // lin != null: *covers = patternCovers(*lin), answered with the same synchronisation
    void closeStencils(const double* x_dev, double dTol, std::vector<std::array<int, 4>>& ids, std::vector<double>& d2, const HipLinSysSolver* lin = nullptr,
        int* covers = nullptr);
    unsigned long long setsVersion = 0; // bumped whenever the sets change (buildConstraintSet, uploadSets): what a cached coverage answer is stamped with
    double maxSurfaceSpeed(const double* p_dev); // max_{v in SVI} |p_v|  (CFL bound, Optimizer.cpp:1947-1953)
    // lagged friction of the self-contact set: MMActiveSet_lastH, MMLambda_lastH, MMDistCoord, MMTanBasis
    std::vector<std::array<int, 4>> fricSet;
    DevBuf<int> d_fricSet;
    DevBuf<double> d_fricLambda, d_fricCoord, d_fricBasis;
    double fricScaleSelf = 1.0, fricScaleObst = 1.0; // MeshCO::friction beside selfFric: factors on the lagged normal forces by stencil kind
    void frictionLagClear();
    void frictionLagUpdate(const double* x_dev, double dHat, double kappa); // lags the current `active` set (Optimizer.cpp:1578-1598)
    void frictionGet(double* lambda, double* coord2, double* basis6);
    double frictionEnergy(const double* x_dev, const double* xt_dev, double eps2, double coef, DevBuf<double>& partial, double* scalar_dev);
    void frictionGradientAdd(const double* x_dev, const double* xt_dev, double eps2, double coef, double* grad_dev);
    void frictionHessianAdd(const double* x_dev, const double* xt_dev, const int* dbc_dev, const HipLinSysSolver& lin, double eps2, double coef,
        int projectDBC, double* a_dev);
    void frictionConnectivity(std::vector<std::pair<int, int>>& pairs) const; // appends

private:
    struct GridHost {
        double lo[3], h;
        int dim[3];
        long long nCells;
    };
    GridHost makeGrid(const HipMesh& mesh, const double* x_dev, const double* p_dev, double alpha, double minCell);
    void buildCells(const GridHost& g, int nPrim, int nv, const int* prim_dev, const double* x_dev, const double* p_dev, double alpha, double infl,
        DevBuf<int>& cnt, DevBuf<int>& start, DevBuf<int>& items);
    DevBuf<int> d_cand_, d_ids_;
    DevBuf<double> d_vals_;
    DevBuf<unsigned long long> ccdOut_, ccdHits_;
    DevBuf<int> d_codimPoints;
    DevBuf<int> cellCountT_, cellCountE_, cellStartT_, cellStartE_, cellItemsT_, cellItemsE_, outPT_, outEE_, counters_;
    DevBuf<int> d_v2sv, refVbox_, refCount_, refStart_, refItems_; // reference-mode sweep: node -> surface index, index boxes, the three cell structures in one
    DevBuf<double> bboxPartial_;
    bool haveBox_ = false; // box_: the bounding box the last constraint-set build measured (the next build's grid is laid over it)
    double box_[6] = { 0, 0, 0, 0, 0, 0 };
    DevBuf<int> gridCount_, gridStart_, gridItems_; // the narrow phase's grids, triangles and edges in one array of 2 nCells + 1 cells (k_grid_insert_both)
    // on-device assembly of the sets (buildConstraintSet)
    int nActive_ = 0, nPara_ = 0, nCand_ = 0;
    mutable bool hostStale_ = false;
    DevBuf<unsigned long long> sortKeyIn_, sortKeyOut_, flags_, flagPos_;
    DevBuf<int> permPT_, dupTuple_, dupSorted_, closeIdx_;
    // counting sorts of the record lists (by first primitive) and of the duplicate candidates (by vertex): counters, bucket starts, bucket contents, runs
    DevBuf<int> bucketCount_, bucketStart_, bucketSeg_, dupCount_, dupStart_, runs_, runPos_;
    void readbackInit();
    bool hessErrPending_ = false, refDirty_ = false;
    bool countersDirty_ = false; // a build was left half-way (exception): the counters are cleared before the next one
    PinnedBuf<unsigned long long> readback_; // BuildReadback: what the host reads between the stages of a build (mapped memory, written by the kernels)
    DevBuf<double> closeVal_;
    DevBuf<char> scanTmp_;
    // deterministic scatter of the barrier / friction terms (hip_contact.hip "Deterministic scatter"): per-stencil slots, keys, bucket counters / starts /
    // contents of the counting sort
    DevBuf<double> detVals_;
    DevBuf<unsigned> detKey_;
    DevBuf<int> detCount_, detStart_, detSeg_, detSorted_, detRow_, hessPerm_; // hessPerm_: the two lists' indices binned by stencil kind (k_bin_stencils)
    bool detDirty_ = false; // a pass was left between its kernel and its fill (exception): the counters are cleared before the next one
    void detBegin(size_t nSlots, int valsPerSlot, bool withRow, bool fillKeys, size_t nKeys);
    void detBuckets(size_t nSlots, size_t nKeys, int div);
    void detReduce3(size_t nSlots, size_t nKeys, double* grad_dev);
    void detReduceBlocks(size_t nSlots, size_t nKeys, const int* ia_dev, double* a_dev);
};

} // namespace ipcgpu
