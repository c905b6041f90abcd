// GPU multifrontal LL^T: numeric factorisation and triangular solves on the assembly tree produced by
// mf_symbolic.  Plays the role of cholmod_factorize / cholmod_solve (CHOLMODSolver.cpp:130-154).
#pragma once
#include "common.h"
#include "mf_symbolic.h"

namespace ipcgpu {

class MfNumeric {
public:
    MfNumeric() = default;
    ~MfNumeric();
    MfNumeric(const MfNumeric&) = delete;
    MfNumeric& operator=(const MfNumeric&) = delete;

    // uploads maps, allocates fronts; ia_dev / ja_dev: the user's CSR pattern (0-based, symmetric upper) in HBM, nnzPattern entries -- the destinations of its
    // entries in the fronts are computed on the device (sym.aDst / aFront are not read)
    void setup(const MfSymbolic& sym, hipStream_t stream, const int* ia_dev, const int* ja_dev, long long nnzPattern);
    // Multi-GPU (one process per GPU): the assembly tree is cut below its top separators, every rank factorises and solves the
    // subtrees it owns.  Round 5: a front above the cut is executed by ONE rank (the executor of its most expensive child, mf_assign_executors) instead
    // of being repeated by all of them, and what a parent on another rank needs -- the packed update matrix of a child in the factorisation, its update
    // vector in the forward sweep, the solution entries of an ancestor in the backward sweep -- travels POINT TO POINT to exactly the ranks that
    // need it (`exchange`: one group of sends / receives per level of the cut; ncclSend / ncclRecv between ncclGroupStart / End in the RCCL binding).
    // Two all-reduces remain: the pivot flag (one double per factorisation) and the solution every rank ends up with (3 nV doubles per solve,
    // every entry contributed by its executor).  Call before setup().
    typedef int (*AllreduceFn)(void* user, void* buf_dev, long long count, int op);
    // stream-ordered variant (ipcgpu_opt_set_allreduce_stream): enqueued on the solver's stream, no host synchronisation around it
    typedef int (*AllreduceStreamFn)(void* user, void* buf_dev, long long count, int op, void* hipStream);
    struct P2POp { // == ipcgpu_p2p_op (include/ipcgpu.h)
        void* buf_dev;
        long long count;
        int peer;
        int send;
    };
    typedef int (*ExchangeFn)(void* user, int nOps, const P2POp* ops); // host-ordered: the solver drains its stream first
    typedef int (*ExchangeStreamFn)(void* user, int nOps, const P2POp* ops, void* hipStream); // enqueued on the solver's stream
    void setShard(int rank, int world, AllreduceFn fn, void* user, AllreduceStreamFn sfn = nullptr, void* streamUser = nullptr)
    {
        rank_ = rank;
        world_ = world;
        setHooks(fn, user, sfn, streamUser);
    }
    // the two parameters of the two-level blocking a caller may set (ipcgpu_linsys_set_tuning; read by the next setup()): no mesh of the test suite is big enough
    // for the default threshold, so the tests force the path on through these
    void setBulkTuning(double minMB, int block)
    {
        bulkMinMB_ = minMB;
        bulkBlock_ = block < 64 ? 64 : (block / 32) * 32;
    }
    void setExchangeHooks(ExchangeFn fn, void* user, ExchangeStreamFn sfn, void* streamUser)
    {
        exchange_ = fn;
        exchangeUser_ = user;
        exchangeStream_ = sfn;
        exchangeStreamUser_ = streamUser;
    }
    bool hasExchangeHook() const { return exchange_ || exchangeStream_; }
    // the hooks alone (the caller attached or detached a communicator after the solver was sharded): nothing to re-analyse
    void setHooks(AllreduceFn fn, void* user, AllreduceStreamFn sfn, void* streamUser)
    {
        allreduce_ = fn;
        allreduceUser_ = user;
        allreduceStream_ = sfn;
        allreduceStreamUser_ = streamUser;
    }
    int world() const { return world_; }
    // per node of the CALLER's numbering: the rank whose subtree eliminates it, -1 above the cut (the rows of those nodes are assembled by every rank: a
    // subtree's fronts read entries of the separator rows above them); all -1 on one rank.
    // What the owner-computes sharding of the assembly needs: a rank assembles the CSR rows of the nodes it owns or shares.
    void nodeOwners(std::vector<int>& ownerOfNode) const;
    long long exchangedBytes() const { return commBytes_; } // bytes THIS rank sent + received point to point + the buffers it all-reduced, factorisations and solves so far
    long long exchangeCalls() const { return commCalls_; }
    long long sentBytes() const { return sentBytes_; }
    long long receivedBytes() const { return recvBytes_; }
    // time this rank's stream spent inside the point-to-point groups so far (HIP events around every group: a rank that executes nothing at a level of the cut
    // waits in its receive for the rank that does) -- the measured counterpart of `steps above the cut` in the strong-scaling model (bench.py expected_speedup_model)
    double exchangeWaitMs();
    // inputs of that model: out5 = { dependent 32-column pivot steps on the critical path of the tree (sum over the levels of the widest front's steps),
    // the same over the fronts above the cut only, the largest rank's share of the flops below the cut (1 / world when balanced), levels, levels that
    // hold a front above the cut }
    void criticalPath(double* out5) const;
    double sharedFlopFraction() const { return sharedFlops_; } // share of the factorisation flops above the cut (executed once each since round 5, on the chain of the cut's levels)
    // a_dev: CSR values (device).  Returns false when a non-positive pivot was met.
    bool factorize(const double* a_dev);
    // rhs_dev / x_dev: device vectors in the user's ordering
    void solve(const double* rhs_dev, double* x_dev);
    // factorize(a) and solve(rhs) in one go, the forward sweep overlapped with the factorisation (single rank); returns false when a
    // non-positive pivot was met (x is then meaningless)
    // wait = false: everything is enqueued and the call returns true without synchronising (the pivot flag is on its way to pinned memory:
    // lastPivotsOk() after the caller's own synchronisation of the stream).  Returns false when the call had to take the synchronous
    // two-call sequence (sharded / graph runs) and the factorisation failed there.
    bool factorizeSolve(const double* a_dev, const double* rhs_dev, double* x_dev, bool wait = true);
    bool lastPivotsOk() const { return pivotsOk(); }
    bool pivotsOk() const; // false: a non-positive pivot
    bool ready() const { return ns_ > 0; }
    // diagnosis / tests: the slot of every entry of the user's matrix in the front buffer, as the set-up's device kernel computed it (nnz values)
    void entryDestinations(long long* out, size_t nnz) const { entryDst_.download(out, nnz, stream_); }
    size_t front_bytes() const { return fronts_.n * sizeof(double); }

private:
    void enqueueFactor(const double* a_dev, bool overlapForward = false);
    void enqueueSolve(const double* rhs_dev, double* x_dev);
    struct Range {
        int off = 0, cnt = 0;
    };
    struct LevelPlan {
        Range small; // into smallList_
        size_t smallLds = 0, solveLds = 0, triLds = 0, bwdLds = 0;
        int smallThreads = 256; // workgroup size of the fused kernel on this level
        Range ea; // extend-add descriptors
        Range bigFronts; // into bigList_
        std::vector<Range> step; // fused factor steps: launch 0 factors panel 0, launch j+1 applies panel j / factors j+1
        std::vector<Range> bulk; // per step launch: the bulk updates of the wide fronts that follow it (k_big_bulk; empty ranges elsewhere)
        Range schur; // one-pass Schur complement tiles of the big fronts
        bool schur64 = false; // ... as 64 x 64 tiles (k_big_schur64) instead of 32 x 32 with the columns split over the waves
        bool stepTop = false; // the level's step launches carry role C, the explicit inverse growing by bordering (k_big_step<true>)
        bool fuseEA = false; // the level's Schur kernel gathers the children of the update block itself (k_big_schur64_ea); the extend-add only writes own columns
        Range fwdRect, bwdInit; // descriptors of the row-/column-parallel halves of the big-front solves
        Range bigTri; // into triList_: big fronts whose triangle is swept by one workgroup (no explicit inverse)
        Range xinvFwd, xinvBwd; // into xinvDesc_: row / column blocks of the fronts with an explicit inverse
    };
    int rank_ = 0, world_ = 1;
    const long long schur64Min_ = 512; // levels with at least this many 32 x 32 Schur tiles take the 64 x 64 kernel (profiles/r03r_schur_tile_ab.txt)
    double bulkMinMB_ = 48.0; // levels whose step launches read + write at least that many MB of own columns factor them in outer blocks (two-level blocking, k_big_bulk); ipcgpu_linsys_set_tuning "bulk_min_mb"
    int bulkBlock_ = 256; // width of an outer block; ipcgpu_linsys_set_tuning "bulk_block"
    bool xinvBorder_ = true; // X = L11^-1 by bordering inside the step launches (false: recursive doubling on the side stream, the round-4 scheme)
    AllreduceFn allreduce_ = nullptr;
    AllreduceStreamFn allreduceStream_ = nullptr;
    void* allreduceUser_ = nullptr;
    void* allreduceStreamUser_ = nullptr; // its own slot: attaching RCCL after a host hook must not replace that hook's user pointer
    double sharedFlops_ = 0.0;
    ExchangeFn exchange_ = nullptr;
    ExchangeStreamFn exchangeStream_ = nullptr;
    void* exchangeUser_ = nullptr;
    void* exchangeStreamUser_ = nullptr;
    std::vector<int> owner_; // per front: owning rank, -1 = above the cut
    std::vector<int> exec_; // per front: the rank that factorises and solves it (== owner_ below the cut)
    std::vector<unsigned long long> group_; // per front: ranks that execute a front of its subtree
    long long commBytes_ = 0, commCalls_ = 0, sentBytes_ = 0, recvBytes_ = 0;
    double waitMs_ = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> waitPending_, waitFree_; // event pairs around exchange groups not yet read / free for reuse
    struct Xchg {
        Range pack; // into xchgDesc_: (front, staging offset lo, hi, offset of its update vector) of the fronts of this level this rank SENDS to their parent's rank
        Range unpack; // ... and of the children (of this level) of fronts this rank executes that it RECEIVES
        std::vector<P2POp> opsM, opsW, opsX; // the level's groups: update matrices (factorisation), update vectors (forward sweep), solution segments (backward sweep)
        long long count = 0; // doubles exchanged after the level's factorisation (update matrices)
        long long countW = 0; // ... and after its forward sweep (update vectors)
    };
    std::vector<Xchg> xchg_;
    DevBuf<int4> xchgDesc_;
    DevBuf<double> xchgBuf_;
    DevBuf<int> nodeExec_; // per permuted node: executing rank
    void allreduceSum(double* dev, long long count);
    void exchange(const std::vector<P2POp>& ops);
    // mf_exchange.hip: what crosses ranks, per level of the cut
    void exchangeUpdateMatrices(int level);
    void exchangeUpdateVectors(int level);
    void reduceSolution();
    void allreduceFlag();
    // mf_sweeps.hip
    void configureSweepKernels(size_t maxSolveLds, size_t maxBwdLds, size_t maxTriLds);
    void enqueuePermuteRhs(const double* rhs_dev, hipStream_t st);
    const MfSymbolic* sym_ = nullptr;
    hipStream_t stream_ = nullptr;
    int ns_ = 0, nLevels_ = 0;
    long long nDiagBlocks_ = 0;
    std::vector<LevelPlan> plan_;
    DevBuf<double> fronts_, w_, yperm_, bperm_, xsol_;
    // explicit inverses X = L11^-1 of the widest fronts (and the scratch T of their recursive doubling)
    DevBuf<double> xinvX_, xinvT_;
    DevBuf<long long> xinvOff_;
    DevBuf<int4> xinvDesc_;
    DevBuf<int> triList_;
    struct XinvLevel {
        Range blocks; // diagonal blocks of the level's inverse fronts (cnt > 0: the level has inverses to form by recursive doubling)
        Range init; // into xinvDesc_
        std::vector<std::pair<Range, Range>> rounds; // per doubling: the two GEMM launches (descriptor pairs)
    };
    std::vector<XinvLevel> xinvLevel_;
    hipStream_t side_ = nullptr; // inverses are formed here, beside the chain of the upper levels
    bool xcdOrder_ = true; // Schur tiles dealt to the XCDs front by front (false: front after front over all XCDs, as before round 5; profiles/r05_solver_ab_xcd_occupancy.txt)
    int fwdStride_ = 4; // levels handed to the forward stream per event (1 = every level, as before round 4; profiles/r05_knob_sweep.txt)
    bool fwdJoined_ = false; // the root's forward sweep went onto the main stream (factorizeSolve)
    // factorizeSolve(): the forward sweep of a level is enqueued on its own stream as soon as that level's factor kernels are, so that it
    // runs beside the latency-bound pivot chain of the levels above instead of behind the whole factorisation
    hipStream_t fwd_ = nullptr;
    hipEvent_t evRhs_ = nullptr, evFwdDone_ = nullptr;
    std::vector<hipEvent_t> evFactLevel_;
    void enqueueForwardLevel(int l, hipStream_t st);
    void enqueueBackward(double* x_dev);
    hipEvent_t evSide_ = nullptr;
    std::vector<hipEvent_t> evLevel_, evInvDone_;
    bool sidePending_ = false; // the last factorisation left work on the side stream that nothing has waited for yet
    void enqueueInverses(int level, hipStream_t st);
    size_t xinvLds_ = 8;
    DevBuf<double> dinv_; // explicit inverses of the 32x32 diagonal blocks of L, 1024 doubles each
    DevBuf<int> idx_, idxPtr_, firstNode_, childPtr_, child_, invPtr_, inv_, newOf_, flag_;
    DevBuf<long long> frontOff_, wOff_, dinvOff_;
    std::vector<long long> hDinvOff_; // host copy: the step records carry a front's first inverse block
    DevBuf<int> aSrc_, aLoc_; // entries of A per fused front: CSR source index, offset inside the LDS panel
    DevBuf<double> aPerm_; // values of A gathered into fused-front order at the start of every factorisation
    int nFusedA_ = 0;
    std::vector<int> aPtrHost_; // front -> first entry (goes into the packed descriptors)
    DevBuf<int> bigFd_; // packed records of the other fronts (k_extend_add)
    DevBuf<int> fdesc_; // packed descriptors of the fused fronts (64 ints each, launch order)
    DevBuf<int> bigASrc_; // entries of A of the other fronts, grouped by extend-add tile: source index ...
    DevBuf<long long> bigADst_; // ... and destination in the front buffer
    DevBuf<long long> entryDst_; // set-up scratch: per entry of A its slot in the front buffer ...
    DevBuf<int> entryBucket_, bucketHist_, bucketStart_, nodeFront_; // ... its bucket, the buckets' counts + tickets and starts, the front of every permuted node
    DevBuf<int4> frontInfo_;
    DevBuf<int> eaAPtr_; // per extend-add tile: first of its entries in bigASrc_ / bigADst_ (the extend-add kernel adds them, round 5)
    DevBuf<int4> eaDesc_;
    DevBuf<int> smallList_, bigList_;
    DevBuf<int4> desc_; // all big-front step descriptors
    PinnedBuf<int> hflag_;
};

} // namespace ipcgpu
