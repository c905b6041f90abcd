// extern "C" boundary of libipcgpu.so (include/ipcgpu.h).  No exception leaves this file.
#include "../../include/ipcgpu.h"
#include "hip_ipc.h"
#include "msh_io.h"
#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace ipcgpu;

namespace {
thread_local std::string g_err;

template <class Fn>
int guarded(Fn&& fn)
{
    try {
        return fn();
    }
    catch (const ArgError& e) {
        g_err = e.what();
        return IPCGPU_ERR_ARG;
    }
    catch (const StateError& e) {
        g_err = e.what();
        return IPCGPU_ERR_STATE;
    }
    catch (const HipError& e) {
        g_err = e.what();
        return IPCGPU_ERR_HIP;
    }
    catch (const std::exception& e) {
        g_err = e.what();
        return IPCGPU_ERR_HIP;
    }
    catch (...) {
        g_err = "unknown error";
        return IPCGPU_ERR_HIP;
    }
}

void need(bool cond, const char* what)
{
    if (!cond) throw StateError(what);
}
void needArg(bool cond, const char* what)
{
    if (!cond) throw ArgError(what);
}
HipMesh& M(ipcgpu_ctx* c)
{
    needArg(c != nullptr, "null context");
    need(c->mesh && c->mesh->nV > 0, "no mesh: call ipcgpu_set_mesh first");
    return *c->mesh;
}
HipLinSysSolver& L(ipcgpu_ctx* c)
{
    needArg(c != nullptr, "null context");
    return *c->lin;
}
HipOptimizer& O(ipcgpu_ctx* c)
{
    M(c);
    return *c->opt;
}
void bind(ipcgpu_ctx* c) { HIP_CHECK(hipSetDevice(c->device)); }
// Every entry point that changes what the next assembly depends on (positions, constraints, material, time step, ...) calls this first: the stepper
// may hold an assembly it enqueued ahead for the state it had (HipOptimizer::speculativeAssembly), and that one must not be swapped in afterwards.
void specChanged(ipcgpu_ctx* c)
{
    if (c && c->opt) c->opt->specAsmValid = false;
}
// After an owner-computes assembly on a sharded context (HipOptimizer::ownerMode) a[] holds this rank's rows only.  Entry points that consume the WHOLE
// matrix refuse to work on such a partial one instead of returning a rank's share silently; ipcgpu_opt_complete_matrix (a collective: never called
// implicitly) completes it.  ADVICE round 4.
void needWholeMatrix(ipcgpu_ctx* c)
{
    need(!(c->opt && c->opt->ownerMode() && !c->opt->matrixComplete),
        "the matrix holds this rank's rows only (owner-computes assembly): call ipcgpu_opt_complete_matrix on every rank first");
}
// contact-pair lists shard with the elements (ipcgpu_ctx_set_shard): the handler is created lazily, so both places call this
void applyContactShard(ipcgpu_ctx* c)
{
    if (!c->contact) return;
    c->contact->shardRank = c->rank;
    c->contact->shardWorld = c->worldSize;
    HipOptimizer* o = c->opt.get();
    c->contact->shardReduce = [o](double* d, long long n) { o->reduceSum(d, n); };
}

// column-major nV x 3 (host) <-> xyz interleaved (device)
void uploadColMajor(ipcgpu_ctx* c, const double* Vcm, DevBuf<double>& dst)
{
    const int nV = c->mesh->nV;
    std::vector<double> aos(3 * (size_t)nV);
    for (int v = 0; v < nV; ++v)
        for (int k = 0; k < 3; ++k) aos[3 * (size_t)v + k] = Vcm[v + (size_t)nV * k];
    dst.upload(aos, c->stream);
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
void downloadColMajor(ipcgpu_ctx* c, const DevBuf<double>& src, double* Vcm)
{
    const int nV = c->mesh->nV;
    std::vector<double> aos(3 * (size_t)nV);
    src.download(aos.data(), aos.size(), c->stream);
    for (int v = 0; v < nV; ++v)
        for (int k = 0; k < 3; ++k) Vcm[v + (size_t)nV * k] = aos[3 * (size_t)v + k];
}
} // namespace

extern "C" {

const char* ipcgpu_last_error(void) { return g_err.c_str(); }
int ipcgpu_version(void) { return 100; }

int ipcgpu_ctx_create(int device_id, ipcgpu_ctx** out)
{
    return guarded([&] {
        needArg(out != nullptr, "null out pointer");
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) throw HipError("no HIP device visible (this library has no CPU fallback)");
        needArg(device_id >= 0 && device_id < count, "device id out of range");
        HIP_CHECK(hipSetDevice(device_id));
        auto* c = new ipcgpu_ctx;
        c->device = device_id;
        {
            // the context's stream carries the dependent chain of every Newton iteration (pivot steps of the top separators): highest priority, so that its
            // workgroups are placed ahead of the bulk work the solver keeps on its own streams beside it (Schur passes, forward sweep)
            int lo = 0, hi = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_CHECK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
        }
        c->mesh.reset(new HipMesh);
        c->lin.reset(new HipLinSysSolver(c->stream));
        c->opt.reset(new HipOptimizer(*c->mesh, *c->lin, c->stream));
        *out = c;
        return IPCGPU_OK;
    });
}
int ipcgpu_ctx_destroy(ipcgpu_ctx* c)
{
    return guarded([&] {
        if (!c) return IPCGPU_OK;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        c->contact.reset();
        c->opt.reset();
        c->lin.reset();
        c->mesh.reset();
        (void)hipStreamDestroy(c->stream);
        delete c;
        return IPCGPU_OK;
    });
}
int ipcgpu_ctx_set_solver(ipcgpu_ctx* c, int type)
{
    specChanged(c);
    return guarded([&] {
        needArg(c && (type == IPCGPU_SOLVER_MULTIFRONTAL || type == IPCGPU_SOLVER_ROCSOLVER_CSRRF), "unknown solver type"); // LinSysSolver.cpp:24-26
        c->lin->solverType = type;
        return IPCGPU_OK;
    });
}
int ipcgpu_ctx_set_shard(ipcgpu_ctx* c, int rank, int world)
{
    specChanged(c);
    return guarded([&] {
        needArg(c && world >= 1 && rank >= 0 && rank < world, "bad shard");
        c->rank = rank;
        c->worldSize = world;
        c->opt->rank = rank;
        c->opt->worldSize = world;
        applyContactShard(c);
        if (c->mesh->nT) {
            c->opt->tetBegin = (int)((long long)c->mesh->nT * rank / world);
            c->opt->tetEnd = (int)((long long)c->mesh->nT * (rank + 1) / world);
        }
        return IPCGPU_OK;
    });
}

int ipcgpu_set_mesh(ipcgpu_ctx* c, int nV, int nT, const double* Vr, const int* F, double YM, double PR, double rho)
{
    return guarded([&] {
        needArg(c && Vr && (F || nT == 0), "null argument");
        bind(c);
        c->mesh->computeFeatures(nV, nT, Vr, F, YM, PR, rho, c->stream);
        c->opt->tetBegin = (int)((long long)nT * c->rank / c->worldSize);
        c->opt->tetEnd = (int)((long long)nT * (c->rank + 1) / c->worldSize);
        c->opt->initialised = false;
        c->opt->selfCollision = false; // surface + contact state belong to the previous mesh
        c->opt->contact = nullptr;
        c->opt->planes.clear();
        c->contact.reset();
        return IPCGPU_OK;
    });
}
int ipcgpu_set_component_material(ipcgpu_ctx* c, int nodeBegin, int nodeEnd, int tetBegin, int tetEnd, double rho, double YM, double PR)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        needArg(0 <= nodeBegin && nodeBegin <= nodeEnd && nodeEnd <= m.nV && 0 <= tetBegin && tetBegin <= tetEnd && tetEnd <= m.nT, "range out of bounds");
        needArg(rho > 0 && YM > 0 && PR > -1.0 && PR < 0.5, "bad material");
        m.setComponentMaterial(nodeBegin, nodeEnd, tetBegin, tetEnd, rho, YM, PR, c->stream);
        return IPCGPU_OK;
    });
}
struct ipcgpu_tetmesh {
    ipcgpu::TetMeshFile m;
};
int ipcgpu_read_tet_mesh(const char* path, ipcgpu_tetmesh** mesh, int* nV, int* nT, int* nSF)
{
    return guarded([&] {
        needArg(path && mesh && nV && nT && nSF, "null argument");
        std::unique_ptr<ipcgpu_tetmesh> h(new ipcgpu_tetmesh);
        readTetMesh(path, h->m, true);
        *nV = (int)(h->m.V.size() / 3);
        *nT = (int)(h->m.T.size() / 4);
        *nSF = (int)(h->m.SF.size() / 3);
        *mesh = h.release();
        return IPCGPU_OK;
    });
}
int ipcgpu_tet_mesh_get(const ipcgpu_tetmesh* h, double* V, int* T, int* SF)
{
    return guarded([&] {
        needArg(h != nullptr, "null mesh");
        const size_t nV = h->m.V.size() / 3, nT = h->m.T.size() / 4, nSF = h->m.SF.size() / 3;
        if (V)
            for (size_t v = 0; v < nV; ++v)
                for (int c = 0; c < 3; ++c) V[v + nV * c] = h->m.V[3 * v + c];
        if (T)
            for (size_t t = 0; t < nT; ++t)
                for (int k = 0; k < 4; ++k) T[t + nT * k] = h->m.T[4 * t + k];
        if (SF)
            for (size_t t = 0; t < nSF; ++t)
                for (int k = 0; k < 3; ++k) SF[t + nSF * k] = h->m.SF[3 * t + k];
        return IPCGPU_OK;
    });
}
void ipcgpu_tet_mesh_free(ipcgpu_tetmesh* h) { delete h; }
int ipcgpu_save_tet_mesh(const char* path, int nV, int nT, const double* V, const int* T)
{
    return guarded([&] {
        needArg(path && V && T && nV >= 4 && nT >= 1, "bad argument");
        std::vector<double> v(3 * (size_t)nV);
        std::vector<int> t(4 * (size_t)nT), sf;
        for (int i = 0; i < nV; ++i)
            for (int c = 0; c < 3; ++c) v[3 * (size_t)i + c] = V[i + (size_t)nV * c];
        for (int i = 0; i < nT; ++i)
            for (int k = 0; k < 4; ++k) {
                const int n = T[i + (size_t)nT * k];
                needArg(n >= 0 && n < nV, "element refers to a node that does not exist");
                t[4 * (size_t)i + k] = n;
            }
        findSurfaceTris(nT, t.data(), sf);
        saveTetMesh(path, nV, nT, v.data(), t.data(), sf);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_energy_type(ipcgpu_ctx* c, int energyType)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        needArg(energyType == 0 || energyType == 1, "energy type: 0 = NH, 1 = FCR");
        m.energyType = energyType;
        return IPCGPU_OK;
    });
}
int ipcgpu_set_dbc(ipcgpu_ctx* c, int n, const int* ids, int type)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        needArg(type >= 0 && type <= 2, "bad DirichletBCType");
        for (int i = 0; i < n; ++i) {
            needArg(ids[i] >= 0 && ids[i] < m.nV, "vertex id out of range");
            m.dbcType[ids[i]] = type;
        }
        m.uploadDBC(c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_clear_dbc(ipcgpu_ctx* c)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        std::fill(m.dbcType.begin(), m.dbcType.end(), 0);
        m.uploadDBC(c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_positions(ipcgpu_ctx* c, const double* V)
{
    specChanged(c);
    return guarded([&] {
        M(c);
        bind(c);
        needArg(V != nullptr, "null V");
        uploadColMajor(c, V, c->mesh->d_x);
        return IPCGPU_OK;
    });
}
int ipcgpu_get_positions(ipcgpu_ctx* c, double* V)
{
    return guarded([&] {
        M(c);
        bind(c);
        needArg(V != nullptr, "null V");
        downloadColMajor(c, c->mesh->d_x, V);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_xtilde(ipcgpu_ctx* c, const double* V)
{
    specChanged(c);
    return guarded([&] {
        M(c);
        bind(c);
        needArg(V != nullptr, "null xTilta");
        uploadColMajor(c, V, c->mesh->d_xTilde);
        return IPCGPU_OK;
    });
}
int ipcgpu_get_features(ipcgpu_ctx* c, double* A, double* vol, double* mass, double* mu, double* lam)
{
    return guarded([&] {
        HipMesh& m = M(c);
        if (A)
            for (int t = 0; t < m.nT; ++t)
                for (int k = 0; k < 9; ++k) A[9 * (size_t)t + k] = m.restTriInv[(size_t)k * m.nT + t];
        if (vol) std::memcpy(vol, m.triArea.data(), sizeof(double) * m.nT);
        if (mass) std::memcpy(mass, m.mass.data(), sizeof(double) * m.nV);
        if (mu) std::memcpy(mu, m.mu.data(), sizeof(double) * m.nT);
        if (lam) std::memcpy(lam, m.lam.data(), sizeof(double) * m.nT);
        return IPCGPU_OK;
    });
}
int ipcgpu_get_mesh_dims(ipcgpu_ctx* c, int* nV, int* nT)
{
    return guarded([&] {
        need(c != nullptr, "null context");
        if (nV) *nV = c->mesh ? c->mesh->nV : 0;
        if (nT) *nT = c->mesh ? c->mesh->nT : 0;
        return IPCGPU_OK;
    });
}
int ipcgpu_set_mesh_features(ipcgpu_ctx* c, const double* A, const double* vol, const double* mass, const double* mu, const double* lam)
{
    specChanged(c);
    return guarded([&] {
        bind(c);
        HipMesh& m = M(c);
        need(m.nV > 0 && m.nT > 0, "call ipcgpu_set_mesh first");
        if (A) {
            for (int t = 0; t < m.nT; ++t)
                for (int k = 0; k < 9; ++k) m.restTriInv[(size_t)k * m.nT + t] = A[9 * (size_t)t + k];
            m.d_A.upload(m.restTriInv, c->stream);
        }
        if (vol) {
            m.triArea.assign(vol, vol + m.nT);
            m.d_vol.upload(m.triArea, c->stream);
        }
        if (mass) {
            m.mass.assign(mass, mass + m.nV);
            m.d_mass.upload(m.mass, c->stream);
        }
        if (mu) {
            m.mu.assign(mu, mu + m.nT);
            m.d_mu.upload(m.mu, c->stream);
        }
        if (lam) {
            m.lam.assign(lam, lam + m.nT);
            m.d_lam.upload(m.lam, c->stream);
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return IPCGPU_OK;
    });
}
int ipcgpu_check_inversion(ipcgpu_ctx* c, int* ok)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        *ok = o.checkInversion() ? 1 : 0;
        return IPCGPU_OK;
    });
}

// ---- Energy ------------------------------------------------------------------------------------------
int ipcgpu_elastic_energy(ipcgpu_ctx* c, double coef, double* E)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        launch_energy(o.view(), coef, false, o.rank == 0, o.d_partial.p, (int)o.d_partial.n, o.d_scalar.p, c->stream);
        o.reduceSum(o.d_scalar.p, 1);
        *E = o.readScalar(o.d_scalar.p);
        return IPCGPU_OK;
    });
}
int ipcgpu_elastic_energy_per_elem(ipcgpu_ctx* c, double* out)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        DevBuf<double> tmp;
        tmp.alloc(c->mesh->nT);
        launch_energy_per_elem(o.view(), tmp.p, c->stream);
        tmp.download(out, c->mesh->nT, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_elastic_gradient(ipcgpu_ctx* c, double coef, int projectDBC, double* g)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        o.d_gradient.zero(c->stream);
        launch_assemble(o.view(), coef, projectDBC, o.d_gradient.p, nullptr, c->stream);
        o.reduceSum(o.d_gradient.p, 3LL * c->mesh->nV);
        o.d_gradient.download(g, 3 * (size_t)c->mesh->nV, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_elastic_hessian_add(ipcgpu_ctx* c, double coef, int projectDBC)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        need(!c->lin->rowBase.empty(), "call ipcgpu_linsys_set_pattern first");
        launch_assemble(o.view(), coef, projectDBC, nullptr, c->lin->d_a.p, c->stream);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return IPCGPU_OK;
    });
}
int ipcgpu_filter_step_size(ipcgpu_ctx* c, const double* p, double* step)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        DevBuf<double> dp;
        dp.upload(p, 3 * (size_t)c->mesh->nV, c->stream);
        *step = o.filterStepSize(dp.p, *step);
        return IPCGPU_OK;
    });
}

// ---- LinSysSolver ------------------------------------------------------------------------------------
int ipcgpu_linsys_set_pattern(ipcgpu_ctx* c, int nExtra, const int* pairs)
{
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        c->lin->set_pattern(m, nExtra, pairs);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_set_pattern_csr(ipcgpu_ctx* c, int n, const int* ia, const int* ja)
{
    return guarded([&] {
        needArg(c && ia && ja, "null argument");
        bind(c);
        c->lin->set_pattern_csr(n, ia, ja);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_get_dims(ipcgpu_ctx* c, int* n, int* nnz)
{
    return guarded([&] {
        if (n) *n = L(c).getNumRows();
        if (nnz) *nnz = L(c).getNumNonzeros();
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_get_pattern(ipcgpu_ctx* c, int* ia, int* ja)
{
    return guarded([&] {
        HipLinSysSolver& l = L(c);
        if (ia) std::memcpy(ia, l.ia.data(), sizeof(int) * l.ia.size());
        if (ja) std::memcpy(ja, l.ja.data(), sizeof(int) * l.ja.size());
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_set_zero(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        L(c).setZero();
        if (c->opt) c->opt->matrixComplete = true; // whatever the caller adds from here on is whole on every rank (the handler's entry points return whole sums)
        specChanged(c);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_get_values(ipcgpu_ctx* c, double* a)
{
    return guarded([&] {
        bind(c);
        needWholeMatrix(c);
        L(c).d_a.download(a, L(c).ja.size(), c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_set_values(ipcgpu_ctx* c, const double* a)
{
    return guarded([&] {
        bind(c);
        HipLinSysSolver& l = L(c);
        need(l.numRows > 0, "no pattern");
        HIP_CHECK(hipMemcpyAsync(l.d_a.p, a, sizeof(double) * l.ja.size(), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_apply_host_updates(ipcgpu_ctx* c, const double* delta, const unsigned char* isSet, const double* setVal)
{
    return guarded([&] {
        bind(c);
        HipLinSysSolver& l = L(c);
        need(l.numRows > 0, "no pattern");
        need(delta != nullptr, "delta must not be NULL");
        need((isSet == nullptr) == (setVal == nullptr), "isSet and setVal go together");
        const size_t nnz = l.ja.size();
        l.hostDelta.upload(delta, nnz, c->stream);
        if (isSet) {
            l.hostSetMask.upload(isSet, nnz, c->stream);
            l.hostSetVal.upload(setVal, nnz, c->stream);
        }
        launch_apply_host_updates((long long)nnz, l.hostDelta.p, isSet ? l.hostSetMask.p : nullptr, isSet ? l.hostSetVal.p : nullptr, l.d_a.p,
            c->stream);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return IPCGPU_OK;
    });
}
static int coeff(ipcgpu_ctx* c, int row, int col, double v, bool add)
{
    return guarded([&] {
        bind(c);
        HipLinSysSolver& l = L(c);
        if (row > col) return IPCGPU_OK; // LinSysSolver.hpp:331-339, 402-410: lower-triangle writes are ignored
        const int k = l.findEntry(row, col);
        needArg(k >= 0, "entry not in the sparsity pattern");
        double cur = 0.0;
        if (add) {
            HIP_CHECK(hipMemcpyAsync(&cur, l.d_a.p + k, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        cur = add ? cur + v : v;
        HIP_CHECK(hipMemcpyAsync(l.d_a.p + k, &cur, sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_add_coeff(ipcgpu_ctx* c, int row, int col, double v) { return coeff(c, row, col, v, true); }
int ipcgpu_linsys_set_coeff(ipcgpu_ctx* c, int row, int col, double v) { return coeff(c, row, col, v, false); }
int ipcgpu_linsys_multiply(ipcgpu_ctx* c, const double* x, double* y)
{
    return guarded([&] {
        bind(c);
        HipLinSysSolver& l = L(c);
        need(l.numRows > 0, "no pattern");
        needWholeMatrix(c);
        DevBuf<double> dx, dy;
        dx.upload(x, l.numRows, c->stream);
        dy.alloc(l.numRows);
        l.multiply(dx.p, dy.p);
        dy.download(y, l.numRows, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_analyze_pattern(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        L(c).analyze_pattern(c->mesh->nV ? c->mesh.get() : nullptr);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_factorize(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        return L(c).factorize() ? IPCGPU_OK : IPCGPU_NOT_PD;
    });
}
int ipcgpu_linsys_solve(ipcgpu_ctx* c, const double* rhs, double* x)
{
    return guarded([&] {
        bind(c);
        HipLinSysSolver& l = L(c);
        DevBuf<double> db, dx;
        db.upload(rhs, l.numRows, c->stream);
        dx.alloc(l.numRows);
        l.solve(db.p, dx.p);
        dx.download(x, l.numRows, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_precondition_diag(ipcgpu_ctx* c, const double* in, double* out)
{
    return guarded([&] {
        bind(c);
        HipLinSysSolver& l = L(c);
        needWholeMatrix(c);
        DevBuf<double> di, dout;
        di.upload(in, l.numRows, c->stream);
        dout.alloc(l.numRows);
        l.precondition_diag(di.p, dout.p);
        dout.download(out, l.numRows, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_set_shard(ipcgpu_ctx* c, int rank, int world)
{
    specChanged(c);
    return guarded([&] {
        needArg(c && world >= 1 && rank >= 0 && rank < world, "bad shard");
        need(world == 1 || c->opt->allreduce != nullptr || c->opt->allreduceStream != nullptr, "set the all-reduce hook first (ipcgpu_opt_set_allreduce)");
        need(world == 1 || c->lin->hasExchangeHook(), "set the exchange hook first (ipcgpu_opt_set_exchange / ipcgpu_opt_set_exchange_stream): the sharded solver sends point to point");
        c->lin->setShard(rank, world, c->opt->allreduce, c->opt->allreduceUser, c->opt->allreduceStream, c->opt->allreduceStreamUser);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_comm_stats(ipcgpu_ctx* c, double* out6)
{
    return guarded([&] {
        needArg(c && out6, "null argument");
        HipOptimizer& o = O(c);
        out6[0] = (double)o.commBytes;
        out6[1] = (double)o.commCalls;
        out6[2] = (double)L(c).exchangedBytes();
        out6[3] = (double)L(c).exchangeCalls();
        out6[4] = (double)(o.ownerMode() ? o.ownerNeededNodes : c->mesh->nV);
        out6[5] = (double)c->mesh->nV;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_complete_matrix(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        O(c).completeMatrix();
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_exchange_stats(ipcgpu_ctx* c, double* out4)
{
    return guarded([&] {
        needArg(c && out4, "null argument");
        out4[0] = (double)L(c).sentBytes();
        out4[1] = (double)L(c).receivedBytes();
        out4[2] = (double)L(c).exchangeCalls();
        out4[3] = L(c).exchangeWaitMs();
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_entry_destinations(ipcgpu_ctx* c, long long* dst_nnz)
{
    return guarded([&] {
        bind(c);
        needArg(c && dst_nnz, "null argument");
        need(L(c).solverType == IPCGPU_SOLVER_MULTIFRONTAL && L(c).analyzed(), "the multifrontal solver has not analysed a pattern yet");
        L(c).entryDestinations(dst_nnz);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_shard_stats(ipcgpu_ctx* c, double* out2)
{
    return guarded([&] {
        out2[0] = L(c).solverWorld();
        out2[1] = L(c).sharedFlopFraction();
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_set_tuning(ipcgpu_ctx* c, double bulk_min_mb, int bulk_block)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        needArg(bulk_min_mb >= 0.0 && bulk_block >= 64 && bulk_block <= 4096, "set_tuning: bulk_min_mb >= 0, 64 <= bulk_block <= 4096");
        L(c).setBulkTuning(bulk_min_mb, bulk_block);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_critical_path(ipcgpu_ctx* c, double* out5)
{
    return guarded([&] {
        needArg(c && out5, "null argument");
        L(c).criticalPath(out5);
        return IPCGPU_OK;
    });
}
int ipcgpu_linsys_stats(ipcgpu_ctx* c, double* st)
{
    return guarded([&] {
        const MfSymbolic& s = L(c).symbolic();
        st[0] = (double)s.nnzL;
        st[1] = s.flops;
        st[2] = s.ns;
        st[3] = (double)s.levelPtr.size() - 1;
        return IPCGPU_OK;
    });
}

// ---- SelfCollisionHandler ----------------------------------------------------------------------------
static HipContact& CT(ipcgpu_ctx* c)
{
    M(c);
    if (!c->contact) {
        c->contact.reset(new HipContact(c->stream));
        applyContactShard(c);
    }
    return *c->contact;
}
int ipcgpu_set_surface(ipcgpu_ctx* c, int nSF, const int* SF)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        needArg(nSF >= 0 && (SF || nSF == 0), "bad surface");
        m.addSurfaceEdges(nSF, SF);
        CT(c).setSurface(m, nSF, SF);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_surface_codim(ipcgpu_ctx* c, int nSF, const int* SF, int nCE, const int* CE)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        needArg(nSF >= 0 && (SF || nSF == 0), "bad surface");
        needArg(nCE >= 0 && (CE || nCE == 0), "bad codimensional segments");
        m.addSurfaceEdges(nSF, SF, nCE, CE);
        CT(c).setSurface(m, nSF, SF, nCE, CE);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_exact_predicates(ipcgpu_ctx* c, int on)
{
    specChanged(c);
    return guarded([&] {
        CT(c).exactPredicates = on != 0;
        return IPCGPU_OK;
    });
}
int ipcgpu_get_surface(ipcgpu_ctx* c, int* counts, int* SVI, int* SFE)
{
    return guarded([&] {
        HipContact& k = CT(c);
        if (counts) {
            counts[0] = k.nSVI;
            counts[1] = k.nSF;
            counts[2] = k.nSFE;
        }
        if (SVI) std::memcpy(SVI, k.SVI.data(), sizeof(int) * k.SVI.size());
        if (SFE)
            for (size_t i = 0; i < k.SFEdges.size(); ++i) {
                SFE[2 * i] = k.SFEdges[i].first;
                SFE[2 * i + 1] = k.SFEdges[i].second;
            }
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_build(ipcgpu_ctx* c, double dHat, int* counts)
{
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        needArg(dHat > 0, "dHat must be positive");
        HipContact& k = CT(c);
        k.buildConstraintSet(m, m.d_x.p, m.d_dbc.p, dHat);
        if (counts) {
            counts[0] = k.nActive();
            counts[1] = k.nPara();
            counts[2] = k.nCand();
        }
        return IPCGPU_OK;
    });
}
int ipcgpu_set_codim_nodes(ipcgpu_ctx* c, int n, const int* ids, const double* mass)
{
    specChanged(c);
    return guarded([&] {
        HipMesh& m = M(c);
        bind(c);
        needArg(n >= 0 && ((ids && mass) || !n), "bad codimensional node list");
        need(!c->opt->initialised, "ipcgpu_set_codim_nodes must come before ipcgpu_opt_init (dHat and kappa derive from the mesh extent and mass)");
        m.setCodimNodes(n, ids, mass, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_obstacle_nodes(ipcgpu_ctx* c, int n, const int* ids, int only)
{
    specChanged(c);
    return guarded([&] {
        bind(c);
        HipContact& k = CT(c);
        need(k.surfaceSet, "call ipcgpu_set_surface first");
        needArg(n >= 0 && (ids || !n), "bad obstacle node list");
        k.setObstacle(M(c).nV, n, ids, only != 0);
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_get(ipcgpu_ctx* c, int* a4, int* p4, int* pe2, int* cs2)
{
    return guarded([&] {
        bind(c);
        HipContact& k = CT(c);
        k.syncHost();
        for (size_t i = 0; a4 && i < k.active.size(); ++i)
            for (int j = 0; j < 4; ++j) a4[4 * i + j] = k.active[i][j];
        for (size_t i = 0; i < k.para.size(); ++i) {
            for (int j = 0; p4 && j < 4; ++j) p4[4 * i + j] = k.para[i][j];
            if (pe2) {
                pe2[2 * i] = k.paraEIEJ[i][0];
                pe2[2 * i + 1] = k.paraEIEJ[i][1];
            }
        }
        for (size_t i = 0; cs2 && i < k.csPTEE.size(); ++i) {
            cs2[2 * i] = k.csPTEE[i][0];
            cs2[2 * i + 1] = k.csPTEE[i][1];
        }
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_set(ipcgpu_ctx* c, int nA, const int* a4, int nP, const int* p4, const int* pe2)
{
    return guarded([&] {
        bind(c);
        HipContact& k = CT(c);
        need(k.surfaceSet, "call ipcgpu_set_surface first");
        needArg(nA >= 0 && nP >= 0 && (a4 || !nA) && ((p4 && pe2) || !nP), "bad constraint set");
        k.setSets(nA, a4, nP, p4, pe2);
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_counts(ipcgpu_ctx* c, int* counts3)
{
    return guarded([&] {
        needArg(c != nullptr && counts3 != nullptr, "null argument");
        HipContact& k = CT(c);
        counts3[0] = k.nActive();
        counts3[1] = k.nPara();
        counts3[2] = k.nCand();
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_energy(ipcgpu_ctx* c, double dHat, double kappa, double* E)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        *E = CT(c).energy(c->mesh->d_x.p, dHat, kappa, o.d_partial, o.d_scalar.p + 4);
        return IPCGPU_OK;
    });
}
// every node a kernel will read, checked per stencil kind exactly as the device decodes the tuple (hip_contact.hip decode()): c[0] >= 0: an edge-edge tuple, four
// nodes; else -c[0] - 1 and c[1] are nodes, c[2] is one unless negative (point-point), c[3] is one when c[2] and c[3] are non-negative (point-triangle); a negative
// c[3] behind a node c[2] is the multiplicity of a point-edge tuple
static void checkTuples(ipcgpu_ctx* c, int n, const int* t)
{
    const int nV = c->mesh->nV;
    auto node = [&](int v) { return v >= 0 && v < nV; };
    for (int i = 0; i < n; ++i) {
        const int* q = t + 4 * (size_t)i;
        bool ok;
        if (q[0] >= 0) ok = node(q[0]) && node(q[1]) && node(q[2]) && node(q[3]);
        else ok = node(-q[0] - 1) && node(q[1]) && (q[2] < 0 || (node(q[2]) && (q[3] < 0 || node(q[3]))));
        needArg(ok, "MMCVID node id out of range");
    }
}
int ipcgpu_contact_evaluate(ipcgpu_ctx* c, int n, const int* mmcvid_4n, double* val_n)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        bind(c);
        needArg(n >= 0 && (n == 0 || (mmcvid_4n && val_n)), "null argument");
        checkTuples(c, n, mmcvid_4n);
        CT(c).evaluateTuples(c->mesh->d_x.p, n, mmcvid_4n, val_n);
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_jt_multiply(ipcgpu_ctx* c, int n, const int* mmcvid_4n, const double* input_n, double coef, double* out_3nV_inout)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        bind(c);
        needArg(n >= 0 && out_3nV_inout && (n == 0 || (mmcvid_4n && input_n)), "null argument");
        checkTuples(c, n, mmcvid_4n);
        CT(c).jtMultiplyTuples(c->mesh->d_x.p, c->mesh->nV, n, mmcvid_4n, input_n, coef, out_3nV_inout);
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_gradient_add(ipcgpu_ctx* c, double dHat, double kappa, int projectDBC, double* g)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        const size_t n3 = 3 * (size_t)c->mesh->nV;
        HIP_CHECK(hipMemcpyAsync(o.d_gradient.p, g, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        CT(c).gradientAdd(c->mesh->d_x.p, c->mesh->d_dbc.p, c->mesh->nV, dHat, kappa, projectDBC, o.d_gradient.p);
        o.d_gradient.download(g, n3, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_hessian_add(ipcgpu_ctx* c, double dHat, double kappa, int projectDBC)
{
    return guarded([&] {
        M(c);
        bind(c);
        need(c->lin->numRows == 3 * c->mesh->nV, "call ipcgpu_linsys_set_pattern first");
        CT(c).hessianAdd(c->mesh->d_x.p, c->mesh->d_dbc.p, *c->lin, dHat, kappa, projectDBC, c->lin->d_a.p);
        return IPCGPU_OK;
    });
}
int ipcgpu_contact_connectivity(ipcgpu_ctx* c, int cap, int* pairs, int* n)
{
    return guarded([&] {
        std::vector<std::pair<int, int>> p;
        CT(c).connectivity(p);
        if (n) *n = (int)p.size();
        for (size_t i = 0; pairs && i < p.size() && (int)i < cap; ++i) {
            pairs[2 * i] = p[i].first;
            pairs[2 * i + 1] = p[i].second;
        }
        return IPCGPU_OK;
    });
}

int ipcgpu_ccd_partial(ipcgpu_ctx* c, const double* p, double slackness, double* step, int* pair2)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(p && step && slackness > 0 && slackness < 1, "bad ccd argument");
        HIP_CHECK(hipMemcpyAsync(o.d_searchDir.p, p, 3 * (size_t)c->mesh->nV * sizeof(double), hipMemcpyHostToDevice, c->stream));
        *step = CT(c).ccdPartial(c->mesh->d_x.p, o.d_searchDir.p, slackness, *step, pair2);
        return IPCGPU_OK;
    });
}
int ipcgpu_ccd_full(ipcgpu_ctx* c, const double* p, double slackness, double* step, int* pair2, int* nCand)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(p && step && slackness > 0 && slackness < 1, "bad ccd argument");
        HIP_CHECK(hipMemcpyAsync(o.d_searchDir.p, p, 3 * (size_t)c->mesh->nV * sizeof(double), hipMemcpyHostToDevice, c->stream));
        *step = CT(c).ccdFull(*c->mesh, c->mesh->d_x.p, o.d_searchDir.p, c->mesh->d_dbc.p, slackness, *step, pair2, nCand);
        return IPCGPU_OK;
    });
}
int ipcgpu_ccd_full_reference(ipcgpu_ctx* c, const double* p, double slackness, double* step, double* alphaCapped, int* arg3, int* nCand)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(p && step && slackness > 0 && slackness < 1, "bad ccd argument");
        HIP_CHECK(hipMemcpyAsync(o.d_searchDir.p, p, 3 * (size_t)c->mesh->nV * sizeof(double), hipMemcpyHostToDevice, c->stream));
        *step = CT(c).ccdFullReference(*c->mesh, c->mesh->d_x.p, o.d_searchDir.p, c->mesh->d_dbc.p, slackness, *step, alphaCapped, arg3, nCand);
        return IPCGPU_OK;
    });
}
int ipcgpu_set_ccd_mode(ipcgpu_ctx* c, int mode)
{
    specChanged(c);
    return guarded([&] {
        needArg(mode == 0 || mode == 1, "ccd mode is 0 (swept boxes, PT / EE pairs) or 1 (the reference's sweep)");
        CT(c).ccdMode = mode;
        return IPCGPU_OK;
    });
}
int ipcgpu_is_intersected(ipcgpu_ctx* c, int* flag)
{
    return guarded([&] {
        M(c);
        bind(c);
        *flag = CT(c).isIntersected(*c->mesh, c->mesh->d_x.p, c->mesh->d_dbc.p) ? 1 : 0;
        return IPCGPU_OK;
    });
}

// ---- Optimizer building blocks ----------------------------------------------------------------------
int ipcgpu_assemble_newton(ipcgpu_ctx* c, double dtSq, int projectDBC, double* grad)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        const double keep = o.dtSq;
        o.dtSq = dtSq;
        o.computePrecondMtr(projectDBC != 0, grad != nullptr);
        o.dtSq = keep;
        if (grad) o.d_gradient.download(grad, 3 * (size_t)c->mesh->nV, c->stream);
        else HIP_CHECK(hipStreamSynchronize(c->stream));
        if (o.selfCollision && o.contact) o.contact->takeHessianError(); // (the stepper leaves the barrier Hessian's "pair outside the pattern" flag to its next synchronisation: here)
        return IPCGPU_OK;
    });
}
int ipcgpu_incremental_potential(ipcgpu_ctx* c, double dtSq, double* E)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        const double keep = o.dtSq;
        o.dtSq = dtSq;
        *E = o.computeEnergyVal();
        o.dtSq = keep;
        return IPCGPU_OK;
    });
}
int ipcgpu_gradient(ipcgpu_ctx* c, double dtSq, int projectDBC, double* grad)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        const double keep = o.dtSq;
        o.dtSq = dtSq;
        o.computeGradient(projectDBC != 0);
        o.dtSq = keep;
        o.d_gradient.download(grad, 3 * (size_t)c->mesh->nV, c->stream);
        return IPCGPU_OK;
    });
}

// ---- Optimizer --------------------------------------------------------------------------------------
int ipcgpu_opt_init(ipcgpu_ctx* c, double dt, int withGravity)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        needArg(dt > 0, "dt must be positive");
        o.rank = c->rank;
        o.worldSize = c->worldSize;
        o.init(dt, withGravity != 0);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_rel_tol(ipcgpu_ctx* c, double tol)
{
    specChanged(c);
    return guarded([&] {
        needArg(tol > 0, "relTol must be positive"); // Optimizer.cpp:392
        O(c).setRelGL2Tol(tol);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_twist(ipcgpu_ctx* c, int nL, const int* l, int nR, const int* r, double angVel)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        for (int i = 0; i < nL; ++i) needArg(l[i] >= 0 && l[i] < c->mesh->nV, "handle id out of range");
        for (int i = 0; i < nR; ++i) needArg(r[i] >= 0 && r[i] < c->mesh->nV, "handle id out of range");
        o.setTwist(nL, l, nR, r, angVel);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_enable_self_collision(ipcgpu_ctx* c, double dHatEps)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(dHatEps > 0, "dHatEps must be positive");
        o.enableSelfCollision(&CT(c), dHatEps);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_pattern_lookahead(ipcgpu_ctx* c, double pad)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        needArg(pad >= 0.0 && pad <= 100.0, "pattern look-ahead: 0 <= pad <= 100 (in units of dHat)");
        o.patternPad = pad;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_add_half_space(ipcgpu_ctx* c, const double* origin, const double* normal, double dHatEps, int* id)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(origin && normal && dHatEps > 0, "bad half-space argument");
        const int k = o.addHalfSpace(&CT(c), origin, normal, dHatEps);
        if (id) *id = k;
        return IPCGPU_OK;
    });
}
static HipHalfSpace& HS(ipcgpu_ctx* c, int id)
{
    HipOptimizer& o = O(c);
    needArg(id >= 0 && id < (int)o.planes.size(), "half-space index out of range");
    return *o.planes[id];
}
int ipcgpu_halfspace_build(ipcgpu_ctx* c, int id, double dHat, int cap, int* verts, int* n)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        bind(c);
        const int cnt = h.build(CT(c).nSVI, CT(c).d_SVI.p, c->mesh->d_x.p, c->mesh->d_dbc.p, dHat);
        if (n) *n = cnt;
        needArg(!verts || cap >= cnt, "verts buffer too small");
        if (verts) std::memcpy(verts, h.set.data(), sizeof(int) * cnt);
        return IPCGPU_OK;
    });
}
int ipcgpu_halfspace_set(ipcgpu_ctx* c, int id, int n, const int* verts)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        bind(c);
        for (int i = 0; i < n; ++i) needArg(verts[i] >= 0 && verts[i] < c->mesh->nV, "vertex id out of range");
        h.setSet(n, verts);
        return IPCGPU_OK;
    });
}
int ipcgpu_halfspace_energy(ipcgpu_ctx* c, int id, double dHat, double kappa, double* E)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        bind(c);
        *E = h.energy(c->mesh->d_x.p, dHat, kappa);
        return IPCGPU_OK;
    });
}
int ipcgpu_halfspace_gradient_add(ipcgpu_ctx* c, int id, double dHat, double kappa, double* g)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        HipOptimizer& o = O(c);
        bind(c);
        const size_t n3 = 3 * (size_t)c->mesh->nV;
        HIP_CHECK(hipMemcpyAsync(o.d_gradient.p, g, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        h.gradientAdd(c->mesh->d_x.p, dHat, kappa, o.d_gradient.p);
        o.d_gradient.download(g, n3, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_halfspace_hessian_add(ipcgpu_ctx* c, int id, double dHat, double kappa, int projectDBC)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        bind(c);
        need(c->lin->numRows == 3 * c->mesh->nV, "call ipcgpu_linsys_set_pattern first");
        h.hessianAdd(c->mesh->d_x.p, c->mesh->d_dbc.p, c->lin->d_rowBase.p, c->lin->d_rowLen.p, dHat, kappa, projectDBC, c->lin->d_a.p);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return IPCGPU_OK;
    });
}
int ipcgpu_halfspace_move(ipcgpu_ctx* c, int id, const double* delta3, double slackness, double* stepSizeLeft)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        bind(c);
        needArg(delta3 && slackness > 0.0 && slackness <= 1.0, "bad half-space move");
        const double left = h.move(CT(c).nSVI, CT(c).d_SVI.p, c->mesh->d_x.p, delta3, slackness);
        if (stepSizeLeft) *stepSizeLeft = left;
        return IPCGPU_OK;
    });
}
int ipcgpu_halfspace_step_bound(ipcgpu_ctx* c, int id, const double* p, double slackness, double* step)
{
    return guarded([&] {
        HipHalfSpace& h = HS(c, id);
        HipOptimizer& o = O(c);
        bind(c);
        needArg(p && step && slackness > 0 && slackness <= 1, "bad step-bound argument");
        HIP_CHECK(hipMemcpyAsync(o.d_searchDir.p, p, 3 * (size_t)c->mesh->nV * sizeof(double), hipMemcpyHostToDevice, c->stream));
        *step = h.stepBound(CT(c).nSVI, CT(c).d_SVI.p, c->mesh->d_x.p, c->mesh->d_dbc.p, o.d_searchDir.p, slackness, *step);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_friction(ipcgpu_ctx* c, double selfFric, int fricIterAmt, double epsV)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        needArg(selfFric >= 0.0 && epsV > 0.0, "bad friction parameters");
        o.selfFric = selfFric;
        o.fricIterAmt = fricIterAmt;
        o.epsV = epsV;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_friction_target(ipcgpu_ctx* c, double epsVTarget)
{
    specChanged(c);
    return guarded([&] {
        O(c).epsVTarget = epsVTarget > 0.0 ? epsVTarget : -1.0;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_constructor_dt(ipcgpu_ctx* c, double h)
{
    specChanged(c);
    return guarded([&] {
        needArg(h > 0.0, "the step size must be positive");
        O(c).ctorDt = h;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_parameter_scaling(ipcgpu_ctx* c, int useAbsParameters, double dTolRel, double kappaMinMultiplier)
{
    specChanged(c);
    return guarded([&] {
        O(c).setParameterScaling(useAbsParameters != 0, dTolRel, kappaMinMultiplier);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_kappa(ipcgpu_ctx* c, double kappa)
{
    specChanged(c);
    return guarded([&] {
        needArg(kappa >= 0.0, "negative barrier stiffness");
        O(c).kappaConfig = kappa;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_dhat_target(ipcgpu_ctx* c, double dHatTargetEps)
{
    specChanged(c);
    return guarded([&] {
        needArg(dHatTargetEps == dHatTargetEps, "dHat target is not a number");
        O(c).dHatTargetEps = dHatTargetEps;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_damping(ipcgpu_ctx* c, double dampingStiff)
{
    specChanged(c);
    return guarded([&] {
        needArg(dampingStiff == dampingStiff, "damping stiffness is not a number");
        O(c).setDamping(dampingStiff);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_friction_scales(ipcgpu_ctx* c, double scaleSelf, double scaleObstacle)
{
    specChanged(c);
    return guarded([&] {
        needArg(scaleSelf >= 0.0 && scaleObstacle >= 0.0, "friction scales must not be negative");
        CT(c).fricScaleSelf = scaleSelf;
        CT(c).fricScaleObst = scaleObstacle;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_force_friction_loop(ipcgpu_ctx* c, int on)
{
    return guarded([&] {
        O(c).fricLoopForced = on != 0;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_half_space_friction(ipcgpu_ctx* c, int id, double mu)
{
    specChanged(c);
    return guarded([&] {
        needArg(mu >= 0.0, "negative friction coefficient");
        HS(c, id).friction = mu;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_next_subproblem(ipcgpu_ctx* c, int* more)
{
    specChanged(c);
    return guarded([&] {
        bind(c);
        const bool m = O(c).nextSubproblem();
        if (more) *more = m ? 1 : 0;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_friction_state(ipcgpu_ctx* c, double* sc, double* lambda)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        const size_t n = (o.contact && o.selfCollision) ? o.contact->fricSet.size() : 0;
        if (sc) {
            sc[0] = o.fricDHat;
            sc[1] = (double)n;
            sc[2] = (double)o.fricIterI;
            size_t nh = 0;
            for (const auto& h : o.planes) nh += h->lagSet.size();
            sc[3] = (double)nh;
        }
        if (lambda && n) o.contact->d_fricLambda.download(lambda, n, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_friction_update(ipcgpu_ctx* c, double dHat, double kappa, int* nLagged)
{
    return guarded([&] {
        M(c);
        bind(c);
        CT(c).frictionLagUpdate(c->mesh->d_x.p, dHat, kappa);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (nLagged) *nLagged = (int)CT(c).fricSet.size();
        return IPCGPU_OK;
    });
}
int ipcgpu_friction_get(ipcgpu_ctx* c, double* lambda, double* coord, double* basis)
{
    return guarded([&] {
        M(c);
        bind(c);
        needArg(lambda && coord && basis, "null output");
        CT(c).frictionGet(lambda, coord, basis);
        return IPCGPU_OK;
    });
}
int ipcgpu_friction_energy(ipcgpu_ctx* c, const double* Vt, double eps2, double coef, double* E)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(Vt && E && eps2 > 0, "bad argument");
        uploadColMajor(c, Vt, o.d_x0);
        *E = CT(c).frictionEnergy(c->mesh->d_x.p, o.d_x0.p, eps2, coef, o.d_partial, o.d_scalar.p + 4);
        return IPCGPU_OK;
    });
}
int ipcgpu_friction_gradient_add(ipcgpu_ctx* c, const double* Vt, double eps2, double coef, double* g)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(Vt && g && eps2 > 0, "bad argument");
        const size_t n3 = 3 * (size_t)c->mesh->nV;
        uploadColMajor(c, Vt, o.d_x0);
        HIP_CHECK(hipMemcpyAsync(o.d_gradient.p, g, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        CT(c).frictionGradientAdd(c->mesh->d_x.p, o.d_x0.p, eps2, coef, o.d_gradient.p);
        o.d_gradient.download(g, n3, c->stream);
        return IPCGPU_OK;
    });
}
int ipcgpu_friction_hessian_add(ipcgpu_ctx* c, const double* Vt, double eps2, double coef, int projectDBC)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        need(c->lin->numRows == 3 * c->mesh->nV, "call ipcgpu_linsys_set_pattern first");
        needArg(Vt && eps2 > 0, "bad argument");
        uploadColMajor(c, Vt, o.d_x0);
        CT(c).frictionHessianAdd(c->mesh->d_x.p, o.d_x0.p, c->mesh->d_dbc.p, *c->lin, eps2, coef, projectDBC, c->lin->d_a.p);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_velocity(ipcgpu_ctx* c, const double* vel)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(vel != nullptr, "null velocity");
        o.setVelocity(vel);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_warm_start(ipcgpu_ctx* c, int option)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        needArg(option >= 0 && option <= 5, "warmStart option must be 0..5");
        o.warmStart = option;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_warm_step(ipcgpu_ctx* c, double* out)
{
    return guarded([&] {
        if (out) *out = O(c).warmStepSize;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_time_integration(ipcgpu_ctx* c, int type, double beta, double gamma)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(type == 0 || type == 1, "time integration: 0 = BE, 1 = NM");
        needArg(type == 0 || (beta > 0.0 && gamma >= 0.0), "Newmark needs beta > 0, gamma >= 0");
        o.setTimeIntegration(type, beta, gamma);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_add_dirichlet(ipcgpu_ctx* c, int n, const int* ids, const double* lin3, const double* ang3, double t0, double t1)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(n > 0 && ids && lin3 && ang3, "empty Dirichlet group");
        for (int i = 0; i < n; ++i) needArg(ids[i] >= 0 && ids[i] < c->mesh->nV, "vertex id out of range");
        o.addDirichletBC(n, ids, lin3, ang3, t0, t1);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_end_dirichlet(ipcgpu_ctx* c, int group, double t_end)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        needArg(group >= 0 && group < (int)o.dbcGroups.size(), "no such Dirichlet group");
        o.dbcGroups[group]->t1 = std::min(o.dbcGroups[group]->t1, t_end);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_dirichlet_motion(ipcgpu_ctx* c, int group, const double* lin3, const double* ang3, const double* center3, int forceNonzero)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        needArg(group >= 0 && group < (int)o.dbcGroups.size(), "no such Dirichlet group");
        needArg(lin3 && ang3, "null velocity");
        auto& g = *o.dbcGroups[group];
        for (int k = 0; k < 3; ++k) {
            g.lin[k] = lin3[k];
            g.ang[k] = ang3[k];
            if (center3) g.center[k] = center3[k];
        }
        g.hasCenter = center3 != nullptr;
        g.forceNonzero = forceNonzero != 0;
        o.setDBCVertices();
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_dirichlet_targets(ipcgpu_ctx* c, int group, int n, const double* targets)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        needArg(group >= 0 && group < (int)o.dbcGroups.size(), "no such Dirichlet group");
        auto& g = *o.dbcGroups[group];
        if (!targets) {
            g.hasTargets = false;
            return IPCGPU_OK;
        }
        needArg(n == (int)g.ids.size(), "one target position per node of the group");
        g.d_targets.upload(targets, 3 * (size_t)n, c->stream);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        g.hasTargets = true;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_add_neumann(ipcgpu_ctx* c, int n, const int* ids, const double* accel3, double t0, double t1)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(n > 0 && ids && accel3, "empty Neumann group");
        for (int i = 0; i < n; ++i) needArg(ids[i] >= 0 && ids[i] < c->mesh->nV, "vertex id out of range");
        o.addNeumannBC(n, ids, accel3, t0, t1);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_dbc_state(ipcgpu_ctx* c, double* out4)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(out4 != nullptr, "null output");
        o.getDbcState(out4);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_kinematics(ipcgpu_ctx* c, double* vel, double* acc, double* dx)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        o.getKinematics(vel, acc, dx);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_save_status(ipcgpu_ctx* c, const char* path)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(path != nullptr && path[0] != 0, "empty path");
        o.saveStatus(path);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_load_status(ipcgpu_ctx* c, const char* path)
{
    specChanged(c);
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        needArg(path != nullptr && path[0] != 0, "empty path");
        o.loadStatus(path);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_contact_state(ipcgpu_ctx* c, int* counts6, int* pair2)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        need(o.ipOn(), "neither self collision nor a half-space is enabled");
        if (counts6) {
            counts6[0] = o.selfCollision ? o.contact->nActive() : 0;
            counts6[1] = o.selfCollision ? o.contact->nPara() : 0;
            counts6[2] = o.selfCollision ? o.contact->nCand() : 0;
            counts6[3] = 0;
            for (const auto& h : o.planes) counts6[3] += (int)h->set.size();
            counts6[4] = o.nFullCCD;
            counts6[5] = o.nPatternChanges;
        }
        if (pair2) {
            pair2[0] = o.lastCCDPair[0];
            pair2[1] = o.lastCCDPair[1];
        }
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_precompute(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        O(c).precompute();
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_begin_timestep(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        O(c).beginTimestep();
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_newton_iter(ipcgpu_ctx* c, int* converged)
{
    return guarded([&] {
        bind(c);
        const bool cv = O(c).newtonIter();
        if (converged) *converged = cv ? 1 : 0;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_end_timestep(ipcgpu_ctx* c)
{
    return guarded([&] {
        bind(c);
        O(c).endTimestep();
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_solve_timestep(ipcgpu_ctx* c, int maxIter, int* nIter)
{
    return guarded([&] {
        bind(c);
        const int n = O(c).solveTimestep(maxIter);
        if (nIter) *nIter = n;
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_state(ipcgpu_ctx* c, double* V, double* p, double* g, double* sc)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised, "call ipcgpu_opt_init first");
        const size_t n3 = 3 * (size_t)c->mesh->nV;
        if (V) downloadColMajor(c, c->mesh->d_x, V);
        if (p) o.d_searchDir.download(p, n3, c->stream);
        if (g) o.d_gradient.download(g, n3, c->stream);
        if (sc) {
            sc[0] = o.lastEnergyVal;
            sc[1] = o.lastStepSize;
            sc[2] = o.targetGRes;
            sc[3] = o.innerIterAmt;
            sc[4] = o.globalIterNum;
            sc[5] = o.lastAlphaFeasible;
            sc[6] = o.kappa;
            sc[7] = o.dHat;
        }
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_get_timers(ipcgpu_ctx* c, double* t)
{
    return guarded([&] {
        O(c).resolveEventTimers();
        std::memcpy(t, O(c).timers, sizeof(double) * 16);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_allreduce(ipcgpu_ctx* c, ipcgpu_allreduce_fn fn, void* user)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        c->opt->allreduce = fn;
        c->opt->allreduceUser = user;
        // a solver that was sharded before keeps copies of the hooks: refresh them (no new analysis)
        c->lin->setHooks(c->opt->allreduce, c->opt->allreduceUser, c->opt->allreduceStream, c->opt->allreduceStreamUser);
        return IPCGPU_OK;
    });
}

int ipcgpu_opt_set_allreduce_stream(ipcgpu_ctx* c, ipcgpu_allreduce_stream_fn fn, void* user)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        c->opt->allreduceStream = fn;
        c->opt->allreduceStreamUser = user; // its own slot: a host hook set earlier keeps its user pointer; detaching (fn = NULL) leaves that hook in charge
        c->lin->setHooks(c->opt->allreduce, c->opt->allreduceUser, c->opt->allreduceStream, c->opt->allreduceStreamUser);
        return IPCGPU_OK;
    });
}
static_assert(sizeof(ipcgpu_p2p_op) == sizeof(MfNumeric::P2POp), "ipcgpu_p2p_op and MfNumeric::P2POp must be one layout");
int ipcgpu_opt_set_exchange(ipcgpu_ctx* c, ipcgpu_exchange_fn fn, void* user)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        c->exchange = reinterpret_cast<void*>(fn);
        c->exchangeUser = user;
        c->lin->setExchangeHooks(reinterpret_cast<MfNumeric::ExchangeFn>(c->exchange), c->exchangeUser, reinterpret_cast<MfNumeric::ExchangeStreamFn>(c->exchangeStream),
            c->exchangeStreamUser);
        return IPCGPU_OK;
    });
}
int ipcgpu_opt_set_exchange_stream(ipcgpu_ctx* c, ipcgpu_exchange_stream_fn fn, void* user)
{
    return guarded([&] {
        needArg(c != nullptr, "null context");
        c->exchangeStream = reinterpret_cast<void*>(fn);
        c->exchangeStreamUser = user;
        c->lin->setExchangeHooks(reinterpret_cast<MfNumeric::ExchangeFn>(c->exchange), c->exchangeUser, reinterpret_cast<MfNumeric::ExchangeStreamFn>(c->exchangeStream),
            c->exchangeStreamUser);
        return IPCGPU_OK;
    });
}
int ipcgpu_ctx_get_stream(ipcgpu_ctx* c, void** hipStream)
{
    return guarded([&] {
        needArg(c != nullptr && hipStream != nullptr, "null argument");
        *hipStream = (void*)c->stream;
        return IPCGPU_OK;
    });
}

// ---- measurement ------------------------------------------------------------------------------------
int ipcgpu_bench_assembly(ipcgpu_ctx* c, double dtSq, int reps, double* avg_ms, double* bytes)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(o.initialised && !c->lin->rowBase.empty(), "needs opt_init + a pattern");
        needArg(reps > 0, "reps must be positive");
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        const ElemView v = o.view();
        o.ensurePatchPlan();
        int pb, pe;
        o.patchShard(pb, pe);
        double total = 0.0;
        for (int r = 0; r < reps; ++r) {
            HIP_CHECK(hipEventRecord(e0, c->stream));
            launch_assemble_patches(v, o.patch, pb, pe, dtSq, 1, o.d_gradient.p, c->lin->d_a.p, c->stream);
            HIP_CHECK(hipEventRecord(e1, c->stream));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = total / reps;
        const double nT = o.tetEnd - o.tetBegin;
        // B_asm = 112 nT + 84 nV + 8 nnz
        *bytes = 112.0 * nT + 84.0 * c->mesh->nV + 8.0 * (double)c->lin->ja.size();
        return IPCGPU_OK;
    });
}
int ipcgpu_bench_factor_solve(ipcgpu_ctx* c, int reps, double* fms, double* sms)
{
    return guarded([&] {
        HipOptimizer& o = O(c);
        bind(c);
        need(c->lin->analyzed(), "needs analyze_pattern");
        hipEvent_t e0, e1, e2;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventCreate(&e2));
        double tf = 0, ts = 0;
        for (int r = 0; r < reps; ++r) {
            HIP_CHECK(hipEventRecord(e0, c->stream));
            c->lin->factorize();
            HIP_CHECK(hipEventRecord(e1, c->stream));
            c->lin->solve(o.d_gradient.p, o.d_minusG.p);
            HIP_CHECK(hipEventRecord(e2, c->stream));
            HIP_CHECK(hipEventSynchronize(e2));
            float a = 0, b = 0;
            HIP_CHECK(hipEventElapsedTime(&a, e0, e1));
            HIP_CHECK(hipEventElapsedTime(&b, e1, e2));
            tf += a;
            ts += b;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipEventDestroy(e2);
        *fms = tf / reps;
        *sms = ts / reps;
        return IPCGPU_OK;
    });
}
int ipcgpu_bench_stream(ipcgpu_ctx* c, long long bytes, int reps, double* gbps)
{
    return guarded([&] {
        needArg(c && bytes > 0 && reps > 0, "bad argument");
        bind(c);
        DevBuf<double> a, b;
        a.alloc((size_t)bytes / 8);
        b.alloc((size_t)bytes / 8);
        a.zero(c->stream);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, c->stream));
        HIP_CHECK(hipEventRecord(e0, c->stream));
        for (int r = 0; r < reps; ++r) HIP_CHECK(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, c->stream));
        HIP_CHECK(hipEventRecord(e1, c->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *gbps = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9; // read + write
        return IPCGPU_OK;
    });
}
}
