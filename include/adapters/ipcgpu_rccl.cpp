// RCCL bound to a libipcgpu context from C (see include/ipcgpu_rccl.h).  This file is on the caller's side of the C ABI: it uses
// nothing of the library but ipcgpu.h.  Inside ipc-sim/IPC it would sit next to src/main.cpp, which owns the communicator.
#include <ipcgpu_rccl.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Binding {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};
std::mutex g_mu;
std::map<ipcgpu_ctx*, Binding*> g_bind;
thread_local std::string g_err;

int fail(const std::string& what)
{
    g_err = what;
    return IPCGPU_ERR_HIP;
}

// the hook: one ncclAllReduce on the stream the library hands over (its own)
int allreduce_on_stream(void* user, void* buf, long long count, int op, void* stream)
{
    Binding* b = static_cast<Binding*>(user);
    const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, op == 1 ? ncclMin : ncclSum, b->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : -1;
}
// the point-to-point hook of the sharded solver: one RCCL group of sends / receives on the library's stream
int exchange_on_stream(void* user, int nOps, const ipcgpu_p2p_op* ops, void* stream)
{
    Binding* b = static_cast<Binding*>(user);
    if (ncclGroupStart() != ncclSuccess) return -1;
    bool ok = true;
    for (int i = 0; i < nOps; ++i) {
        const ipcgpu_p2p_op& o = ops[i];
        const ncclResult_t r = o.send ? ncclSend(o.buf_dev, (size_t)o.count, ncclDouble, o.peer, b->comm, static_cast<hipStream_t>(stream))
                                      : ncclRecv(o.buf_dev, (size_t)o.count, ncclDouble, o.peer, b->comm, static_cast<hipStream_t>(stream));
        ok = ok && r == ncclSuccess;
    }
    return (ncclGroupEnd() == ncclSuccess && ok) ? 0 : -1;
}
} // namespace

extern "C" {

const char* ipcgpu_rccl_last_error(void) { return g_err.c_str(); }

int ipcgpu_rccl_unique_id(void* id128)
{
    static_assert(sizeof(ncclUniqueId) <= IPCGPU_RCCL_ID_BYTES, "ncclUniqueId does not fit the id buffer");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return fail("ncclGetUniqueId failed");
    std::vector<char> buf(IPCGPU_RCCL_ID_BYTES, 0);
    std::memcpy(buf.data(), &id, sizeof(id));
    std::memcpy(id128, buf.data(), IPCGPU_RCCL_ID_BYTES);
    return IPCGPU_OK;
}

int ipcgpu_rccl_attach(ipcgpu_ctx* ctx, int rank, int world, const void* id128)
{
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fail("ipcgpu_rccl_attach: bad arguments");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    Binding* b = new Binding;
    b->rank = rank;
    b->world = world;
    // the context was created on its device and made it current; the communicator lives on the same one
    const ncclResult_t r = ncclCommInitRank(&b->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete b;
        return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    if (ipcgpu_opt_set_allreduce_stream(ctx, allreduce_on_stream, b) != IPCGPU_OK || ipcgpu_opt_set_exchange_stream(ctx, exchange_on_stream, b) != IPCGPU_OK) {
        ipcgpu_opt_set_allreduce_stream(ctx, nullptr, nullptr);
        ncclCommDestroy(b->comm);
        delete b;
        return fail(std::string("ipcgpu_opt_set_allreduce_stream / ipcgpu_opt_set_exchange_stream: ") + ipcgpu_last_error());
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_bind[ctx] = b;
    return IPCGPU_OK;
}

int ipcgpu_rccl_detach(ipcgpu_ctx* ctx)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_bind.find(ctx);
    if (it == g_bind.end()) return IPCGPU_OK;
    ipcgpu_opt_set_allreduce_stream(ctx, nullptr, nullptr);
    ipcgpu_opt_set_exchange_stream(ctx, nullptr, nullptr);
    ncclCommDestroy(it->second->comm);
    delete it->second;
    g_bind.erase(it);
    return IPCGPU_OK;
}

int ipcgpu_rccl_selftest(ipcgpu_ctx* ctx, int rank, long long count, int op, double* result)
{
    Binding* b = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_bind.find(ctx);
        if (it == g_bind.end()) return fail("ipcgpu_rccl_selftest: context not attached");
        b = it->second;
    }
    void* stream = nullptr;
    if (ipcgpu_ctx_get_stream(ctx, &stream) != IPCGPU_OK) return fail(ipcgpu_last_error());
    std::vector<double> h((size_t)count, (double)(rank + 1));
    double* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(double) * (size_t)count) != hipSuccess) return fail("hipMalloc failed");
    (void)hipMemcpyAsync(d, h.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
    const int rc = allreduce_on_stream(b, d, count, op, stream);
    (void)hipMemcpyAsync(h.data(), d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    (void)hipFree(d);
    if (rc != 0) return fail("ncclAllReduce failed");
    *result = h[0];
    return IPCGPU_OK;
}

// every rank sends `count` doubles filled with (rank + 1) to rank + 1 and receives from rank - 1 (mod world) through the exchange hook; returns the
// first received value: ((rank - 1 + world) % world) + 1.  A smoke test of the point-to-point binding.
int ipcgpu_rccl_selftest_p2p(ipcgpu_ctx* ctx, int rank, int world, long long count, double* result)
{
    Binding* b = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_bind.find(ctx);
        if (it == g_bind.end()) return fail("ipcgpu_rccl_selftest_p2p: context not attached");
        b = it->second;
    }
    void* stream = nullptr;
    if (ipcgpu_ctx_get_stream(ctx, &stream) != IPCGPU_OK) return fail(ipcgpu_last_error());
    std::vector<double> h((size_t)count, (double)(rank + 1));
    double *ds = nullptr, *dr = nullptr;
    if (hipMalloc((void**)&ds, sizeof(double) * (size_t)count) != hipSuccess || hipMalloc((void**)&dr, sizeof(double) * (size_t)count) != hipSuccess) return fail("hipMalloc failed");
    (void)hipMemcpyAsync(ds, h.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
    int rc = 0;
    if (world > 1) {
        const ipcgpu_p2p_op ops[2] = { { ds, count, (rank + 1) % world, 1 }, { dr, count, (rank + world - 1) % world, 0 } };
        rc = exchange_on_stream(b, 2, ops, stream);
    }
    else (void)hipMemcpyAsync(dr, ds, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
    (void)hipMemcpyAsync(h.data(), dr, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    (void)hipFree(ds);
    (void)hipFree(dr);
    if (rc != 0) return fail("ncclSend / ncclRecv failed");
    *result = h[0];
    return IPCGPU_OK;
}

} // extern "C"
