// GPU multifrontal Cholesky, multi-GPU: what crosses ranks (packed update matrices / vectors, solution segments, the pivot flag) and the hooks that carry it.
// Split out of mf_numeric.hip in round 5.
#include <chrono>
#include <algorithm>
#include "mf_kernels.h"

namespace ipcgpu {

namespace {

// ---- subtree-sharded factorisation: what crosses ranks -----------------------------------------------------------------------
// desc = (front, staging offset lo, hi, offset of the update vector).  The update block of a front whose parent is executed by another rank (its lower
// triangle, packed: m (m + 1) / 2 doubles in the staging buffer, m = N - nc) is packed by the rank that computed it, sent to the parent's rank
// (MfNumeric::exchange: point to point) and unpacked there into the same front: the parent's extend-add then finds its child's contribution in place.
__global__ __launch_bounds__(256) void k_xchg_update(const int4* __restrict__ desc, TreeView tv, double* __restrict__ fronts, double* __restrict__ buf,
    int unpack)
{
    const int4 d = desc[blockIdx.y];
    const int s = d.x;
    const int N = frontN(tv, s), nc = frontNc(tv, s), m = N - nc;
    double* F = fronts + tv.frontOff[s];
    double* B = buf + (((long long)(unsigned)d.z << 32) | (unsigned)d.y);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)m * m; e += (long long)gridDim.x * 256) {
        const int j = (int)(e / m), i = (int)(e - (long long)j * m);
        if (i < j) continue;
        const long long t = (long long)j * m - (long long)j * (j - 1) / 2 + (i - j); // packed lower triangle, column by column
        if (unpack) F[(nc + i) + (long long)N * (nc + j)] = B[t];
        else B[t] = F[(nc + i) + (long long)N * (nc + j)];
    }
}
// the same for the update vectors of the forward sweep (rows >= nc of the front's work vector)
__global__ __launch_bounds__(256) void k_xchg_w(const int4* __restrict__ desc, TreeView tv, const long long* __restrict__ wOff, double* __restrict__ wbuf,
    double* __restrict__ buf, int unpack)
{
    const int4 d = desc[blockIdx.y];
    const int s = d.x;
    const int N = frontN(tv, s), nc = frontNc(tv, s), m = N - nc;
    double* w = wbuf + wOff[s] + nc;
    double* B = buf + d.w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        if (unpack) w[i] = B[i];
        else B[i] = w[i];
    }
}
// the solution: every rank keeps the entries of the fronts it executed, zeros elsewhere; the sum over the ranks is x
__global__ void k_mask_xsol(int nn, const int* __restrict__ nodeExec, int rank, double* __restrict__ xsol)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nn) return;
    if (nodeExec[i / 3] != rank) xsol[i] = 0.0;
}
__global__ void k_flag_to_double(const int* __restrict__ flag, double* __restrict__ buf) { buf[0] = flag[0] ? 1.0 : 0.0; }
__global__ void k_double_to_flag(const double* __restrict__ buf, int* __restrict__ flag) { flag[0] = (flag[0] || buf[0] > 0.0) ? 1 : 0; }

} // namespace

void MfNumeric::allreduceSum(double* dev, long long count)
{
    if (world_ <= 1 || count <= 0) return;
    commBytes_ += 8 * count;
    commCalls_++;
    if (allreduceStream_) { // stream-ordered (RCCL called from C on this stream): nothing to wait for on the host
        if (allreduceStream_(allreduceStreamUser_, dev, count, 0, (void*)stream_) != 0) throw HipError("all-reduce hook failed");
        return;
    }
    if (!allreduce_) throw StateError("sharded solver without an all-reduce hook (ipcgpu_opt_set_allreduce)");
    HIP_CHECK(hipStreamSynchronize(stream_)); // the hook works on the caller's stream: ours has to be drained first
    if (allreduce_(allreduceUser_, dev, count, 0) != 0) throw HipError("all-reduce hook failed");
}

void MfNumeric::exchange(const std::vector<P2POp>& ops)
{
    if (world_ <= 1 || ops.empty()) return;
    for (const P2POp& o : ops) {
        (o.send ? sentBytes_ : recvBytes_) += 8 * o.count;
        commBytes_ += 8 * o.count;
    }
    commCalls_++;
    if (exchangeStream_) { // stream-ordered (ncclSend / ncclRecv in one group on this stream): nothing to wait for on the host
        if (waitPending_.size() >= 256) (void)exchangeWaitMs(); // (drains the stream once every 256 groups: the events are read in bulk)
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (!waitFree_.empty()) {
            ev = waitFree_.back();
            waitFree_.pop_back();
        }
        else {
            HIP_CHECK(hipEventCreate(&ev.first));
            HIP_CHECK(hipEventCreate(&ev.second));
        }
        HIP_CHECK(hipEventRecord(ev.first, stream_));
        if (exchangeStream_(exchangeStreamUser_, (int)ops.size(), ops.data(), (void*)stream_) != 0) throw HipError("exchange hook failed");
        HIP_CHECK(hipEventRecord(ev.second, stream_));
        waitPending_.push_back(ev);
        return;
    }
    if (!exchange_) throw StateError("sharded solver without an exchange hook (ipcgpu_opt_set_exchange / ipcgpu_opt_set_exchange_stream)");
    HIP_CHECK(hipStreamSynchronize(stream_)); // the hook works on the caller's stream: ours has to be drained first
    const auto t0 = std::chrono::steady_clock::now();
    if (exchange_(exchangeUser_, (int)ops.size(), ops.data()) != 0) throw HipError("exchange hook failed");
    waitMs_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

double MfNumeric::exchangeWaitMs()
{
    for (auto& ev : waitPending_) {
        HIP_CHECK(hipEventSynchronize(ev.second));
        float ms = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
        waitMs_ += (double)ms;
        waitFree_.push_back(ev);
    }
    waitPending_.clear();
    return waitMs_;
}

void MfNumeric::criticalPath(double* out5) const
{
    if (!sym_) throw StateError("critical path before analyze_pattern");
    const MfSymbolic& sym = *sym_;
    const int nLevels = (int)sym.levelPtr.size() - 1;
    double steps = 0.0, above = 0.0, levelsAbove = 0.0;
    std::vector<double> load(std::max(world_, 1), 0.0);
    double below = 0.0;
    for (int l = 0; l < nLevels; ++l) {
        int widest = 0, widestAbove = 0;
        for (int q = sym.levelPtr[l]; q < sym.levelPtr[l + 1]; ++q) {
            const int s = sym.levelFronts[q], st = (sym.nc(s) + 31) / 32;
            widest = std::max(widest, st);
            const bool shared = world_ > 1 && owner_[s] < 0;
            if (shared) widestAbove = std::max(widestAbove, st);
            else {
                const double N = sym.N(s), nc = sym.nc(s);
                const double fl = nc * nc * nc / 3.0 + nc * nc * (N - nc) + nc * (N - nc) * (N - nc);
                load[world_ > 1 ? owner_[s] : 0] += fl;
                below += fl;
            }
        }
        steps += widest;
        above += widestAbove;
        if (widestAbove) levelsAbove += 1.0;
    }
    out5[0] = steps;
    out5[1] = above;
    out5[2] = below > 0.0 ? *std::max_element(load.begin(), load.end()) / below : 1.0;
    out5[3] = nLevels;
    out5[4] = levelsAbove;
}

// update matrices of level l whose parent another rank executes: packed by the rank that computed them, sent point to point, unpacked into the same front on
// the parent's rank (a bad pivot anywhere reaches everybody with the one-double all-reduce behind the factorisation)
void MfNumeric::exchangeUpdateMatrices(int l)
{
    const Xchg& X = xchg_[l];
    if (X.opsM.empty()) return;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p, dinvOff_.p };
    if (X.pack.cnt) hipLaunchKernelGGL(k_xchg_update, dim3(64, X.pack.cnt), dim3(256), 0, stream_, xchgDesc_.p + X.pack.off, tv, fronts_.p, xchgBuf_.p, 0);
    exchange(X.opsM);
    if (X.unpack.cnt) hipLaunchKernelGGL(k_xchg_update, dim3(64, X.unpack.cnt), dim3(256), 0, stream_, xchgDesc_.p + X.unpack.off, tv, fronts_.p, xchgBuf_.p, 1);
}

// the same for the update vectors of the forward sweep
void MfNumeric::exchangeUpdateVectors(int l)
{
    const Xchg& X = xchg_[l];
    if (X.opsW.empty()) return;
    TreeView tv{ frontOff_.p, idxPtr_.p, firstNode_.p, childPtr_.p, child_.p, invPtr_.p, inv_.p, idx_.p, dinvOff_.p };
    if (X.pack.cnt) hipLaunchKernelGGL(k_xchg_w, dim3(4, X.pack.cnt), dim3(256), 0, stream_, xchgDesc_.p + X.pack.off, tv, wOff_.p, w_.p, xchgBuf_.p, 0);
    exchange(X.opsW);
    if (X.unpack.cnt) hipLaunchKernelGGL(k_xchg_w, dim3(4, X.unpack.cnt), dim3(256), 0, stream_, xchgDesc_.p + X.unpack.off, tv, wOff_.p, w_.p, xchgBuf_.p, 1);
}

// the solution: every rank zeroes what it did not execute, the sum over the ranks is x
void MfNumeric::reduceSolution()
{
    const int n3 = sym_->n;
    hipLaunchKernelGGL(k_mask_xsol, dim3((n3 + 255) / 256), dim3(256), 0, stream_, sym_->nn, nodeExec_.p, rank_, xsol_.p);
    allreduceSum(xsol_.p, n3);
}

// "a non-positive pivot was met" on any rank -> on every rank: one double
void MfNumeric::allreduceFlag()
{
    hipLaunchKernelGGL(k_flag_to_double, dim3(1), dim3(1), 0, stream_, flag_.p, xchgBuf_.p);
    allreduceSum(xchgBuf_.p, 1);
    hipLaunchKernelGGL(k_double_to_flag, dim3(1), dim3(1), 0, stream_, xchgBuf_.p, flag_.p);
}

} // namespace ipcgpu
