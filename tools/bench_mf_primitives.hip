// Micro-timing of the building blocks of the multifrontal step kernel (cycles via s_memtime), to see which link of the
// per-step critical chain costs what.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I ipc_amd/csrc tools/bench_mf_primitives.hip -o /tmp/bmf && /tmp/bmf
#include "../ipc_amd/csrc/mf_numeric.hip"
#include <cstdio>
#include <vector>

using namespace ipcgpu;

// experiment: batches + the next pivot's reciprocal square root started as soon as its column has been updated
template <int BATCH>
__device__ __forceinline__ bool wave_potrf32c(double* blk, int ld, int w, int lane, double* rdiag)
{
    bool bad = false;
    double row[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) row[k] = (lane < w && k <= lane && k < w) ? blk[k * ld + (lane & (NB - 1))] : 0.0;
    double myRd = 0.0;
    double d0 = bcast_lane(row[0], 0);
    if (!(d0 > 0.0)) {
        bad = true;
        d0 = 1.0;
    }
    double invd = rsqrt_nr(d0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < w) {
            if (lane == j) myRd = invd;
            row[j] *= invd;
            double invdNext = 1.0;
            auto nextPivot = [&]() {
                if (j + 1 < w) {
                    double dn = bcast_lane(row[j + 1], j + 1);
                    if (!(dn > 0.0)) {
                        bad = true;
                        dn = 1.0;
                    }
                    invdNext = rsqrt_nr(dn);
                }
            };
#pragma unroll
            for (int j0 = j + 1, b = 0; j0 < NB; j0 += BATCH, ++b) {
                double m[BATCH];
                if (b == 1) nextPivot(); // column j + 1 is final after batch 0: its rsqrt chain runs under the later batches
#pragma unroll
                for (int q = 0; q < BATCH; ++q)
                    if (j0 + q < NB) m[q] = bcast_lane(row[j], j0 + q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BATCH; ++q)
                    if (j0 + q < NB) row[j0 + q] -= row[j] * m[q];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j + 1 + BATCH >= NB) nextPivot(); // single batch: after it
            invd = invdNext;
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (lane < w && k <= lane && k < w) blk[k * ld + lane] = row[k];
    if (lane < NB) rdiag[lane] = myRd;
    return bad;
}

// experiment: the broadcasts of a column issued in batches ahead of the FMAs that consume them (SGPR latency pipelined)
template <int BATCH>
__device__ __forceinline__ bool wave_potrf32b(double* blk, int ld, int w, int lane, double* rdiag)
{
    bool bad = false;
    double row[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) row[k] = (lane < w && k <= lane && k < w) ? blk[k * ld + (lane & (NB - 1))] : 0.0;
    double myRd = 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < w) {
            double djj = bcast_lane(row[j], j);
            if (!(djj > 0.0)) {
                bad = true;
                djj = 1.0;
            }
            const double invd = rsqrt_nr(djj);
            if (lane == j) myRd = invd;
            row[j] *= invd;
#pragma unroll
            for (int j0 = j + 1; j0 < NB; j0 += BATCH) {
                double m[BATCH];
#pragma unroll
                for (int q = 0; q < BATCH; ++q)
                    if (j0 + q < NB) m[q] = bcast_lane(row[j], j0 + q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BATCH; ++q)
                    if (j0 + q < NB) row[j0 + q] -= row[j] * m[q];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (lane < w && k <= lane && k < w) blk[k * ld + lane] = row[k];
    if (lane < NB) rdiag[lane] = myRd;
    return bad;
}

__global__ __launch_bounds__(WGB) void k_probe(double* A, long long* out, double* sink)
{
    __shared__ double blk[NB * LDP];
    __shared__ double rd[NB];
    __shared__ double Xs[NB * LDI];
    __shared__ double blk2[NB * LDP], blk3[NB * LDP];
    __shared__ double rd2[NB];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += WGB) blk[(e >> 5) * LDP + (e & 31)] = blk2[(e >> 5) * LDP + (e & 31)] = blk3[(e >> 5) * LDP + (e & 31)] = A[e];
    __syncthreads();
    long long u0 = __builtin_readcyclecounter();
    if (tid >= ROWS_B) (void)wave_potrf32c<8>(blk2, LDP, NB, tid - ROWS_B, rd2);
    __syncthreads();
    long long u1 = __builtin_readcyclecounter();
    if (tid >= ROWS_B) (void)wave_potrf32c<16>(blk3, LDP, NB, tid - ROWS_B, rd2);
    __syncthreads();
    long long u2 = __builtin_readcyclecounter();
    long long t0 = __builtin_readcyclecounter();
    if (tid >= ROWS_B) (void)wave_potrf32(blk, LDP, NB, tid - ROWS_B, rd);
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    double x2[2][NB];
    double (&x)[NB] = x2[0];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        x2[0][c] = A[(tid & 31) + 32 * c] + tid;
        x2[1][c] = x2[0][c] + 1.0;
    }
    long long t2 = __builtin_readcyclecounter();
    {
        double(&x1)[1][NB] = reinterpret_cast<double(&)[1][NB]>(x2[0]);
        if (tid < ROWS_B) row_trsm32<1>(x1, blk, LDP, rd);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NB; ++c) s += x2[0][c] + x2[1][c];
    sink[blockIdx.x * WGB + tid] = s;
    __syncthreads();
    long long t3 = __builtin_readcyclecounter();
    if (tid < 64) wave_trinv32(blk, LDP, tid, Xs);
    __syncthreads();
    long long t4 = __builtin_readcyclecounter();
    // 32 x 32 x 32 rank update of one row, as in role B
#pragma unroll 2
    for (int k = 0; k < NB; ++k) {
        const double ak = A[(tid & 31) * 32 + k];
        const double* lpk = blk + k * LDP;
#pragma unroll
        for (int c = 0; c < NB; ++c) x[c] -= ak * lpk[c];
    }
    s = 0;
#pragma unroll
    for (int c = 0; c < NB; ++c) s += x[c];
    sink[blockIdx.x * WGB + tid] += s + Xs[tid & 31];
    __syncthreads();
    long long t5 = __builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x == 0) {
        double md = 0.0;
        for (int k = 0; k < NB; ++k)
            for (int r = k; r < NB; ++r) md = fmax(md, fmax(fabs(blk[k * LDP + r] - blk2[k * LDP + r]), fabs(blk[k * LDP + r] - blk3[k * LDP + r])));
        sink[0] = md;
        out[4] = u1 - u0;
        out[5] = u2 - u1;
        out[0] = t1 - t0;
        out[1] = t3 - t2;
        out[2] = t4 - t3;
        out[3] = t5 - t4;
    }
}

int main()
{
    std::vector<double> A(1024, 0.0);
    for (int k = 0; k < 32; ++k)
        for (int r = k; r < 32; ++r) A[k * 32 + r] = (r == k) ? 40.0 + k : 1.0 / (1 + r + k);
    double *dA, *dS;
    long long* dO;
    hipMalloc(&dA, 8192);
    hipMalloc(&dS, 8 * WGB * 64);
    hipMalloc(&dO, 64);
    hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(rep == 2 ? 64 : 1), dim3(WGB), 0, 0, dA, dO, dS);
        long long o[6];
        double md;
        hipMemcpy(o, dO, 48, hipMemcpyDeviceToHost);
        hipMemcpy(&md, dS, 8, hipMemcpyDeviceToHost);
        std::printf("early next pivot: c<8> -> %lld, c<16> -> %lld ticks, max |dL| %.2e\n", o[4], o[5], md);
        std::printf("grid %2d: potrf32 %lld  row_trsm32<1>(4 waves) %lld  trinv32 %lld  row_update32 %lld  (s_memtime ticks)\n",
            rep == 2 ? 64 : 1, o[0], o[1], o[2], o[3]);
    }
    return 0;
}
