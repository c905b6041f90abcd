// What a dependency between workgroups costs on MI355X, measured four ways (round 4: the design question behind a persistent
// step kernel for the top separators of the multifrontal factorisation):
//   1. a kernel boundary: K dependent launches of a small kernel on one stream
//   2. cooperative_groups::grid_group::sync() in a cooperative launch
//   3. a hand-rolled grid barrier (one agent-scope counter, sense by generation), cooperative launch for co-residency
//   4. a point-to-point flag between two workgroups (release store / acquire spin), same XCD and different XCDs
// Every spin loop carries an iteration cap and gives up (sets an error flag) rather than hang the box.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_sync.hip -o tools/_build/bench_sync && tools/_build/bench_sync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace cg = cooperative_groups;

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr long long SPIN_CAP = 4000000; // ~ tens of ms: a lost signal ends the kernel with err = 1

__global__ void k_tiny(double* p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}

__global__ void k_cg_sync(int rounds, double* p)
{
    cg::grid_group g = cg::this_grid();
    double v = p[blockIdx.x];
    for (int r = 0; r < rounds; ++r) {
        v = v * 1.0000001 + 1.0;
        g.sync();
    }
    if (threadIdx.x == 0) p[blockIdx.x] = v;
}

// counter barrier: generation g is complete when the counter reaches g * gridDim.x
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* err)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE); // agent scope by default for global memory
        long long spins = 0;
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_CAP) {
                *err = 1;
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ void k_hand_barrier(int rounds, double* p, unsigned* ctr, int* err)
{
    double v = p[blockIdx.x];
    for (int r = 0; r < rounds; ++r) {
        v = v * 1.0000001 + 1.0;
        if (!grid_barrier(ctr, (unsigned)(r + 1) * gridDim.x, err)) break;
        if (*(volatile int*)err) break;
    }
    if (threadIdx.x == 0) p[blockIdx.x] = v;
}

// ping-pong between workgroup a and workgroup b (all others exit at once): flag[0] a -> b, flag[1] b -> a
__global__ void k_pingpong(int rounds, int a, int b, unsigned* flag, int* err, long long* cycles, int* xcc)
{
    const int me = blockIdx.x;
    if (threadIdx.x == 0) {
        // XCC_ID: hardware register 20 on gfx94x / gfx950, low 4 bits
        xcc[me] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    }
    if (me != a && me != b) return;
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (me == a) {
            __atomic_store_n(&flag[0], (unsigned)r, __ATOMIC_RELEASE);
            long long spins = 0;
            while (__atomic_load_n(&flag[1], __ATOMIC_ACQUIRE) < (unsigned)r)
                if (++spins > SPIN_CAP) {
                    *err = 1;
                    return;
                }
        }
        else {
            long long spins = 0;
            while (__atomic_load_n(&flag[0], __ATOMIC_ACQUIRE) < (unsigned)r)
                if (++spins > SPIN_CAP) {
                    *err = 1;
                    return;
                }
            __atomic_store_n(&flag[1], (unsigned)r, __ATOMIC_RELEASE);
        }
    }
    if (me == a) cycles[0] = wall_clock64() - t0;
}

// the same with a payload: the producer writes 8 KB (a 32 x 32 block of doubles), then the flag; the consumer reads all of it after the flag
__global__ void k_pingpong_payload(int rounds, int a, int b, unsigned* flag, double* buf, int* err, long long* cycles)
{
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    const int t = threadIdx.x;
    double acc = 0.0;
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        double* mine = buf + (me == a ? 0 : 1024);
        double* theirs = buf + (me == a ? 1024 : 0);
        if (me == a) {
            for (int e = t; e < 1024; e += blockDim.x) mine[e] = acc + e + r;
            __syncthreads();
            if (t == 0) {
                __atomic_store_n(&flag[0], (unsigned)r, __ATOMIC_RELEASE);
                long long spins = 0;
                while (__atomic_load_n(&flag[1], __ATOMIC_ACQUIRE) < (unsigned)r)
                    if (++spins > SPIN_CAP) {
                        *err = 1;
                        break;
                    }
            }
            __syncthreads();
            if (*(volatile int*)err) return;
            for (int e = t; e < 1024; e += blockDim.x) acc += __builtin_nontemporal_load(theirs + e) * 1e-9;
        }
        else {
            if (t == 0) {
                long long spins = 0;
                while (__atomic_load_n(&flag[0], __ATOMIC_ACQUIRE) < (unsigned)r)
                    if (++spins > SPIN_CAP) {
                        *err = 1;
                        break;
                    }
            }
            __syncthreads();
            if (*(volatile int*)err) return;
            for (int e = t; e < 1024; e += blockDim.x) acc += __builtin_nontemporal_load(theirs + e) * 1e-9;
            for (int e = t; e < 1024; e += blockDim.x) mine[e] = acc + e - r;
            __syncthreads();
            if (t == 0) __atomic_store_n(&flag[1], (unsigned)r, __ATOMIC_RELEASE);
        }
    }
    if (me == a && t == 0) cycles[0] = wall_clock64() - t0;
    if (acc == 12345.678) buf[2048] = acc;
}

static double elapsed_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int wallKHz = 0;
    CK(hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0));
    printf("device %s, %d CUs, cooperative launch %d, wall clock %d kHz, shader clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.cooperativeLaunch, wallKHz,
        prop.clockRate);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* p;
    CK(hipMalloc(&p, 1 << 24));
    CK(hipMemset(p, 0, 1 << 24));
    unsigned* ctr;
    CK(hipMalloc(&ctr, 4096));
    int* err;
    CK(hipMalloc(&err, 4096));
    long long* cyc;
    CK(hipMalloc(&cyc, 4096));
    int* xcc;
    CK(hipMalloc(&xcc, 4096 * sizeof(int)));
    const int K = 400;

    // 1. kernel boundaries
    for (int wgs : { 1, 64, 256, 512 }) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_tiny, dim3(wgs), dim3(256), 0, st, p, wgs * 256);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            if (rep) printf("launch chain       %4d workgroups: %.2f us per dependent launch\n", wgs, 1e3 * elapsed_ms(e0, e1) / K);
        }
    }
    // 2. cooperative grid sync, 3. hand-rolled barrier
    for (int wgs : { 8, 32, 64, 128, 256, 512, 1024 }) {
        int rounds = K;
        {
            void* args[] = { &rounds, &p };
            int maxB = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxB, k_cg_sync, 256, 0));
            if (wgs > maxB * prop.multiProcessorCount) {
                printf("cg grid.sync       %4d workgroups: does not fit (%d per CU)\n", wgs, maxB);
            }
            else
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, st));
                    CK(hipLaunchCooperativeKernel((const void*)k_cg_sync, dim3(wgs), dim3(256), args, 0, st));
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    if (rep) printf("cg grid.sync       %4d workgroups: %.2f us per sync\n", wgs, 1e3 * elapsed_ms(e0, e1) / K);
                }
        }
        {
            void* args[] = { &rounds, &p, &ctr, &err };
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 64, st));
                CK(hipMemsetAsync(err, 0, 64, st));
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel((const void*)k_hand_barrier, dim3(wgs), dim3(256), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                int h = 0;
                CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                if (rep) printf("counter barrier    %4d workgroups: %.2f us per barrier%s\n", wgs, 1e3 * elapsed_ms(e0, e1) / K, h ? "  (GAVE UP)" : "");
            }
        }
    }
    // 4. point-to-point flags
    {
        const int wgs = 64;
        std::vector<int> hx(wgs);
        const int pairs[][2] = { { 0, 1 }, { 0, 8 }, { 0, 16 }, { 0, 4 }, { 3, 43 } };
        for (auto& pr : pairs) {
            int rounds = 2000, a = pr[0], b = pr[1];
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 64, st));
                CK(hipMemsetAsync(err, 0, 64, st));
                void* args[] = { &rounds, &a, &b, &ctr, &err, &cyc, &xcc };
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel((const void*)k_pingpong, dim3(wgs), dim3(64), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                int h = 0;
                CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hx.data(), xcc, wgs * sizeof(int), hipMemcpyDeviceToHost));
                if (rep)
                    printf("flag ping-pong     workgroups %2d (xcc %d) <-> %2d (xcc %d): %.3f us per one-way signal%s\n", a, hx[a] & 15, b, hx[b] & 15,
                        1e3 * elapsed_ms(e0, e1) / rounds / 2, h ? "  (GAVE UP)" : "");
            }
            double* buf = p + (1 << 16);
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 64, st));
                CK(hipMemsetAsync(err, 0, 64, st));
                void* args[] = { &rounds, &a, &b, &ctr, &buf, &err, &cyc };
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel((const void*)k_pingpong_payload, dim3(wgs), dim3(256), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                int h = 0;
                CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                if (rep)
                    printf("flag + 8 KB block  workgroups %2d <-> %2d: %.3f us per one-way hand-over%s\n", a, b, 1e3 * elapsed_ms(e0, e1) / rounds / 2, h ? "  (GAVE UP)" : "");
            }
        }
        printf("xcc of workgroups 0..15:");
        for (int i = 0; i < 16; ++i) printf(" %d", hx[i] & 15);
        printf("\n");
    }
    return 0;
}
