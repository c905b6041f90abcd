// Tool (host only): where the flops of a multifrontal factorisation sit, by kernel class of mf_numeric.hip -- fused fronts, the 32-column steps
// of the big fronts (pivot + panel rows + trailing update inside the front's own columns), their Schur complements -- and the chain of
// dependent steps per level.  Built and driven by tools/mf_flops.py.
#include "../ipc_amd/csrc/mf_symbolic.h"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace ipcgpu;
extern "C" int mf_flops_report(int n, const int* ia, const int* ja, const double* coords, int leaf, int fusedMaxKids, double* out8, int* chain, int maxLevels)
{
    MfSymbolic s;
    mf_analyze(n, ia, ja, coords, leaf, s);
    double fused = 0, steps = 0, schur = 0, big = 0;
    int nFused = 0, nBig = 0;
    const int nLevels = (int)s.levelPtr.size() - 1;
    std::vector<int> maxSteps(nLevels, 0);
    for (int f = 0; f < s.ns; ++f) {
        const double N = s.N(f), nc = s.nc(f), m = N - nc;
        const int kids = s.childPtr[f + 1] - s.childPtr[f];
        const size_t lds = ((size_t)nc * (size_t)N + 64) * 8 + (size_t)kids * (size_t)N * 4;
        double own = 0; // factor of the nc x nc block + panel rows: sum_j (N - j - 1)^2 restricted to columns < nc ...
        for (int j = 0; j < (int)nc; ++j) {
            const double r = N - j - 1; // rows below the pivot
            const double c = nc - j - 1; // of which columns inside the front's own block
            own += 2 * r + 1 + 2 * (c * (c + 1) / 2 + c * m); // scale + rank-1 update inside [own columns] x [all rows]
        }
        const double sc = nc * m * (m + 1); // S -= L21 L21^T, lower triangle, 2 flops per multiply-add
        if (kids <= fusedMaxKids && lds <= 64 * 1024) {
            fused += own + sc;
            ++nFused;
        }
        else {
            steps += own;
            schur += sc;
            big += 1;
            ++nBig;
            maxSteps[s.level[f]] = std::max(maxSteps[s.level[f]], ((int)nc + 31) / 32);
        }
    }
    out8[0] = s.flops;
    out8[1] = fused;
    out8[2] = steps;
    out8[3] = schur;
    out8[4] = nFused;
    out8[5] = nBig;
    out8[6] = (double)s.nnzL;
    out8[7] = nLevels;
    for (int l = 0; l < std::min(nLevels, maxLevels); ++l) chain[l] = maxSteps[l];
    return nLevels;
}

// per level of the assembly tree: big fronts, their widths, update-block sizes, 64 x 64 Schur tiles and Schur flops (printed; tools/mf_flops.py --levels)
extern "C" void mf_level_report(int n, const int* ia, const int* ja, const double* coords, int leaf, int fusedMaxKids)
{
    MfSymbolic s;
    mf_analyze(n, ia, ja, coords, leaf, s);
    const int nLevels = (int)s.levelPtr.size() - 1;
    std::printf("level  big  fused |  nc min..max   N-nc min..max | tiles64  schur GFLOP  step GFLOP | fused GFLOP\n");
    for (int l = 0; l < nLevels; ++l) {
        int nb = 0, nf = 0, ncMin = 1 << 30, ncMax = 0, mMin = 1 << 30, mMax = 0;
        long long tiles = 0;
        double schur = 0, steps = 0, fused = 0;
        for (int i = s.levelPtr[l]; i < s.levelPtr[l + 1]; ++i) {
            const int f = s.levelFronts[i];
            const double N = s.N(f), nc = s.nc(f), m = N - nc;
            const int kids = s.childPtr[f + 1] - s.childPtr[f];
            const size_t lds = ((size_t)nc * (size_t)N + 64) * 8 + (size_t)kids * (size_t)N * 4;
            double own = 0;
            for (int j = 0; j < (int)nc; ++j) {
                const double r = N - j - 1, c = nc - j - 1;
                own += 2 * r + 1 + 2 * (c * (c + 1) / 2 + c * m);
            }
            const double sc = nc * m * (m + 1);
            if (kids <= fusedMaxKids && lds <= 64 * 1024) {
                ++nf;
                fused += own + sc;
                continue;
            }
            ++nb;
            ncMin = std::min(ncMin, (int)nc);
            ncMax = std::max(ncMax, (int)nc);
            mMin = std::min(mMin, (int)m);
            mMax = std::max(mMax, (int)m);
            const long long nt = ((long long)m + 63) / 64;
            tiles += nt * (nt + 1) / 2;
            schur += sc;
            steps += own;
        }
        if (nb) std::printf("%5d %4d %6d | %5d..%-5d  %6d..%-6d | %7lld  %11.3f  %10.3f | %10.3f\n", l, nb, nf, ncMin, ncMax, mMin, mMax, tiles, schur / 1e9, steps / 1e9, fused / 1e9);
        else std::printf("%5d %4d %6d |                                 |                                   | %10.3f\n", l, nb, nf, fused / 1e9);
    }
}
