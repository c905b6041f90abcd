// Tet-mesh file readers / writer of the reference's IglUtils (src/Utils/IglUtils.cpp:203-233, 300-361, 451-584), host only.
#pragma once
#include <string>
#include <vector>

namespace ipcgpu {

struct TetMeshFile {
    std::vector<double> V; // xyz interleaved, file order
    std::vector<int> T; // 4 node ids per tet (0-based)
    std::vector<int> SF; // 3 node ids per surface triangle (0-based), outward for positively oriented tets
};

void readTetMesh(const std::string& path, TetMeshFile& m, bool findSurface);
void saveTetMesh(const std::string& path, int nV, int nT, const double* V, const int* T, const std::vector<int>& SF);
void findSurfaceTris(int nT, const int* T, std::vector<int>& SF);

} // namespace ipcgpu
