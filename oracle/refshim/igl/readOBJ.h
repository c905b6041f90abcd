// ORACLE / TEST INFRASTRUCTURE: libigl's readOBJ(path, V, F) restated for what the reference feeds it (triangle meshes): `v x y z`
// lines, `f a b c` lines whose entries may carry /vt/vn suffixes and may be negative (relative to the vertices read so far);
// indices become 0-based.  Faces with more than three corners are split into a fan, as the matrices must be rectangular.
#pragma once
#include <Eigen/Core>
#include <array>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
namespace igl {
template <class DV, class DF>
inline bool readOBJ(const std::string& path, Eigen::MatrixBase<DV>& V, Eigen::MatrixBase<DF>& F)
{
    std::ifstream in(path);
    if (!in) return false;
    std::vector<std::array<double, 3>> vs;
    std::vector<std::array<int, 3>> fs;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") {
            std::array<double, 3> p = { 0, 0, 0 };
            ss >> p[0] >> p[1] >> p[2];
            vs.push_back(p);
        }
        else if (tag == "f") {
            std::vector<int> idx;
            std::string tok;
            while (ss >> tok) {
                const int i = std::atoi(tok.substr(0, tok.find('/')).c_str());
                idx.push_back(i > 0 ? i - 1 : (int)vs.size() + i);
            }
            for (size_t k = 1; k + 1 < idx.size(); ++k) fs.push_back({ idx[0], idx[k], idx[k + 1] });
        }
    }
    V.derived().resize((Eigen::Index)vs.size(), 3);
    for (size_t i = 0; i < vs.size(); ++i)
        for (int c = 0; c < 3; ++c) V((Eigen::Index)i, c) = vs[i][c];
    F.derived().resize((Eigen::Index)fs.size(), 3);
    for (size_t i = 0; i < fs.size(); ++i)
        for (int c = 0; c < 3; ++c) F((Eigen::Index)i, c) = fs[i][c];
    return true;
}
} // namespace igl
