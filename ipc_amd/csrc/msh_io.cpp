// Tetrahedral mesh files: the host-side readers the reference reaches through IglUtils::readTetMesh
// (src/Utils/IglUtils.cpp:451-584): Gmsh MSH 4.1 and 2.2 (what its MshIO dependency parses; ASCII here) and the reference's own
// "msh 4.0" dialect ($Nodes "1 N" / $Elements "1 N" / optional $Surface, IglUtils.cpp:514-584), plus saveTetMesh
// (IglUtils.cpp:300-361).  Nodes are taken in file order and elements refer to them by tag - 1, exactly as the reference does.
// No GPU involved: these are what lets both implementations start from the same paper-scene geometry.
#include "common.h"
#include "msh_io.h"
#include <algorithm>
#include <array>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <limits>
#include <map>
#include <sstream>

namespace ipcgpu {

namespace {

bool nextSection(std::istream& in, const char* name)
{
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line == name) return true;
    }
    return false;
}

void requireGood(std::istream& in, const std::string& what)
{
    if (!in.good() && !in.eof()) throw StateError("msh: malformed " + what);
    if (in.fail()) throw StateError("msh: malformed " + what);
}

// $Nodes / $Elements of MSH 4.1 (entity blocks)
void read41(std::istream& in, TetMeshFile& m)
{
    if (!nextSection(in, "$Nodes")) throw StateError("msh: no $Nodes section");
    size_t nBlocks = 0, nNodes = 0, minTag = 0, maxTag = 0;
    in >> nBlocks >> nNodes >> minTag >> maxTag;
    requireGood(in, "$Nodes header");
    m.V.reserve(3 * nNodes);
    for (size_t b = 0; b < nBlocks; ++b) {
        int dim = 0, tag = 0, parametric = 0;
        size_t n = 0;
        in >> dim >> tag >> parametric >> n;
        requireGood(in, "node block header");
        if (parametric) throw StateError("msh: parametric node blocks are not supported");
        size_t t;
        for (size_t i = 0; i < n; ++i) in >> t; // tags: nodes are used in file order (IglUtils.cpp:481-487)
        for (size_t i = 0; i < n; ++i) {
            double x, y, z;
            in >> x >> y >> z;
            m.V.push_back(x);
            m.V.push_back(y);
            m.V.push_back(z);
        }
        requireGood(in, "node block");
    }
    if (m.V.size() != 3 * nNodes) throw StateError("msh: node count does not match the $Nodes header");
    if (!nextSection(in, "$Elements")) throw StateError("msh: no $Elements section");
    size_t nEl = 0;
    in >> nBlocks >> nEl >> minTag >> maxTag;
    requireGood(in, "$Elements header");
    for (size_t b = 0; b < nBlocks; ++b) {
        int dim = 0, tag = 0, type = 0;
        size_t n = 0;
        in >> dim >> tag >> type >> n;
        requireGood(in, "element block header");
        if (dim != 3 || type != 4) throw StateError("msh: only linear tetrahedra (element type 4) are supported, as in the reference (IglUtils.cpp:473-476)");
        for (size_t i = 0; i < n; ++i) {
            size_t et;
            long long a[4];
            in >> et >> a[0] >> a[1] >> a[2] >> a[3];
            for (int k = 0; k < 4; ++k) m.T.push_back((int)(a[k] - 1));
        }
        requireGood(in, "element block");
    }
}

// $Nodes / $Elements of MSH 2.2
void read22(std::istream& in, TetMeshFile& m)
{
    if (!nextSection(in, "$Nodes")) throw StateError("msh: no $Nodes section");
    size_t nNodes = 0;
    in >> nNodes;
    requireGood(in, "$Nodes header");
    m.V.reserve(3 * nNodes);
    for (size_t i = 0; i < nNodes; ++i) {
        size_t t;
        double x, y, z;
        in >> t >> x >> y >> z;
        m.V.push_back(x);
        m.V.push_back(y);
        m.V.push_back(z);
    }
    requireGood(in, "$Nodes");
    if (!nextSection(in, "$Elements")) throw StateError("msh: no $Elements section");
    size_t nEl = 0;
    in >> nEl;
    requireGood(in, "$Elements header");
    static const int nodesOfType[16] = { 0, 2, 3, 4, 4, 8, 6, 5, 3, 6, 9, 10, 27, 18, 14, 1 };
    for (size_t i = 0; i < nEl; ++i) {
        size_t id;
        int type = 0, nTags = 0;
        in >> id >> type >> nTags;
        requireGood(in, "element");
        for (int k = 0; k < nTags; ++k) {
            long long t;
            in >> t;
        }
        if (type < 1 || type > 15) throw StateError("msh: unsupported element type");
        const int nn = nodesOfType[type];
        long long a[27];
        for (int k = 0; k < nn; ++k) in >> a[k];
        if (type == 4)
            for (int k = 0; k < 4; ++k) m.T.push_back((int)(a[k] - 1));
        else if (type != 15 && type != 1 && type != 2) // lower-dimensional entities are skipped, other volume elements are not tets
            throw StateError("msh: only linear tetrahedra (element type 4) are supported");
    }
    requireGood(in, "$Elements");
}

// the reference's own dialect (IglUtils.cpp:514-584)
void read40(std::istream& in, TetMeshFile& m)
{
    if (!nextSection(in, "$Nodes")) throw StateError("msh: no $Nodes section");
    long long one = 0, n = 0;
    in >> one >> n;
    requireGood(in, "$Nodes header");
    std::string rest;
    std::getline(in, rest);
    std::getline(in, rest); // block header line (skipped by the reference as well)
    for (long long i = 0; i < n; ++i) {
        long long t;
        double x, y, z;
        in >> t >> x >> y >> z;
        m.V.push_back(x);
        m.V.push_back(y);
        m.V.push_back(z);
    }
    requireGood(in, "$Nodes");
    if (!nextSection(in, "$Elements")) throw StateError("msh: no $Elements section");
    in >> one >> n;
    requireGood(in, "$Elements header");
    std::getline(in, rest);
    std::getline(in, rest);
    for (long long i = 0; i < n; ++i) {
        long long t, a[4];
        in >> t >> a[0] >> a[1] >> a[2] >> a[3];
        for (int k = 0; k < 4; ++k) m.T.push_back((int)(a[k] - 1));
    }
    requireGood(in, "$Elements");
    if (nextSection(in, "$Surface")) {
        in >> n;
        requireGood(in, "$Surface header");
        for (long long i = 0; i < n; ++i) {
            long long a[3];
            in >> a[0] >> a[1] >> a[2];
            for (int k = 0; k < 3; ++k) m.SF.push_back((int)(a[k] - 1));
        }
        requireGood(in, "$Surface");
    }
}

} // namespace

// IglUtils::findSurfaceTris (IglUtils.cpp:203-233): the oriented faces (0,2,1), (0,3,2), (0,1,3), (1,2,3) of every tet whose
// reverse does not occur.  The reference walks a std::unordered_map, so its face ORDER is whatever its standard library's
// hash gives; here the order is (tet, local face), which is also what ipc_amd/scene.py and the oracle use.
void findSurfaceTris(int nT, const int* T /*4 per tet, interleaved*/, std::vector<int>& SF)
{
    static const int loc[4][3] = { { 1, 2, 3 }, { 0, 3, 2 }, { 0, 1, 3 }, { 0, 2, 1 } };
    std::map<std::array<int, 3>, int> count;
    auto key = [](int a, int b, int c) {
        std::array<int, 3> k = { a, b, c };
        std::sort(k.begin(), k.end());
        return k;
    };
    for (int t = 0; t < nT; ++t)
        for (int f = 0; f < 4; ++f) count[key(T[4 * t + loc[f][0]], T[4 * t + loc[f][1]], T[4 * t + loc[f][2]])]++;
    SF.clear();
    for (int t = 0; t < nT; ++t)
        for (int f = 0; f < 4; ++f) {
            const int a = T[4 * t + loc[f][0]], b = T[4 * t + loc[f][1]], c = T[4 * t + loc[f][2]];
            if (count[key(a, b, c)] == 1) {
                SF.push_back(a);
                SF.push_back(b);
                SF.push_back(c);
            }
        }
}

void readTetMesh(const std::string& path, TetMeshFile& m, bool findSurface)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) throw StateError("msh: cannot open " + path);
    m = TetMeshFile();
    std::string line;
    std::getline(in, line);
    if (!line.empty() && line.back() == '\r') line.pop_back();
    bool handled = false;
    if (line == "$MeshFormat") {
        double version = 0;
        int fileType = 0, dataSize = 0;
        in >> version >> fileType >> dataSize;
        requireGood(in, "$MeshFormat");
        if (fileType != 0) throw StateError("msh: binary files are not supported (every mesh shipped with the reference is ASCII)");
        if (version >= 4.05 && version < 4.15) {
            read41(in, m);
            handled = true;
        }
        else if (version >= 2.0 && version < 3.0) {
            read22(in, m);
            handled = true;
        }
    }
    if (!handled) { // MshIO throws on anything else and the reference falls back to its own reader (IglUtils.cpp:462-466)
        in.clear();
        in.seekg(0);
        read40(in, m);
    }
    const int nV = (int)(m.V.size() / 3);
    if (nV < 4 || m.T.empty()) throw StateError("msh: no tetrahedra in " + path);
    for (int v : m.T)
        if (v < 0 || v >= nV) throw StateError("msh: element refers to a node that does not exist");
    for (int v : m.SF)
        if (v < 0 || v >= nV) throw StateError("msh: surface triangle refers to a node that does not exist");
    if (m.SF.empty() && findSurface) findSurfaceTris((int)(m.T.size() / 4), m.T.data(), m.SF);
}

// IglUtils::saveTetMesh (IglUtils.cpp:300-361): MSH 4.1 ASCII, one node block, one element block, $Surface appended
void saveTetMesh(const std::string& path, int nV, int nT, const double* V /*xyz interleaved*/, const int* T /*4 per tet*/,
    const std::vector<int>& SF)
{
    std::ofstream out(path, std::ios::out);
    if (!out.is_open()) throw StateError("msh: unable to save mesh to " + path);
    out << std::setprecision(std::numeric_limits<double>::max_digits10);
    out << "$MeshFormat\n4.1 0 8\n$EndMeshFormat\n";
    out << "$Nodes\n1 " << nV << " 1 " << nV << "\n3 0 0 " << nV << "\n";
    for (int v = 0; v < nV; ++v) out << v + 1 << "\n";
    for (int v = 0; v < nV; ++v) out << V[3 * (size_t)v] << " " << V[3 * (size_t)v + 1] << " " << V[3 * (size_t)v + 2] << "\n";
    out << "$EndNodes\n";
    out << "$Elements\n1 " << nT << " 1 " << nT << "\n3 0 4 " << nT << "\n";
    for (int t = 0; t < nT; ++t)
        out << t + 1 << " " << T[4 * (size_t)t] + 1 << " " << T[4 * (size_t)t + 1] + 1 << " " << T[4 * (size_t)t + 2] + 1 << " " << T[4 * (size_t)t + 3] + 1
            << "\n";
    out << "$EndElements\n";
    out << "$Surface\n" << SF.size() / 3 << "\n";
    for (size_t i = 0; i + 2 < SF.size(); i += 3) out << SF[i] + 1 << " " << SF[i + 1] + 1 << " " << SF[i + 2] + 1 << "\n";
    out << "$EndSurface\n";
    if (!out.good()) throw StateError("msh: write error on " + path);
}

} // namespace ipcgpu
