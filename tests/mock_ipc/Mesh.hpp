// TEST INFRASTRUCTURE -- stand-in for the data members of the reference's src/Mesh.hpp that the adapters read
// (Mesh.hpp:58-171).  Same names, same types (through the Eigen stand-in), no behaviour beyond isProjectDBCVertex.
#pragma once
#include "Types.hpp"
#include <Eigen/Eigen>
#include <set>
#include <vector>

namespace IPC {

enum class DirichletBCType { // Mesh.hpp:40-44
    NOT_DBC = 0,
    ZERO = 1,
    NONZERO = 2
};

template <int dim>
class Mesh {
public: // owned data (Mesh.hpp:60-66)
    Eigen::MatrixXd V_rest, V;
    Eigen::MatrixXi F, SF;

public: // owned features (Mesh.hpp:147-164)
    Eigen::SparseMatrix<double> massMatrix;
    double density = 0, m_YM = 0, m_PR = 0;
    Eigen::VectorXd u, lambda;
    Eigen::VectorXd triArea;
    std::set<int> DBCVertexIds;
    std::vector<DirichletBCType> vertexDBCType;
    std::vector<Eigen::Matrix<double, dim, dim>> restTriInv;
    std::vector<std::set<int>> vNeighbor;

    bool isDBCVertex(int vI) const { return vertexDBCType[vI] != DirichletBCType::NOT_DBC; } // :219
    bool isProjectDBCVertex(int vI, bool projectDBC) const // :220-229
    {
        return vertexDBCType[vI] == DirichletBCType::ZERO || (vertexDBCType[vI] == DirichletBCType::NONZERO && projectDBC);
    }
};

} // namespace IPC
