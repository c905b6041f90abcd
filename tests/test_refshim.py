"""Consistency of the dense-matrix stand-in (oracle/refshim/mini_eigen.hpp) that lets the reference's own sources compile here.
Test infrastructure checking test infrastructure: the pieces of it that are algorithms rather than spelling (LDLT, full-pivot LU,
symmetric eigen-decomposition, inverse / determinant, blocks and maps) against residuals and identities, compiled on the fly."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(os.path.dirname(HERE), "oracle", "refshim")

SRC = r"""
#include <Eigen/Eigen>
#include <iostream>
using namespace Eigen;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::cerr << "FAILED: " #c " (line " << __LINE__ << ")\n"; ++fails; } } while (0)
int main()
{
    std::srand(7);
    double wLU = 0, wLDLT = 0, wInv = 0, wEig = 0;
    for (int it = 0; it < 500; ++it) {
        Matrix3d A = Matrix3d::Random();
        Vector3d b = Vector3d::Random();
        if (it % 3 == 0) A.row(it % 3) *= 1e-3;
        if (it % 5 == 0) A.col(it % 3) *= 1e2;
        Vector3d x = A.fullPivLu().solve(b);
        wLU = std::max(wLU, (A * x - b).norm() / (b.norm() + (A.cwiseAbs() * x.cwiseAbs()).norm()));
        Matrix3d S = A.transpose() * A;
        Vector3d y = S.ldlt().solve(b);
        wLDLT = std::max(wLDLT, (S * y - b).norm() / (b.norm() + (S.cwiseAbs() * y.cwiseAbs()).norm()));
        Matrix3d I = A.inverse() * A;
        wInv = std::max(wInv, (I - Matrix3d::Identity()).norm() / (A.norm() * A.inverse().norm()));
        Matrix<double, 6, 6> M = Matrix<double, 6, 6>::Random();
        M = (M + M.transpose()).eval();
        SelfAdjointEigenSolver<Matrix<double, 6, 6>> es(M);
        Matrix<double, 6, 6> R = es.eigenvectors() * es.eigenvalues().asDiagonal() * es.eigenvectors().transpose();
        wEig = std::max(wEig, (R - M).norm() / M.norm());
        for (int k = 0; k + 1 < 6; ++k) CHECK(es.eigenvalues()[k] <= es.eigenvalues()[k + 1]);
    }
    CHECK(wLU < 1e-13);
    CHECK(wLDLT < 1e-13);
    CHECK(wInv < 1e-13);
    CHECK(wEig < 1e-13);
    // a system with a zero pivot column order that needs the column permutation (the case that exposed a bug in the stand-in)
    Matrix3d A;
    A << 9.2e-17, -1.1e-16, -1.0, 1, 0, 0, -2.2e-16, -1, -1;
    Vector3d b(-2.7e-16, 1, -1), x = A.fullPivLu().solve(b);
    CHECK((A * x - b).norm() < 1e-14);
    // rank-deficient: free variable zero, consistent part solved
    Matrix3d Rk;
    Rk << 1, 2, 3, 2, 4, 6, 0, 1, 1;
    Vector3d rb(6, 12, 2), rx = Rk.fullPivLu().solve(rb);
    CHECK(Rk.fullPivLu().rank() == 2);
    CHECK((Rk * rx - rb).norm() < 1e-12);
    // blocks, maps, comma initialiser, vector transposition on assignment, row-major storage
    MatrixXd V(4, 3);
    V << 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3;
    Matrix3d E;
    for (int k = 0; k < 3; ++k) E.col(k) = (V.row(k + 1) - V.row(0)).transpose();
    CHECK(E.determinant() == 6.0);
    Matrix<double, 12, 1> g;
    g.setZero();
    g.segment<3>(3) = 2.0 * (V.row(1) - V.row(0));
    CHECK(g[3] == 2.0 && g[4] == 0.0);
    typedef Matrix<double, Dynamic, Dynamic, RowMajor> RowM;
    RowM Rm(V);
    VectorXd flat = Map<VectorXd>(Rm.data(), 12);
    CHECK(flat[3] == 1.0 && flat[7] == 2.0 && flat[11] == 3.0);
    MatrixXd W = V;
    W = Map<VectorXd>(W.data(), W.size());
    CHECK(W.rows() == 12 && W.cols() == 1 && W(4, 0) == 0.0 && W(6, 0) == 2.0);
    Array<double, 1, 3> a = V.row(3).array().max(V.row(1).array());
    CHECK(a[0] == 1.0 && a[2] == 3.0);
    CHECK((V.colwise().maxCoeff() - RowVector3d(1, 2, 3)).norm() == 0.0);
    Matrix3d Q = (AngleAxisd(0.3, Vector3d::UnitX()) * AngleAxisd(-0.7, Vector3d::UnitY()) * AngleAxisd(1.1, Vector3d::UnitZ())).toRotationMatrix();
    Matrix3d Qm = AngleAxisd(0.3, Vector3d::UnitX()).toRotationMatrix() * AngleAxisd(-0.7, Vector3d::UnitY()).toRotationMatrix() * AngleAxisd(1.1, Vector3d::UnitZ()).toRotationMatrix();
    CHECK((Q - Qm).norm() < 1e-15 && std::abs(Q.determinant() - 1.0) < 1e-15);
    std::cout << "residuals LU " << wLU << " LDLT " << wLDLT << " inverse " << wInv << " eigen " << wEig << "\n";
    return fails;
}
"""


def test_the_dense_matrix_stand_in_is_consistent(tmp_path):
    src = tmp_path / "shim_check.cpp"
    src.write_text(SRC)
    exe = tmp_path / "shim_check"
    subprocess.run(["g++", "-std=c++17", "-O1", f"-I{SHIM}", str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stderr
