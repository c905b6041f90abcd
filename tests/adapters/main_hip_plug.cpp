// TEST INFRASTRUCTURE -- what a maintainer adds to src/LinSysSolver/LinSysSolver.cpp:13-27 (`case LinSysSolverType::HIP:
// return hipCreateLinSysSolver<...>();`), here as the whole factory of the executable that runs the reference's main() on the
// HIP adapters (tests/test_adapters.py::build_main_hip).  The un-vendored CTCD comes from oracle/ref_plug.cpp as in libipcref.so:
// it is only reached in percall mode, where the reference's host code sweeps the contact pairs.
#include "HipOptimizer.hpp"

namespace IPC {
template <typename vectorTypeI, typename vectorTypeS>
LinSysSolver<vectorTypeI, vectorTypeS>* LinSysSolver<vectorTypeI, vectorTypeS>::create(const LinSysSolverType)
{
    return hipCreateLinSysSolver<vectorTypeI, vectorTypeS>();
}
template class LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>;
} // namespace IPC

int ipc_reference_main(int argc, char* argv[]); // src/main.cpp compiled with -Dmain=ipc_reference_main
int main(int argc, char** argv) { return ipc_reference_main(argc, argv); }
