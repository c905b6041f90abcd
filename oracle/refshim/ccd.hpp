// ORACLE / TEST INFRASTRUCTURE: declarations of the CCD-Wrapper front end (un-vendored dependency, pinned at
// 23907da in cmake/recipes/ccd_wrapper.cmake) with the shape the reference's call sites need, so that those call
// sites compile.  The default configuration (CCDMethod FLOATING_POINT_ROOT_FINDER) never goes through ccd::...CCD:
// it calls CTCD directly (see CTCD.h here).  The bodies abort.
#pragma once
#include <Eigen/Core>
#include <array>
#include <cstdlib>
#include <iostream>
namespace ccd {
enum CCDMethod {
    FLOAT = 0,
    MULTIPRECISION_FLOAT,
    RATIONAL,
    BSC,
    TIGHT_CCD,
    ROOT_PARITY,
    RATIONAL_ROOT_PARITY,
    FIXED_ROOT_PARITY,
    RATIONAL_FIXED_ROOT_PARITY,
    MIN_SEPARATION_ROOT_PARITY,
    MIN_SEPARATION_ROOT_FINDER,
    FLOATING_POINT_ROOT_FINDER,
    FLOATING_POINT_ROOT_PARITY,
    UNIVARIATE_INTERVAL_ROOT_FINDER,
    MULTIVARIATE_INTERVAL_ROOT_FINDER,
    REDON_ROOT_FINDER,
    TIGHT_INCLUSION,
    NUM_CCD_METHODS
};
static const char* const method_names[NUM_CCD_METHODS] = { "Float", "MultiprecisionFloat", "Rational", "BSC", "TightCCD",
    "RootParity", "RationalRootParity", "FixedRootParity", "RationalFixedRootParity", "MinSeparationRootParity",
    "MinSeparationRootFinder", "FloatingPointRootFinder", "FloatingPointRootParity", "UnivariateIntervalRootFinder", "MultivariateIntervalRootFinder",
    "RedonRootFinder", "TightInclusion" };
inline bool is_time_of_impact_computed(CCDMethod m) { return m == FLOATING_POINT_ROOT_FINDER || m == TIGHT_INCLUSION; }
inline bool not_provided(const char* what)
{
    std::cerr << "refshim: ccd::" << what << " is not provided (only the CTCD path of the default configuration is)" << std::endl;
    std::abort();
    return false;
}
inline bool vertexFaceCCD(const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&,
    const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&,
    const CCDMethod, const double = 1e-6, const long = 1e7, const std::array<double, 3>& = { { -1, -1, -1 } })
{
    return not_provided("vertexFaceCCD");
}
inline bool edgeEdgeCCD(const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&,
    const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&, const Eigen::Vector3d&,
    const CCDMethod, const double = 1e-6, const long = 1e7, const std::array<double, 3>& = { { -1, -1, -1 } })
{
    return not_provided("edgeEdgeCCD");
}
} // namespace ccd
