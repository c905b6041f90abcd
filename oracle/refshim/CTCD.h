// ORACLE / TEST INFRASTRUCTURE: the four entry points of Vouga's CTCD (inside the un-vendored CCD-Wrapper) that the
// reference calls.  The implementation linked into oracle/_ref is NOT CTCD: ref_ctcd_plug.cpp forwards to the
// per-pair conservative advancement of oracle/orc_contact.cpp (`accd`), i.e. to the same "CCD by contract" the
// oracle and the HIP kernels use.  What the reference-compiled call sites therefore pin is everything AROUND the
// per-pair query (candidate enumeration through the real SpatialHash, the eta = (1 - slackness) * distance rule,
// the t < 1e-6 retry with eta = 0, the slackness rescale, the minimum) -- not the time of impact itself.
#pragma once
#include <Eigen/Core>
class CTCD {
public:
    static bool vertexFaceCTCD(const Eigen::Vector3d& q0start, const Eigen::Vector3d& q1start, const Eigen::Vector3d& q2start, const Eigen::Vector3d& q3start,
        const Eigen::Vector3d& q0end, const Eigen::Vector3d& q1end, const Eigen::Vector3d& q2end, const Eigen::Vector3d& q3end, double eta, double& t);
    static bool edgeEdgeCTCD(const Eigen::Vector3d& q0start, const Eigen::Vector3d& p0start, const Eigen::Vector3d& q1start, const Eigen::Vector3d& p1start,
        const Eigen::Vector3d& q0end, const Eigen::Vector3d& p0end, const Eigen::Vector3d& q1end, const Eigen::Vector3d& p1end, double eta, double& t);
    static bool vertexEdgeCTCD(const Eigen::Vector3d& q0start, const Eigen::Vector3d& q1start, const Eigen::Vector3d& q2start,
        const Eigen::Vector3d& q0end, const Eigen::Vector3d& q1end, const Eigen::Vector3d& q2end, double eta, double& t);
    static bool vertexVertexCTCD(const Eigen::Vector3d& q1start, const Eigen::Vector3d& q2start,
        const Eigen::Vector3d& q1end, const Eigen::Vector3d& q2end, double eta, double& t);
};
