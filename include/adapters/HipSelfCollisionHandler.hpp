// HipSelfCollisionHandler -- the collision-handler side of the drop-in (INTEGRATION.md section 3b as compiled code).
//
// `SelfCollisionHandler<dim>` (src/CollisionObject/SelfCollisionHandler.hpp:21-250) has only static members, called by name from 44 places of
// src/TimeStepper/Optimizer.cpp: there is no virtual to override.  This class carries statics of the SAME names and signatures that forward to
// the C ABI (the constraint sets, the per-constraint distances and Jacobian-transpose products, the barrier Hessian, the two step bounds and the
// intersection test run on the device); everything it does not define is inherited from the reference's class, so the rest of the handler --
// the QP / SQP entry points, friction, the mollified-pair terms -- stays the reference's host code.  A maintainer makes Optimizer.cpp call it by
// including this header AND HipSelfCollisionHandlerRedirect.hpp (the one-line `#define` that redirects the name; a header of its own without an include
// guard, so that it acts wherever it is included -- whatever was included before) behind the other includes of Optimizer.cpp, exactly as
// tests/adapters/main_hook.hpp redirects `new Optimizer` -- the test build pre-includes both (g++ -include) in front of the UNCHANGED Optimizer.cpp.
//
// Who decides: hipCollisionRegistry().ctx.  Null (the default) = every call goes to the reference's implementation.  HipOptimizer sets it in
// percall mode for scenes with `selfCollisionOn` (IPCGPU_PERCALL_CONTACT=host keeps the host code for A/B runs) after handing the surface to the
// library, and every forward is counted in hipCollisionRegistry().calls so that a run can say where its contact was evaluated.
//
// Data contract: MMCVID is four ints (MeshCollisionUtils.hpp:24-27), so std::vector<MMCVID>::data() IS the int4 buffer of the C ABI; positions go
// over as Mesh<3>::V (Eigen column-major nV x 3 = the layout of ipcgpu_set_positions) before every call -- the reference's control flow moves the
// mesh between calls on the host.
// Needs: the reference's SelfCollisionHandler.hpp, include/adapters/HipLinSysSolver.hpp, include/ipcgpu.h, -lipcgpu.
#pragma once
#define IPCGPU_HIP_SELF_COLLISION_HANDLER_DECLARED
#include "SelfCollisionHandler.hpp"
#include "HipLinSysSolver.hpp"
#include <ipcgpu.h>
#include <stdexcept>
#include <vector>

namespace IPC {

struct HipCollisionRegistry {
    ipcgpu_ctx* ctx = nullptr; // the context whose mesh and surface are the ones of the Mesh<3> the handler is called with
    // forwards so far: 0 computeConstraintSet, 1 evaluateConstraints, 2 leftMultiplyConstraintJacobianT, 3 augmentIPHessian,
    // 4 largestFeasibleStepSize, 5 largestFeasibleStepSize_CCD, 6 checkEdgeTriIntersectionIfAny
    long long calls[7] = { 0, 0, 0, 0, 0, 0, 0 };
    long long hostFallbacks = 0; // calls that reached this class with a context set and still had to take the host code (a solver that is not HipLinSysSolver)
};
inline HipCollisionRegistry& hipCollisionRegistry()
{
    static HipCollisionRegistry r;
    return r;
}

template <int dim>
class HipSelfCollisionHandler : public SelfCollisionHandler<dim> {
    static_assert(dim == 3, "the device path is three-dimensional");
    static_assert(sizeof(MMCVID) == 4 * sizeof(int), "MMCVID is expected to be four ints");
    typedef SelfCollisionHandler<dim> Ref;
    static ipcgpu_ctx* dev() { return hipCollisionRegistry().ctx; }
    static void chk(int rc)
    {
        if (rc < 0) throw std::runtime_error(ipcgpu_last_error());
    }
    static void sync(const Mesh<dim>& mesh) { chk(ipcgpu_set_positions(dev(), mesh.V.data())); }
    static const int* tuples(const std::vector<MMCVID>& set) { return set.empty() ? nullptr : reinterpret_cast<const int*>(set.data()); }

public:
    // SelfCollisionHandler.cpp:2149-2478: the MMCVID tuples within dHat (merged PP / PE duplicates carry their multiplicity), the mollified
    // edge-edge pairs and, with getPTEE, the point-triangle / edge-edge candidate list of the partial CCD.  The reference's hash `sh` is not used:
    // the library's own grid finds the same sets (tests/test_gpu_vs_reference.py::test_constraint_sets_against_the_reference).
    static void computeConstraintSet(const Mesh<dim>& mesh, const SpatialHash<dim>& sh, double dHat, std::vector<MMCVID>& constraintSet,
        std::vector<MMCVID>& paraEEMMCVIDSet, std::vector<std::pair<int, int>>& paraEEeIeJSet, bool getPTEE, std::vector<std::pair<int, int>>& cs_PTEE)
    {
        if (!dev()) return Ref::computeConstraintSet(mesh, sh, dHat, constraintSet, paraEEMMCVIDSet, paraEEeIeJSet, getPTEE, cs_PTEE);
        hipCollisionRegistry().calls[0]++;
        sync(mesh);
        int counts[3] = { 0, 0, 0 };
        chk(ipcgpu_contact_build(dev(), dHat, counts));
        constraintSet.assign((size_t)counts[0], MMCVID());
        paraEEMMCVIDSet.assign((size_t)counts[1], MMCVID());
        std::vector<int> eiej(2 * (size_t)counts[1] + 2), cs(2 * (size_t)counts[2] + 2);
        chk(ipcgpu_contact_get(dev(), counts[0] ? reinterpret_cast<int*>(constraintSet.data()) : nullptr,
            counts[1] ? reinterpret_cast<int*>(paraEEMMCVIDSet.data()) : nullptr, eiej.data(), getPTEE ? cs.data() : nullptr));
        paraEEeIeJSet.resize((size_t)counts[1]);
        for (int i = 0; i < counts[1]; ++i) paraEEeIeJSet[i] = std::make_pair(eiej[2 * (size_t)i], eiej[2 * (size_t)i + 1]);
        if (getPTEE) {
            cs_PTEE.resize((size_t)counts[2]);
            for (int i = 0; i < counts[2]; ++i) cs_PTEE[i] = std::make_pair(cs[2 * (size_t)i], cs[2 * (size_t)i + 1]);
        }
    }

    // :64-81: val grows by one squared distance per tuple
    static void evaluateConstraints(const Mesh<dim>& mesh, const std::vector<MMCVID>& activeSet, Eigen::VectorXd& val, double coef = 1.0)
    {
        if (!dev()) return Ref::evaluateConstraints(mesh, activeSet, val, coef);
        hipCollisionRegistry().calls[1]++;
        const int start = (int)val.size();
        val.conservativeResize(start + (int)activeSet.size());
        if (activeSet.empty()) return;
        sync(mesh);
        chk(ipcgpu_contact_evaluate(dev(), (int)activeSet.size(), tuples(activeSet), val.data() + start));
    }

    // :84-148: output += coef * multiplicity_i * input[i] * grad d_i
    static void leftMultiplyConstraintJacobianT(const Mesh<dim>& mesh, const std::vector<MMCVID>& activeSet, const Eigen::VectorXd& input,
        Eigen::VectorXd& output_incremental, double coef = 1.0)
    {
        if (!dev()) return Ref::leftMultiplyConstraintJacobianT(mesh, activeSet, input, output_incremental, coef);
        hipCollisionRegistry().calls[2]++;
        if (activeSet.empty()) return;
        if (input.size() < (long)activeSet.size() || output_incremental.size() != 3 * mesh.V.rows())
            throw std::invalid_argument("HipSelfCollisionHandler::leftMultiplyConstraintJacobianT: vector sizes do not match the set / the mesh");
        sync(mesh);
        chk(ipcgpu_contact_jt_multiply(dev(), (int)activeSet.size(), tuples(activeSet), input.data(), coef, output_incremental.data()));
    }

    // :418-561: the PSD-projected barrier Hessian blocks of the set, added to the matrix in HBM.  Only when the matrix IS in HBM (a HipLinSysSolver on
    // this context); any other solver gets the reference's blocks through addCoeff.
    static void augmentIPHessian(const Mesh<dim>& mesh, const std::vector<MMCVID>& activeSet, LinSysSolver<Eigen::VectorXi, Eigen::VectorXd>* mtr_incremental,
        double dHat, double coef = 1.0, bool projectDBC = true)
    {
        auto* hip = dev() ? dynamic_cast<HipLinSysSolver<Eigen::VectorXi, Eigen::VectorXd>*>(mtr_incremental) : nullptr;
        if (!hip || hip->context() != dev()) {
            if (dev()) hipCollisionRegistry().hostFallbacks++;
            return Ref::augmentIPHessian(mesh, activeSet, mtr_incremental, dHat, coef, projectDBC);
        }
        hipCollisionRegistry().calls[3]++;
        if (activeSet.empty()) return;
        sync(mesh);
        hip->flush(); // host-side addCoeff / setCoeff that are still pending go first: the device adds on top of them
        chk(ipcgpu_contact_set(dev(), (int)activeSet.size(), tuples(activeSet), 0, nullptr, nullptr)); // this set alone; the mollified pairs have their own call
        chk(ipcgpu_contact_hessian_add(dev(), dHat, coef, projectDBC ? 1 : 0));
    }

    // :564-686: the step bound over the candidate pairs of the current constraint sets (the list the library kept at computeConstraintSet)
    static void largestFeasibleStepSize(const Mesh<dim>& mesh, const SpatialHash<dim>& sh, const Eigen::VectorXd& searchDir, double slackness,
        const std::vector<std::pair<int, int>>& constraintSet, std::vector<std::pair<int, int>>& candidates, double& stepSize)
    {
        if (!dev()) return Ref::largestFeasibleStepSize(mesh, sh, searchDir, slackness, constraintSet, candidates, stepSize);
        // The library bounds the step over the candidate list IT kept at the last computeConstraintSet (getPTEE): that is the reference's call only when
        // Optimizer.cpp passes MMActiveSet_CCD of that same build, i.e. in the CFL_FOR_CCD = 1 / 2 builds (Types.hpp:34; Optimizer.cpp:1891-1935).  With
        // CFL_FOR_CCD == 0 the list is empty by construction and the reference's function of this name IS the full sweep: refuse to compile rather than
        // hand back a partial bound.
#if defined(CFL_FOR_CCD)
        static_assert(CFL_FOR_CCD != 0, "HipSelfCollisionHandler::largestFeasibleStepSize forwards the PARTIAL CCD (CFL_FOR_CCD 1 or 2); a CFL_FOR_CCD == 0 build must forward to ipcgpu_ccd_full_reference");
#endif
        {
            int counts[3] = { 0, 0, 0 };
            chk(ipcgpu_contact_counts(dev(), counts));
            if ((size_t)counts[2] != constraintSet.size()) throw std::runtime_error("HipSelfCollisionHandler::largestFeasibleStepSize: the candidate list handed in is not the one of the library's last computeConstraintSet (stale state)");
        }
        hipCollisionRegistry().calls[4]++;
        sync(mesh);
        int pair[2] = { -1, -1 };
        chk(ipcgpu_ccd_partial(dev(), searchDir.data(), slackness, &stepSize, pair));
    }

    // :982-1366: the sweep over every vertex-vertex / vertex-edge / vertex-triangle / edge-edge pair that shares a cell of the swept hash.  The
    // caller has already let SpatialHash::build cap the step (Optimizer.cpp:1965: its `double&` argument); the library applies the same rule to the
    // step it is handed, which then no longer shrinks.
    static void largestFeasibleStepSize_CCD(const Mesh<dim>& mesh, const SpatialHash<dim>& sh, const Eigen::VectorXd& searchDir, double slackness,
        std::vector<std::pair<int, int>>& candidates, double& stepSize)
    {
        if (!dev()) return Ref::largestFeasibleStepSize_CCD(mesh, sh, searchDir, slackness, candidates, stepSize);
        hipCollisionRegistry().calls[5]++;
        sync(mesh);
        double capped = stepSize;
        int arg[3] = { -1, -1, -1 }, nCand = 0;
        chk(ipcgpu_ccd_full_reference(dev(), searchDir.data(), slackness, &stepSize, &capped, arg, &nCand));
    }

    // :3255-3340 (through Optimizer::isIntersected, Optimizer.cpp:2626-2659): false when an edge pierces a triangle
    static bool checkEdgeTriIntersectionIfAny(const Mesh<dim>& mesh, const SpatialHash<dim>& sh)
    {
        if (!dev()) return Ref::checkEdgeTriIntersectionIfAny(mesh, sh);
        hipCollisionRegistry().calls[6]++;
        sync(mesh);
        int flag = 0;
        chk(ipcgpu_is_intersected(dev(), &flag));
        return flag == 0;
    }
};

} // namespace IPC
