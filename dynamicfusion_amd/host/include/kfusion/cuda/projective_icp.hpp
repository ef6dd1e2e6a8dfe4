// kfusion/cuda/projective_icp.hpp -- point-to-plane projective ICP, API-compatible with the class of the same name in the reference
// (/root/reference/kfusion/include/kfusion/cuda/projective_icp.hpp:9-48): same type names, setters / getters and the two
// estimateTransform overloads KinFu calls.  What differs is underneath: the correspondence search, the 27-sum reduction, the 6x6
// solve and the pose update all run on the GPU as one enqueue (dfusion_icp_estimate); setDeviceLoop(false) restores the reference's
// control flow (stream synchronise + host solve per iteration).  The Frame overload, which the reference declares but never
// implements (projective_icp.cpp:113-127 asserts "Not implemented"), is not declared.
#pragma once
#include <vector>
#include <kfusion/types.hpp>

namespace kfusion { namespace cuda {

class ProjectiveICP
{
public:
    static constexpr int MAX_PYRAMID_LEVELS = 4;
    using DepthPyr = std::vector<Depth>;
    using PointsPyr = std::vector<Cloud>;
    using NormalsPyr = std::vector<Normals>;

    ProjectiveICP();
    virtual ~ProjectiveICP();

    // thresholds of the correspondence test (metres / radians) and Gauss-Newton iterations per pyramid level (index 0 = finest)
    void setDistThreshold(float distance);
    float getDistThreshold() const;
    void setAngleThreshold(float angle);
    float getAngleThreshold() const;
    void setIterationsNum(const std::vector<int>& iters);
    int getUsedLevelsNum() const;
    void setDeviceLoop(bool on) { device_loop_ = on; }      // default true; poses of the two flows agree to ~1e-5

    // projective_icp.hpp:30: the Frame form.  The reference's body is CV_Assert(!"Not implemented") (projective_icp.cpp:110-123, its
    // dispatch on the pyramids commented out); the same here: error(), never a transform.
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const Frame& curr, const Frame& prev);

    // curr -> prev rigid motion.  Depth variant: masked depth + normals pyramids ("if depth(y,x) is not zero, normals(y,x) is not
    // qnan"); points variant: float4 vertex + normal pyramids.  Returns false when a normal matrix is singular.
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const DepthPyr& dcurr, const NormalsPyr ncurr, const DepthPyr dprev,
                                   const NormalsPyr nprev);
    virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const PointsPyr& vcurr, const NormalsPyr ncurr, const PointsPyr vprev,
                                   const NormalsPyr nprev);

private:
    bool iterate(Affine3f& affine, const Intr& intr, const void* const* curr, const NormalsPyr& ncurr, const void* const* prev,
                 const NormalsPyr& nprev, const size_t* curr_step, const size_t* prev_step, bool depth_variant);

    DeviceArray<float> buffer_;          // partial sums, the 27 results, the device-resident estimate
    std::vector<int> iters_;
    float dist_thres_, angle_thres_;
    bool device_loop_ = true;
};

} }
