// kfusion/cuda/tsdf_volume.hpp -- the TSDF volume class of the hot path, source compatible with the reference's
// kfusion::cuda::TsdfVolume (/root/reference/kfusion/include/kfusion/cuda/tsdf_volume.hpp:11-100): same name, same method names and
// signatures, same parameter semantics (including the truncation clamp and its setter-order quirk, tsdf_volume.cpp:63-73).
// Every compute method forwards to the C-ABI of include/dfusion.h; the class holds no kernels of its own.
//
// Differences a caller can see:
//   * get_cloud_host() / get_normal_host() return std::vector<Point> (the reference returns a 1 x N CV_32FC4 cv::Mat: the same
//     bytes) and are filled lazily -- compute_points / compute_normals only run the device kernels;
//   * integrate(dists, pose, intr, warp) is an extra overload: the per-voxel warped fusion the reference's surface_fusion is
//     meant to become;
//   * surface_fusion has an extra overload taking the warped points on the device;
//   * getGridOrigin / setGridOrigin are declared but never defined in the reference (:49-50) and are omitted;
//   * setSlab(): the volume becomes ONE Z-SLAB SHARD of the dims it was created with (one process per GPU; kfusion/cuda/zslab.hpp
//     holds the RCCL side).  Every method then works on the owned planes (+ halo planes for the ray-cast), with results identical,
//     bit for bit, to the corresponding planes / pixels of the unsharded volume.
#pragma once
#include <kfusion/types.hpp>

namespace kfusion {
class WarpField;
namespace cuda {

class TsdfVolume
{
public:
    TsdfVolume(const Vec3i& dims);
    virtual ~TsdfVolume();
    void create(const Vec3i& dims);                                       // (re)allocates dims.x * dims.y * dims.z voxels of 4 bytes

    // ---- geometry and fusion parameters
    Vec3i getDims() const;
    Vec3f getVoxelSize() const;                                            // size / dims per axis
    Vec3f getSize() const;            void setSize(const Vec3f& size);      // metres; re-applies the truncation clamp
    float getTruncDist() const;       void setTruncDist(float distance);    // >= 2.1 * largest voxel edge
    int getMaxWeight() const;         void setMaxWeight(int weight);
    Affine3f getPose() const;         void setPose(const Affine3f& pose);   // volume -> world
    float getRaycastStepFactor() const;     void setRaycastStepFactor(float factor);
    float getGradientDeltaFactor() const;   void setGradientDeltaFactor(float factor);
    virtual void applyAffine(const Affine3f& affine);                      // pose <- affine * pose

    // ---- the voxel blob (ushort2 {half tsdf, weight}, x fastest)
    const CudaData data() const;
    CudaData data();
    void swap(CudaData& data);
    virtual void clear();

    // ---- Z-slab shard (extension): own planes [z_own0, z_own0 + z_own_n) of the full dims, stored with `halo` more planes on each
    // side (clipped to the volume).  Reallocates and clears.  setSlab(0, dims.z, 0) is the unsharded volume.
    // integrate_halo: the integrate methods also update the stored HALO planes (the integrate is a pure function of the frame's
    // inputs, so the planes come out exactly as the neighbour computes them and no halo exchange is needed); ray-cast, extraction and
    // the merge still see the non-overlapping own range.
    void setSlab(int z_own0, int z_own_n, int halo, bool integrate_halo = false);
    bool isSlab() const { return has_slab_; }
    bool integratesHalo() const { return integrate_halo_; }
    int slabIntegrate0() const { return integrate_halo_ ? z_store0_ : z_own0_; }      // planes the integrate methods update
    int slabIntegrateN() const { return integrate_halo_ ? z_store_n_ : z_own_n_; }
    int slabStore0() const { return z_store0_; }
    int slabStoreN() const { return z_store_n_; }
    int slabOwn0() const { return z_own0_; }
    int slabOwnN() const { return z_own_n_; }
    // the two-stage sharded ray-cast (include/dfusion.h dfusion_raycast_march / _shade): see kfusion/cuda/zslab.hpp
    // (keys64: a dense cols x rows array of merge keys; points / normals of the shade: any pitch)
    void raycastMarch(const Affine3f& camera_pose, const Intr& intr, int cols, int rows, unsigned rank, DeviceArray<unsigned long long>& keys64) const;
    void raycastShade(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64, Cloud& points,
                      Normals& normals) const;
    // normals only (the points need no exchange), and the points from the merged keys + the SUMMED normals on the rank that wants them
    void raycastShadeNormals(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64, Normals& normals) const;
    void raycastPointsOfKeys(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64, const Normals& normals,
                             Cloud& points) const;
    // the same for the band of pixel rows [row0, row0 + normals.rows()) of a cols x image_rows image (normals / points: the band's views)
    void raycastPointsOfKeysRows(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64, int image_rows, int row0,
                                 const Normals& normals, Cloud& points) const;

    // ---- fusion
    virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr);                        // rigid, tsdf_volume.cpp:110-122
    virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr, const WarpField& warp);  // per-voxel DQB warp first
    // ... the same, enqueued only (the reference's integrate ends with a device synchronise, tsdf_volume.cu:160, and so do the two above;
    // a host that keeps frames in flight -- apps/headless_frame bench -- synchronises once per batch of frames instead)
    void integrateAsync(const Dists& dists, const Affine3f& camera_pose, const Intr& intr, const WarpField& warp);
    // tsdf_volume.cpp:228-255: depth pixels explained by a warped model point are zeroed, the rest is fused rigidly.  The per-point
    // weights loop of the reference (:241-254) has no effect there (its update lines are commented out) and is not run; psdf is
    // handed dists computed from `depth` (the reference binds the millimetre image as half, :235/:281 -- fixed).
    void surface_fusion(const WarpField& warp_field, std::vector<Vec3f> warped, std::vector<Vec3f> canonical, cuda::Depth& depth,
                        const Affine3f& camera_pose, const Intr& intr);
    void surface_fusion(const WarpField& warp_field, DeviceArray<Point>& warped /* float4, on the device */, cuda::Depth& depth,
                        const Affine3f& camera_pose, const Intr& intr);
    // tsdf_volume.cpp:266-292: ro[i] = dists at the projection of warped[i] minus warped[i].z (NaN: NaN point or outside the image);
    // the dists pixels hit by a point are zeroed
    std::vector<float> psdf(const std::vector<Vec3f>& warped, Dists& dists, const Intr& intr);
    float weighting(const std::vector<float>& dist_sqr, int k) const;      // tsdf_volume.cpp:300-306

    // ---- surface prediction
    virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals);
    virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals);

    // ---- surface extraction (tsdf_volume.cpp:181-218, 313-325)
    DeviceArray<Point> fetchCloud(DeviceArray<Point>& cloud_buffer) const;
    void fetchNormals(const DeviceArray<Point>& cloud, DeviceArray<Normal>& normals) const;
    void compute_points();
    void compute_normals();
#ifdef KFUSION_USE_OPENCV
    cv::Mat get_cloud_host() const;                                        // tsdf_volume.hpp:24-28: 1 x N CV_32FC4 (copies of the vectors below)
    cv::Mat get_normal_host() const;
    cv::Mat* get_cloud_host_ptr() const;
    cv::Mat* get_normal_host_ptr() const;
#else
    const std::vector<Point>& get_cloud_host() const { return cloud_host_vector(); }
    const std::vector<Normal>& get_normal_host() const { return normal_host_vector(); }
    const std::vector<Point>* get_cloud_host_ptr() const { return &get_cloud_host(); }      // tsdf_volume.hpp:27-28 (cv::Mat* there)
    const std::vector<Normal>* get_normal_host_ptr() const { return &get_normal_host(); }
#endif
    const std::vector<Point>& cloud_host_vector() const;                  // the host copies themselves, filled on first use after compute_*
    const std::vector<Normal>& normal_host_vector() const;
    const DeviceArray<Point>& get_cloud_device() const { return cloud_; }

private:
    CudaData data_;
    Vec3i dims_;
    Vec3f size_;
    Affine3f pose_;
    float trunc_dist_;
    float max_weight_;                                                     // a float in the reference too (tsdf_volume.hpp:86)
    float gradient_delta_factor_, raycast_step_factor_;
    DeviceArray<Point> cloud_buffer_, cloud_;
    DeviceArray<Normal> normal_buffer_;                                    // grows, never shrinks: cloud_.size() entries are valid
    mutable DeviceArray<unsigned long long> extract_count_;                // fetchCloud's device counter, allocated once
    mutable std::vector<Point> cloud_host_;
    mutable std::vector<Normal> normal_host_;
    mutable bool cloud_host_stale_ = false, normal_host_stale_ = false;
#ifdef KFUSION_USE_OPENCV
    mutable cv::Mat cloud_host_mat_, normal_host_mat_;
#endif
    Dists fusion_dists_;                                                   // scratch of surface_fusion
    bool has_slab_ = false, integrate_halo_ = false;
    int z_store0_ = 0, z_store_n_ = 0, z_own0_ = 0, z_own_n_ = 0;
};

} }
