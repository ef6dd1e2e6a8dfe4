// kfusion_hip.cpp -- host side of the drop-in boundary: kfusion::cuda::{DeviceMemory, TsdfVolume, computeDists} and
// kfusion::WarpField implemented over the C-ABI (include/dfusion.h) and the HIP runtime.  The call sequences mirror
// /root/reference/kfusion/src/tsdf_volume.cpp, device_memory.cpp, imgproc.cpp and warp_field.cpp (cited per function).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime_api.h>
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/warp_field.hpp>
#include "dfusion.h"

using namespace kfusion;
using namespace kfusion::cuda;

// ------------------------------------------------------------------------------------------ errors
// Reference: prints and exit(0)s (safe_call.hpp:13-27, device_memory.cpp:7-11).  Same line; non-zero status.
void kfusion::cuda::error(const char* error_string, const char* file, const int line, const char* func)
{
    std::printf("KinFu2 error: %s\t%s:%d\n", error_string, file, line);
    if (func && *func) std::printf("\t%s\n", func);
    std::fflush(stdout);
    std::exit(1);
}
#define KF_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) kfusion::cuda::error(hipGetErrorString(e_), __FILE__, __LINE__, #expr); } while (0)
#define KF_DF(expr) do { int e_ = (expr); if (e_ != 0) kfusion::cuda::error(dfusion_error_string(e_), __FILE__, __LINE__, #expr); } while (0)

// ------------------------------------------------------------------------------------------ device memory
static std::shared_ptr<void> hip_owned(void* p) { return std::shared_ptr<void>(p, [](void* q) { if (q) (void)hipFree(q); }); }

void DeviceMemory::create(size_t sizeBytes)             // device_memory.cpp:73-91: no-op when the size is unchanged
{
    if (sizeBytes == sizeBytes_ && data_) return;
    release();
    if (!sizeBytes) return;
    void* p = nullptr;
    KF_HIP(hipMalloc(&p, sizeBytes));
    owner_ = hip_owned(p); data_ = p; sizeBytes_ = sizeBytes;
}
void DeviceMemory::copyTo(DeviceMemory& other) const
{
    if (empty()) { other.release(); return; }
    other.create(sizeBytes_);
    KF_HIP(hipMemcpy(other.data_, data_, sizeBytes_, hipMemcpyDeviceToDevice));
}
void DeviceMemory::upload(const void* host_ptr, size_t sizeBytes)
{
    create(sizeBytes);
    if (sizeBytes) KF_HIP(hipMemcpy(data_, host_ptr, sizeBytes, hipMemcpyHostToDevice));
}
void DeviceMemory::download(void* host_ptr) const
{
    if (sizeBytes_) KF_HIP(hipMemcpy(host_ptr, data_, sizeBytes_, hipMemcpyDeviceToHost));
}
void DeviceMemory2D::create(int rows, int colsBytes)    // device_memory.cpp:180-199
{
    if (rows == rows_ && colsBytes == colsBytes_ && data_) return;
    release();
    if (rows <= 0 || colsBytes <= 0) return;
    void* p = nullptr; size_t pitch = 0;
    KF_HIP(hipMallocPitch(&p, &pitch, (size_t)colsBytes, (size_t)rows));
    owner_ = hip_owned(p); data_ = p; step_ = pitch; colsBytes_ = colsBytes; rows_ = rows;
}
void DeviceMemory2D::copyTo(DeviceMemory2D& other) const
{
    if (empty()) { other.release(); return; }
    other.create(rows_, colsBytes_);
    KF_HIP(hipMemcpy2D(other.data_, other.step_, data_, step_, (size_t)colsBytes_, (size_t)rows_, hipMemcpyDeviceToDevice));
}
void DeviceMemory2D::upload(const void* host_ptr, size_t host_step, int rows, int colsBytes)
{
    create(rows, colsBytes);
    KF_HIP(hipMemcpy2D(data_, step_, host_ptr, host_step, (size_t)colsBytes, (size_t)rows, hipMemcpyHostToDevice));
}
void DeviceMemory2D::download(void* host_ptr, size_t host_step) const
{
    if (rows_) KF_HIP(hipMemcpy2D(host_ptr, host_step, data_, step_, (size_t)colsBytes_, (size_t)rows_, hipMemcpyDeviceToHost));
}

// ------------------------------------------------------------------------------------------ imgproc
void kfusion::cuda::waitAllDefaultStream() { KF_HIP(hipDeviceSynchronize()); }

void kfusion::cuda::computeDists(const Depth& depth, Dists& dists, const Intr& intr)   // imgproc.cpp:87-91
{
    dists.create(depth.rows(), depth.cols());
    const float in[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_compute_dists(depth.ptr(), depth.step(), dists.ptr(), dists.step(), depth.cols(), depth.rows(), in, nullptr));
}

// ------------------------------------------------------------------------------------------ TsdfVolume
static DfVolume c_volume(const TsdfVolume& v)
{
    DfVolume d;
    d.data = const_cast<void*>(static_cast<const void*>(v.data().ptr<char>()));
    const Vec3i dims = v.getDims(); const Vec3f vs = v.getVoxelSize();
    for (int i = 0; i < 3; ++i) { d.dims[i] = dims[i]; d.voxel_size[i] = vs[i]; }
    d.trunc_dist = v.getTruncDist();
    d.max_weight = v.getMaxWeight();
    return d;
}

TsdfVolume::TsdfVolume(const Vec3i& dims)               // tsdf_volume.cpp:7-17
    : data_(), trunc_dist_(0.03f), max_weight_(128), dims_(dims), size_(Vec3f::all(3.f)), pose_(Affine3f::Identity()),
      gradient_delta_factor_(0.75f), raycast_step_factor_(0.75f)
{
    create(dims_);
}
TsdfVolume::~TsdfVolume() {}

void TsdfVolume::create(const Vec3i& dims)              // tsdf_volume.cpp:32-39 (size_t, not int: no overflow at 2048^3)
{
    dims_ = dims;
    const size_t voxels_number = (size_t)dims_[0] * dims_[1] * dims_[2];
    data_.create(voxels_number * sizeof(int));
    setTruncDist(trunc_dist_);
    clear();
}
Vec3i TsdfVolume::getDims() const { return dims_; }
Vec3f TsdfVolume::getVoxelSize() const { return Vec3f(size_[0] / dims_[0], size_[1] / dims_[1], size_[2] / dims_[2]); }   // :54-57
const CudaData TsdfVolume::data() const { return data_; }
CudaData TsdfVolume::data() { return data_; }
Vec3f TsdfVolume::getSize() const { return size_; }
void TsdfVolume::setSize(const Vec3f& size) { size_ = size; setTruncDist(trunc_dist_); }                                  // :63-64
float TsdfVolume::getTruncDist() const { return trunc_dist_; }
void TsdfVolume::setTruncDist(float distance)           // :68-73
{
    const Vec3f vsz = getVoxelSize();
    const float max_coeff = std::max<float>(std::max<float>(vsz[0], vsz[1]), vsz[2]);
    trunc_dist_ = std::max(distance, 2.1f * max_coeff);
}
int TsdfVolume::getMaxWeight() const { return (int)max_weight_; }
void TsdfVolume::setMaxWeight(int weight) { max_weight_ = (float)weight; }
Affine3f TsdfVolume::getPose() const { return pose_; }
void TsdfVolume::setPose(const Affine3f& pose) { pose_ = pose; }
float TsdfVolume::getRaycastStepFactor() const { return raycast_step_factor_; }
void TsdfVolume::setRaycastStepFactor(float factor) { raycast_step_factor_ = factor; }
float TsdfVolume::getGradientDeltaFactor() const { return gradient_delta_factor_; }
void TsdfVolume::setGradientDeltaFactor(float factor) { gradient_delta_factor_ = factor; }
void TsdfVolume::swap(CudaData& data) { data_.swap(data); }
void TsdfVolume::applyAffine(const Affine3f& affine) { pose_ = affine * pose_; }                                          // :88

void TsdfVolume::clear()                                // :89-102 (without the five leaked heap objects)
{
    KF_DF(dfusion_clear(c_volume(*this), nullptr, nullptr));
}

void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr)   // :110-122
{
    const Affine3f vol2cam = camera_pose.inv() * pose_;
    float aff[12]; affine_to_aff12(vol2cam, aff);
    const float proj[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_integrate(dists.ptr(), dists.step(), dists.cols(), dists.rows(), c_volume(*this), nullptr, aff, proj, nullptr, nullptr));
    KF_HIP(hipDeviceSynchronize());                     // device::integrate ends with cudaDeviceSynchronize (tsdf_volume.cu:160)
}

void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr, const WarpField& warp)
{
    warp.ensureIndex(*this);
    float v2w[12], w2c[12];
    affine_to_aff12(pose_, v2w);
    affine_to_aff12(camera_pose.inv() * warp.getWarpToLive(), w2c);
    const float proj[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_integrate_warped(dists.ptr(), dists.step(), dists.cols(), dists.rows(), c_volume(*this), nullptr, v2w, w2c, proj,
                                   warp.handle(), warp.k(), 0u, nullptr, nullptr));
    KF_HIP(hipDeviceSynchronize());
}

static void raycast_args(const Affine3f& pose, const Affine3f& camera_pose, const Intr& intr, float aff[12], float Rinv[9], float reproj[4])
{
    const Affine3f cam2vol = pose.inv() * camera_pose;                      // :135 / :162
    affine_to_aff12(cam2vol, aff);
    const Mat3f ri = cam2vol.rotation().inv();                              // :138 / :165 inv(DECOMP_SVD)
    for (int i = 0; i < 9; ++i) Rinv[i] = ri.val[i];
    reproj[0] = 1.f / intr.fx; reproj[1] = 1.f / intr.fy; reproj[2] = intr.cx; reproj[3] = intr.cy;   // precomp.cpp:55
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals)   // :131-148
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_depth(c_volume(*this), nullptr, aff, Rinv, reproj, depth.ptr(), depth.step(), (float*)normals.ptr(),
                                normals.step(), depth.cols(), depth.rows(), raycast_step_factor_, gradient_delta_factor_, nullptr));
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals)  // :157-174 (async: caller syncs)
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_points(c_volume(*this), nullptr, aff, Rinv, reproj, (float*)points.ptr(), points.step(), (float*)normals.ptr(),
                                 normals.step(), points.cols(), points.rows(), raycast_step_factor_, gradient_delta_factor_, nullptr,
                                 nullptr));
}

// ------------------------------------------------------------------------------------------ extraction (tsdf_volume.cpp:181-218,313-325)
DeviceArray<Point> TsdfVolume::fetchCloud(DeviceArray<Point>& cloud_buffer) const
{
    enum { DEFAULT_CLOUD_BUFFER_SIZE = 256 * 256 * 256 };                   // :184
    if (cloud_buffer.empty()) cloud_buffer.create(DEFAULT_CLOUD_BUFFER_SIZE);
    float aff[12]; affine_to_aff12(pose_, aff);
    DeviceArray<unsigned long long> count(1);
    KF_HIP(hipMemset(count.ptr(), 0, sizeof(unsigned long long)));
    KF_DF(dfusion_extract_cloud(c_volume(*this), nullptr, aff, (float*)cloud_buffer.ptr(), cloud_buffer.size(), count.ptr(), nullptr));
    unsigned long long n = 0;
    count.download(&n);                                                     // cudaMemcpyFromSymbol(output_count), tsdf_volume.cu:815
    if (n > cloud_buffer.size()) n = cloud_buffer.size();
    return DeviceArray<Point>(cloud_buffer.ptr(), (size_t)n);               // non-owning view, like the reference (:198)
}

void TsdfVolume::fetchNormals(const DeviceArray<Point>& cloud, DeviceArray<Normal>& normals) const
{
    normals.create(cloud.size());
    if (!cloud.size()) return;
    float aff[12]; affine_to_aff12(pose_, aff);
    const Mat3f ri = pose_.rotation().inv();                                // :214 inv(DECOMP_SVD)
    KF_DF(dfusion_extract_normals(c_volume(*this), nullptr, aff, ri.val, (const float*)cloud.ptr(), cloud.size(), gradient_delta_factor_,
                                  (float*)normals.ptr(), nullptr));
}

void TsdfVolume::compute_points()
{
    cloud_ = fetchCloud(cloud_buffer_);
    cloud_host_.resize(cloud_.size());
    if (cloud_.size()) cloud_.download(cloud_host_.data());
}

void TsdfVolume::compute_normals()
{
    fetchNormals(cloud_, normal_buffer_);
    normal_host_.resize(cloud_.size());
    if (cloud_.size()) normal_buffer_.download(normal_host_.data());
}

// ------------------------------------------------------------------------------------------ psdf / surface_fusion (tsdf_volume.cpp:228-306)
// project_and_remove + the K^-1 arithmetic run in one kernel; `removed` receives the zeros (dists itself when null).
static std::vector<float> psdf_impl(const std::vector<Vec3f>& warped, const Dists& dists, DeviceArray2D<unsigned short>* removed,
                                    const Intr& intr)
{
    std::vector<float> distances(warped.size());
    if (warped.empty()) return distances;
    std::vector<Point> pts(warped.size());
    for (size_t i = 0; i < warped.size(); ++i) { pts[i].x = warped[i][0]; pts[i].y = warped[i][1]; pts[i].z = warped[i][2]; pts[i].data[3] = 0.f; }   // :272-278
    DeviceArray<Point> d_pts; d_pts.upload(pts);
    DeviceArray<float> d_ro(warped.size());
    const float proj[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_project_and_remove(dists.ptr(), dists.step(), removed->ptr(), removed->step(), dists.cols(), dists.rows(),
                                     (float*)d_pts.ptr(), warped.size(), proj, d_ro.ptr(), nullptr, nullptr));
    d_ro.download(distances.data());
    return distances;
}

std::vector<float> TsdfVolume::psdf(const std::vector<Vec3f>& warped, Dists& dists, const Intr& intr)
{
    Dists snapshot;                                                         // samples come from an immutable copy (include/dfusion.h)
    snapshot.create(dists.rows(), dists.cols());
    KF_HIP(hipMemcpy2D(snapshot.ptr(), snapshot.step(), dists.ptr(), dists.step(), dists.cols() * sizeof(unsigned short), dists.rows(),
                       hipMemcpyDeviceToDevice));
    return psdf_impl(warped, snapshot, &dists, intr);
}

float TsdfVolume::weighting(const std::vector<float>& dist_sqr, int k) const
{
    float distances = 0;
    for (float d : dist_sqr) distances += std::sqrt(d);
    return distances / k;
}

void TsdfVolume::surface_fusion(const WarpField& /*warp_field*/, std::vector<Vec3f> warped, std::vector<Vec3f> /*canonical*/,
                                cuda::Depth& depth, const Affine3f& camera_pose, const Intr& intr)
{
    cuda::Dists dists;
    cuda::computeDists(depth, dists, intr);                                 // metres for psdf (the reference samples the mm image as half)
    std::vector<float> ro = psdf_impl(warped, dists, &depth, intr);         // :235  zeroes the explained pixels of `depth`
    (void)ro;
    cuda::computeDists(depth, dists, intr);                                 // :237-238
    integrate(dists, camera_pose, intr);                                    // :239
}

// ------------------------------------------------------------------------------------------ WarpField
WarpField::WarpField(int k) : k_(k), handle_(nullptr), out_dist_sqr_(k), ret_index_(k), index_ok_(false), index_volume_(nullptr)
{
    KF_DF(dfusion_warp_create(&handle_));
}
WarpField::~WarpField() { dfusion_warp_destroy(handle_); }

void WarpField::init(const std::vector<Vec3f>& first_frame)
{
    nodes_.clear();
    for (const Vec3f& p : first_frame) {
        if (std::isnan(p[0])) continue;
        deformation_node n;
        n.vertex = p;
        n.transform = utils::DualQuaternion<float>();
        n.weight = 3.f;                                   // warp_field.cpp:84 (3 * voxel_size with voxel_size forced to 1)
        nodes_.push_back(n);
    }
    commit(true);
}

void WarpField::commit(bool positions_changed)
{
    const size_t M = nodes_.size();
    std::vector<float> dq(M * 8);
    for (size_t i = 0; i < M; ++i) std::memcpy(&dq[8 * i], nodes_[i].transform.raw(), 32);
    DeviceArray<float> d_dq; d_dq.upload(dq);
    if (positions_changed) {
        std::vector<float> pos(M * 3), sigma(M);
        for (size_t i = 0; i < M; ++i) { for (int c = 0; c < 3; ++c) pos[3 * i + c] = nodes_[i].vertex[c]; sigma[i] = nodes_[i].weight; }
        DeviceArray<float> d_pos, d_sigma; d_pos.upload(pos); d_sigma.upload(sigma);
        KF_DF(dfusion_warp_set_nodes(handle_, d_pos.ptr(), d_dq.ptr(), d_sigma.ptr(), (int)M, nullptr));
        index_ok_ = false;
    } else {
        KF_DF(dfusion_warp_set_transforms(handle_, d_dq.ptr(), nullptr));
    }
    KF_HIP(hipDeviceSynchronize());                       // the temporaries above are freed on return
}

void WarpField::ensureIndex(const cuda::TsdfVolume& volume) const
{
    if (index_ok_ && index_volume_ == &volume) return;
    float v2w[12]; affine_to_aff12(volume.getPose(), v2w);
    KF_DF(dfusion_warp_build_index(handle_, c_volume(volume), nullptr, v2w, k_, DF_INDEX_VOXEL_TABLE | DF_INDEX_WEIGHT_TABLE, nullptr));
    index_ok_ = true; index_volume_ = &volume;
}

void WarpField::KNN(Vec3f point) const
{
    DeviceArray<float> q; q.upload(point.val, 3);
    DeviceArray<int> idx(k_); DeviceArray<float> d2(k_);
    KF_DF(dfusion_knn(handle_, k_, q.ptr(), 1, idx.ptr(), d2.ptr(), nullptr));
    std::vector<int> hi; idx.download(hi); d2.download(out_dist_sqr_);
    for (int i = 0; i < k_; ++i) ret_index_[i] = (size_t)hi[i];
}

void WarpField::warp(std::vector<Vec3f>& points, std::vector<Vec3f>& normals) const
{
    static_assert(sizeof(Vec3f) == 12, "Vec3f must be 3 packed floats");
    DeviceArray<float> p, n;
    p.upload(points[0].val, points.size() * 3);
    if (!normals.empty()) n.upload(normals[0].val, normals.size() * 3);
    float live[12]; affine_to_aff12(warp_to_live_, live);
    KF_DF(dfusion_warp_points(handle_, k_, p.ptr(), normals.empty() ? nullptr : n.ptr(), (int)points.size(), live, nullptr));
    p.download(points[0].val);
    if (!normals.empty()) n.download(normals[0].val);
}
