// kfusion_hip.cpp -- host side of the drop-in boundary: kfusion::cuda::{DeviceMemory, TsdfVolume, computeDists} and
// kfusion::WarpField implemented over the C-ABI (include/dfusion.h) and the HIP runtime.  The call sequences mirror
// /root/reference/kfusion/src/tsdf_volume.cpp, device_memory.cpp, imgproc.cpp and warp_field.cpp (cited per function).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime_api.h>
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/warp_field.hpp>
#include <kfusion/cuda/projective_icp.hpp>
#include <kfusion/kinfu.hpp>
#include <algorithm>
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

static void raycast_args(const Affine3f& pose, const Affine3f& camera_pose, const Intr& intr, float aff[12], float Rinv[9], float reproj[4]);
// the slab descriptor of a sharded volume for the C-ABI (null = the whole volume)
struct SlabArg { DfSlab s; bool on; const DfSlab* ptr() const { return on ? &s : nullptr; } };
static SlabArg c_slab(const TsdfVolume& v)
{
    SlabArg a; a.on = v.isSlab();
    a.s.z_store0 = v.slabStore0(); a.s.z_store_n = v.slabStoreN(); a.s.z_own0 = v.slabOwn0(); a.s.z_own_n = v.slabOwnN();
    return a;
}
// the same for the integrate methods (and the k-NN tables they stream): with setSlab(..., integrate_halo = true) they own every stored plane
static SlabArg c_slab_integrate(const TsdfVolume& v)
{
    SlabArg a = c_slab(v);
    a.s.z_own0 = v.slabIntegrate0(); a.s.z_own_n = v.slabIntegrateN();
    return a;
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
    has_slab_ = false; z_store0_ = 0; z_store_n_ = dims[2]; z_own0_ = 0; z_own_n_ = dims[2];
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
    KF_DF(dfusion_clear(c_volume(*this), c_slab(*this).ptr(), nullptr));
}

void TsdfVolume::setSlab(int z_own0, int z_own_n, int halo, bool integrate_halo)
{
    const int Z = dims_[2];
    if (z_own0 < 0 || z_own_n < 0 || z_own0 + z_own_n > Z || halo < 0) kfusion::cuda::error("setSlab: planes outside the volume", __FILE__, __LINE__, "setSlab");
    z_own0_ = z_own0; z_own_n_ = z_own_n;
    z_store0_ = std::max(0, z_own0 - halo);
    z_store_n_ = std::min(Z, z_own0 + z_own_n + halo) - z_store0_;
    has_slab_ = true; integrate_halo_ = integrate_halo;
    data_.create((size_t)dims_[0] * dims_[1] * (size_t)std::max(z_store_n_, 1) * sizeof(int));
    clear();
}

void TsdfVolume::raycastMarch(const Affine3f& camera_pose, const Intr& intr, int cols, int rows, unsigned rank,
                              DeviceArray<unsigned long long>& keys64) const
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    keys64.create((size_t)cols * rows);
    KF_DF(dfusion_raycast_march(c_volume(*this), c_slab(*this).ptr(), aff, reproj, cols, rows, raycast_step_factor_, rank, keys64.ptr(), nullptr));
}
void TsdfVolume::raycastShade(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64, Cloud& points,
                              Normals& normals) const
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_shade(c_volume(*this), c_slab(*this).ptr(), aff, Rinv, reproj, merged_keys64.ptr(),
                                (float*)points.ptr(), points.step(), (float*)normals.ptr(), normals.step(), points.cols(), points.rows(),
                                gradient_delta_factor_, nullptr));
}

void TsdfVolume::raycastShadeNormals(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64,
                                     Normals& normals) const
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_shade(c_volume(*this), c_slab(*this).ptr(), aff, Rinv, reproj, merged_keys64.ptr(), nullptr, 0,
                                (float*)normals.ptr(), normals.step(), normals.cols(), normals.rows(), gradient_delta_factor_, nullptr));
}
void TsdfVolume::raycastPointsOfKeys(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64,
                                     const Normals& normals, Cloud& points) const
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_points_of_keys(aff, Rinv, reproj, merged_keys64.ptr(), (const float*)normals.ptr(), normals.step(), (float*)points.ptr(),
                                         points.step(), points.cols(), points.rows(), nullptr));
}

void TsdfVolume::raycastPointsOfKeysRows(const Affine3f& camera_pose, const Intr& intr, const DeviceArray<unsigned long long>& merged_keys64, int image_rows,
                                         int row0, const Normals& normals, Cloud& points) const
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_points_of_keys_rows(aff, Rinv, reproj, merged_keys64.ptr(), (const float*)normals.ptr(), normals.step(), (float*)points.ptr(),
                                              points.step(), points.cols(), image_rows, row0, normals.rows(), nullptr));
}

void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr)   // :110-122
{
    const Affine3f vol2cam = camera_pose.inv() * pose_;
    float aff[12]; affine_to_aff12(vol2cam, aff);
    const float proj[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_integrate(dists.ptr(), dists.step(), dists.cols(), dists.rows(), c_volume(*this), c_slab_integrate(*this).ptr(), aff, proj, nullptr, nullptr));
    KF_HIP(hipDeviceSynchronize());                     // device::integrate ends with cudaDeviceSynchronize (tsdf_volume.cu:160)
}

void TsdfVolume::integrateAsync(const Dists& dists, const Affine3f& camera_pose, const Intr& intr, const WarpField& warp)
{
    warp.ensureIndex(*this);
    float v2w[12], w2c[12];
    affine_to_aff12(pose_, v2w);
    affine_to_aff12(camera_pose.inv() * warp.getWarpToLive(), w2c);
    const float proj[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_integrate_warped(dists.ptr(), dists.step(), dists.cols(), dists.rows(), c_volume(*this), c_slab_integrate(*this).ptr(), v2w, w2c, proj,
                                   warp.handle(), warp.k(), 0u, nullptr, nullptr));
}
void TsdfVolume::integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr, const WarpField& warp)
{
    integrateAsync(dists, camera_pose, intr, warp);
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
    KF_DF(dfusion_raycast_depth(c_volume(*this), c_slab(*this).ptr(), aff, Rinv, reproj, depth.ptr(), depth.step(), (float*)normals.ptr(),
                                normals.step(), depth.cols(), depth.rows(), raycast_step_factor_, gradient_delta_factor_, nullptr));
}

void TsdfVolume::raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals)  // :157-174 (async: caller syncs)
{
    float aff[12], Rinv[9], reproj[4];
    raycast_args(pose_, camera_pose, intr, aff, Rinv, reproj);
    KF_DF(dfusion_raycast_points(c_volume(*this), c_slab(*this).ptr(), aff, Rinv, reproj, (float*)points.ptr(), points.step(), (float*)normals.ptr(),
                                 normals.step(), points.cols(), points.rows(), raycast_step_factor_, gradient_delta_factor_, nullptr,
                                 nullptr));
}

// ------------------------------------------------------------------------------------------ extraction (tsdf_volume.cpp:181-218,313-325)
DeviceArray<Point> TsdfVolume::fetchCloud(DeviceArray<Point>& cloud_buffer) const
{
    enum { DEFAULT_CLOUD_BUFFER_SIZE = 256 * 256 * 256 };                   // :184
    if (cloud_buffer.empty()) cloud_buffer.create(DEFAULT_CLOUD_BUFFER_SIZE);
    float aff[12]; affine_to_aff12(pose_, aff);
    if (extract_count_.empty()) extract_count_.create(1);                   // (a hipMalloc + hipFree per frame otherwise: ~0.1 ms)
    DeviceArray<unsigned long long>& count = extract_count_;
    KF_HIP(hipMemset(count.ptr(), 0, sizeof(unsigned long long)));
    KF_DF(dfusion_extract_cloud(c_volume(*this), c_slab(*this).ptr(), aff, (float*)cloud_buffer.ptr(), cloud_buffer.size(), count.ptr(), nullptr));
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
    KF_DF(dfusion_extract_normals(c_volume(*this), c_slab(*this).ptr(), aff, ri.val, (const float*)cloud.ptr(), cloud.size(), gradient_delta_factor_,
                                  (float*)normals.ptr(), nullptr));
}

void TsdfVolume::compute_points()
{
    cloud_ = fetchCloud(cloud_buffer_);
    cloud_host_stale_ = true;
}

void TsdfVolume::compute_normals()
{
    // fetchNormals(cloud_, normal_buffer_) without its per-frame reallocation (the cloud size changes every frame): the buffer only
    // grows, the first cloud_.size() normals are the current ones
    if (normal_buffer_.size() < cloud_.size()) normal_buffer_.create(cloud_.size() + cloud_.size() / 4 + 1024);
    if (cloud_.size()) {
        float aff[12]; affine_to_aff12(pose_, aff);
        const Mat3f ri = pose_.rotation().inv();                            // :214 inv(DECOMP_SVD)
        KF_DF(dfusion_extract_normals(c_volume(*this), c_slab(*this).ptr(), aff, ri.val, (const float*)cloud_.ptr(), cloud_.size(), gradient_delta_factor_,
                                      (float*)normal_buffer_.ptr(), nullptr));
    }
    normal_host_stale_ = true;
}

const std::vector<Point>& TsdfVolume::cloud_host_vector() const
{
    if (cloud_host_stale_) {
        cloud_host_.resize(cloud_.size());
        if (cloud_.size()) cloud_.download(cloud_host_.data());
        cloud_host_stale_ = false;
    }
    return cloud_host_;
}

const std::vector<Normal>& TsdfVolume::normal_host_vector() const
{
    if (normal_host_stale_) {
        normal_host_.resize(cloud_.size());
        if (cloud_.size()) KF_HIP(hipMemcpy(normal_host_.data(), normal_buffer_.ptr(), cloud_.size() * sizeof(Normal), hipMemcpyDeviceToHost));
        normal_host_stale_ = false;
    }
    return normal_host_;
}
#ifdef KFUSION_USE_OPENCV
// tsdf_volume.cpp:74-77: the reference keeps the clouds as 1 x N CV_32FC4 matrices
template <typename T> static void kf_vector_to_mat(const std::vector<T>& v, cv::Mat& m)
{
    m.create(1, (int)v.size(), CV_32FC4);
    if (!v.empty()) std::memcpy(m.data, v.data(), v.size() * sizeof(T));
}
cv::Mat TsdfVolume::get_cloud_host() const { kf_vector_to_mat(cloud_host_vector(), cloud_host_mat_); return cloud_host_mat_; }
cv::Mat TsdfVolume::get_normal_host() const { kf_vector_to_mat(normal_host_vector(), normal_host_mat_); return normal_host_mat_; }
cv::Mat* TsdfVolume::get_cloud_host_ptr() const { kf_vector_to_mat(cloud_host_vector(), cloud_host_mat_); return &cloud_host_mat_; }
cv::Mat* TsdfVolume::get_normal_host_ptr() const { kf_vector_to_mat(normal_host_vector(), normal_host_mat_); return &normal_host_mat_; }
#endif

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

void TsdfVolume::surface_fusion(const WarpField& /*warp_field*/, DeviceArray<Point>& warped, cuda::Depth& depth, const Affine3f& camera_pose,
                                const Intr& intr)
{
    cuda::computeDists(depth, fusion_dists_, intr);
    const float proj[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    if (warped.size())
        KF_DF(dfusion_project_and_remove(fusion_dists_.ptr(), fusion_dists_.step(), depth.ptr(), depth.step(), depth.cols(), depth.rows(),
                                         (float*)warped.ptr(), warped.size(), proj, nullptr, nullptr, nullptr));
    cuda::computeDists(depth, fusion_dists_, intr);
    integrate(fusion_dists_, camera_pose, intr);
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
    nodes_.clear(); nodes_stale_ = false;
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

#ifdef KFUSION_USE_OPENCV
// warp_field.cpp:41-63: every 50th point of every 50th row of a cloud image.  (KinFu hands it the extracted cloud, 1 x N: every 50th
// point, kinfu.cpp:252.)  The reference sizes its node array for EVERY pixel and fills only the sampled ones, leaving the rest
// zero-positioned with weight 0; here only the sampled, non-NaN points become nodes (as in init(std::vector)).
void WarpField::init(const cv::Mat& first_frame)
{
    std::vector<Vec3f> seeds;
    const int step = 50;                                   // :49
    for (int i = 0; i < first_frame.rows; i += step)
        for (int j = 0; j < first_frame.cols; j += step) {
            const Point point = first_frame.at<Point>(i, j);
            if (!std::isnan(point.x)) seeds.push_back(Vec3f(point.x, point.y, point.z));
        }
    init(seeds);
}
#endif

// warp_field.cpp:238-241
// (the reference's unqualified exp(float) resolves to the C library's double exp -- oracle/ref_glue.cpp pins that with a static_assert --
// so the argument, formed in float, is widened: std::exp(float) would pick expf and differ in the last bit for ~0.3 % of the weights)
float WarpField::weighting(float squared_dist, float weight) const { return (float)std::exp((double)(-squared_dist / (2 * weight * weight))); }

// warp_field.cpp:225-230: the k-NN on the GPU (KNN above), the weights on the host with the reference's expression; entries past k() are 0
void WarpField::getWeightsAndUpdateKNN(const Vec3f& vertex, float weights[KNN_NEIGHBOURS]) const
{
    KNN(vertex);
    for (int i = 0; i < KNN_NEIGHBOURS; ++i)
        weights[i] = i < k_ ? weighting(out_dist_sqr_[i], nodes_[ret_index_[i]].weight) : 0.f;
}

// warp_field.cpp:203-217: the blend at ONE vertex -- the per-point form of what dfusion_warp_points / dfusion_integrate_warped do on the
// device, operation for operation (sums in neighbour order, normalize, DualQuaternion(translation, rotation))
utils::DualQuaternion<float> WarpField::DQB(const Vec3f& vertex) const
{
    float weights[KNN_NEIGHBOURS];
    getWeightsAndUpdateKNN(vertex, weights);
    pullNodes();
    utils::Quaternion<float> translation_sum(0, 0, 0, 0), rotation_sum(0, 0, 0, 0);
    for (int i = 0; i < k_; ++i) {
        translation_sum = translation_sum + weights[i] * nodes_[ret_index_[i]].transform.getTranslation();     // :211
        rotation_sum = rotation_sum + weights[i] * nodes_[ret_index_[i]].transform.getRotation();              // :212
    }
    rotation_sum.normalize();                                                                                  // :214
    return utils::DualQuaternion<float>(translation_sum, rotation_sum);
}

// warp_field.cpp:98-108 (two asserts on the image sizes), :168-172 and :298-301 (empty bodies): nothing to do here either
void WarpField::energy(const cuda::Cloud& frame, const cuda::Normals& normals, const Affine3f&, const cuda::TsdfVolume&,
                       const std::vector<std::pair<utils::DualQuaternion<float>, utils::DualQuaternion<float>>>&)
{
    if (normals.cols() != frame.cols() || normals.rows() != frame.rows()) error("WarpField::energy: normals and frame differ in size", __FILE__, __LINE__, "");
}
void WarpField::energy_reg(const std::vector<std::pair<utils::DualQuaternion<float>, utils::DualQuaternion<float>>>&) {}
void WarpField::clear() {}

void WarpField::commit(bool positions_changed)
{
    pullNodes();                                           // (edits were made through getNodes(), which had pulled already)
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

void WarpField::setTransformsDevice(const cuda::DeviceArray<float>& dq8)
{
    const size_t M = nodes_.size();
    if (!M || dq8.size() != M * 8) error("WarpField::setTransformsDevice: 8 floats per node expected", __FILE__, __LINE__, "");
    KF_DF(dfusion_warp_set_transforms(handle_, dq8.ptr(), nullptr));
    solve_dq_.create(M * 8);                                 // what the host node store is refreshed from when it is next looked at
    KF_HIP(hipMemcpyAsync(solve_dq_.ptr(), dq8.ptr(), M * 8 * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    nodes_stale_ = true;
}

void WarpField::energy_data(const cuda::DeviceArray<float>& canonical_vertices, const cuda::DeviceArray<float>& live_vertices, int n)
{
    const size_t M = nodes_.size();
    if (!M || n <= 0) return;
    solve_dq_.create(M * 8); solve_energy_.create(2);      // no-ops after the first frame
    KF_DF(dfusion_warp_solve_data_term(handle_, k_, canonical_vertices.ptr(), live_vertices.ptr(), n, solver_iters_, solver_lambda_, solve_dq_.ptr(),
                                       track_energy_ ? solve_energy_.ptr() : nullptr, nullptr));
    if (track_energy_) solve_energy_.download(last_energy_);
    nodes_stale_ = true;                                   // the host node store follows (updateWarp, optimisation.hpp:211-218) -- when it is looked at
}

void WarpField::pullNodes() const
{
    if (!nodes_stale_) return;
    nodes_stale_ = false;
    const size_t M = nodes_.size();
    if (!M || solve_dq_.size() != M * 8) return;
    std::vector<float> dq(M * 8);
    solve_dq_.download(dq.data());                         // (synchronises with the solve)
    for (size_t i = 0; i < M; ++i) std::memcpy((void*)nodes_[i].transform.raw(), &dq[8 * i], 32);
}

void WarpField::energy_data(const std::vector<Vec3f>& canonical_vertices, const std::vector<Vec3f>& /*canonical_normals*/,
                            const std::vector<Vec3f>& live_vertices, const std::vector<Vec3f>& /*live_normals*/)
{
    const size_t n = std::min(canonical_vertices.size(), live_vertices.size());
    if (!n) return;
    DeviceArray<float> c, l;
    c.upload(canonical_vertices[0].val, n * 3);
    l.upload(live_vertices[0].val, n * 3);
    energy_data(c, l, (int)n);
    KF_HIP(hipDeviceSynchronize());                       // the solve reads c and l, which are freed on return (hipFree does not wait for it)
}

std::vector<Vec3f> WarpField::getNodesAsVector() const                 // warp_field.cpp:284-293
{
    pullNodes();
    std::vector<Vec3f> m(nodes_.size());
    for (size_t i = 0; i < nodes_.size(); ++i) {
        float x, y, z;
        nodes_[i].transform.getTranslation(x, y, z);
        m[i] = Vec3f(x, y, z) + nodes_[i].vertex;                         // matrix.at(i) += vertex
    }
    return m;
}
const WarpField::NodesMat WarpField::getNodesAsMat() const
{
#ifdef KFUSION_USE_OPENCV
    const std::vector<Vec3f> v = getNodesAsVector();
    cv::Mat matrix(1, (int)v.size(), CV_32FC3);                            // warp_field.cpp:286
    for (size_t i = 0; i < v.size(); ++i) matrix.at<cv::Vec3f>((int)i) = v[i];
    return matrix;
#else
    return getNodesAsVector();
#endif
}

void WarpField::ensureIndex(const cuda::TsdfVolume& volume, bool tables) const
{
    // the index belongs to one volume GEOMETRY (dims, voxel size, pose), not to an object address: setPose / setSize / applyAffine
    // on the same TsdfVolume must rebuild it (dfusion_integrate_warped would refuse the stale one with DF_E_NO_INDEX)
    float v2w[12]; affine_to_aff12(volume.getPose(), v2w);
    const Vec3i d = volume.getDims(); const Vec3f vs = volume.getVoxelSize();
    float key[20] = {(float)d[0], (float)d[1], (float)d[2], vs[0], vs[1], vs[2]};
    std::memcpy(key + 6, v2w, sizeof(v2w));
    key[18] = (float)volume.slabIntegrate0(); key[19] = volume.isSlab() ? (float)volume.slabIntegrateN() : -1.f;
    const bool same = index_ok_ && index_volume_ == &volume && std::memcmp(key, index_key_, sizeof(key)) == 0;
    if (same && (index_tables_ || !tables)) return;
    std::memcpy(index_key_, key, sizeof(key));
    KF_DF(dfusion_warp_build_index(handle_, c_volume(volume), c_slab_integrate(volume).ptr(), v2w, k_, tables ? (DF_INDEX_VOXEL_TABLE | DF_INDEX_WEIGHT_TABLE | DF_INDEX_TABLES_ON_DEMAND) : 0u, nullptr));
    index_ok_ = true; index_volume_ = &volume; index_tables_ = tables;
}

std::vector<unsigned long long> WarpField::aliveBlocksPerLayer(const cuda::TsdfVolume& volume) const
{
    const int layers = volume.getDims()[2] / 8;
    std::vector<unsigned long long> out;
    if (layers <= 0) return out;
    cuda::DeviceArray<unsigned long long> dev((size_t)layers);
    if (hipMemset(dev.ptr(), 0, (size_t)layers * sizeof(unsigned long long)) != hipSuccess) return out;
    const int z0 = volume.isSlab() ? volume.slabOwn0() : 0, zn = volume.isSlab() ? volume.slabOwnN() : volume.getDims()[2];
    if (dfusion_warp_alive_blocks(handle_, z0, zn, dev.ptr(), layers, nullptr) != DF_OK) return out;      // (DF_E_NO_INDEX: nothing swept yet)
    out.resize((size_t)layers);
    dev.download(out.data());
    return out;
}

void WarpField::KNN(Vec3f point) const
{
    DeviceArray<float> q; q.upload(point.val, 3);
    DeviceArray<int> idx(k_); DeviceArray<float> d2(k_);
    KF_DF(dfusion_knn(handle_, k_, q.ptr(), 1, idx.ptr(), d2.ptr(), nullptr));
    std::vector<int> hi; idx.download(hi); d2.download(out_dist_sqr_);
    for (int i = 0; i < k_; ++i) ret_index_[i] = (size_t)hi[i];
}

void WarpField::warp(cuda::DeviceArray<float>& points, cuda::DeviceArray<float>& normals, int n) const
{
    float live[12]; affine_to_aff12(warp_to_live_, live);
    KF_DF(dfusion_warp_points(handle_, k_, points.ptr(), normals.empty() ? nullptr : normals.ptr(), n, live, nullptr));
}

void WarpField::warp(std::vector<Vec3f>& points, std::vector<Vec3f>& normals) const
{
    static_assert(sizeof(Vec3f) == 12, "Vec3f must be 3 packed floats");
    if (points.empty()) return;
    DeviceArray<float> p, n;
    p.upload(points[0].val, points.size() * 3);
    if (!normals.empty()) n.upload(normals[0].val, normals.size() * 3);
    float live[12]; affine_to_aff12(warp_to_live_, live);
    KF_DF(dfusion_warp_points(handle_, k_, p.ptr(), normals.empty() ? nullptr : n.ptr(), (int)points.size(), live, nullptr));
    p.download(points[0].val);
    if (!normals.empty()) n.download(normals[0].val);
}

// ------------------------------------------------------------------------------------------ depth front-end (imgproc.cpp:10-141)
void kfusion::cuda::depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth)
{
    out.create(in.rows(), in.cols());
    KF_DF(dfusion_bilateral_filter(in.ptr(), in.step(), out.ptr(), out.step(), in.cols(), in.rows(), ksz, sigma_spatial, sigma_depth, nullptr));
}
void kfusion::cuda::cloudToDepth(const Cloud& cloud, Depth& depth)
{
    depth.create(cloud.rows(), cloud.cols());
    KF_DF(dfusion_cloud_to_depth((const float*)cloud.ptr(), cloud.step(), depth.ptr(), depth.step(), cloud.cols(), cloud.rows(), nullptr));
}
void kfusion::cuda::depthTruncation(Depth& depth, float threshold)
{
    KF_DF(dfusion_truncate_depth(depth.ptr(), depth.step(), depth.cols(), depth.rows(), threshold, nullptr));
}
void kfusion::cuda::depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth)
{
    pyramid.create(depth.rows() / 2, depth.cols() / 2);
    KF_DF(dfusion_depth_pyramid(depth.ptr(), depth.step(), depth.cols(), depth.rows(), pyramid.ptr(), pyramid.step(), sigma_depth, nullptr));
}
void kfusion::cuda::computeNormalsAndMaskDepth(const Intr& intr, Depth& depth, Normals& normals)
{
    normals.create(depth.rows(), depth.cols());
    const float in[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_compute_normals_mask_depth(depth.ptr(), depth.step(), (float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(), in, nullptr));
}
void kfusion::cuda::computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals)
{
    points.create(depth.rows(), depth.cols());
    normals.create(depth.rows(), depth.cols());
    const float in[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_compute_point_normals(depth.ptr(), depth.step(), (float*)points.ptr(), points.step(), (float*)normals.ptr(), normals.step(),
                                        depth.cols(), depth.rows(), in, nullptr));
}
void kfusion::cuda::resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out)
{
    depth_out.create(depth.rows() / 2, depth.cols() / 2);
    normals_out.create(normals.rows() / 2, normals.cols() / 2);
    KF_DF(dfusion_resize_depth_normals(depth.ptr(), depth.step(), (const float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(),
                                       depth_out.ptr(), depth_out.step(), (float*)normals_out.ptr(), normals_out.step(), nullptr));
}
void kfusion::cuda::resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out)
{
    points_out.create(points.rows() / 2, points.cols() / 2);
    normals_out.create(normals.rows() / 2, normals.cols() / 2);
    KF_DF(dfusion_resize_points_normals((const float*)points.ptr(), points.step(), (const float*)normals.ptr(), normals.step(), points.cols(),
                                        points.rows(), (float*)points_out.ptr(), points_out.step(), (float*)normals_out.ptr(),
                                        normals_out.step(), nullptr));
}

// ------------------------------------------------------------------------------------------ views (imgproc.cpp:152-201)
void kfusion::cuda::renderImage(const Depth& depth, const Normals& normals, const Intr& intr, const Vec3f& light_pose, Image& image)
{
    image.create(depth.rows(), depth.cols());
    const float in[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
    KF_DF(dfusion_render_image_depth(depth.ptr(), depth.step(), (const float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(), in,
                                     light_pose.val, (unsigned char*)image.ptr(), image.step(), nullptr));
}
void kfusion::cuda::renderImage(const Cloud& points, const Normals& normals, const Intr& /*intr*/, const Vec3f& light_pose, Image& image)
{
    image.create(points.rows(), points.cols());
    KF_DF(dfusion_render_image_points((const float*)points.ptr(), points.step(), (const float*)normals.ptr(), normals.step(), points.cols(),
                                      points.rows(), light_pose.val, (unsigned char*)image.ptr(), image.step(), nullptr));
}
void kfusion::cuda::renderTangentColors(const Normals& normals, Image& image)
{
    image.create(normals.rows(), normals.cols());
    KF_DF(dfusion_render_tangent_colors((const float*)normals.ptr(), normals.step(), normals.cols(), normals.rows(), (unsigned char*)image.ptr(),
                                        image.step(), nullptr));
}

// ------------------------------------------------------------------------------------------ ProjectiveICP (projective_icp.cpp:64-213)
ProjectiveICP::ProjectiveICP() : angle_thres_(deg2rad(20.f)), dist_thres_(0.1f)
{
    const int iters[] = {10, 5, 4, 0};
    setIterationsNum(std::vector<int>(iters, iters + 4));
}
ProjectiveICP::~ProjectiveICP() {}
float ProjectiveICP::getDistThreshold() const { return dist_thres_; }
void ProjectiveICP::setDistThreshold(float distance) { dist_thres_ = distance; }
float ProjectiveICP::getAngleThreshold() const { return angle_thres_; }
void ProjectiveICP::setAngleThreshold(float angle) { angle_thres_ = angle; }
void ProjectiveICP::setIterationsNum(const std::vector<int>& iters)
{
    if (iters.size() >= MAX_PYRAMID_LEVELS) iters_.assign(iters.begin(), iters.begin() + MAX_PYRAMID_LEVELS);
    else { iters_ = std::vector<int>(MAX_PYRAMID_LEVELS, 0); std::copy(iters.begin(), iters.end(), iters_.begin()); }
}
int ProjectiveICP::getUsedLevelsNum() const
{
    int i = MAX_PYRAMID_LEVELS - 1;
    for (; i >= 0 && !iters_[i]; --i);
    return i + 1;
}

// A r = b for the symmetric 6x6 normal equations, and det(A): LU with partial pivoting in double.  (The reference calls
// cv::determinant and cv::solve(DECOMP_SVD) on float matrices, projective_icp.cpp:150,159 -- OpenCV is not part of the
// reference tree; any backward-stable solver agrees with it to ~1e-6 relative on these well-scaled systems.)
static bool solve6(const float A[36], const float b[6], double& det, float r[6])
{
    double M[6][7];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) M[i][j] = A[6 * i + j]; M[i][6] = b[i]; }
    det = 1.0;
    bool singular = false;
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int i = c + 1; i < 6; ++i) if (std::fabs(M[i][c]) > std::fabs(M[p][c])) p = i;
        if (M[p][c] == 0.0 || std::isnan(M[p][c])) { det = std::isnan(M[p][c]) ? M[p][c] : 0.0; singular = true; break; }
        if (p != c) { for (int j = 0; j < 7; ++j) std::swap(M[p][j], M[c][j]); det = -det; }
        det *= M[c][c];
        for (int i = c + 1; i < 6; ++i) {
            const double f = M[i][c] / M[c][c];
            for (int j = c; j < 7; ++j) M[i][j] -= f * M[c][j];
        }
    }
    if (singular) return false;
    for (int i = 5; i >= 0; --i) {
        double s = M[i][6];
        for (int j = i + 1; j < 6; ++j) s -= M[i][j] * (double)r[j];
        r[i] = (float)(s / M[i][i]);
    }
    return true;
}

bool ProjectiveICP::iterate(Affine3f& affine, const Intr& intr, const void* const* curr, const NormalsPyr& ncurr, const void* const* prev,
                            const NormalsPyr& nprev, const size_t* curr_step, const size_t* prev_step, bool depth_variant)
{
    const int LEVELS = getUsedLevelsNum();
    const float dist2_thres = dist_thres_ * dist_thres_;               // ComputeIcpHelper ctor, projective_icp.cpp:11-15
    const float min_cosine = std::cos(angle_thres_);
    affine = Affine3f::Identity();
    if (device_loop_ && LEVELS > 0) {
        // the whole loop as one enqueue: sums, 6x6 solve and pose update stay on the GPU (no per-iteration stream synchronise)
        DfIcpLevel lv[MAX_PYRAMID_LEVELS];
        for (int l = 0; l < LEVELS; ++l) {
            lv[l].curr = curr[l]; lv[l].curr_pitch = curr_step[l]; lv[l].ncurr = (const float*)ncurr[l].ptr(); lv[l].ncurr_pitch = ncurr[l].step();
            lv[l].prev = prev[l]; lv[l].prev_pitch = prev_step[l]; lv[l].nprev = (const float*)nprev[l].ptr(); lv[l].nprev_pitch = nprev[l].step();
            lv[l].cols = nprev[l].cols(); lv[l].rows = nprev[l].rows(); lv[l].iters = iters_[l];
        }
        const size_t need = (size_t)dfusion_icp_workspace_floats(lv[0].cols, lv[0].rows) + 27 + 16;
        if (buffer_.size() < need) buffer_.create(need);
        float* state_dev = buffer_.ptr() + (need - 16);
        const float in[4] = {intr.fx, intr.fy, intr.cx, intr.cy};
        KF_DF(dfusion_icp_estimate(lv, LEVELS, depth_variant ? 1 : 0, in, dist2_thres, min_cosine, buffer_.ptr(), state_dev, nullptr));
        float st[13];
        KF_HIP(hipMemcpy(st, state_dev, sizeof(st), hipMemcpyDeviceToHost));
        affine = aff12_to_affine(st);
        return st[12] != 0.f;
    }
    for (int level_index = LEVELS - 1; level_index >= 0; --level_index) {
        const Normals& n = nprev[level_index];
        const int rows = n.rows(), cols = n.cols();
        const int div = 1 << level_index;                              // setLevelIntr, :17-23
        const float li[4] = {intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div};
        const size_t need = (size_t)dfusion_icp_workspace_floats(cols, rows) + 27;
        if (buffer_.size() < need) buffer_.create(need);
        float* sums_dev = buffer_.ptr() + (need - 27);
        for (int iter = 0; iter < iters_[level_index]; ++iter) {
            float aff[12]; affine_to_aff12(affine, aff);
            if (depth_variant)
                KF_DF(dfusion_icp_sums_depth((const unsigned short*)curr[level_index], curr_step[level_index], (const float*)ncurr[level_index].ptr(),
                                             ncurr[level_index].step(), (const unsigned short*)prev[level_index], prev_step[level_index],
                                             (const float*)n.ptr(), n.step(), cols, rows, aff, li, dist2_thres, min_cosine, buffer_.ptr(), sums_dev,
                                             nullptr, nullptr));
            else
                KF_DF(dfusion_icp_sums_points((const float*)curr[level_index], curr_step[level_index], (const float*)ncurr[level_index].ptr(),
                                              ncurr[level_index].step(), (const float*)prev[level_index], prev_step[level_index],
                                              (const float*)n.ptr(), n.step(), cols, rows, aff, li, dist2_thres, min_cosine, buffer_.ptr(), sums_dev,
                                              nullptr, nullptr));
            float s[27];
            KF_HIP(hipMemcpy(s, sums_dev, sizeof(s), hipMemcpyDeviceToHost));   // cudaMemcpyAsync + cudaStreamSynchronize (:45)
            float A[36], b[6];
            int shift = 0;                                                  // StreamHelper::get, :43-61
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 7; ++j) {
                    const float value = s[shift++];
                    if (j == 6) b[i] = value; else A[j * 6 + i] = A[i * 6 + j] = value;
                }
            double det; float r[6];
            const bool solved = solve6(A, b, det, r);
            if (std::fabs(det) < 1e-15 || std::isnan(det) || !solved) {     // :152-156
                if (std::isnan(det)) std::printf("qnan\n");
                return false;
            }
            const Affine3f Tinc(Vec3f(r[0], r[1], r[2]), Vec3f(r[3], r[4], r[5]));   // :162
            affine = Tinc * affine;
        }
    }
    return true;
}

bool ProjectiveICP::estimateTransform(Affine3f& /*affine*/, const Intr& /*intr*/, const Frame& /*curr*/, const Frame& /*prev*/)   // projective_icp.cpp:110-123
{
    error("Not implemented", __FILE__, __LINE__, "ProjectiveICP::estimateTransform(Frame)");
    return false;
}
bool ProjectiveICP::estimateTransform(Affine3f& affine, const Intr& intr, const DepthPyr& dcurr, const NormalsPyr ncurr, const DepthPyr dprev,
                                      const NormalsPyr nprev)
{
    const void* c[MAX_PYRAMID_LEVELS] = {}; const void* p[MAX_PYRAMID_LEVELS] = {}; size_t cs[MAX_PYRAMID_LEVELS] = {}, ps[MAX_PYRAMID_LEVELS] = {};
    for (int i = 0; i < getUsedLevelsNum(); ++i) { c[i] = dcurr[i].ptr(); cs[i] = dcurr[i].step(); p[i] = dprev[i].ptr(); ps[i] = dprev[i].step(); }
    return iterate(affine, intr, c, ncurr, p, nprev, cs, ps, true);
}
bool ProjectiveICP::estimateTransform(Affine3f& affine, const Intr& intr, const PointsPyr& vcurr, const NormalsPyr ncurr, const PointsPyr vprev,
                                      const NormalsPyr nprev)
{
    const void* c[MAX_PYRAMID_LEVELS] = {}; const void* p[MAX_PYRAMID_LEVELS] = {}; size_t cs[MAX_PYRAMID_LEVELS] = {}, ps[MAX_PYRAMID_LEVELS] = {};
    for (int i = 0; i < getUsedLevelsNum(); ++i) { c[i] = vcurr[i].ptr(); cs[i] = vcurr[i].step(); p[i] = vprev[i].ptr(); ps[i] = vprev[i].step(); }
    return iterate(affine, intr, c, ncurr, p, nprev, cs, ps, false);
}

// ------------------------------------------------------------------------------------------ KinFu (kinfu.cpp)
KinFuParams KinFuParams::default_params_dynamicfusion()                // kinfu.cpp:15-50
{
    const int iters[] = {10, 5, 4, 0};
    KinFuParams p;
    p.cols = 640; p.rows = 480;
    p.intr = Intr(570.342f, 570.342f, 320.f, 240.f);
    p.volume_dims = Vec3i::all(256);
    p.volume_size = Vec3f::all(1.f);
    p.volume_pose = Affine3f().translate(Vec3f(-p.volume_size[0] / 2, -p.volume_size[1] / 2, 0.5f));
    p.bilateral_sigma_depth = 0.04f; p.bilateral_sigma_spatial = 4.5; p.bilateral_kernel_size = 7;
    p.icp_truncate_depth_dist = 0.f; p.icp_dist_thres = 0.1f; p.icp_angle_thres = deg2rad(30.f);
    p.icp_iter_num.assign(iters, iters + 4);
    p.tsdf_min_camera_movement = 0.f; p.tsdf_trunc_dist = 0.04f; p.tsdf_max_weight = 64;
    p.raycast_step_factor = 0.75f; p.gradient_delta_factor = 0.5f;
    p.light_pose = Vec3f::all(0.f);
    return p;
}
KinFuParams KinFuParams::default_params()                              // kinfu.cpp:55-89
{
    KinFuParams p = default_params_dynamicfusion();
    p.intr = Intr(525.f, 525.f, p.cols / 2 - 0.5f, p.rows / 2 - 0.5f);
    p.volume_dims = Vec3i::all(512);
    p.volume_size = Vec3f::all(3.f);
    p.volume_pose = Affine3f().translate(Vec3f(-p.volume_size[0] / 2, -p.volume_size[1] / 2, 0.5f));
    return p;
}

KinFu::KinFu(const KinFuParams& params) : frame_counter_(0), params_(params)     // kinfu.cpp:95-125
{
    if (params.volume_dims[0] % 32 != 0) kfusion::cuda::error("volume_dims[0] % 32 == 0", __FILE__, __LINE__, "KinFu");
    volume_.reset(new cuda::TsdfVolume(params_.volume_dims));
    warp_.reset(new WarpField());
    volume_->setTruncDist(params_.tsdf_trunc_dist);                     // NB the reference's order: trunc before size (clamp quirk kept)
    volume_->setMaxWeight(params_.tsdf_max_weight);
    volume_->setSize(params_.volume_size);
    volume_->setPose(params_.volume_pose);
    volume_->setRaycastStepFactor(params_.raycast_step_factor);
    volume_->setGradientDeltaFactor(params_.gradient_delta_factor);
    icp_.reset(new cuda::ProjectiveICP());
    icp_->setDistThreshold(params_.icp_dist_thres);
    icp_->setAngleThreshold(params_.icp_angle_thres);
    icp_->setIterationsNum(params_.icp_iter_num);
    allocate_buffers();
    reset();
}

void KinFu::allocate_buffers()                                         // kinfu.cpp:150-190
{
    const int LEVELS = cuda::ProjectiveICP::MAX_PYRAMID_LEVELS;
    int cols = params_.cols, rows = params_.rows;
    dists_.create(rows, cols);
    for (cuda::Frame* f : {&curr_, &prev_, &first_}) { f->depth_pyr.resize(LEVELS); f->normals_pyr.resize(LEVELS); f->points_pyr.resize(LEVELS); }
    for (int i = 0; i < LEVELS; ++i) {
        for (cuda::Frame* f : {&curr_, &prev_, &first_}) { f->depth_pyr[i].create(rows, cols); f->normals_pyr[i].create(rows, cols); f->points_pyr[i].create(rows, cols); }
        cols /= 2; rows /= 2;
    }
}

void KinFu::reset()                                                    // kinfu.cpp:195-206
{
    if (frame_counter_) std::printf("Reset\n");
    frame_counter_ = 0;
    poses_.clear();
    poses_.reserve(30000);
    poses_.push_back(Affine3f::Identity());
    volume_->clear();
}

void KinFu::optimiseWarp(std::vector<Vec3f>& canonical, std::vector<Vec3f>& canonical_normals, const std::vector<Vec3f>& live)
{
    if (params_.warp_solver_iterations <= 0) return;
    warp_->setSolverIterations(params_.warp_solver_iterations);
    warp_->energy_data(canonical, canonical_normals, live, canonical_normals);
}

// kinfu.cpp:312-343 / :408-436.  `image.create` of the render wrappers is a no-op on the user-pointer halves of the side-by-side view.
static void render_views(const KinFuParams& p, const cuda::Depth* depth, const cuda::Cloud* points, const cuda::Normals& normals,
                         cuda::Image& image, int flag)
{
    image.create(p.rows, flag != 3 ? p.cols : p.cols * 2);
    auto phong = [&](cuda::Image& dst) {
        if (depth) cuda::renderImage(*depth, normals, p.intr, p.light_pose, dst);
        else cuda::renderImage(*points, normals, p.intr, p.light_pose, dst);
    };
    if (flag < 1 || flag > 3) phong(image);
    else if (flag == 2) cuda::renderTangentColors(normals, image);
    else if (flag == 1) phong(image);       // (the reference's `flag < 1 || flag > 3` test sends 1 to the last branch; with flag == 1 the
                                            // image is only p.cols wide and its i2 half would be written out of bounds -- shaded view only)
    else {
        cuda::Image i1(p.rows, p.cols, image.ptr(), image.step());
        cuda::Image i2(p.rows, p.cols, image.ptr() + p.cols, image.step());
        phong(i1);
        cuda::renderTangentColors(normals, i2);
    }
}
void KinFu::renderImage(cuda::Image& image, int flag)
{
    if (params_.use_depth_pyramids) render_views(params_, &prev_.depth_pyr[0], nullptr, prev_.normals_pyr[0], image, flag);
    else render_views(params_, nullptr, &prev_.points_pyr[0], prev_.normals_pyr[0], image, flag);
}
void KinFu::renderImage(cuda::Image& image, const Affine3f& pose, int flag)
{
    const KinFuParams& p = params_;
    normals_.create(p.rows, p.cols);
    if (p.use_depth_pyramids) {
        depths_.create(p.rows, p.cols);
        volume_->raycast(pose, p.intr, depths_, normals_);
        render_views(p, &depths_, nullptr, normals_, image, flag);
    } else {
        points_.create(p.rows, p.cols);
        volume_->raycast(pose, p.intr, points_, normals_);
        render_views(p, nullptr, &points_, normals_, image, flag);
    }
}

Affine3f KinFu::getCameraPose(int time) const                          // kinfu.cpp:213-218
{
    if (time > (int)poses_.size() || time < 0) time = (int)poses_.size() - 1;
    return poses_[time];
}

bool KinFu::operator()(const cuda::Depth& depth, const cuda::Image& /*image*/)    // kinfu.cpp:220-304 (default build: points pyramids)
{
    const KinFuParams& p = params_;
    const int LEVELS = icp_->getUsedLevelsNum();
    cuda::computeDists(depth, dists_, p.intr);
    cuda::depthBilateralFilter(depth, curr_.depth_pyr[0], p.bilateral_kernel_size, p.bilateral_sigma_spatial, p.bilateral_sigma_depth);
    if (p.icp_truncate_depth_dist > 0) cuda::depthTruncation(curr_.depth_pyr[0], p.icp_truncate_depth_dist);
    for (int i = 1; i < LEVELS; ++i) cuda::depthBuildPyramid(curr_.depth_pyr[i - 1], curr_.depth_pyr[i], p.bilateral_sigma_depth);
    for (int i = 0; i < LEVELS; ++i) {
        if (p.use_depth_pyramids) cuda::computeNormalsAndMaskDepth(p.intr(i), curr_.depth_pyr[i], curr_.normals_pyr[i]);   // #if defined USE_DEPTH, :236-237
        else cuda::computePointNormals(p.intr(i), curr_.depth_pyr[i], curr_.points_pyr[i], curr_.normals_pyr[i]);          // :239
    }
    if (p.use_depth_pyramids)   // dynamicfusion() is handed curr_.points_pyr[0] in either build (:283): keep it valid
        cuda::computePointNormals(p.intr(0), curr_.depth_pyr[0], curr_.points_pyr[0], df_live_normals_);
    cuda::waitAllDefaultStream();

    if (frame_counter_ == 0) {                                          // :246-265
        volume_->integrate(dists_, poses_.back(), p.intr);
        volume_->compute_points();
        volume_->compute_normals();
        // warp_->init(cloud): the reference's cv::Mat overload keeps every 50th point (warp_field.cpp:49-60) and leaves the rest of
        // the node array zero-initialised; here only the kept points become nodes, with a larger stride if they would not fit
        // The extraction order depends on atomics (here and in the reference); the seed set must not, or every run would deform
        // differently: the cloud is put in (z, y, x) order before sampling.
        std::vector<Point> cloud = volume_->cloud_host_vector();
        std::sort(cloud.begin(), cloud.end(), [](const Point& a, const Point& b) {
            return a.z != b.z ? a.z < b.z : (a.y != b.y ? a.y < b.y : a.x < b.x); });
        size_t step = 50;
        const size_t max_nodes = (size_t)std::max(1, std::min(params_.max_warp_nodes, 65535));
        while (cloud.size() / step > max_nodes) ++step;
        std::vector<Vec3f> seeds;
        for (size_t i = 0; i < cloud.size(); i += step) seeds.push_back(Vec3f(cloud[i].x, cloud[i].y, cloud[i].z));
        if (seeds.size() >= (size_t)warp_->k()) warp_->init(seeds);
        if (p.use_depth_pyramids) { curr_.depth_pyr.swap(prev_.depth_pyr); curr_.depth_pyr.swap(first_.depth_pyr); }        // :254-256
        else { curr_.points_pyr.swap(prev_.points_pyr); curr_.points_pyr.swap(first_.points_pyr); }                         // :258-259
        curr_.normals_pyr.swap(prev_.normals_pyr);
        curr_.normals_pyr.swap(first_.normals_pyr);
        return ++frame_counter_, false;
    }

    Affine3f affine;                                                    // curr -> prev
    const bool ok = p.use_depth_pyramids
        ? icp_->estimateTransform(affine, p.intr, curr_.depth_pyr, curr_.normals_pyr, prev_.depth_pyr, prev_.normals_pyr)      // :272
        : icp_->estimateTransform(affine, p.intr, curr_.points_pyr, curr_.normals_pyr, prev_.points_pyr, prev_.normals_pyr);   // :274
    if (!ok) return reset(), false;
    poses_.push_back(poses_.back() * affine);                           // curr -> global
    cuda::Depth d = curr_.depth_pyr[0];
    dynamicfusion(d, curr_.points_pyr[0], curr_.normals_pyr[0]);

    if (p.use_depth_pyramids) {                                         // :292-295
        volume_->raycast(poses_.back(), p.intr, prev_.depth_pyr[0], prev_.normals_pyr[0]);
        for (int i = 1; i < LEVELS; ++i)
            cuda::resizeDepthNormals(prev_.depth_pyr[i - 1], prev_.normals_pyr[i - 1], prev_.depth_pyr[i], prev_.normals_pyr[i]);
    } else {                                                            // :296-299
        volume_->raycast(poses_.back(), p.intr, prev_.points_pyr[0], prev_.normals_pyr[0]);
        for (int i = 1; i < LEVELS; ++i)
            cuda::resizePointsNormals(prev_.points_pyr[i - 1], prev_.normals_pyr[i - 1], prev_.points_pyr[i], prev_.normals_pyr[i]);
    }
    cuda::waitAllDefaultStream();
    return ++frame_counter_, true;
}

void KinFu::dynamicfusion(cuda::Depth& depth, cuda::Cloud live_frame, cuda::Normals /*current_normals*/)   // kinfu.cpp:344-400
{
    const Affine3f camera_pose = poses_.back();
    if (warp_->nodeCount() < (size_t)warp_->k()) {                      // no usable warp field (empty first frame): plain KinFu fusion
        cuda::Dists dists; cuda::computeDists(depth, dists, params_.intr);
        volume_->integrate(dists, camera_pose, params_.intr);
        return;
    }
    const size_t n = (size_t)depth.rows() * depth.cols();
    warp_->ensureIndex(*volume_, params_.warped_fusion);                // brick lists: exact k-NN of the points below without scanning all nodes
    KF_DF(dfusion_warp_set_point_tiling(warp_->handle(), depth.cols()));  // the point sets below are images: 8 x 8 pixel tiles per wave
    if (params_.device_resident) {
        // kinfu.cpp:346-393 with every point set kept on the GPU: raycast -> canonical = inverse_pose * point (and the float4 ->
        // float3 repack) -> warp twice (as the reference does, :387 and :391) -> psdf / removal -> fusion
        df_cloud_.create(depth.rows(), depth.cols());
        df_normals_.create(depth.rows(), depth.cols());
        volume_->raycast(camera_pose, params_.intr, df_cloud_, df_normals_);
        if (df_points3_.size() < 3 * n) { df_points3_.create(3 * n); df_normals3_.create(3 * n); df_warped4_.create(n); }
        float inv12[12]; affine_to_aff12(camera_pose.inv(), inv12);
        KF_DF(dfusion_transform_points((const float*)df_cloud_.ptr(), df_cloud_.step(), 4, df_points3_.ptr(), (size_t)depth.cols() * 12, 3,
                                       depth.cols(), depth.rows(), inv12, nullptr));
        KF_DF(dfusion_transform_points((const float*)df_normals_.ptr(), df_normals_.step(), 4, df_normals3_.ptr(), (size_t)depth.cols() * 12, 3,
                                       depth.cols(), depth.rows(), nullptr, nullptr));
        warp_->warp(df_points3_, df_normals3_, (int)n);                      // :387
        if (params_.warp_solver_iterations > 0) {                           // :389 optimiser_->optimiseWarpData(canonical, normals, live, normals)
            if (df_live3_.size() < 3 * n) df_live3_.create(3 * n);
            KF_DF(dfusion_transform_points((const float*)live_frame.ptr(), live_frame.step(), 4, df_live3_.ptr(), (size_t)depth.cols() * 12, 3,
                                           depth.cols(), depth.rows(), nullptr, nullptr));
            warp_->setSolverIterations(params_.warp_solver_iterations);
            warp_->energy_data(df_points3_, df_live3_, (int)n);
        }
        warp_->warp(df_points3_, df_normals3_, (int)n);                      // :391
        if (params_.warped_fusion) {
            cuda::Dists dists; cuda::computeDists(depth, dists, params_.intr);
            volume_->integrate(dists, camera_pose, params_.intr, *warp_);
        } else {
            KF_DF(dfusion_transform_points(df_points3_.ptr(), 3 * n * sizeof(float), 3, (float*)df_warped4_.ptr(), n * sizeof(Point), 4, (int)n, 1,
                                           nullptr, nullptr));
            cuda::DeviceArray<Point> warped(df_warped4_.ptr(), n);
            volume_->surface_fusion(*warp_, warped, depth, camera_pose, params_.intr);
        }
        volume_->compute_points();
        volume_->compute_normals();
        return;
    }
    cuda::Cloud cloud; cuda::Normals normals;
    cloud.create(depth.rows(), depth.cols());
    normals.create(depth.rows(), depth.cols());
    volume_->raycast(camera_pose, params_.intr, cloud, normals);
    std::vector<Point> cloud_host(n), normal_host(n), live_host(n);
    cloud.download(cloud_host.data(), (size_t)depth.cols() * sizeof(Point));
    normals.download(normal_host.data(), (size_t)depth.cols() * sizeof(Point));
    live_frame.download(live_host.data(), (size_t)depth.cols() * sizeof(Point));
    const Affine3f inverse_pose = camera_pose.inv();
    std::vector<Vec3f> canonical(n), canonical_normals(n), live(n);
    for (size_t i = 0; i < n; ++i) {
        canonical[i] = inverse_pose * Vec3f(cloud_host[i].x, cloud_host[i].y, cloud_host[i].z);
        canonical_normals[i] = Vec3f(normal_host[i].x, normal_host[i].y, normal_host[i].z);
        live[i] = Vec3f(live_host[i].x, live_host[i].y, live_host[i].z);
    }
    std::vector<Vec3f> canonical_visible(canonical);
    warp_->warp(canonical, canonical_normals);                          // :387
    optimiseWarp(canonical, canonical_normals, live);                   // :389
    warp_->warp(canonical, canonical_normals);                          // :391
    if (params_.warped_fusion) {
        cuda::Dists dists; cuda::computeDists(depth, dists, params_.intr);
        volume_->integrate(dists, camera_pose, params_.intr, *warp_);
    } else {
        volume_->surface_fusion(*warp_, canonical, canonical_visible, depth, camera_pose, params_.intr);   // :393
    }
    volume_->compute_points();                                          // :398-399
    volume_->compute_normals();
}
