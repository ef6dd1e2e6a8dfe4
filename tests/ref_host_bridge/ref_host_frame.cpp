// ref_host_frame.cpp -- TEST DRIVER: the hot-path call sequence of KinFu::operator() (kinfu.cpp:226,248,297,301,398-399) made on the
// REFERENCE'S OWN kfusion::cuda::TsdfVolume / cuda::computeDists (compiled unmodified from /root/reference/kfusion/src, see build_ref_host.py),
// whose kfusion::device::* calls hip_bridge.cpp forwards to libdfusion_hip.so.  Same raw-file protocol as
// dynamicfusion_amd/host/apps/headless_frame.cpp (rigid part), so tests/test_gpu_ref_host_bridge.py reads both the same way:
//   ref_host_frame <dims> <size_m> <cols> <rows> <frames> <in.bin> <out.bin>
// in.bin : volume pose f32[12], intrinsics f32[4], per frame { depth u16[rows*cols], camera pose f32[12] }
// out.bin: volume u32[dims^3], points f32[rows*cols*4], normals f32[rows*cols*4] (Points cast of the last frame), depth u16[rows*cols] +
//          normals f32[rows*cols*4] (Depth cast of the last frame), count u64, cloud f32[count*4], normals f32[count*4]; then the volume
//          again after TsdfVolume::clear() u32[dims^3].
// Round 6 -- every forward of hip_bridge.cpp, through the reference's own wrappers (kfusion/src/imgproc.cpp, TsdfVolume::psdf): when in.bin
// ends with a second set of intrinsics f32[4] (non-square: both axes of the bridge's focal_of()), the driver goes on, on the LAST frame's raw
// depth, with cuda::depthBilateralFilter, depthTruncation, depthBuildPyramid, computeNormalsAndMaskDepth, computePointNormals,
// resizeDepthNormals, resizePointsNormals, renderImage x 2, renderTangentColors, TsdfVolume::psdf (-> device::project_and_remove(const&)) on
// the last Points cast, and the one forward no reference code reaches -- device::project_and_remove(PtrStepSz<ushort>&), internal.hpp:108 --
// called directly.  out.bin continues (W = cols, H = rows, h = half sizes):
//   bilateral u16[H*W], truncated u16[H*W], pyramid u16[h*w], masked depth u16[H*W] + normals f32[H*W*4], points f32[H*W*4] + normals f32[H*W*4]
//   (computePointNormals), resized depth u16[h*w] + normals f32[h*w*4], resized points f32[h*w*4] + normals f32[h*w*4], three images u8[H*W*4],
//   psdf distances f32[H*W] + dists after u16[H*W], direct call: points f32[H*W*4] + dists after u16[H*W], cloudToDepth of the points u16[H*W].
// The last stdout line reports which forwards ran (hip_bridge_forward_report).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <kfusion/cuda/tsdf_volume.hpp>        // the reference's headers (-I /root/reference/kfusion/include)
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/warp_field.hpp>
#include "internal.hpp"                        // the reference's private header (-I /root/reference/kfusion/src): device::project_and_remove

extern "C" int hip_bridge_forward_report(char* buf, int cap);

using namespace kfusion;

// TsdfVolume::surface_fusion (tsdf_volume.cpp:228-255) names two WarpField members; kfusion/src/warp_field.cpp, which defines them, needs
// Ceres and is not compiled here, and this driver never calls surface_fusion.
void kfusion::WarpField::KNN(Vec3f) const { std::fprintf(stderr, "WarpField::KNN: warp_field.cpp is not part of this build\n"); std::abort(); }
std::vector<float>* kfusion::WarpField::getDistSquared() const { std::abort(); }

static Affine3f read_affine(FILE* f)
{
    float a[12];
    if (std::fread(a, 4, 12, f) != 12) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    Affine3f r;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) r.R(i, j) = a[3 * i + j]; r.t[i] = a[9 + i]; }
    return r;
}

int main(int argc, char** argv)
{
    if (argc != 8) { std::fprintf(stderr, "usage: %s dims size cols rows frames in.bin out.bin\n", argv[0]); return 2; }
    const int dims = std::atoi(argv[1]); const float size = (float)std::atof(argv[2]);
    const int cols = std::atoi(argv[3]), rows = std::atoi(argv[4]), frames = std::atoi(argv[5]);
    FILE* in = std::fopen(argv[6], "rb");
    if (!in) { std::perror("in"); return 2; }
    const Affine3f volume_pose = read_affine(in);
    float iv[4];
    if (std::fread(iv, 4, 4, in) != 4) return 2;
    const Intr intr(iv[0], iv[1], iv[2], iv[3]);

    cuda::TsdfVolume volume(Vec3i(dims, dims, dims));        // the reference's constructor: create() -> clear() -> device::clear_volume
    volume.setSize(Vec3f::all(size));                        // KinFu::KinFu, kinfu.cpp:99-107
    volume.setTruncDist(0.04f);
    volume.setMaxWeight(64);
    volume.setPose(volume_pose);
    volume.setRaycastStepFactor(0.75f);
    volume.setGradientDeltaFactor(0.5f);

    std::vector<unsigned short> depth((size_t)rows * cols);
    cuda::Depth depth_device;
    cuda::Dists dists;
    cuda::Cloud points; cuda::Normals normals;
    points.create(rows, cols); normals.create(rows, cols);
    Affine3f cam;
    for (int f = 0; f < frames; ++f) {
        if (std::fread(depth.data(), 2, depth.size(), in) != depth.size()) return 2;
        cam = read_affine(in);
        depth_device.upload(depth.data(), (size_t)cols * 2, rows, cols);             // demo.cpp:89
        cuda::computeDists(depth_device, dists, intr);                               // kinfu.cpp:226 (the reference's imgproc.cpp:87)
        volume.integrate(dists, cam, intr);                                          // kinfu.cpp:248 (tsdf_volume.cpp:110-122)
        volume.raycast(cam, intr, points, normals);                                  // kinfu.cpp:297 (tsdf_volume.cpp:154-174)
        cuda::waitAllDefaultStream();                                                // kinfu.cpp:301
    }
    float iv2[4];
    const bool have_intr2 = std::fread(iv2, 4, 4, in) == 4;                          // (round 6: the front-end section's own intrinsics)
    std::fclose(in);

    FILE* out = std::fopen(argv[7], "wb");
    if (!out) { std::perror("out"); return 2; }
    std::vector<unsigned int> vol((size_t)dims * dims * dims);
    volume.data().download(vol.data());
    std::fwrite(vol.data(), 4, vol.size(), out);
    std::vector<float> p((size_t)rows * cols * 4), n(p.size());
    points.download(p.data(), (size_t)cols * 16);
    normals.download(n.data(), (size_t)cols * 16);
    std::fwrite(p.data(), 4, p.size(), out);
    std::fwrite(n.data(), 4, n.size(), out);
    cuda::Depth cast_depth; cast_depth.create(rows, cols);                           // the Depth variant (tsdf_volume.cpp:131-146; kinfu.cpp:421)
    volume.raycast(cam, intr, cast_depth, normals);
    cuda::waitAllDefaultStream();
    std::vector<unsigned short> cd((size_t)rows * cols);
    cast_depth.download(cd.data(), (size_t)cols * 2);
    normals.download(n.data(), (size_t)cols * 16);
    std::fwrite(cd.data(), 2, cd.size(), out);
    std::fwrite(n.data(), 4, n.size(), out);
    volume.compute_points();                                                         // kinfu.cpp:398 (tsdf_volume.cpp:312-317 -> fetchCloud -> device::extractCloud)
    volume.compute_normals();                                                        // kinfu.cpp:399 (-> fetchNormals -> device::extractNormals)
    const cv::Mat cloud = volume.get_cloud_host(), cnrm = volume.get_normal_host();
    const unsigned long long cnt = (unsigned long long)cloud.cols;
    std::fwrite(&cnt, 8, 1, out);
    if (cnt) {
        std::fwrite(cloud.ptr<Point>(), 16, cnt, out);
        std::fwrite(cnrm.ptr<Normal>(), 16, cnt, out);
    }
    volume.clear();                                                                  // tsdf_volume.cpp:89-102
    volume.data().download(vol.data());
    std::fwrite(vol.data(), 4, vol.size(), out);
    if (have_intr2) {
        const Intr intr2(iv2[0], iv2[1], iv2[2], iv2[3]);
        const int hr = rows / 2, hc = cols / 2;
        auto put_u16 = [&](const cuda::Depth& d) { std::vector<unsigned short> h((size_t)d.rows() * d.cols()); d.download(h.data(), (size_t)d.cols() * 2); std::fwrite(h.data(), 2, h.size(), out); };
        auto put_f4 = [&](const cuda::DeviceArray2D<Point>& a) { std::vector<float> h((size_t)a.rows() * a.cols() * 4); a.download(h.data(), (size_t)a.cols() * 16); std::fwrite(h.data(), 4, h.size(), out); };
        auto put_n4 = [&](const cuda::Normals& a) { std::vector<float> h((size_t)a.rows() * a.cols() * 4); a.download(h.data(), (size_t)a.cols() * 16); std::fwrite(h.data(), 4, h.size(), out); };
        auto put_img = [&](const cuda::Image& a) { std::vector<unsigned char> h((size_t)a.rows() * a.cols() * 4); a.download(h.data(), (size_t)a.cols() * 4); std::fwrite(h.data(), 1, h.size(), out); };
        cuda::Depth bil, tr, pyr, md, dh;
        cuda::Normals mn, pn, nh, pnh;
        cuda::Cloud pc, ph;
        cuda::depthBilateralFilter(depth_device, bil, 7, 4.5f, 0.04f);               // kinfu.cpp:192 with the default parameters (:30-32)
        put_u16(bil);
        bil.copyTo(tr);
        cuda::depthTruncation(tr, 1.2f);                                             // kinfu.cpp:195 (a threshold that cuts the back plane)
        put_u16(tr);
        cuda::depthBuildPyramid(bil, pyr, 0.04f);                                    // kinfu.cpp:201
        put_u16(pyr);
        bil.copyTo(md);
        cuda::computeNormalsAndMaskDepth(intr2, md, mn);                             // kinfu.cpp:198 (USE_DEPTH)
        put_u16(md); put_n4(mn);
        cuda::computePointNormals(intr2, bil, pc, pn);                               // kinfu.cpp:203
        put_f4(pc); put_n4(pn);
        cuda::resizeDepthNormals(md, mn, dh, nh);                                    // kinfu.cpp:437
        put_u16(dh); put_n4(nh);
        cuda::resizePointsNormals(pc, pn, ph, pnh);                                  // kinfu.cpp:306
        put_f4(ph); put_n4(pnh);
        cuda::Image im1, im2, im3;
        const Vec3f light(0.3f, -0.2f, -0.5f);
        cuda::renderImage(md, mn, intr2, light, im1);                                // kinfu.cpp:424 (Depth variant)
        cuda::renderImage(pc, pn, intr2, light, im2);                                // kinfu.cpp:324 (Cloud variant)
        cuda::renderTangentColors(pn, im3);                                          // kinfu.cpp:329
        put_img(im1); put_img(im2); put_img(im3);
        // TsdfVolume::psdf (tsdf_volume.cpp:266-292) on the last Points cast as the "warped" points: projects them, reads and removes
        // the dists they explain (device::project_and_remove(const PtrStepSz<ushort>&, ...)), returns the host-side differences
        std::vector<Vec3f> warped((size_t)rows * cols);
        for (size_t i = 0; i < warped.size(); ++i) warped[i] = Vec3f(p[4 * i], p[4 * i + 1], p[4 * i + 2]);
        cuda::Dists d2;
        cuda::computeDists(depth_device, d2, intr2);
        const std::vector<float> ro = volume.psdf(warped, d2, intr2);
        cuda::waitAllDefaultStream();
        std::fwrite(ro.data(), 4, ro.size(), out);
        put_u16(d2);
        // ... and the non-const overload (internal.hpp:108): declared, bridged, called by nothing in the reference -- called here
        cuda::Dists d3;
        cuda::computeDists(depth_device, d3, intr2);
        cuda::Cloud p4;
        p4.upload(p.data(), (size_t)cols * 16, rows, cols);
        device::PtrStepSz<ushort> view = d3;
        device::Projector proj(intr2.fx, intr2.fy, intr2.cx, intr2.cy);
        device::project_and_remove(view, (device::Points&)p4, proj);
        cuda::waitAllDefaultStream();
        put_f4(p4); put_u16(d3);
        cuda::Depth c2d;
        cuda::cloudToDepth(pc, c2d);                                                 // imgproc.cpp:98-103 -> device::cloud_to_depth (no caller in the reference)
        cuda::waitAllDefaultStream();
        put_u16(c2d);
        (void)hr; (void)hc;
    }
    std::fclose(out);
    char rep[1024];
    const int ran = hip_bridge_forward_report(rep, (int)sizeof(rep));
    std::printf("hip_bridge forwards %s\n", rep);
    (void)ran;
    std::printf("ref_host_frame ok: the reference's TsdfVolume, %d frames\n", frames);
    return 0;
}
