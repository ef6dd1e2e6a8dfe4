// kinfu_headless.cpp -- apps/demo.cpp (/root/reference/apps/demo.cpp:60-110) without OpenNI capture and the viz window: feeds depth
// frames from a file to kfusion::KinFu::operator() and records what the demo would display -- the camera pose per frame.
//   kinfu_headless <cols> <rows> <frames> <dims> <size_m> <in.bin> <out.bin> [warped|host|warped-host]
//   (warped: per-voxel warped integrate instead of surface_fusion; host: the reference's host-staged data flow; nosolver: skip the
//   warp data-term solve; depth: the reference's USE_DEPTH build -- depth pyramids and masked-depth ICP)
// in.bin : intrinsics fx fy cx cy f32[4], then per frame depth u16[rows*cols] (mm).
// out.bin: per frame { tracked i32 (operator()'s return value), pose f32[12] (R row-major, t) }, then the extracted surface
//          count u64 and the volume u32[dims^3].
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include <kfusion/kinfu.hpp>
#include <kfusion/cuda/imgproc.hpp>

using namespace kfusion;

int main(int argc, char** argv)
{
    if (argc != 8 && argc != 9) { std::fprintf(stderr, "usage: %s cols rows frames dims size in.bin out.bin [warped|host|nosolver|depth, '-' separated]\n", argv[0]); return 2; }
    const int cols = std::atoi(argv[1]), rows = std::atoi(argv[2]), frames = std::atoi(argv[3]), dims = std::atoi(argv[4]);
    const float size = (float)std::atof(argv[5]);
    FILE* in = std::fopen(argv[6], "rb");
    if (!in) { std::perror("in"); return 2; }
    float iv[4];
    if (std::fread(iv, 4, 4, in) != 4) return 2;

    KinFuParams p = KinFuParams::default_params_dynamicfusion();        // demo.cpp:120
    p.cols = cols; p.rows = rows;
    p.intr = Intr(iv[0], iv[1], iv[2], iv[3]);
    p.volume_dims = Vec3i::all(dims);
    p.volume_size = Vec3f::all(size);
    p.volume_pose = Affine3f().translate(Vec3f(-size / 2, -size / 2, 0.5f));
    const std::string mode = argc == 9 ? argv[8] : "";
    p.warped_fusion = mode.find("warped") != std::string::npos;
    p.device_resident = mode.find("host") == std::string::npos;
    if (mode.find("nosolver") != std::string::npos) p.warp_solver_iterations = 0;
    p.use_depth_pyramids = mode.find("depth") != std::string::npos;
    KinFu kinfu(p);

    FILE* out = std::fopen(argv[7], "wb");
    if (!out) { std::perror("out"); return 2; }
    std::vector<unsigned short> depth((size_t)rows * cols);
    cuda::Depth depth_device;
    double total_ms = 0.0; int timed = 0;
    for (int f = 0; f < frames; ++f) {
        if (std::fread(depth.data(), 2, depth.size(), in) != depth.size()) return 2;
        depth_device.upload(depth.data(), (size_t)cols * 2, rows, cols);    // demo.cpp:89
        const auto t0 = std::chrono::steady_clock::now();
        const int tracked = kinfu(depth_device) ? 1 : 0;                    // demo.cpp:93
        cuda::waitAllDefaultStream();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (f >= 2) { total_ms += ms; ++timed; }
        float pose[12]; affine_to_aff12(kinfu.getCameraPose(), pose);
        if (mode.find("trace") != std::string::npos) {                      // per-frame checksums (debugging aid): volume, node transforms
            std::vector<unsigned int> v((size_t)dims * dims * dims);
            kinfu.tsdf().data().download(v.data());
            unsigned long long hv = 1469598103934665603ull, hn = hv;
            for (unsigned int w : v) { hv ^= w; hv *= 1099511628211ull; }
            for (const auto& nd : *kinfu.getWarp().getNodes()) {
                const unsigned int* r = (const unsigned int*)nd.transform.raw();
                for (int i = 0; i < 8; ++i) { hn ^= r[i]; hn *= 1099511628211ull; }
            }
            std::fprintf(stderr, "trace frame %d volume %016llx nodes %016llx pose %08x\n", f, hv, hn, *(unsigned int*)&pose[9]);
        }
        std::fwrite(&tracked, 4, 1, out);
        std::fwrite(pose, 4, 12, out);
    }
    std::fclose(in);
    kinfu.tsdf().compute_points();
    const unsigned long long cnt = kinfu.tsdf().get_cloud_host().size();
    std::fwrite(&cnt, 8, 1, out);
    std::vector<unsigned int> vol((size_t)dims * dims * dims);
    kinfu.tsdf().data().download(vol.data());
    std::fwrite(vol.data(), 4, vol.size(), out);
    std::fclose(out);
    std::printf("kinfu_headless ok: %d frames, %zu warp nodes, %llu surface points, %.3f ms/frame (KinFu::operator(), frames 2.., wall clock)\n",
                frames, kinfu.getWarp().getNodes()->size(), cnt, timed ? total_ms / timed : 0.0);
    return 0;
}
