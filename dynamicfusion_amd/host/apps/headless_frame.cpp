// headless_frame.cpp -- the hot-path call sequence of KinFu::operator() / KinFu::dynamicfusion
// (/root/reference/kfusion/src/kinfu.cpp:226,248,297,351,385,391) through the source-compatible C++ API, without the GUI,
// ICP or solver.  Inputs and outputs are raw binary files so that tests can drive it and diff against the oracle:
//   headless_frame <dims> <size_m> <cols> <rows> <frames> <nodes> <k> <in.bin> <out.bin> [surface_fusion]
// in.bin : per frame { depth u16[rows*cols], camera pose f32[12] (R row-major, t) } , then nodes { pos f32[3M], dq f32[8M]
//          per frame, sigma f32[M] } ; intrinsics fx fy cx cy f32[4] ; volume pose f32[12] first of all.
// out.bin: volume u32[dims^3], points f32[rows*cols*4], normals f32[rows*cols*4] of the last frame, then the extracted
//          surface (kinfu.cpp:398-399 compute_points / compute_normals): count u64, cloud f32[count*4], normals f32[count*4].
//          With the trailing `surface_fusion` argument the body of KinFu::dynamicfusion (kinfu.cpp:344-391, minus the solver)
//          then runs on the last frame: ray-cast points -> canonical -> WarpField::warp -> TsdfVolume::surface_fusion, and
//          out.bin continues with: warped f32[rows*cols*3], depth after removal u16[rows*cols], volume u32[dims^3].
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/warp_field.hpp>

using namespace kfusion;

static Affine3f read_affine(FILE* f)
{
    float a[12];
    if (std::fread(a, 4, 12, f) != 12) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    return aff12_to_affine(a);
}

// `headless_frame bench <dims> <size_m> <cols> <rows> <frames> <nodes> <k> <prime> <warmup> <in.bin>` (round 6): the
// HEADLINE frame -- set_transforms + computeDists + TsdfVolume::integrate(..., warp) + raycast(Points) -- timed through the C++ mirror
// with every input resident on the device before the clock starts, exactly the sequence bench.py times through the Python mirror:
// `prime` untimed frames (first tables / models), the volume cleared, `warmup` untimed frames, then frames - prime - warmup timed ones
// between two device synchronises, one new pose per frame.  in.bin as above (depth + pose per frame, then node positions, per-frame
// transforms, dg_w).  Prints "cxx_host_ms_per_frame <ms> over <n> frames"; with an out.bin, writes the final volume and the last frame's
// points and normals behind it.
static int bench_mode(int argc, char** argv)
{
    if (argc != 12 && argc != 13) { std::fprintf(stderr, "usage: %s bench dims size cols rows frames nodes k prime warmup in.bin [out.bin]\n", argv[0]); return 2; }
    const int dims = std::atoi(argv[2]); const float size = (float)std::atof(argv[3]);
    const int cols = std::atoi(argv[4]), rows = std::atoi(argv[5]), frames = std::atoi(argv[6]), M = std::atoi(argv[7]), k = std::atoi(argv[8]);
    const int prime = std::atoi(argv[9]), warmup = std::atoi(argv[10]);
    FILE* in = std::fopen(argv[11], "rb");
    if (!in || M <= 0 || frames <= prime + warmup) { std::fprintf(stderr, "bench: bad arguments\n"); return 2; }
    const Affine3f volume_pose = read_affine(in);
    float intr_v[4];
    if (std::fread(intr_v, 4, 4, in) != 4) return 2;
    const Intr intr(intr_v[0], intr_v[1], intr_v[2], intr_v[3]);
    cuda::TsdfVolume volume(Vec3i(dims, dims, dims));
    volume.setSize(Vec3f::all(size)); volume.setTruncDist(0.04f); volume.setMaxWeight(64); volume.setPose(volume_pose);
    volume.setRaycastStepFactor(0.75f); volume.setGradientDeltaFactor(0.5f);
    std::vector<cuda::Depth> depth_dev(frames);
    std::vector<Affine3f> cam(frames);
    {
        std::vector<unsigned short> d((size_t)rows * cols);
        for (int f = 0; f < frames; ++f) {
            if (std::fread(d.data(), 2, d.size(), in) != d.size()) return 2;
            cam[f] = read_affine(in);
            depth_dev[f].upload(d.data(), (size_t)cols * 2, rows, cols);
        }
    }
    WarpField warp(k);
    std::vector<cuda::DeviceArray<float> > dq_dev(frames);
    {
        std::vector<float> pos((size_t)M * 3), sigma(M), dq((size_t)M * 8);
        if (std::fread(pos.data(), 4, pos.size(), in) != pos.size()) return 2;
        for (int f = 0; f < frames; ++f) { if (std::fread(dq.data(), 4, dq.size(), in) != dq.size()) return 2; dq_dev[f].upload(dq); }
        if (std::fread(sigma.data(), 4, sigma.size(), in) != sigma.size()) return 2;
        std::vector<Vec3f> pts(M);
        for (int i = 0; i < M; ++i) pts[i] = Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        warp.init(pts);
        for (int i = 0; i < M; ++i) (*warp.getNodes())[i].weight = sigma[i];
        warp.commit(true);
    }
    std::fclose(in);
    cuda::Dists dists;
    cuda::Cloud points; cuda::Normals normals;
    points.create(rows, cols); normals.create(rows, cols);
    auto frame = [&](int f) {
        warp.setTransformsDevice(dq_dev[f]);                                         // the solver's output, already on the device
        cuda::computeDists(depth_dev[f], dists, intr);                               // kinfu.cpp:226
        volume.integrateAsync(dists, cam[f], intr, warp);                            // the north-star fusion
        volume.raycast(cam[f], intr, points, normals);                               // kinfu.cpp:297
    };
    for (int f = 0; f < prime; ++f) frame(f);
    volume.clear();
    for (int f = prime; f < prime + warmup; ++f) frame(f);
    cuda::waitAllDefaultStream();
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = prime + warmup; f < frames; ++f) frame(f);
    cuda::waitAllDefaultStream();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const int n = frames - prime - warmup;
    std::printf("cxx_host_ms_per_frame %.6f over %d frames (%d nodes, k = %d, %d^3)\n", ms / n, n, M, k, dims);
    if (argc == 13) {
        // what the asynchronous sequence left behind, for a parity check (tests/test_gpu_cxx_host.py): the volume, then the last frame's
        // points and normals
        FILE* out = std::fopen(argv[12], "wb");
        if (!out) { std::perror("out"); return 2; }
        std::vector<unsigned int> vol((size_t)dims * dims * dims);
        volume.data().download(vol.data());
        std::fwrite(vol.data(), 4, vol.size(), out);
        std::vector<float> p((size_t)rows * cols * 4), nn((size_t)rows * cols * 4);
        points.download(p.data(), (size_t)cols * 16);
        normals.download(nn.data(), (size_t)cols * 16);
        std::fwrite(p.data(), 4, p.size(), out);
        std::fwrite(nn.data(), 4, nn.size(), out);
        std::fclose(out);
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && !std::strcmp(argv[1], "bench")) return bench_mode(argc, argv);
    if (argc != 10 && argc != 11) { std::fprintf(stderr, "usage: %s dims size cols rows frames nodes k in.bin out.bin\n", argv[0]); return 2; }
    const int dims = std::atoi(argv[1]); const float size = (float)std::atof(argv[2]);
    const int cols = std::atoi(argv[3]), rows = std::atoi(argv[4]), frames = std::atoi(argv[5]), M = std::atoi(argv[6]), k = std::atoi(argv[7]);
    FILE* in = std::fopen(argv[8], "rb");
    if (!in) { std::perror("in"); return 2; }
    const Affine3f volume_pose = read_affine(in);
    float intr_v[4];
    if (std::fread(intr_v, 4, 4, in) != 4) return 2;
    const Intr intr(intr_v[0], intr_v[1], intr_v[2], intr_v[3]);

    cuda::TsdfVolume volume(Vec3i(dims, dims, dims));        // KinFu::KinFu, kinfu.cpp:99-107 (size before trunc, see tests)
    volume.setSize(Vec3f::all(size));
    volume.setTruncDist(0.04f);
    volume.setMaxWeight(64);
    volume.setPose(volume_pose);
    volume.setRaycastStepFactor(0.75f);
    volume.setGradientDeltaFactor(0.5f);

    std::vector<std::vector<unsigned short> > depth(frames, std::vector<unsigned short>((size_t)rows * cols));
    std::vector<Affine3f> cam(frames);
    for (int f = 0; f < frames; ++f) {
        if (std::fread(depth[f].data(), 2, depth[f].size(), in) != depth[f].size()) return 2;
        cam[f] = read_affine(in);
    }
    WarpField warp(k);
    std::vector<std::vector<float> > dq(frames, std::vector<float>((size_t)M * 8));
    if (M > 0) {
        std::vector<float> pos((size_t)M * 3), sigma(M);
        if (std::fread(pos.data(), 4, pos.size(), in) != pos.size()) return 2;
        for (int f = 0; f < frames; ++f) if (std::fread(dq[f].data(), 4, dq[f].size(), in) != dq[f].size()) return 2;
        if (std::fread(sigma.data(), 4, sigma.size(), in) != sigma.size()) return 2;
        std::vector<Vec3f> pts(M);
        for (int i = 0; i < M; ++i) pts[i] = Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        warp.init(pts);
        for (int i = 0; i < M; ++i) (*warp.getNodes())[i].weight = sigma[i];
        warp.commit(true);
    }
    std::fclose(in);

    cuda::Depth depth_device;
    cuda::Dists dists;
    cuda::Cloud points; cuda::Normals normals;
    points.create(rows, cols); normals.create(rows, cols);
    for (int f = 0; f < frames; ++f) {
        depth_device.upload(depth[f].data(), (size_t)cols * 2, rows, cols);          // demo.cpp:89
        cuda::computeDists(depth_device, dists, intr);                               // kinfu.cpp:226
        if (M > 0) {
            for (int i = 0; i < M; ++i) std::memcpy((void*)(*warp.getNodes())[i].transform.raw(), &dq[f][8 * (size_t)i], 32);
            warp.commit(false);                                                      // solver write-back stand-in (kinfu.cpp:387)
            volume.integrate(dists, cam[f], intr, warp);                             // the north-star fusion (kinfu.cpp:391)
        } else {
            volume.integrate(dists, cam[f], intr);                                   // kinfu.cpp:248
        }
        volume.raycast(cam[f], intr, points, normals);                               // kinfu.cpp:297 / :351
        cuda::waitAllDefaultStream();                                                // kinfu.cpp:301
    }

    FILE* out = std::fopen(argv[9], "wb");
    if (!out) { std::perror("out"); return 2; }
    std::vector<unsigned int> vol((size_t)dims * dims * dims);
    volume.data().download(vol.data());
    std::fwrite(vol.data(), 4, vol.size(), out);
    std::vector<float> p((size_t)rows * cols * 4), n(p.size());
    points.download(p.data(), (size_t)cols * 16);
    normals.download(n.data(), (size_t)cols * 16);
    std::fwrite(p.data(), 4, p.size(), out);
    std::fwrite(n.data(), 4, n.size(), out);
    volume.compute_points();                                                         // kinfu.cpp:398 (and :249 on frame 0)
    volume.compute_normals();                                                        // kinfu.cpp:399
    const unsigned long long cnt = volume.get_cloud_host().size();
    std::fwrite(&cnt, 8, 1, out);
    if (cnt) {
        std::fwrite(volume.get_cloud_host().data(), 16, cnt, out);
        std::fwrite(volume.get_normal_host().data(), 16, cnt, out);
    }
    if (argc == 11 && M > 0) {
        const Affine3f camera_pose = cam[frames - 1];
        const Affine3f inverse_pose = camera_pose.inv();                             // kinfu.cpp:357
        std::vector<Vec3f> canonical((size_t)rows * cols), canonical_normals((size_t)rows * cols);
        for (size_t i = 0; i < canonical.size(); ++i) {
            canonical[i] = inverse_pose * Vec3f(p[4 * i], p[4 * i + 1], p[4 * i + 2]);   // :358-363
            canonical_normals[i] = Vec3f(n[4 * i], n[4 * i + 1], n[4 * i + 2]);          // :378-383
        }
        std::vector<Vec3f> canonical_visible(canonical);                             // :385
        warp.warp(canonical, canonical_normals);                                     // :387 (the solver, :389, is out of scope)
        volume.surface_fusion(warp, canonical, canonical_visible, depth_device, camera_pose, intr);   // :393
        std::vector<unsigned short> depth_after((size_t)rows * cols);
        depth_device.download(depth_after.data(), (size_t)cols * 2);                 // :395-396 "Depth diff"
        std::fwrite(canonical[0].val, 12, canonical.size(), out);
        std::fwrite(depth_after.data(), 2, depth_after.size(), out);
        volume.data().download(vol.data());
        std::fwrite(vol.data(), 4, vol.size(), out);
    }
    std::fclose(out);
    std::printf("headless_frame ok: %d frames, %d nodes\n", frames, M);
    return 0;
}
