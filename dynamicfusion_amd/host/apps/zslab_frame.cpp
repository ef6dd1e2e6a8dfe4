// zslab_frame.cpp -- the hot-path frame of headless_frame.cpp on a Z-slab-sharded volume, in C++ over RCCL: one process per GPU
// (RANK / WORLD_SIZE / LOCAL_RANK from the environment, as torchrun and mpirun set them; WORLD_SIZE absent = 1).
//   zslab_frame <dims> <size_m> <cols> <rows> <frames> <nodes> <k> <in.bin> <out.bin> <id_file> [exchange|recompute] [slab=<r>/<n>] [bounds=0,b1,..,Z]
// in.bin as headless_frame's.  Rank 0 reads it and broadcasts every frame's depth image and node transforms (ZSlabComm::broadcast);
// every rank integrates the planes it owns (warped when nodes > 0), the halo planes are exchanged (ncclSend/Recv) or recomputed, the
// ray-cast is the two-stage sharded one.  out.bin (rank 0): points f32[rows*cols*4], normals f32[rows*cols*4] of the last frame, then
// this rank's OWN planes u32[z_own_n * dims^2].
// slab=<r>/<n> (single process, no communicator): integrate only slab r of n and write its own planes -- lets a one-GPU test compare
// every shard of the C++ path with the unsharded volume.  With `recompute` the shard is setSlab(z0, n, halo, integrate_halo = true)
// and ALL its stored planes (own + the halo planes it integrated itself) are written.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include <kfusion/cuda/zslab.hpp>
#include <kfusion/cuda/imgproc.hpp>
#include <kfusion/warp_field.hpp>

using namespace kfusion;

static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
static Affine3f to_affine(const float a[12])
{
    return aff12_to_affine(a);
}

int main(int argc, char** argv)
{
    if (argc == 5 && (!std::strcmp(argv[1], "bounds") || !std::strcmp(argv[1], "bounds-minmax"))) {     // zslab_frame bounds[-minmax] <world> <halo> <weights.f64>: ZSlabComm::slabBounds[MinMax] of the file's planes
        std::vector<double> w;
        if (FILE* f = std::fopen(argv[4], "rb")) { double v; while (std::fread(&v, 8, 1, f) == 1) w.push_back(v); std::fclose(f); }
        const std::vector<int> b = !std::strcmp(argv[1], "bounds") ? cuda::ZSlabComm::slabBounds((int)w.size(), std::atoi(argv[2]), std::atoi(argv[3]), w)
                                                                   : cuda::ZSlabComm::slabBoundsMinMax((int)w.size(), std::atoi(argv[2]), std::atoi(argv[3]), w);
        for (size_t i = 0; i < b.size(); ++i) std::printf("%s%d", i ? "," : "", b[i]);
        std::printf("\n");
        return 0;
    }
    if (argc < 11) { std::fprintf(stderr, "usage: %s dims size cols rows frames nodes k in.bin out.bin id_file [exchange|recompute] [slab=r/n]\n", argv[0]); return 2; }
    const int dims = std::atoi(argv[1]); const float size = (float)std::atof(argv[2]);
    const int cols = std::atoi(argv[3]), rows = std::atoi(argv[4]), frames = std::atoi(argv[5]), M = std::atoi(argv[6]), k = std::atoi(argv[7]);
    bool exchange = true, row_bands = false; int only_r = -1, only_n = 0;
    std::vector<int> bounds;                                // bounds=0,b1,...,Z : explicit slab boundaries (work-balanced partitions)
    for (int i = 11; i < argc; ++i) {
        if (!std::strcmp(argv[i], "recompute")) exchange = false;
        else if (!std::strcmp(argv[i], "rows")) row_bands = true;         // the ray-cast's normals reduce-scattered by pixel rows: every rank writes its band
        else if (!std::strncmp(argv[i], "bounds=", 7)) { for (const char* q = argv[i] + 7; *q;) { bounds.push_back(std::atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; } }
        else if (!std::strncmp(argv[i], "slab=", 5)) std::sscanf(argv[i] + 5, "%d/%d", &only_r, &only_n);
    }
    const int world = only_n > 0 ? only_n : env_int("WORLD_SIZE", 1), rank = only_n > 0 ? only_r : env_int("RANK", 0);
    if (hipSetDevice(only_n > 0 ? 0 : env_int("LOCAL_RANK", 0)) != hipSuccess) { std::fprintf(stderr, "hipSetDevice failed\n"); return 1; }

    // ---- inputs: rank 0 reads the file, everybody gets the small fixed part through the file too (it is tiny and static);
    // the per-frame data goes over RCCL below
    FILE* in = std::fopen(argv[8], "rb");
    if (!in) { std::perror("in"); return 2; }
    float pose12[12], intr_v[4];
    if (std::fread(pose12, 4, 12, in) != 12 || std::fread(intr_v, 4, 4, in) != 4) return 2;
    const Intr intr(intr_v[0], intr_v[1], intr_v[2], intr_v[3]);
    std::vector<std::vector<unsigned short> > depth(frames, std::vector<unsigned short>((size_t)rows * cols));
    std::vector<Affine3f> cam(frames);
    for (int f = 0; f < frames; ++f) {
        float c12[12];
        if (std::fread(depth[f].data(), 2, depth[f].size(), in) != depth[f].size() || std::fread(c12, 4, 12, in) != 12) return 2;
        cam[f] = to_affine(c12);
    }
    std::vector<float> pos((size_t)M * 3), sigma(M);
    std::vector<std::vector<float> > dq(frames, std::vector<float>((size_t)M * 8));
    if (M > 0) {
        if (std::fread(pos.data(), 4, pos.size(), in) != pos.size()) return 2;
        for (int f = 0; f < frames; ++f) if (std::fread(dq[f].data(), 4, dq[f].size(), in) != dq[f].size()) return 2;
        if (std::fread(sigma.data(), 4, sigma.size(), in) != sigma.size()) return 2;
    }
    std::fclose(in);

    // ---- the shard
    cuda::TsdfVolume volume(Vec3i(8, 8, 8));                 // (re-created below with the real dims, then cut down to the slab)
    const float vz = size / dims;
    const float trunc = std::max(0.04f, 2.1f * vz);
    const int halo = world > 1 ? cuda::ZSlabComm::haloPlanes(trunc, 0.75f, 0.5f, vz) : 0;
    std::string why;
    if ((int)bounds.size() != world + 1 && !cuda::ZSlabComm::partitionOk(dims, world, halo, &why)) { std::fprintf(stderr, "zslab_frame: %s\n", why.c_str()); return 3; }
    int z0, zn; cuda::ZSlabComm::slabRange(dims, rank, world, z0, zn);
    if ((int)bounds.size() == world + 1) { z0 = bounds[rank]; zn = bounds[rank + 1] - bounds[rank]; if (zn < 1 || (world > 1 && zn < halo)) { std::fprintf(stderr, "zslab_frame: bad bounds\n"); return 3; } }
    volume.create(Vec3i(dims, dims, dims));
    volume.setSize(Vec3f::all(size)); volume.setTruncDist(0.04f); volume.setMaxWeight(64); volume.setPose(to_affine(pose12));
    volume.setRaycastStepFactor(0.75f); volume.setGradientDeltaFactor(0.5f);
    // halo recompute: the rank also INTEGRATES its halo planes -- the integrate is a pure function of the broadcast inputs, so the
    // planes come out exactly as the neighbour computes them and no exchange is needed.  The own range stays the non-overlapping one:
    // it is what the ray-cast's march / shade and the merge partition the rays by.
    if (only_n > 0 && exchange) volume.setSlab(z0, zn, 0);
    else volume.setSlab(z0, zn, halo, !exchange);

    cuda::ZSlabComm* comm = only_n > 0 ? nullptr : new cuda::ZSlabComm(rank, world, argv[10]);
    if (comm && !comm->ok()) { std::fprintf(stderr, "zslab_frame: %s\n", comm->lastError().c_str()); return 4; }
    WarpField warp(k);
    if (M > 0) {
        std::vector<Vec3f> pts(M);
        for (int i = 0; i < M; ++i) pts[i] = Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        warp.init(pts);
        for (int i = 0; i < M; ++i) (*warp.getNodes())[i].weight = sigma[i];
        warp.commit(true);
    }
    cuda::Depth depth_device; depth_device.create(rows, cols);
    cuda::DeviceArray<float> dq_device((size_t)std::max(M, 1) * 8);
    cuda::Dists dists;
    cuda::Cloud points; cuda::Normals normals;
    std::vector<float> dq_host((size_t)M * 8);
    for (int f = 0; f < frames; ++f) {
        // the sensor frame and the solver output live on rank 0: broadcast (one ncclBroadcast each), then every rank proceeds alike
        if (rank == 0 || !comm) {
            depth_device.upload(depth[f].data(), (size_t)cols * 2, rows, cols);
            if (M > 0) dq_device.upload(dq[f].data(), (size_t)M * 8);
        }
        if (comm) {
            comm->broadcast(depth_device.ptr(), depth_device.step() * (size_t)rows);        // the pitched image as it lies (same pitch on every rank)
            if (M > 0) comm->broadcast(dq_device.ptr(), (size_t)M * 8 * sizeof(float));
            if (!comm->ok()) { std::fprintf(stderr, "zslab_frame: %s\n", comm->lastError().c_str()); return 4; }
        }
        cuda::computeDists(depth_device, dists, intr);
        if (M > 0) {
            dq_device.download(dq_host.data());
            for (int i = 0; i < M; ++i) std::memcpy((void*)(*warp.getNodes())[i].transform.raw(), &dq_host[8 * (size_t)i], 32);
            warp.commit(false);
            volume.integrate(dists, cam[f], intr, warp);
        } else {
            volume.integrate(dists, cam[f], intr);
        }
        if (comm && exchange) comm->exchangeHalos(volume, halo);
        if (comm && row_bands) comm->raycastRowBands(volume, cam[f], intr, cols, rows, points, normals);
        else if (comm) comm->raycast(volume, cam[f], intr, cols, rows, points, normals, 0);
        if (comm && !comm->ok()) { std::fprintf(stderr, "zslab_frame: %s\n", comm->lastError().c_str()); return 4; }
    }
    if (comm) { if (!comm->barrier()) { std::fprintf(stderr, "zslab_frame: %s\n", comm->lastError().c_str()); return 4; } } else cuda::waitAllDefaultStream();

    if (comm && row_bands) {                                // every rank's band of the merged image: <out>.band<rank> = {row0, nrows, points, normals}
        const std::string name = std::string(argv[9]) + ".band" + std::to_string(rank);
        if (FILE* out = std::fopen(name.c_str(), "wb")) {
            const int hdr[2] = {comm->bandRow0(rows), comm->bandRows(rows)};
            std::vector<float> p((size_t)hdr[1] * cols * 4), n(p.size());
            if (hdr[1] > 0) { points.download(p.data(), (size_t)cols * 16); normals.download(n.data(), (size_t)cols * 16); }
            std::fwrite(hdr, 4, 2, out); std::fwrite(p.data(), 4, p.size(), out); std::fwrite(n.data(), 4, n.size(), out);
            std::fclose(out);
        }
    }
    if (comm && rank > 0) {                                 // the other ranks leave their OWN planes beside rank 0's file (tests): <out>.r<rank>
        const std::string name = std::string(argv[9]) + ".r" + std::to_string(rank);
        if (FILE* out = std::fopen(name.c_str(), "wb")) {
            const size_t plane = (size_t)dims * dims;
            std::vector<unsigned int> vol(plane * (size_t)volume.slabStoreN());
            volume.data().download(vol.data());
            std::fwrite(vol.data() + (size_t)(z0 - volume.slabStore0()) * plane, 4, plane * (size_t)zn, out);
            std::fclose(out);
        }
    }
    if (rank == 0 || !comm) {
        FILE* out = std::fopen(argv[9], "wb");
        if (!out) { std::perror("out"); return 2; }
        std::vector<float> p((size_t)rows * cols * 4, 0.f), n(p.size(), 0.f);
        if (comm && !row_bands) { points.download(p.data(), (size_t)cols * 16); normals.download(n.data(), (size_t)cols * 16); }
        std::fwrite(p.data(), 4, p.size(), out);
        std::fwrite(n.data(), 4, n.size(), out);
        const size_t plane = (size_t)dims * dims;
        std::vector<unsigned int> vol(plane * (size_t)volume.slabStoreN());
        volume.data().download(vol.data());
        if (only_n > 0 && !exchange) {                        // slab=r/n recompute: every STORED plane (own + recomputed halos)
            std::fwrite(vol.data(), 4, plane * (size_t)volume.slabStoreN(), out);
        } else {
            const int own_first = z0 - volume.slabStore0();
            std::fwrite(vol.data() + (size_t)own_first * plane, 4, plane * (size_t)zn, out);
        }
        std::fclose(out);
    }
    unsigned long long alive = 0;                           // the re-balance's input: what the last sweep's verdict pass kept, per 8-plane layer
    if (M > 0) for (unsigned long long a : warp.aliveBlocksPerLayer(volume)) alive += a;
    std::printf("zslab_frame ok: rank %d of %d, planes [%d, %d), halo %d (%s), %d frames, %d nodes, %llu alive blocks in the own planes\n", rank, world, z0,
                z0 + zn, halo, exchange ? "exchanged" : "recomputed", frames, M, alive);
    delete comm;
    return 0;
}
