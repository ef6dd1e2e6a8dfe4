// demo_calls.cpp -- makes exactly the kfusion calls of the reference's demo (/root/reference/apps/demo.cpp), in its order, against
// the source-compatible headers of this repository: construction (:26-27), the per-frame loop (:89-104), show_raycasted's two
// renderImage overloads + download (:47-53) and show_warp's getNodesAsMat (:67).  OpenCV (imread / imshow / viz) is what the demo
// wraps around those calls; it is absent from this image, so frames come from a file and the "windows" are checksums and a PPM.
//   demo_calls <cols> <rows> <frames> <dims> <size_m> <in.bin> <out_prefix>
// in.bin : fx fy cx cy f32[4], then per frame depth u16[rows*cols] (mm)
// writes <out_prefix>.views.bin: per rendered frame the side-by-side view (rows x 2*cols BGRA) of renderImage(image, 3) followed by the
//        one of renderImage(image, pose, 3); <out_prefix>.nodes.bin: getNodesAsMat() of the last frame (f32[N*3]); <out_prefix>.ppm
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <kfusion/kinfu.hpp>
using namespace kfusion;

struct DynamicFusionApp
{
    DynamicFusionApp(int cols, int rows, int dims, float size, const float iv[4]) : interactive_mode_(false)
    {
        KinFuParams params = KinFuParams::default_params_dynamicfusion();      // demo.cpp:26
        params.cols = cols; params.rows = rows; params.intr = Intr(iv[0], iv[1], iv[2], iv[3]);
        params.volume_dims = Vec3i::all(dims); params.volume_size = Vec3f::all(size);
        params.volume_pose = Affine3f().translate(Vec3f(-size / 2, -size / 2, 0.5f));
        kinfu_ = KinFu::Ptr(new KinFu(params));                                 // :27
    }
    void show_raycasted(KinFu& kinfu, FILE* out)                                // :44-62
    {
        const int mode = 3;
        for (int pass = 0; pass < 2; ++pass) {
            interactive_mode_ = pass == 1;
            if (interactive_mode_) kinfu.renderImage(view_device_, viewer_pose_, mode);   // viz.getViewerPose()
            else kinfu.renderImage(view_device_, mode);
            view_host_.resize((size_t)view_device_.rows() * view_device_.cols() * 4);     // view_host_.create(rows, cols, CV_8UC4)
            view_device_.download(view_host_.data(), (size_t)view_device_.cols() * 4);
            std::fwrite(view_host_.data(), 1, view_host_.size(), out);
        }
    }
    size_t show_warp(KinFu& kinfu, FILE* out)                                   // :64-68
    {
        const WarpField::NodesMat warp_host = kinfu.getWarp().getNodesAsMat();
        if (out && !warp_host.empty()) std::fwrite(warp_host[0].val, 12, warp_host.size(), out);
        return warp_host.size();
    }
    bool interactive_mode_;
    KinFu::Ptr kinfu_;
    Affine3f viewer_pose_;
    std::vector<unsigned char> view_host_;
    cuda::Image view_device_;
    cuda::Depth depth_device_;
};

int main(int argc, char** argv)
{
    if (argc != 8) { std::fprintf(stderr, "usage: %s cols rows frames dims size in.bin out_prefix\n", argv[0]); return 2; }
    const int cols = std::atoi(argv[1]), rows = std::atoi(argv[2]), frames = std::atoi(argv[3]), dims = std::atoi(argv[4]);
    const float size = (float)std::atof(argv[5]);
    FILE* in = std::fopen(argv[6], "rb");
    if (!in) { std::perror("in"); return 2; }
    float iv[4];
    if (std::fread(iv, 4, 4, in) != 4) return 2;
    const std::string prefix = argv[7];
    DynamicFusionApp app(cols, rows, dims, size, iv);
    KinFu& dynamic_fusion = *app.kinfu_;
    FILE* views = std::fopen((prefix + ".views.bin").c_str(), "wb");
    if (!views) return 2;
    std::vector<unsigned short> depth((size_t)rows * cols);
    int shown = 0; size_t n_nodes = 0;
    for (int i = 0; i < frames; ++i) {
        if (std::fread(depth.data(), 2, depth.size(), in) != depth.size()) return 2;
        app.depth_device_.upload(depth.data(), (size_t)cols * 2, rows, cols);           // :89
        const bool has_image = dynamic_fusion(app.depth_device_);                        // :93
        app.viewer_pose_ = dynamic_fusion.getCameraPose();                               // :104 viz.setViewerPose(getCameraPose())
        if (has_image) { app.show_raycasted(dynamic_fusion, views); ++shown; }           // :96-97
        n_nodes = app.show_warp(dynamic_fusion, nullptr);                                // :108
    }
    std::fclose(views); std::fclose(in);
    FILE* nodes = std::fopen((prefix + ".nodes.bin").c_str(), "wb");
    app.show_warp(dynamic_fusion, nodes);
    std::fclose(nodes);
    if (!app.view_host_.empty()) {                                                       // the last view as a picture
        FILE* ppm = std::fopen((prefix + ".ppm").c_str(), "wb");
        const int w = app.view_device_.cols(), h = app.view_device_.rows();
        std::fprintf(ppm, "P6\n%d %d\n255\n", w, h);
        for (size_t p = 0; p < (size_t)w * h; ++p) { const unsigned char* b = &app.view_host_[4 * p]; const unsigned char rgb[3] = {b[2], b[1], b[0]}; std::fwrite(rgb, 1, 3, ppm); }
        std::fclose(ppm);
    }
    std::printf("demo_calls ok: %d frames, %d shown, view %d x %d, %zu warp nodes\n", frames, shown, app.view_device_.cols(), app.view_device_.rows(), n_nodes);
    return 0;
}
