// tests/opencv_stub/opencv2/viz/vizcore.hpp -- TEST INFRASTRUCTURE (see core/core.hpp): the cv::viz names demo.cpp uses.  The viewer
// shows nothing; WCloud(mat) writes the cloud (the demo's "warp_field") to <DFUSION_CVSTUB_OUT>/warp_field.bin.
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/core/affine.hpp>
namespace cv { namespace viz
{
    struct KeyboardEvent
    {
        enum Action { KEY_UP = 0, KEY_DOWN = 1 };
        Action action; String symbol; unsigned char code; int modifiers;
    };
    struct Color { double b, g, r; Color(double b_ = 0, double g_ = 0, double r_ = 0) : b(b_), g(g_), r(r_) {} static Color apricot() { return Color(177, 206, 251); } static Color white() { return Color(255, 255, 255); } };
    struct Widget {};
    struct WCube : Widget { WCube(const Vec3d& = Vec3d::all(-0.5), const Vec3d& = Vec3d::all(0.5), bool = true, const Color& = Color::white()) {} };
    struct WCoordinateSystem : Widget { WCoordinateSystem(double = 1.0) {} };
    struct WCloud : Widget { WCloud(const Mat& cloud, const Color& = Color::white()); };
    class Viz3d
    {
    public:
        typedef void (*KeyboardCallback)(const KeyboardEvent&, void*);
        Viz3d(const String& = String()) : cb_(nullptr), cookie_(nullptr) {}
        void showWidget(const String&, const Widget&, const Affine3d& = Affine3d::Identity()) {}
        void registerKeyboardCallback(KeyboardCallback cb, void* cookie = nullptr) { cb_ = cb; cookie_ = cookie; }
        Affine3d getViewerPose() { return pose_; }
        void setViewerPose(const Affine3d& pose) { pose_ = pose; }
        bool wasStopped() const { return false; }
        void spinOnce(int = 1, bool = false) {}
    private:
        KeyboardCallback cb_; void* cookie_; Affine3d pose_;
    };
} }
