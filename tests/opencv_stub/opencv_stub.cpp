// tests/opencv_stub/opencv_stub.cpp -- TEST INFRASTRUCTURE: the few out-of-line functions of the OpenCV stand-in (see opencv2/core/core.hpp)
#include <dirent.h>
#include <opencv2/highgui/highgui.hpp>
#include <opencv2/viz/vizcore.hpp>

namespace
{
    std::string out_dir() { const char* e = std::getenv("DFUSION_CVSTUB_OUT"); return e ? std::string(e) : std::string(); }
}
namespace cv
{
    void glob(String pattern, std::vector<String>& result, bool)
    {
        result.clear();
        DIR* d = opendir(pattern.c_str());
        if (!d) return;
        while (dirent* e = readdir(d))
            if (e->d_name[0] != '.') result.push_back(pattern + "/" + e->d_name);
        closedir(d);
    }
    Mat imread(const String& filename, int)
    {
        Mat m;
        FILE* f = std::fopen(filename.c_str(), "rb");
        if (!f) return m;
        char magic[4]; int hdr[3];
        if (std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "DFRW", 4) == 0 && std::fread(hdr, 4, 3, f) == 3) {
            m.create(hdr[0], hdr[1], hdr[2]);
            if (std::fread(m.data, 1, (size_t)m.rows * m.step, f) != (size_t)m.rows * m.step) m = Mat();
        }
        std::fclose(f);
        return m;
    }
    bool imwrite(const String&, const Mat&) { return true; }
    void imshow(const String& winname, const Mat& mat)
    {
        const std::string dir = out_dir();
        if (dir.empty() || winname != "Scene" || mat.empty()) return;
        FILE* f = std::fopen((dir + "/Scene.bin").c_str(), "ab");
        if (!f) return;
        for (int r = 0; r < mat.rows; ++r) std::fwrite(mat.ptr<unsigned char>(r), 1, (size_t)mat.cols * mat.elemSize(), f);
        std::fclose(f);
    }
    int waitKey(int) { return -1; }
    namespace viz
    {
        WCloud::WCloud(const Mat& cloud, const Color&)
        {
            const std::string dir = out_dir();
            if (dir.empty()) return;
            FILE* f = std::fopen((dir + "/warp_field.bin").c_str(), "wb");
            if (!f) return;
            if (!cloud.empty()) std::fwrite(cloud.data, 1, (size_t)cloud.rows * cloud.step, f);
            std::fclose(f);
        }
    }
}
int cvWaitKey(int) { return -1; }
