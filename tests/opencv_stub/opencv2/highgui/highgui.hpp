// tests/opencv_stub/opencv2/highgui/highgui.hpp -- TEST INFRASTRUCTURE (see core/core.hpp): windows are files, keys never pressed.
// imread: a "picture" is a raw file {char magic[4] = "DFRW"; int32 rows, cols, type; bytes}; imshow(name, m): when the environment
// variable DFUSION_CVSTUB_OUT names a directory, window "Scene" APPENDS its image bytes to <dir>/Scene.bin (other windows: nothing).
#pragma once
#include <opencv2/core/core.hpp>
#define CV_LOAD_IMAGE_COLOR 1
#define CV_LOAD_IMAGE_ANYDEPTH 2
namespace cv
{
    Mat imread(const String& filename, int flags = 1);
    bool imwrite(const String& filename, const Mat& img);
    void imshow(const String& winname, const Mat& mat);
    int waitKey(int delay = 0);
}
int cvWaitKey(int delay = 0);
