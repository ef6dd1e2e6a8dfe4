// tests/opencv_stub/opencv2/imgproc/imgproc.hpp -- TEST INFRASTRUCTURE (see core/core.hpp); demo.cpp includes it and uses nothing of it
#pragma once
#include <opencv2/core/core.hpp>
