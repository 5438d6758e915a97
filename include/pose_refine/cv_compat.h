// cv_compat.h -- the reference API passes depth images and intrinsics as cv::Mat
// (renderer.h:195-196, depth_scene.h:16-27, pcd_scene.h:54-58).  With OpenCV installed the real
// headers are used; without it (this image has none) a minimal cv::Mat with the members the path
// touches (rows, cols, type(), data, at<T>, ptr<T>, the (rows, cols, type, void*) view constructor
// and an owning (rows, cols, type) constructor) keeps the same source compiling.
#pragma once
#if __has_include(<opencv2/core/core.hpp>) && !defined(POSE_REFINE_NO_OPENCV)
#include <opencv2/core/core.hpp>
#else
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#ifndef CV_8U
#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_32SC1 CV_32S
#define CV_16UC1 CV_16U
#define CV_32FC1 CV_32F
#endif
namespace cv {
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    size_t step = 0;                                               // bytes per row (always dense here; cv::Mat::step converts to size_t the same way)
    Mat() {}
    Mat(int r, int c, int type) : rows(r), cols(c), step((size_t)c * elem(type)), type_(type), own_(new std::vector<unsigned char>((size_t)r * c * elem(type), 0))
    { data = own_->data(); }
    Mat(int r, int c, int type, void *ext) : rows(r), cols(c), data(static_cast<unsigned char *>(ext)), step((size_t)c * elem(type)), type_(type) {}
    bool isContinuous() const { return true; }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    size_t elemSize() const { return elem(type_); }
    template <class T> T &at(int r, int c = 0) { return reinterpret_cast<T *>(data)[(size_t)r * cols + c]; }
    template <class T> const T &at(int r, int c = 0) const { return reinterpret_cast<const T *>(data)[(size_t)r * cols + c]; }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data) + (size_t)r * cols; }
    template <class T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data) + (size_t)r * cols; }
private:
    static size_t elem(int t) { return t == CV_8U ? 1 : (t == CV_16U ? 2 : 4); }
    int type_ = CV_8U;
    std::shared_ptr<std::vector<unsigned char>> own_;
};
}  // namespace cv
#endif
