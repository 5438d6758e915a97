// thrust_compat.h -- the reference's device_vector_holder<T> exposes thrust::device_ptr<T> (begin_thr / end_thr / data_thr:
// cuda_renderer/renderer.h:175-177, cuda_icp/scene/common.h:30-33) and its drivers call thrust::copy on them
// (cuda_renderer/test.cpp:90,135, pose_renderer.cpp:12).  Thrust is a device-compiler library: when this translation unit is
// compiled by hipcc and rocThrust is on the include path the holders get those members (POSE_REFINE_HAVE_THRUST); a plain host
// compiler builds the adapters without them and uses upload() / download().  Nothing in the library itself uses Thrust.
#pragma once
#if defined(__HIPCC__) && !defined(POSE_REFINE_NO_THRUST)
#if __has_include(<thrust/device_ptr.h>)
#include <thrust/copy.h>
#include <thrust/device_ptr.h>
#define POSE_REFINE_HAVE_THRUST 1
#endif
#endif
