// Shared plumbing of the C++ host layer: libtorch tensors in, plain pointers + the current HIP stream out.
// The host layer mirrors the reference's C++ operator surface (gsplat/gsplat_wapper.hpp, include/raw_gs_model.h,
// include/raw_gs_param.h, slam/slam_pipeline.h, the ITMBasicEngine calls of slam/InfiniTAM_tools.cpp) on top of the
// C-ABI of include/gps_slam_hip.h.  libtorch is used for memory, streams and autograd bookkeeping only.
#pragma once
#include <torch/torch.h>
#include <c10/hip/HIPStream.h>

#include <map>
#include <string>

#include "gps_slam_hip.h"

typedef std::map<std::string, torch::Tensor> TensorDict;  // include/dataset_reader.h:17

namespace gpsh {

inline gps_stream current_stream() { return (gps_stream)c10::hip::getCurrentHIPStream().stream(); }

// C-ABI status -> C++ exception, like the reference's TORCH_CHECK / AT_ERROR (gsplat/rasterizer/bindings.h:11-17)
inline void check(int status, const char* what) {
    TORCH_CHECK(status == GPS_OK, what, " failed: gps_status ", status);
}

inline void check_f32_dev(const torch::Tensor& t, const char* name) {
    TORCH_CHECK(t.defined(), name, " is undefined");
    TORCH_CHECK(t.is_cuda(), name, " must be a device tensor");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

template <class T>
inline T* ptr(const torch::Tensor& t) { return t.defined() ? reinterpret_cast<T*>(t.data_ptr()) : nullptr; }
inline float* fptr(const torch::Tensor& t) { return ptr<float>(t); }
inline int32_t* iptr(const torch::Tensor& t) { return ptr<int32_t>(t); }

inline torch::TensorOptions f32(const torch::Device& d) { return torch::TensorOptions().dtype(torch::kFloat32).device(d); }
inline torch::TensorOptions i32(const torch::Device& d) { return torch::TensorOptions().dtype(torch::kInt32).device(d); }
inline torch::TensorOptions i64(const torch::Device& d) { return torch::TensorOptions().dtype(torch::kInt64).device(d); }
inline torch::TensorOptions u8(const torch::Device& d) { return torch::TensorOptions().dtype(torch::kUInt8).device(d); }

// Flat key -> value configuration standing in for the YAML::Node arguments of the reference (yaml-cpp is an I/O
// dependency outside the hot path).  Keys are the reference's YAML keys ("MODEL.sh_degree" style is accepted too).
struct Config {
    std::map<std::string, double> num;
    std::map<std::string, std::string> str;
    double get(const std::string& k, double dflt) const { auto it = num.find(k); return it == num.end() ? dflt : it->second; }
    std::string gets(const std::string& k, const std::string& dflt) const { auto it = str.find(k); return it == str.end() ? dflt : it->second; }
};

}  // namespace gpsh
