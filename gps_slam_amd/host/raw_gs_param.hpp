// RawGaussianParams (include/raw_gs_param.h:7-85, src/raw_gs_param.cpp) and Camera (include/dataset_reader.h:111-169)
// as the hot path uses them.
//
// MI355X layout: the seven tensors are [:N] views into capacity-sized device buffers that are allocated once
// (1M Gaussians incl. Adam state is under 1 GB of the 288 GB), so add / remove never reallocate -- the reference
// re-`cat`s and re-indexes all tensors into fresh allocations on every add and prune (raw_gs_param.cpp:123-157).
#pragma once
#include "gps_host_common.hpp"
#include "gsplat_wapper.hpp"

struct Camera {
    int id = -1;
    int width = 0, height = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    bool has_depth = false;
    torch::Tensor image;     // [H,W,3] float in [0,1]
    torch::Tensor c2w;       // camera to world (dataset)
    torch::Tensor c2w_slam;  // camera to world (SLAM estimate)
    torch::Tensor depth;     // [H,W,1] metres
    torch::Tensor K;

    Camera() {}
    Camera(int width, int height, float fx, float fy, float cx, float cy, bool has_depth, const torch::Tensor& c2w);

    // curr_cam.toGPU() (slam_pipeline.cpp:84).  viewmat = poseInv(c2w_slam), K and the camera position are packed on
    // the host and go up in ONE 28-float copy (the reference launches ~8 tiny device kernels for poseInv).
    void toGPU(const torch::Device& device = torch::kCUDA);
    // the same when the frame already sits in HBM as the uchar4 image UpdateView uploaded and no float image was kept for this
    // camera: image := rgba[..., :3] / 255 and the pack in ONE launch (gps_rgba8_to_rgbf_and_floats)
    void toGPU(const torch::Device& device, const torch::Tensor& frame_rgba_u8);
    void invalidate() { pack_ = torch::Tensor(); }
    const float* viewmat() const { return pack_.data_ptr<float>(); }
    const float* Kmat() const { return pack_.data_ptr<float>() + 16; }
    const float* cam_pos() const { return pack_.data_ptr<float>() + 25; }
    bool on_device() const { return pack_.defined(); }
    torch::Tensor viewmat_tensor() const { return pack_.slice(0, 0, 16).view({4, 4}); }
    torch::Tensor K_tensor() const { return pack_.slice(0, 16, 25).view({3, 3}); }
    torch::Tensor cam_pos_tensor() const { return pack_.slice(0, 25, 28); }
    const torch::Tensor& pack_tensor() const { return pack_; }  // device float[28] = viewmat | K | cam_pos
    uint64_t pack_serial() const { return pack_.defined() ? pack_serial_ : 0; }  // identity of this upload (addresses get recycled)

private:
    torch::Tensor pack_;  // device float[28] = viewmat(16) | K(9) | cam_pos(3)
    uint64_t pack_serial_ = 0;
};

torch::Tensor poseInv(const torch::Tensor& c2w);  // src/tensor_math.cpp:56-67 (host tensors)

class RawGaussianParams {
    friend class RawGaussianModel;

public:
    static constexpr int NUM = 6;  // means, scales, quats, featuresDc, featuresRest, opacities
    RawGaussianParams() = default;

    // raw_gs_param.cpp:11-74: initial tensors from points / colours / normals -> replaces the contents
    void init(const torch::Tensor& xyz, const torch::Tensor& rgb, const torch::Tensor& normals, int max_sh_degree,
              float init_opacs, float max_scale = -1, float min_scale = -1, int exposure_num = 1);
    // the seven tensors init() would produce, without storing them (used by add paths)
    static std::vector<torch::Tensor> make(const torch::Tensor& xyz, const torch::Tensor& rgb,
                                           const torch::Tensor& normals, int max_sh_degree, float init_opacs,
                                           float max_scale, float min_scale);

    bool isDefined() const { return N_ > 0; }
    uint64_t version() const { return version_; }
    int getGaussianNum() const { return (int)N_; }
    void reserve(int64_t capacity, int sh_k, const torch::Device& device);
    // make() + add() in place: the new rows are initialised directly behind the existing ones
    void appendInit(const torch::Tensor& xyz, const torch::Tensor& rgb, const torch::Tensor& normals, int max_sh_degree,
                    float init_opacs, float max_scale = -1, float min_scale = -1);
    void add(const RawGaussianParams& other);            // raw_gs_param.cpp:123-145
    void add(const std::vector<torch::Tensor>& tensors);  // same, from loose tensors (NUM entries, reference order)
    void remove(const torch::Tensor& mask);               // raw_gs_param.cpp:148-157: mask = rows to delete
    // raw_gs_param.cpp:159-254: 3DGS PLY (binary little endian, raw parameters, property order of the reference) and the
    // libtorch archive with the reference's keys (means, scales, quats, featuresDc, featuresRest, opacities, exposure)
    void savePly(const std::string& filename) const;
    void saveTensor(const std::string& filename) const;
    void loadTensor(const std::string& filename);
    void toGPU() {}                                        // buffers always live on the device
    void requireGrad(bool) {}

    torch::Tensor getMeans() const { return view(0); }
    torch::Tensor getScales() const { return view(1); }
    torch::Tensor getQuats() const { return view(2); }
    torch::Tensor getFeaturesDc() const { return view(3); }
    torch::Tensor getFeaturesRest() const { return view(4); }
    torch::Tensor getOpacities() const { return view(5); }
    torch::Tensor getExposure() const { return exposure; }
    torch::Tensor getRealMeans() const { return view(0); }
    torch::Tensor getRealScales() const { return torch::exp(view(1)); }
    torch::Tensor getRealOpacities() const { return torch::sigmoid(view(5)); }

    int64_t capacity() const { return cap_; }
    int shK() const { return K_; }
    torch::Tensor buffer(int k) const { return buf_[k]; }  // capacity-sized storage of tensor k
    torch::Tensor keep_index() const;                      // int64 indices kept by the last remove()
    int64_t removeKeep(const torch::Tensor& keep_mask);    // remove(~keep_mask); returns the number of rows kept

protected:
    torch::Tensor view(int k) const { return buf_[k].defined() ? buf_[k].slice(0, 0, N_) : torch::Tensor(); }
    torch::Tensor buf_[NUM], alt_[NUM];
    torch::Tensor exposure;
    mutable torch::Tensor keep_idx_;
    torch::Tensor keep_ids32_, host_count_;
    int64_t N_ = 0, cap_ = 0;
    uint64_t version_ = 0;  // bumped by every change of the row set (add / remove / reserve / load): part of RawGaussianModel's prefetch key
    int K_ = 16;
    torch::Device device_ = torch::kCUDA;
};
