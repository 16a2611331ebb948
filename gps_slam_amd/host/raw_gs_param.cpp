#include "raw_gs_param.hpp"
#include <atomic>

#include <fstream>

using namespace gpsh;
using torch::indexing::Slice;

// ------------------------------------------------------------------------------------------------ Camera
Camera::Camera(int width, int height, float fx, float fy, float cx, float cy, bool has_depth, const torch::Tensor& c2w)
    : width(width), height(height), fx(fx), fy(fy), cx(cx), cy(cy), has_depth(has_depth), c2w(c2w) {
    c2w_slam = c2w;
    K = torch::tensor({{fx, 0.0f, cx}, {0.0f, fy, cy}, {0.0f, 0.0f, 1.0f}}, torch::kFloat32);
}

torch::Tensor poseInv(const torch::Tensor& c2w) {
    auto R = c2w.index({Slice(0, 3), Slice(0, 3)});
    auto T = c2w.index({Slice(0, 3), Slice(3, 4)});
    auto Rinv = R.transpose(0, 1);
    auto out = torch::eye(4, c2w.options());
    out.index_put_({Slice(0, 3), Slice(0, 3)}, Rinv);
    out.index_put_({Slice(0, 3), Slice(3, 4)}, torch::matmul(-Rinv, T));
    return out;
}

void Camera::toGPU(const torch::Device& device) { toGPU(device, torch::Tensor()); }

void Camera::toGPU(const torch::Device& device, const torch::Tensor& frame_rgba_u8) {
    const bool convert = frame_rgba_u8.defined() && !image.defined();
    if (convert) image = torch::empty({frame_rgba_u8.size(0), frame_rgba_u8.size(1), 3}, f32(frame_rgba_u8.device()));
    bool converted = false;
    if (!pack_.defined()) {
        auto c = c2w_slam.to(torch::kCPU, torch::kFloat32).contiguous();
        float h[28];
        const float* m = c.data_ptr<float>();
        // viewmat = [R^T | -R^T t] row-major
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) h[4 * r + k] = m[4 * k + r];
            h[4 * r + 3] = -(m[0 * 4 + r] * m[3] + m[1 * 4 + r] * m[7] + m[2 * 4 + r] * m[11]);
        }
        h[12] = h[13] = h[14] = 0.f; h[15] = 1.f;
        auto Kc = K.to(torch::kCPU, torch::kFloat32).contiguous();
        for (int k = 0; k < 9; k++) h[16 + k] = Kc.data_ptr<float>()[k];
        h[25] = m[3]; h[26] = m[7]; h[27] = m[11];
        // through the kernel argument buffer: no pinned staging tensor, no copy-engine latency on the frame stream
        pack_ = torch::empty({28}, f32(device));
        static std::atomic<uint64_t> serial{0};
        pack_serial_ = ++serial;
        if (convert) {
            check(gps_rgba8_to_rgbf_and_floats((int)(frame_rgba_u8.size(0) * frame_rgba_u8.size(1)), ptr<uint8_t>(frame_rgba_u8),
                                               fptr(image), fptr(pack_), h, 28, current_stream()), "gps_rgba8_to_rgbf_and_floats");
            converted = true;
        } else {
            check(gps_upload_floats(fptr(pack_), h, 28, current_stream()), "gps_upload_floats");
        }
    }
    if (convert && !converted)
        check(gps_rgba8_to_rgbf((int)(frame_rgba_u8.size(0) * frame_rgba_u8.size(1)), ptr<uint8_t>(frame_rgba_u8), fptr(image),
                                current_stream()), "gps_rgba8_to_rgbf");
    if (image.defined() && !image.is_cuda()) image = image.to(device);
    if (depth.defined() && !depth.is_cuda()) depth = depth.to(device);
}

// ------------------------------------------------------------------------------------------------ params
namespace {

// tensor_math.cpp:184-201 computeQuat + quaternionFromAxisAngle
const int64_t TAIL[RawGaussianParams::NUM][2] = {{3, 0}, {3, 0}, {4, 0}, {3, 0}, {15, 3}, {1, 0}};

}  // namespace

namespace {
// gps_init_gaussians on k points into the rows the six pointers name
void init_rows(const torch::Tensor& xyz, const torch::Tensor& rgb, const torch::Tensor& normals, int K, float init_opacs,
               float max_scale, float min_scale, float* const out[6]) {
    const int64_t P = xyz.size(0);
    auto x = xyz.contiguous(), c = rgb.contiguous();
    auto n = normals.defined() ? normals.contiguous() : torch::Tensor();
    check_f32_dev(x, "xyz"); check_f32_dev(c, "rgb");
    auto knn = distCUDA2(x);
    check(gps_init_gaussians((int)P, fptr(x), fptr(c), fptr(n), fptr(knn), K, init_opacs, max_scale, min_scale, out[0], out[1],
                             out[2], out[3], out[4], out[5], current_stream()), "gps_init_gaussians");
}
}  // namespace

// One launch (gps_init_gaussians) + the KNN kernel instead of the reference's ~35 tensor ops (raw_gs_param.cpp:24-62); the
// op-for-op torch sequence this replaces is what tests/test_init_prune_raycast_gpu.py::_ref_init restates.
std::vector<torch::Tensor> RawGaussianParams::make(const torch::Tensor& xyz, const torch::Tensor& rgb,
                                                   const torch::Tensor& normals, int max_sh_degree, float init_opacs,
                                                   float max_scale, float min_scale) {
    const int64_t P = xyz.size(0);
    const auto opt = xyz.options();
    const int K = numShBases(max_sh_degree);
    std::vector<torch::Tensor> t = {torch::empty({P, 3}, opt), torch::empty({P, 3}, opt), torch::empty({P, 4}, opt),
                                    torch::empty({P, 3}, opt), torch::empty({P, K - 1, 3}, opt), torch::empty({P, 1}, opt)};
    float* out[6];
    for (int k = 0; k < 6; k++) out[k] = fptr(t[k]);
    if (P > 0) init_rows(xyz, rgb, normals, K, init_opacs, max_scale, min_scale, out);
    return t;
}

// init() of `xyz.size(0)` new Gaussians written straight behind the existing ones (make() + add() without the six copies)
void RawGaussianParams::appendInit(const torch::Tensor& xyz, const torch::Tensor& rgb, const torch::Tensor& normals,
                                   int max_sh_degree, float init_opacs, float max_scale, float min_scale) {
    const int64_t n = xyz.size(0);
    const int K = numShBases(max_sh_degree);
    if (!buf_[0].defined() || N_ + n > cap_ || K != K_)
        reserve(std::max<int64_t>(2 * cap_, std::max<int64_t>(1 << 19, N_ + n)), K, xyz.device());
    if (n == 0) return;
    float* out[6];
    for (int k = 0; k < 6; k++) out[k] = fptr(buf_[k]) + N_ * (buf_[k].numel() / buf_[k].size(0));
    // quats rows are 16 bytes: any row offset keeps the float4 store aligned
    init_rows(xyz, rgb, normals, K, init_opacs, max_scale, min_scale, out);
    N_ += n;
    version_++;
}

void RawGaussianParams::reserve(int64_t capacity, int sh_k, const torch::Device& device) {
    if (capacity <= cap_ && sh_k == K_ && buf_[0].defined()) return;
    device_ = device;
    const int old_k = K_;
    K_ = sh_k;
    for (int k = 0; k < NUM; k++) {
        std::vector<int64_t> shape = {capacity};
        if (k == 4) { shape.push_back(K_ - 1); shape.push_back(3); }
        else shape.push_back(TAIL[k][0]);
        auto nb = torch::empty(shape, f32(device));
        if (N_ > 0 && buf_[k].defined() && old_k == K_) nb.slice(0, 0, N_).copy_(buf_[k].slice(0, 0, N_));
        buf_[k] = nb;
        alt_[k] = torch::empty_like(nb);
    }
    if (old_k != K_) N_ = 0;
    cap_ = capacity;
    version_++;
}

void RawGaussianParams::init(const torch::Tensor& xyz, const torch::Tensor& rgb, const torch::Tensor& normals,
                             int max_sh_degree, float init_opacs, float max_scale, float min_scale, int exposure_num) {
    auto t = make(xyz, rgb, normals, max_sh_degree, init_opacs, max_scale, min_scale);
    reserve(std::max<int64_t>(cap_, std::max<int64_t>(1 << 19, 2 * xyz.size(0))), numShBases(max_sh_degree), xyz.device());
    N_ = 0;
    add(t);
    exposure = torch::eye(3, 4, xyz.options()).unsqueeze(0).repeat({exposure_num, 1, 1});  // raw_gs_param.cpp:70-73
}

void RawGaussianParams::add(const std::vector<torch::Tensor>& t) {
    TORCH_CHECK((int)t.size() == NUM, "expected ", NUM, " tensors");
    const int64_t n = t[0].size(0);
    if (!buf_[0].defined() || N_ + n > cap_)
        reserve(std::max<int64_t>(2 * cap_, std::max<int64_t>(1 << 19, N_ + n)), K_, t[0].device());
    for (int k = 0; k < NUM; k++) buf_[k].slice(0, N_, N_ + n).copy_(t[k]);
    N_ += n;
    version_++;
}

void RawGaussianParams::add(const RawGaussianParams& other) {
    std::vector<torch::Tensor> t;
    for (int k = 0; k < NUM; k++) t.push_back(other.view(k));
    add(t);
}

void RawGaussianParams::remove(const torch::Tensor& mask) { removeKeep(~mask); }

// Keeps the rows whose byte in keep_mask is set: ordered compaction of the mask into row ids (gps_compact_mask: nonzero()
// without its launches), one read of their number, ONE gather launch for the six tensors into the alternate buffers.
int64_t RawGaussianParams::removeKeep(const torch::Tensor& keep_mask) {
    TORCH_CHECK(keep_mask.scalar_type() == torch::kBool && keep_mask.numel() == N_, "keep mask: bool [N]");
    auto km = keep_mask.contiguous();
    const auto dev = buf_[0].device();
    auto ids = torch::empty({std::max<int64_t>(N_, 1)}, i32(dev));
    auto count = torch::empty({1}, i32(dev));
    if (!host_count_.defined()) host_count_ = torch::zeros({16}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    auto ws = torch::empty({gps_compact_mask_workspace_bytes((int)N_)}, u8(dev));
    auto stream = c10::hip::getCurrentHIPStream();
    check(gps_compact_mask((int)N_, reinterpret_cast<const uint8_t*>(km.data_ptr<bool>()), iptr(ids), iptr(count),
                           host_count_.data_ptr<int32_t>(), ws.data_ptr(), ws.numel(), (gps_stream)stream.stream()), "gps_compact_mask");
    stream.synchronize();
    const int64_t m = host_count_.data_ptr<int32_t>()[0];
    keep_ids32_ = ids.slice(0, 0, m);
    keep_idx_ = torch::Tensor();  // int64 copy made on demand (keep_index())
    if (m == N_) return m;
    const float* srcs[NUM]; float* dsts[NUM]; int32_t rows[NUM];
    for (int k = 0; k < NUM; k++) {
        srcs[k] = fptr(buf_[k]); dsts[k] = fptr(alt_[k]);
        rows[k] = (int32_t)(buf_[k].numel() / buf_[k].size(0));
    }
    check(gps_gather_rows((int)m, iptr(ids), NUM, srcs, dsts, rows, (gps_stream)stream.stream()), "gps_gather_rows");
    for (int k = 0; k < NUM; k++) std::swap(buf_[k], alt_[k]);
    N_ = m;
    version_++;
    return m;
}

torch::Tensor RawGaussianParams::keep_index() const {
    if (!keep_idx_.defined() && keep_ids32_.defined()) keep_idx_ = keep_ids32_.to(torch::kInt64);
    return keep_idx_;
}

// ------------------------------------------------------------------------------------------------ persistence
void RawGaussianParams::savePly(const std::string& filename) const {
    std::ofstream o(filename, std::ios_base::out | std::ios_base::binary);
    TORCH_CHECK((bool)o, "savePly: cannot open ", filename);
    const int64_t numPoints = N_;
    auto meansCpu = view(0).cpu().contiguous(), scalesCpu = view(1).cpu().contiguous(), quatsCpu = view(2).cpu().contiguous();
    auto dcCpu = view(3).cpu().contiguous(), opacCpu = view(5).cpu().contiguous();
    auto restCpu = view(4).cpu().transpose(1, 2).reshape({numPoints, -1}).contiguous();  // channel-major like the reference
    o << "ply\n" << "format binary_little_endian 1.0\n" << "element vertex " << numPoints << "\n";
    for (const char* p : {"x", "y", "z", "nx", "ny", "nz"}) o << "property float " << p << "\n";
    for (int64_t i = 0; i < dcCpu.size(1); i++) o << "property float f_dc_" << i << "\n";
    for (int64_t i = 0; i < restCpu.size(1); i++) o << "property float f_rest_" << i << "\n";
    o << "property float opacity\n";
    for (int i = 0; i < 3; i++) o << "property float scale_" << i << "\n";
    for (int i = 0; i < 4; i++) o << "property float rot_" << i << "\n";
    o << "end_header\n";
    // one row per Gaussian; assembled once instead of seven small writes per point
    auto rows = torch::cat({meansCpu, torch::zeros({numPoints, 3}), dcCpu, restCpu, opacCpu.view({numPoints, 1}), scalesCpu, quatsCpu}, 1)
                    .contiguous();
    o.write(reinterpret_cast<const char*>(rows.data_ptr<float>()), (std::streamsize)rows.nbytes());
}

void RawGaussianParams::saveTensor(const std::string& filename) const {
    torch::serialize::OutputArchive archive;
    static const char* names[NUM] = {"means", "scales", "quats", "featuresDc", "featuresRest", "opacities"};
    for (int k = 0; k < NUM; k++) archive.write(names[k], view(k).contiguous());
    archive.write("exposure", exposure.defined() ? exposure : torch::eye(3, 4, f32(device_)).unsqueeze(0));
    archive.save_to(filename);
}

void RawGaussianParams::loadTensor(const std::string& filename) {
    torch::serialize::InputArchive archive;
    archive.load_from(filename);
    static const char* names[NUM] = {"means", "scales", "quats", "featuresDc", "featuresRest", "opacities"};
    std::vector<torch::Tensor> t(NUM);
    for (int k = 0; k < NUM; k++) { archive.read(names[k], t[k]); t[k] = t[k].to(device_); }
    archive.read("exposure", exposure);
    exposure = exposure.to(device_);
    N_ = 0;
    add(t);
}
