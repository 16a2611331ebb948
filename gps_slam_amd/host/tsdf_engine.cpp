#include <cstring>
#include "tsdf_engine.hpp"

#include <hip/hip_runtime_api.h>
#include <sys/stat.h>

#include <cstdio>
#include <fstream>

using namespace gpsh;

namespace {
torch::Tensor zeros_bytes(int64_t n, const torch::Device& d) { return torch::zeros({n}, u8(d)); }
}  // namespace

TsdfEngine::TsdfEngine(int width, int height, float fx, float fy, float cx, float cy, float voxel_size, float mu,
                               float view_frustum_min, float view_frustum_max, int n_blocks, int n_buckets,
                               int n_excess, torch::Device device)
    : device_(device),
      free_image_(width, height, torch::Tensor()),
      free_vertex_(width, height, torch::Tensor()),
      live_vertex_(width, height, torch::Tensor()) {
    const int64_t P = (int64_t)width * height, n_total = (int64_t)n_buckets + n_excess;
    const auto F = f32(device), I = i32(device);
    vba_ = zeros_bytes((int64_t)n_blocks * 512 * 8, device);
    vba_alloc_list_ = torch::zeros({n_blocks}, I);
    hash_ = zeros_bytes(n_total * 16, device);
    excess_list_ = torch::zeros({n_excess}, I);
    counters_ = torch::zeros({16}, I);
    alloc_prio_ = torch::zeros({n_total}, I);
    scan_scratch_ = zeros_bytes(gps_tsdf_scratch_bytes(width, height, n_buckets, n_excess) + 16, device);
    visible_type_ = zeros_bytes(n_total, device);
    visible_ids_ = torch::zeros({n_blocks}, I);
    depth_ = torch::zeros({P}, F);
    minmax_ = torch::zeros({P * 2}, F);
    raycast_ = torch::zeros({P * 4}, F);
    icp_points_ = torch::zeros({P * 4}, F);
    icp_normals_ = torch::zeros({P * 4}, F);
    fv_visible_ids_ = torch::zeros({n_blocks}, I);
    fv_minmax_ = torch::zeros({P * 2}, F);
    fv_raycast_ = torch::zeros({P * 4}, F);
    fv_colour_ = zeros_bytes(P * 4, device);
    gps_tsdf_state& s = state_;
    s.width = width; s.height = height;
    s.fx = fx; s.fy = fy; s.cx = cx; s.cy = cy;
    s.voxel_size = voxel_size; s.mu = mu; s.view_frustum_min = view_frustum_min; s.view_frustum_max = view_frustum_max;
    s.max_w = 100;  // ITMLibSettings.cpp:10
    s.n_blocks = n_blocks; s.n_buckets = n_buckets; s.n_excess = n_excess;
    s.vba = reinterpret_cast<gps_voxel*>(vba_.data_ptr());
    s.vba_alloc_list = iptr(vba_alloc_list_);
    s.hash = reinterpret_cast<gps_hash_entry*>(hash_.data_ptr());
    s.excess_list = iptr(excess_list_);
    s.counters = iptr(counters_);
    s.alloc_prio = reinterpret_cast<uint32_t*>(alloc_prio_.data_ptr());
    s.scan_scratch = reinterpret_cast<int32_t*>(scan_scratch_.data_ptr());
    s.visible_type = ptr<uint8_t>(visible_type_);
    s.visible_ids = iptr(visible_ids_);
    s.depth = fptr(depth_);
    s.rgb = nullptr;
    s.minmax = fptr(minmax_); s.raycast = fptr(raycast_);
    s.icp_points = fptr(icp_points_); s.icp_normals = fptr(icp_normals_);
    s.fv_visible_ids = iptr(fv_visible_ids_); s.fv_minmax = fptr(fv_minmax_); s.fv_raycast = fptr(fv_raycast_);
    s.fv_colour = ptr<uint8_t>(fv_colour_);
    // the state must be complete for gps_tsdf_reset's validity check: point rgb at the colour buffer until a frame arrives
    s.rgb = ptr<uint8_t>(fv_colour_);
    free_image_ = ITMUChar4Image(width, height, fv_colour_);
    free_vertex_ = ITMFloat4Image(width, height, fv_raycast_);
    live_vertex_ = ITMFloat4Image(width, height, raycast_);
    intrinsics_d.SetFrom(width, height, fx, fy, cx, cy);
    resetAll();
}

void TsdfEngine::resetAll() {
    check(gps_tsdf_reset(&state_, current_stream()), "gps_tsdf_reset");
    framesProcessed = 0;
    tracked_frames_ = evals_total_ = rode_total_ = used_total_ = 0;
    camPoses.clear();
    camIntrincs.clear();
    check(gps_track_state_reset(&track_state_), "gps_track_state_reset");
    if (track_mailbox_.defined()) track_state_.host_mailbox = track_mailbox_.data_ptr();
    track_state_.mailbox_bytes = mailboxBytes();
    track_state_.dev_arg_line = bar_arg_line ? track_arg_line_.get() : nullptr;
}

void TsdfEngine::turnOnTracking(const char* levels, int numIterC, int numIterF, float outlierSpaceC, float outlierSpaceF,
                                    float minstep, float tukeyCutOff, int framesToSkip, int framesToWeight) {
    check(gps_track_config_init(&track_cfg_, levels, numIterC, numIterF, outlierSpaceC, outlierSpaceF, minstep, tukeyCutOff,
                                framesToSkip, framesToWeight), "gps_track_config_init");
    if (!track_scratch_.defined()) {
        track_scratch_ = torch::empty({gps_track_scratch_bytes(state_.width, state_.height)}, u8(device_));
        // (room for the answer blocks and the host-summed row tables of every group: gps_track_state.mailbox_bytes)
        track_mailbox_ = torch::empty({(GPS_TRACK_MAILBOX_BLOCK_BYTES + GPS_TRACK_MAILBOX_ROWS_BYTES) / 4 * (1 + kMaxRidingAlong)},
                                      torch::TensorOptions().dtype(torch::kFloat32).pinned_memory(true));
        std::memset(track_mailbox_.data_ptr(), 0, (size_t)track_mailbox_.nbytes());   // (not torch::zeros: tsdf_engine.hpp, Image)
        void* line = nullptr;
        check(gps_track_arg_line_alloc(&line), "gps_track_arg_line_alloc");
        if (line) track_arg_line_ = std::shared_ptr<void>(line, [](void* p) { (void)gps_track_arg_line_free(p); });
    }
    track_state_.host_mailbox = track_mailbox_.data_ptr();
    track_state_.mailbox_bytes = mailboxBytes();
    track_state_.dev_arg_line = bar_arg_line ? track_arg_line_.get() : nullptr;
    trackingActive = true;
}

ITMTrackingState* TsdfEngine::ProcessFrame(const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16) {
    TORCH_CHECK(rgb_u8.is_cuda() && rgb_u8.scalar_type() == torch::kUInt8 && rgb_u8.is_contiguous() &&
                    rgb_u8.size(-1) == 4, "rgb must be a contiguous uint8 [H,W,4] device tensor (uchar4)");
    TORCH_CHECK(depth_mm_i16.is_cuda() && depth_mm_i16.scalar_type() == torch::kInt16 && depth_mm_i16.is_contiguous(),
                "depth must be a contiguous int16 [H,W] device tensor (millimetres)");
    {
        std::lock_guard<std::mutex> lk(state_mu_);
        frame_inputs_ = {rgb_u8, depth_mm_i16};  // keep alive while kernels may read them
        state_.rgb = ptr<uint8_t>(rgb_u8);
    }
    if (trackingActive) {
        if (!track_scratch_.defined()) turnOnTracking();  // ITMLibSettings defaults
        std::function<void()> gate;
        gate.swap(beforeNextFusion);
        check(gps_tsdf_process_frame_tracked_gated(&state_, ptr<int16_t>(depth_mm_i16), &track_cfg_, &track_state_,
                                                   track_scratch_.data_ptr(), track_scratch_.numel(), current_stream(),
                                                   gate ? +[](void* f) { (*static_cast<std::function<void()>*>(f))(); } : nullptr,
                                                   gate ? &gate : nullptr),
              "gps_tsdf_process_frame_tracked");
        pose_d_.SetBoth(track_state_.pose_M, track_state_.pose_invM);
        tracked_frames_++;
        for (int l = 0; l < 8; l++) evals_total_ += (int64_t)track_state_.diag[l];
        rode_total_ += (int64_t)track_state_.diag[12]; used_total_ += (int64_t)track_state_.diag[13];
    } else {
        TORCH_CHECK((int)gtC2wPoses.size() > framesProcessed, "gtC2wPoses must hold the pose of frame ", framesProcessed);
        auto c2w = gtC2wPoses[framesProcessed].to(torch::kCPU, torch::kFloat32).contiguous();
        pose_d_.SetInvM(c2w.data_ptr<float>());
        pose_d_.Coerce();
        if (beforeNextFusion) { std::function<void()> gate; gate.swap(beforeNextFusion); gate(); }
        check(gps_tsdf_process_frame(&state_, ptr<int16_t>(depth_mm_i16), pose_d_.GetM(), pose_d_.GetInvM(),
                                     current_stream()), "gps_tsdf_process_frame");
    }
    camPoses.push_back(pose_d_);
    camIntrincs.push_back(intrinsics_d);
    framesProcessed++;
    return &tracking_state_;
}

ITMTrackingState* TsdfEngine::ProcessFrame(ITMUChar4Image* rgbImage, ITMShortImage* rawDepthImage) {
    TORCH_CHECK(rgbImage && rawDepthImage, "ProcessFrame: null image");
    TORCH_CHECK(rgbImage->noDims.x == state_.width && rgbImage->noDims.y == state_.height &&
                    rawDepthImage->noDims.x == state_.width && rawDepthImage->noDims.y == state_.height,
                "ProcessFrame: image size differs from the engine's");
    const int64_t P = (int64_t)state_.width * state_.height;
    const int slot = framesProcessed & 1;
    auto on_device = [&](const torch::Tensor& dev, const torch::Tensor& host, torch::Tensor& stage, int64_t bytes) {
        if (dev.defined()) return dev;
        TORCH_CHECK(host.defined(), "ProcessFrame: image has neither a host nor a device copy");
        if (!stage.defined()) stage = torch::empty({bytes}, u8(device_));
        // view->rgb->SetFrom(rgbImage, CPU_TO_CUDA) (ITMViewBuilder_CUDA.cu:61-62): pinned host -> HBM on the frame's stream
        TORCH_CHECK(hipMemcpyAsync(stage.data_ptr(), host.data_ptr(), (size_t)bytes, hipMemcpyHostToDevice,
                                   (hipStream_t)current_stream()) == hipSuccess, "UpdateView: upload failed");
        return stage;
    };
    auto rgb = on_device(rgbImage->tensor(), rgbImage->host_tensor(), stage_rgb_[slot], P * 4);
    auto dep = on_device(rawDepthImage->tensor(), rawDepthImage->host_tensor(), stage_depth_[slot], P * 2);
    return ProcessFrame(rgb.view({state_.height, state_.width, 4}), dep.view(torch::kInt16).view({state_.height, state_.width}));
}

void TsdfEngine::runRaycast(ORUtils::SE3Pose* pose, ITMLib::ITMIntrinsics* intrinsics) {
    TORCH_CHECK(pose != nullptr, "runRaycast(NULL, NULL) (re-render of the live view, ITMBasicEngine.tpp:503-518) is not on "
                "SLAMPipeline's path and is not implemented; pass the pose of the view");
    gps_tsdf_state s;
    { std::lock_guard<std::mutex> lk(state_mu_); s = state_; }
    if (intrinsics) {
        TORCH_CHECK(intrinsics->imgSize.x == s.width && intrinsics->imgSize.y == s.height,
                    "runRaycast: the free-view render state has the depth camera's image size");
        s.fx = intrinsics->projectionParamsSimple.fx; s.fy = intrinsics->projectionParamsSimple.fy;
        s.cx = intrinsics->projectionParamsSimple.px; s.cy = intrinsics->projectionParamsSimple.py;
    }
    check(gps_tsdf_free_raycast(&s, pose->GetM(), pose->GetInvM(), current_stream()), "gps_tsdf_free_raycast");
}

// view k's render state, created (and initialised: gps_tsdf_view_init) on first use
void TsdfEngine::ensureView(int k, const gps_tsdf_state& s) {
    while ((int)views_.size() <= k) {
        const int64_t P = (int64_t)s.width * s.height;
        const auto I = torch::TensorOptions().dtype(torch::kInt32).device(device_);
        const auto F = torch::TensorOptions().dtype(torch::kFloat32).device(device_);
        std::unique_ptr<FreeView> v(new FreeView());
        v->visible_ids = torch::zeros({s.n_blocks}, I);
        v->minmax = torch::zeros({P * 2}, F);
        v->raycast = torch::zeros({P * 4}, F);
        v->colour = zeros_bytes(P * 4, device_);
        v->scratch = zeros_bytes(gps_tsdf_scratch_bytes(s.width, s.height, s.n_buckets, s.n_excess) + 16, device_);
        v->counters = torch::zeros({GPS_TSDF_N_COUNTERS}, I);
        v->image_p.reset(new ITMUChar4Image(s.width, s.height, v->colour));
        v->vertex_p.reset(new ITMFloat4Image(s.width, s.height, v->raycast));
        gps_tsdf_view r;
        memset(&r, 0, sizeof(r));
        r.minmax = fptr(v->minmax); r.counters = iptr(v->counters);
        check(gps_tsdf_view_init(&s, &r, current_stream()), "gps_tsdf_view_init");
        views_.push_back(std::move(v));
    }
}

void TsdfEngine::reserveViews(int n) {
    gps_tsdf_state s;
    { std::lock_guard<std::mutex> lk(state_mu_); s = state_; }
    if (n <= 0) return;
    ensureView(std::min(n, 12) - 1, s);
    // (the views are used from whatever stream a later runRaycastBatch runs on: finish their initialisation here)
    TORCH_CHECK(hipStreamSynchronize((hipStream_t)current_stream()) == hipSuccess, "reserveViews: hipStreamSynchronize");
}

void TsdfEngine::runRaycastBatch(const std::vector<ORUtils::SE3Pose>& poses, ITMLib::ITMIntrinsics* intrinsics,
                                 const std::vector<ViewMaps>* maps, const std::vector<ITMLib::ITMIntrinsics>* per_view_intrinsics) {
    const int n = (int)poses.size();
    if (n == 0) return;
    TORCH_CHECK(n <= 12, "runRaycastBatch: at most 12 views per call");
    gps_tsdf_state s;
    { std::lock_guard<std::mutex> lk(state_mu_); s = state_; }
    float fx = s.fx, fy = s.fy, cx = s.cx, cy = s.cy;
    if (intrinsics) {
        TORCH_CHECK(intrinsics->imgSize.x == s.width && intrinsics->imgSize.y == s.height,
                    "runRaycastBatch: the free-view render states have the depth camera's image size");
        fx = intrinsics->projectionParamsSimple.fx; fy = intrinsics->projectionParamsSimple.fy;
        cx = intrinsics->projectionParamsSimple.px; cy = intrinsics->projectionParamsSimple.py;
    }
    TORCH_CHECK(!maps || (int)maps->size() == n, "runRaycastBatch: one ViewMaps per pose");
    TORCH_CHECK(!per_view_intrinsics || (int)per_view_intrinsics->size() == n, "runRaycastBatch: one ITMIntrinsics per pose");
    std::vector<gps_tsdf_view> recs(n);
    memset(recs.data(), 0, sizeof(gps_tsdf_view) * (size_t)n);
    for (int k = 0; k < n; k++) {
        ensureView(k, s);
        FreeView& v = *views_[k];
        gps_tsdf_view& r = recs[k];
        memcpy(r.M, poses[k].GetM(), 64);
        memcpy(r.invM, poses[k].GetInvM(), 64);
        r.fx = fx; r.fy = fy; r.cx = cx; r.cy = cy;
        if (per_view_intrinsics) {  // runRaycast(pose, intrinsics) per view (slam_pipeline.cpp:367-379: camIntrincs[cam.id] or the camera's own)
            const ITMLib::ITMIntrinsics& in = (*per_view_intrinsics)[k];
            TORCH_CHECK(in.imgSize.x == s.width && in.imgSize.y == s.height, "runRaycastBatch: the free-view render states have the depth camera's image size");
            r.fx = in.projectionParamsSimple.fx; r.fy = in.projectionParamsSimple.fy;
            r.cx = in.projectionParamsSimple.px; r.cy = in.projectionParamsSimple.py;
        }
        r.visible_ids = iptr(v.visible_ids); r.minmax = fptr(v.minmax); r.raycast = fptr(v.raycast);
        r.colour = ptr<uint8_t>(v.colour); r.scratch = reinterpret_cast<int32_t*>(v.scratch.data_ptr()); r.counters = iptr(v.counters);
        if (maps) {
            const ViewMaps& o = (*maps)[k];
            memcpy(r.w2c, o.w2c_row_major, 64);
            r.color_map = o.color_map; r.vertex_map = o.vertex_map; r.confidence_map = o.confidence_map;
            r.depth_map = o.depth_map; r.depth_map_clamped = o.depth_map_clamped;
        }
    }
    if (!view_table_.defined()) view_table_ = zeros_bytes(gps_tsdf_view_table_bytes(12), device_);
    check(gps_tsdf_free_raycast_batch(&s, n, recs.data(), view_table_.data_ptr(), current_stream()), "gps_tsdf_free_raycast_batch");
}

bool TsdfEngine::checkRenderingBlocks() {
    auto c = counters_.cpu();
    const bool over = c.data_ptr<int32_t>()[GPS_TSDF_OVERFLOW] != 0;
    if (over && !warned_rendering_blocks_) {
        fprintf(stderr, "gps_slam_amd: more than 262144 rendering blocks in a CreateExpectedDepths call: the ray z-ranges of "
                        "that view were computed from a truncated block list (as the reference does, silently)\n");
        warned_rendering_blocks_ = true;
    }
    return over;
}

// ------------------------------------------------------------------------------------------------ meshing
std::pair<torch::Tensor, torch::Tensor> TsdfEngine::MeshScene(int64_t maxTriangles) {
    auto tri = torch::empty({maxTriangles, 7, 3}, f32(device_));
    auto counts = torch::zeros({2}, i64(device_));
    const int64_t ws_bytes = gps_tsdf_mesh_workspace_bytes(&state_);
    auto ws = torch::empty({ws_bytes}, u8(device_));
    check(gps_tsdf_mesh_scene(&state_, maxTriangles, fptr(tri), ptr<int64_t>(counts), ws.data_ptr(), ws_bytes, current_stream()),
          "gps_tsdf_mesh_scene");
    return {tri, counts};
}

int64_t TsdfEngine::SaveSceneToMesh(const char* fileName, int64_t maxTriangles) {
    auto mesh = MeshScene(maxTriangles);
    const int64_t n = mesh.second.cpu().data_ptr<int64_t>()[0];
    auto host = mesh.first.slice(0, 0, n).cpu().contiguous();
    const float* t = host.data_ptr<float>();
    FILE* f = fopen(fileName, "w");
    TORCH_CHECK(f != nullptr, "SaveSceneToMesh: cannot open ", fileName);
    fprintf(f, "ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
               "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face %d\n"
               "property list uchar int vertex_indices\nend_header\n", (int)(n * 3), (int)n);
    for (int64_t i = 0; i < n; i++) {
        const float* q = t + i * 21;
        for (int v = 0; v < 3; v++)
            fprintf(f, "%f %f %f %d %d %d\n", q[3 * v], q[3 * v + 1], q[3 * v + 2],
                    static_cast<unsigned char>(q[9 + 3 * v] * 255), static_cast<unsigned char>(q[9 + 3 * v + 1] * 255),
                    static_cast<unsigned char>(q[9 + 3 * v + 2] * 255));
    }
    for (int64_t i = 0; i < n; i++) fprintf(f, "3 %d %d %d\n", (int)(i * 3), (int)(i * 3 + 1), (int)(i * 3 + 2));
    fclose(f);
    return n;
}

// ------------------------------------------------------------------------------------------------ persistence
namespace {
std::string with_slash(const std::string& d) { return (!d.empty() && d.back() == '/') ? d : d + "/"; }

void write_block(const std::string& path, const torch::Tensor& dev, size_t elem_bytes) {
    auto h = dev.cpu().contiguous();
    std::ofstream fs(path.c_str(), std::ios::binary);
    TORCH_CHECK((bool)fs, "Could not open ", path, " for writing");
    const size_t count = (size_t)h.nbytes() / elem_bytes;  // MemoryBlockPersister::WriteBlock: dataSize, then the elements
    fs.write(reinterpret_cast<const char*>(&count), sizeof(size_t));
    fs.write(reinterpret_cast<const char*>(h.data_ptr()), (std::streamsize)h.nbytes());
    TORCH_CHECK((bool)fs, "Could not write memory block data: ", path);
}

void read_block(const std::string& path, torch::Tensor& dev, size_t elem_bytes) {
    std::ifstream fs(path.c_str(), std::ios::binary);
    TORCH_CHECK((bool)fs, "Could not open ", path, " for reading");
    size_t count = 0;
    TORCH_CHECK((bool)fs.read(reinterpret_cast<char*>(&count), sizeof(size_t)), "Could not read memory block size");
    TORCH_CHECK(count * elem_bytes == (size_t)dev.nbytes(), "Could not read data into a memory block of the wrong size: ", path);
    auto h = torch::empty_like(dev, dev.options().device(torch::kCPU));
    TORCH_CHECK((bool)fs.read(reinterpret_cast<char*>(h.data_ptr()), (std::streamsize)h.nbytes()), "Could not read memory block data");
    dev.copy_(h);
}
}  // namespace

void TsdfEngine::SaveToFile(const std::string& saveOutputDirectory) {
    const std::string d = with_slash(saveOutputDirectory), sc = d + "Scene/";
    mkdir(d.c_str(), 0755); mkdir((d + "Relocaliser/").c_str(), 0755); mkdir(sc.c_str(), 0755);
    auto c = counters_.cpu();
    const int32_t* ch = c.data_ptr<int32_t>();
    write_block(sc + "voxel.dat", vba_, 8);
    write_block(sc + "alloc.dat", vba_alloc_list_, 4);
    { std::ofstream ofs((sc + "vba.txt").c_str()); TORCH_CHECK((bool)ofs, "Could not open vba.txt"); ofs << ch[GPS_TSDF_LAST_FREE_BLOCK] << ' ' << (int64_t)state_.n_blocks * 512; }
    write_block(sc + "hash.dat", hash_, 16);
    write_block(sc + "excess.dat", excess_list_, 4);
    { std::ofstream ofs((sc + "last.txt").c_str()); TORCH_CHECK((bool)ofs, "Could not open last.txt"); ofs << ch[GPS_TSDF_LAST_FREE_EXCESS]; }
}

void TsdfEngine::LoadFromFile(const std::string& saveInputDirectory) {
    const std::string sc = with_slash(saveInputDirectory) + "Scene/";
    resetAll();
    read_block(sc + "voxel.dat", vba_, 8);
    read_block(sc + "alloc.dat", vba_alloc_list_, 4);
    int last_block = 0, last_excess = 0;
    int64_t alloc_size = 0;
    { std::ifstream ifs((sc + "vba.txt").c_str()); TORCH_CHECK((bool)ifs, "Could not open vba.txt for reading"); ifs >> last_block >> alloc_size; }
    read_block(sc + "hash.dat", hash_, 16);
    read_block(sc + "excess.dat", excess_list_, 4);
    { std::ifstream ifs((sc + "last.txt").c_str()); TORCH_CHECK((bool)ifs, "Count not open last.txt for reading"); ifs >> last_excess; }
    auto c = counters_.cpu();
    c.data_ptr<int32_t>()[GPS_TSDF_LAST_FREE_BLOCK] = last_block;
    c.data_ptr<int32_t>()[GPS_TSDF_LAST_FREE_EXCESS] = last_excess;
    counters_.copy_(c);
    check(gps_tsdf_rebuild_index(&state_, current_stream()), "gps_tsdf_rebuild_index");  // the table was written from outside
}
