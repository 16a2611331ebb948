#include "slam_pipeline.hpp"

#include <atomic>

#include <chrono>
#include <cstdlib>

#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <hip/hip_runtime_api.h>

#include <cmath>

using namespace gpsh;

namespace {
inline void hip_ok(hipError_t e, const char* what) { TORCH_CHECK(e == hipSuccess, what, ": ", hipGetErrorString(e)); }
struct MapStream { c10::hip::HIPStream s; };
// kind 0 / 1: torch's high- / normal-priority pool; 2 / 3 / 4: a stream of this library's own (hipStreamCreateWithPriority,
// non-blocking) at the lowest / highest / default priority, wrapped for the stream guards.  ONE stream per (device, kind) for the
// life of the process, shared by every pipeline: ROCm places a new stream on a hardware queue by what exists at that moment, and
// creating a stream costs ~10 ms of a blocked runtime -- creating the three once keeps the placement the same for every scene of
// a process and the cost out of all but the first.  Two pipelines alive at once share the streams: ordered, just not concurrent
// with each other.  (The 90 ms / 2.1 s holes in the first frames of round 5's early whole-sequence runs, first blamed on queue
// creation, were the container's CPU quota: LABBOOK section 14, dist_util.cap_host_threads.)
c10::hip::HIPStream make_stream(int kind) {
    const auto dev = c10::hip::current_device();
    if (kind == 0) return c10::hip::getStreamFromPool(/*isHighPriority=*/true, dev);
    if (kind == 1) return c10::hip::getStreamFromPool(/*isHighPriority=*/false, dev);
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> own;
    std::lock_guard<std::mutex> lock(mu);
    auto it = own.find({(int)dev, kind});
    if (it == own.end()) {
        int least = 0, greatest = 0;
        hip_ok(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        hipStream_t st = nullptr;
        hip_ok(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, kind == 2 ? least : kind == 3 ? greatest : 0), "hipStreamCreateWithPriority");
        // first use now, not in somebody's frame: the queue behind the stream exists when this returns
        void* word = nullptr;
        hip_ok(hipMalloc(&word, 256), "hipMalloc");
        hip_ok(hipMemsetAsync(word, 0, 256, st), "hipMemsetAsync");
        hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
        (void)hipFree(word);
        it = own.emplace(std::make_pair((int)dev, kind), st).first;
    }
    return c10::hip::getStreamFromExternal(it->second, dev);
}
}  // namespace
using torch::indexing::Slice;

unsigned long long getGPUMemoryUsage(int gpu_id) {
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return ~0ull;
    size_t free_b = 0, total_b = 0;
    const bool ok = hipSetDevice(gpu_id) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    (void)hipSetDevice(prev);
    return ok ? (unsigned long long)((total_b - free_b) / (1024 * 1024)) : ~0ull;
}

torch::Tensor computeNormalMap(const torch::Tensor& vertex_map_in) {
    auto vertex_map = vertex_map_in.contiguous();
    check_f32_dev(vertex_map, "vertex_map");
    const int H = (int)vertex_map.size(0), W = (int)vertex_map.size(1);
    auto out = torch::empty_like(vertex_map);
    check(gps_normal_map(W, H, fptr(vertex_map), fptr(out), current_stream()), "gps_normal_map");
    return out;
}

SLAMPipeline::SLAMPipeline(TsdfEngine* tsdf_engine, SLAMGaussianModel* model_, uint64_t seed, bool use_gt_pose)
    : main_engine(tsdf_engine), model(model_), rng_kf_(seed ^ 0x9E3779B97F4A7C15ull), rng_(seed), gen_(at::detail::createCPUGenerator(seed)) {
    if (use_gt_pose) main_engine->turnOffTracking();
    device = model->device;
    voxel_size = main_engine->getVoxelSize();
}

SLAMPipeline::SLAMPipeline(uint64_t seed) : rng_kf_(seed ^ 0x9E3779B97F4A7C15ull), rng_(seed), gen_(at::detail::createCPUGenerator(seed)) {}

void SLAMPipeline::setTsdfEngine(InfiniTAM::Engine::CLIEngine* engine) {
    tsdf_engine = engine;
    auto* be = dynamic_cast<ITMLib::ITMBasicEngine<ITMVoxel, ITMVoxelIndex>*>(engine->getMainEngine());
    TORCH_CHECK(be != nullptr, "setTsdfEngine: the main engine must be an ITMBasicEngine<ITMVoxel, ITMVoxelIndex>");
    main_engine = be;
    voxel_size = be->getVoxelSize();
    // CLIEngine::Shutdown() frees the engine (ownsInputs): finish what is in flight and let go of it first
    engine->beforeShutdown = [this] { detachTsdfEngine(); };
}

void SLAMPipeline::detachTsdfEngine() {
    if (!tsdf_engine) return;
    try { flush(); } catch (...) {}   // (a worker error has been, or will be, reported by the call that hit it)
    (void)hipDeviceSynchronize();
    if (main_engine) main_engine->beforeNextFusion = nullptr;
    if (tsdf_engine->beforeShutdown) tsdf_engine->beforeShutdown = nullptr;
    main_engine = nullptr;
    tsdf_engine = nullptr;
}

static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// host time the frame thread spent WAITING for the map worker inside the last processFrame call: for the previous update to finish
// (hand-over at a keyframe) and for this update's raycasts to be enqueued (gate of the next frame's fusion)
static thread_local double g_gate_wait_ms = 0.0, g_handover_wait_ms = 0.0;

// slam_pipeline.cpp:52-173 with its LOG_PIPELINE_TIME clock: `times` holds what the reference prints as "[PIPELINE AVG TIME]"
void SLAMPipeline::SLAMTrainCams(SLAMGaussianModel& model_, std::vector<Camera>& cams) {
    model = &model_;
    device = model->device;
    times = PipelineTimes();
    frame_ms.clear(); frame_wait_ms.clear();
    const double t0 = now_ms();
    for (size_t i = 0; i < cams.size(); i++) {
        const double tf = now_ms();
        processFrame((int)i, cams[i]);
        const double dt = now_ms() - tf;
        if (keep_frame_ms) { frame_ms.push_back((float)dt); frame_wait_ms.push_back((float)(g_gate_wait_ms + g_handover_wait_ms)); }
        if (i >= 30 && dt > times.max_frame_after_30) { times.max_frame_after_30 = dt; times.max_frame_id = (int)i; }
    }
    flush();
    hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
    times.slam_total = now_ms() - t0;
    times.frames = (int)cams.size();
    if (times.frames == 0) return;   // (nothing to average: the reference would print NaN into time_log.txt)
    // slam_pipeline.cpp:168-171: emptyCache(), then the device memory in use (what run/read_results.py reads as "GPU memory usage")
    c10::hip::HIPCachingAllocator::emptyCache();
    times.gpu_memory_mb = (long long)getGPUMemoryUsage((int)c10::hip::current_device());
    if (log_pipeline_time) {
        if (FILE* f = fopen((workspace_dir + "/time_log.txt").c_str(), "w")) {   // (the file run/read_results.py parses)
            fprintf(f, "[PIPELINE AVG TIME] GS num: %d, per frame fusion time: %f, localFrameRaycast time: %f, keyFrameRaycast time: %f, "
                       "initNewGaussians time: %f, localOptimize time: %f, FPS: %f\n", model->getGaussianNum(), times.per_frame / times.frames,
                    times.localFrameRaycast / times.frames, times.keyFrameRaycast / times.frames, times.initNewGaussians / times.frames,
                    times.localOptimize / times.frames, times.fps());
            fprintf(f, "GPU memory usage: %d MB\n", (int)times.gpu_memory_mb);
            fclose(f);
        }
        printf("GPU memory usage: %d MB\n", (int)times.gpu_memory_mb);
    }
    if (log_pipeline_time)
        printf("[PIPELINE AVG TIME] GS num: %d, per frame fusion time: %f, localFrameRaycast time: %f, keyFrameRaycast time: %f, "
               "initNewGaussians time: %f, localOptimize time: %f, FPS: %f\n", model->getGaussianNum(), times.per_frame / times.frames,
               times.localFrameRaycast / times.frames, times.keyFrameRaycast / times.frames, times.initNewGaussians / times.frames,
               times.localOptimize / times.frames, times.fps());
}

void SLAMPipeline::processFrame(int i, Camera& cam) {
    TORCH_CHECK(tsdf_engine != nullptr && model != nullptr, "processFrame(i, cam): setTsdfEngine() and a model first");
    processFrame(i, cam, torch::Tensor(), torch::Tensor());
}

void SLAMPipeline::loadConfig(const Config& c) {
    new_gs_sample_ratio = (float)c.get("new_gs_sample_ratio", new_gs_sample_ratio);
    color_error_thres = (float)c.get("color_error_thres", color_error_thres);
    localframe_cam_window_length = (int)c.get("localframe_cam_window_length", localframe_cam_window_length);
    localframe_cam_window_interval = (int)c.get("localframe_cam_window_interval", localframe_cam_window_interval);
    local_opt_iters = (int)c.get("local_opt_iters", local_opt_iters);
    local_opt_interval = (int)c.get("local_opt_interval", local_opt_interval);
    keyframe_theta_thres = (float)c.get("keyframe_theta_thres", keyframe_theta_thres);
    keyframe_trans_thres = (float)c.get("keyframe_trans_thres", keyframe_trans_thres);
    keyframe_select_max = (int)c.get("keyframe_select_max", keyframe_select_max);
    depth_vis_max = (float)c.get("depth_vis_max", depth_vis_max);
    depth_vis_min = (float)c.get("depth_vis_min", depth_vis_min);
    alpha_vis_max = (float)c.get("alpha_vis_max", alpha_vis_max);
    large_scale_thres = (float)c.get("large_scale_thres", large_scale_thres);
    small_scale_thres = (float)c.get("small_scale_thres", small_scale_thres);
    low_opac_thres = (float)c.get("low_opac_thres", low_opac_thres);
    scene_scale = (float)c.get("scene_scale", scene_scale);
    work_mode = c.gets("work_mode", work_mode);
    ssim_weight = (float)c.get("ssim_weight", ssim_weight);    // LOSS section of the configs (office0.yaml:37-39)
    depth_weight = (float)c.get("depth_weight", depth_weight);
    sample_method = c.gets("sample_method", sample_method);   // keyframe_sample_configs
    loss_thres = (float)c.get("loss_thres", loss_thres);
    TORCH_CHECK(sample_method == "random" || sample_method == "ours", "sample_method: 'random' or 'ours', got '", sample_method, "'");
}

// ------------------------------------------------------------------ raycast -> tensors (runRaycastByCam :362-415)
TensorDict SLAMPipeline::runRaycastByCam(const Camera& cam, bool use_cam_depth) {
    TensorDict m = raycastCam(cam, main_engine->camPoses);
    if (use_cam_depth) {  // slam_pipeline.cpp:405-408: the sensor depth instead of the raycast's (no call site of the loop uses it)
        TORCH_CHECK(cam.depth.defined(), "runRaycastByCam(use_cam_depth = true): the camera has no depth image");
        m["depth_map"] = cam.depth.contiguous().to(device);
        m["depth_map_clamped"] = RawGaussianModel::clampRefDepth(m["depth_map"]);
    }
    return m;
}

// slam_pipeline.cpp:367-379: a camera of the sequence is rendered with the intrinsics the engine stored for its frame
// (camIntrincs[cam.id]), any other camera with its own fx / fy / cx / cy
static ITMLib::ITMIntrinsics intrinsicsOf(const Camera& cam, const TsdfEngine* eng) {
    if (cam.id >= 0 && cam.id < (int)eng->camIntrincs.size()) return eng->camIntrincs[cam.id];
    ITMLib::ITMIntrinsics in;
    in.SetFrom(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy);
    return in;
}

TensorDict SLAMPipeline::raycastCam(const Camera& cam, const std::vector<ORUtils::SE3Pose>& poses, void** ev_out) {
    TsdfEngine* eng = main_engine;
    if (ev_out && !rc_stream_) beginAsyncRaycasts();  // (a caller that skipped localFrameRaycast: order the raycast stream now)
    ORUtils::SE3Pose pose;
    if (cam.id >= 0 && cam.id < (int)poses.size()) {
        pose = poses[cam.id];
    } else {
        auto c = cam.c2w.to(torch::kCPU, torch::kFloat32).contiguous();
        pose.SetInvM(c.data_ptr<float>());
        pose.Coerce();
    }
    const int H = cam.height, W = cam.width;
    const auto F = f32(device);
    // (the result tensors are allocated on the CONSUMER's stream, before the guard below: the caching allocator ties a block
    // to the stream that was current at allocation)
    TensorDict m;
    m["color_map"] = torch::empty({H, W, 3}, F);
    m["vertex_map"] = torch::empty({H, W, 3}, F);
    m["confidence_map"] = torch::empty({H, W, 1}, F);
    m["depth_map"] = torch::empty({H, W, 1}, F);
    m["depth_map_clamped"] = torch::empty({H, W, 1}, F);
    auto w2c = poseInv(cam.c2w.to(torch::kCPU, torch::kFloat32)).contiguous();  // poseInv(cam.c2w): dataset pose (:398)
    c10::optional<c10::hip::HIPStreamGuard> on_rc;
    if (ev_out) on_rc.emplace(static_cast<MapStream*>(rc_stream_)->s);
    ITMLib::ITMIntrinsics intr = intrinsicsOf(cam, eng);
    eng->runRaycast(&pose, &intr);
    check(gps_raycast_to_maps(W, H, reinterpret_cast<const float*>(eng->GetFreeVertex()->GetData(MEMORYDEVICE_CUDA)),
                              reinterpret_cast<const uint8_t*>(eng->GetFreeImage()->GetData(MEMORYDEVICE_CUDA)),
                              eng->getVoxelSize(), w2c.data_ptr<float>(), fptr(m["color_map"]), fptr(m["vertex_map"]),
                              fptr(m["confidence_map"]), fptr(m["depth_map"]), fptr(m["depth_map_clamped"]),
                              current_stream()), "gps_raycast_to_maps");
    if (ev_out) {
        if (rc_event_next_ == rc_events_.size()) {
            hipEvent_t ev;
            hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
            rc_events_.push_back(ev);
        }
        *ev_out = rc_events_[rc_event_next_++];
        hip_ok(hipEventRecord((hipEvent_t)*ev_out, c10::hip::getCurrentHIPStream().stream()), "hipEventRecord");
    }
    stats.raycasts++;
    return m;
}

void SLAMPipeline::beginAsyncRaycasts() {
    window_raycast_events_.clear(); opt_raycast_events_.clear();
    rc_event_next_ = 0;
    if (!async_raycasts) return;
    if (!rc_stream_) {
        rc_stream_ = new MapStream{make_stream(raycast_stream_kind)};
        hipEvent_t ev;
        hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        ev_rc_begin_ = ev;
    }
    // the raycast stream starts behind everything the current stream holds (the fusion of the keyframe; the previous
    // update's last readers of the engine's free-view scratch)
    hip_ok(hipEventRecord((hipEvent_t)ev_rc_begin_, c10::hip::getCurrentHIPStream().stream()), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(static_cast<MapStream*>(rc_stream_)->s.stream(), (hipEvent_t)ev_rc_begin_, 0), "hipStreamWaitEvent");
}

void SLAMPipeline::waitRaycast(void* ev) {
    if (ev) hip_ok(hipStreamWaitEvent(c10::hip::getCurrentHIPStream().stream(), (hipEvent_t)ev, 0), "hipStreamWaitEvent");
}

void SLAMPipeline::waitAllRaycasts() {
    if (last_raycast_event_) { waitRaycast(last_raycast_event_); return; }   // (an adopted job's views: its last batch)
    if (rc_event_next_ > 0) waitRaycast(rc_events_[rc_event_next_ - 1]);  // one stream: the last event covers all
}

// localFrameRaycast + keyFrameRaycast of the keyframe just fused, by the frame thread, into a job the map worker adopts later
// (pipeline_raycasts): the same two batches, the same draws, the same tensors as raycastWindow + raycastKeyframes.
void SLAMPipeline::buildUpdateViews(MapJob& out) {
    const std::vector<ORUtils::SE3Pose>& poses = main_engine->camPoses;
    void** evs = job_events_[job_parity_];
    job_parity_ ^= 1;
    std::vector<const Camera*> cams;
    for (const Camera& cam : localframe_cam_window) cams.push_back(&cam);
    void* ev = nullptr;
    const double t0 = now_ms();
    for (TensorDict& m : raycastCams(cams, poses, &ev, evs[0])) { out.window_raycasts.push_back(m); out.window_events.push_back(evs[0]); }
    const double t1 = now_ms();
    times.localFrameRaycast += t1 - t0;
    out.opt_cams.assign(localframe_cam_window.begin(), localframe_cam_window.end());
    out.window_len = localframe_cam_window.size();
    out.opt_raycasts.assign(out.window_raycasts.begin(), out.window_raycasts.end());
    out.opt_events = out.window_events;
    out.last_event = cams.empty() ? nullptr : evs[0];
    const int n = sample_method == "random" ? std::min<int>(keyframe_select_max, (int)keyframe_cam_list.size()) : 0;   // :538
    if (n > 0) {
        RandomSelector<Camera> sel(keyframe_cam_list, rng_kf_);
        std::vector<const Camera*> kcams;
        for (int k = 0; k < n; k++) {
            const Camera* cam = sel.getNext().second;
            out.opt_cams.push_back(*cam);
            kcams.push_back(cam);
        }
        for (TensorDict& m : raycastCams(kcams, poses, &ev, evs[1])) { out.opt_raycasts.push_back(m); out.opt_events.push_back(evs[1]); }
        out.last_event = evs[1];
    }
    times.keyFrameRaycast += now_ms() - t1;
    out.views_ready = true;
}

// ------------------------------------------------------------------ frame bookkeeping (updateFrameList :319-360)
void SLAMPipeline::updateFrameList() {
    if (curr_frame_id == 0) return;
    if (curr_frame_id % localframe_cam_window_interval == 0) {
        localframe_cam_window.push_back(curr_cam);
        if ((int)localframe_cam_window.size() == localframe_cam_window_length + 1) localframe_cam_window.pop_front();
    }
    bool is_key = false;
    if (keyframe_cam_list.empty()) {
        is_key = true;
    } else {
        const Camera& last = keyframe_cam_list.back();
        auto a = last.c2w_slam.to(torch::kCPU, torch::kFloat32).contiguous();
        auto b = curr_cam.c2w_slam.to(torch::kCPU, torch::kFloat32).contiguous();
        const float* A = a.data_ptr<float>();
        const float* B = b.data_ptr<float>();
        float tr = 0.f;  // trace(Rp^T Rc)
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) tr += A[4 * k + r] * B[4 * k + r];
        const float cos_t = std::max(-1.0f, std::min(1.0f, (tr - 1.f) / 2.f));
        const float theta = std::acos(cos_t) * 180.0f / (float)M_PI;
        const float dx = A[3] - B[3], dy = A[7] - B[7], dz = A[11] - B[11];
        const float trans = std::sqrt(dx * dx + dy * dy + dz * dz);
        is_key = theta > keyframe_theta_thres || trans > keyframe_trans_thres;
    }
    if (is_key) {
        keyframe_cam_list.push_back(curr_cam);
        std::lock_guard<std::mutex> lk(loss_mu_);
        keyframe_loss_dict[curr_cam.id] = {0.1f, (float)curr_frame_id, 0.f, 0.f, 0.f};   // slam_pipeline.cpp:355
    }
}

void SLAMPipeline::localFrameRaycast() { raycastWindow(localframe_cam_window, main_engine->camPoses); }
void SLAMPipeline::keyFrameRaycast() { update_frame_id_ = curr_frame_id; raycastKeyframes(localframe_cam_window, keyframe_cam_list, main_engine->camPoses); }
void SLAMPipeline::initNewGaussians(TensorDict& rm) { initNewGaussiansFor(rm, curr_cam); }

// runRaycastByCam for several cameras of ONE volume state: one batched free-view chain (TsdfEngine::runRaycastBatch) instead
// of one chain per camera, then each view's tensor glue.  Same tensors as raycastCam per camera; one event covers them all.
std::vector<TensorDict> SLAMPipeline::raycastCams(const std::vector<const Camera*>& cams,
                                                  const std::vector<ORUtils::SE3Pose>& poses, void** ev_out, void* use_event) {
    std::vector<TensorDict> out;
    if (cams.empty()) return out;
    TsdfEngine* eng = main_engine;
    const auto F = f32(device);
    if (ev_out && !rc_stream_ && !use_event) beginAsyncRaycasts();  // (keyFrameRaycast() without a preceding localFrameRaycast())
    // Whose pool the result tensors come from.  They are WRITTEN on the raycast stream and READ on the consumer's (the stream
    // current here: the map stream, or the frames stream in the sequential schedule).  Rounds 4-5 allocated them with the
    // consumer's stream current -- fine while the worker enqueued the raycasts itself (the raycast stream was ordered behind
    // the map stream), a write-after-read hazard once the FRAME thread enqueues update k+1's views while the worker is still
    // running update k on the map stream: the caching allocator may hand out a block the worker freed a moment ago that queued
    // map-stream kernels still read, and the raycast stream only waits for the frame's fusion (round-5 advisor finding).  Now:
    // allocated with the RAYCAST stream current (a block of that pool is only reused in that stream's order) and the consumer
    // registered with recordStream, so that a freed block waits for the consumer's queued work before the raycast stream gets
    // it again; the consumer itself waits on the batch events as before.  (Only the allocations run under the raycast stream's
    // guard: the pose glue below must not inherit it.)
    const c10::hip::HIPStream consumer = c10::hip::getCurrentHIPStream();
    auto result = [&](int64_t h, int64_t w, int64_t c) {
        c10::optional<c10::hip::HIPStreamGuard> alloc_on_rc;
        if (ev_out) alloc_on_rc.emplace(static_cast<MapStream*>(rc_stream_)->s);
        torch::Tensor t = torch::empty({h, w, c}, F);
        // (the allocator's own entry point: Tensor::record_stream wants a c10::Stream of the masquerading "cuda" device type)
        if (ev_out) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), consumer);
        return t;
    };
    if (!raycast_pool_warm_) {
        // The result tensors of an update (5 per view, ~11 MB per 640x480 view) are allocated on the consumer's stream and freed
        // when the next update replaces them, so in steady state the caching allocator hands the same blocks out again -- but
        // while the keyframe list is still filling every update has one view more than the last, i.e. a fresh hipMalloc under
        // the allocator's lock in the middle of an update (measured: the frame thread's own 3.7 MB image allocation then waited
        // 3-6 ms for that lock in about one run in six).  Take the full set once, up front, and give it back to the cache.
        // (two sets where the frame thread enqueues update k+1's views while update k's are still being read)
        std::vector<torch::Tensor> warm;
        const int64_t H = cams[0]->height, W = cams[0]->width;
        const int sets = mapping_thread && pipeline_raycasts && async_raycasts ? 2 : 1;
        for (int k = 0; k < sets * (localframe_cam_window_length + keyframe_select_max); k++) {
            warm.push_back(result(H, W, 3)); warm.push_back(result(H, W, 3));
            warm.push_back(result(H, W, 1)); warm.push_back(result(H, W, 1));
            warm.push_back(result(H, W, 1));
        }
        raycast_pool_warm_ = true;
    }
    std::vector<ORUtils::SE3Pose> view_poses(cams.size());
    std::vector<ITMLib::ITMIntrinsics> view_intr(cams.size());
    std::vector<torch::Tensor> w2c(cams.size());
    for (size_t k = 0; k < cams.size(); k++) {
        const Camera& cam = *cams[k];
        if (cam.id >= 0 && cam.id < (int)poses.size()) {
            view_poses[k] = poses[cam.id];
        } else {
            auto c = cam.c2w.to(torch::kCPU, torch::kFloat32).contiguous();
            view_poses[k].SetInvM(c.data_ptr<float>());
            view_poses[k].Coerce();
        }
        view_intr[k] = intrinsicsOf(cam, eng);
        TensorDict m;
        m["color_map"] = result(cam.height, cam.width, 3);
        m["vertex_map"] = result(cam.height, cam.width, 3);
        m["confidence_map"] = result(cam.height, cam.width, 1);
        m["depth_map"] = result(cam.height, cam.width, 1);
        m["depth_map_clamped"] = result(cam.height, cam.width, 1);
        out.push_back(m);
        w2c[k] = poseInv(cam.c2w.to(torch::kCPU, torch::kFloat32)).contiguous();  // poseInv(cam.c2w): dataset pose (:398)
    }
    c10::optional<c10::hip::HIPStreamGuard> on_rc;
    if (ev_out) on_rc.emplace(static_cast<MapStream*>(rc_stream_)->s);
    constexpr size_t kMaxBatch = 12;  // views per gps_tsdf_free_raycast_batch call; a later chunk reuses the render states in stream order
    for (size_t base = 0; base < cams.size(); base += kMaxBatch) {
        const size_t cnt = std::min(kMaxBatch, cams.size() - base);
        std::vector<TsdfEngine::ViewMaps> maps(cnt);
        for (size_t k = 0; k < cnt; k++) {
            TensorDict& m = out[base + k];
            maps[k] = {w2c[base + k].data_ptr<float>(), fptr(m["color_map"]), fptr(m["vertex_map"]), fptr(m["confidence_map"]),
                       fptr(m["depth_map"]), fptr(m["depth_map_clamped"])};
        }
        // (the tensor glue of every view -- gps_raycast_to_maps -- is written by the batch's last kernel)
        const std::vector<ITMLib::ITMIntrinsics> intr(view_intr.begin() + base, view_intr.begin() + base + cnt);
        eng->runRaycastBatch(std::vector<ORUtils::SE3Pose>(view_poses.begin() + base, view_poses.begin() + base + cnt), nullptr, &maps, &intr);
    }
    if (ev_out) {
        if (use_event) {
            *ev_out = use_event;
        } else {
            if (rc_event_next_ == rc_events_.size()) {
                hipEvent_t ev;
                hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
                rc_events_.push_back(ev);
            }
            *ev_out = rc_events_[rc_event_next_++];
        }
        hip_ok(hipEventRecord((hipEvent_t)*ev_out, c10::hip::getCurrentHIPStream().stream()), "hipEventRecord");
    }
    stats.raycasts += (int64_t)cams.size();
    return out;
}

void SLAMPipeline::raycastWindow(const std::deque<Camera>& window, const std::vector<ORUtils::SE3Pose>& poses) {
    localframe_raycast_window.clear();
    beginAsyncRaycasts();
    std::vector<const Camera*> cams;
    for (const Camera& cam : window) cams.push_back(&cam);
    void* ev = nullptr;
    for (TensorDict& m : raycastCams(cams, poses, async_raycasts ? &ev : nullptr)) {
        localframe_raycast_window.push_back(m);
        window_raycast_events_.push_back(ev);
    }
}

void SLAMPipeline::raycastKeyframes(const std::deque<Camera>& window, const std::vector<Camera>& keyframes,
                                    const std::vector<ORUtils::SE3Pose>& poses) {
    opt_cam_list.assign(window.begin(), window.end());
    opt_window_len_ = window.size(); opt_frame_id_ = update_frame_id_;   // (what checkKeyFrameError indexes / stamps with)
    opt_raycast_list.assign(localframe_raycast_window.begin(), localframe_raycast_window.end());
    opt_raycast_events_ = window_raycast_events_;
    const int n = sample_method == "random" ? std::min<int>(keyframe_select_max, (int)keyframes.size()) : 0;   // :538
    if (n == 0) return;
    RandomSelector<Camera> sel(keyframes, rng_kf_);
    std::vector<const Camera*> cams;
    for (int k = 0; k < n; k++) {
        const Camera* cam = sel.getNext().second;
        opt_cam_list.push_back(*cam);
        cams.push_back(cam);
    }
    void* ev = nullptr;
    for (TensorDict& m : raycastCams(cams, poses, async_raycasts ? &ev : nullptr)) {
        opt_raycast_list.push_back(m);
        opt_raycast_events_.push_back(ev);
    }
}

// localFrameRaycast + keyFrameRaycast of one keyframe update as ONE batch (what the keyframe-step functions call): same lists,
// same random draws, same tensors; the window's and the keyframes' views share the launches.
void SLAMPipeline::raycastWindowAndKeyframes(const std::deque<Camera>& window, const std::vector<Camera>& keyframes,
                                             const std::vector<ORUtils::SE3Pose>& poses) {
    localframe_raycast_window.clear();
    beginAsyncRaycasts();
    std::vector<const Camera*> cams;
    for (const Camera& cam : window) cams.push_back(&cam);
    opt_cam_list.assign(window.begin(), window.end());
    opt_window_len_ = window.size(); opt_frame_id_ = update_frame_id_;   // (what checkKeyFrameError indexes / stamps with)
    const int n = sample_method == "random" ? std::min<int>(keyframe_select_max, (int)keyframes.size()) : 0;   // :538
    RandomSelector<Camera> sel(keyframes, rng_kf_);
    for (int k = 0; k < n; k++) {
        const Camera* cam = sel.getNext().second;
        opt_cam_list.push_back(*cam);
        cams.push_back(cam);
    }
    void* ev = nullptr;
    std::vector<TensorDict> res = raycastCams(cams, poses, async_raycasts ? &ev : nullptr);
    for (size_t k = 0; k < window.size(); k++) {
        localframe_raycast_window.push_back(res[k]);
        window_raycast_events_.push_back(ev);
    }
    opt_raycast_list.assign(localframe_raycast_window.begin(), localframe_raycast_window.end());
    opt_raycast_events_ = window_raycast_events_;
    for (size_t k = window.size(); k < res.size(); k++) {
        opt_raycast_list.push_back(res[k]);
        opt_raycast_events_.push_back(ev);
    }
}

// ------------------------------------------------------------------ initNewGaussians :450-526
void SLAMPipeline::initNewGaussiansFor(TensorDict& rm, const Camera& cam) {
    torch::NoGradGuard no_grad;
    if (!window_raycast_events_.empty()) waitRaycast(window_raycast_events_.back());  // rm is the newest window camera's result
    const auto &depth = rm.at("depth_map"), &color = rm.at("color_map"), &vertex = rm.at("vertex_map");
    int frame_num = local_opt_interval;
    // valid = depth in (min, max) & vertex.sum(2) != 0;  mask = mean|src - image| > thres & valid [& alpha < max]: one launch
    // (gps_new_gaussian_mask) instead of the reference's ~12 tensor ops (:455-480)
    const int64_t H = cam.image.size(0), W = cam.image.size(1);
    auto image = cam.image.contiguous(), depth_c = depth.contiguous(), vertex_c = vertex.contiguous();
    auto mask = torch::empty({H, W, 1}, torch::TensorOptions().dtype(torch::kBool).device(depth.device()));
    torch::Tensor src = color.contiguous(), alpha;
    if (model->getGaussianNum() == 0) {
        frame_num += 1;
    } else {
        auto res = model->forward(cam, depth, color);
        src = res.at("rgb").contiguous();
        alpha = res.at("alpha").contiguous();
    }
    check(gps_new_gaussian_mask((int)W, (int)H, fptr(depth_c), fptr(src), fptr(image), fptr(vertex_c), fptr(alpha), depth_vis_min,
                                depth_vis_max, color_error_thres, alpha_vis_max, reinterpret_cast<uint8_t*>(mask.data_ptr<bool>()),
                                current_stream()), "gps_new_gaussian_mask");
    rm["normal_map"] = computeNormalMap(vertex);
    stats.added += model->addGaussians(cam, rm, mask, new_gs_sample_ratio, frame_num, gen_);
}

// ------------------------------------------------------------------ checkKeyFrameError :293-317
// The loss record of every history keyframe in the current optimise list (the entries behind the local window's).
void SLAMPipeline::checkKeyFrameError() {
    torch::NoGradGuard no_grad;
    Config wc;
    wc.num["ssim_weight"] = ssim_weight; wc.num["depth_weight"] = depth_weight;
    // The window length and frame number are those of the update the lists belong to (recorded when the lists were built): with
    // overlapped / threaded mapping this check runs while the frame thread keeps pushing into localframe_cam_window and
    // advancing curr_frame_id, or one update late.  keyframe_loss_dict is shared with updateFrameList (frame thread): loss_mu_.
    for (size_t k = opt_window_len_; k < opt_cam_list.size(); k++) {
        const Camera& cam = opt_cam_list[k];
        TensorDict& rc = opt_raycast_list[k];
        if (k < opt_raycast_events_.size()) waitRaycast(opt_raycast_events_[k]);
        auto res = model->forward(cam, rc.at("depth_map"), rc.at("color_map"));
        auto loss = model->computeLoss(res, cam, wc, rc.at("depth_map") > 0);
        const float total = loss.at("total").item<float>();
        const float confidence_mean = rc.at("confidence_map").mean().item<float>();
        std::lock_guard<std::mutex> lk(loss_mu_);
        auto it = keyframe_loss_dict.find(cam.id);
        float opt_count = it != keyframe_loss_dict.end() && it->second.size() > 3 ? it->second[3] : 0.f;
        if (total > loss_thres) opt_count += 1.f;
        keyframe_loss_dict[cam.id] = {total, (float)opt_frame_id_, confidence_mean, opt_count};
    }
}

// ------------------------------------------------------------------ localOptimize :195-289
void SLAMPipeline::localOptimize() {
    localOptimizeBegin();
    optimizeIterations(opt_pending_);
}

void SLAMPipeline::localOptimizeBegin() {
    opt_pending_ = 0;
    if (model->getGaussianNum() == 0) return;
    model->initOptimizers(-1, scene_scale);
    opt_loader_.reset(new RandomSelector<Camera>(opt_cam_list, rng_));
    opt_peek_valid_ = false;
    opt_pending_ = local_opt_iters;
}

// the next `count` iterations of the current localOptimize (cameras drawn in the same order as the all-at-once loop)
void SLAMPipeline::optimizeIterations(int count) {
    for (; count > 0 && opt_pending_ > 0; count--, opt_pending_--) {
        // the camera of THIS iteration was drawn one iteration early (same draws in the same order), so that the previous
        // iteration's backward kernel could run its preprocessing forward in its tail (RawGaussianModel::trainStep next_cam)
        auto pick = opt_peek_valid_ ? opt_peek_ : opt_loader_->getNext();
        opt_peek_valid_ = false;
        const Camera& cam = *pick.second;
        TensorDict& rc = opt_raycast_list[pick.first];
        if (pick.first < (int)opt_raycast_events_.size()) waitRaycast(opt_raycast_events_[pick.first]);
        const Camera* next_cam = nullptr;
        if (prefetch_next_preprocess && opt_pending_ > 1 && !(ssim_weight > 0 || depth_weight > 0)) {
            opt_peek_ = opt_loader_->getNext();
            opt_peek_valid_ = true;
            next_cam = opt_peek_.second;
        }
        if (ssim_weight > 0 || depth_weight > 0) {
            // losses beyond L1: the reference's own sequence (slam_pipeline.cpp:247-254) through the autograd route
            Config wc;
            wc.num["ssim_weight"] = ssim_weight; wc.num["depth_weight"] = depth_weight;
            auto res = model->forward(cam, rc.at("depth_map"), rc.at("color_map"));
            auto loss = model->computeLoss(res, cam, wc);
            loss.at("total").backward();
            model->optimizersStep();
            model->optimizersZeroGrad();
        } else {
            model->trainStep(cam, rc.at("depth_map"), rc.at("color_map"), rc.at("depth_map_clamped"), next_cam);
        }
        stats.opt_iters++;
    }
}

// ------------------------------------------------------------------ removeRedundantGs :564-586
void SLAMPipeline::removeRedundantGs() {
    torch::NoGradGuard no_grad;
    if (model->getGaussianNum() == 0) return;
    // delete = max real scale < small or > large, or real opacity < low (:566-575): one launch; the compaction of the keep
    // mask synchronises with the map stream once (the reference syncs 5 times here for its printf)
    const int64_t N = model->getGaussianNum();
    const auto dev = model->getMeans().device();
    auto del = torch::empty({N}, torch::TensorOptions().dtype(torch::kBool).device(dev));
    auto keep = torch::empty({N}, torch::TensorOptions().dtype(torch::kBool).device(dev));
    check(gps_prune_mask((int)N, fptr(model->getScales()), fptr(model->getOpacities()), small_scale_thres, large_scale_thres,
                         low_opac_thres, reinterpret_cast<uint8_t*>(del.data_ptr<bool>()),
                         reinterpret_cast<uint8_t*>(keep.data_ptr<bool>()), current_stream()), "gps_prune_mask");
    const int64_t kept = model->pruneKeep(keep);
    // the host is synchronised with the map stream right here: look at the capacity flags the kernels cannot raise as exceptions
    model->checkBinningCapacity();
    main_engine->checkRenderingBlocks();
    stats.pruned += N - kept;
}

// ------------------------------------------------------------------ renderEvalImgs :588-695 (tensors instead of image files)
std::vector<TensorDict> SLAMPipeline::renderEvalImgs(const std::vector<Camera>& cams, const std::vector<std::string>& names) {
    torch::NoGradGuard no_grad;
    flush();
    std::vector<TensorDict> out;
    for (const Camera& cam : cams) {
        TensorDict r;
        TORCH_CHECK(cam.on_device(), "Camera::toGPU() must run before the camera is rendered");
        auto rc = runRaycastByCam(cam, false);
        r["raycast_color"] = rc.at("color_map");
        r["raycast_depth"] = rc.at("depth_map");
        // what the reference writes to disk, as it quantises it (cv_utils.cpp:57-76 tensorToImage: (t * 255.0).toType(kU8),
        // truncation; :79-101 tensorToDepth: convertTo(CV_16UC1, 1000) = saturate_cast<ushort>(round-half-even(d * 1000)))
        r["raycast_color_u8"] = (r["raycast_color"] * 255.0).toType(torch::kUInt8);
        r["raycast_depth_u16"] = torch::round(r["raycast_depth"] * 1000.0f).clamp(0.0, 65535.0).to(torch::kInt32);
        if (cam.image.defined()) r["gt_u8"] = (cam.image.to(device) * 255.0).toType(torch::kUInt8);
        if (model->getGaussianNum() > 0) {
            auto res = model->forward(cam, rc.at("depth_map"), rc.at("color_map"));
            for (const std::string& name : names) {
                if (name == "rgb") {
                    r["rgb"] = torch::clamp(res.at("rgb"), 0, 1);
                    r["rgb_u8"] = (r["rgb"] * 255.0).toType(torch::kUInt8);   // render/<frame>.color.jpg before the JPEG encoder
                    if (cam.image.defined()) {
                        auto mse = torch::mean(torch::square(r["rgb"] - cam.image.to(device)));
                        r["psnr"] = -10.0 * torch::log10(mse);
                        // the number the reference reports: scripts/metric.py reads the 8-bit render / gt files back as
                        // u8 / 255 (to_tensor) and takes 20 log10(1 / sqrt(mse)) (scripts/utils/image_utils.py:19-21); the lossy
                        // JPEG encoder in between is I/O outside this library
                        auto a = r["rgb_u8"].to(torch::kFloat32) / 255.0f, b = r["gt_u8"].to(torch::kFloat32) / 255.0f;
                        r["psnr_u8"] = 20.0 * torch::log10(1.0 / torch::sqrt(torch::mean(torch::square(a - b))));
                    }
                } else if (name == "alpha" || name == "depth") {
                    r[name] = res.at(name).clone();
                }
            }
            if (r.count("rgb")) r["rgb"] = r["rgb"].clone();  // forward() returns views of buffers the next camera overwrites
        }
        out.push_back(r);
    }
    return out;
}

// ------------------------------------------------------------------ one SLAM frame (body of SLAMTrainCams :69-132)
static std::atomic<double> g_job_post_ms{0.0};  // (debug aid only: when the frame thread woke the mapping thread)

void SLAMPipeline::processFrameImpl(int i, Camera& cam, const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16) {
    curr_frame_id = i;
    const double tt0 = now_ms();
    g_gate_wait_ms = g_handover_wait_ms = 0.0;
    if (!main_engine->trackingActive && (int)main_engine->gtC2wPoses.size() <= main_engine->framesProcessed)
        main_engine->gtC2wPoses.push_back(cam.c2w);
    ITMTrackingState* ts;
    torch::Tensor frame_rgba;
    if (rgb_u8.defined()) {
        ts = main_engine->ProcessFrame(rgb_u8, depth_mm_i16);
    } else {
        // slam_pipeline.cpp:77-78: the CLIEngine owns the sequence (host memory) and uploads the frame
        TORCH_CHECK(tsdf_engine != nullptr && i == tsdf_engine->currentFrameNo, "frame ", i, " is not the CLIEngine's next frame");
        tsdf_engine->ProcessFrame();
        ts = main_engine->GetTrackingState();
        // (no float copy of the image kept for this camera: it is derived below, with the pose pack, from the uchar4 frame
        // UpdateView just put into HBM -- 3 of its 4 bytes per pixel instead of uploading 12 more as Camera::toGPU would)
        if (!cam.image.defined()) frame_rgba = main_engine->currentRgb();  // [H,W,4] u8, contiguous
    }
    if (trace_frames) {
        trace_live.push_back(main_engine->GetLiveVertex()->tensor().clone());
        trace_counters.push_back(main_engine->counters().clone());
        auto t = torch::empty({2, 16}, torch::kFloat32);
        const ORUtils::SE3Pose& p = main_engine->camPoses.back();
        for (int k = 0; k < 16; k++) { t[0][k] = p.GetM()[k]; t[1][k] = p.GetInvM()[k]; }
        trace_poses.push_back(t);
    }
    // est_pose = pose_d->GetInvM() (:81-82): ORUtils column-major -> row-major tensor
    auto est = torch::empty({4, 4}, torch::kFloat32);
    const float* invM = ts->pose_d->GetInvM();
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) est.data_ptr<float>()[4 * r + c] = invM[4 * c + r];
    cam.c2w_slam = est;
    // `curr_cam = cams[i]; curr_cam.toGPU();` (:83-84): a COPY of the caller's camera goes to the device -- the caller's keeps the
    // pose and no device tensors, so a frame's 12-byte-per-pixel image lives exactly as long as the pipeline's lists hold it
    // (held by the caller's camera it stayed for the whole run: 3.7 MB per frame at 640x480, i.e. a fresh 20 MB hipMalloc by the
    // frame thread every fifth frame instead of the caching allocator handing the previous frame's block out again)
    curr_cam = cam;
    curr_cam.invalidate();
    const double tt1 = now_ms();
    curr_cam.toGPU(device, frame_rgba);
    const double tt2 = now_ms();
    if (frame_rgba.defined()) tsdf_engine->markConsumed();  // the staging slot's last reader is the conversion toGPU just enqueued
    updateFrameList();
    stats.frames++;
    const double tt_lists = now_ms();
    times.per_frame += tt_lists - tt0;   // the reference's perFrame_start .. perFrame_end (slam_pipeline.cpp:73-96)
    if (work_mode == "recon") return;
    if (i % local_opt_interval == 0 && i > 0) {
        if (!views_reserved_) {  // every free-view render state an update can need, once, before the first update
            main_engine->reserveViews(localframe_cam_window_length + keyframe_select_max);
            views_reserved_ = true;
        }
        if (overlap_mapping && mapping_thread) keyframeStepThreaded();
        else if (overlap_mapping) keyframeStepOverlapped();
        else keyframeStep();
        times.keyframe_step += now_ms() - tt_lists;
    }
    const double thr = frame_report_ms, tt3 = now_ms();
    if (thr >= 0.0 && tt3 - tt0 > thr)
        fprintf(stderr, "[pipe] frame %d: %.3f ms = engine %.3f (tracker %.3f, fusion enqueue %.3f, gate wait %.3f) + toGPU %.3f + lists / keyframe step %.3f (hand-over wait %.3f)\n",
                i, tt3 - tt0, tt1 - tt0, main_engine->trackDiag(14), main_engine->trackDiag(15), g_gate_wait_ms, tt2 - tt1, tt3 - tt2, g_handover_wait_ms);
}

// ------------------------------------------------------------------ tracking / mapping overlap (see slam_pipeline.hpp)

void SLAMPipeline::processFrame(int i, Camera& cam, const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16) {
    // room for the tracker's evaluations beside the strip backward only while the two chains really run side by side (the strip
    // kernel alone is faster without the reserve: include/gps_slam_hip.h)
    gps_set_frame_chain_reserve(overlap_mapping ? frame_chain_reserve : 0);
    if (!overlap_mapping) { processFrameImpl(i, cam, rgb_u8, depth_mm_i16); return; }
    // overlap: frames run on a HIGH-priority stream of their own, so that the short, latency-bound tracker kernels are
    // dispatched ahead of the map stream's long rasterizer kernels; ordered after the caller's stream (inputs), and the
    // caller's stream is re-joined in flush()
    ensureStreams();
    const hipStream_t caller = c10::hip::getCurrentHIPStream().stream();
    c10::hip::HIPStream& fs = static_cast<MapStream*>(frame_stream_)->s;
    if (caller != fs.stream()) {
        hip_ok(hipEventRecord((hipEvent_t)ev_caller_, caller), "hipEventRecord");
        hip_ok(hipStreamWaitEvent(fs.stream(), (hipEvent_t)ev_caller_, 0), "hipStreamWaitEvent");
    }
    rethrowWorkerError();
    pumpMapping(pump_iters_per_frame);  // keep the map stream fed before this thread starts spinning on the tracker
    c10::hip::HIPStreamGuard guard(fs);
    processFrameImpl(i, cam, rgb_u8, depth_mm_i16);
}

void SLAMPipeline::ensureStreams() {
    if (map_stream_) return;
    map_stream_ = new MapStream{make_stream(map_stream_kind)};
    frame_stream_ = new MapStream{make_stream(frame_stream_kind)};
    for (void** e : {&ev_frame_, &ev_raycasts_, &ev_map_, &ev_caller_, &job_events_[0][0], &job_events_[0][1], &job_events_[1][0], &job_events_[1][1]}) {
        hipEvent_t ev;
        hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        *e = ev;
    }
}

void SLAMPipeline::keyframeStep() {
    // two batches: initNewGaussians only needs the window's views and starts while the keyframes' views render
    const double t0 = now_ms();
    localFrameRaycast();
    const double t1 = now_ms();
    keyFrameRaycast();
    const double t2 = now_ms();
    initNewGaussians(localframe_raycast_window.back());
    const double t3 = now_ms();
    localOptimize();
    const double t4 = now_ms();
    removeRedundantGs();
    const double t5 = now_ms();
    if (sample_method == "ours") checkKeyFrameError();   // slam_pipeline.cpp:130-131
    waitAllRaycasts();  // the next frame's fusion modifies the volume (results no iteration drew would still be in flight)
    times.localFrameRaycast += t1 - t0; times.keyFrameRaycast += t2 - t1; times.initNewGaussians += t3 - t2;
    times.localOptimize += t4 - t3; times.removeGaussian += t5 - t4; times.checkError += now_ms() - t5;
}

void SLAMPipeline::keyframeStepOverlapped() {
    ensureStreams();
    const hipStream_t frames = c10::hip::getCurrentHIPStream().stream();
    c10::hip::HIPStream& ms = static_cast<MapStream*>(map_stream_)->s;
    // the previous update must be complete before its camera / raycast lists are replaced (host wait: B is idle afterwards)
    pumpMapping(opt_pending_);
    if (map_in_flight_) { hip_ok(hipEventSynchronize((hipEvent_t)ev_map_), "hipEventSynchronize"); map_in_flight_ = false; }
    hip_ok(hipEventRecord((hipEvent_t)ev_frame_, frames), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(ms.stream(), (hipEvent_t)ev_frame_, 0), "hipStreamWaitEvent");  // raycasts see frame i's volume
    {
        c10::hip::HIPStreamGuard guard(ms);
        if (prune_pending_) {   // update k's prune (and loss records), before update k+1 reads the model and replaces the lists
            removeRedundantGs();
            if (sample_method == "ours") checkKeyFrameError();
            prune_pending_ = false;
        }
        update_frame_id_ = curr_frame_id;
        if (merge_keyframe_raycasts) raycastWindowAndKeyframes(localframe_cam_window, keyframe_cam_list, main_engine->camPoses);
        else { localFrameRaycast(); keyFrameRaycast(); }
        if (async_raycasts) waitAllRaycasts();  // (this arrangement keeps its single map stream: the gate below covers them)
        hip_ok(hipEventRecord((hipEvent_t)ev_raycasts_, ms.stream()), "hipEventRecord");
        initNewGaussians(localframe_raycast_window.back());
        localOptimizeBegin();
        map_update_open_ = true;
    }
    // a few iterations now, the rest a few at a time from the following processFrame calls: enqueueing all 20 at once keeps the
    // host (and with it the frame stream) busy for ~1 ms while the map stream only needs to stay ahead of the GPU
    pumpMapping(pump_iters_first);
    // the next frame's fusion must not modify the volume (or reuse the engine's free-view scratch) before the raycasts read it;
    // its tracking may overlap them: the wait sits in the engine's before-fusion hook
    (void)frames;
    main_engine->beforeNextFusion = [this] {
        hip_ok(hipStreamWaitEvent(c10::hip::getCurrentHIPStream().stream(), (hipEvent_t)ev_raycasts_, 0), "hipStreamWaitEvent");
    };
}

// ------------------------------------------------------------------ mapping thread
// The classic SLAM split: this (the caller's) thread tracks and fuses, a worker thread owns the Gaussian model and runs each
// keyframe's map update start to finish on the map stream -- its host-side waits (mask counts in addGaussians / prune, 240
// kernel launches) no longer stall the frame stream at all.  Hand-over at keyframe i: wait for update i-10, record "frame i
// fused", snapshot the camera lists / poses the update reads (the frame thread keeps appending to the originals), wake the
// worker, wait (~0.3 ms) until it has enqueued the raycasts and recorded their event, make the frame stream wait for it.
void SLAMPipeline::keyframeStepThreaded() {
    ensureStreams();
    const hipStream_t frames = c10::hip::getCurrentHIPStream().stream();
    if (pipeline_raycasts && async_raycasts) {
        // (1) this keyframe's free views, now: the raycast stream starts behind frame i's fusion; the result tensors are allocated
        // with the CONSUMER's stream current (the caching allocator ties a block to the stream it was allocated under)
        if (!rc_stream_) {
            rc_stream_ = new MapStream{make_stream(raycast_stream_kind)};
            hipEvent_t ev;
            hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
            ev_rc_begin_ = ev;
        }
        hip_ok(hipEventRecord((hipEvent_t)ev_frame_, frames), "hipEventRecord");
        const hipStream_t rcs = static_cast<MapStream*>(rc_stream_)->s.stream();
        hip_ok(hipStreamWaitEvent(rcs, (hipEvent_t)ev_frame_, 0), "hipStreamWaitEvent");
        MapJob next;
        next.curr_cam = curr_cam;
        next.frame_id = curr_frame_id;
        {
            c10::hip::HIPStreamGuard as_consumer(static_cast<MapStream*>(map_stream_)->s);
            buildUpdateViews(next);
        }
        hip_ok(hipEventRecord((hipEvent_t)ev_raycasts_, rcs), "hipEventRecord");
        // (2) the next frame's fusion must not modify the volume before these raycasts have read it: a stream-side wait, no host wait
        main_engine->beforeNextFusion = [this] {
            hip_ok(hipStreamWaitEvent(c10::hip::getCurrentHIPStream().stream(), (hipEvent_t)ev_raycasts_, 0), "hipStreamWaitEvent");
        };
        // (3) hand the job over once the worker has finished the previous update
        std::unique_lock<std::mutex> lk(mu_);
        if (!worker_.joinable()) worker_ = std::thread([this, dev = (int)c10::hip::current_device()] { mapWorker(dev); });
        const double tw0 = now_ms();
        cv_.wait(lk, [&] { return done_seq_ == job_seq_ || worker_error_; });
        g_handover_wait_ms = now_ms() - tw0;
        if (worker_error_) { lk.unlock(); rethrowWorkerError(); }
        job_ = std::move(next);
        job_seq_++;
        g_job_post_ms = now_ms();
        cv_.notify_all();
        return;
    }
    std::unique_lock<std::mutex> lk(mu_);
    if (!worker_.joinable()) worker_ = std::thread([this, dev = (int)c10::hip::current_device()] { mapWorker(dev); });
    const double tw0 = now_ms();
    cv_.wait(lk, [&] { return done_seq_ == job_seq_ || worker_error_; });
    g_handover_wait_ms = now_ms() - tw0;
    if (worker_error_) { lk.unlock(); rethrowWorkerError(); }
    hip_ok(hipEventRecord((hipEvent_t)ev_frame_, frames), "hipEventRecord");
    job_.curr_cam = curr_cam;
    job_.frame_id = curr_frame_id;
    job_.window = localframe_cam_window;
    job_.keyframes = keyframe_cam_list;
    job_.poses = main_engine->camPoses;
    job_seq_++;
    g_job_post_ms = now_ms();
    cv_.notify_all();
    // The next frame's fusion must not modify the volume (or reuse the engine's free-view scratch) before the update's raycasts
    // have read it -- but its TRACKING may run meanwhile: the wait (for the worker to have recorded the event, then the
    // stream-side wait on it) is deferred to the engine's before-fusion hook of the next ProcessFrame.
    const int64_t want = job_seq_;
    (void)frames;
    main_engine->beforeNextFusion = [this, want] {
        {
            const double tw1 = now_ms();
            std::unique_lock<std::mutex> lk2(mu_);
            cv_.wait(lk2, [&] { return raycasts_seq_ >= want || worker_error_; });
            g_gate_wait_ms = now_ms() - tw1;
        }
        rethrowWorkerError();
        hip_ok(hipStreamWaitEvent(c10::hip::getCurrentHIPStream().stream(), (hipEvent_t)ev_raycasts_, 0), "hipStreamWaitEvent");
    };
}

void SLAMPipeline::mapWorker(int device_index) {
    try {
        c10::hip::set_device((c10::DeviceIndex)device_index);
        c10::hip::HIPStream& ms = static_cast<MapStream*>(map_stream_)->s;
        c10::hip::HIPStreamGuard guard(ms);
        int64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || job_seq_ > seen; });
                if (stop_) return;
                seen = job_seq_;
            }
            const double t_woke = now_ms();
            // job_ is stable until done_seq_ catches up (the frame thread waits for that before it writes the next one)
            hip_ok(hipStreamWaitEvent(ms.stream(), (hipEvent_t)ev_frame_, 0), "hipStreamWaitEvent");  // raycasts see frame i's volume
            update_frame_id_ = job_.frame_id;
            if (job_.views_ready) {   // the frame thread has enqueued this update's views (and the gate event) already: adopt them
                localframe_raycast_window = std::move(job_.window_raycasts);
                window_raycast_events_ = std::move(job_.window_events);
                opt_cam_list = std::move(job_.opt_cams);
                opt_raycast_list = std::move(job_.opt_raycasts);
                opt_raycast_events_ = std::move(job_.opt_events);
                opt_window_len_ = job_.window_len; opt_frame_id_ = update_frame_id_;
                last_raycast_event_ = job_.last_event;
                { std::lock_guard<std::mutex> lk(mu_); raycasts_seq_ = seen; }
            } else {
                last_raycast_event_ = nullptr;
                const double r0 = now_ms();
                if (merge_keyframe_raycasts) {
                    raycastWindowAndKeyframes(job_.window, job_.keyframes, job_.poses);
                    times.localFrameRaycast += now_ms() - r0;
                } else {
                    raycastWindow(job_.window, job_.poses);
                    const double r1 = now_ms();
                    raycastKeyframes(job_.window, job_.keyframes, job_.poses);
                    times.localFrameRaycast += r1 - r0; times.keyFrameRaycast += now_ms() - r1;
                }
                // the gate of the next frame's fusion: the last raycast (on the raycast stream when they run beside the iterations)
                hip_ok(hipEventRecord((hipEvent_t)ev_raycasts_, async_raycasts && rc_stream_ ? static_cast<MapStream*>(rc_stream_)->s.stream()
                                                                                             : ms.stream()), "hipEventRecord");
                { std::lock_guard<std::mutex> lk(mu_); raycasts_seq_ = seen; }
            }
            cv_.notify_all();
            if (frame_report_ms >= 0.0 && now_ms() - g_job_post_ms > 1.0)
                fprintf(stderr, "[pipe] update %lld: raycasts enqueued %.3f ms after the hand-over (woke after %.3f)\n", (long long)seen,
                        now_ms() - g_job_post_ms, t_woke - g_job_post_ms);
            // the stages' host time on THIS thread (the reference's stage timers, slam_pipeline.cpp:116-135, with the update moved
            // here; read by the frame thread only after flush()); the wait for the map stream at the end is the iterations' GPU time
            // the enqueues ran ahead of: booked under localOptimize, which it mostly is
            const double s0 = now_ms();
            initNewGaussiansFor(localframe_raycast_window.back(), job_.curr_cam);
            const double s1 = now_ms();
            localOptimize();
            const double s2 = now_ms();
            removeRedundantGs();
            const double s3 = now_ms();
            if (sample_method == "ours") checkKeyFrameError();
            const double s4 = now_ms();
            waitAllRaycasts();
            hip_ok(hipStreamSynchronize(ms.stream()), "hipStreamSynchronize");
            times.initNewGaussians += s1 - s0; times.localOptimize += (s2 - s1) + (now_ms() - s4); times.removeGaussian += s3 - s2;
            times.checkError += s4 - s3;
            { std::lock_guard<std::mutex> lk(mu_); done_seq_ = seen; }
            cv_.notify_all();
        }
    } catch (...) {
        std::lock_guard<std::mutex> lk(mu_);
        worker_error_ = std::current_exception();
        cv_.notify_all();
    }
}

void SLAMPipeline::rethrowWorkerError() {
    std::exception_ptr e;
    { std::lock_guard<std::mutex> lk(mu_); e = worker_error_; worker_error_ = nullptr; }
    if (e) std::rethrow_exception(e);
}

// enqueue up to `count` pending optimise iterations on the map stream; closing the update records its completion event
void SLAMPipeline::pumpMapping(int count) {
    if (!map_update_open_) return;
    c10::hip::HIPStream& ms = static_cast<MapStream*>(map_stream_)->s;
    c10::hip::HIPStreamGuard guard(ms);
    optimizeIterations(count);
    if (opt_pending_ == 0) {
        hip_ok(hipEventRecord((hipEvent_t)ev_map_, ms.stream()), "hipEventRecord");
        map_update_open_ = false;
        map_in_flight_ = true;
        prune_pending_ = true;
    }
}

void SLAMPipeline::flush() {
    main_engine->beforeNextFusion = nullptr;  // no further frame: everything is joined below anyway
    if (worker_.joinable()) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return done_seq_ == job_seq_ || worker_error_; });
        lk.unlock();
        rethrowWorkerError();
    }
    pumpMapping(opt_pending_);
    if (frame_stream_) hip_ok(hipStreamSynchronize(static_cast<MapStream*>(frame_stream_)->s.stream()), "hipStreamSynchronize");
    if (map_in_flight_) { hip_ok(hipEventSynchronize((hipEvent_t)ev_map_), "hipEventSynchronize"); map_in_flight_ = false; }
    if (prune_pending_) {
        c10::hip::HIPStreamGuard guard(static_cast<MapStream*>(map_stream_)->s);
        removeRedundantGs();
        if (sample_method == "ours") checkKeyFrameError();
        prune_pending_ = false;
        hip_ok(hipStreamSynchronize(static_cast<MapStream*>(map_stream_)->s.stream()), "hipStreamSynchronize");
    }
}

SLAMPipeline::~SLAMPipeline() {
    if (tsdf_engine) {   // the engine outlives this pipeline: its hooks must not call into a dead object
        if (main_engine) main_engine->beforeNextFusion = nullptr;
        tsdf_engine->beforeShutdown = nullptr;
    }
    if (worker_.joinable()) {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        worker_.join();
    }
    if (map_stream_) {
        (void)hipStreamSynchronize(static_cast<MapStream*>(map_stream_)->s.stream());
        (void)hipStreamSynchronize(static_cast<MapStream*>(frame_stream_)->s.stream());
        for (void* e : {ev_frame_, ev_raycasts_, ev_map_, ev_caller_, job_events_[0][0], job_events_[0][1], job_events_[1][0], job_events_[1][1]}) if (e) (void)hipEventDestroy((hipEvent_t)e);
        delete static_cast<MapStream*>(map_stream_);
        delete static_cast<MapStream*>(frame_stream_);
    }
    // the free views' stream and its events (round 5: they used to outlive the pipeline -- one stream of the lowest-priority pool and
    // a dozen events leaked per scene of a process that builds many)
    if (rc_stream_) {
        (void)hipStreamSynchronize(static_cast<MapStream*>(rc_stream_)->s.stream());
        for (void* e : rc_events_) if (e) (void)hipEventDestroy((hipEvent_t)e);
        if (ev_rc_begin_) (void)hipEventDestroy((hipEvent_t)ev_rc_begin_);
        delete static_cast<MapStream*>(rc_stream_);
    }
}

void SLAMPipeline::SLAMTrainCams(std::vector<Camera>& cams, const std::vector<torch::Tensor>& rgb_u8,
                                 const std::vector<torch::Tensor>& depth_mm_i16) {
    for (size_t i = 0; i < cams.size(); i++) processFrame((int)i, cams[i], rgb_u8[i], depth_mm_i16[i]);
    flush();
}
