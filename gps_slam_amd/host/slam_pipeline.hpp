// SLAMPipeline (slam/slam_pipeline.{h,cpp}): the per-frame loop that drives the hot path.
//
//    SLAMTrainCams      :52-173   per frame: TSDF ProcessFrame; every local_opt_interval frames:
//    localFrameRaycast  :417-448  runRaycastByCam for the <= 2 window frames
//    keyFrameRaycast    :528-561  + <= 7 random keyframes
//    initNewGaussians   :450-526  error-mask sampling -> addGaussians
//    localOptimize      :195-289  20 x (forward, L1, backward, 7 x Adam)
//    removeRedundantGs  :564-586  prune by scale / opacity
//
// All compute goes through the C-ABI (RawGaussianModel, TsdfEngine); this file is bookkeeping.  Random choices the
// reference seeds from std::random_device (dataset_reader.h:39) are seeded here so that runs are reproducible.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <random>

#include "raw_gs_model.hpp"
#include "infinitam_tools.hpp"
#include "tsdf_engine.hpp"

// dataset_reader.h:26-100 (uniform branch): draw without replacement, refill when exhausted
template <class T>
class RandomSelector {
public:
    RandomSelector(const std::vector<T>& items, std::mt19937_64& rng) : rng_(rng) {
        for (size_t i = 0; i < items.size(); i++) original_.push_back({(int)i, &items[i]});
        current_ = original_;
    }
    std::pair<int, const T*> getNext() {
        if (current_.empty()) current_ = original_;
        std::uniform_int_distribution<size_t> d(0, current_.size() - 1);
        const size_t i = d(rng_);
        auto v = current_[i];
        current_[i] = current_.back();
        current_.pop_back();
        return v;
    }

private:
    std::vector<std::pair<int, const T*>> original_, current_;
    std::mt19937_64& rng_;
};

// include/file_utils.h:35, src/file_utils.cpp:172-210: device memory in use, MiB (the reference asks NVML for `memory.used`; here
// hipMemGetInfo of the device: total - free, the same quantity -- every process's allocations on that GPU)
unsigned long long getGPUMemoryUsage(int gpu_id = 0);
torch::Tensor computeNormalMap(const torch::Tensor& vertex_map);  // src/tensor_math.cpp:278-300 -> gps_normal_map

class SLAMPipeline {
public:
    // use_gt_pose: TSDF.use_gt_pose of the configs (true in every shipped one) -> engine->turnOffTracking(), as
    // createTsdfEngine does (InfiniTAM_tools.cpp:59-62); false keeps the depth tracker active
    SLAMPipeline(TsdfEngine* tsdf_engine, SLAMGaussianModel* model, uint64_t seed = 1234, bool use_gt_pose = true);
    // the reference's construction sequence (slam_trainer.cpp:26-33): SLAMPipeline pipe; pipe.setTsdfEngine(createTsdfEngine(
    // reader, config)); ... pipe.SLAMTrainCams(model, cams).  Frames then come from the CLIEngine's host-resident sequence
    // (per-frame upload inside the loop) instead of device tensors handed to processFrame.
    explicit SLAMPipeline(uint64_t seed = 1234);
    void setTsdfEngine(InfiniTAM::Engine::CLIEngine* tsdf_engine);  // slam_pipeline.h:22-29
    void detachTsdfEngine();   // flush + drop the pointers into the engine (CLIEngine::Shutdown calls it through beforeShutdown)
    void SLAMTrainCams(SLAMGaussianModel& model, std::vector<Camera>& cams);  // slam_pipeline.cpp:52-173
    void processFrame(int i, Camera& cam);  // one iteration of that loop: tsdf_engine->ProcessFrame() + the Gaussian block

    void loadConfig(const gpsh::Config& config);  // PIPELINE section keys

    // body of the SLAMTrainCams frame loop (:69-132) for frame i
    void processFrame(int i, Camera& cam, const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16);
    void SLAMTrainCams(std::vector<Camera>& cams, const std::vector<torch::Tensor>& rgb_u8,
                       const std::vector<torch::Tensor>& depth_mm_i16);

    TensorDict runRaycastByCam(const Camera& cam, bool use_cam_depth = true);  // :362-415
    void updateFrameList();                                                    // :319-360
    void localFrameRaycast();
    void keyFrameRaycast();
    void initNewGaussians(TensorDict& raycast_maps);
    void localOptimize();
    void removeRedundantGs();

    // renderEvalImgs (slam_pipeline.cpp:588-695), the compute half: for every camera the free-view raycast with the stored pose
    // and, if the model is not empty, forward() under NoGradGuard.  Returns per camera the tensors the reference turns into
    // image files with OpenCV ("raycast_color", "raycast_depth" and, per requested name, "rgb" clamped to [0,1], "alpha",
    // "depth") plus "psnr" -- and the same images AS THE REFERENCE QUANTISES THEM for its files: "rgb_u8", "gt_u8",
    // "raycast_color_u8" ((t * 255.0).toType(kU8), cv_utils.cpp:57-76), "raycast_depth_u16" (millimetres, cv_utils.cpp:79-101)
    // and "psnr_u8" = scripts/metric.py's PSNR of the 8-bit render against the 8-bit ground truth.  The JPEG / PNG encoders
    // themselves are I/O outside this library.
    std::vector<TensorDict> renderEvalImgs(const std::vector<Camera>& cams, const std::vector<std::string>& names = {"rgb"});

    // slam_pipeline.h:32-49: mesh / engine state files under workspace_dir (empty names are skipped like the reference)
    std::string workspace_dir = ".", saved_mesh, saved_engine;
    void saveMesh() { if (!saved_mesh.empty()) main_engine->SaveSceneToMesh((workspace_dir + "/" + saved_mesh).c_str()); }
    void saveEngine() { if (!saved_engine.empty()) main_engine->SaveToFile(workspace_dir + "/" + saved_engine); }
    void loadEngine() { main_engine->LoadFromFile(workspace_dir + "/" + saved_engine); }

    torch::Device device = torch::kCUDA;
    std::string work_mode = "train";
    InfiniTAM::Engine::CLIEngine* tsdf_engine = nullptr;
    TsdfEngine* main_engine = nullptr;
    SLAMGaussianModel* model = nullptr;
    float voxel_size;

    int curr_frame_id = -1;
    int localframe_cam_window_length = 2, localframe_cam_window_interval = 5;
    int local_opt_iters = 20, local_opt_interval = 10;
    int keyframe_select_max = 7;
    Camera curr_cam;
    std::deque<Camera> localframe_cam_window;
    std::deque<TensorDict> localframe_raycast_window;
    std::vector<Camera> keyframe_cam_list;
    std::vector<Camera> opt_cam_list;
    std::vector<TensorDict> opt_raycast_list;
    float keyframe_theta_thres = 30.0f, keyframe_trans_thres = 0.3f;
    // keyframe_sample_configs (slam_pipeline.cpp:130, 293-317, 538): "random" draws keyframe_select_max history keyframes per update;
    // "ours" -- as the reference ships it -- adds NO history views (keyFrameRaycast has a "random" branch only) and keeps a loss
    // record per keyframe: {loss, frame id of the check, mean raycast confidence, number of checks with loss > loss_thres}
    std::string sample_method = "random";
    float loss_thres = 0.0f;
    std::map<int, std::vector<float>> keyframe_loss_dict;
    void checkKeyFrameError();
    float new_gs_sample_ratio = 0.25f, color_error_thres = 0.05f;
    float depth_vis_max = 5.0f, depth_vis_min = 0.0f, alpha_vis_max = 5.0f;
    float large_scale_thres = 0.1f, small_scale_thres = 0.003f, low_opac_thres = 0.005f;
    float scene_scale = 1.1f * 3.0f;
    float ssim_weight = 0.0f, depth_weight = 0.0f;  // both 0 in every shipped config -> the fused L1 trainStep

    // LOG_PIPELINE_TIME of the reference (slam_pipeline.cpp:54-67, 73-96, 141-167): host wall-clock totals of one SLAMTrainCams run in
    // milliseconds.  `per_frame` is the reference's "per frame fusion time" (ProcessFrame + pose + toGPU + updateFrameList); FPS =
    // frames / (slam_total / 1000); run/read_results.py:38-39 derives Fusion-FPS = 1000 / per_frame and Gaussian-FPS =
    // 1000 / (1000 / FPS - per_frame) from them.  The five per-stage totals are taken in the sequential keyframe step only (the
    // overlapped schedules run the stages on another stream / thread; their host share of the frame thread is `keyframe_step`).
    // slam_total here ends AFTER flush() and a device synchronise (the reference stops its clock with kernels still in flight).
    // Under overlap_mapping / mapping_thread only per_frame (-> fusion_fps), slam_total (-> fps) and gpu_memory_mb mean what the
    // reference's timers mean: localFrameRaycast / keyFrameRaycast are then the frame thread's ENQUEUE time, localOptimize includes
    // the map stream's synchronisation wait -- not comparable with the reference's per-stage numbers (sequential schedule: they are).
    struct PipelineTimes {
        int frames = 0;
        double slam_total = 0, per_frame = 0, keyframe_step = 0, localFrameRaycast = 0, keyFrameRaycast = 0, initNewGaussians = 0,
               localOptimize = 0, removeGaussian = 0, checkError = 0;
        long long gpu_memory_mb = -1;    // getGPUMemoryUsage after emptyCache() at the end of the run (slam_pipeline.cpp:168-171)
        double max_frame_after_30 = 0;   // slowest processFrame call (host wall) from frame 30 on
        int max_frame_id = -1;
        double fps() const { return slam_total > 0 ? frames / (slam_total / 1000.0) : 0.0; }
        double fusion_fps() const { return per_frame > 0 ? 1000.0 / (per_frame / frames) : 0.0; }
        double gaussian_fps() const {
            const double rest = slam_total > 0 && frames > 0 ? (slam_total - per_frame) / frames : 0.0;
            return rest > 0 ? 1000.0 / rest : 0.0;
        }
    } times;
    bool log_pipeline_time = false;        // print the reference's "[PIPELINE AVG TIME]" line at the end of SLAMTrainCams
    double frame_report_ms = -1.0;         // debug aid: a processFrame call that took longer than this many ms of host time prints where it went
    std::vector<float> frame_ms;           // host wall of every processFrame call of the last SLAMTrainCams (filled when keep_frame_ms)
    std::vector<float> frame_wait_ms;      // ... of which: waiting for the map worker (previous update's completion / this update's raycasts)
    bool keep_frame_ms = false;

    // counters are bumped by the frame thread AND (mapping_thread) by the map worker: atomics
    struct Stats { std::atomic<int64_t> frames{0}, opt_iters{0}, raycasts{0}, added{0}, pruned{0}; } stats;

    // Tracking / mapping overlap.  The reference runs a keyframe's map update (raycasts, new Gaussians, 20 optimise iterations,
    // prune) to completion before it looks at the next frame.  Nothing in frames i+1 .. i+9 reads the Gaussian model, and the
    // model update reads the TSDF volume only through the free-view raycasts at its start -- so with overlap_mapping the
    // update runs on a second HIP stream while a high-priority stream tracks and fuses the following frames (the latency-bound
    // LM loop of the tracker fits into the shadow of the throughput-bound rasterizer kernels).  Ordering is kept with events:
    // map stream waits for frame i's fusion; frame i+1 may TRACK at once but waits for the raycasts before it fuses
    // (TsdfEngine::beforeNextFusion); the next keyframe waits for the previous update; without a mapping thread the prune
    // of update k runs at the start of update k+1 (or in flush()).
    // Results are those of the sequential schedule.  Off by default: with it on, processFrame() returns while the update is
    // still in flight and the model may only be read after flush() (SLAMTrainCams and the accessors below call it).
    bool overlap_mapping = false;
    // overlapped arrangements: the window's and the keyframes' free views of an update as ONE batch (the next frame's fusion waits
    // for all of them anyway: 0.68 instead of 0.23 + 0.54 ms) or as two (initNewGaussians then starts after the window's views
    // alone).  Measured: one batch is + 1 % on some boxes of the pool and - 5 % on others at 640x480 (the map update and the ten
    // frames it runs beside are equally long there, so which of the two waits for the other flips) and - 6 % at 1280x720: two.
    bool merge_keyframe_raycasts = false;
    bool views_reserved_ = false, raycast_pool_warm_ = false;
    bool async_raycasts = true;   // the update's free-view raycasts run beside its optimise iterations (same results)
    // which stream each chain gets (slam_pipeline.cpp: make_stream): 0 / 1 = torch's high- / normal-priority pool, 2 / 3 / 4 = a
    // stream of the pipeline's own at the lowest / highest / default priority
    // Round 5: ALL THREE are streams of the library's own now (one per priority for the life of the process: make_stream).  torch hands
    // its pool streams out round-robin and never destroys them, ROCm spreads the streams of one priority level over <= 4 hardware
    // queues by reference count -- so which chains ended up sharing a queue depended on how many scenes the PROCESS had built
    // before: the same 1,000-frame run gave 1,120 frames/s in a fresh process and 900 after six other scenes, a test process's
    // tenth scene ran its frames and its map update on ONE queue (540 instead of 850 frames/s, every period's fifth frame 6 ms
    // late); with own streams every measurement is history-free (bench.py: cfg1 1,070 -> 1,640 and cfgR 710 -> 820 frames/s when
    // they run after the whole-sequence leg).
    int frame_stream_kind = 3, map_stream_kind = 4, raycast_stream_kind = 2;
    // an optimise iteration's backward + Adam kernel also runs the NEXT iteration's preprocessing forward (the next camera is drawn
    // one iteration early: same draws, same order): one launch and one pass over the parameters less per iteration, same results
    bool prefetch_next_preprocess = true;
    // what gps_set_frame_chain_reserve gets while the schedules overlap (1: the strip backward leaves register room for a tracker wave
    // and the forward launches row-major; 0: off; other bits: experiment switches of the kernel library)
    int frame_chain_reserve = 1;
    // with mapping_thread + async_raycasts: the free views of update k + 1 are enqueued by the FRAME thread the moment keyframe k + 1 is
    // fused, whether or not the worker has finished update k (they read the volume, not the model; their results go into the job the
    // worker adopts when it gets there).  Before, the worker raycast at the start of its job: 0.6-0.8 ms at the head of a chain that
    // carries the period once the update is longer than its ten frames, and the next frame's fusion waited for the worker to get that far.
    bool pipeline_raycasts = true;
    // with overlap_mapping: run the map update on a worker thread of its own (tracking thread + mapping thread) instead of
    // interleaving its host work with the frames on the caller's thread
    bool mapping_thread = false;
    int pump_iters_first = 4, pump_iters_per_frame = 3;  // optimise iterations enqueued at the keyframe / per following frame
    void flush();
    // Test hook: with trace_frames every frame leaves, ENQUEUED on the stream the frame ran on (no host synchronisation, so the
    // schedule under test is undisturbed), a copy of the live raycast image and of the engine's counters, plus the pose the
    // frame was fused with (host values).  Read them after flush().
    bool trace_frames = false;
    std::vector<torch::Tensor> trace_live, trace_counters, trace_poses;
    ~SLAMPipeline();

private:
    void processFrameImpl(int i, Camera& cam, const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16);
    void ensureStreams();
    // `ev_out` != nullptr: run on the raycast stream and hand back the event that marks the result complete
    TensorDict raycastCam(const Camera& cam, const std::vector<ORUtils::SE3Pose>& poses, void** ev_out = nullptr);
    // the same for several cameras of one volume state in ONE batched free-view chain; one event for all results
    // (use_event: record THIS event behind the batch instead of one of the update's pooled events)
    std::vector<TensorDict> raycastCams(const std::vector<const Camera*>& cams, const std::vector<ORUtils::SE3Pose>& poses,
                                        void** ev_out = nullptr, void* use_event = nullptr);
    void raycastWindowAndKeyframes(const std::deque<Camera>& window, const std::vector<Camera>& keyframes,
                                   const std::vector<ORUtils::SE3Pose>& poses);
    void raycastWindow(const std::deque<Camera>& window, const std::vector<ORUtils::SE3Pose>& poses);
    void raycastKeyframes(const std::deque<Camera>& window, const std::vector<Camera>& keyframes,
                          const std::vector<ORUtils::SE3Pose>& poses);
    void initNewGaussiansFor(TensorDict& raycast_maps, const Camera& cam);
    void keyframeStep();
    void keyframeStepOverlapped();
    void keyframeStepThreaded();
    void mapWorker(int device_index);
    void rethrowWorkerError();
    struct MapJob {
        Camera curr_cam; int frame_id = 0; std::deque<Camera> window; std::vector<Camera> keyframes; std::vector<ORUtils::SE3Pose> poses;
        // pipeline_raycasts: the update's views, already enqueued by the frame thread (the worker adopts them instead of raycasting)
        bool views_ready = false;
        std::vector<Camera> opt_cams; std::vector<TensorDict> opt_raycasts; std::vector<void*> opt_events;
        std::deque<TensorDict> window_raycasts; std::vector<void*> window_events;
        size_t window_len = 0; void* last_event = nullptr;
    };
    MapJob job_;
    void buildUpdateViews(MapJob& out);           // localFrameRaycast + keyFrameRaycast of the keyframe at hand into `out` (frame thread)
    void* job_events_[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // {window batch, keyframe batch} per update parity
    int job_parity_ = 0;
    void* last_raycast_event_ = nullptr;          // the adopted job's last batch (waitAllRaycasts)
    std::mt19937_64 rng_kf_;                      // the history keyframes' draw (its own generator: the frame thread draws while the worker's update draws cameras)
    size_t opt_window_len_ = 0;                  // length of the local window at the front of opt_cam_list (the rest: history keyframes)
    int opt_frame_id_ = 0, update_frame_id_ = 0; // frame number of the update opt_cam_list belongs to / of the update being built
    std::mutex loss_mu_;                         // keyframe_loss_dict: written by updateFrameList (frame thread) and checkKeyFrameError (map thread)
    std::thread worker_;
    std::mutex mu_;
    std::condition_variable cv_;
    int64_t job_seq_ = 0, raycasts_seq_ = 0, done_seq_ = 0;
    bool stop_ = false;
    std::exception_ptr worker_error_;
    void localOptimizeBegin();
    void optimizeIterations(int count);
    void pumpMapping(int count);
    std::unique_ptr<RandomSelector<Camera>> opt_loader_;
    std::pair<int, const Camera*> opt_peek_{0, nullptr};   // the next iteration's camera, drawn one iteration early (optimizeIterations)
    bool opt_peek_valid_ = false;
    int opt_pending_ = 0;
    bool map_update_open_ = false;
    std::mt19937_64 rng_;
    at::Generator gen_;
    void *map_stream_ = nullptr, *frame_stream_ = nullptr;  // c10 stream handle storage (see .cpp)
    // A keyframe's free-view raycasts on a stream of their own, next to the optimise iterations (they only read the volume; the
    // rasterizer kernels fill the GPU their latency-bound tails leave idle): every result carries an event, and whoever reads
    // a result (initNewGaussians, an optimise iteration) makes its stream wait for it.
    void* rc_stream_ = nullptr;
    void* ev_rc_begin_ = nullptr;                 // "the volume the raycasts read is complete" (recorded on the issuing stream)
    std::vector<void*> rc_events_;                // hipEvent_t pool, one per raycast of the current update
    size_t rc_event_next_ = 0;
    std::vector<void*> window_raycast_events_, opt_raycast_events_;  // parallel to localframe_raycast_window / opt_raycast_list
    void beginAsyncRaycasts();                    // call once per update, on the stream that is ordered after the volume
    void waitRaycast(void* ev);                   // current stream waits for one result (no-op for nullptr)
    void waitAllRaycasts();                       // ... for all of this update's
    void *ev_frame_ = nullptr, *ev_raycasts_ = nullptr, *ev_map_ = nullptr, *ev_caller_ = nullptr;  // hipEvent_t
    bool map_in_flight_ = false, prune_pending_ = false;
};
