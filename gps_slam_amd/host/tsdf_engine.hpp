// The TSDF engine surface SLAMPipeline drives (slam/slam_pipeline.cpp:69-83, 362-415; slam/InfiniTAM_tools.cpp:3-67;
// ITMLib/Core/ITMBasicEngine.{h,tpp}) on the C-ABI, in two layers:
//
//   TsdfEngine                                 the engine proper: owns the device buffers (through libtorch, where the reference
//                                              uses ORUtils::MemoryBlock), fills gps_tsdf_state once, afterwards only passes
//                                              pointers; tensor-based ProcessFrame for callers whose frames are already in HBM
//   ITMLib::ITMBasicEngine<TVoxel, TIndex>     the reference's class template over it (infinitam_tools.hpp): same constructor,
//     : ITMLib::ITMMainEngine, TsdfEngine      ProcessFrame(ITMUChar4Image*, ITMShortImage*, ITMIMUMeasurement*), runRaycast(
//                                              SE3Pose*, ITMIntrinsics*), camPoses / camIntrincs / gtC2wPoses, so that code
//                                              written against ITMBasicEngine.h:54-92 (slam_pipeline.cpp's dynamic_casts) reads as is
//   InfiniTAM::Engine::CLIEngine, createTsdfEngine    infinitam_tools.hpp
//
// Method and member names follow the reference: ProcessFrame, runRaycast, GetFreeImage, GetFreeVertex, getVoxelSize,
// GetTrackingState()->pose_d->GetInvM(), camPoses, camIntrincs, gtC2wPoses, turnOffTracking, SaveToFile, LoadFromFile,
// SaveSceneToMesh.
#pragma once
#include <cstring>
#include <memory>
#include <functional>
#include <mutex>

#include "gps_host_common.hpp"

enum MemoryDeviceType { MEMORYDEVICE_CPU, MEMORYDEVICE_CUDA };

namespace ORUtils {

template <class T> struct Vector2 {  // ORUtils/Vector.h (the two fields the engine surface uses)
    T x, y;
    Vector2() : x(0), y(0) {}
    Vector2(T x_, T y_) : x(x_), y(y_) {}
    T width() const { return x; }
    T height() const { return y; }
};

// ORUtils/SE3Pose.h as used here: SetInvM(c2w) + Coerce() -> GetM / GetInvM (ORUtils column-major float[16])
class SE3Pose {
public:
    SE3Pose() { for (int i = 0; i < 16; i++) M_[i] = invM_[i] = (i % 5 == 0) ? 1.f : 0.f; }
    void SetInvM(const float* c2w_row_major) { gpsh::check(gps_pose_from_c2w(c2w_row_major, M_, invM_), "gps_pose_from_c2w"); }
    void SetBoth(const float* M, const float* invM) { for (int i = 0; i < 16; i++) { M_[i] = M[i]; invM_[i] = invM[i]; } }
    void Coerce() {}  // done by gps_pose_from_c2w
    const float* GetM() const { return M_; }
    const float* GetInvM() const { return invM_; }

private:
    float M_[16], invM_[16];
};

// ORUtils/Image.h + MemoryBlock.h as the pipeline uses them: an image with a host copy (pinned, so that UpdateView's
// CPU->device transfer can be asynchronous), a device copy, or both; GetData(MEMORYDEVICE_CUDA) is a pointer
// torch::from_blob can wrap (src/cv_utils.cpp:324-336).  Storage is torch tensors.
template <class T>
class Image {
public:
    Vector2<int> noDims;
    Image(int w, int h, torch::Tensor device_storage) : noDims(w, h), dev_(device_storage) {}
    // ORUtils::Image(noDims, allocate_CPU, allocate_CUDA) (Image.h:27-33)
    Image(Vector2<int> dims, bool allocate_CPU, bool allocate_CUDA, torch::Device device = torch::kCUDA) : noDims(dims) {
        const int64_t bytes = (int64_t)dims.x * dims.y * (int64_t)sizeof(T);
        // (empty + memset, not torch::zeros: the fill of a megabyte is a parallel region of torch's intra-op pool -- one thread per
        // core the machine SHOWS, each spinning for a while afterwards; two such regions per frame of a sequence exhaust a
        // container's CPU quota and the kernel then freezes every thread of the process, the tracking thread included, until the
        // end of the 100 ms accounting period: LABBOOK section 14, tools/probe/early_stall.py)
        if (allocate_CPU) {
            host_ = torch::empty({bytes}, torch::TensorOptions().dtype(torch::kUInt8).pinned_memory(true));
            std::memset(host_.data_ptr(), 0, (size_t)bytes);
        }
        if (allocate_CUDA) dev_ = torch::zeros({bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(device));
    }
    T* GetData(MemoryDeviceType t) const {
        const torch::Tensor& s = t == MEMORYDEVICE_CUDA ? dev_ : host_;
        TORCH_CHECK(s.defined(), "Image: no ", t == MEMORYDEVICE_CUDA ? "device" : "host", " copy allocated");
        return reinterpret_cast<T*>(s.data_ptr());
    }
    bool isAllocated_CPU() const { return host_.defined(); }
    bool isAllocated_CUDA() const { return dev_.defined(); }
    size_t dataSize() const { return (size_t)noDims.x * noDims.y; }
    const torch::Tensor& tensor() const { return dev_; }         // device storage
    const torch::Tensor& host_tensor() const { return host_; }   // pinned host storage (raw bytes)

private:
    torch::Tensor dev_, host_;
};

}  // namespace ORUtils

struct Vector4u { unsigned char x, y, z, w; };
struct Vector4f { float x, y, z, w; };
typedef ORUtils::Vector2<int> Vector2i;
typedef ORUtils::Image<Vector4u> ITMUChar4Image;
typedef ORUtils::Image<Vector4f> ITMFloat4Image;
typedef ORUtils::Image<short> ITMShortImage;

// ITMLib/Objects/Tracking/ITMTrackingState.h:20-27, 38
struct ITMTrackingState {
    enum TrackingResult { TRACKING_GOOD = 2, TRACKING_POOR = 1, TRACKING_FAILED = 0 };
    ORUtils::SE3Pose* pose_d;
    TrackingResult trackerResult = TRACKING_GOOD;
};

namespace ITMLib {

// ITMLib/Objects/Camera/ITMIntrinsics.h:17-63
class ITMIntrinsics {
public:
    struct ProjectionParamsSimple { Vector4f all; float fx, fy, px, py; } projectionParamsSimple;
    Vector2i imgSize;
    void SetFrom(int width, int height, float fx, float fy, float cx, float cy) {
        imgSize = Vector2i(width, height);
        projectionParamsSimple.fx = fx; projectionParamsSimple.fy = fy; projectionParamsSimple.px = cx; projectionParamsSimple.py = cy;
        projectionParamsSimple.all = Vector4f{fx, fy, cx, cy};
    }
    ITMIntrinsics() { SetFrom(640, 480, 580.f, 580.f, 320.f, 240.f); }  // ITMIntrinsics.cpp:11-15
};

struct ITMDisparityCalib { void SetStandard() {} };  // depth = mm * 0.001, no affine/Kinect disparity (InfiniTAM_tools.cpp:10)

// ITMLib/Objects/Camera/ITMRGBDCalib.h
class ITMRGBDCalib {
public:
    ITMIntrinsics intrinsics_rgb, intrinsics_d;
    ITMDisparityCalib disparityCalib;
};

// ITMLib/Utils/ITMSceneParams.h + ITMLibSettings.{h,cpp} (the fields createTsdfEngine sets, reference defaults :10)
struct ITMSceneParams { float voxelSize = 0.005f, mu = 0.02f, viewFrustum_min = 0.2f, viewFrustum_max = 3.0f; int maxW = 100; };
class ITMLibSettings {
public:
    ITMSceneParams sceneParams;
    // hash table / voxel block array capacities (ITMVoxelBlockHash.h:18-22; runtime parameters here)
    int noTotalEntries_blocks = 0x40000, noBuckets = 0x100000, excessListSize = 0x20000;
};

class ITMIMUMeasurement;

}  // namespace ITMLib

struct ITMVoxel_s_rgb {};        // ITMLib/Objects/Scene/ITMVoxelTypes.h:41-69 (layout: gps_voxel)
struct ITMVoxelBlockHash {};     // ITMLib/Objects/Scene/ITMVoxelBlockHash.h (layout: gps_hash_entry)
typedef ITMVoxel_s_rgb ITMVoxel;            // ITMLib/ITMLibDefines.h:18
typedef ITMVoxelBlockHash ITMVoxelIndex;    // ITMLib/ITMLibDefines.h:25

class TsdfEngine {
public:
    // capacities: ITMLib/Objects/Scene/ITMVoxelBlockHash.h:18-22
    TsdfEngine(int width, int height, float fx, float fy, float cx, float cy, float voxel_size = 0.005f,
                   float mu = 0.02f, float view_frustum_min = 0.2f, float view_frustum_max = 10.0f,
                   int n_blocks = 0x40000, int n_buckets = 0x100000, int n_excess = 0x20000,
                   torch::Device device = torch::kCUDA);

    void resetAll();
    // ITMBasicEngine::ProcessFrame with tracking off: pose := gtC2wPoses[framesProcessed] (ITMBasicEngine.tpp:260-385).
    // rgb uint8[H,W,4] (uchar4) and depth int16[H,W] (mm) device tensors are read in place.
    ITMTrackingState* ProcessFrame(const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16);
    // ITMBasicEngine::runRaycast(pose, intrinsics) (ITMBasicEngine.tpp:501-526): free-view FindVisibleBlocks +
    // CreateExpectedDepths + RenderImage into the free-view render state.  intrinsics == NULL: the depth camera's.
    // (pose == NULL && intrinsics == NULL -- re-render of the live view, :503-518 -- is not on SLAMPipeline's path: throws.)
    void runRaycast(ORUtils::SE3Pose* pose = nullptr, ITMLib::ITMIntrinsics* intrinsics = nullptr);
    // ITMViewBuilder::UpdateView (ViewBuilding/CUDA/ITMViewBuilder_CUDA.cu:32-91) + ProcessFrame: images with a host copy only
    // are uploaded (asynchronously, pinned -> the engine's staging buffers) on the current stream first; images with a device
    // copy are read in place.
    ITMTrackingState* ProcessFrame(ITMUChar4Image* rgbImage, ITMShortImage* rawDepthImage);
    // runRaycast for several poses at once (gps_tsdf_free_raycast_batch): every launch of the free-view chain covers all
    // views, each view renders into a render state of its own (kept by the engine, created on first use) -- view k's images
    // are GetFreeImage(k) / GetFreeVertex(k).  Same images as runRaycast(pose k) would leave in GetFreeImage() / GetFreeVertex().
    // `maps` (optional, one per pose): the runRaycastByCam tensor glue of the view (gps_raycast_to_maps), written by the batch.
    struct ViewMaps { const float* w2c_row_major; float *color_map, *vertex_map, *confidence_map, *depth_map, *depth_map_clamped; };
    void runRaycastBatch(const std::vector<ORUtils::SE3Pose>& poses, ITMLib::ITMIntrinsics* intrinsics = nullptr,
                         const std::vector<ViewMaps>* maps = nullptr,
                         const std::vector<ITMLib::ITMIntrinsics>* per_view_intrinsics = nullptr);
    // creates the render states of views 0 .. n-1 now (runRaycastBatch otherwise creates them on first use: ~20 MB of device
    // allocations per view in the middle of a keyframe update)
    void reserveViews(int n);
    ITMUChar4Image* GetFreeImage(int view) { return views_.at(view)->image_p.get(); }
    ITMFloat4Image* GetFreeVertex(int view) { return views_.at(view)->vertex_p.get(); }
    ITMUChar4Image* GetFreeImage() { return &free_image_; }
    ITMFloat4Image* GetFreeVertex() { return &free_vertex_; }
    ITMFloat4Image* GetLiveVertex() { return &live_vertex_; }
    float getVoxelSize() const { return state_.voxel_size; }
    ITMTrackingState* GetTrackingState() { return &tracking_state_; }

    // ITMBasicEngine::SaveSceneToMesh (ITMBasicEngine.tpp:105-117): MeshScene (gps_tsdf_mesh_scene, triangles in the CPU
    // engine's deterministic order) + ITMMesh::WritePLY (ITMMesh.h:39-106).  Returns noTotalTriangles.
    int64_t SaveSceneToMesh(const char* fileName, int64_t maxTriangles = (int64_t)1 << 24);
    // {triangles float[maxTriangles,7,3] (p0 p1 p2 c0 c1 c2 clr), counts int64[2]} on the device, no host sync
    std::pair<torch::Tensor, torch::Tensor> MeshScene(int64_t maxTriangles = (int64_t)1 << 24);
    // ITMBasicEngine::SaveToFile / LoadFromFile (ITMBasicEngine.tpp:119-171): <dir>Scene/{voxel.dat, alloc.dat, vba.txt,
    // hash.dat, excess.dat, last.txt} in the reference's MemoryBlockPersister format (size_t count + raw elements)
    void SaveToFile(const std::string& saveOutputDirectory);
    void LoadFromFile(const std::string& saveInputDirectory);
    // MAX_RENDERING_BLOCKS (262144) exceeded in some CreateExpectedDepths since the last check?  The reference drops the excess
    // blocks silently (ITMVisualisationEngine_CUDA.tcu:150-163); here the kernel raises counters[GPS_TSDF_OVERFLOW] and this
    // (blocking) read-back turns it into one warning on stderr per engine.  Returns the flag.
    bool checkRenderingBlocks();
    const gps_tsdf_state& state() const { return state_; }
    torch::Tensor currentRgb() const { return frame_inputs_.empty() ? torch::Tensor() : frame_inputs_[0]; }  // view->rgb, [H,W,4] u8
    torch::Tensor counters() const { return counters_; }

    // ITMBasicEngine::turnOffTracking (ITMBasicEngine.tpp:532; createTsdfEngine calls it when use_gt_pose is true,
    // InfiniTAM_tools.cpp:59-62): poses then come from gtC2wPoses.  With tracking active (the reference's default) the
    // depth-only ExtendedTracker of ITMLibSettings.cpp:54-57 estimates them (gps_tsdf_process_frame_tracked).
    void turnOffTracking() { trackingActive = false; }
    // the tracker's argument line through the BAR (default, when the device memory is host-visible) or in the pinned mailbox
    void setBarArgLine(bool on) { bar_arg_line = on; track_state_.dev_arg_line = on ? track_arg_line_.get() : nullptr; }
    bool usesBarArgLine() const { return track_state_.dev_arg_line != nullptr; }
    // how many of the poses the LM loop would evaluate after a REJECTION ride along with every evaluation (0..2; BAR line only;
    // gps_track_state.mailbox_bytes).  Same poses, fewer round trips; the switch exists for A/B measurements and the equality test.
    void setPosesRidingAlong(int n) {
        poses_riding_along = n < 0 ? 0 : n > kMaxRidingAlong ? kMaxRidingAlong : n;
        track_state_.mailbox_bytes = mailboxBytes();
    }
    // the evaluation's workgroups store their rows of partial sums straight into the pinned mailbox and the tracking thread adds
    // them (default), or a summing workgroup on the device does and only the totals travel (round 4); same poses, bit for bit
    void setHostSummedRows(bool on) { host_summed_rows = on; track_state_.mailbox_bytes = mailboxBytes(); }
    bool hostSummedRows() const { return host_summed_rows; }
    int32_t mailboxBytes() const {
        return (1 + poses_riding_along) * (GPS_TRACK_MAILBOX_BLOCK_BYTES + (host_summed_rows ? GPS_TRACK_MAILBOX_ROWS_BYTES : 0));
    }
    int posesRidingAlong() const { return poses_riding_along; }
    // of the last tracked frame: {poses that rode along with evaluations, poses the loop consumed}
    float trackDiag(int k) const { return k >= 0 && k < 16 ? track_state_.diag[k] : 0.0f; }
    std::pair<int, int> ridingAlongStats() const { return {(int)track_state_.diag[12], (int)track_state_.diag[13]}; }
    // since construction / resetAll: {tracked frames, evaluations the LM loop consumed, poses that rode along, of those consumed}
    std::vector<int64_t> trackerTotals() const { return {tracked_frames_, evals_total_, rode_total_, used_total_}; }
    void turnOnTracking(const char* levels = "rrbb", int numIterC = 20, int numIterF = 50, float outlierSpaceC = 0.1f,
                        float outlierSpaceF = 0.004f, float minstep = 1e-4f, float tukeyCutOff = 8.0f, int framesToSkip = 20,
                        int framesToWeight = 50);
    const gps_track_state& trackState() const { return track_state_; }
    // gps_track_poll_profile of this engine's tracker scratch (blocking; measurement only): {ticks waiting for the host's
    // argument line, ticks evaluating, evaluations, launches retired unused}, cumulative, 100 MHz ticks
    std::vector<int64_t> trackPollProfile() const {
        uint32_t out[4] = {0, 0, 0, 0};
        if (track_scratch_.defined())
            gpsh::check(gps_track_poll_profile(track_scratch_.data_ptr(), state_.width, state_.height, out, gpsh::current_stream()),
                        "gps_track_poll_profile");
        return {out[0], out[1], out[2], out[3]};
    }

    // One-shot hook for the next ProcessFrame: called on the calling thread after the frame's tracking and before its fusion
    // (gps_tsdf_process_frame_tracked_gated); with given poses it runs right before the fusion kernels are enqueued.
    std::function<void()> beforeNextFusion;

    std::vector<ORUtils::SE3Pose> camPoses;       // pose used for every processed frame (ITMBasicEngine.tpp:382)
    std::vector<ITMLib::ITMIntrinsics> camIntrincs;  // ... and its intrinsics (ITMBasicEngine.tpp:383)
    ITMLib::ITMIntrinsics intrinsics_d;            // view->calib.intrinsics_d
    std::vector<torch::Tensor> gtC2wPoses;         // dataset poses, [4,4] float CPU tensors (push before ProcessFrame)
    bool trackingActive = true;
    bool warned_rendering_blocks_ = false;
    int framesProcessed = 0;

private:
    gps_tsdf_state state_{};
    // state_.rgb / frame_inputs_ change per frame (frame thread) while a mapping worker may snapshot state_ for a free-view
    // raycast: the snapshot and the per-frame update take this lock (device-side ordering is the caller's events)
    mutable std::mutex state_mu_;
    torch::Device device_;
    torch::Tensor vba_, vba_alloc_list_, hash_, excess_list_, counters_, alloc_prio_, scan_scratch_, visible_type_,
        visible_ids_, depth_, minmax_, raycast_, icp_points_, icp_normals_, fv_visible_ids_, fv_minmax_, fv_raycast_,
        fv_colour_;
    struct FreeView {  // one ITMRenderState_VH worth of buffers + scratch + counters (gps_tsdf_view)
        torch::Tensor visible_ids, minmax, raycast, colour, scratch, counters;
        std::unique_ptr<ITMUChar4Image> image_p;
        std::unique_ptr<ITMFloat4Image> vertex_p;
    };
    std::vector<std::unique_ptr<FreeView>> views_;
    void ensureView(int k, const gps_tsdf_state& s);
    torch::Tensor view_table_;
    std::vector<torch::Tensor> frame_inputs_;
    torch::Tensor stage_rgb_[2], stage_depth_[2];  // UpdateView staging (host-resident input images), alternating per frame
    gps_track_config track_cfg_{};
    gps_track_state track_state_{};
    torch::Tensor track_scratch_, track_mailbox_;
    // the tracker's argument line in host-writable device memory (gps_track_arg_line_alloc; null without a large BAR).  The
    // switch exists for A/B measurements and for the tests that cover both hand-over paths.
    std::shared_ptr<void> track_arg_line_;
    bool bar_arg_line = true, host_summed_rows = true;
    static constexpr int kMaxRidingAlong = 3;
    int64_t tracked_frames_ = 0, evals_total_ = 0, rode_total_ = 0, used_total_ = 0;
    int poses_riding_along = 1;   // (0 / 1 / 2 measured on the 640x480 loop: 973 / 993 / 980 frames/s sequential, 1,316 / 1,324 / 1,306 overlap)
    ORUtils::SE3Pose pose_d_;
    ITMTrackingState tracking_state_{&pose_d_};
    ITMUChar4Image free_image_;
    ITMFloat4Image free_vertex_, live_vertex_;
};
