// The TSDF engine surface SLAMPipeline drives (slam/slam_pipeline.cpp:69-83, 362-415; slam/InfiniTAM_tools.cpp:3-67;
// ITMLib/Core/ITMBasicEngine.{h,tpp}) on the C-ABI.  ITMBasicEngine<ITMVoxel_s_rgb, ITMVoxelBlockHash> as the
// shipped configs run it: tracking off (use_gt_pose: true), no swapping, no meshing on the hot path.
//
// The engine owns the device buffers (through libtorch, where the reference uses ORUtils::MemoryBlock), fills the
// gps_tsdf_state once and afterwards only passes pointers.  Method and member names follow the reference so that
// slam_pipeline-style code reads the same: ProcessFrame, runRaycast, GetFreeImage, GetFreeVertex, getVoxelSize,
// GetTrackingState()->pose_d->GetInvM(), camPoses, gtC2wPoses.
#pragma once
#include <functional>

#include "gps_host_common.hpp"

enum MemoryDeviceType { MEMORYDEVICE_CPU, MEMORYDEVICE_CUDA };

namespace ORUtils {

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

// ORUtils/Image.h as the pipeline uses it: a device image whose GetData(MEMORYDEVICE_CUDA) pointer torch::from_blob
// can wrap (src/cv_utils.cpp:324-336)
template <class T>
class Image {
public:
    struct { int x, y; } noDims;
    Image(int w, int h, torch::Tensor storage) : storage_(storage) { noDims.x = w; noDims.y = h; }
    T* GetData(MemoryDeviceType t) const {
        TORCH_CHECK(t == MEMORYDEVICE_CUDA, "the engine keeps images on the device only");
        return reinterpret_cast<T*>(storage_.data_ptr());
    }
    const torch::Tensor& tensor() const { return storage_; }

private:
    torch::Tensor storage_;
};

}  // namespace ORUtils

struct Vector4u { unsigned char x, y, z, w; };
struct Vector4f { float x, y, z, w; };
typedef ORUtils::Image<Vector4u> ITMUChar4Image;
typedef ORUtils::Image<Vector4f> ITMFloat4Image;

struct ITMTrackingState { ORUtils::SE3Pose* pose_d; };

class ITMBasicEngine {
public:
    // capacities: ITMLib/Objects/Scene/ITMVoxelBlockHash.h:18-22
    ITMBasicEngine(int width, int height, float fx, float fy, float cx, float cy, float voxel_size = 0.005f,
                   float mu = 0.02f, float view_frustum_min = 0.2f, float view_frustum_max = 10.0f,
                   int n_blocks = 0x40000, int n_buckets = 0x100000, int n_excess = 0x20000,
                   torch::Device device = torch::kCUDA);

    void resetAll();
    // ITMBasicEngine::ProcessFrame with tracking off: pose := gtC2wPoses[framesProcessed] (ITMBasicEngine.tpp:260-385).
    // rgb uint8[H,W,4] (uchar4) and depth int16[H,W] (mm) device tensors are read in place.
    ITMTrackingState* ProcessFrame(const torch::Tensor& rgb_u8, const torch::Tensor& depth_mm_i16);
    // ITMBasicEngine::runRaycast(pose, intrinsics) (ITMBasicEngine.tpp:519-525)
    void runRaycast(ORUtils::SE3Pose* pose);
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
    const gps_tsdf_state& state() const { return state_; }
    torch::Tensor counters() const { return counters_; }

    // ITMBasicEngine::turnOffTracking (ITMBasicEngine.tpp:532; createTsdfEngine calls it when use_gt_pose is true,
    // InfiniTAM_tools.cpp:59-62): poses then come from gtC2wPoses.  With tracking active (the reference's default) the
    // depth-only ExtendedTracker of ITMLibSettings.cpp:54-57 estimates them (gps_tsdf_process_frame_tracked).
    void turnOffTracking() { trackingActive = false; }
    void turnOnTracking(const char* levels = "rrbb", int numIterC = 20, int numIterF = 50, float outlierSpaceC = 0.1f,
                        float outlierSpaceF = 0.004f, float minstep = 1e-4f, float tukeyCutOff = 8.0f, int framesToSkip = 20,
                        int framesToWeight = 50);
    const gps_track_state& trackState() const { return track_state_; }

    // One-shot hook for the next ProcessFrame: called on the calling thread after the frame's tracking and before its fusion
    // (gps_tsdf_process_frame_tracked_gated); with given poses it runs right before the fusion kernels are enqueued.
    std::function<void()> beforeNextFusion;

    std::vector<ORUtils::SE3Pose> camPoses;       // pose used for every processed frame
    std::vector<torch::Tensor> gtC2wPoses;         // dataset poses, [4,4] float CPU tensors (push before ProcessFrame)
    bool trackingActive = true;
    int framesProcessed = 0;

private:
    gps_tsdf_state state_{};
    torch::Device device_;
    torch::Tensor vba_, vba_alloc_list_, hash_, excess_list_, counters_, alloc_prio_, scan_scratch_, visible_type_,
        visible_ids_, depth_, minmax_, raycast_, icp_points_, icp_normals_, fv_visible_ids_, fv_minmax_, fv_raycast_,
        fv_colour_;
    std::vector<torch::Tensor> frame_inputs_;
    gps_track_config track_cfg_{};
    gps_track_state track_state_{};
    torch::Tensor track_scratch_, track_mailbox_;
    ORUtils::SE3Pose pose_d_;
    ITMTrackingState tracking_state_{&pose_d_};
    ITMUChar4Image free_image_;
    ITMFloat4Image free_vertex_, live_vertex_;
};
