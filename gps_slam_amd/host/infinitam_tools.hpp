// The TsdfFusion facade slam_trainer.cpp:26-29 and slam_pipeline.{h,cpp} program against:
//
//   ITMLib::ITMMainEngine / ITMLib::ITMBasicEngine<TVoxel, TIndex>   ITMLib/Core/ITMMainEngine.h, ITMBasicEngine.h:54-92
//   InfiniTAM::Engine::CLIEngine                                      slam/TsdfFusion/CLIEngine.{h,cpp}
//   createTsdfEngine(const DatasetReader&, config)                    slam/InfiniTAM_tools.{h,cpp}
//
// with the reference's names, constructor / method signatures and member names, over TsdfEngine (tsdf_engine.hpp) and the
// C-ABI.  What differs, and why: `config` is gpsh::Config instead of YAML::Node and DatasetReader is the plain struct below
// instead of the OpenCV/yaml-cpp reader of include/dataset_reader.h (I/O, out of scope); CLIEngine keeps the whole sequence in
// PINNED host memory and, knowing the next frame, uploads it on a copy stream while the current one is processed (the
// reference's UpdateView is a blocking cudaMemcpy per frame, ITMViewBuilder_CUDA.cu:61-62) -- every frame's 6 bytes per
// pixel still cross PCIe inside the frame loop.
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include "raw_gs_param.hpp"
#include "tsdf_engine.hpp"

namespace ITMLib {

// ITMLib/Core/ITMMainEngine.h:39-88 (the members GPS-SLAM calls)
class ITMMainEngine {
public:
    virtual ~ITMMainEngine() {}
    virtual ITMTrackingState* GetTrackingState(void) = 0;
    virtual ITMTrackingState::TrackingResult ProcessFrame(ITMUChar4Image* rgbImage, ITMShortImage* rawDepthImage,
                                                          ITMIMUMeasurement* imuMeasurement = NULL) = 0;
    virtual void SaveSceneToMesh(const char* fileName) = 0;
    virtual void SaveToFile(const std::string& saveOutputDirectory) = 0;
    virtual void LoadFromFile(const std::string& saveInputDirectory) = 0;
    virtual Vector2i GetImageSize(void) const = 0;
    virtual void turnOnTracking() = 0;
    virtual void turnOffTracking() = 0;
};

// ITMLib/Core/ITMBasicEngine.h:21-115.  TVoxel / TIndex select the storage layouts; the one combination GPS-SLAM
// instantiates (ITMVoxel_s_rgb + ITMVoxelBlockHash, ITMLibDefines.h:18-25) is what the kernels implement.
template <typename TVoxel, typename TIndex>
class ITMBasicEngine : public ITMMainEngine, public TsdfEngine {
    static_assert(std::is_same<TVoxel, ITMVoxel_s_rgb>::value && std::is_same<TIndex, ITMVoxelBlockHash>::value,
                  "the gfx950 kernels implement ITMVoxel_s_rgb voxels in an ITMVoxelBlockHash index");

public:
    // ITMBasicEngine.h:104: imgSize_d = (-1,-1) means "same as rgb"
    ITMBasicEngine(const ITMLibSettings* settings, const ITMRGBDCalib& calib, Vector2i imgSize_rgb,
                   Vector2i imgSize_d = Vector2i(-1, -1))
        : TsdfEngine(imgSize_d.x == -1 ? imgSize_rgb.x : imgSize_d.x, imgSize_d.x == -1 ? imgSize_rgb.y : imgSize_d.y,
                     calib.intrinsics_d.projectionParamsSimple.fx, calib.intrinsics_d.projectionParamsSimple.fy,
                     calib.intrinsics_d.projectionParamsSimple.px, calib.intrinsics_d.projectionParamsSimple.py,
                     settings->sceneParams.voxelSize, settings->sceneParams.mu, settings->sceneParams.viewFrustum_min,
                     settings->sceneParams.viewFrustum_max, settings->noTotalEntries_blocks, settings->noBuckets,
                     settings->excessListSize) {
        TORCH_CHECK(imgSize_d.x == -1 || (imgSize_d.x == imgSize_rgb.x && imgSize_d.y == imgSize_rgb.y),
                    "rgb and depth images of one size (as createTsdfEngine passes them)");
        intrinsics_d = calib.intrinsics_d;
    }

    ITMTrackingState* GetTrackingState(void) override { return TsdfEngine::GetTrackingState(); }
    ITMTrackingState::TrackingResult ProcessFrame(ITMUChar4Image* rgbImage, ITMShortImage* rawDepthImage,
                                                  ITMIMUMeasurement* imuMeasurement = NULL) override {
        TORCH_CHECK(imuMeasurement == NULL, "IMU measurements are not used by GPS-SLAM");
        return TsdfEngine::ProcessFrame(rgbImage, rawDepthImage)->trackerResult;
    }
    using TsdfEngine::ProcessFrame;  // + the tensor overload for frames that already live in HBM
    void SaveSceneToMesh(const char* fileName) override { (void)TsdfEngine::SaveSceneToMesh(fileName); }
    void SaveToFile(const std::string& d) override { TsdfEngine::SaveToFile(d); }
    void LoadFromFile(const std::string& d) override { TsdfEngine::LoadFromFile(d); }
    Vector2i GetImageSize(void) const override { return Vector2i(state().width, state().height); }
    void turnOnTracking() override { TsdfEngine::turnOnTracking(); }
    void turnOffTracking() override { TsdfEngine::turnOffTracking(); }
};

}  // namespace ITMLib

namespace InfiniTAM {
namespace Engine {

// slam/TsdfFusion/CLIEngine.h:14-66: process-wide singleton that owns the input sequence and feeds it to the main engine
class CLIEngine {
    static CLIEngine* instance;
    std::vector<ITMUChar4Image*> rgb_images;
    std::vector<ITMShortImage*> depth_images;
    ITMLib::ITMMainEngine* mainEngine = nullptr;

public:
    static CLIEngine* Instance(void) {
        if (instance == NULL) instance = new CLIEngine();
        return instance;
    }
    float processedTime = 0.f;
    int currentFrameNo = 0;  // (private in the reference; slam_pipeline.cpp:77 asserts on it, which only compiles under NDEBUG)

    void Initialise(std::vector<ITMUChar4Image*> rgb_images, std::vector<ITMShortImage*> depth_images,
                    ITMLib::ITMMainEngine* mainEngine);
    void Shutdown();
    void Run();
    bool ProcessFrame();
    // call on the stream of the LAST reader of the frame ProcessFrame() just consumed (e.g. after deriving the camera's float
    // image from currentRgb()): the staging slot is not overwritten by a later upload before that reader has run
    void markConsumed();
    Vector2i GetDepthSize() { return depth_images[0]->noDims; }
    Vector2i GetRGBSize() { return rgb_images[0]->noDims; }
    ITMLib::ITMMainEngine* getMainEngine() { return mainEngine; }

    // Upload pipelining (see the file comment).  prefetch = false reproduces the reference's schedule: the frame's upload is
    // enqueued on the frame's own stream right before its kernels.
    bool prefetch = true;
    int64_t uploadedBytes = 0;
    // called first thing by Shutdown(): a SLAMPipeline attached with setTsdfEngine() flushes its map worker and drops its raw
    // pointers into this engine (and the engine's beforeNextFusion hook that captures the pipeline) before the engine is freed
    // (round-5 advisor finding: bench.Scene.close() shuts the engine down while the pipeline is still alive)
    std::function<void()> beforeShutdown;
    bool ownsInputs = false;   // Shutdown() deletes the images and the engine (set by createTsdfEngine, which allocated them)

private:
    void upload(int frame);  // pinned host images of `frame` -> staging slot frame % 3, on the copy stream
    struct Staging;
    std::shared_ptr<Staging> staging_;
};

}  // namespace Engine
}  // namespace InfiniTAM

// include/dataset_reader.h:111-169, 171-230: what createTsdfEngine reads from the reader -- image size, pinhole intrinsics
// and the training cameras (image float [H,W,3] in [0,1], depth float [H,W,1] metres, c2w [4,4]), on the host.  createTsdfEngine also
// takes the image as uint8 [H,W,3] and the depth as (u)int16 millimetres -- what the dataset's files hold -- with the same bytes out
struct DatasetReader {
    int width = 0, height = 0;
    float fx = 0.f, fy = 0.f, cx = 0.f, cy = 0.f;
    std::vector<Camera> train_vec;
};

// slam/InfiniTAM_tools.cpp:3-67.  config keys: voxel_size, trunc_dist, viewFrustum_min, viewFrustum_max, use_gt_pose (0/1).
// Image conversion as the reference: rgb = (image * 255).toType(uint8), alpha 255 (cv_utils.cpp:57-76, 221-246);
// depth = round-half-even(depth * 1000) saturated to uint16, stored in shorts (cv_utils.cpp:79-101, 248-268).
InfiniTAM::Engine::CLIEngine* createTsdfEngine(const DatasetReader& data_reader, const gpsh::Config& config);
