#include "infinitam_tools.hpp"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstring>
#include <map>
#include <mutex>

using namespace gpsh;
using namespace InfiniTAM::Engine;
using namespace ITMLib;

CLIEngine* CLIEngine::instance = NULL;

namespace {
inline void hip_ok(hipError_t e, const char* what) { TORCH_CHECK(e == hipSuccess, what, ": ", hipGetErrorString(e)); }
}  // namespace

// Three device slots: frame n is read by its kernels from slot n % 3 while frame n + 1 is uploaded into the next one; a slot
// is overwritten two frames later, after the event recorded behind its last reader.
struct CLIEngine::Staging {
    hipStream_t copy = nullptr;
    std::unique_ptr<ITMUChar4Image> rgb[3];
    std::unique_ptr<ITMShortImage> depth[3];
    hipEvent_t uploaded[3] = {nullptr, nullptr, nullptr}, consumed[3] = {nullptr, nullptr, nullptr};
    int frame_in_slot[3] = {-1, -1, -1};
    bool used[3] = {false, false, false};
    ~Staging() {
        if (copy) (void)hipStreamSynchronize(copy);   // (the stream itself lives as long as the process: copy_stream())
        for (int k = 0; k < 3; k++) {
            if (uploaded[k]) (void)hipEventDestroy(uploaded[k]);
            if (consumed[k]) (void)hipEventDestroy(consumed[k]);
        }
    }
};

// The upload stream: one per device for the life of the process (creating a stream blocks the runtime for ~10 ms; the copy
// engine's queue behind it is created at its first copy): made, and used once, the first time any sequence is initialised.
static hipStream_t copy_stream() {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    int dev = 0;
    hip_ok(hipGetDevice(&dev), "hipGetDevice");
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(dev);
    if (it == streams.end()) {
        hipStream_t st = nullptr;
        hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
        void *d = nullptr, *h = nullptr;
        hip_ok(hipMalloc(&d, 4096), "hipMalloc");
        hip_ok(hipHostMalloc(&h, 4096, hipHostMallocDefault), "hipHostMalloc");
        hip_ok(hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, st), "hipMemcpyAsync");   // (the copy engine's queue too)
        hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
        (void)hipFree(d); (void)hipHostFree(h);
        it = streams.emplace(dev, st).first;
    }
    return it->second;
}

void CLIEngine::Initialise(std::vector<ITMUChar4Image*> rgb_images_, std::vector<ITMShortImage*> depth_images_,
                           ITMMainEngine* mainEngine_) {
    TORCH_CHECK(!rgb_images_.empty() && rgb_images_.size() == depth_images_.size() && mainEngine_, "CLIEngine::Initialise");
    rgb_images = rgb_images_;
    depth_images = depth_images_;
    mainEngine = mainEngine_;
    currentFrameNo = 0;
    uploadedBytes = 0;
    staging_.reset(new Staging());
    staging_->copy = copy_stream();
    for (int k = 0; k < 3; k++) {
        staging_->rgb[k].reset(new ITMUChar4Image(GetRGBSize(), false, true));
        staging_->depth[k].reset(new ITMShortImage(GetDepthSize(), false, true));
        hip_ok(hipEventCreateWithFlags(&staging_->uploaded[k], hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&staging_->consumed[k], hipEventDisableTiming), "hipEventCreate");
    }
    // The staging images were zero-filled on the CURRENT stream just now (ORUtils::Image clears what it allocates); the first
    // uploads run on the copy stream: it waits for those fills, or a fill lands on top of frame 0 (seen once the copy stream
    // stopped being created -- a device-wide stall -- per engine: tests/test_tsdf_facade_gpu.py, counters of the two routes apart)
    hip_ok(hipEventRecord(staging_->consumed[0], (hipStream_t)current_stream()), "hipEventRecord");
    hip_ok(hipStreamWaitEvent(staging_->copy, staging_->consumed[0], 0), "hipStreamWaitEvent");
}

void CLIEngine::upload(int frame) {
    Staging& st = *staging_;
    const int slot = frame % 3;
    if (st.frame_in_slot[slot] == frame) return;
    ITMUChar4Image* rgb = rgb_images[frame];
    ITMShortImage* dep = depth_images[frame];
    TORCH_CHECK(rgb->isAllocated_CPU() && dep->isAllocated_CPU(), "CLIEngine: input images live in host memory");
    if (st.used[slot]) hip_ok(hipStreamWaitEvent(st.copy, st.consumed[slot], 0), "hipStreamWaitEvent");
    const size_t nb_rgb = rgb->dataSize() * sizeof(Vector4u), nb_d = dep->dataSize() * sizeof(short);
    hip_ok(hipMemcpyAsync(st.rgb[slot]->GetData(MEMORYDEVICE_CUDA), rgb->GetData(MEMORYDEVICE_CPU), nb_rgb, hipMemcpyHostToDevice,
                          st.copy), "UpdateView: rgb upload");
    hip_ok(hipMemcpyAsync(st.depth[slot]->GetData(MEMORYDEVICE_CUDA), dep->GetData(MEMORYDEVICE_CPU), nb_d, hipMemcpyHostToDevice,
                          st.copy), "UpdateView: depth upload");
    hip_ok(hipEventRecord(st.uploaded[slot], st.copy), "hipEventRecord");
    st.frame_in_slot[slot] = frame;
    uploadedBytes += (int64_t)(nb_rgb + nb_d);
}

bool CLIEngine::ProcessFrame() {
    if (currentFrameNo >= (int)rgb_images.size()) return false;
    const auto t0 = std::chrono::steady_clock::now();
    if (prefetch) {
        Staging& st = *staging_;
        const hipStream_t frames = (hipStream_t)current_stream();
        const int slot = currentFrameNo % 3;
        upload(currentFrameNo);                                                   // no-op when the previous call prefetched it
        hip_ok(hipStreamWaitEvent(frames, st.uploaded[slot], 0), "hipStreamWaitEvent");
        if (currentFrameNo + 1 < (int)rgb_images.size()) upload(currentFrameNo + 1);  // overlaps this frame's kernels
        mainEngine->ProcessFrame(st.rgb[slot].get(), st.depth[slot].get());
        hip_ok(hipEventRecord(st.consumed[slot], frames), "hipEventRecord");
        st.used[slot] = true;
    } else {
        mainEngine->ProcessFrame(rgb_images[currentFrameNo], depth_images[currentFrameNo]);  // engine uploads on its stream
        uploadedBytes += (int64_t)(rgb_images[currentFrameNo]->dataSize() * 6);
    }
    processedTime = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    currentFrameNo++;
    return true;
}

// The staging slot of the frame ProcessFrame() handed to the engine has readers BEHIND the engine's own kernels: the pipeline
// derives the camera's float image from it (Camera::toGPU -> gps_rgba8_to_rgbf_and_floats) on the frame stream.  Re-recording
// the slot's `consumed` event here moves it behind that last reader; upload(n + 3) waits for the re-recorded event.
void CLIEngine::markConsumed() {
    if (!prefetch || !staging_ || currentFrameNo == 0) return;
    Staging& st = *staging_;
    const int slot = (currentFrameNo - 1) % 3;
    if (!st.used[slot]) return;
    hip_ok(hipEventRecord(st.consumed[slot], (hipStream_t)current_stream()), "hipEventRecord");
}

void CLIEngine::Run() {
    while (ProcessFrame()) {}
}

void CLIEngine::Shutdown() {
    if (beforeShutdown) { auto detach = std::move(beforeShutdown); beforeShutdown = nullptr; detach(); }
    staging_.reset();
    // createTsdfEngine handed this object the sequence's images and the engine it built for it (ownsInputs): a process that builds
    // scene after scene -- bench.py, the tests -- gets the pinned images (1.8 MB per 640x480 frame) and the engine's 1.5 GB back
    // here; the reference's Shutdown leaves both to the end of the process (CLIEngine.cpp:69-77)
    if (ownsInputs) {
        (void)hipDeviceSynchronize();
        for (auto* im : rgb_images) delete im;
        for (auto* im : depth_images) delete im;
        delete mainEngine;
    }
    ownsInputs = false;
    rgb_images.clear();
    depth_images.clear();
    mainEngine = nullptr;
    delete instance;
    instance = NULL;
}

CLIEngine* createTsdfEngine(const DatasetReader& data_reader, const Config& config) {
    // 1. calibration (InfiniTAM_tools.cpp:5-10)
    ITMRGBDCalib rgbd_calib;
    rgbd_calib.intrinsics_rgb.SetFrom(data_reader.width, data_reader.height, data_reader.fx, data_reader.fy, data_reader.cx,
                                      data_reader.cy);
    rgbd_calib.intrinsics_d = rgbd_calib.intrinsics_rgb;
    rgbd_calib.disparityCalib.SetStandard();
    // 2. InfiniTAM images, host memory (:12-46)
    const int image_num = (int)data_reader.train_vec.size();
    TORCH_CHECK(image_num > 0, "createTsdfEngine: empty dataset");
    std::vector<ITMUChar4Image*> rgb_images(image_num);
    std::vector<ITMShortImage*> depth_images(image_num);
    std::vector<torch::Tensor> gt_c2w_poses(image_num);
    const Vector2i dims(data_reader.width, data_reader.height);
    const int64_t P = (int64_t)dims.x * dims.y;
    for (int i = 0; i < image_num; i++) {
        const Camera& cam = data_reader.train_vec[i];
        TORCH_CHECK(cam.image.defined() && cam.depth.defined(), "createTsdfEngine: camera ", i, " has no image / depth");
        // The image as the reader holds it -- float [H,W,3] in [0,1] (the reference's DatasetReader) -- or as the dataset's FILES hold
        // it, uint8 [H,W,3]: a thousand-frame sequence is 4.9 GB as floats and 1.5 GB as bytes.  The reference turns a file's
        // byte k into the float k / 255 (imageToTensor, cv_utils.cpp:208-213) and back with (t * 255).toType(uint8) (truncation): the 256-entry
        // table below is that round trip, so both inputs give the same uchar4 frame.
        rgb_images[i] = new ITMUChar4Image(dims, true, false);
        {
            if (cam.image.scalar_type() == torch::kUInt8) {
                static const torch::Tensor lut = ((torch::arange(256, torch::kFloat32) / 255.0) * 255.0).toType(torch::kUInt8);
                auto raw = cam.image.detach().to(torch::kCPU).contiguous();
                TORCH_CHECK(raw.dim() == 3 && raw.size(0) == dims.y && raw.size(1) == dims.x && raw.size(2) == 3,
                            "Only images with 3 channels are supported");
                const uint8_t* t = lut.data_ptr<uint8_t>();
                const uint8_t* src = raw.data_ptr<uint8_t>();
                Vector4u* dst = rgb_images[i]->GetData(MEMORYDEVICE_CPU);
                for (int64_t k = 0; k < P; k++) dst[k] = Vector4u{t[src[3 * k]], t[src[3 * k + 1]], t[src[3 * k + 2]], 255};
            } else {
                auto img = cam.image.detach().to(torch::kCPU, torch::kFloat32);
                TORCH_CHECK(img.dim() == 3 && img.size(0) == dims.y && img.size(1) == dims.x && img.size(2) == 3,
                            "Only images with 3 channels are supported");
                auto u8img = (img * 255.0).toType(torch::kUInt8).contiguous();  // tensorToImage: truncation, not rounding
                const uint8_t* src = u8img.data_ptr<uint8_t>();
                Vector4u* dst = rgb_images[i]->GetData(MEMORYDEVICE_CPU);
                for (int64_t k = 0; k < P; k++) dst[k] = Vector4u{src[3 * k], src[3 * k + 1], src[3 * k + 2], 255};
            }
        }
        TORCH_CHECK(cam.depth.numel() == P, "Only images with 1 channels are supported");
        depth_images[i] = new ITMShortImage(dims, true, false);
        if (cam.depth.scalar_type() == torch::kInt16 || cam.depth.scalar_type() == torch::kUInt16) {
            // millimetres as the depth PNG holds them: the reference's float metres (mm / 1000.f) come back as the same integer
            // under convertTo(CV_16UC1, 1000) for every uint16 (|error| < 0.008 mm), so the shorts are copied as they are
            auto raw = cam.depth.detach().to(torch::kCPU).contiguous();
            std::memcpy(depth_images[i]->GetData(MEMORYDEVICE_CPU), raw.data_ptr(), (size_t)P * sizeof(short));
        } else {
            auto d = cam.depth.detach().to(torch::kCPU, torch::kFloat32);
            // cv::Mat::convertTo(CV_16UC1, 1000): saturate_cast<ushort>(cvRound(v * 1000)), round half to even
            auto mm = torch::round(d * 1000.0f).clamp(0.0, 65535.0).to(torch::kInt32).contiguous();
            const int32_t* src = mm.data_ptr<int32_t>();
            short* dst = depth_images[i]->GetData(MEMORYDEVICE_CPU);
            for (int64_t k = 0; k < P; k++) dst[k] = (short)(unsigned short)src[k];
        }
        gt_c2w_poses[i] = cam.c2w.to(torch::kCPU, torch::kFloat32).contiguous();
    }
    // 3. main engine (:48-63)
    ITMLibSettings* internalSettings = new ITMLibSettings();
    internalSettings->sceneParams.voxelSize = (float)config.get("voxel_size", 0.005);
    internalSettings->sceneParams.mu = (float)config.get("trunc_dist", 0.02);
    internalSettings->sceneParams.viewFrustum_min = (float)config.get("viewFrustum_min", 0.2);
    internalSettings->sceneParams.viewFrustum_max = (float)config.get("viewFrustum_max", 10.0);
    // table sizes: compile-time constants of the reference (SDF_LOCAL_BLOCK_NUM 0x40000, SDF_BUCKET_NUM 0x100000, SDF_EXCESS_LIST_SIZE
    // 0x20000, ITMLib/Objects/Scene/ITMVoxelBlockHash.h:15-22) and the defaults here; smaller tables are a test aid (the CPU oracle's
    // per-frame cost is its sweeps over them)
    internalSettings->noTotalEntries_blocks = (int)config.get("sdf_local_block_num", internalSettings->noTotalEntries_blocks);
    internalSettings->noBuckets = (int)config.get("sdf_bucket_num", internalSettings->noBuckets);
    internalSettings->excessListSize = (int)config.get("sdf_excess_list_size", internalSettings->excessListSize);
    ITMMainEngine* mainEngine =
        new ITMBasicEngine<ITMVoxel, ITMVoxelIndex>(internalSettings, rgbd_calib, rgb_images[0]->noDims, depth_images[0]->noDims);
    if (config.get("use_gt_pose", 1.0) != 0.0) {
        auto* be = dynamic_cast<ITMBasicEngine<ITMVoxel, ITMVoxelIndex>*>(mainEngine);
        be->turnOffTracking();
        be->gtC2wPoses = gt_c2w_poses;
    }
    // 4. CLI engine (:64-67)
    CLIEngine* tsdf_engine = CLIEngine::Instance();
    tsdf_engine->Initialise(rgb_images, depth_images, mainEngine);
    tsdf_engine->ownsInputs = true;
    return tsdf_engine;
}
