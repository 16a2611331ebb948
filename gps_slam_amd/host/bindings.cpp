// pybind11 view of the C++ host layer (module gps_slam_amd._host): lets the Python tests and bench.py drive exactly the
// C++ classes a C++ slam_trainer would link against.
#include <torch/extension.h>

#include "slam_pipeline.hpp"

namespace py = pybind11;

static gpsh::Config config_from_dict(const py::dict& d) {
    gpsh::Config c;
    for (auto item : d) {
        const std::string k = py::str(item.first);
        if (py::isinstance<py::str>(item.second)) c.str[k] = py::str(item.second);
        else c.num[k] = item.second.cast<double>();
    }
    return c;
}

PYBIND11_MODULE(_host, m) {
    m.doc() = "C++ host layer of gps_slam_amd over the C-ABI (libgpsslam_hip.so)";

    // ---- operator surface (gsplat_wapper.hpp)
    m.def("SphericalHarmonicsNew", [](int deg, torch::Tensor dirs, torch::Tensor coeffs, torch::Tensor masks) {
        return SphericalHarmonicsNew::apply(deg, dirs, coeffs, masks);
    });
    m.def("FullyFusedProjection", [](torch::Tensor means, torch::Tensor quats, torch::Tensor scales, torch::Tensor viewmats,
                                    torch::Tensor Ks, int w, int h, float eps2d, float near_plane, float far_plane,
                                    float radius_clip) {
        return FullyFusedProjection::apply(means, c10::nullopt, quats, scales, viewmats, Ks, w, h, eps2d, near_plane,
                                           far_plane, radius_clip, false, std::string("pinhole"));
    });
    m.def("RasterizeToPixelsGes_NewParallel",
          [](torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities, torch::Tensor radiis,
             torch::Tensor ref_depth_map, torch::Tensor base_color_map, int w, int h, int tile_size,
             torch::Tensor isect_offsets, torch::Tensor flatten_ids, torch::Tensor group_gs_ids, torch::Tensor group_starts,
             float delta_depth) {
              return RasterizeToPixelsGes_NewParallel::apply(means2d, conics, colors, opacities, radiis, ref_depth_map,
                                                             base_color_map, c10::nullopt, c10::nullopt, w, h, tile_size,
                                                             isect_offsets, flatten_ids, group_gs_ids, group_starts,
                                                             false, delta_depth);
          });
    m.def("RasterizeToPixelsGes",
          [](torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
             torch::Tensor ref_depth_map, torch::Tensor base_color_map, int w, int h, int tile_size,
             torch::Tensor isect_offsets, torch::Tensor flatten_ids, float delta_depth) {
              return RasterizeToPixelsGes::apply(means2d, conics, colors, opacities, ref_depth_map, base_color_map, c10::nullopt,
                                                 c10::nullopt, w, h, tile_size, isect_offsets, flatten_ids, false, delta_depth);
          });
    m.def("RasterizeToPixels",
          [](torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
             c10::optional<torch::Tensor> backgrounds, int w, int h, int tile_size, torch::Tensor isect_offsets,
             torch::Tensor flatten_ids, bool absgrad) {
              return RasterizeToPixels::apply(means2d, conics, colors, opacities, backgrounds, c10::nullopt, w, h, tile_size,
                                              isect_offsets, flatten_ids, absgrad);
          });
    m.def("FusedSSIMMap", [](double C1, double C2, torch::Tensor img1, torch::Tensor img2, std::string padding, bool train) {
        return FusedSSIMMap::apply(C1, C2, img1, img2, padding, train);
    });
    m.def("isectTiles", &isectTiles, py::arg("means2d"), py::arg("radii"), py::arg("depths"), py::arg("tile_size"),
          py::arg("tile_width"), py::arg("tile_height"), py::arg("sort") = true);
    m.def("isectOffsetEncode", &isectOffsetEncode);
    m.def("isectTilesNoDepth", &isectTilesNoDepth, py::arg("means2d"), py::arg("radii"), py::arg("depths"),
          py::arg("tile_size"), py::arg("tile_width"), py::arg("tile_height"), py::arg("sort") = true);
    m.def("isectOffsetEncodeNoDepth", &isectOffsetEncodeNoDepth);
    m.def("distCUDA2", &distCUDA2);
    m.def("simpleKNN", &simpleKNN);
    m.def("degFromSh", &degFromSh);
    m.def("numShBases", &numShBases);
    m.def("rgb2sh", &rgb2sh);
    m.def("sh2rgb", &sh2rgb);
    m.def("poseInv", &poseInv);
    m.def("RawGaussianParamsMake", &RawGaussianParams::make, py::arg("xyz"), py::arg("rgb"), py::arg("normals"),
          py::arg("max_sh_degree") = 3, py::arg("init_opacs") = 0.5f, py::arg("max_scale") = 0.01f, py::arg("min_scale") = -1.0f);
    m.def("computeNormalMap", &computeNormalMap);
    m.def("getGPUMemoryUsage", &getGPUMemoryUsage, py::arg("gpu_id") = 0);

    // ---- Camera
    py::class_<Camera>(m, "Camera")
        .def(py::init<int, int, float, float, float, float, bool, const torch::Tensor&>())
        .def_readwrite("id", &Camera::id)
        .def_readwrite("width", &Camera::width)
        .def_readwrite("height", &Camera::height)
        .def_readwrite("fx", &Camera::fx)
        .def_readwrite("fy", &Camera::fy)
        .def_readwrite("cx", &Camera::cx)
        .def_readwrite("cy", &Camera::cy)
        .def_readwrite("image", &Camera::image)
        .def_readwrite("depth", &Camera::depth)
        .def_readwrite("c2w", &Camera::c2w)
        .def_readwrite("c2w_slam", &Camera::c2w_slam)
        .def("toGPU", [](Camera& c) { c.toGPU(); })
        .def("invalidate", &Camera::invalidate)
        .def("viewmat", &Camera::viewmat_tensor)
        .def("K_dev", &Camera::K_tensor)
        .def("cam_pos", &Camera::cam_pos_tensor);

    // ---- model
    py::class_<RawGaussianParams>(m, "RawGaussianParams")
        .def("getGaussianNum", &RawGaussianParams::getGaussianNum)
        .def("getMeans", &RawGaussianParams::getMeans)
        .def("getScales", &RawGaussianParams::getScales)
        .def("getQuats", &RawGaussianParams::getQuats)
        .def("getFeaturesDc", &RawGaussianParams::getFeaturesDc)
        .def("getFeaturesRest", &RawGaussianParams::getFeaturesRest)
        .def("getOpacities", &RawGaussianParams::getOpacities)
        .def("add", [](RawGaussianParams& p, std::vector<torch::Tensor> t) { p.add(t); })
        .def("remove", &RawGaussianParams::remove)
        .def("savePly", &RawGaussianParams::savePly)
        .def("saveTensor", &RawGaussianParams::saveTensor)
        .def("loadTensor", &RawGaussianParams::loadTensor)
        .def_static("make", &RawGaussianParams::make);

    py::class_<SLAMGaussianModel>(m, "SLAMGaussianModel")
        .def(py::init<>())
        .def("loadConfig", [](SLAMGaussianModel& s, const py::dict& d) { s.loadConfig(config_from_dict(d)); })
        .def("forward", &SLAMGaussianModel::forward, py::arg("cam"), py::arg("ref_depth"), py::arg("base_color"))
        .def("rawForward", &SLAMGaussianModel::rawForward)
        .def("setBackgrounds", [](SLAMGaussianModel& s, c10::optional<torch::Tensor> bg) {
            s.backgrounds = bg.has_value() ? *bg : torch::Tensor();
        })
        .def_readwrite("render_method", &SLAMGaussianModel::render_method)
        .def("computeLoss", [](SLAMGaussianModel& s, TensorDict& r, const Camera& cam, const py::dict& w) {
            return s.computeLoss(r, cam, config_from_dict(w));
        })
        .def("trainStep", [](SLAMGaussianModel& s, const Camera& cam, const torch::Tensor& ref_depth,
                             const torch::Tensor& base_color, c10::optional<torch::Tensor> clamped, const Camera* next_cam) {
            s.trainStep(cam, ref_depth, base_color, clamped.has_value() ? *clamped : torch::Tensor(), next_cam);
        }, py::arg("cam"), py::arg("ref_depth"), py::arg("base_color"), py::arg("ref_depth_clamped") = py::none(),
             py::arg("next_cam") = (const Camera*)nullptr)
        .def("reserveWorkspace", &SLAMGaussianModel::reserveWorkspace)
        .def("checkBinningCapacity", &SLAMGaussianModel::checkBinningCapacity)
        .def_readonly("binning_overflows", &SLAMGaussianModel::binning_overflows)
        .def("capacity", [](SLAMGaussianModel& s) { return s.getGaussianParms().capacity(); })
        .def("adamState", [](SLAMGaussianModel& s) { return s.adamState(); })
        .def("lossSum", &SLAMGaussianModel::lossSum)
        .def("initOptimizers", &SLAMGaussianModel::initOptimizers, py::arg("max_iterations") = -1,
             py::arg("scene_scale") = 1.0f)
        .def("optimizersStep", &SLAMGaussianModel::optimizersStep)
        .def("optimizersZeroGrad", &SLAMGaussianModel::optimizersZeroGrad)
        .def("prunePoints", &SLAMGaussianModel::prunePoints)
        .def("grads", &SLAMGaussianModel::grads)
        .def("getGaussianNum", &SLAMGaussianModel::getGaussianNum)
        .def("getGaussianParms", &SLAMGaussianModel::getGaussianParms, py::return_value_policy::reference_internal)
        .def("getRealScales", &SLAMGaussianModel::getRealScales)
        .def("getRealOpacities", &SLAMGaussianModel::getRealOpacities)
        .def("addGaussians", [](SLAMGaussianModel& s, const Camera& cam, const TensorDict& maps, const torch::Tensor& mask,
                                float ratio, int frame_num) { return s.addGaussians(cam, maps, mask, ratio, frame_num); });

    // ---- TSDF engine
    py::class_<TsdfEngine>(m, "ITMBasicEngine")
        .def(py::init([](int w, int h, float fx, float fy, float cx, float cy, float voxel_size, float mu, float vmin,
                         float vmax) { return new TsdfEngine(w, h, fx, fy, cx, cy, voxel_size, mu, vmin, vmax); }),
             py::arg("width"), py::arg("height"), py::arg("fx"), py::arg("fy"), py::arg("cx"), py::arg("cy"),
             py::arg("voxel_size") = 0.005f, py::arg("mu") = 0.02f, py::arg("view_frustum_min") = 0.2f,
             py::arg("view_frustum_max") = 10.0f)
        .def("pushGtPose", [](TsdfEngine& e, const torch::Tensor& c2w) { e.gtC2wPoses.push_back(c2w); })
        .def("turnOffTracking", &TsdfEngine::turnOffTracking)
        .def("turnOnTracking", [](TsdfEngine& e) { e.turnOnTracking(); })
        .def("setBarArgLine", &TsdfEngine::setBarArgLine)
        .def("usesBarArgLine", &TsdfEngine::usesBarArgLine)
        .def("setPosesRidingAlong", &TsdfEngine::setPosesRidingAlong)
        .def("posesRidingAlong", &TsdfEngine::posesRidingAlong)
        .def("setHostSummedRows", &TsdfEngine::setHostSummedRows)
        .def("hostSummedRows", &TsdfEngine::hostSummedRows)
        .def("ridingAlongStats", &TsdfEngine::ridingAlongStats)
        .def("trackerTotals", &TsdfEngine::trackerTotals)
        .def("lastPose", [](TsdfEngine& e) {
            auto t = torch::empty({2, 16}, torch::kFloat32);
            const ORUtils::SE3Pose& p = e.camPoses.back();
            for (int i = 0; i < 16; i++) { t[0][i] = p.GetM()[i]; t[1][i] = p.GetInvM()[i]; }
            return t;
        })
        .def("trackDiag", [](TsdfEngine& e) {
            auto t = torch::empty({16}, torch::kFloat32);
            for (int i = 0; i < 16; i++) t[i] = e.trackState().diag[i];
            return t;
        })
        .def("trackPollProfile", &TsdfEngine::trackPollProfile)
        .def("ProcessFrame", [](TsdfEngine& e, const torch::Tensor& rgb, const torch::Tensor& depth) {
            e.ProcessFrame(rgb, depth);
        })
        .def("runRaycastC2w", [](TsdfEngine& e, const torch::Tensor& c2w) {
            ORUtils::SE3Pose pose;
            auto c = c2w.to(torch::kCPU, torch::kFloat32).contiguous();
            pose.SetInvM(c.data_ptr<float>());
            e.runRaycast(&pose);
        })
        .def("GetFreeImage", [](TsdfEngine& e) { return e.GetFreeImage()->tensor(); })
        .def("GetFreeVertex", [](TsdfEngine& e) { return e.GetFreeVertex()->tensor(); })
        .def("GetLiveVertex", [](TsdfEngine& e) { return e.GetLiveVertex()->tensor(); })
        .def("getVoxelSize", &TsdfEngine::getVoxelSize)
        .def("counters", &TsdfEngine::counters)
        .def("MeshScene", &TsdfEngine::MeshScene, py::arg("maxTriangles") = (int64_t)1 << 24)
        .def("SaveSceneToMesh", [](TsdfEngine& e, const std::string& f, int64_t m) { return e.SaveSceneToMesh(f.c_str(), m); },
             py::arg("fileName"), py::arg("maxTriangles") = (int64_t)1 << 24)
        .def("SaveToFile", &TsdfEngine::SaveToFile)
        .def("LoadFromFile", &TsdfEngine::LoadFromFile)
        .def("runRaycastIntrinsics", [](TsdfEngine& e, const torch::Tensor& c2w, float fx, float fy, float cx, float cy) {
            // ITMBasicEngine::runRaycast(pose, intrinsics) with an explicit ITMIntrinsics (slam_pipeline.cpp:367-376)
            ORUtils::SE3Pose pose;
            auto c = c2w.to(torch::kCPU, torch::kFloat32).contiguous();
            pose.SetInvM(c.data_ptr<float>());
            ITMLib::ITMIntrinsics in;
            in.SetFrom(e.state().width, e.state().height, fx, fy, cx, cy);
            e.runRaycast(&pose, &in);
        })
        .def("camIntrincs", [](TsdfEngine& e) {
            auto t = torch::empty({(int64_t)e.camIntrincs.size(), 4}, torch::kFloat32);
            for (size_t i = 0; i < e.camIntrincs.size(); i++) {
                const auto& p = e.camIntrincs[i].projectionParamsSimple;
                t[i][0] = p.fx; t[i][1] = p.fy; t[i][2] = p.px; t[i][3] = p.py;
            }
            return t;
        })
        .def_readonly("framesProcessed", &TsdfEngine::framesProcessed);

    // ---- the reference's construction path: DatasetReader -> createTsdfEngine -> CLIEngine (slam_trainer.cpp:20-33)
    py::class_<DatasetReader>(m, "DatasetReader")
        .def(py::init([](int w, int h, float fx, float fy, float cx, float cy) {
            auto* r = new DatasetReader();
            r->width = w; r->height = h; r->fx = fx; r->fy = fy; r->cx = cx; r->cy = cy;
            return r;
        }))
        .def("addTrainCamera", [](DatasetReader& r, const Camera& c) { r.train_vec.push_back(c); })
        .def("size", [](DatasetReader& r) { return r.train_vec.size(); });
    py::class_<InfiniTAM::Engine::CLIEngine, std::unique_ptr<InfiniTAM::Engine::CLIEngine, py::nodelete>>(m, "CLIEngine")
        .def("ProcessFrame", &InfiniTAM::Engine::CLIEngine::ProcessFrame)
        .def("Shutdown", &InfiniTAM::Engine::CLIEngine::Shutdown)
        .def("GetDepthSize", [](InfiniTAM::Engine::CLIEngine& e) { auto d = e.GetDepthSize(); return std::make_pair(d.x, d.y); })
        .def("GetRGBSize", [](InfiniTAM::Engine::CLIEngine& e) { auto d = e.GetRGBSize(); return std::make_pair(d.x, d.y); })
        .def("getMainEngine", [](InfiniTAM::Engine::CLIEngine& e) {
            return static_cast<TsdfEngine*>(dynamic_cast<ITMLib::ITMBasicEngine<ITMVoxel, ITMVoxelIndex>*>(e.getMainEngine()));
        }, py::return_value_policy::reference)
        .def_readwrite("prefetch", &InfiniTAM::Engine::CLIEngine::prefetch)
        .def_readonly("uploadedBytes", &InfiniTAM::Engine::CLIEngine::uploadedBytes)
        .def_readonly("currentFrameNo", &InfiniTAM::Engine::CLIEngine::currentFrameNo);
    m.def("createTsdfEngine", [](const DatasetReader& r, const py::dict& cfg) { return createTsdfEngine(r, config_from_dict(cfg)); },
          py::return_value_policy::reference);

    // ---- pipeline
    py::class_<SLAMPipeline::PipelineTimes>(m, "PipelineTimes")
        .def_readonly("frames", &SLAMPipeline::PipelineTimes::frames)
        .def_readonly("slam_total", &SLAMPipeline::PipelineTimes::slam_total)
        .def_readonly("per_frame", &SLAMPipeline::PipelineTimes::per_frame)
        .def_readonly("keyframe_step", &SLAMPipeline::PipelineTimes::keyframe_step)
        .def_readonly("localFrameRaycast", &SLAMPipeline::PipelineTimes::localFrameRaycast)
        .def_readonly("keyFrameRaycast", &SLAMPipeline::PipelineTimes::keyFrameRaycast)
        .def_readonly("initNewGaussians", &SLAMPipeline::PipelineTimes::initNewGaussians)
        .def_readonly("localOptimize", &SLAMPipeline::PipelineTimes::localOptimize)
        .def_readonly("removeGaussian", &SLAMPipeline::PipelineTimes::removeGaussian)
        .def_readonly("checkError", &SLAMPipeline::PipelineTimes::checkError)
        .def_readonly("max_frame_after_30", &SLAMPipeline::PipelineTimes::max_frame_after_30)
        .def_readonly("max_frame_id", &SLAMPipeline::PipelineTimes::max_frame_id)
        .def_readonly("gpu_memory_mb", &SLAMPipeline::PipelineTimes::gpu_memory_mb)
        .def("fps", &SLAMPipeline::PipelineTimes::fps)
        .def("fusion_fps", &SLAMPipeline::PipelineTimes::fusion_fps)
        .def("gaussian_fps", &SLAMPipeline::PipelineTimes::gaussian_fps);
    py::class_<SLAMPipeline>(m, "SLAMPipeline")
        .def(py::init<TsdfEngine*, SLAMGaussianModel*, uint64_t, bool>(), py::arg("engine"), py::arg("model"),
             py::arg("seed") = 1234, py::arg("use_gt_pose") = true, py::keep_alive<1, 2>(), py::keep_alive<1, 3>())
        .def(py::init<uint64_t>(), py::arg("seed") = 1234)
        .def("setTsdfEngine", &SLAMPipeline::setTsdfEngine)
        .def("setModel", [](SLAMPipeline& p, SLAMGaussianModel* m) { p.model = m; p.device = m->device; }, py::keep_alive<1, 2>())
        .def("SLAMTrainCamsModel", [](SLAMPipeline& p, SLAMGaussianModel& m, std::vector<Camera>& cams) {
            p.SLAMTrainCams(m, cams);
            return cams;  // with c2w_slam filled in
        }, py::call_guard<py::gil_scoped_release>())
        // the reference's whole-run clock (LOG_PIPELINE_TIME): SLAMTrainCams over `cams` from frame 0, -> the [PIPELINE AVG TIME] numbers
        .def("SLAMTrainCamsTimed", [](SLAMPipeline& p, SLAMGaussianModel& m, std::vector<Camera>& cams) {
            { py::gil_scoped_release nogil; p.SLAMTrainCams(m, cams); }
            return p.times;
        })
        .def_readonly("times", &SLAMPipeline::times)
        .def_readwrite("log_pipeline_time", &SLAMPipeline::log_pipeline_time)
        .def_readwrite("frame_report_ms", &SLAMPipeline::frame_report_ms)
        .def_readwrite("keep_frame_ms", &SLAMPipeline::keep_frame_ms)
        .def_readonly("frame_ms", &SLAMPipeline::frame_ms)
        .def_readonly("frame_wait_ms", &SLAMPipeline::frame_wait_ms)
        .def("processFrameCLI", [](SLAMPipeline& p, int i, Camera& cam) { p.processFrame(i, cam); },
             py::call_guard<py::gil_scoped_release>())
        .def("loadConfig", [](SLAMPipeline& p, const py::dict& d) { p.loadConfig(config_from_dict(d)); })
        // the optimise loop may run the autograd engine (losses beyond L1): it must not hold the GIL
        .def("processFrame", py::overload_cast<int, Camera&, const torch::Tensor&, const torch::Tensor&>(&SLAMPipeline::processFrame),
             py::call_guard<py::gil_scoped_release>())
        .def("SLAMTrainCams", py::overload_cast<std::vector<Camera>&, const std::vector<torch::Tensor>&,
                                                const std::vector<torch::Tensor>&>(&SLAMPipeline::SLAMTrainCams),
             py::call_guard<py::gil_scoped_release>())
        .def("runRaycastByCam", &SLAMPipeline::runRaycastByCam, py::arg("cam"), py::arg("use_cam_depth") = true)
        .def_readwrite("overlap_mapping", &SLAMPipeline::overlap_mapping)
        .def_readwrite("mapping_thread", &SLAMPipeline::mapping_thread)
        .def_readwrite("async_raycasts", &SLAMPipeline::async_raycasts)
        .def_readwrite("frame_stream_kind", &SLAMPipeline::frame_stream_kind)
        .def_readwrite("map_stream_kind", &SLAMPipeline::map_stream_kind)
        .def_readwrite("raycast_stream_kind", &SLAMPipeline::raycast_stream_kind)
        .def_readwrite("prefetch_next_preprocess", &SLAMPipeline::prefetch_next_preprocess)
        .def_readwrite("frame_chain_reserve", &SLAMPipeline::frame_chain_reserve)
        .def_readwrite("pipeline_raycasts", &SLAMPipeline::pipeline_raycasts)
        .def_readwrite("merge_keyframe_raycasts", &SLAMPipeline::merge_keyframe_raycasts)
        .def_readwrite("pump_iters_first", &SLAMPipeline::pump_iters_first)
        .def_readwrite("pump_iters_per_frame", &SLAMPipeline::pump_iters_per_frame)
        .def("flush", &SLAMPipeline::flush, py::call_guard<py::gil_scoped_release>())
        .def_readwrite("trace_frames", &SLAMPipeline::trace_frames)
        .def("frameTrace", [](SLAMPipeline& p) {
            { py::gil_scoped_release nogil; p.flush(); }
            return std::make_tuple(p.trace_live, p.trace_counters, p.trace_poses);
        })
        .def("removeRedundantGs", &SLAMPipeline::removeRedundantGs)
        .def("checkKeyFrameError", [](SLAMPipeline& p) { { py::gil_scoped_release nogil; p.flush(); } p.checkKeyFrameError(); })
        .def("keyframeLossDict", [](SLAMPipeline& p) { { py::gil_scoped_release nogil; p.flush(); } return p.keyframe_loss_dict; })
        .def("appendOptView", [](SLAMPipeline& p, const Camera& cam, TensorDict rc) {   // tests: a history view by hand
            p.opt_cam_list.push_back(cam); p.opt_raycast_list.push_back(std::move(rc));
        })
        .def_readwrite("sample_method", &SLAMPipeline::sample_method)
        .def_readwrite("loss_thres", &SLAMPipeline::loss_thres)
        .def_readwrite("large_scale_thres", &SLAMPipeline::large_scale_thres)
        .def_readwrite("small_scale_thres", &SLAMPipeline::small_scale_thres)
        .def_readwrite("low_opac_thres", &SLAMPipeline::low_opac_thres)
        .def("stats", [](SLAMPipeline& p) {
            { py::gil_scoped_release nogil; p.flush(); }
            py::dict d;
            d["frames"] = p.stats.frames.load(); d["opt_iters"] = p.stats.opt_iters.load(); d["raycasts"] = p.stats.raycasts.load();
            d["added"] = p.stats.added.load(); d["pruned"] = p.stats.pruned.load();
            return d;
        })
        .def("keyframeCount", [](SLAMPipeline& p) { { py::gil_scoped_release nogil; p.flush(); } return (int)p.keyframe_cam_list.size(); })
        .def("optCams", [](SLAMPipeline& p) { { py::gil_scoped_release nogil; p.flush(); } return p.opt_cam_list; })
        .def("optRaycasts", [](SLAMPipeline& p) { { py::gil_scoped_release nogil; p.flush(); } return p.opt_raycast_list; })
        .def_readwrite("workspace_dir", &SLAMPipeline::workspace_dir)
        .def_readwrite("saved_mesh", &SLAMPipeline::saved_mesh)
        .def_readwrite("saved_engine", &SLAMPipeline::saved_engine)
        .def("renderEvalImgs", &SLAMPipeline::renderEvalImgs, py::arg("cams"), py::arg("names") = std::vector<std::string>{"rgb"},
             py::call_guard<py::gil_scoped_release>())
        .def("saveMesh", &SLAMPipeline::saveMesh)
        .def("saveEngine", &SLAMPipeline::saveEngine)
        .def("loadEngine", &SLAMPipeline::loadEngine)
        .def_readwrite("work_mode", &SLAMPipeline::work_mode);
}
