"""Host-side mirror of SLAMPipeline (slam/slam_pipeline.{h,cpp}): the per-frame loop that drives the hot path.

    SLAMTrainCams      :52-173   per frame: TSDF ProcessFrame; every local_opt_interval frames:
    localFrameRaycast  :417-448  runRaycastByCam for the <= 2 window frames
    keyFrameRaycast    :528-561  + <= 7 random keyframes
    initNewGaussians   :450-526  error-mask sampling -> addGaussians
    localOptimize      :195-289  20 x (forward, L1, backward, 7 x Adam)
    removeRedundantGs  :564-586  prune by scale / opacity

All compute goes through the C-ABI (gs_model.py, tsdf_engine.py); this file is bookkeeping.  Random choices the
reference seeds from std::random_device (dataset_reader.h:39) are seeded here so runs are reproducible.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from ._lib import check, lib
from .gs_model import Camera, SLAMGaussianModel, pose_inv  # noqa: F401

DEFAULT_PIPE = dict(new_gs_sample_ratio=0.25, color_error_thres=0.05, localframe_cam_window_length=2,
                    localframe_cam_window_interval=5, local_opt_iters=20, local_opt_interval=10,
                    keyframe_theta_thres=30.0, keyframe_trans_thres=0.3, keyframe_select_max=7,
                    depth_vis_max=5.0, depth_vis_min=0.0, alpha_vis_max=5.0, large_scale_thres=0.1,
                    small_scale_thres=0.003, low_opac_thres=0.005, scene_scale=1.1 * 3.0)


class RandomSelector:
    """dataset_reader.h:26-100 (uniform branch): draw without replacement, refill when exhausted."""

    def __init__(self, items, rng):
        self.original = list(enumerate(items))
        self.current = list(self.original)
        self.rng = rng

    def getNext(self):
        if not self.current:
            self.current = list(self.original)
        i = self.rng.randrange(len(self.current))
        v = self.current[i]
        self.current[i] = self.current[-1]
        self.current.pop()
        return v  # (originalIndex, value)


def compute_normal_map(vertex_map):
    """computeNormalMap (tensor_math.cpp:278-300) -> gps_normal_map"""
    H, W, _ = vertex_map.shape
    vertex_map = vertex_map.contiguous()
    out = torch.empty_like(vertex_map)
    check(lib.gps_normal_map(W, H, vertex_map.data_ptr(), out.data_ptr(),
                             C.c_void_p(torch.cuda.current_stream(vertex_map.device).cuda_stream)), "gps_normal_map")
    return out


class SLAMPipeline:
    def __init__(self, tsdf_engine, model, pipe_cfg=None, seed=1234, work_mode="train", use_gt_pose=True):
        self.tsdf = tsdf_engine
        self.model = model
        self.cfg = dict(DEFAULT_PIPE)
        self.cfg.update(pipe_cfg or {})
        self.work_mode = work_mode
        # TSDF.use_gt_pose of the configs (true in every shipped one; createTsdfEngine then calls turnOffTracking,
        # InfiniTAM_tools.cpp:59-62); False keeps the depth-only ExtendedTracker active
        self.use_gt_pose = use_gt_pose
        if not use_gt_pose and not hasattr(tsdf_engine, "track_cfg"):
            tsdf_engine.turnOnTracking()
        self.rng = random.Random(seed)
        self.gen = torch.Generator().manual_seed(seed)  # host generator (see SLAMGaussianModel.addGaussians)
        self.device = model.device
        self.localframe_cam_window = []
        self.localframe_raycast_window = []
        self.keyframe_cam_list = []
        self.opt_cam_list, self.opt_raycast_list = [], []
        self.curr_frame_id = 0
        self.curr_cam = None
        self.stats = dict(frames=0, opt_iters=0, raycasts=0, added=0, pruned=0)
        self.workspace_dir, self.saved_mesh, self.saved_engine = ".", "", ""

    # ------------------------------------------------------------------ slam_pipeline.h:32-49
    def saveMesh(self):
        if self.saved_mesh:
            self.tsdf.SaveSceneToMesh(self.workspace_dir + "/" + self.saved_mesh)

    def saveEngine(self):
        if self.saved_engine:
            self.tsdf.SaveToFile(self.workspace_dir + "/" + self.saved_engine)

    def loadEngine(self):
        self.tsdf.LoadFromFile(self.workspace_dir + "/" + self.saved_engine)

    # ------------------------------------------------------------------ raycast -> tensors (runRaycastByCam :362-415)
    def runRaycastByCam(self, cam):
        eng = self.tsdf
        if 0 <= cam.id < len(eng.camPoses):
            pose = eng.camPoses[cam.id]
            eng.runRaycast(pose=pose)
        else:
            eng.runRaycast(c2w=cam.c2w.numpy())
        H, W = cam.height, cam.width
        d = self.device
        color = torch.empty((H, W, 3), device=d)
        vertex = torch.empty((H, W, 3), device=d)
        conf = torch.empty((H, W, 1), device=d)
        depth = torch.empty((H, W, 1), device=d)
        depth_c = torch.empty((H, W, 1), device=d)
        w2c = np.ascontiguousarray(pose_inv(cam.c2w).numpy().astype(np.float32))  # poseInv(cam.c2w): dataset pose (:398)
        check(lib.gps_raycast_to_maps(W, H, eng.fv_raycast.data_ptr(), eng.fv_colour.data_ptr(), eng.getVoxelSize(),
                                      w2c.ctypes.data, color.data_ptr(), vertex.data_ptr(), conf.data_ptr(),
                                      depth.data_ptr(), depth_c.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream(d).cuda_stream)),
              "gps_raycast_to_maps")
        self.stats["raycasts"] += 1
        return dict(color_map=color, vertex_map=vertex, confidence_map=conf, depth_map=depth,
                    depth_map_clamped=depth_c)

    # ------------------------------------------------------------------ frame bookkeeping (updateFrameList :319-360)
    def updateFrameList(self):
        c = self.cfg
        if self.curr_frame_id == 0:
            return
        if self.curr_frame_id % c["localframe_cam_window_interval"] == 0:
            self.localframe_cam_window.append(self.curr_cam)
            if len(self.localframe_cam_window) == c["localframe_cam_window_length"] + 1:
                self.localframe_cam_window.pop(0)
        is_key = False
        if not self.keyframe_cam_list:
            is_key = True
        else:
            last = self.keyframe_cam_list[-1]
            Rp, Rc = last.c2w_slam[:3, :3], self.curr_cam.c2w_slam[:3, :3]
            cos_t = float((torch.trace(Rp.t() @ Rc) - 1) / 2)
            theta = math.degrees(math.acos(max(-1.0, min(1.0, cos_t))))
            trans = float(torch.norm(last.c2w_slam[:3, 3] - self.curr_cam.c2w_slam[:3, 3]))
            is_key = theta > c["keyframe_theta_thres"] or trans > c["keyframe_trans_thres"]
        if is_key:
            self.keyframe_cam_list.append(self.curr_cam)

    def localFrameRaycast(self):
        self.localframe_raycast_window = [self.runRaycastByCam(cam) for cam in self.localframe_cam_window]

    def keyFrameRaycast(self):
        self.opt_cam_list = list(self.localframe_cam_window)
        self.opt_raycast_list = list(self.localframe_raycast_window)
        n = min(self.cfg["keyframe_select_max"], len(self.keyframe_cam_list))
        sel = RandomSelector(self.keyframe_cam_list, self.rng)
        for _ in range(n):
            _, cam = sel.getNext()
            self.opt_cam_list.append(cam)
            self.opt_raycast_list.append(self.runRaycastByCam(cam))

    # ------------------------------------------------------------------ initNewGaussians :450-526
    def initNewGaussians(self, raycast_maps):
        c, model, cam = self.cfg, self.model, self.curr_cam
        depth, color, vertex = raycast_maps["depth_map"], raycast_maps["color_map"], raycast_maps["vertex_map"]
        frame_num = c["local_opt_interval"]
        valid = (depth > c["depth_vis_min"]) & (depth < c["depth_vis_max"])
        valid = valid & ~((vertex.sum(2) == 0).unsqueeze(-1))
        if model.getGaussianNum() == 0:
            err = torch.mean(torch.abs(color - cam.image), -1, True)
            mask = (err > c["color_error_thres"]) & valid
            frame_num += 1
        else:
            res = model.forward(cam, depth, color)
            err = torch.mean(torch.abs(res["rgb"] - cam.image), -1, True)
            mask = (err > c["color_error_thres"]) & valid & (res["alpha"] < c["alpha_vis_max"])
        raycast_maps["normal_map"] = compute_normal_map(vertex)
        n = model.addGaussians(cam, raycast_maps, mask, c["new_gs_sample_ratio"], frame_num, generator=self.gen)
        self.stats["added"] += n

    # ------------------------------------------------------------------ localOptimize :195-289
    def localOptimize(self):
        model = self.model
        if model.getGaussianNum() == 0:
            return
        model.initOptimizers(-1, self.cfg["scene_scale"])
        loader = RandomSelector(self.opt_cam_list, self.rng)
        for _ in range(self.cfg["local_opt_iters"]):
            idx, cam = loader.getNext()
            rc = self.opt_raycast_list[idx]
            model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image,
                             ref_depth_clamped=rc["depth_map_clamped"])
            self.stats["opt_iters"] += 1

    # ------------------------------------------------------------------ removeRedundantGs :564-586
    def removeRedundantGs(self):
        model, c = self.model, self.cfg
        if model.getGaussianNum() == 0:
            return
        smax = model.getRealScales().max(-1).values
        mask = (smax < c["small_scale_thres"]) | (smax > c["large_scale_thres"]) | \
               (model.getRealOpacities().squeeze(-1) < c["low_opac_thres"])
        n = int(mask.sum())  # the reference syncs 5 times here for its printf; once is enough
        model.check_binning_capacity()  # the host is synchronised here anyway: read the kernels' sticky overflow flag
        if n > 0:
            model.prunePoints(mask)
            self.stats["pruned"] += n

    # ------------------------------------------------------------------ one SLAM frame (body of SLAMTrainCams :69-132)
    def process_frame(self, i, cam, rgb_u8_dev, depth_mm_dev):
        self.curr_frame_id = i
        if self.use_gt_pose:
            M, invM = self.tsdf.ProcessFrame(rgb_u8_dev, depth_mm_dev, cam.c2w.numpy())
        else:
            M, invM = self.tsdf.ProcessFrameTracked(rgb_u8_dev, depth_mm_dev)
        # est_pose = pose_d->GetInvM() (:81-82): ORUtils layout -> row-major tensor
        cam.c2w_slam = torch.from_numpy(invM.reshape(4, 4).T.copy())
        cam.invalidate()
        self.curr_cam = cam
        cam.toGPU()
        self.updateFrameList()
        self.stats["frames"] += 1
        if self.work_mode == "recon":
            return
        if i % self.cfg["local_opt_interval"] == 0 and i > 0:
            self.localFrameRaycast()
            self.keyFrameRaycast()
            self.initNewGaussians(self.localframe_raycast_window[-1])
            self.localOptimize()
            self.removeRedundantGs()
