"""Host-side mirror of the reference's TSDF engine surface on top of the C-ABI:
ITMBasicEngine<ITMVoxel_s_rgb, ITMVoxelBlockHash> (InfiniTAM/ITMLib/Core/ITMBasicEngine.{h,tpp}) as
configured by createTsdfEngine (slam/InfiniTAM_tools.cpp:3-67) and driven by CLIEngine::ProcessFrame
(slam/TsdfFusion/CLIEngine.cpp:34-58) and SLAMPipeline::runRaycastByCam (slam/slam_pipeline.cpp:362-415).

Owns the device buffers (torch tensors; the reference owns them through ORUtils::MemoryBlock), fills the
gps_tsdf_state struct once and afterwards only passes pointers.  All compute is in libgpsslam_hip.so.
"""
import ctypes as C

import numpy as np
import weakref

import torch

from ._lib import TrackConfig, TrackState, TsdfState, check, lib

# reference capacities (ITMLib/Objects/Scene/ITMVoxelBlockHash.h:18-22)
SDF_LOCAL_BLOCK_NUM = 0x40000
SDF_BUCKET_NUM = 0x100000
SDF_EXCESS_LIST_SIZE = 0x20000

VOXEL_DT = np.dtype([("sdf", "<i2"), ("w_depth", "u1"), ("clr", "u1", (3,)), ("w_color", "u1"), ("pad", "u1")])
HASH_DT = np.dtype([("pos", "<i2", (3,)), ("pad", "<i2"), ("offset", "<i4"), ("ptr", "<i4")])


def pose_from_c2w(c2w):
    """ORUtils::SE3Pose: SetInvM(c2w); Coerce(); -> (M, invM) as float32[16] in ORUtils layout (host only)."""
    c = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(4, 4))
    M = np.zeros(16, np.float32)
    invM = np.zeros(16, np.float32)
    check(lib.gps_pose_from_c2w(c.ctypes.data, M.ctypes.data, invM.ctypes.data), "gps_pose_from_c2w")
    return M, invM


def ray_stats_of(counters):
    """{steps, reads, rays} from a counter block (int32[16] as numpy / tensor): castRay steps as the reference's loop counts
    them, voxel reads of the kernel's own loop, rays cast -- of one raycast launch (S-bar = steps / rays)"""
    c = np.ascontiguousarray(np.asarray(counters, dtype=np.int32)[10:16]).view(np.uint64)
    return {"steps": int(c[0]), "reads": int(c[1]), "rays": int(c[2])}


class TsdfEngine:
    """ITMBasicEngine with tracking switched off (use_gt_pose: true in every shipped config)."""

    def __init__(self, width, height, fx, fy, cx, cy, voxel_size=0.005, mu=0.02, view_frustum_min=0.2,
                 view_frustum_max=10.0, n_blocks=SDF_LOCAL_BLOCK_NUM, n_buckets=SDF_BUCKET_NUM,
                 n_excess=SDF_EXCESS_LIST_SIZE, device="cuda:0"):
        self.W, self.H = int(width), int(height)
        self.device = torch.device(device)
        self.voxel_size = float(voxel_size)
        d = self.device
        P = self.W * self.H
        n_total = n_buckets + n_excess
        nblk = (n_total + 1023) // 1024
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=d)
        self.vba = z(n_blocks * 512 * 8, torch.uint8)
        self.vba_alloc_list = z(n_blocks, torch.int32)
        self.hash = z(n_total * 16, torch.uint8)
        self.excess_list = z(n_excess, torch.int32)
        self.counters = z(16, torch.int32)
        self.alloc_prio = z(n_total, torch.int32)
        self.scan_scratch = z((int(lib.gps_tsdf_scratch_bytes(self.W, self.H, n_buckets, n_excess)) + 3) // 4, torch.int32)
        self.visible_type = z(n_total, torch.uint8)
        self.visible_ids = z(n_blocks, torch.int32)
        self.depth = z(P, torch.float32)
        self.rgb = z(P * 4, torch.uint8)
        self.minmax = z(P * 2, torch.float32)
        self.raycast = z(P * 4, torch.float32)
        self.icp_points = z(P * 4, torch.float32)
        self.icp_normals = z(P * 4, torch.float32)
        self.fv_visible_ids = z(n_blocks, torch.int32)
        self.fv_minmax = z(P * 2, torch.float32)
        self.fv_raycast = z(P * 4, torch.float32)
        self.fv_colour = z(P * 4, torch.uint8)
        self.depth_mm = z(P, torch.int16)
        s = TsdfState()
        s.width, s.height = self.W, self.H
        s.fx, s.fy, s.cx, s.cy = float(fx), float(fy), float(cx), float(cy)
        s.voxel_size, s.mu = float(voxel_size), float(mu)
        s.view_frustum_min, s.view_frustum_max = float(view_frustum_min), float(view_frustum_max)
        s.max_w = 100  # ITMLibSettings.cpp:10
        s.n_blocks, s.n_buckets, s.n_excess = n_blocks, n_buckets, n_excess
        for name in ("vba", "vba_alloc_list", "hash", "excess_list", "counters", "alloc_prio", "scan_scratch",
                     "visible_type", "visible_ids", "depth", "rgb", "minmax", "raycast", "icp_points", "icp_normals",
                     "fv_visible_ids", "fv_minmax", "fv_raycast", "fv_colour"):
            setattr(s, name, getattr(self, name).data_ptr())
        self.state = s
        self.n_blocks, self.n_buckets, self.n_excess, self.n_total = n_blocks, n_buckets, n_excess, n_total
        self.frames_processed = 0
        # ITMBasicEngine::camPoses / gtC2wPoses (ITMBasicEngine.h:54-56)
        self.camPoses = []
        self.reset()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        check(lib.gps_tsdf_reset(C.byref(self.state), self._stream()), "gps_tsdf_reset")
        self.frames_processed = 0
        self.camPoses = []

    # ---- ITMBasicEngine::ProcessFrame (tracking off: pose := gtC2wPoses[framesProcessed]; Coerce())
    def ProcessFrame(self, rgb_u8, depth_mm_i16, gt_c2w):
        """rgb_u8: uint8 [H,W,4] device tensor (uchar4, as ITMUChar4Image; [H,W,3] is padded with a copy);
        depth_mm_i16: int16 [H,W] millimetres; gt_c2w 4x4.  The kernels read both tensors in place (the reference
        uploads the pre-converted ITM images every frame, ITMViewBuilder_CUDA.cu:61-62)."""
        if rgb_u8.shape[-1] == 3:
            rgb_u8 = torch.cat([rgb_u8, torch.full_like(rgb_u8[..., :1], 255)], -1)
        assert rgb_u8.is_contiguous() and depth_mm_i16.is_contiguous()
        self._frame_inputs = (rgb_u8, depth_mm_i16)  # keep alive while kernels may read them
        self.state.rgb = rgb_u8.data_ptr()
        M, invM = pose_from_c2w(gt_c2w)
        check(lib.gps_tsdf_process_frame(C.byref(self.state), depth_mm_i16.data_ptr(), M.ctypes.data,
                                         invM.ctypes.data, self._stream()), "gps_tsdf_process_frame")
        self.camPoses.append((M, invM))
        self.frames_processed += 1
        return M, invM

    # ---- ITMBasicEngine::ProcessFrame with the tracker ON (use_gt_pose: false)
    def turnOnTracking(self, levels="rrbb", num_iter_coarse=20, num_iter_fine=50, thresh_coarse=0.1, thresh_fine=0.004,
                       term_thresh=1e-4, tukey_cutoff=8.0, frames_to_skip=20, frames_to_weight=50, bar_arg_line=True,
                       poses_riding_along=1, host_summed_rows=True):
        """Depth-only ExtendedTracker with the parameters of ITMLibSettings.cpp:54-57 (defaults).
        poses_riding_along: how many of the poses the LM loop would evaluate next after a REJECTION are evaluated together with
        every evaluation (gps_track_state.mailbox_bytes; BAR argument line only; 0 = one pose per evaluation; 0..3).  Same poses
        whatever the number; 1 measured best on the 640x480 loop (0 / 1 / 2: 973 / 993 / 980 frames/s sequential).
        host_summed_rows: the evaluation's workgroups store their rows of partial sums into the pinned mailbox and the tracking call
        adds them (same order, same bits) instead of a summing workgroup on the device."""
        self.track_cfg = TrackConfig()
        check(lib.gps_track_config_init(C.byref(self.track_cfg), levels.encode(), num_iter_coarse, num_iter_fine, thresh_coarse,
                                        thresh_fine, term_thresh, tukey_cutoff, frames_to_skip, frames_to_weight),
              "gps_track_config_init")
        self.track_state = TrackState()
        check(lib.gps_track_state_reset(C.byref(self.track_state)), "gps_track_state_reset")
        nbytes = int(lib.gps_track_scratch_bytes(self.W, self.H))
        self.track_scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        groups = 1 + max(0, min(3, int(poses_riding_along)))
        # kernel -> host, no memcpy: an answer block per group and (host_summed_rows) a table of the workgroups' rows per group, which
        # the tracking call adds up itself instead of a summing workgroup on the device (gps_track_state.mailbox_bytes)
        per_group = 256 + (32768 if host_summed_rows else 0)
        # (empty + a numpy fill, not torch.zeros: the fill would be a parallel region of torch's intra-op pool -- dist_util.cap_host_threads)
        self._mailbox = torch.empty(per_group // 4 * groups, dtype=torch.float32, pin_memory=True)
        self._mailbox.numpy().fill(0.0)
        self.track_state.host_mailbox = self._mailbox.data_ptr()
        self.track_state.mailbox_bytes = per_group * groups
        # the argument line in host-writable device memory (written through the BAR; None without a large BAR)
        if not hasattr(self, "_arg_line"):
            self._arg_line = None
        if bar_arg_line and self._arg_line is None:
            line = C.c_void_p()
            check(lib.gps_track_arg_line_alloc(C.byref(line)), "gps_track_arg_line_alloc")
            self._arg_line = line if line.value else None
            if self._arg_line is not None:
                weakref.finalize(self, lib.gps_track_arg_line_free, self._arg_line)
        self.track_state.dev_arg_line = self._arg_line if bar_arg_line else None

    def ProcessFrameTracked(self, rgb_u8, depth_mm_i16):
        """-> (M, invM) estimated by the tracker (ORUtils layout, numpy float32[16]).  Host-synchronous (see
        gps_tsdf_track_camera)."""
        if rgb_u8.shape[-1] == 3:
            rgb_u8 = torch.cat([rgb_u8, torch.full_like(rgb_u8[..., :1], 255)], -1)
        assert rgb_u8.is_contiguous() and depth_mm_i16.is_contiguous()
        self._frame_inputs = (rgb_u8, depth_mm_i16)
        self.state.rgb = rgb_u8.data_ptr()
        check(lib.gps_tsdf_process_frame_tracked(C.byref(self.state), depth_mm_i16.data_ptr(), C.byref(self.track_cfg),
                                                 C.byref(self.track_state), self.track_scratch.data_ptr(),
                                                 self.track_scratch.numel(), self._stream()),
              "gps_tsdf_process_frame_tracked")
        M = np.array(self.track_state.pose_M, dtype=np.float32)
        invM = np.array(self.track_state.pose_invM, dtype=np.float32)
        self.camPoses.append((M, invM))
        self.frames_processed += 1
        return M, invM

    def track_poll_profile(self):
        """gps_track_poll_profile of this engine's tracker scratch -> [spin ticks, evaluation ticks, evaluations, retired launches]
        (cumulative, 100 MHz ticks; blocking read-back: measurement only)"""
        out = (C.c_uint32 * 4)()
        check(lib.gps_track_poll_profile(self.track_scratch.data_ptr(), self.W, self.H, out, self._stream()), "gps_track_poll_profile")
        return [int(v) for v in out]

    def track_poll_phases(self):
        """gps_track_poll_phases -> {"level0": [evaluations, loop ticks, rows-wait ticks, tail ticks], "coarse": [...]} (cumulative)"""
        out = (C.c_uint32 * 8)()
        check(lib.gps_track_poll_phases(self.track_scratch.data_ptr(), self.W, self.H, out, self._stream()), "gps_track_poll_phases")
        v = [int(x) for x in out]
        return {"level0": v[:4], "coarse": v[4:]}

    def track_diag(self):
        return np.array(self.track_state.diag, dtype=np.float32)

    # ---- ITMBasicEngine::runRaycast(pose, intrinsics) + GetFreeImage / GetFreeVertex
    def runRaycast(self, c2w=None, pose=None):
        M, invM = pose if pose is not None else pose_from_c2w(c2w)
        check(lib.gps_tsdf_free_raycast(C.byref(self.state), M.ctypes.data, invM.ctypes.data, self._stream()),
              "gps_tsdf_free_raycast")
        return M, invM

    # ---- the same for several poses in one chain of launches (gps_tsdf_free_raycast_batch); view k's images:
    #      GetFreeVertex(k) / GetFreeImage(k)
    def runRaycastBatch(self, poses):
        from ._lib import TsdfView
        n = len(poses)
        MAX_VIEWS = 12  # views per gps_tsdf_free_raycast_batch call (gps_tsdf_view_table_bytes rejects more)
        if n > MAX_VIEWS:
            raise ValueError("runRaycastBatch: at most %d views per call (got %d); call it per chunk as the C++ host does" % (MAX_VIEWS, n))
        views = getattr(self, "_views", [])
        P = self.W * self.H
        z = lambda cnt, dt: torch.zeros(cnt, dtype=dt, device=self.device)
        while len(views) < n:
            v = dict(visible_ids=z(self.n_blocks, torch.int32), minmax=z(P * 2, torch.float32), raycast=z(P * 4, torch.float32),
                     colour=z(P * 4, torch.uint8), scratch=torch.zeros_like(self.scan_scratch), counters=z(16, torch.int32))
            rec = TsdfView()
            for k, t in v.items():
                setattr(rec, k, t.data_ptr())
            check(lib.gps_tsdf_view_init(C.byref(self.state), C.byref(rec), self._stream()), "gps_tsdf_view_init")
            views.append(v)
        self._views = views
        arr = (TsdfView * n)()
        for k, (M, invM) in enumerate(poses):
            for name, t in views[k].items():
                setattr(arr[k], name, t.data_ptr())
            arr[k].M[:] = M.reshape(-1).tolist(); arr[k].invM[:] = invM.reshape(-1).tolist()
            arr[k].fx, arr[k].fy, arr[k].cx, arr[k].cy = self.state.fx, self.state.fy, self.state.cx, self.state.cy
        if getattr(self, "_view_table", None) is None:
            nbytes = int(lib.gps_tsdf_view_table_bytes(MAX_VIEWS))
            assert nbytes > 0, nbytes
            self._view_table = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        check(lib.gps_tsdf_free_raycast_batch(C.byref(self.state), n, arr, self._view_table.data_ptr(), self._stream()),
              "gps_tsdf_free_raycast_batch")

    def GetFreeImage(self, view=None):
        if view is not None:
            return self._views[view]["colour"].view(self.H, self.W, 4)
        return self.fv_colour.view(self.H, self.W, 4)

    def GetFreeVertex(self, view=None):
        if view is not None:
            return self._views[view]["raycast"].view(self.H, self.W, 4)
        return self.fv_raycast.view(self.H, self.W, 4)

    def GetLiveVertex(self):
        return self.raycast.view(self.H, self.W, 4)

    def getVoxelSize(self):
        return self.voxel_size

    # ---- meshing (ITMBasicEngine::SaveSceneToMesh, Core/ITMBasicEngine.tpp:105-117)
    def MeshScene(self, max_triangles=1 << 24):
        """ITMMeshingEngine::MeshScene -> (triangles float32 [max_triangles, 7, 3] device tensor, counts int64[2] device):
        counts[0] = noTotalTriangles.  Rows: p0 p1 p2 c0 c1 c2 clr (ITMMesh::Triangle).  No host sync."""
        tri = torch.empty((max_triangles, 7, 3), dtype=torch.float32, device=self.device)
        counts = torch.zeros(2, dtype=torch.int64, device=self.device)
        nbytes = int(lib.gps_tsdf_mesh_workspace_bytes(C.byref(self.state)))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        check(lib.gps_tsdf_mesh_scene(C.byref(self.state), max_triangles, tri.data_ptr(), counts.data_ptr(), ws.data_ptr(), nbytes,
                                      self._stream()), "gps_tsdf_mesh_scene")
        return tri, counts

    def SaveSceneToMesh(self, file_name, max_triangles=1 << 24):
        """MeshScene + ITMMesh::WritePLY (Objects/Meshing/ITMMesh.h:39-106): ascii PLY, 3 vertices per triangle with uchar
        colours (static_cast<unsigned char>(c * 255)), then the faces."""
        tri, counts = self.MeshScene(max_triangles)
        n = int(counts[0].item())
        from . import tsdf_io
        tsdf_io.write_mesh_ply(file_name, tri[:n].cpu().numpy())
        return n

    # ---- persistence (ITMBasicEngine::SaveToFile / LoadFromFile, Core/ITMBasicEngine.tpp:119-171): formats in tsdf_io.py
    def SaveToFile(self, save_output_directory):
        import os
        from . import tsdf_io
        d = save_output_directory if save_output_directory.endswith("/") else save_output_directory + "/"
        os.makedirs(d + "Relocaliser/", exist_ok=True)
        c = self.counters_host()
        tsdf_io.save_scene(d + "Scene/", self.vba.cpu().numpy(), self.vba_alloc_list.cpu().numpy(), c[0],
                           self.hash.cpu().numpy(), self.excess_list.cpu().numpy(), c[1])

    def LoadFromFile(self, save_input_directory):
        from . import tsdf_io
        d = save_input_directory if save_input_directory.endswith("/") else save_input_directory + "/"
        self.reset()  # resetAll() (ITMBasicEngine.tpp:146)
        sc = tsdf_io.load_scene(d + "Scene/", self.n_blocks, self.n_total, self.n_excess)
        self.vba.copy_(torch.from_numpy(sc["vba"]).to(self.device))
        self.vba_alloc_list.copy_(torch.from_numpy(sc["alloc_list"].copy()).to(self.device))
        self.hash.copy_(torch.from_numpy(sc["hash"]).to(self.device))
        self.excess_list.copy_(torch.from_numpy(sc["excess_list"].copy()).to(self.device))
        c = self.counters.cpu()
        c[0], c[1] = sc["last_free_block"], sc["last_free_excess"]
        self.counters.copy_(c)
        check(lib.gps_tsdf_rebuild_index(C.byref(self.state), self._stream()), "gps_tsdf_rebuild_index")

    # ---- host views for tests / persistence (sync)
    def counters_host(self):
        return self.counters.cpu().numpy()

    def ray_stats(self):
        """ray statistics of the LAST raycast launch on this scene (live, or a single free view): gps_tsdf_ray_stats sums the
        raycaster's per-wave rows into the counter block, see ray_stats_of (blocking read-back: measurement only)"""
        check(lib.gps_tsdf_ray_stats(C.byref(self.state), self._stream()), "gps_tsdf_ray_stats")
        return ray_stats_of(self.counters_host())

    def hash_host(self):
        return self.hash.cpu().numpy().view(HASH_DT)

    def vba_host(self, ptrs):
        idx = torch.as_tensor(np.asarray(ptrs, dtype=np.int64), device=self.device)
        return self.vba.view(self.n_blocks, 512 * 8)[idx].cpu().numpy().view(VOXEL_DT).reshape(-1, 512)

