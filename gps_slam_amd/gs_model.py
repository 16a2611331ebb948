"""Host-side mirror of the reference's Gaussian model on top of the C-ABI:
RawGaussianParams (include/raw_gs_param.h, src/raw_gs_param.cpp), RawGaussianModel
(include/raw_gs_model.h, src/raw_gs_model.cpp: gesForward :188-367, computeLoss :369-417, prunePoints :635-644,
initOptimizers/optimizersStep :654-705) and SLAMGaussianModel::addGaussians (slam/slam_gs_model.cpp:5-56).

Tensors live in torch; every per-Gaussian / per-pixel computation of the optimisation step is a C-ABI kernel
(preprocess -> binning -> ges raster -> compose+L1 -> raster bwd -> preprocess bwd -> fused Adam): 10 launch
sites per iteration instead of the reference's ~150 libtorch + gsplat launches and 2 host syncs.
Host plumbing that the reference does with libtorch ops (masked_select / randperm / cat in addGaussians,
boolean-mask compaction in prunePoints) is done with the same torch ops here.
"""
import math

import torch

import ctypes as C

from . import gsplat_ops as ops
from ._lib import SplatStep, check, lib

C0 = 0.28209479177387814  # gsplat_wapper.cpp:117


def rgb2sh(rgb):
    return (rgb - 0.5) / C0


def sh2rgb(sh):
    return torch.clamp(sh * C0 + 0.5, 0.0, 1.0)


def numShBases(degree):
    return (1, 4, 9, 16)[degree] if degree < 4 else 25


def pose_inv(c2w):
    """tensor_math.cpp:56-67 poseInv"""
    R, T = c2w[:3, :3], c2w[:3, 3:4]
    Rinv = R.transpose(0, 1)
    out = torch.eye(4, dtype=c2w.dtype, device=c2w.device)
    out[:3, :3] = Rinv
    out[:3, 3:4] = torch.matmul(-Rinv, T)
    return out


class Camera:
    """dataset_reader.h:111-169 (fields the hot path reads)"""

    def __init__(self, cam_id, width, height, fx, fy, cx, cy, c2w, image=None, depth=None, device="cuda:0"):
        self.id, self.width, self.height = cam_id, width, height
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.device = torch.device(device)
        self.c2w = torch.as_tensor(c2w, dtype=torch.float32)
        self.c2w_slam = self.c2w.clone()
        self.K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32)
        self.image, self.depth = image, depth  # [H,W,3] float in [0,1], [H,W,1] float metres
        self._dev = None

    def toGPU(self):
        if self._dev is None:
            d = self.device
            c2w = self.c2w_slam.to(d)
            self._dev = dict(viewmat=pose_inv(c2w).contiguous(), K=self.K.to(d).contiguous(),
                             cam_pos=c2w[:3, 3].contiguous())
            if self.image is not None:
                self.image = self.image.to(d)
            if self.depth is not None:
                self.depth = self.depth.to(d)
        return self._dev

    def invalidate(self):
        self._dev = None


class RawGaussianParams:
    NAMES = ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        self.means = self.scales = self.quats = self.featuresDc = self.featuresRest = self.opacities = None
        self.exposure = None

    def isDefined(self):
        return self.means is not None

    def getGaussianNum(self):
        return 0 if self.means is None else self.means.shape[0]

    def tensors(self):
        return [getattr(self, n) for n in self.NAMES]


def knn_mean_dist2(points):
    """distCUDA2 (gsplat/rasterizer/simple_knn.cu:191-240): mean squared distance to the 3 nearest neighbours.
    Host plumbing for Gaussian creation (every 10 frames, P ~ 1e2..1e4); exact brute force in chunks."""
    P = points.shape[0]
    if P <= 1:
        return torch.zeros(P, device=points.device)
    out = torch.empty(P, device=points.device)
    k = min(3, P - 1)
    for s in range(0, P, 4096):
        d = torch.cdist(points[s:s + 4096], points).pow(2)
        vals = torch.topk(d, k + 1, dim=1, largest=False).values[:, 1:]
        out[s:s + 4096] = vals.sum(1) / 3.0
    return out


def compute_quat(init_vec, target_vec):
    """tensor_math.cpp:184-201 computeQuat + quaternionFromAxisAngle"""
    axis = torch.cross(init_vec, target_vec, dim=1)
    axis = axis / (torch.norm(axis, 2, -1, True) + 1e-8)
    angle = torch.acos(torch.sum(init_vec * target_vec, 1)).unsqueeze(-1)
    naxis = axis / (torch.norm(axis, 2, -1, True) + 1e-8)
    half = angle / 2
    return torch.cat([torch.cos(half), naxis * torch.sin(half)], 1)


class RawGaussianModel:
    def __init__(self, cfg=None, device="cuda:0"):
        cfg = dict(cfg or {})
        self.device = torch.device(device)
        self.opt_gs_params = RawGaussianParams(device)
        # raw_gs_model.h:283-288 + configs/release/replica/office0.yaml MODEL section
        self.eps2d, self.near_plane, self.far_plane, self.radius_clip = 0.3, 0.01, 1e10, 0.0
        self.tile_size = 16
        self.max_gs_radii = cfg.get("max_gs_radii", 100)
        self.delta_depth = cfg.get("delta_depth", 0.1)
        self.maxSH = cfg.get("sh_degree", 3)
        self.degreesToUse = self.maxSH
        self.maxInitScale = cfg.get("max_init_scale", 0.01)
        self.minInitScale = cfg.get("min_init_scale", -1)
        self.defaultOpacities = cfg.get("default_opacities", 0.5)
        self.lrs = dict(means=cfg.get("means_lr", 1.6e-4), scales=cfg.get("scales_lr", 5e-3),
                        quats=cfg.get("quats_lr", 1e-3), featuresDc=cfg.get("featuresDc_lr", 2.5e-3),
                        featuresRest=cfg.get("featuresRest_lr", 5e-4), opacities=cfg.get("opacities_lr", 5e-2))
        self.isect_capacity = cfg.get("isect_capacity", None)
        self._opt = None
        self._isect = None
        self._bufs = {}
        self._step = None
        self._step_key = None
        self._step_bufs = None

    # ------------------------------------------------------------------ parameters
    def getGaussianNum(self):
        return self.opt_gs_params.getGaussianNum()

    def getRealScales(self):
        return torch.exp(self.opt_gs_params.scales)

    def getRealOpacities(self):
        return torch.sigmoid(self.opt_gs_params.opacities)

    def _buf(self, name, shape, dtype=torch.float32):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    # ------------------------------------------------------------------ forward (gesForward)
    def _render(self, cam, ref_depth, base_color):
        p = self.opt_gs_params
        N = p.getGaussianNum()
        W, H = cam.width, cam.height
        tw, th = math.ceil(W / self.tile_size), math.ceil(H / self.tile_size)
        c = cam.toGPU()
        # ref_depth_clamped = where(ref < 0.01, 1000, ref) (raw_gs_model.cpp:205-207)
        ref_clamped = torch.where(ref_depth < 0.01, torch.full_like(ref_depth, 1000.0), ref_depth)
        out = (self._buf("radii", (N,), torch.int32), self._buf("means2d", (N, 2)), self._buf("depths", (N,)),
               self._buf("conics", (N, 3)), self._buf("colors", (N, 4)), self._buf("opac", (N,)))
        radii, means2d, depths, conics, colors, opac = ops.gauss_preprocess_fwd(
            p.means, p.scales, p.quats, p.opacities.view(-1), p.featuresDc, p.featuresRest, self.degreesToUse,
            c["viewmat"], c["K"], c["cam_pos"], W, H, self.eps2d, self.near_plane, self.far_plane, self.radius_clip,
            self.max_gs_radii, out=out)
        if self._isect is not None and (self._isect.tiles_per_gauss.numel() != N or self._isect.tile_width != tw or
                                        self._isect.tile_height != th):
            self._isect = None
        self._isect = ops.isect_tiles_no_depth(means2d.view(1, N, 2), radii.view(1, N), self.tile_size, tw, th,
                                               isect_capacity=self.isect_capacity,
                                               group_capacity=None if self.isect_capacity is None else 2 * self.isect_capacity,
                                               out=self._isect)
        rc, ra, _ = ops.rasterize_to_pixels_fwd_ges(means2d, conics, colors, opac, ref_clamped, W, H, self.tile_size,
                                                    self._isect, self.delta_depth)
        return dict(radii=radii, means2d=means2d, depths=depths, conics=conics, colors=colors, opac=opac,
                    ref_clamped=ref_clamped, render_colors=rc, weight_sum=ra, cam=c, W=W, H=H)

    def forward(self, cam, ref_depth, base_color):
        """gesForward under NoGradGuard -> {rgb, depth, alpha, radiis, means2d}"""
        st = self._render(cam, ref_depth, base_color)
        rgb, depth, _, _, _ = ops.compose_l1(st["render_colors"], st["weight_sum"], base_color, ref_depth, None)
        return dict(rgb=rgb, depth=depth, alpha=st["weight_sum"][0], radiis=st["radii"], means2d=st["means2d"])

    # ------------------------------------------------------------------ optimisation step
    def initOptimizers(self, max_iterations=-1, scene_scale=1.0):
        """raw_gs_model.cpp:654-675: all Adam state is re-created (step counts restart at 1)."""
        p = self.opt_gs_params
        params = p.tensors()
        lrs = [self.lrs["means"] * scene_scale, self.lrs["scales"], self.lrs["quats"], self.lrs["featuresDc"],
               self.lrs["featuresRest"], self.lrs["opacities"]]
        self._opt = dict(m=[torch.zeros_like(t) for t in params], v=[torch.zeros_like(t) for t in params],
                         g=[torch.empty_like(t) for t in params], lrs=lrs, step=0)
        self._step = None  # re-bind the step struct to the fresh state

    def _step_struct(self, W, H):
        """Persistent gps_splat_step for the current N / image size (rebuilt after add/prune)."""
        p = self.opt_gs_params
        N = p.getGaussianNum()
        key = (N, W, H, p.means.data_ptr())
        if self._step is not None and self._step_key == key:
            return self._step
        d = self.device
        tw, th = math.ceil(W / self.tile_size), math.ceil(H / self.tile_size)
        icap = int(self.isect_capacity or max(1 << 20, 16 * N))
        gcap = 2 * icap
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=d)
        i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=d)
        B = dict(radii=i32(N), means2d=f(N, 2), depths=f(N), conics=f(N, 3), colors=f(N, 4), opacities=f(N),
                 tiles_per_gauss=i32(N), flatten_ids=i32(icap), group_gs_ids=i32(gcap), group_starts=i32(gcap),
                 tile_offsets=i32(th * tw), counts=torch.zeros(4, dtype=torch.int64, device=d),
                 workspace=torch.empty(int(lib.gps_isect_workspace_bytes(N, icap)), dtype=torch.uint8, device=d),
                 render_colors=f(1, H, W, 4), weight_sum=f(1, H, W, 1), rgb=f(H, W, 3), loss=torch.zeros(1, device=d),
                 v_render_colors=f(1, H, W, 4), v_render_alphas=f(1, H, W, 1), v_means2d=f(N, 2), v_conics=f(N, 3),
                 v_colors=f(N, 4), v_opacities=f(N))
        st = SplatStep()
        st.N, st.K, st.sh_degree, st.width, st.height = N, 1 + p.featuresRest.shape[1], self.degreesToUse, W, H
        st.max_gs_radii = int(self.max_gs_radii)
        st.eps2d, st.near_plane, st.far_plane, st.radius_clip = self.eps2d, self.near_plane, self.far_plane, self.radius_clip
        st.delta_depth = self.delta_depth
        for name, t in (("means", p.means), ("log_scales", p.scales), ("quats", p.quats), ("opac_logit", p.opacities),
                        ("sh_dc", p.featuresDc), ("sh_rest", p.featuresRest)):
            assert t.is_contiguous()
            setattr(st, name, t.data_ptr())
        for name, t in B.items():
            setattr(st, name, t.data_ptr())
        st.isect_capacity, st.group_capacity, st.workspace_bytes = icap, gcap, B["workspace"].numel()
        st.beta1, st.beta2, st.adam_eps = 0.9, 0.999, 1e-15
        self._step, self._step_key, self._step_bufs = st, key, B
        self._bind_optimizer()
        return st

    def _bind_optimizer(self):
        st, o = self._step, self._opt
        if st is None or o is None:
            return
        order = (0, 1, 2, 5, 3, 4)  # struct order means, log_scales, quats, opac_logit, sh_dc, sh_rest vs NAMES order
        names = ("means", "log_scales", "quats", "opac_logit", "sh_dc", "sh_rest")
        for n, k in zip(names, order):
            setattr(st, "g_" + n, o["g"][k].data_ptr())
            setattr(st, "m_" + n, o["m"][k].data_ptr())
            setattr(st, "v_" + n, o["v"][k].data_ptr())
        lr = o["lrs"]
        # lr order in the struct: means, log_scales, quats, sh_dc, sh_rest, opac_logit == NAMES order
        for j in range(6):
            st.lr[j] = float(lr[j])

    def _bind_camera(self, st, cam, ref_depth_clamped, base_color, gt_rgb):
        c = cam.toGPU()
        st.viewmat, st.Kmat, st.cam_pos = c["viewmat"].data_ptr(), c["K"].data_ptr(), c["cam_pos"].data_ptr()
        st.ref_depth_clamped = ref_depth_clamped.data_ptr()
        st.base_color = base_color.data_ptr()
        st.gt_rgb = 0 if gt_rgb is None else gt_rgb.data_ptr()

    def train_step(self, cam, ref_depth, base_color, gt_rgb, ref_depth_clamped=None):
        """model.forward -> computeLoss -> loss.backward -> optimizersStep/ZeroGrad (slam_pipeline.cpp:247-254) as one
        C-ABI call (gps_splat_train_step).  The L1 loss accumulates in a device scalar (read with last_loss())."""
        if ref_depth_clamped is None:
            ref_depth_clamped = torch.where(ref_depth < 0.01, torch.full_like(ref_depth, 1000.0), ref_depth)
        st = self._step_struct(cam.width, cam.height)
        self._bind_camera(st, cam, ref_depth_clamped, base_color, gt_rgb)
        o = self._opt
        o["step"] += 1
        check(lib.gps_splat_train_step(C.byref(st), o["step"],
                                       C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
              "gps_splat_train_step")
        self._keep = (ref_depth_clamped, base_color, gt_rgb)  # keep the tensors alive until the next call

    def loss_sum(self):
        return self._step_bufs["loss"]

    # ------------------------------------------------------------------ structure edits (every 10 frames)
    def prunePoints(self, delete_mask):
        """raw_gs_model.cpp:635-644 (+ removeFromOptimizer): boolean-mask compaction of params and Adam state"""
        keep = ~delete_mask
        p = self.opt_gs_params
        for n in p.NAMES:
            setattr(p, n, getattr(p, n)[keep].contiguous())
        if self._opt is not None:
            for k in ("m", "v", "g"):
                self._opt[k] = [t[keep].contiguous() for t in self._opt[k]]
        self._isect = None
        self._bufs = {}
        self._step = None


class SLAMGaussianModel(RawGaussianModel):
    def init_params(self, xyz, rgb, normals):
        """RawGaussianParams::init (raw_gs_param.cpp:11-74) -> dict of new tensors"""
        P = xyz.shape[0]
        raw_scales = torch.sqrt(knn_mean_dist2(xyz))
        lo = self.minInitScale if self.minInitScale is not None else None
        raw_scales = raw_scales.clamp(lo, self.maxInitScale).unsqueeze(1).repeat(1, 3)
        quats = torch.ones((P, 4), device=self.device)
        if normals is not None:
            raw_scales[:, 2] = raw_scales[:, 2] * 0.1
            z_axis = torch.zeros_like(raw_scales)
            z_axis[:, 2] = 1
            quats = compute_quat(z_axis, normals)
        K = numShBases(self.maxSH)
        shs = torch.zeros((P, K, 3), device=self.device)
        shs[:, 0, :3] = rgb2sh(rgb)
        opac = torch.logit(self.defaultOpacities * torch.ones((P, 1), device=self.device))
        return dict(means=xyz.contiguous(), scales=raw_scales.log().contiguous(), quats=quats.contiguous(),
                    featuresDc=shs[:, 0, :].contiguous(), featuresRest=shs[:, 1:, :].contiguous(),
                    opacities=opac.contiguous())

    def add_params(self, new):
        p = self.opt_gs_params
        for n in p.NAMES:
            cur = getattr(p, n)
            setattr(p, n, new[n] if cur is None else torch.cat([cur, new[n]], 0))
        self._isect = None
        self._bufs = {}
        self._step = None

    def addGaussians(self, cam, frame_maps, sample_mask, new_gs_sample_ratio, frame_num, generator=None):
        """slam/slam_gs_model.cpp:5-56"""
        H, W = cam.image.shape[0], cam.image.shape[1]
        m = sample_mask.expand(H, W, 3)
        verts = torch.masked_select(frame_maps["vertex_map"], m).reshape(-1, 3)
        cols = torch.masked_select(cam.image, m).reshape(-1, 3)
        norms = torch.masked_select(frame_maps["normal_map"], m).reshape(-1, 3)
        n = verts.shape[0]
        num_select = int(n * new_gs_sample_ratio)
        if num_select <= 0:
            return 0
        perm = torch.randperm(n, device=verts.device, generator=generator)[:num_select]
        self.add_params(self.init_params(verts[perm], cols[perm], norms[perm]))
        return num_select
