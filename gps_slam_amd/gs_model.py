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
import itertools
import math
import os
import sys

import numpy as np
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


_PACK_SERIAL = itertools.count(1)

# RawGaussianModel::initOptimizers computes eps and the betas in FLOAT variables (raw_gs_model.cpp:661-664) which AdamOptions
# widens: beta1 = 0.8999999761581421, beta2 = 0.9990000128746033, eps = 1.0000000036274937e-15 (tests/test_adam_libtorch_gpu.py)
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = float(np.float32(0.9)), float(np.float32(0.999)), float(np.float32(1e-15))


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
        """curr_cam.toGPU() (slam_pipeline.cpp:84): viewmat = poseInv(c2w_slam), K and the camera position go up in ONE
        28-float upload; poseInv's 3x3 algebra runs on the host (the reference launches ~8 tiny device kernels for it)."""
        if self._dev is None:
            d = self.device
            c2w = self.c2w_slam.to(torch.float32)
            pack = torch.cat([pose_inv(c2w).reshape(-1), self.K.reshape(-1), c2w[:3, 3].reshape(-1)]).to(d)
            # serial: identity of this upload (an address can come back from the allocator with another pose in it)
            self._dev = dict(viewmat=pack[:16].view(4, 4), K=pack[16:25].view(3, 3), cam_pos=pack[25:28], _pack=pack,
                             serial=next(_PACK_SERIAL))
            if self.image is not None:
                self.image = self.image.to(d)
            if self.depth is not None:
                self.depth = self.depth.to(d)
        return self._dev

    def invalidate(self):
        self._dev = None


class RawGaussianParams:
    """include/raw_gs_param.h:7-85.  MI355X layout: every tensor is a [:N] view into a capacity-sized buffer that
    is allocated once (288 GB of HBM: 1M Gaussians incl. Adam state = 0.94 GB), so add / remove never reallocate --
    the reference re-`cat`s / re-indexes all seven tensors into fresh allocations on every add and prune
    (raw_gs_param.cpp:123-157), which on any caching allocator means new hipMalloc calls as N drifts."""
    NAMES = ("means", "scales", "quats", "featuresDc", "featuresRest", "opacities")

    def __init__(self, device="cuda:0", capacity=1 << 19, sh_k=16):
        self.device = torch.device(device)
        self.K = sh_k
        self.N = 0
        self.cap = 0
        self._buf, self._alt = {}, {}
        self.exposure = None
        self.version = 0   # bumped by every add / remove / load: part of the key of a forward run ahead (GaussianModel.train_step)
        self._reserve(capacity)

    def _shapes(self):
        return dict(means=(3,), scales=(3,), quats=(4,), featuresDc=(3,), featuresRest=(self.K - 1, 3), opacities=(1,))

    def _reserve(self, capacity):
        if capacity <= self.cap:
            return
        for n, shp in self._shapes().items():
            nb = torch.empty((capacity,) + shp, dtype=torch.float32, device=self.device)
            if self.N:
                nb[:self.N] = self._buf[n][:self.N]
            self._buf[n] = nb
            self._alt[n] = torch.empty_like(nb)
        self.cap = capacity

    def isDefined(self):
        return self.N > 0

    def getGaussianNum(self):
        return self.N

    def tensors(self):
        return [self._buf[n][:self.N] for n in self.NAMES]

    def add(self, new):
        """RawGaussianParams::add (raw_gs_param.cpp:123-145): append in place"""
        n = new["means"].shape[0]
        if self.N + n > self.cap:
            self._reserve(max(2 * self.cap, self.N + n))
        for name in self.NAMES:
            self._buf[name][self.N:self.N + n] = new[name]
        self.N += n
        self.version += 1

    def savePly(self, filename):
        """RawGaussianParams::savePly (raw_gs_param.cpp:159-217): binary little-endian 3DGS PLY, one float row per Gaussian:
        x y z | nx ny nz (zeros) | f_dc_0..2 | f_rest_* (channel-major: featuresRest.transpose(1,2)) | opacity | scale_0..2 |
        rot_0..3 -- raw (log / logit) parameters, as the reference stores them."""
        write_gaussian_ply(filename, *[t.detach().cpu().numpy() for t in self.tensors()])

    def loadPly(self, filename):
        """inverse of savePly (the reference has no reader; the 3DGS viewers it targets define the format)"""
        new = read_gaussian_ply(filename)
        self.N = 0
        self.add({k: torch.from_numpy(v).to(self.device) for k, v in new.items()})

    def remove(self, keep_idx):
        """RawGaussianParams::remove (raw_gs_param.cpp:148-157): stable compaction into the alternate buffers"""
        m = keep_idx.shape[0]
        for name in self.NAMES:
            torch.index_select(self._buf[name][:self.N], 0, keep_idx, out=self._alt[name][:m])
            self._buf[name], self._alt[name] = self._alt[name], self._buf[name]
        self.N = m
        self.version += 1


def gaussian_ply_properties(n_dc, n_rest):
    """property names in file order (raw_gs_param.cpp:167-196)"""
    return (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(n_dc)] + ["f_rest_%d" % i for i in range(n_rest)] +
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])


def write_gaussian_ply(filename, means, scales, quats, features_dc, features_rest, opacities):
    """numpy arrays in RawGaussianParams.NAMES order -> the reference's PLY bytes"""
    import numpy as np
    n = means.shape[0]
    rest = np.ascontiguousarray(np.transpose(features_rest, (0, 2, 1))).reshape(n, -1)  # .transpose(1, 2).reshape({N, -1})
    props = gaussian_ply_properties(features_dc.shape[1], rest.shape[1])
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % p for p in props) + \
             "end_header\n"
    rows = np.concatenate([means, np.zeros((n, 3), np.float32), features_dc, rest, opacities.reshape(n, 1), scales, quats], 1)
    with open(filename, "wb") as f:
        f.write(header.encode())
        f.write(np.ascontiguousarray(rows, dtype="<f4").tobytes())


def read_gaussian_ply(filename):
    import numpy as np
    raw = open(filename, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode().split("\n")
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
    n = int(lines[2].split()[-1])
    props = [l.split()[-1] for l in lines[3:] if l.startswith("property float")]
    rows = np.frombuffer(raw[end:], "<f4").reshape(n, len(props)).copy()
    col = {p: i for i, p in enumerate(props)}
    n_rest = sum(p.startswith("f_rest_") for p in props)
    pick = lambda names: rows[:, [col[x] for x in names]]
    rest = pick(["f_rest_%d" % i for i in range(n_rest)]).reshape(n, 3, n_rest // 3).transpose(0, 2, 1)
    return dict(means=pick(["x", "y", "z"]), scales=pick(["scale_0", "scale_1", "scale_2"]),
                quats=pick(["rot_0", "rot_1", "rot_2", "rot_3"]), featuresDc=pick(["f_dc_0", "f_dc_1", "f_dc_2"]),
                featuresRest=np.ascontiguousarray(rest), opacities=pick(["opacity"]))


for _n in RawGaussianParams.NAMES:
    setattr(RawGaussianParams, _n, property(lambda self, _n=_n: self._buf[_n][:self.N]))


KNN_GRID_MIN_POINTS = 8192   # include/gps_slam_hip.h GPS_KNN_GRID_MIN_POINTS


def knn_mean_dist2(points, method=None):
    """distCUDA2 (gsplat/rasterizer/simple_knn.cu:191-240) -> gps_knn_mean_dist2 (tiled brute force) up to a few thousand
    points, gps_knn_mean_dist2_grid (exact uniform-grid search, same bits) above; method = "brute" / "grid" forces one"""
    points = points.contiguous()
    P = points.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=points.device)
    stream = C.c_void_p(torch.cuda.current_stream(points.device).cuda_stream)
    if method == "grid" or (method is None and P > KNN_GRID_MIN_POINTS):
        nbytes = int(lib.gps_knn_grid_workspace_bytes(P))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
        check(lib.gps_knn_mean_dist2_grid(P, points.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, stream),
              "gps_knn_mean_dist2_grid")
    else:
        check(lib.gps_knn_mean_dist2(P, points.data_ptr(), out.data_ptr(), stream), "gps_knn_mean_dist2")
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
        self.opt_gs_params = RawGaussianParams(device, capacity=cfg.get("capacity", 1 << 19),
                                               sh_k=numShBases(self.maxSH))
        self.isect_capacity = cfg.get("isect_capacity", None)
        # gps_splat_step.fuse_sh_rest_adam: 0 = gradients are written and a separate Adam kernel steps (grads() is valid),
        # 2 = what the C++ host runs (host/raw_gs_model.cpp): every tensor stepped inside the backward kernel
        self.fuse_adam = int(cfg.get("fuse_sh_rest_adam", 0))
        # the fused iteration's binning + backward rasterizer: superblock counting sort + column strips (default) or the
        # sorted-key binning + 32-pixel-group kernel of the operator-level entry points
        self.strip_backward = bool(cfg.get("strip_backward", True))
        self._opt = None       # Adam state: capacity-sized m / v / g buffers + step count
        self._step = None      # persistent gps_splat_step
        self._step_key = None
        self._B = None         # capacity-sized intermediates the step struct points at
        self._keep = None

    # ------------------------------------------------------------------ parameters
    def getGaussianNum(self):
        return self.opt_gs_params.getGaussianNum()

    def getRealScales(self):
        return torch.exp(self.opt_gs_params.scales)

    def getRealOpacities(self):
        return torch.sigmoid(self.opt_gs_params.opacities)

    # ------------------------------------------------------------------ persistent launch descriptor
    def _step_struct(self, W, H):
        """gps_splat_step for the current buffers.  Intermediates are sized by the parameter CAPACITY, so the struct
        survives add / prune; only N and (after a prune's buffer swap) the parameter pointers are refreshed."""
        p = self.opt_gs_params
        key = (p.cap, W, H)
        d = self.device
        if self._step is None or self._step_key != key:
            self._prefetched = None   # a forward run ahead lived in the buffers replaced below
            cap = p.cap
            tw, th = math.ceil(W / self.tile_size), math.ceil(H / self.tile_size)
            icap = int(self.isect_capacity or max(1 << 20, 16 * cap))
            gcap = 2 * icap
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=d)
            i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=d)
            B = dict(radii=i32(cap), means2d=f(cap, 2), depths=f(cap), conics=f(cap, 3), colors=f(cap, 4),
                     opacities=f(cap), records=f(cap, 12), tiles_per_gauss=i32(cap), flatten_ids=i32(icap), group_gs_ids=i32(gcap),
                     group_starts=i32(gcap), tile_offsets=i32(th * tw), counts=torch.zeros(4, dtype=torch.int64, device=d),
                     # (zero-filled: the superblock binning's count tables are kept zero between launches by the kernels)
                     workspace=torch.zeros(int(lib.gps_isect_workspace_bytes(cap, icap)), dtype=torch.uint8, device=d),
                     render_colors=f(1, H, W, 4), weight_sum=f(1, H, W, 1), rgb=f(H, W, 3), depth=f(H, W, 1),
                     loss=torch.zeros(1, device=d), v_render_colors=f(1, H, W, 4), v_render_alphas=f(1, H, W, 1),
                     v_means2d=f(cap, 2), v_conics=f(cap, 3), v_colors=f(cap, 4), v_opacities=f(cap))
            if self.strip_backward:  # buffers of the strip backward (gps_splat_step: all set -> superblock binning + strips)
                B.update(v_rows=f(cap, 12), pix2=f(H * W, 2), cls_ids=i32(5, cap), cls_counts=torch.zeros(8, dtype=torch.int32, device=d))
            st = SplatStep()
            st.K, st.sh_degree, st.width, st.height = p.K, self.degreesToUse, W, H
            st.max_gs_radii = int(self.max_gs_radii)
            st.eps2d, st.near_plane, st.far_plane, st.radius_clip = self.eps2d, self.near_plane, self.far_plane, self.radius_clip
            st.delta_depth = self.delta_depth
            for name, t in B.items():
                if hasattr(st, name):
                    setattr(st, name, t.data_ptr())
            st.isect_capacity, st.group_capacity, st.workspace_bytes = icap, gcap, B["workspace"].numel()
            st.cls_stride = cap
            st.beta1, st.beta2, st.adam_eps = ADAM_BETA1, ADAM_BETA2, ADAM_EPS
            st.fuse_sh_rest_adam = self.fuse_adam
            self._step, self._step_key, self._B = st, key, B
        st = self._step
        st.N = p.N
        for name, src in (("means", "means"), ("log_scales", "scales"), ("quats", "quats"), ("opac_logit", "opacities"),
                          ("sh_dc", "featuresDc"), ("sh_rest", "featuresRest")):
            setattr(st, name, p._buf[src].data_ptr())
        o = self._opt
        if o is not None:
            order = (("means", 0), ("log_scales", 1), ("quats", 2), ("opac_logit", 5), ("sh_dc", 3), ("sh_rest", 4))
            for n, k in order:
                setattr(st, "g_" + n, o["g"][k].data_ptr())
                setattr(st, "m_" + n, o["m"][k].data_ptr())
                setattr(st, "v_" + n, o["v"][k].data_ptr())
            for j in range(6):  # struct lr order == NAMES order: means, scales, quats, featuresDc, featuresRest, opacities
                st.lr[j] = float(o["lrs"][j])
        return st

    def _bind_camera(self, st, cam, ref_depth_clamped, base_color, gt_rgb, consumes_prefetch=False):
        if getattr(self, "_prefetched", None) is not None and not consumes_prefetch:
            # a forward was run ahead for a train step that is not coming: its counts must leave the binning's tables
            check(lib.gps_splat_discard_prefetch(C.byref(st), self._stream()), "gps_splat_discard_prefetch")
        c = cam.toGPU()
        st.viewmat, st.Kmat, st.cam_pos = c["viewmat"].data_ptr(), c["K"].data_ptr(), c["cam_pos"].data_ptr()
        st.ref_depth_clamped = ref_depth_clamped.data_ptr()
        st.base_color = base_color.data_ptr()
        st.gt_rgb = 0 if gt_rgb is None else gt_rgb.data_ptr()
        st.next_viewmat = st.next_Kmat = st.next_cam_pos = None
        st.preprocessed = 0
        self._prefetched = None   # (whatever runs on the step buffers invalidates a prefetched forward; train_step re-arms it)
        self._keep = (ref_depth_clamped, base_color, gt_rgb)  # alive until the next call

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def clamp_ref_depth(ref_depth):
        """ref_depth_clamped = where(ref < 0.01, 1000, ref) (raw_gs_model.cpp:205-207)"""
        return torch.where(ref_depth < 0.01, torch.full_like(ref_depth, 1000.0), ref_depth)

    # ------------------------------------------------------------------ forward (gesForward under NoGradGuard)
    def forward(self, cam, ref_depth, base_color, ref_depth_clamped=None):
        """-> {rgb[H,W,3], depth[H,W,1], alpha[H,W,1], radiis[N], means2d[N,2]} (views into persistent buffers)"""
        if ref_depth_clamped is None:
            ref_depth_clamped = self.clamp_ref_depth(ref_depth)
        st = self._step_struct(cam.width, cam.height)
        self._bind_camera(st, cam, ref_depth_clamped, base_color, None)
        check(lib.gps_splat_render(C.byref(st), self._stream()), "gps_splat_render")
        B = self._B
        check(lib.gps_compose_l1(cam.width, cam.height, B["render_colors"].data_ptr(), B["weight_sum"].data_ptr(),
                                 base_color.data_ptr(), ref_depth.data_ptr(), None, B["rgb"].data_ptr(),
                                 B["depth"].data_ptr(), None, None, None, self._stream()), "gps_compose_l1")
        N = self.getGaussianNum()
        return dict(rgb=B["rgb"], depth=B["depth"], alpha=B["weight_sum"][0], radiis=B["radii"][:N],
                    means2d=B["means2d"][:N])

    # ------------------------------------------------------------------ optimisation
    def initOptimizers(self, max_iterations=-1, scene_scale=1.0):
        """raw_gs_model.cpp:654-675: all Adam state is re-created (step counts restart at 1)."""
        p = self.opt_gs_params
        # (float members, float product, widened for AdamOptions: raw_gs_model.cpp:26-32, :666-671)
        f32 = np.float32
        lrs = [float(f32(self.lrs["means"]) * f32(scene_scale))] + [float(f32(self.lrs[k])) for k in ("scales", "quats", "featuresDc", "featuresRest", "opacities")]
        if self._opt is None or self._opt["cap"] != p.cap:
            mk = lambda: [torch.zeros_like(p._buf[n]) for n in p.NAMES]
            self._opt = dict(m=mk(), v=mk(), g=mk(), cap=p.cap)
        # (an existing state is NOT zeroed: step 1 of gps_splat_train_step / gps_adam_step takes the moments as zero without reading
        # them and writes every live row -- include/gps_slam_hip.h, gps_adam_step)
        self._opt.update(lrs=lrs, step=0)

    def grads(self):
        """parameter gradients of the last train_step, NAMES order ([:N] views)"""
        N = self.getGaussianNum()
        return [t[:N] for t in self._opt["g"]]

    def train_step(self, cam, ref_depth, base_color, gt_rgb, ref_depth_clamped=None, next_cam=None):
        """model.forward -> computeLoss -> loss.backward -> optimizersStep/ZeroGrad (slam_pipeline.cpp:247-254) as one
        C-ABI call (gps_splat_train_step).  The L1 loss accumulates in a device scalar (loss_sum()).
        next_cam: the camera of the NEXT train_step call (same size) -- its preprocessing forward then runs in the tail of this
        step's backward kernel (gps_splat_step::next_viewmat) and the next call skips its preprocessing launch, provided it is
        called with that camera and nothing else has used the model in between (any other call, an add / prune / load of the
        parameters: the forward run ahead is discarded and the step preprocesses again; editing the parameter tensors in place is
        the one thing this cannot see); ignored where the step cannot prefetch."""
        if ref_depth_clamped is None:
            ref_depth_clamped = self.clamp_ref_depth(ref_depth)
        st = self._step_struct(cam.width, cam.height)
        c = cam.toGPU()
        ver = self.opt_gs_params.version
        key = (c["serial"], int(st.N), cam.width, cam.height, ver)
        skip = getattr(self, "_prefetched", None) == key
        self._bind_camera(st, cam, ref_depth_clamped, base_color, gt_rgb, consumes_prefetch=skip)
        st.preprocessed = 1 if skip else 0
        if next_cam is not None and (next_cam.width, next_cam.height) == (cam.width, cam.height) and lib.gps_splat_can_prefetch(C.byref(st)):
            n = next_cam.toGPU()
            st.next_viewmat, st.next_Kmat, st.next_cam_pos = n["viewmat"].data_ptr(), n["K"].data_ptr(), n["cam_pos"].data_ptr()
            armed = (n["serial"], int(st.N), cam.width, cam.height, ver)
        else:
            armed = None
        o = self._opt
        o["step"] += 1
        check(lib.gps_splat_train_step(C.byref(st), o["step"], self._stream()), "gps_splat_train_step")  # raises on error: nothing armed
        self._prefetched = armed

    def loss_sum(self):
        return self._B["loss"]

    def check_binning_capacity(self):
        """RawGaussianModel::checkBinningCapacity (host/raw_gs_model.cpp): blocking read-back of the binning counts.  The
        kernels clamp n_isects / n_groups to the buffer capacities and raise a sticky flag where the reference would have
        allocated exact sizes; an overflow since the last check raises, more than half of a capacity in use doubles the
        buffers before the next iteration.  -> (n_isects, n_groups) of the last launch."""
        if self._B is None:
            return 0, 0
        ni, ng, overflow, _ = self._B["counts"].cpu().tolist()
        icap = int(self._step.isect_capacity)
        want, need = icap, max(ni, (ng + 1) // 2)
        while want < 2 * need:
            want *= 2
        if overflow:
            want = max(want, 2 * icap)
        if want != icap:
            self.isect_capacity = want
            self._step = None  # re-created (with zeroed counts) by the next _step_struct
        if overflow:  # grow and carry on, loudly (the reference sizes the buffers exactly per forward and never fails here)
            import warnings
            self.binning_overflows = getattr(self, "binning_overflows", 0) + 1
            warnings.warn("tile-intersection buffers overflowed (capacity %d): Gaussians were dropped from a render or a backward "
                          "pass since the last check; capacity raised to %d for the following iterations" % (icap, want))
        return ni, ng

    # ------------------------------------------------------------------ structure edits (every 10 frames)
    def prunePoints(self, delete_mask):
        """raw_gs_model.cpp:635-644 (+ removeFromOptimizer): stable compaction of params and Adam state"""
        p = self.opt_gs_params
        N = p.N
        keep_idx = torch.nonzero(~delete_mask, as_tuple=False).squeeze(1)
        if self._opt is not None and self._opt["cap"] == p.cap:
            for k in ("m", "v"):
                for j, name in enumerate(p.NAMES):
                    t = self._opt[k][j]
                    torch.index_select(t[:N], 0, keep_idx, out=p._alt[name][:keep_idx.shape[0]])
                    t[:keep_idx.shape[0]] = p._alt[name][:keep_idx.shape[0]]
        p.remove(keep_idx)


class SLAMGaussianModel(RawGaussianModel):
    def init_params(self, xyz, rgb, normals):
        """RawGaussianParams::init (raw_gs_param.cpp:11-74) -> dict of new tensors"""
        P = xyz.shape[0]
        raw_scales = torch.sqrt(knn_mean_dist2(xyz))
        raw_scales = raw_scales.clamp(self.minInitScale, self.maxInitScale).unsqueeze(1).repeat(1, 3)
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
        return dict(means=xyz, scales=raw_scales.log(), quats=quats, featuresDc=shs[:, 0, :],
                    featuresRest=shs[:, 1:, :], opacities=opac)

    def add_params(self, new):
        self.opt_gs_params.add(new)

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
        # uniformly random subset of num_select pixels (the reference: torch::randperm(n)[:num_select]).  Drawn on the
        # host: n is already known there (masked_select synced) and the CUDA randperm/sort path stalls for ms on ROCm.
        # The subset is the reference's; its ORDER is not: appended in pixel order instead of permutation order, so that
        # Gaussians with neighbouring ids splat onto neighbouring pixels.  The Gaussian-parallel backward walks Gaussians in
        # id order and gathers the gradient image per pixel -- with random ids every XCD's 4 MB L2 thrashes over the whole
        # 7 MB image (rocprofv3 FETCH_SIZE: 190 MB per launch); with pixel order its working set is a narrow band.
        perm = torch.randperm(n, generator=generator)[:num_select].sort().values.to(verts.device)  # CPU generator: n is host-known
        self.add_params(self.init_params(verts[perm].contiguous(), cols[perm], norms[perm]))
        return num_select
