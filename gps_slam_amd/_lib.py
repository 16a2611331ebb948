"""ctypes binding of include/gps_slam_hip.h.  Fails loudly if the HIP library is missing:
there is NO CPU fallback for any operator in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libgpsslam_hip.so")
_lib = None

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


class TsdfState(C.Structure):
    """gps_tsdf_state (include/gps_slam_hip.h)"""
    _fields_ = [("width", i32), ("height", i32), ("fx", f32), ("fy", f32), ("cx", f32), ("cy", f32),
                ("voxel_size", f32), ("mu", f32), ("view_frustum_min", f32), ("view_frustum_max", f32),
                ("max_w", i32), ("n_blocks", i32), ("n_buckets", i32), ("n_excess", i32),
                ("vba", vp), ("vba_alloc_list", vp), ("hash", vp), ("excess_list", vp), ("counters", vp),
                ("alloc_prio", vp), ("scan_scratch", vp), ("visible_type", vp), ("visible_ids", vp), ("depth", vp),
                ("rgb", vp), ("minmax", vp), ("raycast", vp), ("icp_points", vp), ("icp_normals", vp),
                ("fv_visible_ids", vp), ("fv_minmax", vp), ("fv_raycast", vp), ("fv_colour", vp)]


class TsdfView(C.Structure):
    """gps_tsdf_view: one free view of gps_tsdf_free_raycast_batch"""
    _fields_ = [("M", f32 * 16), ("invM", f32 * 16), ("fx", f32), ("fy", f32), ("cx", f32), ("cy", f32),
                ("visible_ids", vp), ("minmax", vp), ("raycast", vp), ("colour", vp), ("scratch", vp), ("counters", vp),
                ("w2c", f32 * 16), ("color_map", vp), ("vertex_map", vp), ("confidence_map", vp), ("depth_map", vp),
                ("depth_map_clamped", vp)]


class TrackConfig(C.Structure):
    """gps_track_config"""
    _fields_ = [("n_levels", i32), ("iter_type", i32 * 8), ("n_iter", i32 * 8), ("space_thresh", f32 * 8),
                ("term_thresh", f32), ("tukey_cutoff", f32), ("frames_to_skip", i32), ("frames_to_weight", i32)]


class TrackState(C.Structure):
    """gps_track_state (host memory)"""
    _fields_ = [("pose_M", f32 * 16), ("pose_invM", f32 * 16), ("pose_pc_M", f32 * 16), ("age_point_cloud", i32),
                ("frames_processed", i32), ("diag", f32 * 16), ("host_mailbox", vp), ("mail_seq", i32), ("scratch_epoch", i32), ("dev_arg_line", vp), ("mailbox_bytes", i32)]


class SplatStep(C.Structure):
    """gps_splat_step (include/gps_slam_hip.h)"""
    _fields_ = ([(n, i32) for n in ("N", "K", "sh_degree", "width", "height", "max_gs_radii")] +
                [(n, f32) for n in ("eps2d", "near_plane", "far_plane", "radius_clip", "delta_depth")] +
                [(n, vp) for n in ("means", "log_scales", "quats", "opac_logit", "sh_dc", "sh_rest", "viewmat", "Kmat",
                                   "cam_pos", "ref_depth_clamped", "base_color", "gt_rgb", "radii", "means2d", "depths",
                                   "conics", "colors", "opacities", "records")] +
                [(n, i64) for n in ("isect_capacity", "group_capacity", "workspace_bytes")] +
                [(n, vp) for n in ("tiles_per_gauss", "flatten_ids", "group_gs_ids", "group_starts", "tile_offsets",
                                   "counts", "workspace", "render_colors", "weight_sum", "rgb", "loss",
                                   "v_render_colors", "v_render_alphas", "v_means2d", "v_conics", "v_colors",
                                   "v_opacities")] +
                [(p + n, vp) for p in ("g_", "m_", "v_") for n in ("means", "log_scales", "quats", "opac_logit", "sh_dc",
                                                                   "sh_rest")] +
                [("lr", f64 * 6), ("beta1", f64), ("beta2", f64), ("adam_eps", f64), ("fuse_sh_rest_adam", i32)] +
                [(n, vp) for n in ("v_rows", "pix2", "cls_ids", "cls_counts")] + [("cls_stride", i64)] +
                [(n, vp) for n in ("next_viewmat", "next_Kmat", "next_cam_pos")] + [("preprocessed", i32)])


class AdamSegment(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("numel", i64), ("lr", f64)]


# name -> (restype, argtypes); mirrors include/gps_slam_hip.h one to one
PROTOTYPES = {
    "gps_version": (C.c_char_p, []),
    "gps_build_flags": (C.c_char_p, []),
    "gps_launch_timing_start": (i32, [i32]),
    "gps_launch_timing_stop": (i32, []),
    "gps_launch_timing_read": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "gps_proj_fwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, vp, vp, vp, vp]),
    "gps_proj_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gps_sh_fwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp]),
    "gps_sh_bwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_isect_workspace_bytes": (i64, [i32, i64]),
    "gps_isect_workspace_init": (i32, [vp, i64, vp]),
    "gps_isect_tiles_no_depth": (i32, [i32, vp, vp, i32, i32, i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "gps_ssim_fwd": (i32, [i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_ssim_bwd": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gps_upload_floats": (i32, [vp, vp, i32, vp]),
    "gps_isect_tiles": (i32, [i32, vp, vp, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, i64, vp]),
    "gps_raster_raw_fwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_raster_raw_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                 vp]),
    "gps_raster_ges_fwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp]),
    "gps_raster_ges_bwd_gs": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp,
                                    i32, vp]),
    "gps_raster_ges_bwd_strips": (i32, [i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp]),
    "gps_set_frame_chain_reserve": (None, [i32]),
    "gps_set_raster_fwd_persistent": (None, [i32]),

    "gps_raster_pack_records": (i32, [i32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_raster_pair_image": (i32, [i32, i32, vp, vp, f32, vp, vp]),
    "gps_raster_ges_bwd_exact": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_compose_l1": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gps_adam_step": (i32, [C.POINTER(AdamSegment), i32, f64, f64, f64, i32, vp]),
    "gps_gauss_preprocess_fwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, f32,
                                       i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gps_raster_ges_fwd_rec": (i32, [i32, vp, vp, i32, i32, vp, vp, vp, f32, vp, vp, vp]),
    "gps_raster_ges_fwd_rec_ordered": (i32, [i32, vp, vp, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp]),
    "gps_isect_workspace_tile_order": (vp, [vp, i32, i64]),
    "gps_gauss_preprocess_bwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gps_gauss_preprocess_bwd_adam": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp,
                                            vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, vp, f64, f64, f64, i32, vp]),
    "gps_tsdf_scratch_bytes": (i64, [i32, i32, i32, i32]),
    "gps_tsdf_reset": (i32, [C.POINTER(TsdfState), vp]),
    "gps_tsdf_rebuild_index": (i32, [C.POINTER(TsdfState), vp]),
    "gps_tsdf_mesh_workspace_bytes": (i64, [C.POINTER(TsdfState)]),
    "gps_tsdf_mesh_scene": (i32, [C.POINTER(TsdfState), i64, vp, vp, vp, i64, vp]),
    "gps_track_config_init": (i32, [C.POINTER(TrackConfig), C.c_char_p, i32, i32, f32, f32, f32, f32, i32, i32]),
    "gps_track_state_reset": (i32, [C.POINTER(TrackState)]),
    "gps_track_arg_line_alloc": (i32, [C.POINTER(vp)]),
    "gps_track_arg_line_free": (i32, [vp]),
    "gps_track_scratch_bytes": (i64, [i32, i32]),
    "gps_track_poll_profile": (i32, [vp, i32, i32, vp, vp]),
    "gps_track_poll_phases": (i32, [vp, i32, i32, vp, vp]),
    "gps_tsdf_track_camera": (i32, [C.POINTER(TsdfState), C.POINTER(TrackConfig), C.POINTER(TrackState), vp, i64, vp]),
    "gps_tsdf_process_frame_tracked": (i32, [C.POINTER(TsdfState), vp, C.POINTER(TrackConfig), C.POINTER(TrackState), vp, i64,
                                             vp]),
    "gps_tsdf_process_frame_tracked_gated": (i32, [C.POINTER(TsdfState), vp, C.POINTER(TrackConfig), C.POINTER(TrackState), vp, i64,
                                                   vp, vp, vp]),
    "gps_tsdf_convert_depth": (i32, [C.POINTER(TsdfState), vp, vp]),
    "gps_tsdf_allocate": (i32, [C.POINTER(TsdfState), vp, vp, vp]),
    "gps_tsdf_integrate": (i32, [C.POINTER(TsdfState), vp, vp]),
    "gps_tsdf_expected_depths": (i32, [C.POINTER(TsdfState), vp, i32, vp]),
    "gps_tsdf_expected_depths_partial": (i32, [C.POINTER(TsdfState), vp, i32, vp]),
    "gps_tsdf_raycast": (i32, [C.POINTER(TsdfState), vp, i32, i32, vp]),
    "gps_tsdf_expected_depths_and_raycast": (i32, [C.POINTER(TsdfState), vp, vp, i32, i32, vp]),
    "gps_tsdf_icp_maps": (i32, [C.POINTER(TsdfState), vp, vp]),
    "gps_tsdf_ray_stats": (i32, [C.POINTER(TsdfState), vp]),
    "gps_tsdf_ray_wave_rows": (i32, [C.POINTER(TsdfState), vp, i32, vp]),
    "gps_tsdf_find_visible": (i32, [C.POINTER(TsdfState), vp, vp]),
    "gps_tsdf_render_colour": (i32, [C.POINTER(TsdfState), vp]),
    "gps_tsdf_process_frame": (i32, [C.POINTER(TsdfState), vp, vp, vp, vp]),
    "gps_tsdf_free_raycast": (i32, [C.POINTER(TsdfState), vp, vp, vp]),
    "gps_tsdf_view_table_bytes": (i64, [i32]),
    "gps_tsdf_view_init": (i32, [C.POINTER(TsdfState), C.POINTER(TsdfView), vp]),
    "gps_tsdf_free_raycast_batch": (i32, [C.POINTER(TsdfState), i32, C.POINTER(TsdfView), vp, vp]),
    "gps_pose_from_c2w": (i32, [vp, vp, vp]),
    "gps_raycast_to_maps": (i32, [i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_knn_mean_dist2": (i32, [i32, vp, vp, vp]),
    "gps_knn_grid_workspace_bytes": (i64, [i32]),
    "gps_knn_mean_dist2_grid": (i32, [i32, vp, vp, vp, i64, vp]),
    "gps_normal_map": (i32, [i32, i32, vp, vp, vp]),
    "gps_zero_floats": (i32, [i32, vp, vp, vp]),
    "gps_rgba8_to_rgbf": (i32, [i32, vp, vp, vp]),
    "gps_rgba8_to_rgbf_and_floats": (i32, [i32, vp, vp, vp, vp, i32, vp]),
    "gps_prune_mask": (i32, [i32, vp, vp, f32, f32, f32, vp, vp, vp]),
    "gps_gather_rows": (i32, [i32, vp, i32, vp, vp, vp, vp]),
    "gps_init_gaussians": (i32, [i32, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "gps_new_gaussian_mask": (i32, [i32, i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp, vp]),
    "gps_compact_mask_workspace_bytes": (i64, [i32]),
    "gps_compact_mask": (i32, [i32, vp, vp, vp, vp, vp, i64, vp]),
    "gps_gather_pixels": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gps_splat_can_prefetch": (i32, [C.POINTER(SplatStep)]),
    "gps_splat_discard_prefetch": (i32, [C.POINTER(SplatStep), vp]),
    "gps_splat_render": (i32, [C.POINTER(SplatStep), vp]),
    "gps_splat_train_step": (i32, [C.POINTER(SplatStep), i32, vp]),
}


def load_library(path=None):
    """Load libgpsslam_hip.so (building is done by __graft_entry__.build / _build.py, never implicitly here)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GPS_SLAM_HIP_LIB") or _LIBPATH  # env override: A/B builds of the same ABI
    # torch ships its own libamdhip64.so.7; it must be the (single) HIP runtime of the process so that the
    # streams and device pointers torch hands us are valid inside libgpsslam_hip.so -> import torch first.
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise RuntimeError(
            "gps_slam_amd: %s not found. Build it with `python -m gps_slam_amd._build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % p)
    lib_ = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib_, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib_
    return lib_


class _Lazy:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _Lazy()


class GpsError(RuntimeError):
    pass


_ERR = {-1: "GPS_ERR_ARG", -2: "GPS_ERR_LAUNCH", -3: "GPS_ERR_CAPACITY"}


def check(status, what):
    if status != 0:
        raise GpsError("%s failed: %s (%d)" % (what, _ERR.get(status, "?"), status))
