"""Digests of a TSDF engine state, the same for the reference engine's dump (oracle/tsdf_ref.run), the CPU restatement
(oracle.tsdf_ref.TsdfOracle) and the HIP engine (tests.test_tsdf_gpu.EngineView): CRC-32 per array with NaNs canonicalised
(every NaN -> one quiet-NaN pattern, as bits_equal treats them) and the min/max image restricted to the window the raycaster
consumes.  Used by the full-size golden fixture (tests/golden/make_tsdf_fullsize_golden.py)."""
import zlib

import numpy as np


def crc(a):
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        u = a.view(np.uint32).copy()
        u[np.isnan(a)] = 0x7FC00000
        a = u
    return np.uint32(zlib.crc32(a.tobytes()) & 0xFFFFFFFF)


def minmax_window(img, W, H):
    return img[:(H + 7) // 8, :(W + 7) // 8]


def frame_digest(get, f, W, H):
    """get(name, frame) -> array, names as ref_driver dumps them"""
    return {"counts": np.asarray(get("counts", f), np.int32), "visible_ids": crc(get("visible_ids", f)),
            "hash": crc(get("hash", f)), "vis_type_nz": crc(get("vis_type_nz", f)), "depth_f": crc(get("depth_f", f)),
            "minmax_window": crc(minmax_window(get("minmax", f), W, H)), "raycast": crc(get("raycast", f)),
            "icp_points": crc(get("icp_points", f)), "icp_normals": crc(get("icp_normals", f)),
            "vba_crc": np.uint32(np.asarray(get("vba_crc", f)).reshape(-1)[0])}


def free_view_digest(get, tag, W, H):
    return {"fv_counts": np.asarray(get("fv_counts", tag), np.int32).reshape(-1)[:1], "fv_visible_ids": crc(get("fv_visible_ids", tag)),
            "fv_minmax_window": crc(minmax_window(get("fv_minmax", tag), W, H)), "fv_raycast": crc(get("fv_raycast", tag)),
            "fv_colour": crc(get("fv_colour", tag))}


def engine_getter(o):
    """the accessor interface of TsdfOracle / EngineView as a get(name, frame) function over the CURRENT state"""
    def get(name, _f):
        if name == "counts":
            return np.array([o.n_visible, o.last_free_block, o.last_free_excess], np.int32)
        if name == "visible_ids":
            return np.asarray(o.visible_ids(), np.int32)
        if name == "hash":
            return o.hash_rows()
        if name == "vis_type_nz":
            vt = np.asarray(o.visible_type())
            nz = np.nonzero(vt)[0]
            return np.stack([nz, vt[nz]], 1).astype(np.int32)
        if name == "depth_f":
            return o.image("depth")
        if name in ("minmax", "raycast", "icp_points", "icp_normals", "fv_minmax", "fv_raycast", "fv_colour"):
            return o.image(name)
        if name == "vba_crc":
            return np.array([zlib.crc32(np.ascontiguousarray(o.allocated_blocks()).tobytes()) & 0xFFFFFFFF], np.uint32)
        if name == "fv_counts":
            return np.array([o.fv_n_visible], np.int32)
        if name == "fv_visible_ids":
            return np.asarray(o.fv_visible_ids(), np.int32)
        raise KeyError(name)
    return get
