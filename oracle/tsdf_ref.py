"""Runs oracle/_ref/itm_ref (the reference's own ITMLib CPU engine, see ref_build.sh / ref_driver.cpp)
on a synthetic sequence and parses its dump.  TEST INFRASTRUCTURE ONLY; used to validate the C
restatement (tsdf_oracle.c) and to generate the golden fixtures under tests/golden/."""
import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(_HERE, "_ref", "itm_ref")


def available():
    return os.path.exists(BIN)


BIN_OMP = os.path.join(_HERE, "_ref", "itm_ref_omp")


def _write_input(f, seq, n, voxel, mu, vfmin, vfmax, free_views=(), dump_vba_every=0):
    W, H = seq["W"], seq["H"]
    f.write(struct.pack("<6i8f", 0x47505331, W, H, n, len(free_views), dump_vba_every, seq["fx"], seq["fy"],
                        seq["cx"], seq["cy"], voxel, mu, vfmin, vfmax))
    for k in range(n):
        rgba = np.concatenate([seq["rgb"][k], np.full((H, W, 1), 255, np.uint8)], -1)
        f.write(np.ascontiguousarray(rgba).tobytes())
        f.write(np.ascontiguousarray(seq["depth"][k].astype(np.int16)).tobytes())
        f.write(np.ascontiguousarray(seq["c2w"][k], dtype=np.float32).tobytes())
    for fr, c2w in free_views:
        f.write(struct.pack("<i", fr))
        f.write(np.ascontiguousarray(c2w, dtype=np.float32).tobytes())


def time_reference(seq, n_frames, voxel, mu, vfmin, vfmax, threads=None, openmp=True, track=False):
    """Time the reference's own ProcessFrame loop (TSDF-only `recon` mode) on the first n_frames of seq with the timing
    mode of ref_driver.  openmp=True uses itm_ref_omp (built like upstream: -O3 + OpenMP) with `threads` OpenMP threads.
    -> dict(frames, seconds, threads) or None if the binary is not there."""
    import json
    binary = BIN_OMP if openmp else BIN
    if not os.path.exists(binary):
        return None
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(int(threads))
    with tempfile.TemporaryDirectory() as td:
        fin = os.path.join(td, "in.bin")
        with open(fin, "wb") as f:
            _write_input(f, seq, n_frames, voxel, mu, vfmin, vfmax)
        out = subprocess.check_output([binary, fin, "-", "timetrack" if track else "time"], env=env, timeout=600).decode()
    return json.loads(out.strip().splitlines()[-1])


def run(seq, voxel, mu, vfmin, vfmax, free_views=(), dump_vba_every=0, track=False, mesh=False, save_dir=None):
    """seq: dict from tests.synth.make_sequence; free_views: list of (frame_idx, c2w[4,4]).
    returns {(name, frame): np.ndarray(bytes)} decoded by `decode`."""
    W, H = seq["W"], seq["H"]
    n = seq["rgb"].shape[0]
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<6i8f", 0x47505331, W, H, n, len(free_views), dump_vba_every, seq["fx"], seq["fy"],
                                seq["cx"], seq["cy"], voxel, mu, vfmin, vfmax))
            for k in range(n):
                rgba = np.concatenate([seq["rgb"][k], np.full((H, W, 1), 255, np.uint8)], -1)
                f.write(np.ascontiguousarray(rgba).tobytes())
                f.write(np.ascontiguousarray(seq["depth"][k].astype(np.int16)).tobytes())
                f.write(np.ascontiguousarray(seq["c2w"][k], dtype=np.float32).tobytes())
            for fr, c2w in free_views:
                f.write(struct.pack("<i", fr))
                f.write(np.ascontiguousarray(c2w, dtype=np.float32).tobytes())
        # mesh=True: the CPU meshing engine's triangles of the final scene ("mesh" chunk [T,3,3]); save_dir: additionally the
        # reference's own SaveToDirectory files + WritePLY output (mesh.ply) are written there
        assert not (track and mesh)
        env = dict(os.environ)
        if save_dir is not None:
            env["GPS_REF_SAVE_DIR"] = str(save_dir)
        subprocess.check_call([BIN, fin, fout] + (["track"] if track else ["mesh"] if mesh else []), env=env,
                              stdout=subprocess.DEVNULL)
        raw = open(fout, "rb").read()
    out = {}
    off = 0
    while off < len(raw):
        name = raw[off:off + 32].split(b"\0")[0].decode()
        frame, nbytes = struct.unpack_from("<iq", raw, off + 32)
        off += 44
        out[(name, frame)] = raw[off:off + nbytes]
        off += nbytes
    return decode(out, W, H)


_DT = {"M": np.float32, "invM": np.float32, "counts": np.int32, "visible_ids": np.int32, "vis_type_nz": np.int32,
       "hash": np.int32, "vba_crc": np.uint32, "vba": np.uint8, "depth_f": np.float32, "minmax": np.float32,
       "raycast": np.float32, "icp_points": np.float32, "icp_normals": np.float32, "fv_M": np.float32,
       "fv_invM": np.float32, "fv_counts": np.int32, "fv_visible_ids": np.int32, "fv_minmax": np.float32,
       "fv_raycast": np.float32, "fv_colour": np.uint8, "sizeof": np.int32, "trk_score": np.float32, "mesh": np.float32}


def decode(chunks, W, H):
    res = {}
    for (name, frame), b in chunks.items():
        a = np.frombuffer(b, dtype=_DT[name]).copy()
        if name in ("raycast", "icp_points", "icp_normals", "fv_raycast"):
            a = a.reshape(H, W, 4)
        elif name in ("minmax", "fv_minmax"):
            a = a.reshape(H, W, 2)
        elif name == "depth_f":
            a = a.reshape(H, W)
        elif name == "fv_colour":
            a = a.reshape(H, W, 4)
        elif name == "hash":
            a = a.reshape(-1, 6)
        elif name == "vis_type_nz":
            a = a.reshape(-1, 2)
        elif name == "vba":
            a = a.reshape(-1, 512, 8)
        elif name == "mesh":
            a = a.reshape(-1, 3, 3)
        res[(name, frame)] = a
    return res


# ----------------------------------------------------------------------------------------------
# ctypes front-end of the C restatement (oracle/tsdf_oracle.c)
# ----------------------------------------------------------------------------------------------
import ctypes as C  # noqa: E402

_LIB = None
VOXEL_DT = np.dtype([("sdf", "<i2"), ("w_depth", "u1"), ("clr", "u1", (3,)), ("w_color", "u1"), ("pad", "u1")])
HASH_DT = np.dtype([("pos", "<i2", (3,)), ("pad", "<i2"), ("offset", "<i4"), ("ptr", "<i4")])

# reference capacities (Objects/Scene/ITMVoxelBlockHash.h:18-22)
REF_BLOCKS, REF_BUCKETS, REF_EXCESS = 0x40000, 0x100000, 0x20000


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_tsdf.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, so])
        _LIB = C.CDLL(so)
        _LIB.orc_tsdf_create.restype = C.c_void_p
        _LIB.orc_tsdf_ray_steps.restype = C.c_int64
        _LIB.orc_tsdf_rays.restype = C.c_int64
        for n in ("hash", "vba", "vba_alloc_list", "excess_list", "visible_ids", "visible_type", "minmax", "raycast", "icp_points", "icp_normals",
                  "depth", "fv_visible_ids", "fv_minmax", "fv_raycast", "fv_colour", "trk_diag"):
            getattr(_LIB, "orc_tsdf_" + n).restype = C.c_void_p
    return _LIB


def pose_from_c2w(c2w):
    """SE3Pose::SetInvM(c2w) + Coerce() -> (M, invM) in ORUtils layout (flat 16, m[col*4+row])."""
    c2w = np.ascontiguousarray(c2w, dtype=np.float32)
    M = np.zeros(16, np.float32)
    invM = np.zeros(16, np.float32)
    _lib().orc_pose_from_c2w(c2w.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p),
                             invM.ctypes.data_as(C.c_void_p))
    return M, invM


class TrackCfg(C.Structure):
    """TrackCfg of tsdf_oracle.c"""
    _fields_ = [("n_levels", C.c_int), ("iter_type", C.c_int * 8), ("n_iter", C.c_int * 8), ("space_thresh", C.c_float * 8),
                ("term_thresh", C.c_float), ("tukey_cutoff", C.c_float), ("vf_min", C.c_float), ("vf_max", C.c_float),
                ("frames_to_skip", C.c_int), ("frames_to_weight", C.c_int)]


def track_config(vf_min, vf_max, levels="rrbb", num_iter_coarse=20, num_iter_fine=50, thresh_coarse=0.1, thresh_fine=0.004,
                 term_thresh=1e-4, tukey=8.0, frames_to_skip=20, frames_to_weight=50):
    """The tracker configuration of ITMLibSettings.cpp:54-57 (defaults) -> TrackCfg"""
    c = TrackCfg()
    _lib().orc_track_config(levels.encode(), num_iter_coarse, num_iter_fine, C.c_float(thresh_coarse), C.c_float(thresh_fine),
                            C.c_float(term_thresh), C.c_float(tukey), frames_to_skip, frames_to_weight, C.c_float(vf_min),
                            C.c_float(vf_max), C.byref(c))
    return c


def track_camera(W, H, intr, depth, points, normals, scene_pose_M, pose_M, cfg, frames_processed):
    """orc_track_camera on raw arrays -> (M, invM, diag[16])"""
    depth = np.ascontiguousarray(depth, np.float32)
    points, normals = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(normals, np.float32)
    intr = np.ascontiguousarray(intr, np.float32)
    sp = np.ascontiguousarray(scene_pose_M, np.float32)
    M = np.ascontiguousarray(pose_M, np.float32).copy()
    invM = np.zeros(16, np.float32)
    scratch = np.zeros(W * H, np.float32)
    diag = np.zeros(16, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib().orc_track_camera(W, H, p(intr), p(depth), p(points), p(normals), p(sp), p(M), p(invM), C.byref(cfg),
                            int(frames_processed), p(scratch), p(diag))
    return M, invM, diag


class TsdfOracle:
    def __init__(self, W, H, fx, fy, cx, cy, voxel, mu, vf_min, vf_max, n_blocks=REF_BLOCKS, n_buckets=REF_BUCKETS,
                 n_excess=REF_EXCESS):
        self.W, self.H = W, H
        self.n_blocks, self.n_total, self.n_excess = n_blocks, n_buckets + n_excess, n_excess
        self.h = C.c_void_p(_lib().orc_tsdf_create(W, H, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                                   C.c_float(voxel), C.c_float(mu), C.c_float(vf_min),
                                                   C.c_float(vf_max), n_blocks, n_buckets, n_excess))

    def close(self):
        if self.h:
            _lib().orc_tsdf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, rgb, depth_u16, M, invM):
        H, W = self.H, self.W
        rgba = np.ascontiguousarray(np.concatenate([rgb, np.full((H, W, 1), 255, np.uint8)], -1))
        d = np.ascontiguousarray(depth_u16.astype(np.int16))
        M, invM = np.ascontiguousarray(M, np.float32), np.ascontiguousarray(invM, np.float32)
        _lib().orc_tsdf_process_frame(self.h, rgba.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                      M.ctypes.data_as(C.c_void_p), invM.ctypes.data_as(C.c_void_p))

    def process_frame_tracked(self, rgb, depth_u16, cfg):
        """ProcessFrame with the depth-only ExtendedTracker active (pose estimated); cfg = track_config(...).
        -> (M, invM) of the frame, ORUtils layout."""
        H, W = self.H, self.W
        rgba = np.ascontiguousarray(np.concatenate([rgb, np.full((H, W, 1), 255, np.uint8)], -1))
        d = np.ascontiguousarray(depth_u16.astype(np.int16))
        M, invM = np.zeros(16, np.float32), np.zeros(16, np.float32)
        _lib().orc_tsdf_process_frame_tracked(self.h, rgba.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                              C.byref(cfg), M.ctypes.data_as(C.c_void_p), invM.ctypes.data_as(C.c_void_p))
        return M, invM

    def track_diag(self):
        return self._arr("trk_diag", np.float32, (16,)).copy()

    def free_raycast(self, M, invM):
        M, invM = np.ascontiguousarray(M, np.float32), np.ascontiguousarray(invM, np.float32)
        _lib().orc_tsdf_free_raycast(self.h, M.ctypes.data_as(C.c_void_p), invM.ctypes.data_as(C.c_void_p))

    def _arr(self, name, dtype, shape):
        p = getattr(_lib(), "orc_tsdf_" + name)(self.h)
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return np.frombuffer((C.c_char * n).from_address(p), dtype=dtype).reshape(shape)

    @property
    def n_visible(self):
        return _lib().orc_tsdf_n_visible(self.h)

    def ray_stats(self):
        """{steps, rays}: trips of castRay's loop (Shared.h:158-190) and rays cast since creation (live + free views)"""
        return {"steps": int(_lib().orc_tsdf_ray_steps(self.h)), "rays": int(_lib().orc_tsdf_rays(self.h))}

    @property
    def fv_n_visible(self):
        return _lib().orc_tsdf_fv_n_visible(self.h)

    @property
    def last_free_block(self):
        return _lib().orc_tsdf_last_free_block(self.h)

    @property
    def last_free_excess(self):
        return _lib().orc_tsdf_last_free_excess(self.h)

    def hash(self):
        return self._arr("hash", HASH_DT, (self.n_total,))

    def vba(self):
        return self._arr("vba", VOXEL_DT, (self.n_blocks, 512))

    def vba_alloc_list(self):
        return self._arr("vba_alloc_list", np.int32, (self.n_blocks,))

    def excess_list(self):
        return self._arr("excess_list", np.int32, (self.n_excess,))

    def visible_ids(self):
        return self._arr("visible_ids", np.int32, (self.n_blocks,))[:self.n_visible]

    def visible_type(self):
        return self._arr("visible_type", np.uint8, (self.n_total,))

    def image(self, name):
        H, W = self.H, self.W
        spec = {"minmax": (np.float32, (H, W, 2)), "raycast": (np.float32, (H, W, 4)),
                "icp_points": (np.float32, (H, W, 4)), "icp_normals": (np.float32, (H, W, 4)),
                "depth": (np.float32, (H, W)), "fv_minmax": (np.float32, (H, W, 2)),
                "fv_raycast": (np.float32, (H, W, 4)), "fv_colour": (np.uint8, (H, W, 4))}[name]
        return self._arr(name, *spec)

    def fv_visible_ids(self):
        return self._arr("fv_visible_ids", np.int32, (self.n_blocks,))[:self.fv_n_visible]

    def mesh(self, max_triangles=1 << 22):
        """ITMMeshingEngine::MeshScene of the current scene -> float32 [T, 7, 3]: p0 p1 p2 c0 c1 c2 clr (ITMMesh::Triangle)"""
        buf = np.zeros((max_triangles, 21), np.float32)
        lib = _lib()
        lib.orc_tsdf_mesh.restype = C.c_int64
        n = lib.orc_tsdf_mesh(self.h, C.c_int64(max_triangles), buf.ctypes.data_as(C.c_void_p))
        return buf[:n].reshape(n, 7, 3).copy()

    # canonical views used by every comparison
    def hash_rows(self):
        """rows (idx, x, y, z, offset, ptr) of every non-default entry, like ref_driver's "hash" chunk"""
        h = self.hash()
        keep = ~((h["ptr"] == -2) & (h["offset"] == 0) & (h["pos"] == 0).all(1))
        idx = np.nonzero(keep)[0]
        e = h[idx]
        return np.concatenate([idx[:, None], e["pos"].astype(np.int64), e["offset"][:, None], e["ptr"][:, None]],
                              1).astype(np.int32)

    def allocated_blocks(self):
        """voxel payload [n,512,7 bytes] of allocated entries in hash-index order (pad byte dropped)"""
        h = self.hash()
        ptr = h["ptr"][h["ptr"] >= 0]
        v = self.vba()[ptr]
        return voxels_canonical(v)


def voxels_canonical(v):
    """structured voxels [...,512] -> uint8 [...,512,7] (sdf lo, sdf hi, w_depth, r, g, b, w_color)"""
    raw = v.view(np.uint8).reshape(v.shape + (8,))
    return raw[..., :7]
