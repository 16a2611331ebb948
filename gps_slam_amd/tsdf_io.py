"""On-disk formats of the TSDF side (host code, no device work): the mesh PLY and the scene dumps the reference writes.

  ITMMesh::WritePLY                      Objects/Meshing/ITMMesh.h:39-106
  ITMScene::SaveToDirectory / Load       Objects/Scene/ITMScene.h:34-44
  ITMLocalVBA::SaveToDirectory / Load    Objects/Scene/ITMLocalVBA.h:36-66      voxel.dat, alloc.dat, vba.txt
  ITMVoxelBlockHash::SaveToDirectory     Objects/Scene/ITMVoxelBlockHash.h:129-155  hash.dat, excess.dat, last.txt
  MemoryBlockPersister::WriteBlock       ORUtils/MemoryBlockPersister.h:350-364  size_t element count + raw elements
"""
import os

import numpy as np

VOXEL_BYTES, HASH_BYTES = 8, 16


def write_mesh_ply(file_name, tris):
    """tris float32 [T, >=6, 3]: p0 p1 p2 c0 c1 c2 (ITMMesh::Triangle).  Ascii PLY with 3 vertices per triangle."""
    tris = np.asarray(tris, np.float32)
    n = tris.shape[0]
    col = (tris[:, 3:6] * np.float32(255)).astype(np.uint8)  # static_cast<unsigned char>(c * 255): truncation
    with open(file_name, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face %d\n"
                "property list uchar int vertex_indices\nend_header\n" % (n * 3, n))
        pos = tris[:, :3].reshape(n * 3, 3).tolist()
        c = col.reshape(n * 3, 3).tolist()
        f.write("".join("%f %f %f %d %d %d\n" % (p[0], p[1], p[2], q[0], q[1], q[2]) for p, q in zip(pos, c)))
        f.write("".join("3 %d %d %d\n" % (3 * i, 3 * i + 1, 3 * i + 2) for i in range(n)))


def _write_block(path, arr, elem_bytes):
    a = np.ascontiguousarray(arr)
    with open(path, "wb") as f:
        f.write(np.uint64(a.nbytes // elem_bytes).tobytes())
        a.tofile(f)


def _read_block(path, nbytes, elem_bytes):
    with open(path, "rb") as f:
        n = int(np.frombuffer(f.read(8), np.uint64)[0])
        if n * elem_bytes != nbytes:
            raise RuntimeError("Could not read data into a memory block of the wrong size: " + path)
        return np.fromfile(f, dtype=np.uint8, count=nbytes)


def save_scene(scene_dir, vba, alloc_list, last_free_block, hash_table, excess_list, last_free_excess):
    """vba: bytes of [n_blocks*512] voxels; alloc_list int32[n_blocks]; hash_table: bytes of [n_total] entries;
    excess_list int32[n_excess].  scene_dir is created."""
    d = scene_dir if scene_dir.endswith("/") else scene_dir + "/"
    os.makedirs(d, exist_ok=True)
    vba = np.ascontiguousarray(vba).view(np.uint8).reshape(-1)
    _write_block(d + "voxel.dat", vba, VOXEL_BYTES)
    _write_block(d + "alloc.dat", np.asarray(alloc_list, np.int32), 4)
    with open(d + "vba.txt", "w") as f:
        f.write("%d %d" % (int(last_free_block), vba.size // VOXEL_BYTES))  # lastFreeBlockId, allocatedSize
    _write_block(d + "hash.dat", np.ascontiguousarray(hash_table).view(np.uint8).reshape(-1), HASH_BYTES)
    _write_block(d + "excess.dat", np.asarray(excess_list, np.int32), 4)
    with open(d + "last.txt", "w") as f:
        f.write("%d" % int(last_free_excess))


def load_scene(scene_dir, n_blocks, n_total, n_excess):
    """-> dict(vba u8, alloc_list i32, last_free_block, hash u8, excess_list i32, last_free_excess); sizes must match."""
    d = scene_dir if scene_dir.endswith("/") else scene_dir + "/"
    out = {}
    out["vba"] = _read_block(d + "voxel.dat", n_blocks * 512 * VOXEL_BYTES, VOXEL_BYTES)
    out["alloc_list"] = _read_block(d + "alloc.dat", n_blocks * 4, 4).view(np.int32)
    out["last_free_block"] = int(open(d + "vba.txt").read().split()[0])
    out["hash"] = _read_block(d + "hash.dat", n_total * HASH_BYTES, HASH_BYTES)
    out["excess_list"] = _read_block(d + "excess.dat", n_excess * 4, 4).view(np.int32)
    out["last_free_excess"] = int(open(d + "last.txt").read().split()[0])
    return out
