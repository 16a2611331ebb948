"""Golden 3DGS PLY for RawGaussianParams::savePly (src/raw_gs_param.cpp:159-217): tests/golden/gaussian_ply_n2_k16.ply + the
parameter tensors it was written from (gaussian_ply_n2_k16.npz).

The reference's writer cannot be built here (raw_gs_param.h pulls in nvml.h, Eigen and yaml-cpp through file_utils.h), so the
file is DATA derived from reading its writer, independent of every writer in this repository (nothing of gps_slam_amd is
imported): the header lines in the order :164-200 emits them ("ply", "format binary_little_endian 1.0", "element vertex N", the
62 "property float ..." lines, "end_header", each ended by std::endl = one '\\n'), then per point the 62 little-endian floats
in the order :213-221 writes them: means[3], three zeros (nx ny nz), featuresDc[3], featuresRest transposed to channel-major
and flattened ([N,15,3] -> transpose(1,2) -> [N,45]: f_rest_0..14 = red of SH 1..15, then green, then blue), opacities[1],
scales[3], quats[4].  Values are small distinct dyadic numbers so that a swapped field or a wrong transpose changes bytes.

    python tests/golden/make_gaussian_ply_golden.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PROPS = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] +
         ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])


def tensors(n=2, k=16):
    v = lambda *shape, base: (base + np.arange(int(np.prod(shape)), dtype=np.float32).reshape(shape) / 64.0).astype(np.float32)
    return dict(means=v(n, 3, base=1.0), scales=v(n, 3, base=-5.0), quats=v(n, 4, base=0.25), featuresDc=v(n, 3, base=2.0),
                featuresRest=v(n, k - 1, 3, base=-3.0), opacities=v(n, 1, base=8.0))


def main():
    t = tensors()
    n = t["means"].shape[0]
    out = bytearray()
    for line in ["ply", "format binary_little_endian 1.0", "element vertex %d" % n] + ["property float " + p for p in PROPS] + ["end_header"]:
        out += line.encode("ascii") + b"\n"
    for i in range(n):
        row = list(t["means"][i]) + [0.0, 0.0, 0.0] + list(t["featuresDc"][i])
        for c in range(3):                       # channel-major: all 15 coefficients of red, then green, then blue
            row += [t["featuresRest"][i, j, c] for j in range(15)]
        row += list(t["opacities"][i]) + list(t["scales"][i]) + list(t["quats"][i])
        assert len(row) == len(PROPS) == 62
        out += struct.pack("<62f", *[float(x) for x in row])
    open(os.path.join(HERE, "gaussian_ply_n2_k16.ply"), "wb").write(bytes(out))
    np.savez(os.path.join(HERE, "gaussian_ply_n2_k16.npz"), **t)
    print(len(out), "bytes")


if __name__ == "__main__":
    main()
