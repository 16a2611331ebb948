"""Builds gps_slam_amd/libgpsslam_hip.so (the C-ABI library) with hipcc for gfx950.

In-tree build so the .so travels with the repo snapshot to the GPU box.  TSDF
kernels are compiled with -ffp-contract=off (bit-exact voxel updates need the
same unfused mul/add sequence as the CPU oracle); splat kernels use the default
contraction.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpsslam_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-DGPS_BUILDING_DLL"]
PER_FILE = {
    # file prefix -> extra flags
    "tsdf_": ["-ffp-contract=off"],
}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "gps_slam_hip.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            extra = [fl for pre, fls in PER_FILE.items() if f.startswith(pre) for fl in fls]
            cmd = [HIPCC] + COMMON + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % f)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
