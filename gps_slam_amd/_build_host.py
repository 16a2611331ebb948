"""Builds the C++ host layer gps_slam_amd/_host*.so (libtorch + pybind11) with g++, linked against the in-tree
libgpsslam_hip.so.  In-tree build (the .so travels with the snapshot); objects are cached under gps_slam_amd/build/."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "host")
CXX = os.environ.get("CXX", "g++")
SRCS = ["hip_bindings.cpp", "gsplat_wapper.cpp", "raw_gs_param.cpp", "raw_gs_model.cpp", "tsdf_engine.cpp", "infinitam_tools.cpp", "slam_pipeline.cpp", "bindings.cpp"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def target_path():
    return os.path.join(HERE, "_host" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(verbose=False, force=False):
    import torch
    from torch.utils import cpp_extension as ce
    tdir = os.path.dirname(torch.__file__)
    inc = ce.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"], os.path.join(os.path.dirname(HERE), "include"), HOST]
    flags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_host",
             "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
             "-Wno-deprecated-declarations"] + ["-I" + p for p in inc]
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "gps_slam_hip.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for f in SRCS:
        src, obj = os.path.join(HOST, f), os.path.join(objdir, "host_" + f.replace(".cpp", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [CXX] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("%s failed on %s" % (CXX, f))
    tgt = target_path()
    if force or procs or _stale(tgt, objs):
        lib = os.path.join(tdir, "lib")
        cmd = [CXX, "-shared", "-o", tgt] + objs + ["-L" + lib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
                                                    "-ltorch_python", "-L" + HERE, "-lgpsslam_hip",
                                                    "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return tgt


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv, force="-f" in sys.argv))
