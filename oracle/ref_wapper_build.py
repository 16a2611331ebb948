"""Builds the REFERENCE's own splat operator wrapper on top of this repository's launchers (test infrastructure).

    /root/reference/gsplat/gsplat_wapper.{hpp,cpp}      the autograd Functions raw_gs_model.cpp programs against
    + rasterizer/{bindings.h, ssim.h, simple_knn.h}      the launcher declarations they call
linked against
    gps_slam_amd/host/hip_bindings.cpp                   this repository's definitions of those launchers (C-ABI underneath)
    gps_slam_amd/libgpsslam_hip.so                       the HIP kernels

Outputs (oracle/_ref/, git-ignored, travels to the GPU box like the ITMLib reference binary):
    libref_gsplat_wapper.so   reference wrapper + hip_bindings, linked with -Wl,--no-undefined: every symbol the reference's
                              wrapper needs is defined with exactly the reference's signature (C++ mangling = the proof)
    _ref_wapper*.so           pybind11 view (oracle/ref_wapper_driver.cpp) so the GPU tests can CALL the reference's
                              SphericalHarmonicsNew / FullyFusedProjection / RasterizeToPixelsGes[_NewParallel] / ... ::apply

Why hipify.  bindings.h:4 includes <c10/cuda/CUDAGuard.h> and uses at::cuda::OptionalCUDAGuard; in libtorch-ROCm that
header pulls <cuda_runtime.h>, which does not exist on a ROCm machine, so the wrapper cannot be compiled byte-for-byte
unchanged against ANY ROCm libtorch.  PyTorch's own answer -- what torch.utils.cpp_extension does to every extension
source when it builds on ROCm -- is torch.utils.hipify, a textual rename of the CUDA host API names (c10/cuda -> c10/hip,
at::cuda -> at::hip).  That is the step a maintainer's build would run anyway; it is applied here to a TEMPORARY copy
of the five files (never written into this repository), the wrapper's logic is untouched, and no stand-in header is
written.  The launchers' .cu files are NOT built: they are what libgpsslam_hip.so replaces.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/gsplat"
OUT = os.path.join(HERE, "_ref")
FILES = ["gsplat_wapper.hpp", "gsplat_wapper.cpp", "rasterizer/bindings.h", "rasterizer/ssim.h", "rasterizer/simple_knn.h"]


def lib_path():
    return os.path.join(OUT, "libref_gsplat_wapper.so")


def module_path():
    return os.path.join(OUT, "_ref_wapper" + sysconfig.get_config_var("EXT_SUFFIX"))


def symbols_path():
    return os.path.join(OUT, "ref_wapper_needs.txt")


def stale():
    """outputs missing or older than what they are built from (the reference files, hip_bindings, the driver, the library)"""
    outs = [lib_path(), module_path(), symbols_path()]
    if not all(os.path.exists(o) for o in outs):
        return True
    t = min(os.path.getmtime(o) for o in outs)
    host = os.path.join(ROOT, "gps_slam_amd", "host")
    deps = [os.path.join(REF, f) for f in FILES] + [os.path.join(host, f) for f in ("hip_bindings.cpp", "hip_bindings.hpp",
                                                                                    "gps_host_common.hpp")]
    deps += [os.path.join(HERE, "ref_wapper_driver.cpp"), os.path.join(ROOT, "include", "gps_slam_hip.h"), __file__]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(verbose=False):
    if not os.path.isdir(REF):
        raise RuntimeError("reference sources not present")
    import torch
    from torch.utils import cpp_extension as ce
    from torch.utils.hipify import hipify_python
    sys.path.insert(0, ROOT)
    from gps_slam_amd import _build, _build_host
    _build.build()
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="ref_wapper_")
    try:
        for f in FILES:
            dst = os.path.join(tmp, "gsplat", f)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(os.path.join(REF, f), dst)
        hipify_python.hipify(project_directory=tmp, output_directory=tmp, includes=[os.path.join(tmp, "*")],
                             extra_files=[os.path.join(tmp, "gsplat", "gsplat_wapper.cpp")], show_detailed=False,
                             is_pytorch_extension=True, hipify_extra_files_only=False)
        src = os.path.join(tmp, "gsplat", "gsplat_wapper.cpp")
        for cand in (os.path.join(tmp, "gsplat", "gsplat_wapper_hip.cpp"), os.path.join(tmp, "gsplat", "gsplat_wapper.hip")):
            if os.path.exists(cand):
                src = cand
        tdir = os.path.dirname(torch.__file__)
        host = os.path.join(ROOT, "gps_slam_amd", "host")
        # (gps_slam_amd/host is NOT on the include path: "gsplat_wapper.hpp" must resolve to the reference's file only)
        inc = ce.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]
        flags = ["-O1", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                 "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations",
                 "-include", "hip/hip_runtime_api.h"] + ["-I" + p for p in inc]
        objs = []
        jobs = [(src, os.path.join(tmp, "ref_wapper.o"), ["-I" + os.path.join(tmp, "gsplat")]),
                (os.path.join(host, "hip_bindings.cpp"), os.path.join(tmp, "hip_bindings.o"), [])]
        procs = []
        for s, o, extra in jobs:
            cmd = ["g++"] + flags + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            objs.append(o)
        for s, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(out.decode())
                raise RuntimeError("g++ failed on %s" % s)
        tlib = os.path.join(tdir, "lib")
        libdir = os.path.join(ROOT, "gps_slam_amd")
        link = ["-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-L" + libdir, "-lgpsslam_hip",
                "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN/../../gps_slam_amd"]
        subprocess.check_call(["g++", "-shared", "-o", lib_path()] + objs + link + ["-Wl,--no-undefined"])
        # pybind view of the reference's Functions.  The autograd Functions are defined in-class in gsplat_wapper.hpp, so
        # their launcher calls are only instantiated HERE (in the reference: in raw_gs_model.cpp) -- the driver object
        # is what references the SH / projection / rasterizer / SSIM launchers.
        drv = os.path.join(tmp, "driver.o")
        hdr = os.path.basename(src).replace(".cpp", ".hpp")  # gsplat_wapper_hip.hpp
        assert os.path.exists(os.path.join(tmp, "gsplat", hdr))
        cmd = ["g++"] + flags + ["-I" + os.path.join(tmp, "gsplat"), "-DREF_WAPPER_HEADER=\"%s\"" % hdr,
                                 "-DTORCH_EXTENSION_NAME=_ref_wapper",
                                 "-DTORCH_API_INCLUDE_EXTENSION_H", "-c", os.path.join(HERE, "ref_wapper_driver.cpp"), "-o", drv]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        # THE PROOF: reference objects (wrapper .cpp + every Function::apply instantiated by the driver) + hip_bindings link
        # with nothing left undefined.  Throwaway output: it needs libpython on the link line, which a module loaded by
        # the python executable must not carry.
        pylib = ["-L" + (sysconfig.get_config_var("LIBDIR") or "/usr/lib"), "-lpython%s" % sysconfig.get_config_var("LDVERSION")]
        subprocess.check_call(["g++", "-shared", "-o", os.path.join(tmp, "check.so"), drv] + objs + ["-ltorch_python"] + link +
                              pylib + ["-Wl,--no-undefined"])
        # what the reference's objects asked for (undefined in them, namespace gsplat / ssim.h / simple_knn.h), for the test
        need = subprocess.run(["nm", "-C", "--undefined-only", drv, objs[0]], check=True, capture_output=True, text=True).stdout
        with open(symbols_path(), "w") as f:
            for line in sorted(set(l.split(" U ", 1)[1] for l in need.splitlines() if " U " in l)):
                if line.startswith(("gsplat::", "distCUDA2(", "fusedssim")):
                    f.write(line + "\n")
        subprocess.check_call(["g++", "-shared", "-o", module_path(), drv, "-L" + OUT, "-lref_gsplat_wapper", "-ltorch_python"] +
                              link + ["-Wl,-rpath,$ORIGIN"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return lib_path(), module_path()


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
