"""Builds the REFERENCE's own fused-SSIM kernels for gfx950 (test infrastructure: the pin of the SSIM oracle and of csrc/splat_ssim.hip).

    /root/reference/gsplat/rasterizer/ssim.cu + ssim.h      -> oracle/_ref/_ref_ssim*.so  (git-ignored, travels to the GPU box)

Of the reference's splat kernels this is the one translation unit that compiles here without anything the image lacks:
ssim.cu includes only its own header, cooperative_groups and libtorch.  Every other rasterizer .cu includes types.cuh ->
<glm/glm.hpp> (glm is neither vendored nor installed) and simple_knn.cu includes the CUDA toolkit's
device_launch_parameters.h: writing stand-ins for those headers is not a reference build, so those stay unbuildable and their
oracles stay "unpinned by the reference" (DESIGN.md section 2).

As in ref_wapper_build.py the only transformation is torch.utils.hipify -- the textual CUDA -> HIP rename of host API names that
torch.utils.cpp_extension applies to every extension source on a ROCm machine -- run on a TEMPORARY copy; nothing of the
reference is written into this repository.  The kernels' arithmetic is untouched: what runs on the MI355X is the reference's
own fusedssimCUDA / fusedssim_backwardCUDA, so `tests/test_reference_ssim_gpu.py` compares HIP == reference kernel == oracle and
`tests/golden/make_ssim_ref_golden.py` stores its outputs as fixtures for the CPU suite.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/gsplat/rasterizer"
OUT = os.path.join(HERE, "_ref")
FILES = ["ssim.cu", "ssim.h"]


def module_path():
    return os.path.join(OUT, "_ref_ssim" + sysconfig.get_config_var("EXT_SUFFIX"))


def stale():
    out = module_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(REF, f) for f in FILES] + [os.path.join(HERE, "ref_ssim_driver.cpp"), __file__]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(verbose=False):
    if not os.path.isdir(REF):
        raise RuntimeError("reference sources not present")
    import torch
    from torch.utils import cpp_extension as ce
    from torch.utils.hipify import hipify_python
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="ref_ssim_")
    try:
        for f in FILES:
            shutil.copyfile(os.path.join(REF, f), os.path.join(tmp, f))
        hipify_python.hipify(project_directory=tmp, output_directory=tmp, includes=[os.path.join(tmp, "*")],
                             extra_files=[os.path.join(tmp, "ssim.cu")], show_detailed=False, is_pytorch_extension=True,
                             hipify_extra_files_only=False)
        src = os.path.join(tmp, "ssim.hip")
        assert os.path.exists(src), os.listdir(tmp)
        tdir = os.path.dirname(torch.__file__)
        inc = ce.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"], tmp]
        common = ["-O3", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                  "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
        common += ["-I" + p for p in inc]
        kobj, dobj = os.path.join(tmp, "ssim.o"), os.path.join(tmp, "driver.o")
        cmds = [["/opt/rocm/bin/hipcc", "--offload-arch=gfx950"] + common + ["-c", src, "-o", kobj],
                ["g++"] + common + ["-include", "hip/hip_runtime_api.h", "-DTORCH_EXTENSION_NAME=_ref_ssim", "-DTORCH_API_INCLUDE_EXTENSION_H",
                                    "-c", os.path.join(HERE, "ref_ssim_driver.cpp"), "-o", dobj]]
        procs = []
        for cmd in cmds:
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for cmd, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(out.decode())
                raise RuntimeError("compile failed: %s" % cmd[-3])
        tlib = os.path.join(tdir, "lib")
        link = ["-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python", "-L/opt/rocm/lib",
                "-lamdhip64", "-Wl,-rpath," + tlib]
        subprocess.check_call(["g++", "-shared", "-o", module_path(), dobj, kobj] + link)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return module_path()


def load():
    """the built module (GPU box: prebuilt file only)"""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location("_ref_ssim", module_path())
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
