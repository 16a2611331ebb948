"""TEST INFRASTRUCTURE: builds oracle/_libtorch_adam*.so -- libtorch's own torch::optim::Adam behind a tiny pybind11 class
(oracle/libtorch_adam.cpp), the optimiser the reference instantiates (src/raw_gs_model.cpp:654-674).  In-tree so that it
travels to the GPU box; g++ only (the optimiser's kernels are ATen's, already in libtorch)."""
import os
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "libtorch_adam.cpp")


def target_path():
    return os.path.join(HERE, "_libtorch_adam" + sysconfig.get_config_var("EXT_SUFFIX"))


def stale():
    t = target_path()
    return not os.path.exists(t) or os.path.getmtime(SRC) > os.path.getmtime(t)


def build():
    import torch
    from torch.utils import cpp_extension as ce
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"]]
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_libtorch_adam", "-DTORCH_API_INCLUDE_EXTENSION_H",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
           + ["-I" + p for p in inc] + [SRC, "-o", target_path(), "-L" + lib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
                                        "-Wl,-rpath," + lib])
    subprocess.check_call(cmd)
    return target_path()


def load():
    """the module (built on demand where a compiler is present; the GPU box uses the prebuilt file)"""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    if stale():
        build()
    spec = importlib.util.spec_from_file_location("_libtorch_adam", target_path())
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build())
