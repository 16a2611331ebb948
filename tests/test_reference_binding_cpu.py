"""Launcher-level drop-in proof: the reference's OWN gsplat/gsplat_wapper.{hpp,cpp} -- the autograd
Functions raw_gs_model.cpp programs against -- builds and links against this repository's definitions of the
`gsplat::*_tensor` launchers (gps_slam_amd/host/hip_bindings.cpp over the C-ABI) with NO unresolved symbol.

The link is the check: C++ mangling encodes namespace, name and every parameter type, so `-Wl,--no-undefined` succeeding
means each launcher the wrapper calls is defined here with exactly the signature rasterizer/bindings.h / ssim.h /
simple_knn.h declare.  oracle/ref_wapper_build.py explains the one build step applied to a temporary copy of the five
reference files (torch's own hipify rename of c10/cuda -> c10/hip: <cuda_runtime.h> does not exist on ROCm) -- nothing of
the reference enters this repository.  Skipped where /root/reference is absent (the GPU box uses the prebuilt library).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/gsplat")

LAUNCHERS = ["gsplat::compute_sh_fwd_tensor", "gsplat::compute_sh_bwd_tensor", "gsplat::fully_fused_projection_fwd_tensor",
             "gsplat::fully_fused_projection_bwd_tensor", "gsplat::isect_tiles_tensor(", "gsplat::isect_offset_encode_tensor(",
             "gsplat::isect_tiles_tensor_no_depth", "gsplat::isect_offset_encode_tensor_no_depth",
             "gsplat::rasterize_to_pixels_fwd_tensor", "gsplat::rasterize_to_pixels_bwd_tensor",
             "gsplat::rasterize_to_pixels_fwd_ges_tensor", "gsplat::rasterize_to_pixels_bwd_ges_tensor",
             "gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor", "distCUDA2(", "fusedssim(", "fusedssim_backward("]


def _nm(path, *flags):
    return subprocess.run(["nm", "-C", "-D"] + list(flags) + [path], check=True, capture_output=True, text=True).stdout


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference sources (this container only)")
def test_reference_wrapper_links_against_hip_bindings_with_no_unresolved_symbol():
    from oracle import ref_wapper_build as rb
    lib, mod = rb.build()  # raises if g++ -Wl,--no-undefined leaves anything unresolved
    assert os.path.exists(lib) and os.path.exists(mod)
    defined = _nm(lib, "--defined-only")
    undefined = _nm(lib, "--undefined-only")
    # what the reference's objects reference (gsplat_wapper.cpp + the Function::apply instantiations): all 16 launchers ...
    needs = open(rb.symbols_path()).read()
    for name in LAUNCHERS:
        assert name in needs, "the reference's wrapper does not reference " + name + "?"
    # ... and each of those exact (demangled) signatures is defined by hip_bindings.cpp
    for sig in needs.splitlines():
        assert sig in defined, "signature mismatch / missing launcher: " + sig
        assert sig not in undefined
    # the reference's free functions are in the library (its code, compiled from its sources)
    for name in ("isectTilesNoDepth(", "isectOffsetEncodeNoDepth(", "isectTiles(", "simpleKNN(", "getDuration(", "rgb2sh("):
        assert name in defined, name
    # and the kernels come from the C-ABI library
    needed = subprocess.run(["readelf", "-d", lib], check=True, capture_output=True, text=True).stdout
    assert "libgpsslam_hip.so" in needed


def test_hip_bindings_declares_the_reference_signatures():
    """Token-level comparison of the declarations in host/hip_bindings.hpp with rasterizer/bindings.h (when present), so a
    drift shows up even before the link test: same parameter type list for every launcher the wrapper calls."""
    import re
    ours = open(os.path.join(ROOT, "gps_slam_amd", "host", "hip_bindings.hpp")).read()

    def params(text, name):
        m = re.search(r"\b" + re.escape(name) + r"\s*\(", text)
        assert m, name
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        body = re.sub(r"//[^\n]*", "", text[m.end():i - 1])
        out = []
        for p in body.split(","):
            toks = re.sub(r"\s+", " ", p.replace("&", " & ")).strip().split(" ")
            out.append(" ".join(toks[:-1]))  # drop the parameter name
        return out

    for name in ("fully_fused_projection_fwd_tensor", "rasterize_to_pixels_fwd_ges_tensor", "compute_sh_bwd_tensor"):
        assert len(params(ours, name)) > 3
    if not HAVE_REF:
        pytest.skip("reference headers not present: declaration comparison runs in the build container only")
    ref = open("/root/reference/gsplat/rasterizer/bindings.h").read()
    for name in [n.split("::")[1].rstrip("(") for n in LAUNCHERS if n.startswith("gsplat::")]:
        assert params(ours, name) == params(ref, name), name
