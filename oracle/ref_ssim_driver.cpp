// pybind11 view of the REFERENCE's fused-SSIM kernels (test infrastructure, built by oracle/ref_ssim_build.py into
// oracle/_ref/_ref_ssim*.so).  Includes the reference's own ssim.h from the build's temporary copy and only forwards arguments:
// fusedssim / fusedssim_backward below are the reference's launch functions over the reference's kernels, compiled for gfx950.
#include <torch/extension.h>

#include "ssim.h"

PYBIND11_MODULE(_ref_ssim, m) {
    m.doc() = "the reference's gsplat/rasterizer/ssim.cu compiled for gfx950";
    m.def("fusedssim", [](double C1, double C2, torch::Tensor img1, torch::Tensor img2, bool train) {
        return fusedssim((float)C1, (float)C2, img1, img2, train);
    });
    m.def("fusedssim_backward", [](double C1, double C2, torch::Tensor img1, torch::Tensor img2, torch::Tensor dL_dmap,
                                   torch::Tensor dm_dmu1, torch::Tensor dm_dsigma1_sq, torch::Tensor dm_dsigma12) {
        return fusedssim_backward((float)C1, (float)C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
    });
}
