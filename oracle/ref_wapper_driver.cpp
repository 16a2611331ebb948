// pybind11 view of the REFERENCE's splat operator wrapper (test infrastructure, built by oracle/ref_wapper_build.py into
// oracle/_ref/_ref_wapper*.so).  This file includes the reference's own gsplat/gsplat_wapper.hpp (from the build's temporary
// copy) and only forwards arguments: every ::apply below runs the reference's autograd code (save_for_backward, the
// launcher calls, the returned gradient lists) on top of this repository's launchers (gps_slam_amd/host/hip_bindings.cpp)
// and HIP kernels.  The GPU tests compare it with the repository's own wrapper mirror (gps_slam_amd._host).
#include <torch/extension.h>

#include REF_WAPPER_HEADER  // the reference's gsplat_wapper.hpp as the build prepared it (ref_wapper_build.py)

namespace py = pybind11;

PYBIND11_MODULE(_ref_wapper, m) {
    m.doc() = "the reference's gsplat_wapper.{hpp,cpp} linked against gps_slam_amd's launchers";
    m.def("SphericalHarmonicsNew", [](int deg, torch::Tensor dirs, torch::Tensor coeffs, torch::Tensor masks) {
        return SphericalHarmonicsNew::apply(deg, dirs, coeffs, masks);
    });
    m.def("FullyFusedProjection", [](torch::Tensor means, torch::Tensor quats, torch::Tensor scales, torch::Tensor viewmats,
                                    torch::Tensor Ks, int w, int h, float eps2d, float near_plane, float far_plane,
                                    float radius_clip) {
        at::optional<torch::Tensor> covars;
        return FullyFusedProjection::apply(means, covars, quats, scales, viewmats, Ks, w, h, eps2d, near_plane, far_plane,
                                           radius_clip, false, std::string("pinhole"));
    });
    m.def("RasterizeToPixelsGes_NewParallel",
          [](torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities, torch::Tensor radiis,
             torch::Tensor ref_depth_map, torch::Tensor base_color_map, int w, int h, int tile_size,
             torch::Tensor isect_offsets, torch::Tensor flatten_ids, torch::Tensor group_gs_ids, torch::Tensor group_starts,
             float delta_depth) {
              at::optional<torch::Tensor> backgrounds, masks;
              return RasterizeToPixelsGes_NewParallel::apply(means2d, conics, colors, opacities, radiis, ref_depth_map,
                                                             base_color_map, backgrounds, masks, w, h, tile_size,
                                                             isect_offsets, flatten_ids, group_gs_ids, group_starts, false,
                                                             delta_depth);
          });
    m.def("RasterizeToPixelsGes",
          [](torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
             torch::Tensor ref_depth_map, torch::Tensor base_color_map, int w, int h, int tile_size,
             torch::Tensor isect_offsets, torch::Tensor flatten_ids, float delta_depth) {
              at::optional<torch::Tensor> backgrounds, masks;
              return RasterizeToPixelsGes::apply(means2d, conics, colors, opacities, ref_depth_map, base_color_map, backgrounds,
                                                 masks, w, h, tile_size, isect_offsets, flatten_ids, false, delta_depth);
          });
    m.def("RasterizeToPixels",
          [](torch::Tensor means2d, torch::Tensor conics, torch::Tensor colors, torch::Tensor opacities,
             at::optional<torch::Tensor> backgrounds, int w, int h, int tile_size, torch::Tensor isect_offsets,
             torch::Tensor flatten_ids, bool absgrad) {
              at::optional<torch::Tensor> masks;
              return RasterizeToPixels::apply(means2d, conics, colors, opacities, backgrounds, masks, w, h, tile_size,
                                              isect_offsets, flatten_ids, absgrad);
          });
    m.def("FusedSSIMMap", [](double C1, double C2, torch::Tensor img1, torch::Tensor img2, std::string padding, bool train) {
        return FusedSSIMMap::apply((float)C1, (float)C2, img1, img2, padding, train);
    });
    m.def("isectTiles", &isectTiles, py::arg("means2d"), py::arg("radii"), py::arg("depths"), py::arg("tile_size"),
          py::arg("tile_width"), py::arg("tile_height"), py::arg("sort") = true);
    m.def("isectOffsetEncode", &isectOffsetEncode);
    m.def("isectTilesNoDepth", &isectTilesNoDepth, py::arg("means2d"), py::arg("radii"), py::arg("depths"),
          py::arg("tile_size"), py::arg("tile_width"), py::arg("tile_height"), py::arg("sort") = true);
    m.def("isectOffsetEncodeNoDepth", &isectOffsetEncodeNoDepth);
    m.def("simpleKNN", &simpleKNN);
    m.def("degFromSh", &degFromSh);
    m.def("numShBases", &numShBases);
    m.def("rgb2sh", &rgb2sh);
    m.def("sh2rgb", &sh2rgb);
    m.def("getDuration", [](double s0, double ns0, double s1, double ns1) {
        struct timespec a, b;
        a.tv_sec = (time_t)s0; a.tv_nsec = (long)ns0; b.tv_sec = (time_t)s1; b.tv_nsec = (long)ns1;
        return getDuration(a, b);
    });
}
