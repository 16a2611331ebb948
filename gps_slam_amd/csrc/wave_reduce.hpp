// wave64 sum of several per-lane values at once (gfx950).
// v_permlane32_swap / v_permlane16_swap exchange half-waves / 16-lane rows between TWO registers in one instruction, so
// "swap, add" halves the lane span of two values at once: 4 values -> 2 registers (one value per half-wave) -> 1 register (one
// value per 16-lane row) in 3 swaps + 3 adds; the remaining sum inside each row is 4 DPP row_shr adds.  10 VALU ops per 4
// values instead of 24 for four full DPP reductions (or 24 ds_bpermute for shuffles), totals in lane 15 of each row.
#pragma once
#include <hip/hip_runtime.h>

namespace gps {

__device__ __forceinline__ float sum_halves(float a, float b) {  // lanes 0-31: a summed over halves; lanes 32-63: b
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_row_pairs(float a, float b) {  // rows: {a01, b01, a23, b23}
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add_raw(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v = dpp_add_raw<0x111>(v);  // row_shr:1
    v = dpp_add_raw<0x112>(v);  // row_shr:2
    v = dpp_add_raw<0x114>(v);  // row_shr:4
    v = dpp_add_raw<0x118>(v);  // row_shr:8
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_masked(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
    return v + __int_as_float(moved);
}
// values (a, b, c, d) -> one register whose rows 0..3 hold the wave totals of (a, c, b, d) in lane 15 of the row
__device__ __forceinline__ float reduce4(float a, float b, float c, float d) {
    return row_sum_to_lane15(sum_row_pairs(sum_halves(a, b), sum_halves(c, d)));
}

}  // namespace gps
