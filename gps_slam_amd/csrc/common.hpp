// Shared host/device helpers for the gfx950 kernels (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gps_slam_hip.h"

#define GPS_WAVE 64

// Every C-ABI entry point returns GPS_OK or a negative error; launches are checked
// with hipGetLastError so a bad configuration is reported, never silently skipped.
#define GPS_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            fprintf(stderr, "[gps_slam_hip] %s:%d launch failed: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return GPS_ERR_LAUNCH;                           \
        }                                                    \
    } while (0)

// hipGetLastError() is per-thread and shared with every other HIP user in the process (e.g. torch);
// clear whatever was left behind so GPS_LAUNCH_CHECK only reports our own launches.
#define GPS_ENTER() (void)hipGetLastError()

#define GPS_REQUIRE(cond)                 \
    do {                                  \
        if (!(cond)) return GPS_ERR_ARG;  \
    } while (0)

static inline int gps_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- build-time probe switches (gps_build_flags) ----
// The kernels keep a few compile-time tunables (-DGPS_...=n) for A/B builds (tools/probe/variant.py).  Every translation unit
// reports the ones it was compiled with: a tunable whose value differs from the shipped default, or a defined probe switch,
// lands in the string gps_build_flags() returns.  The shipped library reports "" -- tests/test_abi_cpu.py and
// __graft_entry__.smoke() assert it.
namespace gps {
void report_build_flag(const char* name, long value);   // splat_project.hip
struct BuildFlag {
    BuildFlag(const char* name, long value, long shipped) { if (value != shipped) report_build_flag(name, value); }
};
}  // namespace gps
#define GPS_TUNABLE_REPORT(name, shipped) static const gps::BuildFlag gps_build_flag_##name(#name, (long)(name), (long)(shipped))
#define GPS_SWITCH_REPORT(name) static const gps::BuildFlag gps_build_flag_##name(#name, 1L, 0L)

// ---- wave64 reductions (DPP-free, shuffle based; all 64 lanes active) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan_i(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}
