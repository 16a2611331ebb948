// In-loop launch timing (gps_launch_timing_*): while it is on, every wave of an instrumented kernel stamps the device's 100 MHz
// wall clock when it starts and when it ends into its workgroup's {first start, last end} slot (atomic min / max; the host
// reduces a launch's slots after the window: last end - first start): the kernel's own execution interval inside the running schedule, what a
// rocprofv3 kernel trace reports minus the dispatch's ramp-up and the end-of-kernel release (~1-2 us each; compared launch by
// launch in profiles/r06_bench_kernel_stats.md).  This is what bench.py's `roofline` prices -- the same kernel timed alone in
// a back-to-back loop is faster: nothing collides with it and its inputs are warm.  Off (a NULL slot: one scalar test per
// wave) unless a caller turned it on; results never depend on it.
// Tried first and dropped: HIP events around the launch (hipEventRecord: +16 us on a 79 us kernel -- two barrier packets with
// system-scope releases) and hipExtLaunchKernelGGL's start / stop events (+4 us in a one-stream schedule; with a second stream
// busy the start marker is stamped long before the kernel's waves run: a 145 us raycast read 230 us).
#pragma once
#include <hip/hip_runtime.h>

namespace gps {

enum TimedKernel {   // == GPS_TIMED_* of include/gps_slam_hip.h
    TK_PREPROCESS_BWD = 0, TK_PREPROCESS_FWD, TK_RASTER_FWD, TK_RASTER_BWD_STRIPS, TK_SB_SCAN, TK_SB_SCATTER, TK_INTEGRATE, TK_RAYCAST,
    TK_COUNT
};

struct LaunchStamp { unsigned long long* slots; };   // {first start, last end} per WORKGROUP of one launch, or NULL

bool launch_timing_on();
// the slots of one launch of `kind` with n_workgroups workgroups (NULL: off, or the window's slots are used up -- counted as dropped)
LaunchStamp launch_timing_slots(int kind, int flag, size_t n_workgroups);

#if defined(__HIPCC__)
// first statement of an instrumented kernel: `gps::StampScope timed(stamp);` -- the destructor stamps every exit path.
// One slot pair per workgroup, touched by that workgroup's <= 16 waves only: a single pair per launch (every wave of the grid
// on two addresses) was measured first and serialised the kernels behind their own atomics (64 -> 228 us for 9,600 waves).
struct StampScope {
    unsigned long long* s;
    __device__ __forceinline__ explicit StampScope(LaunchStamp st) : s(st.slots) {
        if (s) {
            s += 2 * ((size_t)blockIdx.x + (size_t)gridDim.x * ((size_t)blockIdx.y + (size_t)gridDim.y * blockIdx.z));
            if ((threadIdx.x & 63) == 0) atomicMin(s, wall_clock64());
        }
    }
    __device__ __forceinline__ ~StampScope() {
        if (s && (threadIdx.x & 63) == 0) atomicMax(s + 1, wall_clock64());
    }
};

// flag: a caller-defined bit reported separately (preprocess backward: the next iteration's forward rides in this launch);
// the kernel's LAST parameter is its LaunchStamp
template <typename... Args, typename F = void (*)(Args..., LaunchStamp)>
inline void launch_kernel(int kind, int flag, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    const LaunchStamp st = launch_timing_on() ? launch_timing_slots(kind, flag, (size_t)grid.x * grid.y * grid.z) : LaunchStamp{nullptr};
    kernel<<<grid, block, lds, s>>>(args..., st);
}
#endif

}  // namespace gps
