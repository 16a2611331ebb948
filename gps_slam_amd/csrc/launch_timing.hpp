// In-loop launch timing (gps_launch_timing_*): while it is on, the instrumented launch sites dispatch through
// hipExtLaunchKernelGGL with a start and a stop event bound to THE KERNEL'S OWN dispatch packet -- its begin / end timestamps,
// what a rocprofv3 kernel trace reports -- while a host runs its normal schedule.  This is what bench.py's `roofline` prices
// (the same kernel timed alone in a back-to-back loop is faster: nothing collides with it and its inputs are warm).  Off (one
// relaxed load per launch, the plain <<<>>> launch) unless a caller turned it on; results never depend on it.
// (A first version bracketed the launches with hipEventRecord: two barrier packets with system-scope releases per launch
// added ~16 us to a 79 us kernel.)
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace gps {

enum TimedKernel {   // == GPS_TIMED_* of include/gps_slam_hip.h
    TK_PREPROCESS_BWD = 0, TK_PREPROCESS_FWD, TK_RASTER_FWD, TK_RASTER_BWD_STRIPS, TK_SB_SCAN, TK_SB_SCATTER, TK_INTEGRATE, TK_RAYCAST,
    TK_COUNT
};

bool launch_timing_on();
// a fresh (start, stop) pair for one launch of `kind`; false: ring full or no events (launch untimed, counted as dropped)
bool launch_timing_events(int kind, int flag, hipEvent_t* start, hipEvent_t* stop);

// flag: a caller-defined bit reported separately (preprocess backward: the next iteration's forward rides in this launch)
template <typename... Args, typename F = void (*)(Args...)>
inline void launch_kernel(int kind, int flag, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    hipEvent_t e0, e1;
    if (launch_timing_on() && launch_timing_events(kind, flag, &e0, &e1))
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, e0, e1, 0, args...);
    else
        kernel<<<grid, block, lds, s>>>(args...);
}

}  // namespace gps
