// gps_launch_timing_start / _stop / _read: see launch_timing.hpp and include/gps_slam_hip.h.
#include "common.hpp"
#include "launch_timing.hpp"

#include <mutex>
#include <vector>

namespace gps {
namespace {

struct Rec { int kind, flag; size_t first, count; };   // slots [first, first + count) = the launch's workgroups

std::mutex g_mu;
int g_on = 0;                        // (read without the lock on the launch path)
unsigned long long* g_slots = nullptr;   // device: {first start, last end} per workgroup, initialised to {~0, 0}
size_t g_capacity = 0, g_used = 0;       // in slots
std::vector<Rec> g_recs;             // launches of the current window, in slot order over all streams and host threads
double g_total_us[TK_COUNT][2];
long long g_launches[TK_COUNT][2];
double g_max_us[TK_COUNT];
long long g_dropped = 0;
constexpr double TICK_US = 0.01;     // wall_clock64(): the 100 MHz constant clock

__global__ void stamp_init_kernel(unsigned long long* s, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { s[2 * i] = ~0ull; s[2 * i + 1] = 0ull; }
}

}  // namespace

bool launch_timing_on() { return __atomic_load_n(&g_on, __ATOMIC_RELAXED) != 0; }

LaunchStamp launch_timing_slots(int kind, int flag, size_t n_workgroups) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || !g_slots || kind < 0 || kind >= TK_COUNT || n_workgroups == 0) return LaunchStamp{nullptr};
    if (g_used + n_workgroups > g_capacity) { g_dropped++; return LaunchStamp{nullptr}; }
    g_recs.push_back(Rec{kind, flag ? 1 : 0, g_used, n_workgroups});
    g_used += n_workgroups;
    return LaunchStamp{g_slots + 2 * g_recs.back().first};
}

}  // namespace gps

extern "C" {

int gps_launch_timing_start(int capacity) {
    GPS_ENTER();
    GPS_REQUIRE(capacity > 0 && capacity <= (1 << 24));   // workgroup slots (16 bytes each)
    std::lock_guard<std::mutex> lk(gps::g_mu);
    if (gps::g_on) return GPS_ERR_ARG;   // (one window at a time)
    if (gps::g_slots && gps::g_capacity != (size_t)capacity) { (void)hipFree(gps::g_slots); gps::g_slots = nullptr; }
    if (!gps::g_slots && hipMalloc(&gps::g_slots, (size_t)capacity * 16) != hipSuccess) { gps::g_slots = nullptr; return GPS_ERR_LAUNCH; }
    gps::g_capacity = (size_t)capacity;
    gps::stamp_init_kernel<<<gps_div_up(capacity, 256), 256, 0, nullptr>>>(gps::g_slots, (size_t)capacity);
    if (hipDeviceSynchronize() != hipSuccess) return GPS_ERR_LAUNCH;
    gps::g_recs.clear();
    gps::g_used = 0;
    gps::g_dropped = 0;
    for (int k = 0; k < gps::TK_COUNT; k++) {
        gps::g_total_us[k][0] = gps::g_total_us[k][1] = 0.0;
        gps::g_launches[k][0] = gps::g_launches[k][1] = 0;
        gps::g_max_us[k] = 0.0;
    }
    __atomic_store_n(&gps::g_on, 1, __ATOMIC_RELAXED);
    return GPS_OK;
}

int gps_launch_timing_stop(void) {
    GPS_ENTER();
    __atomic_store_n(&gps::g_on, 0, __ATOMIC_RELAXED);
    if (hipDeviceSynchronize() != hipSuccess) return GPS_ERR_LAUNCH;   // every stamped launch has ended
    std::lock_guard<std::mutex> lk(gps::g_mu);
    const size_t n = gps::g_used;
    if (n == 0) return GPS_OK;
    std::vector<unsigned long long> h(2 * n);
    if (hipMemcpy(h.data(), gps::g_slots, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return GPS_ERR_LAUNCH;
    for (const gps::Rec& r : gps::g_recs) {
        unsigned long long a = ~0ull, b = 0ull;
        for (size_t i = r.first; i < r.first + r.count; i++) {
            if (h[2 * i] < a) a = h[2 * i];
            if (h[2 * i + 1] > b) b = h[2 * i + 1];
        }
        if (a == ~0ull || b < a) continue;   // (a launch that was refused)
        const double us = (double)(b - a) * gps::TICK_US;
        gps::g_total_us[r.kind][r.flag] += us;
        gps::g_launches[r.kind][r.flag]++;
        if (us > gps::g_max_us[r.kind]) gps::g_max_us[r.kind] = us;
    }
    gps::g_recs.clear();
    return GPS_OK;
}

int gps_launch_timing_read(int kind, double* total_us, int64_t* launches, double* total_us_flagged, int64_t* launches_flagged,
                           double* max_us, int64_t* dropped) {
    GPS_REQUIRE(kind >= 0 && kind < gps::TK_COUNT);
    std::lock_guard<std::mutex> lk(gps::g_mu);
    if (total_us) *total_us = gps::g_total_us[kind][0] + gps::g_total_us[kind][1];
    if (launches) *launches = gps::g_launches[kind][0] + gps::g_launches[kind][1];
    if (total_us_flagged) *total_us_flagged = gps::g_total_us[kind][1];
    if (launches_flagged) *launches_flagged = gps::g_launches[kind][1];
    if (max_us) *max_us = gps::g_max_us[kind];
    if (dropped) *dropped = gps::g_dropped;
    return GPS_OK;
}

}  // extern "C"
