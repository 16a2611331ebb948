// gps_launch_timing_start / _stop / _read: see launch_timing.hpp and include/gps_slam_hip.h.
// (events are handed out here and bound to the kernel dispatch by hipExtLaunchKernelGGL at the launch site)
#include "common.hpp"
#include "launch_timing.hpp"

#include <mutex>
#include <vector>

namespace gps {
namespace {

struct Pair { hipEvent_t a, b; int kind, flag; bool closed; };

std::mutex g_mu;
int g_on = 0;                 // (read without the lock on the launch path)
size_t g_capacity = 0;
std::vector<Pair> g_pairs;    // launches of the current session, in record order over all streams and host threads
std::vector<hipEvent_t> g_pool;
double g_total_us[TK_COUNT][2];
long long g_launches[TK_COUNT][2];
double g_max_us[TK_COUNT];
long long g_dropped = 0;

hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

}  // namespace

bool launch_timing_on() { return __atomic_load_n(&g_on, __ATOMIC_RELAXED) != 0; }

bool launch_timing_events(int kind, int flag, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on || kind < 0 || kind >= TK_COUNT) return false;
    if (g_pairs.size() >= g_capacity) { g_dropped++; return false; }
    Pair p = {take_event(), take_event(), kind, flag ? 1 : 0, true};
    if (!p.a || !p.b) {
        if (p.a) g_pool.push_back(p.a);
        g_dropped++;
        return false;
    }
    g_pairs.push_back(p);
    *start = p.a; *stop = p.b;
    return true;
}

}  // namespace gps

extern "C" {

int gps_launch_timing_start(int capacity) {
    GPS_REQUIRE(capacity > 0 && capacity <= (1 << 20));
    std::lock_guard<std::mutex> lk(gps::g_mu);
    for (auto& p : gps::g_pairs) { gps::g_pool.push_back(p.a); gps::g_pool.push_back(p.b); }
    gps::g_pairs.clear();
    gps::g_pairs.reserve((size_t)capacity);
    gps::g_capacity = (size_t)capacity;
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
    __atomic_store_n(&gps::g_on, 0, __ATOMIC_RELAXED);
    std::lock_guard<std::mutex> lk(gps::g_mu);
    int rc = GPS_OK;
    for (auto& p : gps::g_pairs) {
        float ms = 0.f;
        if (p.closed && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            gps::g_total_us[p.kind][p.flag] += 1e3 * (double)ms;
            gps::g_launches[p.kind][p.flag]++;
            if (1e3 * (double)ms > gps::g_max_us[p.kind]) gps::g_max_us[p.kind] = 1e3 * (double)ms;
        } else {
            rc = GPS_ERR_LAUNCH;
        }
        gps::g_pool.push_back(p.a);
        gps::g_pool.push_back(p.b);
    }
    gps::g_pairs.clear();
    (void)hipGetLastError();
    return rc;
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
