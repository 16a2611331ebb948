// Which allocation calls make the driver hold this process's queues?  A tiny kernel is launched and waited for in a loop (its
// round trip is ~10 us); 100 ms in, ONE action is performed between two round trips; the longest round trips after it, with the
// time at which they ended, are printed.  hipcc --offload-arch=gfx950 -O2 -o queue_hold queue_hold.hip; ./queue_hold <action>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <emmintrin.h>
#include <unistd.h>
__global__ void tick(unsigned* p) { p[0] += 1; }
__global__ void spin(volatile unsigned* flag, unsigned* out) {   // a polling kernel like the tracker's: ends when the host says so
    long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < 300000000LL) {}
    out[1] = 1;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    std::string act = argc > 1 ? argv[1] : "none";
    const bool polling = argc > 2 && std::string(argv[2]) == "poll";   // round trips = one spin kernel each, released by the host
    unsigned* d; hipMalloc(&d, 256); hipMemset(d, 0, 256);
    unsigned* flag; hipHostMalloc(&flag, 4096, hipHostMallocDefault); flag[0] = 0;
    void* held_pinned = nullptr; hipHostMalloc(&held_pinned, 1 << 20, hipHostMallocDefault); memset(held_pinned, 1, 1 << 20);
    void* held_big = nullptr; hipMalloc(&held_big, 1 << 30); hipMemset(held_big, 0, 1 << 30);
    void* held_vram = nullptr; hipMalloc(&held_vram, 1 << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int i = 0; i < 200; i++) { tick<<<1, 64, 0, st>>>(d); hipStreamSynchronize(st); }
    usleep(300000);   // whatever the set-up above caused has passed
    const double t0 = now();
    bool done = false;
    std::vector<std::pair<double, double>> trips;   // (end time, duration)
    double t_act = 0, t_act_len = 0;
    while (now() - t0 < 500.0) {
        const double a = now();
        if (polling) {
            flag[0] = 0; _mm_sfence();
            spin<<<1, 64, 0, st>>>(flag, d);
            const double b = now();
            while (now() - b < 0.02) {}     // the kernel is running (or held) now
            flag[0] = 1; _mm_sfence();
            hipStreamSynchronize(st);
        } else {
            tick<<<1, 64, 0, st>>>(d);
            hipStreamSynchronize(st);
        }
        const double e = now();
        trips.push_back({e - t0, e - a});
        if (!done && e - t0 > 100.0) {
            done = true;
            const double s = now();
            if (act == "bar") { void* p; hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained); ((volatile unsigned long long*)p)[7] = 5; _mm_sfence(); }
            else if (act == "bar_notouch") { void* p; hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained); }
            else if (act == "pinned") { void* p; hipHostMalloc(&p, 1 << 20, hipHostMallocDefault); memset(p, 1, 1 << 20); }
            else if (act == "pinned_notouch") { void* p; hipHostMalloc(&p, 1 << 20, hipHostMallocDefault); }
            else if (act == "pinned_big") { void* p; hipHostMalloc(&p, 512 << 20, hipHostMallocDefault); memset(p, 1, 512 << 20); }
            else if (act == "pinned_many") { for (int i = 0; i < 600; i++) { void* p; hipHostMalloc(&p, 1 << 20, hipHostMallocDefault); memset(p, 1, 1 << 20); } }
            else if (act == "pinned_many_notouch") { for (int i = 0; i < 600; i++) { void* p; hipHostMalloc(&p, 1 << 20, hipHostMallocDefault); } }
            else if (act == "vram_many") { for (int i = 0; i < 100; i++) { void* p; hipMalloc(&p, 20 << 20); } }
            else if (act == "vram_big_memset") { void* p; hipMalloc(&p, 1 << 30); hipMemsetAsync(p, 0, 1 << 30, st); hipStreamSynchronize(st); }
            else if (act == "vram_big_free") { hipFree(held_big); }
            else if (act == "bar_many") { for (int i = 0; i < 16; i++) { void* p; hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained); ((volatile unsigned long long*)p)[7] = 5; _mm_sfence(); } }
            else if (act == "vram") { void* p; hipMalloc(&p, 1 << 20); }
            else if (act == "vram_big") { void* p; hipMalloc(&p, 1 << 30); }
            else if (act == "vram_free") { hipFree(held_vram); }
            else if (act == "pinned_free") { hipHostFree(held_pinned); }
            else if (act == "stream") { hipStream_t s2; hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1); tick<<<1, 64, 0, s2>>>(d); hipStreamSynchronize(s2); }
            else if (act == "event") { hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming); }
            t_act = s - t0; t_act_len = now() - s;
        }
    }
    std::sort(trips.begin(), trips.end(), [](auto& x, auto& y) { return x.second > y.second; });
    printf("%-15s %s: action at %.1f ms took %.2f ms; %zu round trips; longest:", act.c_str(), polling ? "poll" : "tick", t_act, t_act_len, trips.size());
    for (int i = 0; i < 3; i++) printf("  %.2f ms ending at %.1f ms", trips[i].second, trips[i].first);
    printf("\n");
    return 0;
}
