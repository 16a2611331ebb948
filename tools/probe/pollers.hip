// How long until ALL workgroups of a launch have seen a 64-byte line the host writes through the BAR into fine-grained device
// memory, as a function of how many workgroups poll it (the tracker's pre-launched evaluation: 256 pollers of one line).
// Each workgroup: 16 lanes poll (system-scope loads, s_sleep 1 between polls) until word 0 == round, then lane 0 stores
// wall_clock64() into stamp[blockIdx]; workgroup 0 acknowledges to pinned host memory.  Per round the host then reads the stamps:
// spread = last - first detection.   hipcc --offload-arch=gfx950 -O2 -o tools/probe/pollers tools/probe/pollers.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <immintrin.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void poll(const uint32_t* line, long long* stamp, volatile uint32_t* ack, int round, int pollers, const uint32_t* relay_src,
                     uint32_t* relay) {
    // pollers < gridDim.x: workgroups >= pollers wait on relay[blockIdx % pollers] (agent scope) which poller (blockIdx % pollers) writes
    const long long t0 = wall_clock64();
    const bool direct = (int)blockIdx.x < pollers;
    if (threadIdx.x < 16) {
        for (;;) {
            uint32_t v = direct ? __hip_atomic_load(line + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                : __hip_atomic_load(relay + 16 * (blockIdx.x % pollers) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__shfl(v, 0, 16) == (uint32_t)round) break;
            if (wall_clock64() - t0 > 100000000LL) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (direct && pollers < (int)gridDim.x && threadIdx.x == 0)
            __hip_atomic_store(relay + 16 * blockIdx.x, (uint32_t)round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) stamp[blockIdx.x] = wall_clock64();
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) *ack = (uint32_t)round;
}
int main() {
    const int G = 256, R = 300;
    uint32_t *ack, *line = nullptr, *relay; long long* stamp;
    CK(hipHostMalloc(&ack, 64)); CK(hipHostMalloc(&stamp, G * sizeof(long long)));
    CK(hipExtMallocWithFlags((void**)&line, 4096, hipDeviceMallocFinegrained)); CK(hipMemset(line, 0, 64));
    CK(hipMalloc(&relay, 16 * 4 * G)); CK(hipMemset(relay, 0, 16 * 4 * G));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int pollers : {256, 64, 8, 1}) {
        std::vector<double> spread, mean;
        for (int r = 1; r <= R; r++) {
            const int round = pollers * 1000 + r;
            *ack = 0;
            poll<<<G, 256, 0, st>>>(line, stamp, ack, round, pollers, nullptr, relay);
            for (volatile int w = 0; w < 20000; w++) {}   // let the launch become resident and start polling (~20 us)
            for (int k = 0; k < 16; k += 2) reinterpret_cast<volatile uint64_t*>(line)[k >> 1] = (uint64_t)(uint32_t)round * (k == 0) | 0;
            _mm_sfence();
            CK(hipStreamSynchronize(st));
            long long lo = stamp[0], hi = stamp[0]; double s = 0;
            for (int i = 0; i < G; i++) { lo = std::min(lo, stamp[i]); hi = std::max(hi, stamp[i]); }
            for (int i = 0; i < G; i++) s += (double)(stamp[i] - lo);
            spread.push_back((hi - lo) / 100.0); mean.push_back(s / G / 100.0);
        }
        std::sort(spread.begin(), spread.end()); std::sort(mean.begin(), mean.end());
        printf("%3d direct pollers of %d workgroups: last - first detection median %.2f us (p90 %.2f), mean lag behind the first %.2f us\n",
               pollers, G, spread[R / 2], spread[R * 9 / 10], mean[R / 2]);
    }
    return 0;
}
