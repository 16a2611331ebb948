// Host <-> GPU doorbell latency probe (not part of the library): a persistent single-wave kernel waits for flag == i in
//   mode 0: pinned host memory (the GPU polls across PCIe)
//   mode 1: fine-grained device memory the host writes through the BAR (the GPU polls local memory), plain store
//   mode 2: as mode 1 with an sfence behind the store (the BAR mapping is write-combining: without the fence the store sits in
//           the core's WC buffer until something evicts it -- round 3's first reading of mode 1, 34.8 us, measured THAT)
// and acknowledges by writing i into pinned host memory; the host measures the round trip.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <immintrin.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void pong(const uint32_t* flag, volatile uint32_t* ack, int n) {
    const long long t0 = wall_clock64();
    for (int i = 1; i <= n; i++) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)i) {
            if (wall_clock64() - t0 > 300000000LL) return;  // 3 s
        }
        *ack = (uint32_t)i;
    }
}

int main() {
    const int n = 2000;
    uint32_t *ack, *hflag, *dflag = nullptr;
    CK(hipHostMalloc(&ack, 64));
    CK(hipHostMalloc(&hflag, 64));
    hipError_t e = hipExtMallocWithFlags((void**)&dflag, 4096, hipDeviceMallocFinegrained);
    printf("fine-grained device alloc: %s\n", hipGetErrorString(e));
    for (int mode = 0; mode < 3; mode++) {
        volatile uint32_t* flag = mode == 0 ? hflag : dflag;
        if (!flag) continue;
        if (mode >= 1) {
            hipPointerAttribute_t at;
            CK(hipPointerGetAttributes(&at, dflag));
            printf("device flag: type %d host ptr %p dev ptr %p\n", (int)at.type, at.hostPointer, at.devicePointer);
            CK(hipMemset(dflag, 0, 64));
        } else {
            *flag = 0;
        }
        *ack = 0;
        hipLaunchKernelGGL(pong, 1, 64, 0, 0, (const uint32_t*)flag, (volatile uint32_t*)ack, n);
        auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        for (int i = 1; i <= n && ok; i++) {
            *flag = (uint32_t)i;   // mode 1, 2: a store through the BAR mapping (faults if the memory is not host visible)
            if (mode == 2) _mm_sfence();
            long spins = 0;
            while (*(volatile uint32_t*)ack != (uint32_t)i) if (++spins > 2000000000L) { ok = false; break; }
        }
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        printf("mode %d: %s, %.2f us per round trip\n", mode, ok ? "ok" : "TIMEOUT",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    }
    return 0;
}
