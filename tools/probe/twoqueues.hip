// When does a small kernel on stream B start while a long, chip-filling kernel runs on stream A?  (The overlap schedule
// depends on it: tracker iterations of the frame stream against rasterizer / raycast kernels of the other streams.)
// Stream A: `big` = many workgroups of a fixed spin; stream B: `small` (256 workgroups) launched while A runs; every kernel
// stamps wall_clock64 at its first and last workgroup.  Printed: B's start / end relative to A's start, A's duration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void work(long long* stamp, int iters, float* sink) {
    if (threadIdx.x == 0) atomicMin((unsigned long long*)&stamp[0], (unsigned long long)wall_clock64());
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-5f; }
    if (a == 12345.f) sink[0] = a + b;
    if (threadIdx.x == 0) atomicMax((unsigned long long*)&stamp[1], (unsigned long long)wall_clock64());
}
static void run(const char* name, int prioA, int prioB, int wgsA, int itersA) {
    hipStream_t a, b;
    hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prioA);
    hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prioB);
    long long *st; float* sink;
    hipMalloc(&st, 64); hipMalloc(&sink, 4);
    long long init[4] = {0x7fffffffffffffffLL, 0, 0x7fffffffffffffffLL, 0};
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(st, init, 32, hipMemcpyHostToDevice);
        work<<<64, 256, 0, a>>>(st + 0, 10, sink); work<<<64, 256, 0, b>>>(st + 2, 10, sink);  // warm both queues
        hipDeviceSynchronize();
        hipMemcpy(st, init, 32, hipMemcpyHostToDevice);
        work<<<wgsA, 256, 0, a>>>(st + 0, itersA, sink);
        usleep(100);
        work<<<256, 256, 0, b>>>(st + 2, 2000, sink);
        hipDeviceSynchronize();
        long long h[4]; hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
        if (rep == 2)
            printf("%-44s A lasts %7.1f us; B starts %7.1f us after A, lasts %7.1f us\n", name, (h[1] - h[0]) / 100.0,
                   (h[2] - h[0]) / 100.0, (h[3] - h[2]) / 100.0);
    }
    hipStreamDestroy(a); hipStreamDestroy(b); hipFree(st); hipFree(sink);
}
int main() {
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least, hi = greatest priority (numerically lower)
    printf("priority range: least %d greatest %d\n", lo, hi);
    run("A high (10800 WGs), B high", hi, hi, 10800, 40000);
    run("A normal (10800 WGs), B high", 0, hi, 10800, 40000);
    run("A low (10800 WGs), B high", lo, hi, 10800, 40000);
    run("A high (10800 WGs), B normal", hi, 0, 10800, 40000);
    run("A normal, B normal", 0, 0, 10800, 40000);
    run("A high (1200 WGs, long), B high", hi, hi, 1200, 400000);
    return 0;
}
