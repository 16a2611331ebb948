// host cost of hipLaunchKernel as a function of the kernel-argument size, one stream / two threads launching on two streams
// hipcc --offload-arch=gfx950 -O2 -o tools/probe/launch_cost tools/probe/launch_cost.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
template <int N> struct Blob { int v[N]; };
template <int N> __global__ void k(Blob<N> b, int* out) { if (b.v[0] == 12345) out[0] = b.v[N - 1]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int N> double run(hipStream_t st, int* out, int iters) {
    Blob<N> b{}; 
    for (int i = 0; i < 200; i++) k<N><<<1, 64, 0, st>>>(b, out);
    hipStreamSynchronize(st);
    double t0 = now();
    for (int i = 0; i < iters; i++) k<N><<<1, 64, 0, st>>>(b, out);
    double t1 = now();
    hipStreamSynchronize(st);
    double t2 = now();
    printf("args %4d B: host %.2f us per launch, end-to-end %.2f us per launch\n", (int)sizeof(Blob<N>), (t1 - t0) / iters, (t2 - t0) / iters);
    return (t1 - t0) / iters;
}
int main() {
    int* out; hipMalloc(&out, 4);
    hipStream_t a, b; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo);
    run<4>(a, out, 2000); run<16>(a, out, 2000); run<64>(a, out, 2000); run<128>(a, out, 2000); run<256>(a, out, 2000);
    printf("-- with a second thread launching on another stream --\n");
    volatile bool stop = false;
    std::thread t([&] { Blob<64> bb{}; while (!stop) { for (int i = 0; i < 50; i++) k<64><<<1, 64, 0, b>>>(bb, out); hipStreamSynchronize(b); } });
    run<16>(a, out, 2000); run<128>(a, out, 2000);
    stop = true; t.join();
    return 0;
}
