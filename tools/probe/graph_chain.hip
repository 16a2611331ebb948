// A chain of N short dependent kernels on one stream: launched one by one vs as an instantiated hipGraph (stream capture).
// Measures the GPU-side chain time (events) and the host time per chain.  hipcc --offload-arch=gfx950 -O2 -o tools/probe/graph_chain tools/probe/graph_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Blob { int v[64]; };
__global__ void k(Blob b, int* data, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] = data[i] * 3 + b.v[1];
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 1 << 20, N = 8, REP = 200;
    int* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Blob b{}; b.v[1] = 1;
    for (int grid : {64, 1152, 4096}) {
        for (int w = 0; w < 20; w++) for (int j = 0; j < N; j++) k<<<grid, 256, 0, st>>>(b, d, n);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        double t0 = now();
        for (int r = 0; r < REP; r++) for (int j = 0; j < N; j++) k<<<grid, 256, 0, st>>>(b, d, n);
        double t1 = now();
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %4d  plain launches: GPU %.2f us per chain of %d (%.2f per kernel), host %.2f us per chain\n", grid, ms * 1e3 / REP, N, ms * 1e3 / REP / N, (t1 - t0) / REP);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int j = 0; j < N; j++) k<<<grid, 256, 0, st>>>(b, d, n);
        hipStreamEndCapture(st, &g);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
        for (int w = 0; w < 20; w++) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        t0 = now();
        for (int r = 0; r < REP; r++) hipGraphLaunch(ge, st);
        t1 = now();
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %4d  graph launches: GPU %.2f us per chain of %d (%.2f per kernel), host %.2f us per chain\n", grid, ms * 1e3 / REP, N, ms * 1e3 / REP / N, (t1 - t0) / REP);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
