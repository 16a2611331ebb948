// Does hipExtStreamCreateWithCUMask take effect on this stack, and how are mask bits numbered against XCDs?
// A VALU-bound kernel (4096 workgroups) is timed on streams with different masks; each workgroup also records the XCC_ID
// it ran on, so the per-XCD share of workgroups shows where a mask's closed units sit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(float* out, int* xcc_count, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-5f; }
    if (a == 12345.f) out[0] = a + b;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicAdd(&xcc_count[xcc & 7], 1);
    }
}
static float run(hipStream_t s, float* out, int* cnt, int* host_cnt) {
    hipMemsetAsync(cnt, 0, 32, s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<<<4096, 256, 0, s>>>(out, cnt, 20000);
    hipMemsetAsync(cnt, 0, 32, s);
    hipEventRecord(e0, s);
    spin<<<4096, 256, 0, s>>>(out, cnt, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host_cnt, cnt, 32, hipMemcpyDeviceToHost);
    return ms;
}
int main() {
    float* out; int* cnt; int h[8];
    hipMalloc(&out, 4); hipMalloc(&cnt, 32);
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    printf("no mask            : %.3f ms  per-XCD workgroups", run(s0, out, cnt, h)); for (int k = 0; k < 8; k++) printf(" %d", h[k]); printf("\n");
    struct { const char* name; uint32_t w[8]; } masks[] = {
        {"all ones           ", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"words 0-3 only     ", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
        {"even bits only     ", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
        {"bits i%8==0 only   ", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"word 0 only        ", {~0u, 0, 0, 0, 0, 0, 0, 0}},
    };
    for (auto& m : masks) {
        hipStream_t s; 
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("%s: create failed %s\n", m.name, hipGetErrorString(e)); continue; }
        printf("%s: %.3f ms  per-XCD workgroups", m.name, run(s, out, cnt, h)); for (int k = 0; k < 8; k++) printf(" %d", h[k]); printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
