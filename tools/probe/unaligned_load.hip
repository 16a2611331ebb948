// Do multi-dword raw buffer / global loads work at dword (not natural) alignment on gfx950?  (not part of the library)
// lane l loads 16 / 8 bytes at byte offset 4 * (3 l + 1) (4-byte aligned only) and at 8 * (2 l + 1) (8-byte aligned) and compares
// with the expected floats.   hipcc --offload-arch=gfx950 -O3 -o /tmp/unaligned tools/probe/unaligned_load.hip && /tmp/unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const float* src, int n, int* bad) {
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    int errs = 0;
    {   // b128 at 4-byte alignment
        const unsigned off = 4u * (3u * l + 1u);
        float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        const float e = (float)(3 * l + 1);
        errs += (v.x != e) + (v.y != e + 1) + (v.z != e + 2) + (v.w != e + 3);
    }
    {   // b128 at 8-byte alignment
        const unsigned off = 8u * (2u * l + 1u);
        float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        const float e = (float)(2 * (2 * l + 1));
        errs += 16 * ((v.x != e) + (v.y != e + 1) + (v.z != e + 2) + (v.w != e + 3));
    }
    {   // b64 at 4-byte alignment
        const unsigned off = 4u * (2u * l + 1u);
        float2 v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
        const float e = (float)(2 * l + 1);
        errs += 256 * ((v.x != e) + (v.y != e + 1));
    }
    {   // global_load_dwordx4 at 4-byte alignment through a plain pointer
        const float* p = src + 5 * l + 3;
        float4 v;
        asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        const float e = (float)(5 * l + 3);
        errs += 4096 * ((v.x != e) + (v.y != e + 1) + (v.z != e + 2) + (v.w != e + 3));
    }
    bad[l] = errs;
}

int main() {
    const int n = 4096;
    float* h = new float[n];
    for (int i = 0; i < n; i++) h[i] = (float)i;
    float* d; int* bad;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&bad, 64 * 4));
    CK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, d, n, bad);
    CK(hipDeviceSynchronize());
    int hb[64];
    CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    int tot = 0;
    for (int i = 0; i < 64; i++) tot |= hb[i];
    printf("unaligned loads: error mask 0x%x (0 = every form returns the right floats; 0xf b128@4, 0xf0 b128@8, 0x300 b64@4, 0xf000 global x4@4)\n", tot);
    return 0;
}
