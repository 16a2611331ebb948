// Issue cost of the VALU / cross-lane / LDS instructions the rasterizers are made of (not part of the library).
// Every wave runs REPS trips of 16 independent copies of one instruction; s_memtime around the loop; waves per SIMD = 1, 2, 4, 8.
// Printed: shader cycles per wave-instruction per SIMD (= per-wave cycles / waves per SIMD) -> what "N wave-instructions" cost.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probe/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REPS = 2000;

#define BODY16(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) STMT(8) STMT(9) STMT(10) STMT(11) STMT(12) STMT(13) STMT(14) STMT(15)

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(uint64_t* __restrict__ out, float seed) {
    __shared__ float4 lds[64];
    float a[16];
    float b = seed + 1.0f, c = seed * 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p[16];
    const v2 pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = seed + k + threadIdx.x; p[k] = v2{a[k], a[k] + 1.f}; }
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const uint32_t laddr = (uint32_t)(uintptr_t)&lds[(threadIdx.x >> 5) & 1];
    uint64_t t0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; r++) {
        if (OP == 0) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
        } else if (OP == 1) {
#define S(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
            BODY16(S)
#undef S
        } else if (OP == 2) {
#define S(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 3) {
#define S(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 4) {
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            BODY16(S)
#undef S
        } else if (OP == 5) {
#define S(k) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 6) {
#define S(k) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 7) {
#define S(k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 1) & 15]));
            BODY16(S)
#undef S
        } else if (OP == 8) {
#define S(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
            BODY16(S)
#undef S
        } else if (OP == 9) {
#define S(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
        } else if (OP == 10) {
#define S(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
            BODY16(S)
#undef S
        } else if (OP == 11) {  // broadcast LDS read of 16 bytes (all lanes of a half one address)
            float4 q[16];
#define S(k) asm volatile("ds_read_b128 %0, %1" : "=v"(q[k]) : "v"(laddr));
            BODY16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("" :: "v"(q[k].x), "v"(q[k].y), "v"(q[k].z), "v"(q[k].w));
        } else if (OP == 12) {
#define S(k) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
            BODY16(S)
#undef S
        } else if (OP == 13) {
#define S(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 1) & 15]));
            BODY16(S)
#undef S
        } else if (OP == 14) {
#define S(k) asm volatile("v_log_f32 %0, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 15) {
#define S(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 16) {
#define S(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            BODY16(S)
#undef S
        } else if (OP == 17) {
#define S(k) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 18) {
#define S(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            BODY16(S)
#undef S
        } else if (OP == 20) {  // packed FMA with one operand broadcast from a register half (what scalar x {px0, px1} compiles to)
#define S(k) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(p[k]) : "v"(pb), "v"(pc));
            BODY16(S)
#undef S
        } else if (OP == 21) {
#define S(k) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[k]) : "v"(pb));
            BODY16(S)
#undef S
        } else if (OP == 22) {  // compare into an SGPR pair (VOP3)
#define S(k) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %0, %1" : : "v"(a[k]), "v"(b) : "s20", "s21");
            BODY16(S)
#undef S
        } else if (OP == 23) {  // select by an SGPR-pair mask (VOP3)
#define S(k) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]" : "+v"(a[k]) : : );
            BODY16(S)
#undef S
        } else if (OP == 24) {
#define S(k) asm volatile("v_readlane_b32 s22, %0, s23" : : "v"(a[k]) : "s22");
            BODY16(S)
#undef S
        } else if (OP == 25) {  // 32-bit literal operand
#define S(k) asm volatile("v_min_f32 %0, 0x3f7fbe77, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 26) {  // SGPR operand
#define S(k) asm volatile("v_mul_f32 %0, s24, %0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 27) {  // VOP3 with a source modifier
#define S(k) asm volatile("v_exp_f32_e64 %0, -%0" : "+v"(a[k]));
            BODY16(S)
#undef S
        } else if (OP == 28) {  // the forward rasterizer's pair: 2 LDS reads (b128, b128 broadcast) per 8 FMAs
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
            float4 q0, q1;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(laddr));
            a[0] += q0.x + q1.y;
        } else if (OP == 19) {  // DEPENDENT chain of v_fma (latency)
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            BODY16(S)
#undef S
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k] + p[k].x + p[k].y;
    if (s == 12345.678f) out[0] = 1;  // keep the chains alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    (void)t0;
}

template <int OP>
int run(const char* name, uint64_t* d_out) {
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {
        // ONE workgroup per CU up to 4 waves per SIMD (its waves start together and stay resident together: a steady state --
        // several smaller workgroups per CU start skewed and the average per-wave time then underestimates the contention);
        // 8 per SIMD = two 1,024-thread workgroups per CU
        const int threads = 256 * (wps > 4 ? 4 : wps);
        const int blocks_per_cu = wps > 4 ? wps / 4 : 1;
        const int blocks = 256 * blocks_per_cu;
        const int waves = blocks * threads / 64;
        CK(hipMemset(d_out, 0, (waves + 1) * 8));
        hipLaunchKernelGGL(rate_kernel<OP>, blocks, threads, 0, 0, d_out, 1.0f);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> h(waves + 1);
        CK(hipMemcpy(h.data(), d_out, (waves + 1) * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int w = 0; w < waves; w++) sum += (double)h[1 + w];
        const double per_wave = sum / waves / (REPS * 16.0);
        // s_memtime ticks at 100 MHz on gfx9 (constant clock) unless it reads the shader clock: print both raw readings
        printf("  wps%d: %7.3f /wave %7.3f /simd", wps, per_wave, per_wave / wps);
    }
    printf("\n");
    return 0;
}

int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, (256 * 8 * 8 + 1) * 8));
    printf("s_memtime ticks per wave-instruction (per wave | divided by waves per SIMD).  The kernel's wall time gives the tick unit:\n");
    {   // calibrate the tick: time a known kernel with events
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_kernel<0>, 256, 256, 0, 0, d_out, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<0>, 256, 256, 0, 0, d_out, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t h[2];
        CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
        printf("calibration: v_fma x %d per wave, 1 wave/SIMD: kernel %.1f us (incl. launch), wave 0 measured %llu ticks -> %.3f ticks/ns\n",
               REPS * 16, ms * 1e3, (unsigned long long)h[1], (double)h[1] / (ms * 1e6));
    }
    run<0>("v_fma_f32", d_out);
    run<20>("v_pk_fma_f32 op_sel_hi", d_out);
    run<21>("v_pk_add_f32 neg+op_sel", d_out);
    run<22>("v_cmp_e64 -> sgpr pair", d_out);
    run<23>("v_cndmask_e64 sgpr mask", d_out);
    run<24>("v_readlane_b32", d_out);
    run<25>("v_min_f32 literal", d_out);
    run<26>("v_mul_f32 sgpr operand", d_out);
    run<27>("v_exp_f32_e64 neg", d_out);
    run<28>("16 v_fma + 2 ds_read_b128", d_out);
    run<19>("v_fma_f32 dependent", d_out);
    run<1>("v_pk_fma_f32", d_out);
    run<10>("v_pk_mul_f32", d_out);
    run<2>("v_exp_f32", d_out);
    run<14>("v_log_f32", d_out);
    run<3>("v_rcp_f32", d_out);
    run<4>("v_mul_lo_u32", d_out);
    run<9>("v_mad_u32_u24", d_out);
    run<18>("v_add_u32", d_out);
    run<5>("v_cvt_f32_i32", d_out);
    run<17>("v_cvt_i32_f32", d_out);
    run<16>("v_max_f32", d_out);
    run<8>("v_cndmask_b32", d_out);
    run<12>("v_cmp + v_cndmask", d_out);
    run<6>("v_add_f32 dpp row_shr", d_out);
    run<15>("v_mov_b32 dpp quad_perm", d_out);
    run<7>("v_permlane16_swap", d_out);
    run<13>("v_permlane32_swap", d_out);
    run<11>("ds_read_b128 broadcast", d_out);
    return 0;
}
