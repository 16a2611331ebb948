// Fused SSIM map, forward and backward (the `ssim_weight > 0` loss option).
//   ssim.cu:209-303  fusedssimCUDA           -> ssim_fwd_kernel
//   ssim.cu:305-383  fusedssim_backwardCUDA  -> ssim_bwd_kernel
// 11-tap separable Gaussian window (sigma 1.5), zero padding outside the image, per channel.
//
// The reference runs five (forward) / three (backward) separate separable convolutions per channel through two LDS image
// tiles and one scratch tile, ~25 barriers per channel, channels in a loop.  Here a workgroup loads the (32+10)^2 halo tiles
// once, the x-pass produces ALL moments (x1, x1^2, x2, x2^2, x1 x2) of a row element from the same 22 LDS reads, the y-pass
// reads them back -- two barriers -- and the channel is a grid dimension (900 workgroups at 640x480x3 instead of 300).
// Images may be planar (NCHW, the reference's layout) or interleaved (HWC, how the renderer and the camera hold them): the
// reference's `.permute().contiguous()` copies disappear.  Tap order and the use of fused multiply-adds follow the reference
// (`val += G_k * p` contracted by nvcc).
#include "common.hpp"

namespace {

constexpr int TS = 32;          // output tile edge
constexpr int HALO = 5;
constexpr int TIN = TS + 2 * HALO;  // 42
constexpr int LD_IN = TIN + 1;      // padded row stride of the input tiles
constexpr int LD_H = TS + 1;        // padded row stride of the x-pass results

__device__ __constant__ float G[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                       0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                       0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                       0.0075987582094967365f, 0.001028380123898387f};

struct Layout { int64_t sb, sc, sy, sx; };  // element strides of (batch, channel, row, column)
__host__ __device__ inline Layout make_layout(int CH, int H, int W, int channels_last) {
    Layout l;
    if (channels_last) { l.sb = (int64_t)H * W * CH; l.sy = (int64_t)W * CH; l.sx = CH; l.sc = 1; }
    else { l.sb = (int64_t)CH * H * W; l.sc = (int64_t)H * W; l.sy = W; l.sx = 1; }
    return l;
}

__device__ __forceinline__ float pix(const float* __restrict__ img, const Layout& l, int b, int c, int y, int x, int H, int W) {
    return (x >= W || y >= H || x < 0 || y < 0) ? 0.0f : img[b * l.sb + c * l.sc + y * l.sy + x * l.sx];
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(int H, int W, int CH, float C1, float C2, Layout l,
                                                      const float* __restrict__ img1, const float* __restrict__ img2,
                                                      float* __restrict__ ssim_map, float* __restrict__ dm_dmu1,
                                                      float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12) {
    __shared__ float a[TIN][LD_IN], bb[TIN][LD_IN];
    __shared__ float h[5][TIN][LD_H];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const int c = blockIdx.z % CH, b = blockIdx.z / CH;
    for (int q = tid; q < TIN * TIN; q += 256) {
        const int ly = q / TIN, lx = q - ly * TIN;
        a[ly][lx] = pix(img1, l, b, c, y0 + ly - HALO, x0 + lx - HALO, H, W);
        bb[ly][lx] = pix(img2, l, b, c, y0 + ly - HALO, x0 + lx - HALO, H, W);
    }
    __syncthreads();
    for (int q = tid; q < TIN * TS; q += 256) {  // x-pass: row ly of the halo tile, output column lx
        const int ly = q / TS, lx = q - ly * TS;
        float s1 = 0.f, s11 = 0.f, s2 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float p = a[ly][lx + k], r = bb[ly][lx + k];
            s1 = fmaf(G[k], p, s1);
            s11 = fmaf(G[k], p * p, s11);
            s2 = fmaf(G[k], r, s2);
            s22 = fmaf(G[k], r * r, s22);
            s12 = fmaf(G[k], p * r, s12);
        }
        h[0][ly][lx] = s1; h[1][ly][lx] = s11; h[2][ly][lx] = s2; h[3][ly][lx] = s22; h[4][ly][lx] = s12;
    }
    __syncthreads();
    for (int q = tid; q < TS * TS; q += 256) {  // y-pass + the SSIM expression
        const int ly = q / TS, lx = q - ly * TS;
        float mu1 = 0.f, e11 = 0.f, mu2 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            mu1 = fmaf(G[k], h[0][ly + k][lx], mu1);
            e11 = fmaf(G[k], h[1][ly + k][lx], e11);
            mu2 = fmaf(G[k], h[2][ly + k][lx], mu2);
            e22 = fmaf(G[k], h[3][ly + k][lx], e22);
            e12 = fmaf(G[k], h[4][ly + k][lx], e12);
        }
        const int x = x0 + lx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        const float sigma1_sq = e11 - mu1 * mu1, sigma2_sq = e22 - mu2 * mu2, sigma12 = e12 - mu1 * mu2;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
        const float C = 2.0f * mu1_mu2 + C1, D = 2.0f * sigma12 + C2;
        const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
        const int64_t o = b * l.sb + c * l.sc + y * l.sy + x * l.sx;
        ssim_map[o] = (C * D) / (A * B);
        if (dm_dmu1) {
            dm_dmu1[o] = (mu2 * 2.0f * D) / (A * B) - (mu2 * 2.0f * C) / (A * B) - (mu1 * 2.0f * C * D) / (A * A * B) +
                         (mu1 * 2.0f * C * D) / (A * B * B);
            dm_dsigma1_sq[o] = (-C * D) / (A * B * B);
            dm_dsigma12[o] = (2 * C) / (A * B);
        }
    }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int W, int CH, Layout l, const float* __restrict__ img1,
                                                      const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                      const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                      const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
    __shared__ float t[3][TIN][LD_IN];  // dL * {dm_dmu1, dm_dsigma1_sq, dm_dsigma12} with halo
    __shared__ float h[3][TIN][LD_H];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const int c = blockIdx.z % CH, b = blockIdx.z / CH;
    for (int q = tid; q < TIN * TIN; q += 256) {
        const int ly = q / TIN, lx = q - ly * TIN;
        const int y = y0 + ly - HALO, x = x0 + lx - HALO;
        const float g = pix(dL_dmap, l, b, c, y, x, H, W);
        t[0][ly][lx] = pix(dm_dmu1, l, b, c, y, x, H, W) * g;
        t[1][ly][lx] = pix(dm_dsigma1_sq, l, b, c, y, x, H, W) * g;
        t[2][ly][lx] = pix(dm_dsigma12, l, b, c, y, x, H, W) * g;
    }
    __syncthreads();
    for (int q = tid; q < TIN * TS; q += 256) {
        const int ly = q / TS, lx = q - ly * TS;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            s0 = fmaf(G[k], t[0][ly][lx + k], s0);
            s1 = fmaf(G[k], t[1][ly][lx + k], s1);
            s2 = fmaf(G[k], t[2][ly][lx + k], s2);
        }
        h[0][ly][lx] = s0; h[1][ly][lx] = s1; h[2][ly][lx] = s2;
    }
    __syncthreads();
    for (int q = tid; q < TS * TS; q += 256) {
        const int ly = q / TS, lx = q - ly * TS;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            v0 = fmaf(G[k], h[0][ly + k][lx], v0);
            v1 = fmaf(G[k], h[1][ly + k][lx], v1);
            v2 = fmaf(G[k], h[2][ly + k][lx], v2);
        }
        const int64_t o = b * l.sb + c * l.sc + y * l.sy + x * l.sx;
        const float p1 = img1[o], p2 = img2[o];
        float d = 0.0f;
        d += v0;
        d += p1 * 2.0f * v1;
        d += p2 * v2;
        dL_dimg1[o] = d;
    }
}

}  // namespace

extern "C" {

int gps_ssim_fwd(int B, int CH, int H, int W, int channels_last, float C1, float C2, const float* img1, const float* img2,
                 float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(B > 0 && CH > 0 && H > 0 && W > 0 && img1 && img2 && ssim_map);
    GPS_REQUIRE((dm_dmu1 && dm_dsigma1_sq && dm_dsigma12) || (!dm_dmu1 && !dm_dsigma1_sq && !dm_dsigma12));
    GPS_REQUIRE((int64_t)B * CH <= 65535);
    const dim3 grid(gps_div_up(W, TS), gps_div_up(H, TS), B * CH);
    ssim_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(H, W, CH, C1, C2, make_layout(CH, H, W, channels_last), img1, img2,
                                                           ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_ssim_bwd(int B, int CH, int H, int W, int channels_last, const float* img1, const float* img2, const float* dL_dmap,
                 const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(B > 0 && CH > 0 && H > 0 && W > 0 && img1 && img2 && dL_dmap && dm_dmu1 && dm_dsigma1_sq && dm_dsigma12 && dL_dimg1);
    GPS_REQUIRE((int64_t)B * CH <= 65535);
    const dim3 grid(gps_div_up(W, TS), gps_div_up(H, TS), B * CH);
    ssim_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(H, W, CH, make_layout(CH, H, W, channels_last), img1, img2, dL_dmap,
                                                           dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"
