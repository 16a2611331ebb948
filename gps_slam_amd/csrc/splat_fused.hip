// Model-level fused kernels: everything RawGaussianModel::gesForward does per Gaussian before
// binning (src/raw_gs_model.cpp:207-286) in ONE pass over the parameters, and its adjoint.
//
//   forward : scales = exp(log_scales); FullyFusedProjection; radii = clamp_max(radii, max_gs_radii);
//             dirs = means - cam_T; SphericalHarmonicsNew(deg, dirs, cat(dc, rest), radii > 0);
//             colors = cat(clamp_min(sh + 0.5, 0), depths); opac = sigmoid(opacities)
//   backward: the libtorch autograd chain of the same ops, given the rasterizer's gradients.
//
// The reference runs ~12 libtorch kernels forward (+ a 38 MB torch::cat of the SH coefficients every
// iteration, raw_gs_model.cpp:253) and ~20 backward; here each direction is one HBM stream:
// forward reads 59 floats/Gaussian and writes 15, backward reads the same + 10 gradient floats and writes
// the 59 parameter gradients (plain stores, no memsets, no atomics).
// The per-Gaussian math is splat_math.hpp, shared with the op-level kernels the parity tests pin.
#include "launch_timing.hpp"
#include "splat_adam.hpp"
#include "splat_bin.hpp"
#include "splat_math.hpp"

using namespace gps;

#ifndef GPS_FUSED_ADAM_THREADS
#define GPS_FUSED_ADAM_THREADS 128
#endif
GPS_TUNABLE_REPORT(GPS_FUSED_ADAM_THREADS, 128);

namespace {

struct FusedIn {
    const float* means;       // [N,3]
    const float* log_scales;  // [N,3]
    const float* quats;       // [N,4]
    const float* opac_logit;  // [N]
    const float* sh_dc;       // [N,3]
    const float* sh_rest;     // [N,K-1,3]
    const float* viewmat;     // [16] device
    const float* Kmat;        // [9] device
    const float* cam_pos;     // [3] device (c2w translation, raw_gs_model.cpp:202)
    int N, K, W, H, max_radii;
    float eps2d, near_plane, far_plane, radius_clip;
};

// The SH coefficients are 45 of the 59 floats of a Gaussian.  A thread-per-Gaussian read of its own 180-byte row touches 64
// different cache lines per load instruction (the TA serialises them); in the BACKWARD the workgroup's 256 rows -- one
// contiguous 46 KB block -- are copied to LDS with coalesced float4 loads and each thread reads its row from there (row stride 45
// words: odd, conflict free).  The backward writes its 45 gradients into the same LDS row and the block is stored back
// coalesced.
__device__ __forceinline__ void stage_rows_in(float* __restrict__ lds, const float* __restrict__ g, int64_t first_float,
                                              int n_floats) {
    // first_float is a multiple of 4 (256 rows per workgroup); n_floats may have a tail at the last workgroup.
    // A thread copies ~12 float4 (45 floats per row / 4): issued in batches of STAGE_BATCH loads BEFORE the first LDS store --
    // as a plain loop the compiler emitted load, s_waitcnt vmcnt(0), store per trip, i.e. 12 dependent memory round trips.
    constexpr int STAGE_BATCH = 6;
    const float4* g4 = reinterpret_cast<const float4*>(g + first_float);
    float4* l4 = reinterpret_cast<float4*>(lds);
    const int n4 = n_floats >> 2;
    const int T = blockDim.x;
    for (int e0 = threadIdx.x; e0 < n4; e0 += STAGE_BATCH * T) {
        float4 a[STAGE_BATCH];
#pragma unroll
        // clamped, unconditional loads AND stores (a trip past the end re-copies the last float4: same value, same place):
        // with a bounds test on the store the compiler sinks each load into its branch and the clause is gone again
        for (int u = 0; u < STAGE_BATCH; u++) a[u] = g4[min(e0 + u * T, n4 - 1)];
#pragma unroll
        for (int u = 0; u < STAGE_BATCH; u++) l4[min(e0 + u * T, n4 - 1)] = a[u];
    }
    for (int e = (n4 << 2) + threadIdx.x; e < n_floats; e += blockDim.x) lds[e] = g[first_float + e];
}
__device__ __forceinline__ void stage_rows_out(const float* __restrict__ lds, float* __restrict__ g, int64_t first_float,
                                               int n_floats) {
    float4* g4 = reinterpret_cast<float4*>(g + first_float);
    const float4* l4 = reinterpret_cast<const float4*>(lds);
    const int n4 = n_floats >> 2;
    for (int e = threadIdx.x; e < n4; e += blockDim.x) g4[e] = l4[e];
    for (int e = (n4 << 2) + threadIdx.x; e < n_floats; e += blockDim.x) g[first_float + e] = lds[e];
}

// What the forward writes per Gaussian, and what the binning's block-level passes need of it afterwards.
struct FwdOut { int32_t* radii; float* means2d; float* depths; float* conics; float* colors; float* opac; float4* recs; };
struct FwdBox { int n_tiles, n_groups, vis; float mx, my; int r; };

// The forward of ONE Gaussian from its parameters in registers (the SH-rest row behind `cf`: global memory in the forward
// kernel, the workgroup's LDS tile in the backward kernel's next-iteration tail): projection, SH colour, record, the backward
// rasterizer's zero-fill, the binning's per-Gaussian counts.  One body for both callers.
template <int DEG>
__device__ __forceinline__ FwdBox preprocess_fwd_gaussian(const FusedIn& in, int i, const float (&p)[3], const float (&q)[4],
                                                          const float (&logs)[3], const float (&dc)[3], const float* cf,
                                                          float opac_logit, const FwdOut& w, const BinCountOut& cnt,
                                                          const ZeroGrads& zg) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    FwdBox bx = {0, 0, 0, 0.f, 0.f, 0};
    Cam cam;
    cam_from_arrays(in.viewmat, in.Kmat, in.W, in.H, cam);
    const float s[3] = {expf(logs[0]), expf(logs[1]), expf(logs[2])};
    Proj o = project_gaussian(cam, p, q, s, in.eps2d, in.near_plane, in.far_plane, in.radius_clip);
    if (in.max_radii > 0) o.radius = min(o.radius, in.max_radii);
    w.radii[i] = o.radius;
    *reinterpret_cast<float2*>(w.means2d + 2 * (size_t)i) = make_float2(o.mx, o.my);
    w.depths[i] = o.z;
    w.conics[3 * i] = o.ca; w.conics[3 * i + 1] = o.cb; w.conics[3 * i + 2] = o.cc;
    float r = 0.f, g = 0.f, b = 0.f;
    if (o.radius > 0) {
        const float dx = p[0] - in.cam_pos[0], dy = p[1] - in.cam_pos[1], dz = p[2] - in.cam_pos[2];
        const float inorm = rsqrtf(dx * dx + dy * dy + dz * dz);
        float Y[NB];
        sh_basis<DEG>(dx * inorm, dy * inorm, dz * inorm, Y);
        r = Y[0] * dc[0]; g = Y[0] * dc[1]; b = Y[0] * dc[2];
        // (forward kernel: direct reads -- only visible Gaussians need their row here, staging all 256 through LDS measured slower)
#pragma unroll
        for (int k = 1; k < NB; k++) {
            r += Y[k] * cf[3 * (k - 1)]; g += Y[k] * cf[3 * (k - 1) + 1]; b += Y[k] * cf[3 * (k - 1) + 2];
        }
        r = fmaxf(r + 0.5f, 0.f); g = fmaxf(g + 0.5f, 0.f); b = fmaxf(b + 0.5f, 0.f);
    }
    *reinterpret_cast<float4*>(w.colors + 4 * (size_t)i) = make_float4(r, g, b, o.z);
    const float op = 1.f / (1.f + expf(-opac_logit));
    w.opac[i] = op;
    if (w.recs) pack_record(o, r, g, b, op, w.recs + 3 * (size_t)i);
    if (zg.v_colors) {  // the backward rasterizer's accumulators of this Gaussian (its zero-fill launch, folded in)
        *reinterpret_cast<float4*>(zg.v_colors + 4 * (size_t)i) = make_float4(0.f, 0.f, 0.f, 0.f);
        zg.v_conics[3 * i] = 0.f; zg.v_conics[3 * i + 1] = 0.f; zg.v_conics[3 * i + 2] = 0.f;
        *reinterpret_cast<float2*>(zg.v_means2d + 2 * (size_t)i) = make_float2(0.f, 0.f);
        zg.v_opacities[i] = 0.f;
    }
    if (cnt.tiles_per_gauss) {
        // first pass of the tile binning (count_kernel of splat_bin.hip) on the values still in registers
        if (o.radius > 0) {
            tile_group_count(o.mx, o.my, o.radius, cnt.tile_size, cnt.tw, cnt.th, bx.n_tiles, bx.n_groups); bx.vis = 1;
            bx.mx = o.mx; bx.my = o.my; bx.r = o.radius;
        }
        cnt.tiles_per_gauss[i] = bx.n_tiles;
        cnt.groups_per_gauss[i] = bx.n_groups;
    }
    return bx;
}

// Superblock binning, histogram pass (splat_bin.hpp) for the Gaussians of one workgroup (any size that divides BIN_BLOCK; its
// first Gaussian names the superblock): the (Gaussian, tile) pairs counted per tile in LDS (hist[SB_MAX_TILES]), the non-zero bins
// added to the count table of the superblock; Gaussians per backward class likewise (khist[BWD_KEYS]).  Every thread arrives.
__device__ __forceinline__ void sb_histogram_block(const BinCountOut& cnt, uint32_t* hist, int* khist, const FwdBox& bx,
                                                   int first_gaussian) {
    const int nt = cnt.tw * cnt.th;
    for (int b = threadIdx.x; b < nt; b += blockDim.x) hist[b] = 0;
    if (threadIdx.x < BWD_KEYS) khist[threadIdx.x] = 0;
    __syncthreads();
    unsigned y0 = 0;
    if (bx.n_tiles > 0) {
        const TileBox tb = tile_bbox(bx.mx, bx.my, bx.r, cnt.tile_size, cnt.tw, cnt.th);
        y0 = tb.y0;
        for (uint32_t ty = tb.y0; ty < tb.y1; ty++)
            for (uint32_t tx = tb.x0; tx < tb.x1; tx++) atomicAdd(&hist[ty * (uint32_t)cnt.tw + tx], 1u);
    }
    if (bx.r > 0) atomicAdd(&khist[bwd_key(bx.r, bx.n_tiles, y0, cnt.th)], 1);
    __syncthreads();
    const int sb = (first_gaussian / BIN_BLOCK) >> cnt.sb.sb_shift;
    for (int b = threadIdx.x; b < nt; b += blockDim.x) {
        const uint32_t c = hist[b];
        if (c) atomicAdd(&cnt.sb.C[(size_t)b * SB_MAX + sb], c);
    }
    if (threadIdx.x < BWD_KEYS) {
        const int c = khist[threadIdx.x];
        if (c) atomicAdd(&cnt.sb.cls_count[threadIdx.x * SB_MAX + sb], c);
    }
}

template <int DEG>
__global__ __launch_bounds__(256) void preprocess_fwd_kernel(FusedIn in, FwdOut w, BinCountOut cnt, ZeroGrads zg, LaunchStamp stamp) {
    StampScope timed(stamp);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    FwdBox bx = {0, 0, 0, 0.f, 0.f, 0};
    if (i < in.N) {
        const float p[3] = {in.means[3 * i], in.means[3 * i + 1], in.means[3 * i + 2]};
        const float4 q4 = *reinterpret_cast<const float4*>(in.quats + 4 * (size_t)i);
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float logs[3] = {in.log_scales[3 * i], in.log_scales[3 * i + 1], in.log_scales[3 * i + 2]};
        const float dc[3] = {in.sh_dc[3 * i], in.sh_dc[3 * i + 1], in.sh_dc[3 * i + 2]};
        bx = preprocess_fwd_gaussian<DEG>(in, i, p, q, logs, dc, in.sh_rest + (size_t)i * (in.K - 1) * 3, in.opac_logit[i], w, cnt, zg);
    }
    if (cnt.tiles_per_gauss) bin_block_sums(cnt, bx.n_tiles, bx.n_groups, bx.vis);  // (uniform branch: every thread arrives)
    if (cnt.sb.C) {
        __shared__ uint32_t hist[SB_MAX_TILES];
        __shared__ int khist[BWD_KEYS];
        sb_histogram_block(cnt, hist, khist, bx, (int)(blockIdx.x * blockDim.x));
    }
}

// FUSE_ADAM: the Adam step of the sh_rest tensor (45 of the 59 parameters) happens here, on the LDS tiles, instead of in
// adam_kernel: the gradient never travels to HBM and back and the parameter is read once (-131 of 394 MB per iteration at
// 243 k Gaussians).  Same adam_update, same operands -> bit-identical to the separate step.
struct FusedAdam {
    float* param;       // sh_rest, updated in place
    float* exp_avg;     // its Adam state
    float* exp_avg_sq;
    AdamScalars sc;
    // optionally also the five small tensors (means, log_scales, quats, sh_dc, opac_logit: 14 floats per Gaussian), stepped
    // by the thread that owns the Gaussian right after it has their gradients in registers -> no adam_kernel launch at all
    int small;
    float* sp[5];
    float* sm[5];
    float* sv[5];
    float sstep[5];     // lr_k / (1 - beta1^t)
};

// NEXT ITERATION'S FORWARD in the tail of this iteration's backward (gps_splat_step::next_viewmat): the thread that has just
// stepped a Gaussian's 59 parameters holds them in registers / the LDS tile -- the preprocessing forward of the next optimise
// iteration (another camera of the same update) runs right there instead of re-reading them in a launch of its own: one launch
// and 68 N + 217 Nv bytes of reads less per iteration.  viewmat == nullptr: off.
struct NextFwd {
    const float* viewmat;     // the NEXT camera (device arrays, as FusedIn's)
    const float* Kmat;
    const float* cam_pos;
    int max_radii;
    float near_plane, far_plane, radius_clip;
    FwdOut out;               // the per-Gaussian outputs of the forward (this iteration's values have been read by then)
    BinCountOut cnt;          // the superblock binning's count targets (tables zero: this iteration's scan has cleared them)
};

// The five small tensors (3 + 3 + 4 + 3 + 1 = 14 floats per Gaussian), stepped by the thread that owns the Gaussian: all 42
// loads (parameter, both moments) are issued before the first update -- row by row with the stores in between, every
// component was its own memory round trip (the compiler cannot move a load of m[c+1] above the store of p[c]).
__device__ __forceinline__ void adam_small_rows(const FusedAdam& ad, int i, const float* const g[5], float (&P)[14]) {
    constexpr int L[5] = {3, 3, 4, 3, 1};
    float M[14], V[14];   // (P: the stepped parameters, means[3] | log_scales[3] | quats[4] | sh_dc[3] | opac_logit, for the caller)
    int o = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float* p = ad.sp[k] + (size_t)i * L[k];
#pragma unroll
        for (int c = 0; c < L[k]; c++, o++) { P[o] = p[c]; M[o] = 0.f; V[o] = 0.f; }
    }
    if (!ad.sc.fresh) {   // (uniform; step 1: the moments are zero by definition, AdamScalars::fresh)
        o = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float* m = ad.sm[k] + (size_t)i * L[k];
            const float* v = ad.sv[k] + (size_t)i * L[k];
#pragma unroll
            for (int c = 0; c < L[k]; c++, o++) { M[o] = m[c]; V[o] = v[c]; }
        }
    }
    o = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        AdamScalars sc = ad.sc;
        sc.step_size = ad.sstep[k];
#pragma unroll
        for (int c = 0; c < L[k]; c++, o++) adam_update(sc, g[k][c], M[o], V[o], P[o]);
    }
    o = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float* p = ad.sp[k] + (size_t)i * L[k];
        float* m = ad.sm[k] + (size_t)i * L[k];
        float* v = ad.sv[k] + (size_t)i * L[k];
#pragma unroll
        for (int c = 0; c < L[k]; c++, o++) { p[c] = P[o]; m[c] = M[o]; v[c] = V[o]; }
    }
}

// (fuse is a RUN-TIME flag on purpose: one instantiation per degree -> the gradient math is the same machine code in both
// modes, so the two paths agree bit for bit; as a template parameter the compiler contracted the SH polynomials differently.)
template <int DEG>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(FusedIn in, FusedAdam ad, const int32_t* __restrict__ radii,
                                                             const float* __restrict__ conics,
                                                             const float* __restrict__ v_means2d,
                                                             const float* __restrict__ v_conics,
                                                             const float* __restrict__ v_colors,
                                                             const float* __restrict__ v_opac,
                                                             const float4* __restrict__ v_rows,
                                                             float* __restrict__ v_means,
                                                             float* __restrict__ v_log_scales,
                                                             float* __restrict__ v_quats,
                                                             float* __restrict__ v_opac_logit,
                                                             float* __restrict__ v_sh_dc,
                                                             float* __restrict__ v_sh_rest, NextFwd nf, LaunchStamp stamp) {
    StampScope timed(stamp);
    // sh_tile: this workgroup's sh_rest rows.  Without FUSE_ADAM their gradients overwrite them in place; with it the
    // gradients go to a second tile so that parameter and gradient are both at hand for the update.
    extern __shared__ float sh_tile[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (in.K - 1) * 3;
    const int rows_here = min((int)blockDim.x, in.N - (int)(blockIdx.x * blockDim.x));
    const int64_t tile_first = (int64_t)blockIdx.x * blockDim.x * row;
    if (in.K > 1) {
        stage_rows_in(sh_tile, in.sh_rest, tile_first, rows_here * row);
        __syncthreads();
    }
    const bool live = i < in.N;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float vp[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    float vdc[3] = {0.f, 0.f, 0.f};
    const bool FUSE_ADAM = ad.param != nullptr;
    float* g_tile = FUSE_ADAM ? sh_tile + blockDim.x * row : sh_tile;
    float* vrest = g_tile + threadIdx.x * row;
    const bool vis = live && radii[i] > 0;
    // (strip rows: a Gaussian the rasterizer did not touch has no row -- and no opacity gradient; four arrays: every element is
    // an input, as the operator-level entry point defines it)
    float v_opac_i = (!v_rows && live) ? v_opac[i] : 0.f;
    int written = 0;  // number of sh_rest bands written below
    if (vis) {
        Cam cam;
        cam_from_arrays(in.viewmat, in.Kmat, in.W, in.H, cam);
        const float p[3] = {in.means[3 * i], in.means[3 * i + 1], in.means[3 * i + 2]};
        const float4 q4 = *reinterpret_cast<const float4*>(in.quats + 4 * (size_t)i);
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float s[3] = {expf(in.log_scales[3 * i]), expf(in.log_scales[3 * i + 1]), expf(in.log_scales[3 * i + 2])};
        const float conic[3] = {conics[3 * i], conics[3 * i + 1], conics[3 * i + 2]};
        // the rasterizer's gradients of this Gaussian: one 48-byte row {colors[4], conics[3], means2d[2], opacity} from the strip
        // backward, or the four arrays of the operator-level entry points
        float vm2[2], vc[3];
        float4 vcol;
        if (v_rows) {
            const float4 r0 = v_rows[3 * (size_t)i], r1 = v_rows[3 * (size_t)i + 1], r2 = v_rows[3 * (size_t)i + 2];
            vcol = r0;
            vc[0] = r1.x; vc[1] = r1.y; vc[2] = r1.z;
            vm2[0] = r1.w; vm2[1] = r2.x;
            v_opac_i = r2.y;
        } else {
            vm2[0] = v_means2d[2 * i]; vm2[1] = v_means2d[2 * i + 1];
            vc[0] = v_conics[3 * i]; vc[1] = v_conics[3 * i + 1]; vc[2] = v_conics[3 * i + 2];
            vcol = *reinterpret_cast<const float4*>(v_colors + 4 * (size_t)i);
        }
        // depth channel of `colors` is the projection's depth output (raw_gs_model.cpp:286)
        project_gaussian_vjp(cam, p, q, s, conic, vm2, vcol.w, vc, vp, vq, vs);
        // SH: recompute the un-clamped colour to evaluate clamp_min's mask (grad passes where sh + 0.5 >= 0)
        const float dx = p[0] - in.cam_pos[0], dy = p[1] - in.cam_pos[1], dz = p[2] - in.cam_pos[2];
        const float inorm = rsqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
        float Y[NB];
        sh_basis<DEG>(x, y, z, Y);
        const float* cf = sh_tile + threadIdx.x * row;  // read completely before vrest (the same row) is written
        float c0 = Y[0] * in.sh_dc[3 * i], c1 = Y[0] * in.sh_dc[3 * i + 1], c2 = Y[0] * in.sh_dc[3 * i + 2];
#pragma unroll
        for (int k = 1; k < NB; k++) {
            c0 += Y[k] * cf[3 * (k - 1)]; c1 += Y[k] * cf[3 * (k - 1) + 1]; c2 += Y[k] * cf[3 * (k - 1) + 2];
        }
        const float vr = (c0 + 0.5f >= 0.f) ? vcol.x : 0.f;
        const float vg = (c1 + 0.5f >= 0.f) ? vcol.y : 0.f;
        const float vb = (c2 + 0.5f >= 0.f) ? vcol.z : 0.f;
        vdc[0] = Y[0] * vr; vdc[1] = Y[0] * vg; vdc[2] = Y[0] * vb;
        if (DEG >= 1) {
            float dX[NB], dY[NB], dZ[NB];
            sh_basis_grad<DEG>(x, y, z, dX, dY, dZ);
            float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
            for (int k = 1; k < NB; k++) {
                const float w = cf[3 * (k - 1)] * vr + cf[3 * (k - 1) + 1] * vg + cf[3 * (k - 1) + 2] * vb;
                gx += dX[k] * w; gy += dY[k] * w; gz += dZ[k] * w;
            }
            const float d = gx * x + gy * y + gz * z;
            // dirs = means - cam_T  ->  v_means += v_dirs
            vp[0] += (gx - d * x) * inorm; vp[1] += (gy - d * y) * inorm; vp[2] += (gz - d * z) * inorm;
        }
#pragma unroll
        for (int k = 1; k < NB; k++) {
            vrest[3 * (k - 1)] = Y[k] * vr; vrest[3 * (k - 1) + 1] = Y[k] * vg; vrest[3 * (k - 1) + 2] = Y[k] * vb;
        }
        written = NB - 1;
        // scales = exp(log_scales)
        vs[0] *= s[0]; vs[1] *= s[1]; vs[2] *= s[2];
    }
    if (live)
        for (int k = written; k < in.K - 1; k++) { vrest[3 * k] = 0.f; vrest[3 * k + 1] = 0.f; vrest[3 * k + 2] = 0.f; }
    if (in.K > 1) {
        __syncthreads();
        if (v_sh_rest) stage_rows_out(g_tile, v_sh_rest, tile_first, rows_here * row);
        if (FUSE_ADAM) {
            const int n = rows_here * row, n4 = n >> 2;  // tile_first is a multiple of 4 floats
            const float4* p4 = reinterpret_cast<const float4*>(sh_tile);
            const float4* g4 = reinterpret_cast<const float4*>(g_tile);
            float4* gp = reinterpret_cast<float4*>(ad.param + tile_first);
            float4* gm = reinterpret_cast<float4*>(ad.exp_avg + tile_first);
            float4* gv = reinterpret_cast<float4*>(ad.exp_avg_sq + tile_first);
            // ADAM_BATCH trips' worth of moment loads in flight before the first dependent use (as a plain loop every trip
            // was load, load, s_waitcnt vmcnt(0), update, 3 stores: ~12 dependent round trips per workgroup)
            constexpr int ADAM_BATCH = 4;
            const int T = blockDim.x;
            for (int e0 = threadIdx.x; e0 < n4; e0 += ADAM_BATCH * T) {
                float4 m[ADAM_BATCH], v[ADAM_BATCH];
#pragma unroll
                for (int u = 0; u < ADAM_BATCH; u++) {
                    const int e = min(e0 + u * T, n4 - 1);  // unconditional loads: one clause
                    m[u] = make_float4(0.f, 0.f, 0.f, 0.f); v[u] = m[u];
                    if (!ad.sc.fresh) { m[u] = gm[e]; v[u] = gv[e]; }   // (uniform: step 1 starts from zero moments, AdamScalars::fresh)
                }
#pragma unroll
                for (int u = 0; u < ADAM_BATCH; u++) {
                    const int e = e0 + u * T;
                    if (e < n4) {
                        float4 p = p4[e];
                        const float4 g = g4[e];
                        adam_update(ad.sc, g.x, m[u].x, v[u].x, p.x); adam_update(ad.sc, g.y, m[u].y, v[u].y, p.y);
                        adam_update(ad.sc, g.z, m[u].z, v[u].z, p.z); adam_update(ad.sc, g.w, m[u].w, v[u].w, p.w);
                        gm[e] = m[u]; gv[e] = v[u]; gp[e] = p;
                        if (nf.viewmat) reinterpret_cast<float4*>(sh_tile)[e] = p;   // (the next forward reads the stepped row from LDS)
                    }
                }
            }
            for (int e = (n4 << 2) + threadIdx.x; e < n; e += blockDim.x) {
                float p = sh_tile[e], m = 0.f, v = 0.f;
                if (!ad.sc.fresh) { m = ad.exp_avg[tile_first + e]; v = ad.exp_avg_sq[tile_first + e]; }
                adam_update(ad.sc, g_tile[e], m, v, p);
                ad.exp_avg[tile_first + e] = m; ad.exp_avg_sq[tile_first + e] = v; ad.param[tile_first + e] = p;
                if (nf.viewmat) sh_tile[e] = p;
            }
        }
    }
    if (!live && !nf.viewmat) return;
    float P14[14];
    if (live) {
        // opac = sigmoid(logit): receives gradient for every Gaussian the rasterizer touched (0 otherwise)
        const float o = 1.f / (1.f + expf(-in.opac_logit[i]));
        const float vo = v_opac_i * o * (1.f - o);
        if (v_means) {  // gradient outputs (optional when the small tensors are stepped below)
            v_means[3 * i] = vp[0]; v_means[3 * i + 1] = vp[1]; v_means[3 * i + 2] = vp[2];
            v_log_scales[3 * i] = vs[0]; v_log_scales[3 * i + 1] = vs[1]; v_log_scales[3 * i + 2] = vs[2];
            *reinterpret_cast<float4*>(v_quats + 4 * (size_t)i) = make_float4(vq[0], vq[1], vq[2], vq[3]);
            v_sh_dc[3 * i] = vdc[0]; v_sh_dc[3 * i + 1] = vdc[1]; v_sh_dc[3 * i + 2] = vdc[2];
            v_opac_logit[i] = vo;
        }
        if (ad.small) {  // every read of this Gaussian's parameters is done: step them in place
            const float* const gs[5] = {vp, vs, vq, vdc, &vo};
            adam_small_rows(ad, i, gs, P14);
        }
    }
    if (!nf.viewmat) return;
    // ---- the next iteration's preprocessing forward (launcher: only with all six tensors stepped above and K > 1) ----
    __syncthreads();   // the tile holds the stepped sh_rest rows of every thread's Gaussian; the gradient tile is free
    FusedIn in2 = in;
    in2.viewmat = nf.viewmat; in2.Kmat = nf.Kmat; in2.cam_pos = nf.cam_pos;
    in2.max_radii = nf.max_radii; in2.near_plane = nf.near_plane; in2.far_plane = nf.far_plane; in2.radius_clip = nf.radius_clip;
    FwdBox bx = {0, 0, 0, 0.f, 0.f, 0};
    if (live) {
        const float p[3] = {P14[0], P14[1], P14[2]}, logs[3] = {P14[3], P14[4], P14[5]};
        const float q[4] = {P14[6], P14[7], P14[8], P14[9]}, dc[3] = {P14[10], P14[11], P14[12]};
        const ZeroGrads none = {};
        bx = preprocess_fwd_gaussian<DEG>(in2, i, p, q, logs, dc, sh_tile + threadIdx.x * row, P14[13], nf.out, nf.cnt, none);
    }
    // the binning's histogram pass in the gradient tile's memory (SB_MAX_TILES words + the class keys: <= the tile, see the launcher)
    uint32_t* hist = reinterpret_cast<uint32_t*>(g_tile);
    sb_histogram_block(nf.cnt, hist, reinterpret_cast<int*>(hist + SB_MAX_TILES), bx, (int)(blockIdx.x * blockDim.x));
}

}  // namespace

namespace gps {

// launcher shared by gps_gauss_preprocess_bwd and gps_splat_train_step.  adam_param != nullptr: fuse the Adam step of
// sh_rest (adam_param aliases sh_rest; v_sh_rest may then be NULL = do not write that gradient at all).
int preprocess_bwd_launch(int N, int K, int sh_degree, const float* means, const float* log_scales, const float* quats,
                          const float* opac_logit, const float* sh_dc, const float* sh_rest, const float* viewmat,
                          const float* Kmat, const float* cam_pos, int width, int height, float eps2d,
                          const int32_t* radii, const float* conics, const float* v_means2d, const float* v_conics,
                          const float* v_colors, const float* v_opacities, float* v_means, float* v_log_scales,
                          float* v_quats, float* v_opac_logit, float* v_sh_dc, float* v_sh_rest, float* adam_param,
                          float* adam_m, float* adam_v, AdamScalars sc, const gps_adam_segment* small5,
                          const float* small_step, gps_stream stream, const float* v_rows, const NextForward* next) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0 && sh_degree >= 0 && sh_degree <= 4 && K >= sh_num_bases(sh_degree));
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(means && log_scales && quats && opac_logit && sh_dc && (K == 1 || sh_rest) && viewmat && Kmat && cam_pos);
    GPS_REQUIRE(radii && conics && (v_rows || (v_means2d && v_conics && v_colors && v_opacities)));
    const bool grads_out = v_means != nullptr;
    GPS_REQUIRE(grads_out ? (v_log_scales && v_quats && v_opac_logit && v_sh_dc) : (small5 != nullptr));
    const bool fuse = adam_param != nullptr && K > 1;
    GPS_REQUIRE(!fuse || (adam_param == sh_rest && adam_m && adam_v));
    GPS_REQUIRE(fuse || K == 1 || v_sh_rest);
    FusedIn in = {means, log_scales, quats, opac_logit, sh_dc, sh_rest, viewmat, Kmat, cam_pos, N, K, width, height,
                  0, eps2d, 0.f, 0.f, 0.f};
    FusedAdam ad = {};
    ad.param = adam_param; ad.exp_avg = adam_m; ad.exp_avg_sq = adam_v; ad.sc = sc;
    if (small5) {
        // order: means, log_scales, quats, sh_dc, opac_logit -- and they must be THE parameters this kernel reads
        GPS_REQUIRE(small5[0].param == means && small5[1].param == log_scales && small5[2].param == quats &&
                    small5[3].param == sh_dc && small5[4].param == opac_logit && small_step);
        ad.small = 1;
        for (int k = 0; k < 5; k++) {
            GPS_REQUIRE(small5[k].exp_avg && small5[k].exp_avg_sq);
            ad.sp[k] = small5[k].param; ad.sm[k] = small5[k].exp_avg; ad.sv[k] = small5[k].exp_avg_sq;
            ad.sstep[k] = small_step[k];
        }
    }
    // parameter (+ gradient) rows of the workgroup's Gaussians live in LDS: halve the workgroup until they fit 64 KB
    // (K = 16: 256 / 128 threads; K = 25, SH degree 4: 128 / 64 threads).  rows * threads stays a multiple of 4 floats.
    int threads = fuse ? GPS_FUSED_ADAM_THREADS : 256;
    auto lds_bytes = [&](int t) { return (size_t)t * (K - 1) * 3 * sizeof(float) * (fuse ? 2 : 1); };
    while (threads > 64 && lds_bytes(threads) > 65536) threads >>= 1;
    dim3 g(gps_div_up(N, threads)), b(threads);
    const size_t lds = lds_bytes(threads);
    GPS_REQUIRE(lds <= 65536);
    NextFwd nf = {};
    if (next) {
        // only behind the fully fused step (every parameter stepped in this kernel), with the superblock binning's count targets,
        // a workgroup that divides the binning's 256-Gaussian blocks and a gradient tile that holds the histogram
        GPS_REQUIRE(fuse && small5 && !grads_out && next->viewmat && next->Kmat && next->cam_pos && next->count && next->count->sb.C);
        GPS_REQUIRE(next->radii && next->means2d && next->depths && next->conics && next->colors && next->opacities && next->records);
        GPS_REQUIRE(BIN_BLOCK % threads == 0 && lds / 2 >= (size_t)(SB_MAX_TILES + BWD_KEYS) * 4);
        nf.viewmat = next->viewmat; nf.Kmat = next->Kmat; nf.cam_pos = next->cam_pos;
        nf.max_radii = next->max_gs_radii; nf.near_plane = next->near_plane; nf.far_plane = next->far_plane; nf.radius_clip = next->radius_clip;
        nf.out = {next->radii, next->means2d, next->depths, next->conics, next->colors, next->opacities, reinterpret_cast<float4*>(next->records)};
        nf.cnt = *next->count;
    }
    hipStream_t s = (hipStream_t)stream;
#define GPS_BWD(D)                                                                                                 \
    launch_kernel(TK_PREPROCESS_BWD, next ? 1 : 0, preprocess_bwd_kernel<D>, g, b, lds, s, in, ad, radii, conics,         \
                  v_means2d, v_conics, v_colors, v_opacities, reinterpret_cast<const float4*>(v_rows), v_means,           \
                  v_log_scales, v_quats, v_opac_logit, v_sh_dc, v_sh_rest, nf)
    switch (sh_degree) {
        case 0: GPS_BWD(0); break;
        case 1: GPS_BWD(1); break;
        case 2: GPS_BWD(2); break;
        case 3: GPS_BWD(3); break;
        default: GPS_BWD(4); break;
    }
#undef GPS_BWD
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int preprocess_fwd_launch(int N, int K, int sh_degree, const float* means, const float* log_scales, const float* quats,
                          const float* opac_logit, const float* sh_dc, const float* sh_rest, const float* viewmat,
                          const float* Kmat, const float* cam_pos, int width, int height, float eps2d, float near_plane,
                          float far_plane, float radius_clip, int max_gs_radii, int32_t* radii, float* means2d, float* depths,
                          float* conics, float* colors, float* opacities, float* records, const BinCountOut* count,
                          const ZeroGrads* zero, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0 && sh_degree >= 0 && sh_degree <= 4 && K >= sh_num_bases(sh_degree));
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(means && log_scales && quats && opac_logit && sh_dc && (K == 1 || sh_rest) && viewmat && Kmat && cam_pos);
    GPS_REQUIRE(radii && means2d && depths && conics && colors && opacities);
    float4* recs = reinterpret_cast<float4*>(records);
    FusedIn in = {means, log_scales, quats, opac_logit, sh_dc, sh_rest, viewmat, Kmat, cam_pos, N, K, width, height,
                  max_gs_radii, eps2d, near_plane, far_plane, radius_clip};
    BinCountOut cnt = {};
    if (count) cnt = *count;
    ZeroGrads zg = {};
    if (zero && zero->v_colors) { zg = *zero; GPS_REQUIRE(zg.v_means2d && zg.v_conics && zg.v_colors && zg.v_opacities); }
    static_assert(BIN_BLOCK == 256, "the binning's per-block sums are per preprocessing workgroup");
    dim3 g(gps_div_up(N, 256)), b(256);
    hipStream_t s = (hipStream_t)stream;
    const FwdOut w = {radii, means2d, depths, conics, colors, opacities, recs};
    switch (sh_degree) {
        case 0: launch_kernel(TK_PREPROCESS_FWD, 0, preprocess_fwd_kernel<0>, g, b, 0, s, in, w, cnt, zg); break;
        case 1: launch_kernel(TK_PREPROCESS_FWD, 0, preprocess_fwd_kernel<1>, g, b, 0, s, in, w, cnt, zg); break;
        case 2: launch_kernel(TK_PREPROCESS_FWD, 0, preprocess_fwd_kernel<2>, g, b, 0, s, in, w, cnt, zg); break;
        case 3: launch_kernel(TK_PREPROCESS_FWD, 0, preprocess_fwd_kernel<3>, g, b, 0, s, in, w, cnt, zg); break;
        default: launch_kernel(TK_PREPROCESS_FWD, 0, preprocess_fwd_kernel<4>, g, b, 0, s, in, w, cnt, zg); break;
    }
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // namespace gps

extern "C" {

int gps_gauss_preprocess_fwd(int N, int K, int sh_degree, const float* means, const float* log_scales,
                             const float* quats, const float* opac_logit, const float* sh_dc, const float* sh_rest,
                             const float* viewmat, const float* Kmat, const float* cam_pos, int width, int height,
                             float eps2d, float near_plane, float far_plane, float radius_clip, int max_gs_radii,
                             int32_t* radii, float* means2d, float* depths, float* conics, float* colors,
                             float* opacities, float* records, gps_stream stream) {
    return gps::preprocess_fwd_launch(N, K, sh_degree, means, log_scales, quats, opac_logit, sh_dc, sh_rest, viewmat, Kmat,
                                      cam_pos, width, height, eps2d, near_plane, far_plane, radius_clip, max_gs_radii, radii,
                                      means2d, depths, conics, colors, opacities, records, nullptr, nullptr, stream);
}

int gps_gauss_preprocess_bwd(int N, int K, int sh_degree, const float* means, const float* log_scales,
                             const float* quats, const float* opac_logit, const float* sh_dc, const float* sh_rest,
                             const float* viewmat, const float* Kmat, const float* cam_pos, int width, int height,
                             float eps2d, const int32_t* radii, const float* conics, const float* v_means2d,
                             const float* v_conics, const float* v_colors, const float* v_opacities, float* v_means,
                             float* v_log_scales, float* v_quats, float* v_opac_logit, float* v_sh_dc,
                             float* v_sh_rest, gps_stream stream) {
    GPS_REQUIRE(K == 1 || v_sh_rest);
    return gps::preprocess_bwd_launch(N, K, sh_degree, means, log_scales, quats, opac_logit, sh_dc, sh_rest, viewmat, Kmat,
                                      cam_pos, width, height, eps2d, radii, conics, v_means2d, v_conics, v_colors,
                                      v_opacities, v_means, v_log_scales, v_quats, v_opac_logit, v_sh_dc, v_sh_rest,
                                      nullptr, nullptr, nullptr, gps::AdamScalars{}, nullptr, nullptr, stream);
}

int gps_gauss_preprocess_bwd_adam(int N, int K, int sh_degree, const float* means, const float* log_scales,
                                  const float* quats, const float* opac_logit, const float* sh_dc, float* sh_rest,
                                  const float* viewmat, const float* Kmat, const float* cam_pos, int width, int height,
                                  float eps2d, const int32_t* radii, const float* conics, const float* v_means2d,
                                  const float* v_conics, const float* v_colors, const float* v_opacities, float* v_means,
                                  float* v_log_scales, float* v_quats, float* v_opac_logit, float* v_sh_dc,
                                  float* v_sh_rest, float* exp_avg, float* exp_avg_sq, double lr,
                                  const gps_adam_segment* small5, double beta1, double beta2, double eps, int step,
                                  gps_stream stream) {
    GPS_REQUIRE(K > 1 && sh_rest && exp_avg && exp_avg_sq && step >= 1);
    float sstep[5] = {0, 0, 0, 0, 0};
    if (small5)
        for (int k = 0; k < 5; k++) sstep[k] = gps::adam_scalars(small5[k].lr, beta1, beta2, eps, step).step_size;
    return gps::preprocess_bwd_launch(N, K, sh_degree, means, log_scales, quats, opac_logit, sh_dc, sh_rest, viewmat, Kmat,
                                      cam_pos, width, height, eps2d, radii, conics, v_means2d, v_conics, v_colors,
                                      v_opacities, v_means, v_log_scales, v_quats, v_opac_logit, v_sh_dc, v_sh_rest,
                                      sh_rest, exp_avg, exp_avg_sq, gps::adam_scalars(lr, beta1, beta2, eps, step), small5,
                                      small5 ? sstep : nullptr, stream);
}

}  // extern "C"
