// Per-Gaussian device math shared by the op-level kernels and the fused
// model-level kernels: world->camera, quat/scale -> covariance, pinhole
// projection with the 0.3*tan(fov) clamp, 2x2 conic, radius, and their VJPs;
// real spherical harmonics up to degree 4 with analytic direction gradients.
//
// What is computed follows the reference operators
//   gsplat/rasterizer/fully_fused_projection_{fwd,bwd}.cu, utils.cuh,
//   spherical_harmonics.cuh
// (see oracle/splat_oracle.c for the line-level restatement); how it is
// organised here is register-resident scalar code with symmetric-matrix
// storage (6 floats), no generic matrix classes.
#pragma once
#include "common.hpp"

namespace gps {

struct Cam {
    float R[9];  // world->camera rotation, row-major
    float t[3];
    float fx, fy, cx, cy;
    float lim_x_pos, lim_x_neg, lim_y_pos, lim_y_neg;  // fov clamp limits (utils.cuh:269-279)
    int W, H;
};

__host__ __device__ inline void cam_from_arrays(const float* viewmat, const float* K, int W, int H, Cam& c) {
    c.R[0] = viewmat[0]; c.R[1] = viewmat[1]; c.R[2] = viewmat[2];
    c.R[3] = viewmat[4]; c.R[4] = viewmat[5]; c.R[5] = viewmat[6];
    c.R[6] = viewmat[8]; c.R[7] = viewmat[9]; c.R[8] = viewmat[10];
    c.t[0] = viewmat[3]; c.t[1] = viewmat[7]; c.t[2] = viewmat[11];
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    c.W = W; c.H = H;
    float tan_fovx = 0.5f * (float)W / c.fx, tan_fovy = 0.5f * (float)H / c.fy;
    c.lim_x_pos = ((float)W - c.cx) / c.fx + 0.3f * tan_fovx;
    c.lim_x_neg = c.cx / c.fx + 0.3f * tan_fovx;
    c.lim_y_pos = ((float)H - c.cy) / c.fy + 0.3f * tan_fovy;
    c.lim_y_neg = c.cy / c.fy + 0.3f * tan_fovy;
}

// symmetric 3x3 stored as {xx, xy, xz, yy, yz, zz}
struct Sym3 { float xx, xy, xz, yy, yz, zz; };

__device__ __forceinline__ void quat_to_R(const float q[4], float R[9], float& inv_norm) {
#pragma clang fp contract(off)   // (inlined into several kernels: unfused mul / add give every copy the same bits, see project_gaussian)
    float w = q[0], x = q[1], y = q[2], z = q[3];
    inv_norm = rsqrtf(x * x + y * y + z * z + w * w);
    x *= inv_norm; y *= inv_norm; z *= inv_norm; w *= inv_norm;
    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (y2 + z2); R[1] = 2.f * (xy - wz);       R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz);       R[4] = 1.f - 2.f * (x2 + z2); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy);       R[7] = 2.f * (yz + wx);       R[8] = 1.f - 2.f * (x2 + y2);
}

// M = A * diag(s)  (3x3 row-major), cov = M M^T (symmetric)
__device__ __forceinline__ Sym3 outer_MMt(const float M[9]) {
#pragma clang fp contract(off)   // (inlined into several kernels: unfused mul / add give every copy the same bits, see project_gaussian)
    Sym3 c;
    c.xx = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    c.xy = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    c.xz = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    c.yy = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    c.yz = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    c.zz = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    return c;
}

struct Proj {
    float mx, my;        // means2d
    float z;             // camera depth
    float ca, cb, cc;    // conic
    int radius;          // 0 => culled
};

// camera-space covariance cov_c = (R_cam Rq S)(R_cam Rq S)^T computed as one
// 3x3 product T = R_cam * Rq then scaled columns.
__device__ __forceinline__ void gaussian_cov_cam(const Cam& cam, const float q[4], const float s[3], Sym3& cov_c) {
#pragma clang fp contract(off)   // (inlined into several kernels: unfused mul / add give every copy the same bits, see project_gaussian)
    float Rq[9], inv_norm;
    quat_to_R(q, Rq, inv_norm);
    float T[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            T[3 * r + c] = (cam.R[3 * r] * Rq[c] + cam.R[3 * r + 1] * Rq[3 + c] + cam.R[3 * r + 2] * Rq[6 + c]) * s[c];
    cov_c = outer_MMt(T);
}

__device__ __forceinline__ void persp_jacobian(const Cam& cam, float x, float y, float z, float& j00, float& j02,
                                               float& j11, float& j12, float& tx, float& ty) {
#pragma clang fp contract(off)   // (inlined into several kernels: unfused mul / add give every copy the same bits, see project_gaussian)
    float rz = 1.f / z, rz2 = rz * rz;
    tx = z * fminf(cam.lim_x_pos, fmaxf(-cam.lim_x_neg, x * rz));
    ty = z * fminf(cam.lim_y_pos, fmaxf(-cam.lim_y_neg, y * rz));
    j00 = cam.fx * rz; j11 = cam.fy * rz;
    j02 = -cam.fx * tx * rz2; j12 = -cam.fy * ty * rz2;
}

// Full forward projection of one Gaussian.  Returns radius==0 when culled.
// The forward projection (this function, its helpers and pack_record) is compiled WITHOUT fp contraction: it is inlined into the
// forward kernel, the operator-level kernel and the backward kernel's next-iteration tail (splat_fused.hip), and with contraction
// the compiler fused different mul / add pairs in each copy -- conics differed in the last bit between the stand-alone forward
// and the tail (tools/probe/prefetch_diff.py).  Plain IEEE operations give every copy the same bits (and are what the CPU oracle,
// built without FMA, computes).
__device__ __forceinline__ Proj project_gaussian(const Cam& cam, const float p[3], const float q[4], const float s[3],
                                                 float eps2d, float near_plane, float far_plane, float radius_clip) {
#pragma clang fp contract(off)   // (inlined into several kernels: unfused mul / add give every copy the same bits, see project_gaussian)
    Proj o; o.radius = 0; o.mx = o.my = o.z = o.ca = o.cb = o.cc = 0.f;
    float x = cam.R[0] * p[0] + cam.R[1] * p[1] + cam.R[2] * p[2] + cam.t[0];
    float y = cam.R[3] * p[0] + cam.R[4] * p[1] + cam.R[5] * p[2] + cam.t[1];
    float z = cam.R[6] * p[0] + cam.R[7] * p[1] + cam.R[8] * p[2] + cam.t[2];
    if (z < near_plane || z > far_plane) return o;
    Sym3 C;
    gaussian_cov_cam(cam, q, s, C);
    float j00, j02, j11, j12, tx, ty;
    persp_jacobian(cam, x, y, z, j00, j02, j11, j12, tx, ty);
    // cov2d = J C J^T with J = [[j00,0,j02],[0,j11,j12]]
    float a0 = j00 * C.xx + j02 * C.xz, a1 = j00 * C.xy + j02 * C.yz, a2 = j00 * C.xz + j02 * C.zz;  // row0 of J*C
    float b1 = j11 * C.yy + j12 * C.yz, b2 = j11 * C.yz + j12 * C.zz;                               // row1 (cols 1,2)
    float c00 = a0 * j00 + a2 * j02 + eps2d;
    float c01 = a1 * j11 + a2 * j12;
    float c11 = b1 * j11 + b2 * j12 + eps2d;
    float det = c00 * c11 - c01 * c01;
    if (det <= 0.f) return o;
    float inv_det = 1.f / det;
    float b = 0.5f * (c00 + c11);
    float v1 = b + sqrtf(fmaxf(0.01f, b * b - det));
    float radius = ceilf(3.f * sqrtf(v1));
    if (radius <= radius_clip) return o;
    float rz = 1.f / z;
    float mx = cam.fx * x * rz + cam.cx, my = cam.fy * y * rz + cam.cy;
    if (mx + radius <= 0.f || mx - radius >= (float)cam.W || my + radius <= 0.f || my - radius >= (float)cam.H) return o;
    o.radius = (int)radius; o.mx = mx; o.my = my; o.z = z;
    o.ca = c11 * inv_det; o.cb = -c01 * inv_det; o.cc = c00 * inv_det;
    return o;
}

// VJP of project_gaussian for a visible Gaussian.
//   in : v_m2[2], v_depth, v_conic[3], conic (saved forward output)
//   out: v_p[3], v_q[4], v_s[3]  (overwritten)
__device__ __forceinline__ void project_gaussian_vjp(const Cam& cam, const float p[3], const float q[4],
                                                     const float s[3], const float conic[3], const float v_m2[2],
                                                     float v_depth, const float v_conic[3], float v_p[3],
                                                     float v_q[4], float v_s[3]) {
    // d(cov2d) = -P * vP * P  (P = conic as symmetric 2x2; off-diagonal grads are halved)
    float Pa = conic[0], Pb = conic[1], Pc = conic[2];
    float Va = v_conic[0], Vb = 0.5f * v_conic[1], Vc = v_conic[2];
    float t00 = Pa * Va + Pb * Vb, t01 = Pa * Vb + Pb * Vc, t10 = Pb * Va + Pc * Vb, t11 = Pb * Vb + Pc * Vc;
    float g00 = -(t00 * Pa + t01 * Pb), g01 = -(t00 * Pb + t01 * Pc);
    float g10 = -(t10 * Pa + t11 * Pb), g11 = -(t10 * Pb + t11 * Pc);
    // recompute forward intermediates
    float x = cam.R[0] * p[0] + cam.R[1] * p[1] + cam.R[2] * p[2] + cam.t[0];
    float y = cam.R[3] * p[0] + cam.R[4] * p[1] + cam.R[5] * p[2] + cam.t[1];
    float z = cam.R[6] * p[0] + cam.R[7] * p[1] + cam.R[8] * p[2] + cam.t[2];
    float Rq[9], inv_norm;
    quat_to_R(q, Rq, inv_norm);
    float T[9];  // R_cam * Rq (unscaled)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            T[3 * r + c] = cam.R[3 * r] * Rq[c] + cam.R[3 * r + 1] * Rq[3 + c] + cam.R[3 * r + 2] * Rq[6 + c];
    float Ms[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Ms[3 * r + c] = T[3 * r + c] * s[c];
    Sym3 C = outer_MMt(Ms);
    float j00, j02, j11, j12, tx, ty;
    persp_jacobian(cam, x, y, z, j00, j02, j11, j12, tx, ty);
    float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
    // v_C (3x3, generally non symmetric) = J^T g J
    float J[6] = {j00, 0.f, j02, 0.f, j11, j12};
    float G[4] = {g00, g01, g10, g11};
    float JtG[6];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        JtG[2 * r] = J[r] * G[0] + J[3 + r] * G[2];
        JtG[2 * r + 1] = J[r] * G[1] + J[3 + r] * G[3];
    }
    float vC[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) vC[3 * r + c] = JtG[2 * r] * J[c] + JtG[2 * r + 1] * J[3 + c];
    // v_J = g J C^T + g^T J C, C symmetric -> (g + g^T) J C
    float S00 = 2.f * g00, S01 = g01 + g10, S11 = 2.f * g11;
    float Cm[9] = {C.xx, C.xy, C.xz, C.xy, C.yy, C.yz, C.xz, C.yz, C.zz};
    float JC[6];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        JC[c] = j00 * Cm[c] + j02 * Cm[6 + c];
        JC[3 + c] = j11 * Cm[3 + c] + j12 * Cm[6 + c];
    }
    float vJ00 = S00 * JC[0] + S01 * JC[3];
    float vJ02 = S00 * JC[2] + S01 * JC[5];
    float vJ11 = S01 * JC[1] + S11 * JC[4];
    float vJ12 = S01 * JC[2] + S11 * JC[5];
    float vx = cam.fx * rz * v_m2[0];
    float vy = cam.fy * rz * v_m2[1];
    float vz = -(cam.fx * x * v_m2[0] + cam.fy * y * v_m2[1]) * rz2;
    float xr = x * rz, yr = y * rz;
    if (xr <= cam.lim_x_pos && xr >= -cam.lim_x_neg) vx += -cam.fx * rz2 * vJ02;
    else vz += -cam.fx * rz3 * vJ02 * tx;
    if (yr <= cam.lim_y_pos && yr >= -cam.lim_y_neg) vy += -cam.fy * rz2 * vJ12;
    else vz += -cam.fy * rz3 * vJ12 * ty;
    vz += -cam.fx * rz2 * vJ00 - cam.fy * rz2 * vJ11 + 2.f * cam.fx * tx * rz3 * vJ02 + 2.f * cam.fy * ty * rz3 * vJ12;
    vz += v_depth;
    // back to world: v_p = R_cam^T v
    v_p[0] = cam.R[0] * vx + cam.R[3] * vy + cam.R[6] * vz;
    v_p[1] = cam.R[1] * vx + cam.R[4] * vy + cam.R[7] * vz;
    v_p[2] = cam.R[2] * vx + cam.R[5] * vy + cam.R[8] * vz;
    // cov_c = Ms Ms^T  ->  v_Ms = (vC + vC^T) Ms ; Ms = T diag(s), T = R_cam Rq
    float vMs[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            vMs[3 * r + c] = (vC[3 * r] + vC[r]) * Ms[c] + (vC[3 * r + 1] + vC[3 + r]) * Ms[3 + c] +
                             (vC[3 * r + 2] + vC[6 + r]) * Ms[6 + c];
#pragma unroll
    for (int c = 0; c < 3; c++) v_s[c] = T[c] * vMs[c] + T[3 + c] * vMs[3 + c] + T[6 + c] * vMs[6 + c];
    // v_T = vMs diag(s);  v_Rq = R_cam^T v_T
    float vRq[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            vRq[3 * r + c] = (cam.R[r] * vMs[c] + cam.R[3 + r] * vMs[3 + c] + cam.R[6 + r] * vMs[6 + c]) * s[c];
    // rotation-matrix -> quaternion VJP, then projection onto the tangent space of |q|=1
    float w = q[0] * inv_norm, qx = q[1] * inv_norm, qy = q[2] * inv_norm, qz = q[3] * inv_norm;
#define VR(r, c) vRq[3 * (r) + (c)]
    float n0 = 2.f * (qx * (VR(2, 1) - VR(1, 2)) + qy * (VR(0, 2) - VR(2, 0)) + qz * (VR(1, 0) - VR(0, 1)));
    float n1 = 2.f * (-2.f * qx * (VR(1, 1) + VR(2, 2)) + qy * (VR(1, 0) + VR(0, 1)) + qz * (VR(2, 0) + VR(0, 2)) +
                      w * (VR(2, 1) - VR(1, 2)));
    float n2 = 2.f * (qx * (VR(1, 0) + VR(0, 1)) - 2.f * qy * (VR(0, 0) + VR(2, 2)) + qz * (VR(2, 1) + VR(1, 2)) +
                      w * (VR(0, 2) - VR(2, 0)));
    float n3 = 2.f * (qx * (VR(2, 0) + VR(0, 2)) + qy * (VR(2, 1) + VR(1, 2)) - 2.f * qz * (VR(0, 0) + VR(1, 1)) +
                      w * (VR(1, 0) - VR(0, 1)));
#undef VR
    float d = n0 * w + n1 * qx + n2 * qy + n3 * qz;
    v_q[0] = (n0 - d * w) * inv_norm;
    v_q[1] = (n1 - d * qx) * inv_norm;
    v_q[2] = (n2 - d * qy) * inv_norm;
    v_q[3] = (n3 - d * qz) * inv_norm;
}

// Packed per-Gaussian record the record-streaming rasterizer reads with scalar loads (3 x float4 = 48 B).
// The last two words hold conservative integer pixel bounds of the region where opac * exp(-sigma) >= 1/255:
// sigma <= tau = ln(255 * opac) is an ellipse with half-extents sqrt(2 tau cc / det), sqrt(2 tau ca / det)
// (det = ca*cc - cb^2); they are inflated by 1 % + 0.01 px so that rounding of exp can never exclude a contributing
// pixel, and an empty box (hi < lo) is stored when opac < 1/255 or the Gaussian is culled.
__device__ __forceinline__ void pack_record(const Proj& o, float r, float g, float b, float opac, float4* rec) {
#pragma clang fp contract(off)   // (inlined into several kernels: unfused mul / add give every copy the same bits, see project_gaussian)
    float ex = -1.f, ey = -1.f;
    if (o.radius > 0) {
        const float tau = logf(255.f * opac);
        const float det = o.ca * o.cc - o.cb * o.cb;
        if (tau > 0.f && det > 0.f) {
            ex = sqrtf(2.f * tau * o.cc / det) * 1.01f + 0.01f;
            ey = sqrtf(2.f * tau * o.ca / det) * 1.01f + 0.01f;
        }
    }
    int x_lo = 1, x_hi = 0, y_lo = 1, y_hi = 0;  // empty
    if (ex >= 0.f) {
        // pixel j has centre j + 0.5: j_lo = floor(mx - ex - 0.5), j_hi = ceil(mx + ex - 0.5), clamped to int16
        x_lo = (int)fmaxf(fminf(floorf(o.mx - ex - 0.5f), 32767.f), -32768.f);
        x_hi = (int)fmaxf(fminf(ceilf(o.mx + ex - 0.5f), 32767.f), -32768.f);
        y_lo = (int)fmaxf(fminf(floorf(o.my - ey - 0.5f), 32767.f), -32768.f);
        y_hi = (int)fmaxf(fminf(ceilf(o.my + ey - 0.5f), 32767.f), -32768.f);
    }
    const int xb = (x_lo & 0xffff) | (x_hi << 16), yb = (y_lo & 0xffff) | (y_hi << 16);
    rec[0] = make_float4(o.mx, o.my, o.ca, o.cb);
    rec[1] = make_float4(o.cc, opac, o.z, r);
    rec[2] = make_float4(g, b, __int_as_float(xb), __int_as_float(yb));
}

// ---------------- spherical harmonics ----------------
__host__ __device__ inline int sh_num_bases(int degree) { return (degree + 1) * (degree + 1); }

// Basis values Y[0..nb) at the normalised direction (x,y,z); Sloan's recurrences.
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* Y) {
    Y[0] = 0.2820947917738781f;
    if (DEG < 1) return;
    Y[1] = -0.48860251190292f * y; Y[2] = 0.48860251190292f * z; Y[3] = -0.48860251190292f * x;
    if (DEG < 2) return;
    float z2 = z * z;
    float t0b = -1.092548430592079f * z;
    float c1 = x * x - y * y, s1 = 2.f * x * y;
    Y[4] = 0.5462742152960395f * s1; Y[5] = t0b * y;
    Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[7] = t0b * x; Y[8] = 0.5462742152960395f * c1;
    if (DEG < 3) return;
    float t0c = -2.285228997322329f * z2 + 0.4570457994644658f;
    float t1b = 1.445305721320277f * z;
    float c2 = x * c1 - y * s1, s2 = x * s1 + y * c1;
    Y[9] = -0.5900435899266435f * s2; Y[10] = t1b * s1; Y[11] = t0c * y;
    Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[13] = t0c * x; Y[14] = t1b * c1; Y[15] = -0.5900435899266435f * c2;
    if (DEG < 4) return;
    float t0d = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float t1c = 3.31161143515146f * z2 - 0.47308734787878f;
    float t2b = -1.770130769779931f * z;
    float c3 = x * c2 - y * s2, s3 = x * s2 + y * c2;
    Y[16] = 0.6258357354491763f * s3; Y[17] = t2b * s2; Y[18] = t1c * s1; Y[19] = t0d * y;
    Y[20] = 1.984313483298443f * z * Y[12] - 1.006230589874905f * Y[6];
    Y[21] = t0d * x; Y[22] = t1c * c1; Y[23] = t2b * c2; Y[24] = 0.6258357354491763f * c3;
}

// Basis gradients wrt the normalised (x,y,z).
template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float* dX, float* dY, float* dZ) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
#pragma unroll
    for (int k = 0; k < NB; k++) { dX[k] = 0.f; dY[k] = 0.f; dZ[k] = 0.f; }
    if (DEG < 1) return;
    dY[1] = -0.48860251190292f; dZ[2] = 0.48860251190292f; dX[3] = -0.48860251190292f;
    if (DEG < 2) return;
    float z2 = z * z;
    float c1 = x * x - y * y, s1 = 2.f * x * y;
    float c1x = 2.f * x, c1y = -2.f * y, s1x = 2.f * y, s1y = 2.f * x;
    float t0b = -1.092548430592079f * z;
    float p6z = 2.f * 0.9461746957575601f * z;
    dX[4] = 0.5462742152960395f * s1x; dY[4] = 0.5462742152960395f * s1y;
    dY[5] = t0b; dZ[5] = -1.092548430592079f * y;
    dZ[6] = p6z;
    dX[7] = t0b; dZ[7] = -1.092548430592079f * x;
    dX[8] = 0.5462742152960395f * c1x; dY[8] = 0.5462742152960395f * c1y;
    if (DEG < 3) return;
    float t0c = -2.285228997322329f * z2 + 0.4570457994644658f, t0cz = -2.285228997322329f * 2.f * z;
    float t1b = 1.445305721320277f * z;
    float c2 = x * c1 - y * s1, s2 = x * s1 + y * c1;
    float c2x = c1 + x * c1x - y * s1x, c2y = x * c1y - s1 - y * s1y;
    float s2x = s1 + x * s1x + y * c1x, s2y = x * s1y + c1 + y * c1y;
    float p12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    float p12z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    dX[9] = -0.5900435899266435f * s2x; dY[9] = -0.5900435899266435f * s2y;
    dX[10] = t1b * s1x; dY[10] = t1b * s1y; dZ[10] = 1.445305721320277f * s1;
    dY[11] = t0c; dZ[11] = t0cz * y;
    dZ[12] = p12z;
    dX[13] = t0c; dZ[13] = t0cz * x;
    dX[14] = t1b * c1x; dY[14] = t1b * c1y; dZ[14] = 1.445305721320277f * c1;
    dX[15] = -0.5900435899266435f * c2x; dY[15] = -0.5900435899266435f * c2y;
    if (DEG < 4) return;
    float t0d = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float t0dz = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
    float t1c = 3.31161143515146f * z2 - 0.47308734787878f, t1cz = 2.f * 3.31161143515146f * z;
    float t2b = -1.770130769779931f * z;
    float c3x = c2 + x * c2x - y * s2x, c3y = x * c2y - s2 - y * s2y;
    float s3x = s2 + y * c2x + x * s2x, s3y = x * s2y + c2 + y * c2y;
    dX[16] = 0.6258357354491763f * s3x; dY[16] = 0.6258357354491763f * s3y;
    dX[17] = t2b * s2x; dY[17] = t2b * s2y; dZ[17] = -1.770130769779931f * s2;
    dX[18] = t1c * s1x; dY[18] = t1c * s1y; dZ[18] = t1cz * s1;
    dY[19] = t0d; dZ[19] = t0dz * y;
    dZ[20] = 1.984313483298443f * (p12 + z * p12z) - 1.006230589874905f * p6z;
    dX[21] = t0d; dZ[21] = t0dz * x;
    dX[22] = t1c * c1x; dY[22] = t1c * c1y; dZ[22] = t1cz * c1;
    dX[23] = t2b * c2x; dY[23] = t2b * c2y; dZ[23] = -1.770130769779931f * c2;
    dX[24] = 0.6258357354491763f * c3x; dY[24] = 0.6258357354491763f * c3y;
}

}  // namespace gps
