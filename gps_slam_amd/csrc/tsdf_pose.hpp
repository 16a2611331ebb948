// Pose algebra of ORUtils::SE3Pose (ORUtils/SE3Pose.cpp, ORUtils/Matrix.h) as ITMBasicEngine / the tracker use it, in the
// reference's operation order: general 4x4 inverse by cofactors, exponential / logarithm maps with the small-angle branches,
// and "SetInvM(m); Coerce()".  Host AND device: the host uses it for given poses (gps_pose_from_c2w), the tracker's
// device-side Levenberg-Marquardt loop (tsdf_track.hip) for the pose update of every iteration.  Files including this header
// are compiled with -ffp-contract=off, so both sides run the same unfused operation sequence; sinf / cosf / asinf / acosf come
// from the respective math library (glibc / ocml) and may differ in the last bit.
#pragma once
#include <math.h>

#define GPS_HD __host__ __device__

namespace gpst {

// General 4x4 inverse by cofactors in the operation order of ORUtils/Matrix.h:177-238.
GPS_HD inline bool mat4_inverse(const float* a, float* out) {
    float s[16], t[12];
    for (int i = 0; i < 4; i++) { s[i] = a[i * 4]; s[i + 4] = a[i * 4 + 1]; s[i + 8] = a[i * 4 + 2]; s[i + 12] = a[i * 4 + 3]; }
    t[0] = s[10] * s[15]; t[1] = s[11] * s[14]; t[2] = s[9] * s[15]; t[3] = s[11] * s[13];
    t[4] = s[9] * s[14]; t[5] = s[10] * s[13]; t[6] = s[8] * s[15]; t[7] = s[11] * s[12];
    t[8] = s[8] * s[14]; t[9] = s[10] * s[12]; t[10] = s[8] * s[13]; t[11] = s[9] * s[12];
    out[0] = (t[0] * s[5] + t[3] * s[6] + t[4] * s[7]) - (t[1] * s[5] + t[2] * s[6] + t[5] * s[7]);
    out[1] = (t[1] * s[4] + t[6] * s[6] + t[9] * s[7]) - (t[0] * s[4] + t[7] * s[6] + t[8] * s[7]);
    out[2] = (t[2] * s[4] + t[7] * s[5] + t[10] * s[7]) - (t[3] * s[4] + t[6] * s[5] + t[11] * s[7]);
    out[3] = (t[5] * s[4] + t[8] * s[5] + t[11] * s[6]) - (t[4] * s[4] + t[9] * s[5] + t[10] * s[6]);
    const float det = s[0] * out[0] + s[1] * out[1] + s[2] * out[2] + s[3] * out[3];
    if (det == 0.0f) return false;
    out[4] = (t[1] * s[1] + t[2] * s[2] + t[5] * s[3]) - (t[0] * s[1] + t[3] * s[2] + t[4] * s[3]);
    out[5] = (t[0] * s[0] + t[7] * s[2] + t[8] * s[3]) - (t[1] * s[0] + t[6] * s[2] + t[9] * s[3]);
    out[6] = (t[3] * s[0] + t[6] * s[1] + t[11] * s[3]) - (t[2] * s[0] + t[7] * s[1] + t[10] * s[3]);
    out[7] = (t[4] * s[0] + t[9] * s[1] + t[10] * s[2]) - (t[5] * s[0] + t[8] * s[1] + t[11] * s[2]);
    t[0] = s[2] * s[7]; t[1] = s[3] * s[6]; t[2] = s[1] * s[7]; t[3] = s[3] * s[5];
    t[4] = s[1] * s[6]; t[5] = s[2] * s[5]; t[6] = s[0] * s[7]; t[7] = s[3] * s[4];
    t[8] = s[0] * s[6]; t[9] = s[2] * s[4]; t[10] = s[0] * s[5]; t[11] = s[1] * s[4];
    out[8] = (t[0] * s[13] + t[3] * s[14] + t[4] * s[15]) - (t[1] * s[13] + t[2] * s[14] + t[5] * s[15]);
    out[9] = (t[1] * s[12] + t[6] * s[14] + t[9] * s[15]) - (t[0] * s[12] + t[7] * s[14] + t[8] * s[15]);
    out[10] = (t[2] * s[12] + t[7] * s[13] + t[10] * s[15]) - (t[3] * s[12] + t[6] * s[13] + t[11] * s[15]);
    out[11] = (t[5] * s[12] + t[8] * s[13] + t[11] * s[14]) - (t[4] * s[12] + t[9] * s[13] + t[10] * s[14]);
    out[12] = (t[2] * s[10] + t[5] * s[11] + t[1] * s[9]) - (t[4] * s[11] + t[0] * s[9] + t[3] * s[10]);
    out[13] = (t[8] * s[11] + t[0] * s[8] + t[7] * s[10]) - (t[6] * s[10] + t[9] * s[11] + t[1] * s[8]);
    out[14] = (t[6] * s[9] + t[11] * s[11] + t[3] * s[8]) - (t[10] * s[11] + t[2] * s[8] + t[7] * s[9]);
    out[15] = (t[10] * s[10] + t[4] * s[8] + t[9] * s[9]) - (t[8] * s[9] + t[11] * s[10] + t[5] * s[8]);
    const float inv = 1 / det;
    for (int i = 0; i < 16; i++) out[i] = out[i] * inv;
    return true;
}

GPS_HD inline float dot3(const float* a, const float* b) { float r = 0; r += a[0] * b[0]; r += a[1] * b[1]; r += a[2] * b[2]; return r; }

// exponential map se(3) -> SE(3) with the reference's small-angle branches (SE3Pose.cpp:92-158)
GPS_HD inline void pose_exp(const float* prm, float* M) {
    const float one_6th = 1.0f / 6.0f, one_20th = 1.0f / 20.0f;
    const float t[3] = {prm[0], prm[1], prm[2]}, w[3] = {prm[3], prm[4], prm[5]};
    const float theta_sq = dot3(w, w);
    const float theta = sqrtf(theta_sq);
    const float cr[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
    float A, B, T[3];
    if (theta_sq < 1e-8f) {
        A = 1.0f - one_6th * theta_sq; B = 0.5f;
        for (int k = 0; k < 3; k++) T[k] = t[k] + 0.5f * cr[k];
    } else {
        float C;
        if (theta_sq < 1e-6f) {
            C = one_6th * (1.0f - one_20th * theta_sq);
            A = 1.0f - theta_sq * C;
            B = 0.5f - 0.25f * one_6th * theta_sq;
        } else {
            const float it = 1.0f / theta;
            A = sinf(theta) * it;
            B = (1.0f - cosf(theta)) * (it * it);
            C = (1.0f - A) * (it * it);
        }
        const float c2[3] = {w[1] * cr[2] - w[2] * cr[1], w[2] * cr[0] - w[0] * cr[2], w[0] * cr[1] - w[1] * cr[0]};
        for (int k = 0; k < 3; k++) T[k] = t[k] + B * cr[k] + C * c2[k];
    }
    const float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    M[0] = 1.0f - B * (wy2 + wz2); M[5] = 1.0f - B * (wx2 + wz2); M[10] = 1.0f - B * (wx2 + wy2);
    float a = A * w[2], b = B * (w[0] * w[1]);
    M[4] = b - a; M[1] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    M[8] = b + a; M[2] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    M[9] = b - a; M[6] = b + a;
    M[12] = T[0]; M[13] = T[1]; M[14] = T[2];
    M[3] = 0.0f; M[7] = 0.0f; M[11] = 0.0f; M[15] = 1.0f;
}

// logarithm SE(3) -> se(3) (SE3Pose.cpp:160-243)
GPS_HD inline void pose_log(const float* M, float* prm) {
    const float T[3] = {M[12], M[13], M[14]};
    float rot[3];
    const float cos_angle = (M[0] + M[5] + M[10] - 1.0f) * 0.5f;
    rot[0] = (M[6] - M[9]) * 0.5f; rot[1] = (M[8] - M[2]) * 0.5f; rot[2] = (M[1] - M[4]) * 0.5f;
    const float sin_abs = sqrtf(dot3(rot, rot));
    const double kSqrtHalf = 0.707106781186547524401;
    if ((double)cos_angle > kSqrtHalf) {
        if (sin_abs) { const float p = asinf(sin_abs) / sin_abs; rot[0] *= p; rot[1] *= p; rot[2] *= p; }
    } else if ((double)cos_angle > -kSqrtHalf) {
        const float p = acosf(cos_angle) / sin_abs;
        rot[0] *= p; rot[1] *= p; rot[2] *= p;
    } else {
        const float angle = (float)3.14159265358979323846 - asinf(sin_abs);
        const float d0 = M[0] - cos_angle, d1 = M[5] - cos_angle, d2 = M[10] - cos_angle;
        float r2[3];
        if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) {
            r2[0] = d0; r2[1] = (M[1] + M[4]) * 0.5f; r2[2] = (M[8] + M[2]) * 0.5f;
        } else if (fabsf(d1) > fabsf(d2)) {
            r2[0] = (M[1] + M[4]) * 0.5f; r2[1] = d1; r2[2] = (M[6] + M[9]) * 0.5f;
        } else {
            r2[0] = (M[8] + M[2]) * 0.5f; r2[1] = (M[6] + M[9]) * 0.5f; r2[2] = d2;
        }
        if (dot3(r2, rot) < 0.0f) { r2[0] *= -1.0f; r2[1] *= -1.0f; r2[2] *= -1.0f; }
        const float len = sqrtf(dot3(r2, r2));
        if (len == 0) { r2[0] = r2[1] = r2[2] = 0; } else { r2[0] /= len; r2[1] /= len; r2[2] /= len; }
        rot[0] = angle * r2[0]; rot[1] = angle * r2[1]; rot[2] = angle * r2[2];
    }
    float shtot = 0.5f;
    const float theta = sqrtf(dot3(rot, rot));
    if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
    const float half[6] = {0.0f, 0.0f, 0.0f, rot[0] * -0.5f, rot[1] * -0.5f, rot[2] * -0.5f};
    float HM[16];
    pose_exp(half, HM);
    float rt[3];
    for (int r = 0; r < 3; r++) rt[r] = HM[r] * T[0] + HM[r + 4] * T[1] + HM[r + 8] * T[2];
    float param;
    if (theta > 0.001f) param = dot3(T, rot) * (1 - 2 * shtot) / dot3(rot, rot);
    else param = dot3(T, rot) / 24;
    for (int k = 0; k < 3; k++) { rt[k] -= rot[k] * param; }
    for (int k = 0; k < 3; k++) rt[k] /= 2 * shtot;
    prm[0] = rt[0]; prm[1] = rt[1]; prm[2] = rt[2];
    prm[3] = rot[0]; prm[4] = rot[1]; prm[5] = rot[2];
}


// pose.SetInvM(c2w); pose.Coerce(): c2w in ORUtils layout m[col*4 + row] -> M = GetM(), invM = GetInvM()
// (SetInvM inverts and extracts the parameters, Coerce extracts them again and rebuilds M; GetInvM inverts M)
GPS_HD inline bool pose_set_invM_coerce(const float* c2w, float* M, float* invM) {
    float M0[16], prm[6];
    if (!mat4_inverse(c2w, M0)) return false;
    pose_log(M0, prm);
    pose_log(M0, prm);
    pose_exp(prm, M);
    return mat4_inverse(M, invM);
}

}  // namespace gpst
