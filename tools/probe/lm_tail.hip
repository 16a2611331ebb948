// What would the host half of one LM iteration cost ON THE DEVICE (round-5 review, item 5: the coarse tracker levels as a resident
// device loop)?  One lane of one wave runs the accepted-step tail of LmLoop::apply (csrc/tsdf_track.hip) for a rotation-only level
// -- normalise, damp, 3x3 Cholesky + back-substitution, Tinc, 4x4 product, SetInvM + Coerce (4x4 inverse, two SE3 logs, one exp,
// 4x4 inverse: csrc/tsdf_pose.hpp) -- and, for comparison, the 6x6 solve of the finer levels; wall_clock64 around REPS dependent
// iterations.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I gps_slam_amd/csrc -o /tmp/lm_tail tools/probe/lm_tail.hip && /tmp/lm_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#include "tsdf_pose.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace gpst;

template <int N>
__device__ void chol_solve(const float* mat, const float* v, float* result) {   // ORUtils::Cholesky as the host runs it
    float ch[N * N];
    for (int i = 0; i < N * N; i++) ch[i] = mat[i];
    for (int c = 0; c < N; c++) {
        float inv_diag = 1;
        for (int r = c; r < N; r++) {
            float val = ch[c + r * N];
            for (int c2 = 0; c2 < c; c2++) val -= ch[c + c2 * N] * ch[c2 + r * N];
            if (r == c) { ch[c + r * N] = val; inv_diag = 1.0f / val; }
            else { ch[r + c * N] = val; ch[c + r * N] = val * inv_diag; }
        }
    }
    float y[N];
    for (int i = 0; i < N; i++) { float val = v[i]; for (int j = 0; j < i; j++) val -= ch[j + i * N] * y[j]; y[i] = val; }
    for (int i = 0; i < N; i++) y[i] /= ch[i + i * N];
    for (int i = N - 1; i >= 0; i--) { float val = y[i]; for (int j = i + 1; j < N; j++) val -= ch[i + j * N] * result[j]; result[i] = val; }
}

template <int NP>
__global__ void lm_tail_kernel(const float* in, float* out, unsigned long long* ticks, int reps) {
    if (threadIdx.x != 0) return;
    float invM[16], M[16], hess[36], nabla[6];
    for (int i = 0; i < 16; i++) invM[i] = in[i];
    for (int i = 0; i < 36; i++) hess[i] = in[16 + i];
    for (int i = 0; i < 6; i++) nabla[i] = in[52 + i];
    float lambda = 1.0f, nvalid = 4000.0f;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < reps; it++) {
        float A[36], nb[6];
        for (int i = 0; i < 36; i++) A[i] = hess[i] / nvalid;
        for (int i = 0; i < 6; i++) nb[i] = nabla[i] / nvalid * (1.0f + 1e-3f * (float)(it & 7));   // (a fresh right-hand side per iteration)
        for (int i = 0; i < 6; i++) A[i + i * 6] *= 1.0f + lambda;
        float step[6] = {0, 0, 0, 0, 0, 0};
        if (NP == 3) {
            float small[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) small[r + c * 3] = A[r + c * 6];
            chol_solve<3>(small, nb, step);
        } else {
            chol_solve<6>(A, nb, step);
        }
        float s6[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < NP; i++) s6[i] = step[i];
        float T[16], P[16];
        T[0] = 1.0f; T[4] = s6[2]; T[8] = -s6[1]; T[12] = s6[3];
        T[1] = -s6[2]; T[5] = 1.0f; T[9] = s6[0]; T[13] = s6[4];
        T[2] = s6[1]; T[6] = -s6[0]; T[10] = 1.0f; T[14] = s6[5];
        T[3] = 0.0f; T[7] = 0.0f; T[11] = 0.0f; T[15] = 1.0f;
        for (int col = 0; col < 4; col++)
            for (int row = 0; row < 4; row++) {
                float acc = 0;
                for (int k = 0; k < 4; k++) acc += T[k * 4 + row] * invM[col * 4 + k];
                P[col * 4 + row] = acc;
            }
        pose_set_invM_coerce(P, M, invM);   // (ORUtils layout m[col * 4 + row], as tsdf_track.hip's set_invM_coerce hands it over)
        lambda = lambda > 1e-3f ? lambda / 10.0f : 1.0f;
    }
    const unsigned long long t1 = wall_clock64();
    ticks[0] = t1 - t0;
    for (int i = 0; i < 16; i++) out[i] = invM[i];
}

int main() {
    float h_in[58] = {0};
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.1f, -0.2f, 0.3f, 1};
    memcpy(h_in, I, sizeof(I));
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) h_in[16 + r + c * 6] = (r == c ? 4000.0f * (2.0f + r) : 4000.0f * 0.1f / (1 + r + c));
    for (int i = 0; i < 6; i++) h_in[52 + i] = 4000.0f * 1e-3f * (i + 1);
    float *d_in, *d_out; unsigned long long* d_t;
    CK(hipMalloc(&d_in, sizeof(h_in))); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_t, 8));
    CK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
    const int reps = 2000;
    for (int np : {3, 6}) {
        for (int warm = 0; warm < 2; warm++) {
            if (np == 3) lm_tail_kernel<3><<<1, 64>>>(d_in, d_out, d_t, reps); else lm_tail_kernel<6><<<1, 64>>>(d_in, d_out, d_t, reps);
            CK(hipDeviceSynchronize());
        }
        unsigned long long t; float o[16];
        CK(hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(o, d_out, 64, hipMemcpyDeviceToHost));
        printf("%d parameters: %.2f us per LM tail on one lane (%d dependent iterations; pose[12..14] = %.4f %.4f %.4f)\n", np, (double)t / 100.0 / reps, reps, o[12], o[13], o[14]);
    }
    return 0;
}
