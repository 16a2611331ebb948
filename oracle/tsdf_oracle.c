/*
 * tsdf_oracle.c -- CPU restatement of the GPS-SLAM TSDF path (InfiniTAM ITMLib, voxel-block
 * hashing, ITMVoxel_s_rgb) as driven by ITMBasicEngine::ProcessFrame / runRaycast.
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg).
 * The product (gps_slam_amd/) never includes, links or calls this file.
 *
 * PARITY PINNED: tests/test_oracle_tsdf.py checks this restatement bit-for-bit against
 * (i) golden dumps under tests/golden/ produced by the reference's OWN CPU engine
 * (oracle/_ref/itm_ref, built from /root/reference by oracle/ref_build.sh) and
 * (ii) live runs of that binary when it is present.
 *
 * Plain scalar fp32 C, -ffp-contract=off.  Every function cites the reference file:line
 * (relative to /root/reference/InfiniTAM/ITMLib unless stated) it follows, and keeps the
 * reference's operation order so that float results are bit-identical.
 * Matrices use the ORUtils layout: m[col*4 + row] (ORUtils/Matrix.h:26-35).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#define BLK 8
#define BLK3 512
#define FAR_AWAY 999999.9f
#define VERY_CLOSE 0.05f
#define MINMAX_SUB 8
#define MAX_RENDERING_BLOCKS (65536 * 4)

typedef struct { int16_t sdf; uint8_t w_depth; uint8_t clr[3]; uint8_t w_color; uint8_t pad; } Voxel; /* Objects/Scene/ITMVoxelTypes.h:41-69 */
typedef struct { int16_t pos[3]; int16_t pad; int32_t offset; int32_t ptr; } HashEntry;                /* Objects/Scene/ITMVoxelBlockHash.h:36-48 */
typedef struct { float x, y, z, w; } V4;
typedef struct { float x, y; } V2;

typedef struct {
    int W, H;
    float fx, fy, cx, cy;
    float voxel, mu, vf_min, vf_max;
    int maxW;
    int n_blocks, n_buckets, n_excess, n_total;
    uint32_t hash_mask;
    Voxel *vba;
    int32_t *vba_alloc_list;
    int32_t last_free_block;
    HashEntry *hash;
    int32_t *excess_list;
    int32_t last_free_excess;
    uint8_t *alloc_type;
    int16_t *block_coords; /* [n_total][4] */
    /* live render state */
    int32_t *visible_ids;
    int32_t n_visible;
    uint8_t *visible_type;
    V2 *minmax;
    V4 *raycast;
    V4 *icp_points, *icp_normals;
    float *depth;
    uint8_t *rgb; /* uchar4 */
    /* free-view render state */
    int32_t *fv_visible_ids;
    int32_t fv_n_visible;
    V2 *fv_minmax;
    V4 *fv_raycast;
    uint8_t *fv_colour; /* uchar4 */
    /* tracking state (ITMTrackingState): pose_d, pose_pointCloud, age_pointCloud, framesProcessed */
    float pose_M[16], pose_invM[16], pose_pc_M[16];
    int age_point_cloud, trk_frames;
    float trk_diag[16];
    float *trk_scratch;
    /* ray statistics (SURVEY 8(d) S-bar): trips of castRay's loop and rays cast since creation, live + free views */
    int64_t ray_steps, rays;
} Tsdf;

/* ---------------- construction / reset: Engines/Reconstruction/CPU/ITMSceneReconstructionEngine_CPU.tpp:26-50 ------------- */
static Voxel empty_voxel(void) { Voxel v; memset(&v, 0, sizeof(v)); v.sdf = 32767; return v; }

static void tracking_reset(Tsdf *t) { /* ITMTrackingState::Reset (Objects/Tracking/ITMTrackingState.h:92-99) */
    for (int i = 0; i < 16; i++) t->pose_M[i] = t->pose_invM[i] = t->pose_pc_M[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    t->age_point_cloud = -1;
    t->trk_frames = 0;
    memset(t->trk_diag, 0, sizeof(t->trk_diag));
}

ORC_API void orc_tsdf_reset(Tsdf *t) {
    tracking_reset(t);
    Voxel e = empty_voxel();
    for (size_t i = 0; i < (size_t)t->n_blocks * BLK3; i++) t->vba[i] = e;
    for (int i = 0; i < t->n_blocks; i++) t->vba_alloc_list[i] = i;
    t->last_free_block = t->n_blocks - 1;
    memset(t->hash, 0, sizeof(HashEntry) * (size_t)t->n_total);
    for (int i = 0; i < t->n_total; i++) t->hash[i].ptr = -2;
    for (int i = 0; i < t->n_excess; i++) t->excess_list[i] = i;
    t->last_free_excess = t->n_excess - 1;
    memset(t->visible_type, 0, (size_t)t->n_total);
    t->n_visible = 0;
    t->fv_n_visible = 0;
}

ORC_API Tsdf *orc_tsdf_create(int W, int H, float fx, float fy, float cx, float cy, float voxel, float mu, float vf_min,
                              float vf_max, int n_blocks, int n_buckets, int n_excess) {
    Tsdf *t = (Tsdf *)calloc(1, sizeof(Tsdf));
    t->W = W; t->H = H; t->fx = fx; t->fy = fy; t->cx = cx; t->cy = cy;
    t->voxel = voxel; t->mu = mu; t->vf_min = vf_min; t->vf_max = vf_max; t->maxW = 100; /* Utils/ITMLibSettings.cpp:10 */
    t->n_blocks = n_blocks; t->n_buckets = n_buckets; t->n_excess = n_excess; t->n_total = n_buckets + n_excess;
    t->hash_mask = (uint32_t)n_buckets - 1u;
    size_t P = (size_t)W * H;
    t->vba = (Voxel *)malloc(sizeof(Voxel) * (size_t)n_blocks * BLK3);
    t->vba_alloc_list = (int32_t *)malloc(4 * (size_t)n_blocks);
    t->hash = (HashEntry *)malloc(sizeof(HashEntry) * (size_t)t->n_total);
    t->excess_list = (int32_t *)malloc(4 * (size_t)n_excess);
    t->alloc_type = (uint8_t *)calloc((size_t)t->n_total, 1);
    t->block_coords = (int16_t *)calloc((size_t)t->n_total * 4, 2);
    t->visible_ids = (int32_t *)malloc(4 * (size_t)n_blocks);
    t->visible_type = (uint8_t *)calloc((size_t)t->n_total, 1);
    t->fv_visible_ids = (int32_t *)malloc(4 * (size_t)n_blocks);
    t->minmax = (V2 *)calloc(P, sizeof(V2)); t->fv_minmax = (V2 *)calloc(P, sizeof(V2));
    t->raycast = (V4 *)calloc(P, sizeof(V4)); t->fv_raycast = (V4 *)calloc(P, sizeof(V4));
    t->icp_points = (V4 *)calloc(P, sizeof(V4)); t->icp_normals = (V4 *)calloc(P, sizeof(V4));
    t->depth = (float *)calloc(P, 4); t->rgb = (uint8_t *)calloc(P, 4); t->fv_colour = (uint8_t *)calloc(P, 4);
    orc_tsdf_reset(t);
    return t;
}

ORC_API void orc_tsdf_destroy(Tsdf *t) {
    free(t->vba); free(t->vba_alloc_list); free(t->hash); free(t->excess_list); free(t->alloc_type);
    free(t->block_coords); free(t->visible_ids); free(t->visible_type); free(t->fv_visible_ids); free(t->minmax);
    free(t->fv_minmax); free(t->raycast); free(t->fv_raycast); free(t->icp_points); free(t->icp_normals);
    free(t->depth); free(t->rgb); free(t->fv_colour); free(t);
}

/* ---------------- small math in the reference's operation order ---------------- */
static V4 m4_mul_v4(const float *m, V4 v) { /* ORUtils/Matrix.h:130-137 */
    V4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}
#define ROUNDF(x) (((x) < 0) ? ((x) - 0.5f) : ((x) + 0.5f)) /* ORUtils/MathUtils.h:21-22 */

static int hash_index(int x, int y, int z, uint32_t mask) { /* Objects/Scene/ITMRepresentationAccess.h:8-11 */
    return (int)((((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349669u) ^ ((uint32_t)z * 83492791u)) & mask);
}

/* ORUtils/Matrix.h:177-238 (cofactor inverse).  out = m^-1 */
ORC_API int orc_mat4_inv(const float *m, float *dst) {
    float tmp[12], src[16], det;
    for (int i = 0; i < 4; i++) { src[i] = m[i * 4]; src[i + 4] = m[i * 4 + 1]; src[i + 8] = m[i * 4 + 2]; src[i + 12] = m[i * 4 + 3]; }
    tmp[0] = src[10] * src[15]; tmp[1] = src[11] * src[14]; tmp[2] = src[9] * src[15]; tmp[3] = src[11] * src[13];
    tmp[4] = src[9] * src[14]; tmp[5] = src[10] * src[13]; tmp[6] = src[8] * src[15]; tmp[7] = src[11] * src[12];
    tmp[8] = src[8] * src[14]; tmp[9] = src[10] * src[12]; tmp[10] = src[8] * src[13]; tmp[11] = src[9] * src[12];
    dst[0] = (tmp[0] * src[5] + tmp[3] * src[6] + tmp[4] * src[7]) - (tmp[1] * src[5] + tmp[2] * src[6] + tmp[5] * src[7]);
    dst[1] = (tmp[1] * src[4] + tmp[6] * src[6] + tmp[9] * src[7]) - (tmp[0] * src[4] + tmp[7] * src[6] + tmp[8] * src[7]);
    dst[2] = (tmp[2] * src[4] + tmp[7] * src[5] + tmp[10] * src[7]) - (tmp[3] * src[4] + tmp[6] * src[5] + tmp[11] * src[7]);
    dst[3] = (tmp[5] * src[4] + tmp[8] * src[5] + tmp[11] * src[6]) - (tmp[4] * src[4] + tmp[9] * src[5] + tmp[10] * src[6]);
    det = src[0] * dst[0] + src[1] * dst[1] + src[2] * dst[2] + src[3] * dst[3];
    if (det == 0.0f) return 0;
    dst[4] = (tmp[1] * src[1] + tmp[2] * src[2] + tmp[5] * src[3]) - (tmp[0] * src[1] + tmp[3] * src[2] + tmp[4] * src[3]);
    dst[5] = (tmp[0] * src[0] + tmp[7] * src[2] + tmp[8] * src[3]) - (tmp[1] * src[0] + tmp[6] * src[2] + tmp[9] * src[3]);
    dst[6] = (tmp[3] * src[0] + tmp[6] * src[1] + tmp[11] * src[3]) - (tmp[2] * src[0] + tmp[7] * src[1] + tmp[10] * src[3]);
    dst[7] = (tmp[4] * src[0] + tmp[9] * src[1] + tmp[10] * src[2]) - (tmp[5] * src[0] + tmp[8] * src[1] + tmp[11] * src[2]);
    tmp[0] = src[2] * src[7]; tmp[1] = src[3] * src[6]; tmp[2] = src[1] * src[7]; tmp[3] = src[3] * src[5];
    tmp[4] = src[1] * src[6]; tmp[5] = src[2] * src[5]; tmp[6] = src[0] * src[7]; tmp[7] = src[3] * src[4];
    tmp[8] = src[0] * src[6]; tmp[9] = src[2] * src[4]; tmp[10] = src[0] * src[5]; tmp[11] = src[1] * src[4];
    dst[8] = (tmp[0] * src[13] + tmp[3] * src[14] + tmp[4] * src[15]) - (tmp[1] * src[13] + tmp[2] * src[14] + tmp[5] * src[15]);
    dst[9] = (tmp[1] * src[12] + tmp[6] * src[14] + tmp[9] * src[15]) - (tmp[0] * src[12] + tmp[7] * src[14] + tmp[8] * src[15]);
    dst[10] = (tmp[2] * src[12] + tmp[7] * src[13] + tmp[10] * src[15]) - (tmp[3] * src[12] + tmp[6] * src[13] + tmp[11] * src[15]);
    dst[11] = (tmp[5] * src[12] + tmp[8] * src[13] + tmp[11] * src[14]) - (tmp[4] * src[12] + tmp[9] * src[13] + tmp[10] * src[14]);
    dst[12] = (tmp[2] * src[10] + tmp[5] * src[11] + tmp[1] * src[9]) - (tmp[4] * src[11] + tmp[0] * src[9] + tmp[3] * src[10]);
    dst[13] = (tmp[8] * src[11] + tmp[0] * src[8] + tmp[7] * src[10]) - (tmp[6] * src[10] + tmp[9] * src[11] + tmp[1] * src[8]);
    dst[14] = (tmp[6] * src[9] + tmp[11] * src[11] + tmp[3] * src[8]) - (tmp[10] * src[11] + tmp[2] * src[8] + tmp[7] * src[9]);
    dst[15] = (tmp[10] * src[10] + tmp[4] * src[8] + tmp[9] * src[9]) - (tmp[8] * src[9] + tmp[11] * src[10] + tmp[5] * src[8]);
    float s = 1 / det;
    for (int i = 0; i < 16; i++) dst[i] = dst[i] * s;
    return 1;
}

/* ORUtils/SE3Pose.cpp:92-158 SetModelViewFromParams: params (t, w) -> M */
static void pose_M_from_params(const float *p, float *M) {
    float one_6th = 1.0f / 6.0f, one_20th = 1.0f / 20.0f;
    float tx = p[0], ty = p[1], tz = p[2], wx = p[3], wy = p[4], wz = p[5];
    float theta_sq = 0; theta_sq += wx * wx; theta_sq += wy * wy; theta_sq += wz * wz;
    float theta = sqrtf(theta_sq);
    float A, B;
    float T[3];
    float cx = wy * tz - wz * ty, cy = wz * tx - wx * tz, cz = wx * ty - wy * tx; /* cross(w,t) */
    if (theta_sq < 1e-8f) {
        A = 1.0f - one_6th * theta_sq; B = 0.5f;
        T[0] = tx + 0.5f * cx; T[1] = ty + 0.5f * cy; T[2] = tz + 0.5f * cz;
    } else {
        float C;
        if (theta_sq < 1e-6f) {
            C = one_6th * (1.0f - one_20th * theta_sq);
            A = 1.0f - theta_sq * C;
            B = 0.5f - 0.25f * one_6th * theta_sq;
        } else {
            float inv_theta = 1.0f / theta;
            A = sinf(theta) * inv_theta;
            B = (1.0f - cosf(theta)) * (inv_theta * inv_theta);
            C = (1.0f - A) * (inv_theta * inv_theta);
        }
        float c2x = wy * cz - wz * cy, c2y = wz * cx - wx * cz, c2z = wx * cy - wy * cx; /* cross(w, cross(w,t)) */
        T[0] = tx + B * cx + C * c2x; T[1] = ty + B * cy + C * c2y; T[2] = tz + B * cz + C * c2z;
    }
    float wx2 = wx * wx, wy2 = wy * wy, wz2 = wz * wz;
    float R[9]; /* R.m[c*3 + r]?  SE3Pose uses R.m[row + 3*col] */
    R[0 + 3 * 0] = 1.0f - B * (wy2 + wz2);
    R[1 + 3 * 1] = 1.0f - B * (wx2 + wz2);
    R[2 + 3 * 2] = 1.0f - B * (wx2 + wy2);
    float a, b;
    a = A * wz; b = B * (wx * wy); R[0 + 3 * 1] = b - a; R[1 + 3 * 0] = b + a;
    a = A * wy; b = B * (wx * wz); R[0 + 3 * 2] = b + a; R[2 + 3 * 0] = b - a;
    a = A * wx; b = B * (wy * wz); R[1 + 3 * 2] = b - a; R[2 + 3 * 1] = b + a;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) M[r + 4 * c] = R[r + 3 * c];
    M[0 + 4 * 3] = T[0]; M[1 + 4 * 3] = T[1]; M[2 + 4 * 3] = T[2];
    M[3 + 4 * 0] = 0.0f; M[3 + 4 * 1] = 0.0f; M[3 + 4 * 2] = 0.0f; M[3 + 4 * 3] = 1.0f;
}

static float dot3(const float *a, const float *b) { float r = 0; r += a[0] * b[0]; r += a[1] * b[1]; r += a[2] * b[2]; return r; }

/* ORUtils/SE3Pose.cpp:160-243 SetParamsFromModelView: M -> params */
static void pose_params_from_M(const float *M, float *params) {
    float R[9];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R[r + 3 * c] = M[r + 4 * c];
    float T[3] = {M[0 + 4 * 3], M[1 + 4 * 3], M[2 + 4 * 3]};
    float rot[3];
    /* Matrix3 m00..: ORUtils/Matrix.h names mXY = column X row Y -> m00=R[0], m11=R[4], m22=R[8] */
    float cos_angle = (R[0] + R[4] + R[8] - 1.0f) * 0.5f;
    rot[0] = (R[2 + 3 * 1] - R[1 + 3 * 2]) * 0.5f;
    rot[1] = (R[0 + 3 * 2] - R[2 + 3 * 0]) * 0.5f;
    rot[2] = (R[1 + 3 * 0] - R[0 + 3 * 1]) * 0.5f;
    float sin_angle_abs = sqrtf(dot3(rot, rot));
    const double SQRT1_2 = 0.707106781186547524401; /* comparisons promote the float to double (M_SQRT1_2) */
    if ((double)cos_angle > SQRT1_2) {
        if (sin_angle_abs) {
            float p = asinf(sin_angle_abs) / sin_angle_abs;
            rot[0] *= p; rot[1] *= p; rot[2] *= p;
        }
    } else {
        if ((double)cos_angle > -SQRT1_2) {
            float p = acosf(cos_angle) / sin_angle_abs;
            rot[0] *= p; rot[1] *= p; rot[2] *= p;
        } else {
            float angle = (float)3.14159265358979323846 - asinf(sin_angle_abs);
            float d0 = R[0 + 3 * 0] - cos_angle, d1 = R[1 + 3 * 1] - cos_angle, d2 = R[2 + 3 * 2] - cos_angle;
            float r2[3];
            if (fabsf(d0) > fabsf(d1) && fabsf(d0) > fabsf(d2)) {
                r2[0] = d0; r2[1] = (R[1 + 3 * 0] + R[0 + 3 * 1]) * 0.5f; r2[2] = (R[0 + 3 * 2] + R[2 + 3 * 0]) * 0.5f;
            } else if (fabsf(d1) > fabsf(d2)) {
                r2[0] = (R[1 + 3 * 0] + R[0 + 3 * 1]) * 0.5f; r2[1] = d1; r2[2] = (R[2 + 3 * 1] + R[1 + 3 * 2]) * 0.5f;
            } else {
                r2[0] = (R[0 + 3 * 2] + R[2 + 3 * 0]) * 0.5f; r2[1] = (R[2 + 3 * 1] + R[1 + 3 * 2]) * 0.5f; r2[2] = d2;
            }
            if (dot3(r2, rot) < 0.0f) { r2[0] *= -1.0f; r2[1] *= -1.0f; r2[2] *= -1.0f; }
            float len = sqrtf(dot3(r2, r2)); /* normalize(): vec / length (ORUtils/Vector.h:1331-1336) */
            if (len == 0) { r2[0] = r2[1] = r2[2] = 0; } else { r2[0] /= len; r2[1] /= len; r2[2] /= len; }
            rot[0] = angle * r2[0]; rot[1] = angle * r2[1]; rot[2] = angle * r2[2];
        }
    }
    float shtot = 0.5f;
    float theta = sqrtf(dot3(rot, rot));
    if (theta > 0.00001f) shtot = sinf(theta * 0.5f) / theta;
    float hp[6] = {0.0f, 0.0f, 0.0f, rot[0] * -0.5f, rot[1] * -0.5f, rot[2] * -0.5f};
    float HM[16];
    pose_M_from_params(hp, HM);
    /* rottrans = halfrotor.GetR() * T  (Matrix3 * Vector3, ORUtils/Matrix.h Matrix3 operator*) */
    float rt[3];
    for (int r = 0; r < 3; r++) rt[r] = HM[r + 4 * 0] * T[0] + HM[r + 4 * 1] * T[1] + HM[r + 4 * 2] * T[2];
    if (theta > 0.001f) {
        float denom = dot3(rot, rot);
        float param = dot3(T, rot) * (1 - 2 * shtot) / denom;
        rt[0] -= rot[0] * param; rt[1] -= rot[1] * param; rt[2] -= rot[2] * param;
    } else {
        float param = dot3(T, rot) / 24;
        rt[0] -= rot[0] * param; rt[1] -= rot[1] * param; rt[2] -= rot[2] * param;
    }
    rt[0] /= 2 * shtot; rt[1] /= 2 * shtot; rt[2] /= 2 * shtot;
    params[3] = rot[0]; params[4] = rot[1]; params[5] = rot[2];
    params[0] = rt[0]; params[1] = rt[1]; params[2] = rt[2];
}

/*
 * pose := SetInvM(c2w); Coerce()  (Core/ITMBasicEngine.tpp:278-279, slam_pipeline.cpp:367-371)
 * c2w given row-major (tensor layout); outputs M / invM in ORUtils layout.
 */
ORC_API void orc_pose_from_c2w(const float *c2w_rowmajor, float *M, float *invM) {
    float c2w[16], M0[16], params[6];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) c2w[c * 4 + r] = c2w_rowmajor[r * 4 + c];
    orc_mat4_inv(c2w, M0);               /* SetInvM: invM.inv(M) */
    pose_params_from_M(M0, params);      /*          SetParamsFromModelView */
    pose_params_from_M(M0, params);      /* Coerce:  SetParamsFromModelView */
    pose_M_from_params(params, M);       /*          SetModelViewFromParams */
    orc_mat4_inv(M, invM);               /* GetInvM */
}

/* ---------------- view building: Engines/ViewBuilding/Shared/ITMViewBuilder_Shared.h:27-36 ---------------- */
static void convert_depth(Tsdf *t, const int16_t *d_in) {
    for (int i = 0; i < t->W * t->H; i++) {
        int16_t d = d_in[i];
        t->depth[i] = d <= 0 ? -1.0f : (float)d * (1.0f / 1000.0f) + 0.0f;
    }
}

/* ---------------- allocation: Engines/Reconstruction/Shared/ITMSceneReconstructionEngine_Shared.h:207-323 ---------------- */
static void alloc_pixel(Tsdf *t, int x, int y, const float *invM, float inv_fx, float inv_fy, float oneOverVoxelSize) {
    float depth_measure = t->depth[x + y * t->W];
    float mu = t->mu;
    if (depth_measure <= 0 || (depth_measure - mu) < 0 || (depth_measure - mu) < t->vf_min || (depth_measure + mu) > t->vf_max) return;
    V4 pc;
    pc.z = depth_measure;
    pc.x = pc.z * (((float)x - t->cx) * inv_fx);
    pc.y = pc.z * (((float)y - t->cy) * inv_fy);
    float norm = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
    V4 pb;
    float s = 1.0f - mu / norm;
    pb.x = pc.x * s; pb.y = pc.y * s; pb.z = pc.z * s; pb.w = 1.0f;
    V4 q = m4_mul_v4(invM, pb);
    float point[3] = {q.x * oneOverVoxelSize, q.y * oneOverVoxelSize, q.z * oneOverVoxelSize};
    s = 1.0f + mu / norm;
    pb.x = pc.x * s; pb.y = pc.y * s; pb.z = pc.z * s; pb.w = 1.0f;
    q = m4_mul_v4(invM, pb);
    float point_e[3] = {q.x * oneOverVoxelSize, q.y * oneOverVoxelSize, q.z * oneOverVoxelSize};
    float dir[3] = {point_e[0] - point[0], point_e[1] - point[1], point_e[2] - point[2]};
    norm = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    int noSteps = (int)ceilf(2.0f * norm);
    float dn = (float)(noSteps - 1);
    dir[0] /= dn; dir[1] /= dn; dir[2] /= dn;
    for (int i = 0; i < noSteps; i++) {
        int16_t bp[3] = {(int16_t)floorf(point[0]), (int16_t)floorf(point[1]), (int16_t)floorf(point[2])};
        int hashIdx = hash_index(bp[0], bp[1], bp[2], t->hash_mask);
        int isFound = 0;
        HashEntry he = t->hash[hashIdx];
        if (he.pos[0] == bp[0] && he.pos[1] == bp[1] && he.pos[2] == bp[2] && he.ptr >= -1) {
            t->visible_type[hashIdx] = (he.ptr == -1) ? 2 : 1;
            isFound = 1;
        }
        if (!isFound) {
            int isExcess = 0;
            if (he.ptr >= -1) {
                while (he.offset >= 1) {
                    hashIdx = t->n_buckets + he.offset - 1;
                    he = t->hash[hashIdx];
                    if (he.pos[0] == bp[0] && he.pos[1] == bp[1] && he.pos[2] == bp[2] && he.ptr >= -1) {
                        t->visible_type[hashIdx] = (he.ptr == -1) ? 2 : 1;
                        isFound = 1;
                        break;
                    }
                }
                isExcess = 1;
            }
            if (!isFound) {
                t->alloc_type[hashIdx] = isExcess ? 2 : 1;
                if (!isExcess) t->visible_type[hashIdx] = 1;
                int16_t *bc = t->block_coords + 4 * (size_t)hashIdx;
                bc[0] = bp[0]; bc[1] = bp[1]; bc[2] = bp[2]; bc[3] = 1;
            }
        }
        point[0] += dir[0]; point[1] += dir[1]; point[2] += dir[2];
    }
}

/* ...Shared.h:326-353 checkPointVisibility<false> / :358-422 checkBlockVisibility<false> */
static int point_visible(const float *M, float px, float py, float pz, float fx, float fy, float cx, float cy, int W, int H) {
    V4 p = {px, py, pz, 1.0f};
    V4 b = m4_mul_v4(M, p);
    if (b.z < 1e-10f) return 0;
    b.x = fx * b.x / b.z + cx;
    b.y = fy * b.y / b.z + cy;
    return (b.x >= 0 && b.x < W && b.y >= 0 && b.y < H);
}
static int block_visible(const int16_t *pos, const float *M, float fx, float fy, float cx, float cy, float voxel, int W, int H) {
    float factor = (float)BLK * voxel;
    float x = (float)pos[0] * factor, y = (float)pos[1] * factor, z = (float)pos[2] * factor;
#define PV() if (point_visible(M, x, y, z, fx, fy, cx, cy, W, H)) return 1
    PV();            /* 0 0 0 */
    z += factor; PV(); /* 0 0 1 */
    y += factor; PV(); /* 0 1 1 */
    x += factor; PV(); /* 1 1 1 */
    z -= factor; PV(); /* 1 1 0 */
    y -= factor; PV(); /* 1 0 0 */
    x -= factor; y += factor; PV(); /* 0 1 0 */
    x += factor; y -= factor; z += factor; PV(); /* 1 0 1 */
#undef PV
    return 0;
}

/* Engines/Reconstruction/CPU/ITMSceneReconstructionEngine_CPU.tpp:129-341 (onlyUpdateVisibleList=false, no swapping) */
static void allocate_scene_from_depth(Tsdf *t, const float *M, const float *invM) {
    float oneOverVoxelSize = 1.0f / (t->voxel * BLK);
    float inv_fx = 1.0f / t->fx, inv_fy = 1.0f / t->fy;
    memset(t->alloc_type, 0, (size_t)t->n_total);
    for (int i = 0; i < t->n_visible; i++) t->visible_type[t->visible_ids[i]] = 3;
    for (int loc = 0; loc < t->W * t->H; loc++) {
        int y = loc / t->W, x = loc - y * t->W;
        alloc_pixel(t, x, y, invM, inv_fx, inv_fy, oneOverVoxelSize);
    }
    int lastFreeVoxelBlockId = t->last_free_block, lastFreeExcessListId = t->last_free_excess;
    for (int idx = 0; idx < t->n_total; idx++) {
        int vbaIdx, exlIdx;
        switch (t->alloc_type[idx]) {
        case 1:
            vbaIdx = lastFreeVoxelBlockId; lastFreeVoxelBlockId--;
            if (vbaIdx >= 0) {
                const int16_t *bc = t->block_coords + 4 * (size_t)idx;
                HashEntry he; memset(&he, 0, sizeof(he));
                he.pos[0] = bc[0]; he.pos[1] = bc[1]; he.pos[2] = bc[2];
                he.ptr = t->vba_alloc_list[vbaIdx]; he.offset = 0;
                t->hash[idx] = he;
            } else { t->visible_type[idx] = 0; lastFreeVoxelBlockId++; }
            break;
        case 2:
            vbaIdx = lastFreeVoxelBlockId; lastFreeVoxelBlockId--;
            exlIdx = lastFreeExcessListId; lastFreeExcessListId--;
            if (vbaIdx >= 0 && exlIdx >= 0) {
                const int16_t *bc = t->block_coords + 4 * (size_t)idx;
                HashEntry he; memset(&he, 0, sizeof(he));
                he.pos[0] = bc[0]; he.pos[1] = bc[1]; he.pos[2] = bc[2];
                he.ptr = t->vba_alloc_list[vbaIdx]; he.offset = 0;
                int exlOffset = t->excess_list[exlIdx];
                t->hash[idx].offset = exlOffset + 1;
                t->hash[t->n_buckets + exlOffset] = he;
                t->visible_type[t->n_buckets + exlOffset] = 1;
            } else { lastFreeVoxelBlockId++; lastFreeExcessListId++; }
            break;
        }
    }
    int n = 0;
    for (int idx = 0; idx < t->n_total; idx++) {
        uint8_t vt = t->visible_type[idx];
        if (vt == 3) {
            if (!block_visible(t->hash[idx].pos, M, t->fx, t->fy, t->cx, t->cy, t->voxel, t->W, t->H)) vt = 0;
            t->visible_type[idx] = vt;
        }
        if (vt > 0) t->visible_ids[n++] = idx;
    }
    t->n_visible = n;
    t->last_free_block = lastFreeVoxelBlockId;
    t->last_free_excess = lastFreeExcessListId;
}

/* ---------------- integration: ...Reconstruction_Shared.h:8-54, 105-139, 157-174; CPU.tpp:52-127 ---------------- */
static void integrate(Tsdf *t, const float *M) {
    const float mu = t->mu;
    const int maxW = t->maxW;
    const int W = t->W, H = t->H;
    for (int e = 0; e < t->n_visible; e++) {
        const HashEntry *he = &t->hash[t->visible_ids[e]];
        if (he->ptr < 0) continue;
        int gx = he->pos[0] * BLK, gy = he->pos[1] * BLK, gz = he->pos[2] * BLK;
        Voxel *blk = t->vba + (size_t)he->ptr * BLK3;
        for (int z = 0; z < BLK; z++) for (int y = 0; y < BLK; y++) for (int x = 0; x < BLK; x++) {
            Voxel *v = &blk[x + y * BLK + z * BLK * BLK];
            V4 pm = {(float)(gx + x) * t->voxel, (float)(gy + y) * t->voxel, (float)(gz + z) * t->voxel, 1.0f};
            V4 pc = m4_mul_v4(M, pm);
            if (pc.z <= 0) continue;
            float ix = t->fx * pc.x / pc.z + t->cx, iy = t->fy * pc.y / pc.z + t->cy;
            if ((ix < 1) || (ix > W - 2) || (iy < 1) || (iy > H - 2)) continue;
            float dm = t->depth[(int)(ix + 0.5f) + (int)(iy + 0.5f) * W];
            if (dm <= 0.0f) continue;
            float eta = dm - pc.z;
            if (eta < -mu) continue;
            float oldF = (float)(v->sdf) / 32767.0f;
            int oldW = v->w_depth;
            float newF = (1.0f < eta / mu) ? 1.0f : eta / mu; /* MIN(1.0f, eta/mu) */
            int newW = 1;
            newF = oldW * oldF + newW * newF;
            newW = oldW + newW;
            newF /= newW;
            newW = (newW < maxW) ? newW : maxW;
            v->sdf = (int16_t)(newF * 32767.0f);
            v->w_depth = (uint8_t)newW;
            /* colour (ComputeUpdatedVoxelInfo<true,false>) */
            if ((eta > mu) || (fabsf(eta / mu) > 0.25f)) continue;
            /* M_rgb == M_d: trafo_rgb_to_depth is identity (slam/InfiniTAM_tools.cpp:6-10) */
            V4 pr = m4_mul_v4(M, pm);
            float rx = t->fx * pr.x / pr.z + t->cx, ry = t->fy * pr.y / pr.z + t->cy;
            if ((rx < 1) || (rx > W - 2) || (ry < 1) || (ry > H - 2)) continue;
            /* interpolateBilinear Utils/ITMPixelUtils.h:11-35 */
            int px = (int)floorf(rx), py = (int)floorf(ry);
            float dx = rx - (float)px, dy = ry - (float)py;
            const uint8_t *a = t->rgb + 4 * (size_t)(px + py * W);
            uint8_t zero[4] = {0, 0, 0, 0};
            const uint8_t *b = zero, *c = zero, *d = zero;
            if (dx != 0) b = t->rgb + 4 * (size_t)((px + 1) + py * W);
            if (dy != 0) c = t->rgb + 4 * (size_t)(px + (py + 1) * W);
            if (dx != 0 && dy != 0) d = t->rgb + 4 * (size_t)((px + 1) + (py + 1) * W);
            float oldWc = (float)v->w_color;
            float nw = 1;
            float sumW = oldWc + nw;
            float cw = (sumW < (uint8_t)maxW) ? sumW : (float)(uint8_t)maxW;
            for (int k = 0; k < 3; k++) {
                float m = ((float)a[k] * (1.0f - dx) * (1.0f - dy) + (float)b[k] * dx * (1.0f - dy) +
                           (float)c[k] * (1.0f - dx) * dy + (float)d[k] * dx * dy);
                float rgb_measure = m / 255.0f;
                float oldC = (float)v->clr[k] / 255.0f;
                float newC = oldC * oldWc + rgb_measure * nw;
                newC /= sumW;
                float s = newC * 255.0f;
                int vi = (int)ROUNDF(s);
                vi = (0 < ((255 < vi) ? 255 : vi)) ? ((255 < vi) ? 255 : vi) : 0; /* CLAMP(vi,0,255) = MAX(0, MIN(255, vi)) */
                v->clr[k] = (uint8_t)vi;
            }
            v->w_color = (uint8_t)cw;
        }
    }
}

/* ---------------- expected depths: Visualisation/Shared/...Shared.h:36-122, CPU.tpp:116-185 ---------------- */
static void create_expected_depths(Tsdf *t, const float *M, const int32_t *vis_ids, int n_vis, V2 *minmax) {
    const int W = t->W, H = t->H;
    for (int i = 0; i < W * H; i++) { minmax[i].x = FAR_AWAY; minmax[i].y = VERY_CLOSE; }
    int numRenderingBlocks = 0;
    for (int k = 0; k < n_vis; k++) {
        const HashEntry *he = &t->hash[vis_ids[k]];
        if (he->ptr < 0) continue;
        int ulx = W / MINMAX_SUB, uly = H / MINMAX_SUB, lrx = -1, lry = -1;
        float zmin = FAR_AWAY, zmax = VERY_CLOSE;
        for (int corner = 0; corner < 8; ++corner) {
            int16_t tx = he->pos[0], ty = he->pos[1], tz = he->pos[2];
            tx += (corner & 1) ? 1 : 0; ty += (corner & 2) ? 1 : 0; tz += (corner & 4) ? 1 : 0;
            V4 p = {(float)tx * (float)BLK * t->voxel, (float)ty * (float)BLK * t->voxel, (float)tz * (float)BLK * t->voxel, 1.0f};
            p = m4_mul_v4(M, p);
            if (p.z < 1e-6) continue;
            float px = (t->fx * p.x / p.z + t->cx) / MINMAX_SUB;
            float py = (t->fy * p.y / p.z + t->cy) / MINMAX_SUB;
            if (ulx > floorf(px)) ulx = (int)floorf(px);
            if (lrx < ceilf(px)) lrx = (int)ceilf(px);
            if (uly > floorf(py)) uly = (int)floorf(py);
            if (lry < ceilf(py)) lry = (int)ceilf(py);
            if (zmin > p.z) zmin = p.z;
            if (zmax < p.z) zmax = p.z;
        }
        if (ulx < 0) ulx = 0;
        if (uly < 0) uly = 0;
        if (lrx >= W) lrx = W - 1;
        if (lry >= H) lry = H - 1;
        if (ulx > lrx) continue;
        if (uly > lry) continue;
        if (zmin < VERY_CLOSE) zmin = VERY_CLOSE;
        if (zmax < VERY_CLOSE) continue;
        int rbx = (int)ceilf((float)(lrx - ulx + 1) / 16.0f), rby = (int)ceilf((float)(lry - uly + 1) / 16.0f);
        int required = rbx * rby;
        if (numRenderingBlocks + required >= MAX_RENDERING_BLOCKS) continue;
        numRenderingBlocks += required;
        /* CreateRenderingBlocks + the fill loop only ever min/max over the bounding box */
        for (int y = uly; y <= lry; ++y)
            for (int x = ulx; x <= lrx; ++x) {
                V2 *px2 = &minmax[x + y * W];
                if (px2->x > zmin) px2->x = zmin;
                if (px2->y < zmax) px2->y = zmax;
            }
    }
}

/* ---------------- voxel reads: Objects/Scene/ITMRepresentationAccess.h ---------------- */
typedef struct { int bx, by, bz, ptr; } Cache;
static Voxel read_voxel(const Tsdf *t, int px, int py, int pz, int *vmIndex, Cache *c) { /* :82-119, :14-24 */
    int bx = ((px < 0) ? px - BLK + 1 : px) / BLK;
    int by = ((py < 0) ? py - BLK + 1 : py) / BLK;
    int bz = ((pz < 0) ? pz - BLK + 1 : pz) / BLK;
    int lin = px + (py - bx) * BLK + (pz - by) * BLK * BLK - bz * BLK3;
    if (bx == c->bx && by == c->by && bz == c->bz) { *vmIndex = 1; return t->vba[c->ptr + lin]; }
    int hashIdx = hash_index(bx, by, bz, t->hash_mask);
    while (1) {
        HashEntry he = t->hash[hashIdx];
        if (he.pos[0] == bx && he.pos[1] == by && he.pos[2] == bz && he.ptr >= 0) {
            c->bx = bx; c->by = by; c->bz = bz; c->ptr = he.ptr * BLK3;
            *vmIndex = hashIdx + 1;
            return t->vba[c->ptr + lin];
        }
        if (he.offset < 1) break;
        hashIdx = t->n_buckets + he.offset - 1;
    }
    *vmIndex = 0;
    return empty_voxel();
}
static float read_sdf_uninterp(const Tsdf *t, const float *p, int *vmIndex, Cache *c) { /* :143-149 */
    Voxel v = read_voxel(t, (int)ROUNDF(p[0]), (int)ROUNDF(p[1]), (int)ROUNDF(p[2]), vmIndex, c);
    return (float)(v.sdf) / 32767.0f;
}
static void floor3(const float *p, int *pos, float *coeff) { /* ORUtils/Vector.h toIntFloor(residual) */
    for (int k = 0; k < 3; k++) { float f = floorf(p[k]); coeff[k] = p[k] - f; pos[k] = (int)f; }
}
static float read_sdf_interp(const Tsdf *t, const float *p, int *vmIndex, Cache *c, float *conf /* may be NULL */) { /* :151-177, :179-235 */
    float res1, res2, v1, v2, res1_c = 0, res2_c = 0, v1_c, v2_c;
    float cf[3]; int pos[3];
    floor3(p, pos, cf);
    Voxel v;
#define RV(dx, dy, dz) read_voxel(t, pos[0] + dx, pos[1] + dy, pos[2] + dz, vmIndex, c)
    v = RV(0, 0, 0); v1 = v.sdf; v1_c = v.w_depth;
    v = RV(1, 0, 0); v2 = v.sdf; v2_c = v.w_depth;
    res1 = (1.0f - cf[0]) * v1 + cf[0] * v2;
    res1_c = (1.0f - cf[0]) * v1_c + cf[0] * v2_c;
    v = RV(0, 1, 0); v1 = v.sdf; v1_c = v.w_depth;
    v = RV(1, 1, 0); v2 = v.sdf; v2_c = v.w_depth;
    res1 = (1.0f - cf[1]) * res1 + cf[1] * ((1.0f - cf[0]) * v1 + cf[0] * v2);
    res1_c = (1.0f - cf[1]) * res1_c + cf[1] * ((1.0f - cf[0]) * v1_c + cf[0] * v2_c);
    v = RV(0, 0, 1); v1 = v.sdf; v1_c = v.w_depth;
    v = RV(1, 0, 1); v2 = v.sdf; v2_c = v.w_depth;
    res2 = (1.0f - cf[0]) * v1 + cf[0] * v2;
    res2_c = (1.0f - cf[0]) * v1_c + cf[0] * v2_c;
    v = RV(0, 1, 1); v1 = v.sdf; v1_c = v.w_depth;
    v = RV(1, 1, 1); v2 = v.sdf; v2_c = v.w_depth;
    res2 = (1.0f - cf[1]) * res2 + cf[1] * ((1.0f - cf[0]) * v1 + cf[0] * v2);
    res2_c = (1.0f - cf[1]) * res2_c + cf[1] * ((1.0f - cf[0]) * v1_c + cf[0] * v2_c);
#undef RV
    *vmIndex = 1;
    if (conf) *conf = (1.0f - cf[2]) * res1_c + cf[2] * res2_c;
    return ((1.0f - cf[2]) * res1 + cf[2] * res2) / 32767.0f;
}

/* castRay: Visualisation/Shared/ITMVisualisationEngine_Shared.h:122-221 */
static int cast_ray(Tsdf *t, V4 *out, uint8_t *visible_type /* NULL = do not modify */, int x, int y, const float *invM,
                    V2 mm) {
    float oneOverVoxelSize = 1.0f / t->voxel;
    float ipx = 1.0f / t->fx, ipy = 1.0f / t->fy, ipz = -t->cx, ipw = -t->cy; /* InvertProjectionParams :31-34 */
    float stepScale = t->mu * oneOverVoxelSize;
    V4 pc;
    pc.z = mm.x;
    pc.x = pc.z * (((float)x + ipz) * ipx);
    pc.y = pc.z * (((float)y + ipw) * ipy);
    pc.w = 1.0f;
    float l2 = 0; l2 += pc.x * pc.x; l2 += pc.y * pc.y; l2 += pc.z * pc.z;
    float totalLength = sqrtf(l2) * oneOverVoxelSize;
    V4 q = m4_mul_v4(invM, pc);
    float ps[3] = {q.x * oneOverVoxelSize, q.y * oneOverVoxelSize, q.z * oneOverVoxelSize};
    pc.z = mm.y;
    pc.x = pc.z * (((float)x + ipz) * ipx);
    pc.y = pc.z * (((float)y + ipw) * ipy);
    pc.w = 1.0f;
    l2 = 0; l2 += pc.x * pc.x; l2 += pc.y * pc.y; l2 += pc.z * pc.z;
    float totalLengthMax = sqrtf(l2) * oneOverVoxelSize;
    q = m4_mul_v4(invM, pc);
    float pe[3] = {q.x * oneOverVoxelSize, q.y * oneOverVoxelSize, q.z * oneOverVoxelSize};
    float rd[3] = {pe[0] - ps[0], pe[1] - ps[1], pe[2] - ps[2]};
    float dn = 1.0f / sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
    rd[0] *= dn; rd[1] *= dn; rd[2] *= dn;
    float pr[3] = {ps[0], ps[1], ps[2]};
    Cache cache = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1};
    float sdfValue = 1.0f, confidence = 0.f, stepLength;
    int vmIndex = 0;
    t->rays++;
    while (totalLength < totalLengthMax) {
        t->ray_steps++;
        sdfValue = read_sdf_uninterp(t, pr, &vmIndex, &cache);
        if (visible_type) { if (vmIndex) visible_type[vmIndex - 1] = 1; }
        if (!vmIndex) {
            stepLength = BLK;
        } else {
            if ((sdfValue <= 0.1f) && (sdfValue >= -0.5f)) sdfValue = read_sdf_interp(t, pr, &vmIndex, &cache, NULL);
            if (sdfValue <= 0.0f) break;
            float a = sdfValue * stepScale;
            stepLength = (a < 1.0f) ? 1.0f : a; /* MAX(a, 1.0f) */
        }
        pr[0] += stepLength * rd[0]; pr[1] += stepLength * rd[1]; pr[2] += stepLength * rd[2];
        totalLength += stepLength;
    }
    int found;
    if (sdfValue <= 0.0f) {
        stepLength = sdfValue * stepScale;
        pr[0] += stepLength * rd[0]; pr[1] += stepLength * rd[1]; pr[2] += stepLength * rd[2];
        sdfValue = read_sdf_interp(t, pr, &vmIndex, &cache, &confidence);
        stepLength = sdfValue * stepScale;
        pr[0] += stepLength * rd[0]; pr[1] += stepLength * rd[1]; pr[2] += stepLength * rd[2];
        found = 1;
    } else found = 0;
    out->x = pr[0]; out->y = pr[1]; out->z = pr[2];
    out->w = found ? confidence + 1.0f : 0.0f;
    return found;
}

/* GenericRaycast: Visualisation/CPU/ITMVisualisationEngine_CPU.tpp:187-236 */
static void generic_raycast(Tsdf *t, const float *invM, const V2 *minmax, V4 *rays, uint8_t *visible_type) {
    const int W = t->W, H = t->H;
    for (int loc = 0; loc < W * H; ++loc) {
        int y = loc / W, x = loc - y * W;
        int loc2 = (int)floorf((float)x / MINMAX_SUB) + (int)floorf((float)y / MINMAX_SUB) * W;
        cast_ray(t, &rays[loc], visible_type, x, y, invM, minmax[loc2]);
    }
}

/* processPixelICP<true,false>: Visualisation/Shared/...Shared.h:252-330 (useSmoothing), :438-480 */
static void icp_maps(Tsdf *t, const float *invM) {
    const int W = t->W, H = t->H;
    const V4 *pr = t->raycast;
    float light[3] = {-invM[8], -invM[9], -invM[10]}; /* -invM.getColumn(2) */
    float vs = t->voxel;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        int loc = x + y * W;
        V4 point = pr[loc];
        int found = point.w > 0.0f;
        float n[3] = {0, 0, 0};
        if (found) {
            if (y <= 2 || y >= H - 3 || x <= 2 || x >= W - 3) found = 0;
        }
        if (found) {
            V4 xp = pr[(x + 2) + y * W], yp = pr[x + (y + 2) * W], xm = pr[(x - 2) + y * W], ym = pr[x + (y - 2) * W];
            float dxv[4] = {0, 0, 0, 0}, dyv[4] = {0, 0, 0, 0};
            int doPlus1 = 0;
            if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) doPlus1 = 1;
            else {
                dxv[0] = xp.x - xm.x; dxv[1] = xp.y - xm.y; dxv[2] = xp.z - xm.z; dxv[3] = xp.w - xm.w;
                dyv[0] = yp.x - ym.x; dyv[1] = yp.y - ym.y; dyv[2] = yp.z - ym.z; dyv[3] = yp.w - ym.w;
                float la = dxv[0] * dxv[0] + dxv[1] * dxv[1] + dxv[2] * dxv[2];
                float lb = dyv[0] * dyv[0] + dyv[1] * dyv[1] + dyv[2] * dyv[2];
                float ld = (la < lb) ? lb : la;
                if (ld * vs * vs > (0.15f * 0.15f)) doPlus1 = 1;
            }
            if (doPlus1) {
                xp = pr[(x + 1) + y * W]; yp = pr[x + (y + 1) * W]; xm = pr[(x - 1) + y * W]; ym = pr[x + (y - 1) * W];
                dxv[0] = xp.x - xm.x; dxv[1] = xp.y - xm.y; dxv[2] = xp.z - xm.z;
                dyv[0] = yp.x - ym.x; dyv[1] = yp.y - ym.y; dyv[2] = yp.z - ym.z;
                if (xp.w <= 0 || yp.w <= 0 || xm.w <= 0 || ym.w <= 0) found = 0;
            }
            if (found) {
                n[0] = -(dxv[1] * dyv[2] - dxv[2] * dyv[1]);
                n[1] = -(dxv[2] * dyv[0] - dxv[0] * dyv[2]);
                n[2] = -(dxv[0] * dyv[1] - dxv[1] * dyv[0]);
                float ns = 1.0f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                n[0] *= ns; n[1] *= ns; n[2] *= ns;
                float angle = n[0] * light[0] + n[1] * light[1] + n[2] * light[2];
                if (!(angle > 0.0)) found = 0;
            }
        }
        if (found) {
            V4 o = {point.x * vs, point.y * vs, point.z * vs, point.w};
            t->icp_points[loc] = o;
            V4 nn = {n[0], n[1], n[2], 0.0f};
            t->icp_normals[loc] = nn;
        } else {
            V4 o = {0.0f, 0.0f, 0.0f, -1.0f};
            t->icp_points[loc] = o; t->icp_normals[loc] = o;
        }
    }
}

/* readFromSDF_color4u_interpolated (GPS-SLAM fork: renormalised over w_color >= 1):
 * Objects/Scene/ITMRepresentationAccess.h:344-423; drawPixelColour Visualisation/Shared:...:384-394 */
static void render_colour(Tsdf *t, const V4 *rays, uint8_t *out) {
    for (int loc = 0; loc < t->W * t->H; loc++) {
        V4 r = rays[loc];
        uint8_t *o = out + 4 * (size_t)loc;
        if (!(r.w > 0)) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
        float p[3] = {r.x, r.y, r.z}, cf[3]; int pos[3];
        floor3(p, pos, cf);
        Cache c = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1};
        float ret[3] = {0, 0, 0}, wsum = 0.f;
        int vm;
        for (int k = 0; k < 8; k++) {
            int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
            Voxel v = read_voxel(t, pos[0] + dx, pos[1] + dy, pos[2] + dz, &vm, &c);
            if (v.w_color >= 1) {
                float wx = dx ? cf[0] : (1.0f - cf[0]);
                float wy = dy ? cf[1] : (1.0f - cf[1]);
                float wz = dz ? cf[2] : (1.0f - cf[2]);
                float w = wx * wy * wz; /* ((a)*(b))*(c) as written in the reference */
                ret[0] += w * (float)v.clr[0]; ret[1] += w * (float)v.clr[1]; ret[2] += w * (float)v.clr[2];
                wsum += w;
            }
        }
        ret[0] /= wsum; ret[1] /= wsum; ret[2] /= wsum;
        float c4[3] = {ret[0] / 255.0f, ret[1] / 255.0f, ret[2] / 255.0f};
        o[0] = (uint8_t)(c4[0] * 255.0f); o[1] = (uint8_t)(c4[1] * 255.0f); o[2] = (uint8_t)(c4[2] * 255.0f); o[3] = 255;
    }
}

/* ---------------- public frame-level entry points ---------------- */
/* ITMBasicEngine::ProcessFrame with tracking off (Core/ITMBasicEngine.tpp:260-385):
 * UpdateView -> DenseMapper::ProcessFrame (allocate + integrate) -> TrackingController::Prepare
 * (CreateExpectedDepths + CreateICPMaps with visible-list update). */
ORC_API void orc_tsdf_process_frame(Tsdf *t, const uint8_t *rgb4, const int16_t *depth_mm, const float *M, const float *invM) {
    memcpy(t->rgb, rgb4, 4 * (size_t)t->W * t->H);
    convert_depth(t, depth_mm);
    allocate_scene_from_depth(t, M, invM);
    integrate(t, M);
    create_expected_depths(t, M, t->visible_ids, t->n_visible, t->minmax);
    generic_raycast(t, invM, t->minmax, t->raycast, t->visible_type);
    icp_maps(t, invM);
}

/* ITMBasicEngine::runRaycast(pose, intrinsics) (Core/ITMBasicEngine.tpp:519-525):
 * FindVisibleBlocks (CPU.tpp:36-74) -> CreateExpectedDepths -> RenderImage(COLOUR_FROM_VOLUME) */
ORC_API void orc_tsdf_free_raycast(Tsdf *t, const float *M, const float *invM) {
    int n = 0;
    for (int idx = 0; idx < t->n_total; idx++) {
        const HashEntry *he = &t->hash[idx];
        if (he->ptr >= 0 && block_visible(he->pos, M, t->fx, t->fy, t->cx, t->cy, t->voxel, t->W, t->H)) t->fv_visible_ids[n++] = idx;
    }
    t->fv_n_visible = n;
    create_expected_depths(t, M, t->fv_visible_ids, n, t->fv_minmax);
    generic_raycast(t, invM, t->fv_minmax, t->fv_raycast, NULL);
    render_colour(t, t->fv_raycast, t->fv_colour);
}

/* ======================================================================================================================
 * ExtendedTracker, depth only -- the tracker ITMLibSettings.cpp:54-57 configures ("type=extended,levels=rrbb,useDepth=1,
 * minstep=1e-4,outlierSpaceC=0.1,outlierSpaceF=0.004,numiterC=20,numiterF=50,tukeyCutOff=8,framesToSkip=20,
 * framesToWeight=50") and ITMBasicEngine runs when use_gt_pose is false.
 *   ITMLib/Trackers/Interface/ITMExtendedTracker.cpp:143-177 (SetupLevels), :216-268 (PrepareForEvaluation),
 *   :293-375 (ComputeDelta / HasConverged / ApplyDelta), :470-665 (TrackCamera), :377-468 (UpdatePoseQuality: score only);
 *   Trackers/Shared/ITMExtendedTracker_Shared.h:51-143, 298-328 (per point); Trackers/CPU/ITMExtendedTracker_CPU.cpp:46-156
 *   (accumulation in scan order); Engines/LowLevel/Shared/ITMLowLevelEngine_Shared.h:48-69 (depth pyramid);
 *   Utils/ITMPixelUtils.h:78-106 (bilinear with holes); ORUtils/Cholesky.h.
 * Stateless: everything comes in through arguments, so the same function pins the HIP tracker in tests.
 * ==================================================================================================================== */
#define TRK_MAX_LEVELS 8
enum { TRK_ROTATION = 0, TRK_TRANSLATION = 1, TRK_BOTH = 2, TRK_NONE = 3 };

typedef struct {
    int n_levels;
    int iter_type[TRK_MAX_LEVELS];    /* level 0 = finest */
    int n_iter[TRK_MAX_LEVELS];
    float space_thresh[TRK_MAX_LEVELS];
    float term_thresh, tukey_cutoff, vf_min, vf_max;
    int frames_to_skip, frames_to_weight;
} TrackCfg;

/* levels string as in the config ("rrbb": parsed from the END, so level 0 = 'b'); SetupLevels' float stepping */
ORC_API void orc_track_config(const char *levels, int num_iter_coarse, int num_iter_fine, float thresh_coarse,
                              float thresh_fine, float term_thresh, float tukey, int frames_to_skip, int frames_to_weight,
                              float vf_min, float vf_max, TrackCfg *c) {
    int n = (int)strlen(levels);
    c->n_levels = n;
    for (int i = n - 1, k = 0; i >= 0; --i, ++k)
        c->iter_type[k] = levels[i] == 'r' ? TRK_ROTATION : levels[i] == 't' ? TRK_TRANSLATION : levels[i] == 'b' ? TRK_BOTH : TRK_NONE;
    {
        float step = (float)(num_iter_coarse - num_iter_fine) / (float)(n - 1);
        float val = (float)num_iter_coarse;
        for (int l = n - 1; l >= 0; l--) { c->n_iter[l] = (int)round(val); val -= step; }
    }
    {
        float step = (float)(thresh_coarse - thresh_fine) / (float)(n - 1);
        float val = thresh_coarse;
        for (int l = n - 1; l >= 0; l--) { c->space_thresh[l] = val; val -= step; }
    }
    c->term_thresh = term_thresh; c->tukey_cutoff = tukey; c->frames_to_skip = frames_to_skip;
    c->frames_to_weight = frames_to_weight; c->vf_min = vf_min; c->vf_max = vf_max;
}

ORC_API void orc_filter_subsample_with_holes(const float *in, int w_in, int h_in, float *out) {
    int w = w_in / 2, h = h_in / 2;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        float acc = 0.0f, good = 0.0f, v;
        v = in[(2 * x + 0) + (2 * y + 0) * w_in]; if (v > 0.0f) { acc += v; good++; }
        v = in[(2 * x + 1) + (2 * y + 0) * w_in]; if (v > 0.0f) { acc += v; good++; }
        v = in[(2 * x + 0) + (2 * y + 1) * w_in]; if (v > 0.0f) { acc += v; good++; }
        v = in[(2 * x + 1) + (2 * y + 1) * w_in]; if (v > 0.0f) { acc += v; good++; }
        if (good > 0) acc /= good;
        out[x + y * w] = acc;
    }
}

static V4 bilinear_with_holes(const V4 *src, float px, float py, int W) {
    const short ix = (short)floorf(px), iy = (short)floorf(py);
    const float dx = px - (float)ix, dy = py - (float)iy;
    const V4 a = src[ix + iy * W], b = src[(ix + 1) + iy * W], c = src[ix + (iy + 1) * W], d = src[(ix + 1) + (iy + 1) * W];
    V4 r;
    if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) { r.x = 0; r.y = 0; r.z = 0; r.w = -1.0f; return r; }
    r.x = (a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy);
    r.y = (a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy);
    r.z = (a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy);
    r.w = (a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy);
    return r;
}

/* computePerPointGH_exDepth_Ab; returns 0 if the point does not contribute.  A has 6 (both) or 3 entries. */
static int track_point_Ab(float *A, float *b, float *depth_weight, int x, int y, float depth, const float *view_intr,
                          int sceneW, int sceneH, const float *scene_intr, const float *approxInvPose, const float *scenePose,
                          const V4 *points, const V4 *normals, float space_thresh, const TrackCfg *c, int iter_type,
                          int use_weights) {
    *depth_weight = 0;
    if (depth <= 1e-8f) return 0;
    V4 p, q;
    p.x = depth * (((float)x - view_intr[2]) / view_intr[0]);
    p.y = depth * (((float)y - view_intr[3]) / view_intr[1]);
    p.z = depth; p.w = 1.0f;
    p = m4_mul_v4(approxInvPose, p); p.w = 1.0f;
    q = m4_mul_v4(scenePose, p);
    if (q.z <= 0.0f) return 0;
    const float u = scene_intr[0] * q.x / q.z + scene_intr[2];
    const float v = scene_intr[1] * q.y / q.z + scene_intr[3];
    if (!((u >= 0.0f) && (u <= sceneW - 2) && (v >= 0.0f) && (v <= sceneH - 2))) return 0;
    const V4 cp = bilinear_with_holes(points, u, v, sceneW);
    if (cp.w < 0.0f) return 0;
    const float dx = cp.x - p.x, dy = cp.y - p.y, dz = cp.z - p.z;
    const float dist = dx * dx + dy * dy + dz * dz;
    if (dist > c->tukey_cutoff * space_thresh) return 0;
    const V4 n = bilinear_with_holes(normals, u, v, sceneW);
    float w = 1.0f - (depth - c->vf_min) / (c->vf_max - c->vf_min);
    w = w > 0.0f ? w : 0.0f;  /* MAX(0.0f, .) */
    w *= w;
    if (use_weights) {
        if (cp.w < c->frames_to_skip) return 0;
        w *= (cp.w - c->frames_to_skip) / c->frames_to_weight;
    }
    *depth_weight = w;
    *b = n.x * dx + n.y * dy + n.z * dz;
    if (iter_type == TRK_ROTATION || iter_type == TRK_BOTH) {
        A[0] = +p.z * n.y - p.y * n.z;
        A[1] = -p.z * n.x + p.x * n.z;
        A[2] = +p.y * n.x - p.x * n.y;
        if (iter_type == TRK_BOTH) { A[3] = n.x; A[4] = n.y; A[5] = n.z; }
    } else {
        A[0] = n.x; A[1] = n.y; A[2] = n.z;
    }
    return 1;
}

static float trk_rho(float r, float h) { float t = fabsf(r) - h; t = t > 0.0f ? t : 0.0f; return r * r - t * t; }
static float trk_rho_deriv(float r, float h) { float cl = r < -h ? -h : (r > h ? h : r); return 2.0f * cl; }
static float trk_rho_deriv2(float r, float h) { return fabsf(r) < h ? 2.0f : 0.0f; }

/* ITMExtendedTracker_CPU::ComputeGandH_Depth: scan-order float accumulation; hessian is 6x6 (only noPara x noPara filled) */
ORC_API int orc_track_gh_depth(const float *depth, int vw, int vh, const float *view_intr, const V4 *points, const V4 *normals,
                               int sw, int sh, const float *scene_intr, const float *approxInvPose, const float *scenePose,
                               const TrackCfg *c, int level, int frames_processed, float *f_out, float *nabla, float *hessian) {
    const int iter_type = c->iter_type[level];
    if (iter_type == TRK_NONE) return 0;
    const int short_it = iter_type != TRK_BOTH;
    const int noPara = short_it ? 3 : 6, noParaSQ = short_it ? 6 : 21;
    float sumH[21], sumN[6], sumF = 0.0f;
    int n_valid = 0;
    memset(sumH, 0, sizeof(sumH)); memset(sumN, 0, sizeof(sumN));
    const int use_weights = frames_processed >= 100;
    const float thr = c->space_thresh[level];
    for (int y = 0; y < vh; y++) for (int x = 0; x < vw; x++) {
        float A[6], b, w;
        if (!track_point_Ab(A, &b, &w, x, y, depth[x + y * vw], view_intr, sw, sh, scene_intr, approxInvPose, scenePose, points,
                            normals, thr, c, iter_type, use_weights)) continue;
        n_valid++;
        sumF += trk_rho(b, thr) * w;
        for (int r = 0, counter = 0; r < noPara; r++) {
            sumN[r] += trk_rho_deriv(b, thr) * w * A[r];
            for (int cc = 0; cc <= r; cc++, counter++) sumH[counter] += trk_rho_deriv2(b, thr) * w * A[r] * A[cc];
        }
    }
    (void)noParaSQ;
    for (int r = 0, counter = 0; r < noPara; r++) for (int cc = 0; cc <= r; cc++, counter++) hessian[r + cc * 6] = sumH[counter];
    for (int r = 0; r < noPara; ++r) for (int cc = r + 1; cc < noPara; cc++) hessian[r + cc * 6] = hessian[cc + r * 6];
    memcpy(nabla, sumN, noPara * sizeof(float));
    *f_out = sumF;
    return n_valid;
}

/* ORUtils::Cholesky (GenericCholesky<float>) */
typedef struct { float ch[36]; int size; } Chol;
static void chol_init(Chol *k, const float *mat, int size) {
    k->size = size;
    for (int i = 0; i < size * size; i++) k->ch[i] = mat[i];
    for (int c = 0; c < size; c++) {
        float inv_diag = 1;
        for (int r = c; r < size; r++) {
            float val = k->ch[c + r * size];
            for (int c2 = 0; c2 < c; c2++) val -= k->ch[c + c2 * size] * k->ch[c2 + r * size];
            if (r == c) { k->ch[c + r * size] = val; inv_diag = 1.0f / val; }
            else { k->ch[r + c * size] = val; k->ch[c + r * size] = val * inv_diag; }
        }
    }
}
static void chol_backsub(const Chol *k, float *result, const float *v) {
    const int size = k->size;
    float y[6];
    for (int i = 0; i < size; i++) {
        float val = v[i];
        for (int j = 0; j < i; j++) val -= k->ch[j + i * size] * y[j];
        y[i] = val;
    }
    for (int i = 0; i < size; i++) y[i] /= k->ch[i + i * size];
    for (int i = size - 1; i >= 0; i--) {
        float val = y[i];
        for (int j = i + 1; j < size; j++) val -= k->ch[i + j * size] * result[j];
        result[i] = val;
    }
}
static float chol_det(const Chol *k) {
    float ret = 1.0f;
    for (int i = 0; i < k->size; ++i) ret *= k->ch[i + i * k->size];
    return ret * ret;
}

static void m4_mul(const float *a, const float *b, float *out) { /* ORUtils/Matrix.h Matrix4 operator* : out = a * b */
    float r[16];
    for (int col = 0; col < 4; col++) for (int row = 0; row < 4; row++) {
        float acc = 0;
        for (int k = 0; k < 4; k++) acc += a[k * 4 + row] * b[col * 4 + k];
        r[col * 4 + row] = acc;
    }
    memcpy(out, r, sizeof(r));
}

/* pose_d->SetInvM(m); Coerce(); -> M, invM (ORUtils layout in and out) */
static void pose_set_invM_coerce(const float *invM_in, float *M, float *invM) {
    float rm[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) rm[r * 4 + c] = invM_in[c * 4 + r];
    orc_pose_from_c2w(rm, M, invM);
}

/* ITMExtendedTracker::TrackCamera for useDepth && !useColour.
 * depth0: view->depth [H*W] (metres, <= 0 invalid); points/normals: trackingState->pointCloud (CreateICPMaps output, full
 * resolution); scenePose = pose_pointCloud->GetM(); pose_M in/out = pose_d->GetM(); pose_invM out.
 * diag[0..n_levels-1] = iterations run per level, diag[8] = noValidPoints of the last accepted evaluation,
 * diag[9] = its f, diag[10] = trackerScore (finalResidual_v2).  scratch: >= (W/2)*(H/2)*4/3 floats.
 * Returns the tracker-level framesProcessed to carry to the next call. */
/* accept / reject decisions of the last orc_track_camera call, e.g. "3:AAR 2:AA 1:AARA 0:AA" (a probe aid for tools/: which
 * evaluations of the LM loop are rejections, and where in a level they fall) */
static char g_track_trace[1024];
ORC_API const char *orc_track_trace(void) { return g_track_trace; }

ORC_API int orc_track_camera(int W, int H, const float *intr /* fx fy cx cy */, const float *depth0, const V4 *points,
                             const V4 *normals, const float *scenePose, float *pose_M, float *pose_invM, const TrackCfg *c,
                             int frames_processed, float *scratch, float *diag) {
    /* PrepareForEvaluation: depth pyramid; intrinsics halve per level; the scene side always stays at level 0 */
    const float *dl[TRK_MAX_LEVELS];
    int lw[TRK_MAX_LEVELS], lh[TRK_MAX_LEVELS];
    float lintr[TRK_MAX_LEVELS][4];
    dl[0] = depth0; lw[0] = W; lh[0] = H;
    memcpy(lintr[0], intr, 16);
    float *sp = scratch;
    for (int l = 1; l < c->n_levels; l++) {
        lw[l] = lw[l - 1] / 2; lh[l] = lh[l - 1] / 2;
        orc_filter_subsample_with_holes(dl[l - 1], lw[l - 1], lh[l - 1], sp);
        dl[l] = sp; sp += lw[l] * lh[l];
        for (int k = 0; k < 4; k++) lintr[l][k] = lintr[l - 1][k] * 0.5f;
    }
    float hessian_good[36], nabla_good[6], hessian_depth_good[36], f_depth_good = 0;
    int nvalid_depth_good = 0;
    memset(hessian_good, 0, sizeof(hessian_good)); memset(nabla_good, 0, sizeof(nabla_good));
    memset(hessian_depth_good, 0, sizeof(hessian_depth_good));
    float M[16], invM[16];
    memcpy(M, pose_M, 64);
    orc_mat4_inv(M, invM);
    int last_type = TRK_NONE;
    for (int k = 0; k < 16; k++) diag[k] = 0;
    int tp = 0;
    g_track_trace[0] = 0;
    for (int level = c->n_levels - 1; level >= 0; level--) {
        const int it = c->iter_type[level];
        if (it == TRK_NONE) continue;
        last_type = it;
        if (tp < (int)sizeof(g_track_trace) - 8) tp += sprintf(g_track_trace + tp, "%s%d:", tp ? " " : "", level);
        float approxInvPose[16], lastGoodM[16], lastGoodInvM[16];
        memcpy(approxInvPose, invM, 64);
        memcpy(lastGoodM, M, 64); memcpy(lastGoodInvM, invM, 64);
        float f_old = 3.402823466e+38f, lambda = 1.0f;
        for (int iter = 0; iter < c->n_iter[level]; iter++) {
            float hessian_depth[36], nabla_depth[6], f_depth = 0.f;
            memset(hessian_depth, 0, sizeof(hessian_depth)); memset(nabla_depth, 0, sizeof(nabla_depth));
            int nvalid = orc_track_gh_depth(dl[level], lw[level], lh[level], lintr[level], points, normals, W, H, lintr[0],
                                            approxInvPose, scenePose, c, level, frames_processed, &f_depth, nabla_depth,
                                            hessian_depth);
            if (nvalid > 100) {
                for (int i = 0; i < 36; ++i) hessian_depth[i] /= nvalid;
                for (int i = 0; i < 6; ++i) nabla_depth[i] /= nvalid;
                f_depth /= nvalid;
            } else {
                f_depth = 3.402823466e+38f;
            }
            diag[level] += 1;
            if (tp < (int)sizeof(g_track_trace) - 8) { g_track_trace[tp++] = ((nvalid <= 0) || (f_depth >= f_old)) ? 'R' : 'A'; g_track_trace[tp] = 0; }
            if ((nvalid <= 0) || (f_depth >= f_old)) {
                memcpy(M, lastGoodM, 64); memcpy(invM, lastGoodInvM, 64);
                memcpy(approxInvPose, invM, 64);
                lambda *= 10.0f;
            } else {
                memcpy(lastGoodM, M, 64); memcpy(lastGoodInvM, invM, 64);
                f_old = f_depth;
                memcpy(hessian_good, hessian_depth, sizeof(hessian_good));
                memcpy(nabla_good, nabla_depth, sizeof(nabla_good));
                lambda /= 10.0f;
                nvalid_depth_good = nvalid; f_depth_good = f_depth;
                memcpy(hessian_depth_good, hessian_depth, sizeof(hessian_depth));
            }
            float A[36];
            for (int i = 0; i < 36; ++i) A[i] = hessian_good[i];
            for (int i = 0; i < 6; ++i) A[i + i * 6] *= 1.0f + lambda;
            float step[6] = {0, 0, 0, 0, 0, 0};
            Chol ch;
            if (it != TRK_BOTH) {
                float small[9];
                for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) small[r + cc * 3] = A[r + cc * 6];
                chol_init(&ch, small, 3);
            } else {
                chol_init(&ch, A, 6);
            }
            chol_backsub(&ch, step, nabla_good);
            float s6[6] = {0, 0, 0, 0, 0, 0};
            if (it == TRK_ROTATION) { s6[0] = step[0]; s6[1] = step[1]; s6[2] = step[2]; }
            else if (it == TRK_TRANSLATION) { s6[3] = step[0]; s6[4] = step[1]; s6[5] = step[2]; }
            else { for (int i = 0; i < 6; i++) s6[i] = step[i]; }
            float Tinc[16];
            Tinc[0 * 4 + 0] = 1.0f;   Tinc[1 * 4 + 0] = s6[2];  Tinc[2 * 4 + 0] = -s6[1]; Tinc[3 * 4 + 0] = s6[3];
            Tinc[0 * 4 + 1] = -s6[2]; Tinc[1 * 4 + 1] = 1.0f;   Tinc[2 * 4 + 1] = s6[0];  Tinc[3 * 4 + 1] = s6[4];
            Tinc[0 * 4 + 2] = s6[1];  Tinc[1 * 4 + 2] = -s6[0]; Tinc[2 * 4 + 2] = 1.0f;   Tinc[3 * 4 + 2] = s6[5];
            Tinc[0 * 4 + 3] = 0.0f;   Tinc[1 * 4 + 3] = 0.0f;   Tinc[2 * 4 + 3] = 0.0f;   Tinc[3 * 4 + 3] = 1.0f;
            m4_mul(Tinc, approxInvPose, approxInvPose);
            pose_set_invM_coerce(approxInvPose, M, invM);
            memcpy(approxInvPose, invM, 64);
            int converged = 1;
            for (int i = 0; i < 6; i++) if (fabs(step[i]) > c->term_thresh) { converged = 0; break; }
            if (converged) break;
        }
    }
    memcpy(pose_M, M, 64); memcpy(pose_invM, invM, 64);
    /* UpdatePoseQuality: the residual score; the SVM verdict only feeds failure modes that are off by default
     * (ITMLibSettings.cpp:42 behaviourOnFailure = FAILUREMODE_IGNORE) */
    {
        int n_max = 0;
        for (int i = 0; i < W * H; i++) if (depth0[i] > 0.0f) n_max++;  /* CountValidDepths */
        float score = sqrtf(((float)nvalid_depth_good * f_depth_good + (float)(n_max - nvalid_depth_good) * c->space_thresh[0]) /
                            (float)n_max);
        diag[8] = (float)nvalid_depth_good; diag[9] = f_depth_good; diag[10] = score;
        float det = 0.0f;
        if (last_type == TRK_BOTH) { Chol ch; chol_init(&ch, hessian_depth_good, 6); det = chol_det(&ch); if (isnan(det)) det = 0.0f; }
        diag[11] = det;
    }
    return frames_processed;
}

/* ---------------- meshing: Engines/Meshing/CPU/ITMMeshingEngine_CPU.tpp:8-55, Shared/ITMMeshingEngine_Shared.h:279-471 ----------------
 * Triangles in the CPU engine's order (hash entry id, then z, y, x, then the case table's order), laid out as ITMMesh::Triangle
 * (Objects/Meshing/ITMMesh.h:18-21): p0 p1 p2 c0 c1 c2 clr, 21 floats.  Positions follow the CPU engine; the per-vertex colours
 * and clr are what the CUDA engine additionally stores (ITMMeshingEngine_CUDA.tcu:118-131) -- the CPU engine leaves them 0.
 * Returns noTotalTriangles (the CPU engine stops advancing at max_triangles - 1). */
#include "../gps_slam_amd/csrc/mc_cases.inc"
static const unsigned long long MC_CASES[256] = GPS_MC_CASES_INIT;
static const int MC_CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
static const int MC_EDGE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

static void sdf_interp3(const float *p1, const float *p2, float v1, float v2, float *out) { /* Shared.h:359-369 */
    if (fabsf(0.0f - v1) < 0.00001f) { out[0] = p1[0]; out[1] = p1[1]; out[2] = p1[2]; return; }
    if (fabsf(0.0f - v2) < 0.00001f) { out[0] = p2[0]; out[1] = p2[1]; out[2] = p2[2]; return; }
    if (fabsf(v1 - v2) < 0.00001f) { out[0] = p1[0]; out[1] = p1[1]; out[2] = p1[2]; return; }
    const float f = (0.0f - v1) / (v2 - v1);
    for (int k = 0; k < 3; k++) out[k] = p1[k] + f * (p2[k] - p1[k]);
}

ORC_API int64_t orc_tsdf_mesh(const Tsdf *t, int64_t max_triangles, float *tris /* [max_triangles][21] */) {
    int64_t n = 0;
    for (int entry = 0; entry < t->n_total; entry++) {
        const HashEntry he = t->hash[entry];
        if (he.ptr < 0) continue;
        const int gx = he.pos[0] * BLK, gy = he.pos[1] * BLK, gz = he.pos[2] * BLK;
        for (int z = 0; z < BLK; z++) for (int y = 0; y < BLK; y++) for (int x = 0; x < BLK; x++) {
            float pts[8][3], sdf[8], col[8][3];
            int ok = 1;
            for (int k = 0; k < 8 && ok; k++) { /* findPointNeighbors: stops at the first missing / untouched corner */
                const int px = gx + x + MC_CORNER[k][0], py = gy + y + MC_CORNER[k][1], pz = gz + z + MC_CORNER[k][2];
                Cache c = {0x7fffffff, 0, 0, 0};
                int vm;
                const Voxel v = read_voxel(t, px, py, pz, &vm, &c);
                pts[k][0] = (float)px; pts[k][1] = (float)py; pts[k][2] = (float)pz;
                sdf[k] = (float)v.sdf / 32767.0f;
                for (int q = 0; q < 3; q++) col[k][q] = v.clr[q] / 255.0f;
                if (!vm || sdf[k] == 1.0f) ok = 0;
            }
            if (!ok) continue;
            int cube = 0;
            for (int k = 0; k < 8; k++) if (sdf[k] < 0) cube |= 1 << k;
            const unsigned long long list = MC_CASES[cube];
            if ((list & 0xF) == 0xF) continue; /* edgeTable[cube] == 0 */
            float vert[12][3], vcol[12][3];
            unsigned cut = 0;
            for (int j = 0; j < 15 && ((list >> (4 * j)) & 0xF) != 0xF; j++) cut |= 1u << ((list >> (4 * j)) & 0xF);
            for (int e = 0; e < 12; e++) {
                if (!(cut & (1u << e))) continue;
                const int a = MC_EDGE[e][0], b = MC_EDGE[e][1];
                sdf_interp3(pts[a], pts[b], sdf[a], sdf[b], vert[e]);
                sdf_interp3(col[a], col[b], sdf[a], sdf[b], vcol[e]);
            }
            for (int j = 0; j < 15 && ((list >> (4 * j)) & 0xF) != 0xF; j += 3) {
                float *o = tris + n * 21;
                for (int q = 0; q < 3; q++) {
                    const int e = (int)((list >> (4 * (j + q))) & 0xF);
                    for (int d = 0; d < 3; d++) { o[3 * q + d] = vert[e][d] * t->voxel; o[9 + 3 * q + d] = vcol[e][d]; }
                }
                for (int d = 0; d < 3; d++) o[18 + d] = col[0][d]; /* VoxelColorReader::uninterpolate at (x,y,z) */
                if (n < max_triangles - 1) n++;
            }
        }
    }
    return n;
}

#define GETTER(name, type, expr) ORC_API type orc_tsdf_##name(Tsdf *t) { return expr; }

/* ITMBasicEngine::ProcessFrame with trackingActive (Core/ITMBasicEngine.tpp:260-385), default failure mode IGNORE:
 * UpdateView -> TrackingController::Track (skipped while there is no point cloud) -> fusion -> Prepare
 * (CreateExpectedDepths + CreateICPMaps, pose_pointCloud := pose_d; Core/ITMTrackingController.h:71-105). */
ORC_API void orc_tsdf_process_frame_tracked(Tsdf *t, const uint8_t *rgb4, const int16_t *depth_mm, const TrackCfg *cfg,
                                            float *M_out, float *invM_out) {
    memcpy(t->rgb, rgb4, 4 * (size_t)t->W * t->H);
    convert_depth(t, depth_mm);
    if (t->age_point_cloud != -1) {  /* HasValidPointCloud */
        if (t->age_point_cloud >= 0) t->trk_frames++; else t->trk_frames = 0;
        if (!t->trk_scratch) t->trk_scratch = (float *)malloc(sizeof(float) * (size_t)t->W * t->H);
        const float intr[4] = {t->fx, t->fy, t->cx, t->cy};
        orc_track_camera(t->W, t->H, intr, t->depth, t->icp_points, t->icp_normals, t->pose_pc_M, t->pose_M, t->pose_invM, cfg,
                         t->trk_frames, t->trk_scratch, t->trk_diag);
    }
    allocate_scene_from_depth(t, t->pose_M, t->pose_invM);
    integrate(t, t->pose_M);
    create_expected_depths(t, t->pose_M, t->visible_ids, t->n_visible, t->minmax);
    generic_raycast(t, t->pose_invM, t->minmax, t->raycast, t->visible_type);
    icp_maps(t, t->pose_invM);
    memcpy(t->pose_pc_M, t->pose_M, 64);
    t->age_point_cloud = (t->age_point_cloud == -1) ? -2 : 0;
    memcpy(M_out, t->pose_M, 64); memcpy(invM_out, t->pose_invM, 64);
}
GETTER(trk_diag, void *, t->trk_diag)

/* accessors for the python side */
GETTER(n_visible, int, t->n_visible)
GETTER(ray_steps, int64_t, t->ray_steps)
GETTER(rays, int64_t, t->rays)
GETTER(fv_n_visible, int, t->fv_n_visible)
GETTER(last_free_block, int, t->last_free_block)
GETTER(last_free_excess, int, t->last_free_excess)
GETTER(n_total, int, t->n_total)
GETTER(hash, void *, t->hash)
GETTER(vba, void *, t->vba)
GETTER(vba_alloc_list, void *, t->vba_alloc_list)
GETTER(excess_list, void *, t->excess_list)
GETTER(visible_ids, void *, t->visible_ids)
GETTER(visible_type, void *, t->visible_type)
GETTER(minmax, void *, t->minmax)
GETTER(raycast, void *, t->raycast)
GETTER(icp_points, void *, t->icp_points)
GETTER(icp_normals, void *, t->icp_normals)
GETTER(depth, void *, t->depth)
GETTER(fv_visible_ids, void *, t->fv_visible_ids)
GETTER(fv_minmax, void *, t->fv_minmax)
GETTER(fv_raycast, void *, t->fv_raycast)
GETTER(fv_colour, void *, t->fv_colour)
