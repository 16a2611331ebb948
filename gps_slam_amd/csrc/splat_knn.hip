// Exact sub-quadratic distCUDA2 for large point sets (gps_knn_mean_dist2_grid): mean squared distance to the 3 nearest
// neighbours, the same numbers as the tiled brute force of splat_init.hip (gps_knn_mean_dist2) bit for bit.
//
// The reference (gsplat/rasterizer/simple_knn.cu:67-227) sorts the points along a Morton curve, boxes every 1024 of them and
// prunes boxes by their distance to the query.  On this path the points are samples of SURFACES seen by a depth camera (a first
// keyframe or a newly revealed room adds up to 0.25 x W x H of them: 76,800 at 640x480, 230,400 at 720p), so a uniform grid over
// their bounding box is the better fit for a wide machine: a counting sort by cell (histogram, scan, scatter -- no comparison
// sort, no cub), then eight lanes per point IN CELL ORDER that search the point's own cell and grow the searched cube ring by ring
// until the third-best distance found is no larger than the distance to the nearest face of the cube that still has grid behind
// it -- the exact termination test, so the result is the brute force's: the three smallest squared distances are a set
// property, both kernels keep them sorted ascending and add them in that order, and both compute a squared distance with
// knn_dist2() (one explicit fma chain).  Neighbouring lanes sit in the same or adjacent cells and walk the same candidate
// runs: their loads are broadcasts out of L2.
//
// Cell size: ~16 P cells over the bounding box (at most 2^21), i.e. a handful of points per OCCUPIED cell for surface samples;
// chosen on the device from the bounding box (no host round trip), written to the workspace header for the later launches.
#include <float.h>

#include "common.hpp"
#include "splat_knn.hpp"

namespace {

constexpr int KG_MAX_CELLS = 1 << 21;
constexpr int KG_MAX_DIM = 1024;
constexpr int KG_SCAN_THREADS = 256, KG_SCAN_PER = 16, KG_SCAN_BLOCK = KG_SCAN_THREADS * KG_SCAN_PER;   // cells per scan workgroup
constexpr int KG_MAX_SCAN_BLOCKS = KG_MAX_CELLS / KG_SCAN_BLOCK;                                        // 512

struct KnnGrid {          // workspace header (64 bytes)
    float minx, miny, minz, inv_h;
    float h;
    int gx, gy, gz, n_cells, n_scan_blocks;
    int pad[6];
};   // (written by knn_grid_kernel on every call: the workspace needs no initialisation)
static_assert(sizeof(KnnGrid) == 64, "header");

// workspace: header | counts[KG_MAX_CELLS + 1] (becomes the exclusive starts) | block_sums[KG_MAX_SCAN_BLOCKS] | cell_of[P] |
//            cursor-free scatter uses counts' copy: fill[KG_MAX_CELLS] | sorted float4[P]
struct KnnWs {
    KnnGrid* grid;
    int* starts;      // [n_cells + 1]
    int* block_sums;  // [KG_MAX_SCAN_BLOCKS]
    int* fill;        // [n_cells] running fill count of the scatter
    int* cell_of;     // [P]
    float4* sorted;   // [P] {x, y, z, original index}
};
__host__ __device__ inline int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }
static inline int64_t knn_ws_bytes(int P) {
    return 64 + align16(4 * ((int64_t)KG_MAX_CELLS + 1)) + align16(4 * (int64_t)KG_MAX_SCAN_BLOCKS) + align16(4 * (int64_t)KG_MAX_CELLS) +
           align16(4 * (int64_t)P) + 16 * (int64_t)P;
}
static inline KnnWs knn_ws(void* base, int P) {
    char* p = reinterpret_cast<char*>(base);
    KnnWs w;
    w.grid = reinterpret_cast<KnnGrid*>(p); p += 64;
    w.starts = reinterpret_cast<int*>(p); p += align16(4 * ((int64_t)KG_MAX_CELLS + 1));
    w.block_sums = reinterpret_cast<int*>(p); p += align16(4 * (int64_t)KG_MAX_SCAN_BLOCKS);
    w.fill = reinterpret_cast<int*>(p); p += align16(4 * (int64_t)KG_MAX_CELLS);
    w.cell_of = reinterpret_cast<int*>(p); p += align16(4 * (int64_t)P);
    w.sorted = reinterpret_cast<float4*>(p);
    return w;
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// (1) bounding box in two launches (no float atomics, no tickets): every workgroup reduces its stride of the points to six
// numbers, then one wave reduces the partials and picks the grid.
constexpr int KG_BBOX_BLOCKS = 64, KG_BBOX_THREADS = 256;
__global__ __launch_bounds__(KG_BBOX_THREADS) void knn_bbox_kernel(int P, const float* __restrict__ pts,
                                                                   float* __restrict__ partials /* [6 * KG_BBOX_BLOCKS] */) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; a++) {   // (finite coordinates only: a NaN / inf point must not decide the grid)
            const float v = pts[3 * i + a];
            if (fabsf(v) <= FLT_MAX) { lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
        }
    }
    __shared__ float red[KG_BBOX_THREADS / 64][6];
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = wave_min_f(lo[a]); hi[a] = wave_max_f(hi[a]); }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < KG_BBOX_THREADS / 64; w++)
#pragma unroll
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], red[w][a]); hi[a] = fmaxf(hi[a], red[w][3 + a]); }
#pragma unroll
        for (int a = 0; a < 3; a++) { partials[6 * blockIdx.x + a] = lo[a]; partials[6 * blockIdx.x + 3 + a] = hi[a]; }
    }
}
static_assert(KG_BBOX_BLOCKS == 64, "one partial per lane below");
__global__ __launch_bounds__(64) void knn_grid_kernel(int P, const float* __restrict__ partials, KnnGrid* __restrict__ grid) {
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = wave_min_f(partials[6 * threadIdx.x + a]);
        hi[a] = wave_max_f(partials[6 * threadIdx.x + 3 + a]);
    }
    if (threadIdx.x != 0) return;
    float ext[3], emax = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) { ext[a] = fmaxf(hi[a] - lo[a], 0.f); emax = fmaxf(emax, ext[a]); }
    if (!(emax > 0.f) || !(emax < FLT_MAX)) { emax = 1.f; ext[0] = ext[1] = ext[2] = 0.f; }   // a single point (or nothing finite): one cell
#pragma unroll
    for (int a = 0; a < 3; a++) if (!(lo[a] <= hi[a])) lo[a] = 0.f;   // (no finite coordinate on this axis)
#pragma unroll
    for (int a = 0; a < 3; a++) ext[a] = fmaxf(ext[a], 1e-3f * emax);   // planar sets: no axis thinner than 1/1000 of the longest
    int64_t target = 16 * (int64_t)P;
    target = target < 4096 ? 4096 : (target > KG_MAX_CELLS ? KG_MAX_CELLS : target);
    float h = cbrtf(ext[0] * ext[1] * ext[2] / (float)target);
    int g[3] = {1, 1, 1};
    for (int it = 0; it < 64; it++) {
        int64_t prod = 1;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float c = floorf(ext[a] / h) + 1.f;
            g[a] = c > (float)KG_MAX_DIM ? KG_MAX_DIM + 1 : (int)c;
            prod *= g[a];
        }
        if (prod <= KG_MAX_CELLS && g[0] <= KG_MAX_DIM && g[1] <= KG_MAX_DIM && g[2] <= KG_MAX_DIM) break;
        h *= 1.26f;   // 2^(1/3): half the cells
    }
    grid->minx = lo[0]; grid->miny = lo[1]; grid->minz = lo[2];
    grid->h = h; grid->inv_h = 1.0f / h;
    grid->gx = g[0]; grid->gy = g[1]; grid->gz = g[2];
    grid->n_cells = g[0] * g[1] * g[2];
    grid->n_scan_blocks = (grid->n_cells + KG_SCAN_BLOCK - 1) / KG_SCAN_BLOCK;
}

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_h, int g) {
    const float f = (v - lo) * inv_h;                    // (NaN / inf / out-of-range coordinates: clamped before the conversion)
    return f > 0.f ? (int)fminf(f, (float)(g - 1)) : 0;
}

// (2) zero the counters of the grid in use (+ the end marker); n_cells is only known on the device
__global__ __launch_bounds__(256) void knn_clear_kernel(const KnnGrid* __restrict__ grid, int* __restrict__ starts, int* __restrict__ fill) {
    const int n = grid->n_cells;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) { starts[i] = 0; if (i < n) fill[i] = 0; }
}

// (3) histogram
__global__ __launch_bounds__(256) void knn_count_kernel(int P, const float* __restrict__ pts, const KnnGrid* __restrict__ grid,
                                                        int* __restrict__ counts, int* __restrict__ cell_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const KnnGrid g = *grid;
    const int cx = cell_coord(pts[3 * i], g.minx, g.inv_h, g.gx), cy = cell_coord(pts[3 * i + 1], g.miny, g.inv_h, g.gy),
              cz = cell_coord(pts[3 * i + 2], g.minz, g.inv_h, g.gz);
    const int c = cx + g.gx * (cy + g.gy * cz);
    cell_of[i] = c;
    atomicAdd(&counts[c], 1);
}

// (4) exclusive scan of the counts in two launches over 4096-cell blocks: block sums, then every block adds the sums in front of it
__global__ __launch_bounds__(KG_SCAN_THREADS) void knn_scan_sums_kernel(const KnnGrid* __restrict__ grid, const int* __restrict__ counts,
                                                                        int* __restrict__ block_sums) {
    if ((int)blockIdx.x >= grid->n_scan_blocks) return;
    const int n = grid->n_cells, first = blockIdx.x * KG_SCAN_BLOCK + threadIdx.x * KG_SCAN_PER;
    int s = 0;
#pragma unroll
    for (int k = 0; k < KG_SCAN_PER; k++) s += first + k < n ? counts[first + k] : 0;
    s = wave_sum_i(s);
    __shared__ int ws[KG_SCAN_THREADS / 64];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(KG_SCAN_THREADS) void knn_scan_apply_kernel(const KnnGrid* __restrict__ grid, int* __restrict__ counts,
                                                                         const int* __restrict__ block_sums, int P) {
    const int nb = grid->n_scan_blocks;
    if ((int)blockIdx.x >= nb) return;
    const int n = grid->n_cells, first = blockIdx.x * KG_SCAN_BLOCK + threadIdx.x * KG_SCAN_PER;
    __shared__ int ws[KG_SCAN_THREADS / 64];
    __shared__ int base_s;
    // sums of the blocks in front of this one (<= 511 values: two per thread)
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += KG_SCAN_THREADS) before += block_sums[b];
    before = wave_sum_i(before);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) base_s = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    int v[KG_SCAN_PER], s = 0;
#pragma unroll
    for (int k = 0; k < KG_SCAN_PER; k++) { v[k] = first + k < n ? counts[first + k] : 0; s += v[k]; }
    const int incl = wave_incl_scan_i(s);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = base_s + incl - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += ws[w];
#pragma unroll
    for (int k = 0; k < KG_SCAN_PER; k++) {
        if (first + k < n) counts[first + k] = run;
        run += v[k];
    }
    if (blockIdx.x == (unsigned)(nb - 1) && threadIdx.x == 0) counts[n] = P;
}

// (5) scatter into cell order
__global__ __launch_bounds__(256) void knn_scatter_kernel(int P, const float* __restrict__ pts, const int* __restrict__ cell_of,
                                                          const int* __restrict__ starts, int* __restrict__ fill, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int c = cell_of[i];
    const int slot = starts[c] + atomicAdd(&fill[c], 1);
    sorted[slot] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float(i));
}

// (6) KQ_LANES lanes per point, points in cell order.  The lanes of a point walk the same cells and split every run of
// candidates between them (candidate lo + lane, + KQ_LANES, ...): KQ_LANES independent loads in flight per point instead of one
// dependent chain -- one lane per point made the search latency-bound (94 us for 5 k points: ~100 candidates x one L2 round
// trip each).  Every lane keeps the three smallest distances of ITS candidates; the termination test and the result use their
// merge (three xor-shuffles; the private triples are never overwritten, so no candidate is counted twice).
constexpr int KQ_LANES = 8;
__device__ __forceinline__ void merge3(float& b0, float& b1, float& b2) {
#pragma unroll
    for (int o = 1; o < KQ_LANES; o <<= 1) {
        const float o0 = __shfl_xor(b0, o, 64), o1 = __shfl_xor(b1, o, 64), o2 = __shfl_xor(b2, o, 64);
        gps::keep3(o0, b0, b1, b2); gps::keep3(o1, b0, b1, b2); gps::keep3(o2, b0, b1, b2);
    }
}
__global__ __launch_bounds__(256) void knn_query_kernel(int P, const KnnGrid* __restrict__ grid, const int* __restrict__ starts,
                                                        const float4* __restrict__ sorted, float* __restrict__ out) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / KQ_LANES, sub = threadIdx.x & (KQ_LANES - 1);
    if (t >= P) return;   // (whole sub-groups leave together: the shuffles below stay inside a sub-group)
    const KnnGrid g = *grid;
    const float4 q = sorted[t];
    const int self = __float_as_int(q.w);
    const int cx = cell_coord(q.x, g.minx, g.inv_h, g.gx), cy = cell_coord(q.y, g.miny, g.inv_h, g.gy),
              cz = cell_coord(q.z, g.minz, g.inv_h, g.gz);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;   // of this lane's candidates
    float m0 = FLT_MAX, m1 = FLT_MAX, m2 = FLT_MAX;   // merged over the point's lanes
    if (!(fabsf(q.x) <= FLT_MAX) || !(fabsf(q.y) <= FLT_MAX) || !(fabsf(q.z) <= FLT_MAX)) {
        // a NaN / inf query is at no finite distance from anything: the brute force leaves its three slots at FLT_MAX (every
        // comparison fails) -- the same expression here, without walking the whole grid for it
        if (sub == 0) out[self] = (m0 + m1 + m2) / 3.0f;
        return;
    }
    auto scan_run = [&](int lo, int hi) {
        for (int k = lo + sub; k < hi; k += KQ_LANES) {
            const float4 c = sorted[k];
            if (__float_as_int(c.w) == self) continue;
            gps::keep3(gps::knn_dist2(c.x - q.x, c.y - q.y, c.z - q.z), b0, b1, b2);
        }
    };
    // a point's computed cell can differ from its exact one by rounding at a face: the faces of the searched cube are trusted only
    // up to a margin of 1 % of a cell (the coordinate error is < 1e-4 cells at <= 1024 cells per axis)
    // -- plus, for point sets far from the origin relative to their extent, the rounding of the coordinates themselves: the face
    // distances are taken in the grid's own frame (q - min, one rounding of |q|'s magnitude) and the margin grows by a few ulps of
    // the largest coordinate magnitude in the box
    const float reach = fmaxf(fmaxf(fmaxf(fabsf(g.minx), fabsf(g.minx + (float)g.gx * g.h)), fmaxf(fabsf(g.miny), fabsf(g.miny + (float)g.gy * g.h))),
                              fmaxf(fabsf(g.minz), fabsf(g.minz + (float)g.gz * g.h)));
    const float margin = 0.01f * g.h + 4.0f * 1.1920929e-7f * reach;
    const float sx = q.x - g.minx, sy = q.y - g.miny, sz = q.z - g.minz;
    for (int r = 0;; r++) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.gx - 1), y0 = max(cy - r, 0), y1 = min(cy + r, g.gy - 1),
                  z0 = max(cz - r, 0), z1 = min(cz + r, g.gz - 1);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                // the shell of ring r: whole x-runs on its z / y faces, the two end cells elsewhere
                const bool face = (z == cz - r) | (z == cz + r) | (y == cy - r) | (y == cy + r);
                const int row = g.gx * (y + g.gy * z);
                if (face) {
                    scan_run(starts[row + x0], starts[row + x1 + 1]);   // cells x0..x1 of a row are consecutive: one run of sorted points
                } else {
                    if (cx - r >= 0) scan_run(starts[row + cx - r], starts[row + cx - r + 1]);
                    if (cx + r < g.gx) scan_run(starts[row + cx + r], starts[row + cx + r + 1]);
                }
            }
        m0 = b0; m1 = b1; m2 = b2;
        merge3(m0, m1, m2);
        // distance from the query to the nearest face of the searched cube that has grid behind it
        float dmin = FLT_MAX;
        bool open = false;
        if (cx - r > 0) { open = true; dmin = fminf(dmin, sx - (float)(cx - r) * g.h); }
        if (cx + r < g.gx - 1) { open = true; dmin = fminf(dmin, (float)(cx + r + 1) * g.h - sx); }
        if (cy - r > 0) { open = true; dmin = fminf(dmin, sy - (float)(cy - r) * g.h); }
        if (cy + r < g.gy - 1) { open = true; dmin = fminf(dmin, (float)(cy + r + 1) * g.h - sy); }
        if (cz - r > 0) { open = true; dmin = fminf(dmin, sz - (float)(cz - r) * g.h); }
        if (cz + r < g.gz - 1) { open = true; dmin = fminf(dmin, (float)(cz + r + 1) * g.h - sz); }
        if (!open) break;                       // the cube covers the grid
        dmin -= margin;
        if (dmin > 0.f && m2 <= dmin * dmin) break;   // nothing outside the cube can be closer than the third best
    }
    if (sub == 0) out[self] = (m0 + m1 + m2) / 3.0f;
}

}  // namespace

extern "C" {

int64_t gps_knn_grid_workspace_bytes(int P) { return P < 0 ? (int64_t)GPS_ERR_ARG : knn_ws_bytes(P) + 4 * 6 * KG_BBOX_BLOCKS; }

int gps_knn_mean_dist2_grid(int P, const float* points, float* mean_dist2, void* workspace, int64_t workspace_bytes, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(P >= 0);
    if (P == 0) return GPS_OK;
    GPS_REQUIRE(points && mean_dist2 && workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
    GPS_REQUIRE(workspace_bytes >= gps_knn_grid_workspace_bytes(P));
    hipStream_t st = (hipStream_t)stream;
    const KnnWs w = knn_ws(workspace, P);
    float* partials = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + knn_ws_bytes(P));
    knn_bbox_kernel<<<KG_BBOX_BLOCKS, KG_BBOX_THREADS, 0, st>>>(P, points, partials);
    knn_grid_kernel<<<1, 64, 0, st>>>(P, partials, w.grid);
    knn_clear_kernel<<<1024, 256, 0, st>>>(w.grid, w.starts, w.fill);
    knn_count_kernel<<<gps_div_up(P, 256), 256, 0, st>>>(P, points, w.grid, w.starts, w.cell_of);
    knn_scan_sums_kernel<<<KG_MAX_SCAN_BLOCKS, KG_SCAN_THREADS, 0, st>>>(w.grid, w.starts, w.block_sums);
    knn_scan_apply_kernel<<<KG_MAX_SCAN_BLOCKS, KG_SCAN_THREADS, 0, st>>>(w.grid, w.starts, w.block_sums, P);
    knn_scatter_kernel<<<gps_div_up(P, 256), 256, 0, st>>>(P, points, w.cell_of, w.starts, w.fill, w.sorted);
    knn_query_kernel<<<gps_div_up((int64_t)P * KQ_LANES, 256), 256, 0, st>>>(P, w.grid, w.starts, w.sorted, mean_dist2);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"
