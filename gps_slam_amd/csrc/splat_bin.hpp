// Shared between splat_bin.hip (tile binning) and splat_fused.hip / splat_step.hip: the fused preprocessing kernel of the
// model path can produce the binning's first pass (tiles / 32-pixel groups per Gaussian + per-256 block sums) itself, so an
// optimise iteration does not launch count_kernel over the arrays it has just written.
#pragma once
#include "common.hpp"

namespace gps {

constexpr int BIN_BLOCK = 256;  // Gaussians per count / expand workgroup == threads of the preprocessing kernel

struct TileBox { uint32_t x0, y0, x1, y1; };

// isect_tiles_no_depth.cu:68-80: bbox in tile units; float->uint conversion saturates at 0.
__device__ __forceinline__ TileBox tile_bbox(float mx, float my, int radius_i, int tile_size, int tw, int th) {
    float radius = (float)radius_i;
    float ts = (float)tile_size;
    float tr = radius / ts, tx = mx / ts, ty = my / ts;
    TileBox b;
    float fx0 = floorf(tx - tr), fy0 = floorf(ty - tr), fx1 = ceilf(tx + tr), fy1 = ceilf(ty + tr);
    b.x0 = (uint32_t)fminf(fmaxf(fx0, 0.f), (float)tw);
    b.y0 = (uint32_t)fminf(fmaxf(fy0, 0.f), (float)th);
    b.x1 = (uint32_t)fminf(fmaxf(fx1, 0.f), (float)tw);
    b.y1 = (uint32_t)fminf(fmaxf(fy1, 0.f), (float)th);
    return b;
}

// tiles and 32-pixel groups of one Gaussian (isect_tiles_no_depth.cu:82-90)
__device__ __forceinline__ void tile_group_count(float mx, float my, int r, int tile_size, int tw, int th, int& tiles, int& groups) {
    TileBox b = tile_bbox(mx, my, r, tile_size, tw, th);
    tiles = (int)((b.y1 - b.y0) * (b.x1 - b.x0));
    float rf = (float)r;
    groups = (int)((4 * rf * rf + 32 - 1) / 32);  // fp32 expression of isect_tiles_no_depth.cu:87
}

// ---- superblock binning (splat_bin_sb.hip): the fused model path's tile binning in two launches behind the preprocessing
// kernel.  A superblock = SB_BLOCKS consecutive preprocessing workgroups (256 Gaussians each), at most SB_MAX superblocks.
//   preprocessing kernel : per workgroup an LDS histogram of its (Gaussian, tile) pairs over the <= SB_MAX_TILES tile ids, added
//                          to the count table C[tile][superblock]; Gaussians per backward list key (class, image band) into
//                          cls_count[key][superblock]
//   scan kernel          : per tile the exclusive prefix over superblocks P[tile][sb] + the tile total; C is zeroed again
//   scatter kernel       : one workgroup per superblock enumerates its pairs in Gaussian order and writes each Gaussian id to
//                          tile_start[tile] + P[tile][sb] + (its stable rank inside the superblock): the order a stable sort
//                          by tile id gives (isect_tiles_no_depth.cu:313-327), without ever materialising a key / value
//                          array; tile_offsets = exclusive scan of the tile totals; the backward's class lists ride along,
//                          each ordered by (image band, Gaussian id): a counting sort over the BWD_KEYS keys with the same
//                          three steps, so that a contiguous piece of a class list gathers from one band of the image.
// Invariant: C and cls_count are ZERO between launches (zeroed by gps_isect_workspace_init, then by the scan kernel).
constexpr int SB_MAX = 512;          // superblocks (row length of the tables)
constexpr int SB_MAX_TILES = 4096;   // tile ids the LDS histograms cover; more tiles -> the sorted-key path of splat_bin.hip
constexpr int BWD_CLASSES = GPS_BWD_CLASSES;
constexpr int BWD_BANDS = 16;        // horizontal image bands a class list is ordered by (first tile row of the Gaussian's box)
constexpr int BWD_KEYS = BWD_CLASSES * BWD_BANDS;
struct SbTables {
    uint32_t* C;           // [SB_MAX_TILES][SB_MAX] counts (zero between launches)
    uint32_t* P;           // [SB_MAX_TILES][SB_MAX] exclusive prefixes over superblocks
    uint32_t* tile_total;  // [SB_MAX_TILES]
    int32_t* cls_count;    // [BWD_KEYS][SB_MAX] Gaussians per (class, band) and superblock (zero between launches)
    int32_t* cls_prefix;   // [BWD_KEYS][SB_MAX] exclusive prefixes over superblocks, then a row of the BWD_KEYS totals
    int sb_shift;          // log2(preprocessing workgroups per superblock)
    int32_t* tile_order;   // [SB_MAX_TILES] the tiles by descending list length (sb_scatter's extra workgroup): the forward rasterizer's launch order
};
// splat_raster_bwd.hip: gps_set_frame_chain_reserve(on) -- the host runs tracking / fusion on another stream beside the map update
bool map_runs_beside_frame_chain();
int frame_chain_reserve_bits();   // the value last given to gps_set_frame_chain_reserve (bit 1: experiment, see splat_raster_bwd.hip)
// splat_bin_sb.hip: can the superblock binning take N Gaussians on a tile_width x tile_height grid on THIS device (tile count,
// packed box fields, superblock count, the scatter kernel's dynamic LDS incl. its > 64 KB opt-in)?  False -> sorted-key binning.
bool sb_supported(int N, int tile_width, int tile_height);
int sb_tables_clear(const struct SbTables& t, gps_stream stream);
__host__ __device__ inline int sb_shift_for(int N) {  // smallest power of two of 256-Gaussian blocks with <= SB_MAX superblocks
    int s = 0;
    while ((((int64_t)N + 255) / 256 + ((int64_t)1 << s) - 1) >> s > SB_MAX) s++;
    return s;
}
// backward class of a radius: the smallest k with 4 << k >= r, the last class takes everything wider
__host__ __device__ inline int bwd_class(int r) { return r <= 4 ? 0 : r <= 8 ? 1 : r <= 16 ? 2 : r <= 32 ? 3 : 4; }
// list key of a visible Gaussian: class-major, then the band of its box's first tile row y0 (0 when the box holds no tile)
__host__ __device__ inline int bwd_key(int r, int n_tiles, unsigned y0, int th) {
    const int band = n_tiles > 0 ? (int)(y0 * (unsigned)BWD_BANDS / (unsigned)(th > 0 ? th : 1)) : 0;
    return bwd_class(r) * BWD_BANDS + (band < BWD_BANDS ? band : BWD_BANDS - 1);
}

// where the first pass's results live (pointers into the caller's binning workspace); tiles_per_gauss == nullptr: off
struct BinCountOut {
    int32_t *tiles_per_gauss, *groups_per_gauss, *blk_tiles, *blk_groups, *blk_vis;
    int tile_size, tw, th;
    SbTables sb;   // sb.C == nullptr: no superblock histogram (the sorted-key path follows)
};

// one value triple per thread -> the three per-workgroup sums (all BIN_BLOCK threads must call)
__device__ __forceinline__ void bin_block_sums(const BinCountOut& o, int t, int g, int vis) {
    __shared__ int red[3][BIN_BLOCK / 64];
    int ts = wave_sum_i(t), gs = wave_sum_i(g), vs = wave_sum_i(vis);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ts; red[1][threadIdx.x >> 6] = gs; red[2][threadIdx.x >> 6] = vs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0, c = 0;
        for (int w = 0; w < BIN_BLOCK / 64; w++) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
        o.blk_tiles[blockIdx.x] = a; o.blk_groups[blockIdx.x] = b; o.blk_vis[blockIdx.x] = c;
    }
}

// gradient buffers of the backward rasterizer the preprocessing kernel may zero on the way (v_colors == nullptr: off):
// the backward then starts from pre-zeroed buffers and needs no zero-fill launch
struct ZeroGrads { float *v_means2d, *v_conics, *v_colors, *v_opacities; };

// compose + L1 (gps_compose_l1) as the epilogue of the forward rasterizer (base_color == nullptr: off): a tile's finished
// pixels go straight from registers into rgb / loss / the two image gradients -- no launch, no re-read of the render
struct FwdCompose {
    const float* base_color;  // [P,3]
    const float* gt_rgb;      // [P,3]
    float* rgb;               // [P,3]
    float* loss;              // scalar accumulator
    float* v_render_colors;   // [P,4]
    float* v_render_alphas;   // [P]
    float inv_count;          // 1 / (3 P)
    float* pix2;              // [P,2] {v_render_alpha, ref_depth + delta_depth}: what the strip backward gathers (may be NULL)
};
// splat_raster.hip
int raster_ges_fwd_rec_launch(int N, const float* records, const float* ref_depth_map, int width, int height,
                              const int32_t* tile_offsets, const int32_t* flatten_ids, const int64_t* counts, float delta_depth,
                              float* render_colors, float* render_alphas, const FwdCompose* compose, gps_stream stream, const int32_t* tile_order = nullptr);
// gps_raster_ges_bwd_gs with zero_mode: 0 zero-fill here, 1 accumulate onto the buffers, 2 the buffers are already zero
int raster_ges_bwd_gs_launch(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                             const int32_t* radii, const float* ref_depth_map, int width, int height,
                             const int32_t* group_gs_ids, const int32_t* group_starts, const int64_t* counts, float delta_depth,
                             const float* v_render_colors, const float* v_render_alphas, float* v_means2d, float* v_conics,
                             float* v_colors, float* v_opacities, int zero_mode, gps_stream stream);

// splat_bin.hip: the count targets inside `workspace` for (N, isect_capacity); GPS_OK or an error.  superblock: also the
// tables of the superblock binning (needs tile_width * tile_height <= SB_MAX_TILES and an initialised workspace)
int isect_count_targets(int N, int64_t isect_capacity, int32_t* tiles_per_gauss, int tile_size, int tile_width, int tile_height,
                        void* workspace, int64_t workspace_bytes, bool superblock, BinCountOut* out);
// splat_bin_sb.hip: bytes of the superblock tables; scan + scatter behind a preprocessing kernel that filled `cnt.sb`.
// cls_ids / cls_counts may be NULL (render only).  counts = {n_isects, 0, overflow (sticky), n_visible}.
size_t sb_tables_bytes();
// the persistent tables of a workspace whatever N it is used with next (clearing them after a discarded prefetch)
int isect_workspace_tables(void* workspace, int64_t workspace_bytes, struct SbTables* t);
void sb_tables_carve(char* base, SbTables* t);
int isect_tiles_superblock(int N, const float* means2d, const int32_t* radii, const BinCountOut& cnt, int64_t isect_capacity,
                           const int32_t* tiles_per_gauss, int32_t* flatten_ids, int32_t* tile_offsets, int64_t* counts,
                           int32_t* cls_ids, int32_t* cls_counts, int64_t cls_stride, gps_stream stream);
// splat_raster_bwd.hip
int raster_ges_bwd_strips_launch(int N, const float* records, const int32_t* radii, const int32_t* cls_ids,
                                 const int32_t* cls_counts, int cls_stride, const float* v_render_colors, const float* pix2,
                                 int width, int height, float* v_rows, gps_stream stream);
// splat_bin.hip: gps_isect_tiles_no_depth whose first pass has already been written to isect_count_targets()'s pointers
int isect_tiles_no_depth_counted(int N, const float* means2d, const int32_t* radii, int tile_size, int tile_width,
                                 int tile_height, int64_t isect_capacity, int64_t group_capacity, int32_t* tiles_per_gauss,
                                 int32_t* flatten_ids, int32_t* group_gs_ids, int32_t* group_starts, int32_t* tile_offsets,
                                 int64_t* counts, void* workspace, int64_t workspace_bytes, gps_stream stream);
// splat_fused.hip: gps_gauss_preprocess_fwd that also writes the binning's first pass
int preprocess_fwd_launch(int N, int K, int sh_degree, const float* means, const float* log_scales, const float* quats,
                          const float* opac_logit, const float* sh_dc, const float* sh_rest, const float* viewmat,
                          const float* Kmat, const float* cam_pos, int width, int height, float eps2d, float near_plane,
                          float far_plane, float radius_clip, int max_gs_radii, int32_t* radii, float* means2d, float* depths,
                          float* conics, float* colors, float* opacities, float* records, const BinCountOut* count,
                          const ZeroGrads* zero, gps_stream stream);

}  // namespace gps
