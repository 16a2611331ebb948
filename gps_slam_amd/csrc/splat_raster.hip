// `ges` rasterizer: order-independent depth-cut weighted splat (forward) and the
// Gaussian-parallel backward over each Gaussian's 2r x 2r pixel box.
//
// gps_raster_ges_fwd    <- gsplat::rasterize_to_pixels_fwd_ges_tensor
//                          (gsplat/rasterizer/rasterize_to_pixels_fwd_ges.cu:18-221)
// gps_raster_ges_bwd_gs <- gsplat::rasterize_to_pixels_bwd_ges_gs_parallel_tensor
//                          (gsplat/rasterizer/rasterize_to_pixels_bwd_ges_new_parallel.cu:18-201)
//
// Forward: one 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns pixel
// rows 4w..4w+3.  The tile's sorted Gaussian list is staged through LDS in
// batches of 256 complete records {xy, conic, opacity, depth, rgb} (40 B) so the
// per-pixel loop reads only LDS (broadcast reads, conflict free) -- the
// reference re-reads the depth/colour channels from global memory for every
// (pixel, Gaussian) pair (:165-166).  No MFMA: this is not a contraction.
//
// Backward: the reference gives each 32-lane warp 32 consecutive 32-pixel groups
// and issues 10 atomics per group.  Here a wave64 takes 32 consecutive groups as
// two contiguous runs of 16 (one per half-wave), keeps the 10 partial sums in
// registers while consecutive groups belong to the same Gaussian, and only
// reduces (within the 32-lane half) + atomically adds when the Gaussian changes:
// ~N_visible x 10 atomics instead of n_groups x 10.  Pixel/box semantics
// (int() truncation, +1 offsets, i > y_max guard) are the reference's.
#include "common.hpp"
#include "launch_timing.hpp"
#include "splat_bin.hpp"

namespace {

constexpr int TILE_THREADS = 256;


template <int TILE>
__global__ __launch_bounds__(TILE_THREADS) void raster_ges_fwd_kernel(
    const float2* __restrict__ means2d, const float* __restrict__ conics, const float4* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ ref_depth, int W, int H, int tw, int th,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    const int64_t* __restrict__ counts, float delta_depth, float4* __restrict__ render_colors,
    float* __restrict__ render_alphas, int32_t* __restrict__ last_ids) {
    static_assert(TILE == 16, "one thread per pixel of a 16x16 tile");
    __shared__ float4 rec0[TILE_THREADS];
    __shared__ float4 rec1[TILE_THREADS];
    __shared__ float2 rec2[TILE_THREADS];

    const int tile_id = blockIdx.x;
    const int ty = tile_id / tw, tx = tile_id - ty * tw;
    const int tid = threadIdx.x;
    const int i = ty * TILE + (tid >> 4), j = tx * TILE + (tid & 15);
    const bool inside = (i < H) && (j < W);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const int pix = i * W + j;

    const int n_isects = (int)counts[0];
    const int range_start = tile_offsets[tile_id];
    const int range_end = (tile_id == tw * th - 1) ? n_isects : tile_offsets[tile_id + 1];

    // pixels outside the image never pass the depth test
    const float cut = inside ? ref_depth[pix] + delta_depth : -3.0e38f;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, wsum = 0.f;
    int cur_idx = 0;

    for (int batch_start = range_start; batch_start < range_end; batch_start += TILE_THREADS) {
        __syncthreads();  // previous batch fully consumed
        const int idx = batch_start + tid;
        if (idx < range_end) {
            const int g = flatten_ids[idx];
            const float2 xy = means2d[g];
            const float4 c = colors[g];
            const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
            rec0[tid] = make_float4(xy.x, xy.y, ca, cb);
            rec1[tid] = make_float4(cc, opacities[g], c.w, c.x);
            rec2[tid] = make_float2(c.y, c.z);
        }
        __syncthreads();
        const int batch_size = min(TILE_THREADS, range_end - batch_start);
        for (int t = 0; t < batch_size; ++t) {
            const float4 a = rec0[t];
            const float4 b = rec1[t];
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = 0.5f * (a.z * dx * dx + b.x * dy * dy) + a.w * dx * dy;
            const float alpha = fminf(0.999f, b.y * __expf(-sigma));
            const bool hit = !(b.z > cut) && !(sigma < 0.f) && !(alpha < 1.f / 255.f);
            if (hit) {
                const float2 gb = rec2[t];
                o0 += b.w * alpha; o1 += gb.x * alpha; o2 += gb.y * alpha; o3 += b.z * alpha;
                wsum += alpha;
                cur_idx = batch_start + t;
            }
        }
    }
    if (inside) {
        render_colors[pix] = make_float4(o0, o1, o2, o3);
        render_alphas[pix] = wsum;
        if (last_ids) last_ids[pix] = cur_idx;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Forward from the packed per-Gaussian records of gps_gauss_preprocess_fwd (the fused model path), packed math.
// record = 3 x float4: a = {mx, my, ca, cb}  b = {cc, opac, depth, r}  c = {g, b, xbounds, ybounds}, see pack_record() in
// splat_math.hpp.  The pair evaluation is VALU bound (PMC: ~23 wave
// instructions per Gaussian x 64-pixel strip), so the lever is instructions per pixel:
//   * each lane owns TWO horizontally adjacent pixels and evaluates them with v_pk_{add,mul,fma}_f32 -- dy, c*dy^2 and
//     b*dy are shared by the pair, everything in dx and the five accumulators are 2-wide: ~25 VALU per Gaussian for
//     128 pixels instead of ~23 per 64;
//   * the staging step rewrites the record for the inner loop once per tile: conic pre-multiplied by log2(e) (and 1/2),
//     opacity as -log2(opacity) folded into the exponent, so alpha = exp2(-(sigma' - log2 o)) is one v_exp_f32;
//   * a 16x16 tile is covered by two waves (16x8 pixels each); to keep four waves per workgroup in flight the batch's
//     Gaussian list is split in two halves, waves {0,1} and {2,3} each take one, and the partial sums are added
//     through LDS at the end in a fixed order (deterministic; the sum is order independent in exact arithmetic).
typedef float v2f __attribute__((ext_vector_type(2)));

// one pixel of gps_compose_l1 (splat_optim.hip: compose_l1_kernel) on a render that is still in registers; returns the
// pixel's |gt - rgb| sum
__device__ __forceinline__ float compose_l1_pixel(const gps::FwdCompose& fc, int p, float4 rc, float w, float cut) {
    const float den = w + 1.0f;  // base colour weight is always 1 (raw_gs_model.cpp:321-323)
    const float n0 = rc.x + fc.base_color[3 * p], n1 = rc.y + fc.base_color[3 * p + 1], n2 = rc.z + fc.base_color[3 * p + 2];
    const float c0 = n0 / den, c1 = n1 / den, c2 = n2 / den;
    fc.rgb[3 * p] = c0; fc.rgb[3 * p + 1] = c1; fc.rgb[3 * p + 2] = c2;
    const float d0 = fc.gt_rgb[3 * p] - c0, d1 = fc.gt_rgb[3 * p + 1] - c1, d2 = fc.gt_rgb[3 * p + 2] - c2;
    const float ic = fc.inv_count;
    const float g0 = d0 > 0.f ? -ic : (d0 < 0.f ? ic : 0.f);
    const float g1 = d1 > 0.f ? -ic : (d1 < 0.f ? ic : 0.f);
    const float g2 = d2 > 0.f ? -ic : (d2 < 0.f ? ic : 0.f);
    reinterpret_cast<float4*>(fc.v_render_colors)[p] = make_float4(g0 / den, g1 / den, g2 / den, 0.f);
    const float dd = den * den;
    const float va = -(g0 * n0) / dd - (g1 * n1) / dd - (g2 * n2) / dd;
    fc.v_render_alphas[p] = va;
    // what the strip backward gathers per pixel: {d loss / d weight sum, the depth cut ref_depth + delta_depth}
    if (fc.pix2) reinterpret_cast<float2*>(fc.pix2)[p] = make_float2(va, cut);
    return fabsf(d0) + fabsf(d1) + fabsf(d2);
}

#ifndef GPS_FWD_LIST_SPLIT
#define GPS_FWD_LIST_SPLIT 4
#endif
GPS_TUNABLE_REPORT(GPS_FWD_LIST_SPLIT, 4);
#ifdef GPS_FWD_STAMPS
GPS_SWITCH_REPORT(GPS_FWD_STAMPS);
// probe builds only (tools/probe/fwd_stamps.py): 100 MHz timestamps of workgroup phases, 8 per tile
__device__ unsigned long long gps_fwd_stamps_buf[4096 * 8];
extern "C" GPS_API void* gps_fwd_stamps() { void* p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(gps_fwd_stamps_buf)); return p; }
#define FWD_STAMP(k) do { if (threadIdx.x == 0) gps_fwd_stamps_buf[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define FWD_STAMP(k) do { } while (0)
#endif
// A 16 x 16 tile = two 16 x 8 pixel halves (one wave each, 2 px per lane) x FWD_SPLIT list parts: 2 * FWD_SPLIT waves per tile.
// History of the inner loop (bench scene, 1,200 tiles, ~500 k list entries):
//   round 1   every entry evaluated for every pixel                                                   81.6 us
//   round 2   wave-uniform skip in the loop (test e > 8 for all 128 pixels, ballot, branch)             55 us   (list in 4 parts)
//   round 3   cull at staging time, then blend survivors only (below)                                   see LABBOOK.md section 4
// The round-2 loop was bound by its own dependent chain per entry: LDS read -> 10 VALU -> ballot -> branch -> exp -> blend, ~400
// cycles per entry for a lone wave (tools/probe/fwd_stamps.py), with a third of the issue slots used.  Now the staging thread of an
// entry tests the entry's pixel bounds (pack_record: a conservative box around {alpha >= 1/255}; outside it the entry adds exact
// zeros) against the two halves, and the survivors of each half are compacted, in list order, into an index list in LDS.  A
// wave takes an equal share of ITS half's survivors, holds their record addresses in one or two registers (lane j = j-th
// survivor) and walks them with v_readlane: no test, no ballot, no data-dependent branch in the loop, and no LDS reads for the
// ~45 % of a tile's entries that do not reach a given half.  Sums are added in list order inside a part and the parts in order:
// deterministic; a culled entry would have added +0.
constexpr int FWD_SPLIT = GPS_FWD_LIST_SPLIT;
constexpr int FWD_THREADS = 128 * FWD_SPLIT;
#ifndef GPS_FWD_BATCH
#define GPS_FWD_BATCH 512
#endif
GPS_TUNABLE_REPORT(GPS_FWD_BATCH, 512);
constexpr int FWD_BATCH = GPS_FWD_BATCH;         // list entries staged per batch (a thread stages FWD_TRIPS of them)
constexpr int FWD_TRIPS = (FWD_BATCH + FWD_THREADS - 1) / FWD_THREADS;
constexpr int FWD_SEGS = FWD_BATCH / 64;         // 64-entry segments of a batch (one ballot each)
constexpr int FWD_VECS = (FWD_BATCH / FWD_SPLIT + 63) / 64;   // registers that hold a part's survivor addresses
static_assert(FWD_BATCH % 64 == 0 && FWD_BATCH * 3 < 65536, "segments are whole waves; record slots are 16-bit");
__global__ __launch_bounds__(FWD_THREADS) void raster_ges_fwd_pk_kernel(
    const float4* __restrict__ recs, const float* __restrict__ ref_depth, int W, int H, int tw, int th,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    const int64_t* __restrict__ counts, float delta_depth, float4* __restrict__ render_colors,
    float* __restrict__ render_alphas, gps::FwdCompose fc, const int32_t* __restrict__ tile_order, gps::LaunchStamp stamp) {
    gps::StampScope timed(stamp);
    // 48-byte records {mx, my, 0.5*ca*log2e, cb*log2e | 0.5*cc*log2e, -log2(opac), depth, r | g, b, -, -}; after the last batch the
    // same memory carries the parts' partial sums
    constexpr int PART_FLOATS = (FWD_SPLIT - 1) * 128 * 10;
    constexpr int REC_FLOATS = FWD_BATCH * 12;
    __shared__ float4 lds_rec[(REC_FLOATS > PART_FLOATS ? REC_FLOATS : PART_FLOATS) / 4];
    __shared__ uint16_t sidx[2][FWD_BATCH];    // per pixel half: the batch's surviving entries (record byte offsets / 16), list order
    __shared__ int scnt[2][FWD_SEGS];          // survivors per staging wave
    FWD_STAMP(0);
    const int tile_id = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;   // (longest lists first when the binning provides the order)
    const int ty = tile_id / tw, tx = tile_id - ty * tw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: loop bounds below)
    const int list_part = wave >> 1, pix_half = wave & 1;
    const int row = ty * 16 + pix_half * 8 + (lane >> 3), col = tx * 16 + 2 * (lane & 7);
    const bool in0 = (row < H) && (col < W), in1 = (row < H) && (col + 1 < W);
    const v2f px = {(float)col + 0.5f, (float)col + 1.5f};
    const float py = (float)row + 0.5f;
    const float cut0 = in0 ? ref_depth[row * W + col] + delta_depth : -3.0e38f;
    const float cut1 = in1 ? ref_depth[row * W + col + 1] + delta_depth : -3.0e38f;
    v2f o0 = {0.f, 0.f}, o1 = o0, o2 = o0, o3 = o0, ws = o0;
    const int n_isects = (int)counts[0];
    const int range_start = tile_offsets[tile_id];
    const int range_end = (tile_id == tw * th - 1) ? n_isects : tile_offsets[tile_id + 1];
    constexpr float LOG2E = 1.4426950408889634f;
    const unsigned long long lt = lanemask_lt();

    for (int batch_start = range_start; batch_start < range_end; batch_start += FWD_BATCH) {
        __syncthreads();   // the previous batch's records and lists are consumed
        bool h0[FWD_TRIPS], h1[FWD_TRIPS];
        int r0[FWD_TRIPS], r1[FWD_TRIPS];   // rank among the segment's survivors
#pragma unroll
        for (int u = 0; u < FWD_TRIPS; u++) {
            const int k = u * FWD_THREADS + tid;   // slot in the batch; segment k >> 6 = u * (FWD_THREADS / 64) + wave
            const int idx = batch_start + k;
            h0[u] = false; h1[u] = false;
            if (k < FWD_BATCH && idx < range_end) {
                const size_t g = (size_t)flatten_ids[idx];
                const float4 a = recs[3 * g], b = recs[3 * g + 1], c = recs[3 * g + 2];
                lds_rec[3 * k] = make_float4(a.x, a.y, 0.5f * LOG2E * a.z, LOG2E * a.w);
                lds_rec[3 * k + 1] = make_float4(0.5f * LOG2E * b.x, -__log2f(b.y), b.z, b.w);
                lds_rec[3 * k + 2] = make_float4(c.x, c.y, 0.f, 0.f);
                const int xb = __float_as_int(c.z), yb = __float_as_int(c.w);
                const int x_lo = (int)(short)(xb & 0xffff), x_hi = xb >> 16, y_lo = (int)(short)(yb & 0xffff), y_hi = yb >> 16;
                const bool in_x = x_lo <= x_hi && y_lo <= y_hi && x_lo <= tx * 16 + 15 && x_hi >= tx * 16;
                h0[u] = in_x && y_lo <= ty * 16 + 7 && y_hi >= ty * 16;
                h1[u] = in_x && y_lo <= ty * 16 + 15 && y_hi >= ty * 16 + 8;
            }
            const unsigned long long m0 = __ballot(h0[u]), m1 = __ballot(h1[u]);
            r0[u] = __popcll(m0 & lt); r1[u] = __popcll(m1 & lt);
            const int seg = u * (FWD_THREADS / 64) + wave;
            if (lane == 0 && seg < FWD_SEGS) { scnt[0][seg] = __popcll(m0); scnt[1][seg] = __popcll(m1); }
        }
        FWD_STAMP(1);
        __syncthreads();
        // exclusive prefix of the segments' survivor counts, per half (wave-uniform values); S = all survivors of the half this
        // wave will EVALUATE
        int pre0[FWD_SEGS + 1], pre1[FWD_SEGS + 1];
        pre0[0] = 0; pre1[0] = 0;
#pragma unroll
        for (int w = 0; w < FWD_SEGS; w++) { pre0[w + 1] = pre0[w] + scnt[0][w]; pre1[w + 1] = pre1[w] + scnt[1][w]; }
        int S = pix_half ? pre1[FWD_SEGS] : pre0[FWD_SEGS];
#pragma unroll
        for (int u = 0; u < FWD_TRIPS; u++) {
            const int k = u * FWD_THREADS + tid;
            int b0 = 0, b1 = 0;
#pragma unroll
            for (int w = 0; w < FWD_SEGS; w++)   // (the segment index is wave-uniform: a scalar select)
                if (w == u * (FWD_THREADS / 64) + wave) { b0 = pre0[w]; b1 = pre1[w]; }
            if (h0[u]) sidx[0][b0 + r0[u]] = (uint16_t)(3 * k);
            if (h1[u]) sidx[1][b1 + r1[u]] = (uint16_t)(3 * k);
        }
        __syncthreads();
        FWD_STAMP(2);
        S = __builtin_amdgcn_readfirstlane(S);
        const int lo = list_part * S / FWD_SPLIT, cnt = (list_part + 1) * S / FWD_SPLIT - lo;   // this wave's survivors
        // lane j of vector k holds the LDS byte address of survivor lo + 64 k + j
        int addr[FWD_VECS];
#pragma unroll
        for (int k = 0; k < FWD_VECS; k++) addr[k] = (64 * k + lane < cnt) ? 16 * (int)sidx[pix_half][lo + 64 * k + lane] : 0;
        const char* rec_bytes = reinterpret_cast<const char*>(lds_rec);
        auto blend = [&](int byte_off) {
            const float4 a = *reinterpret_cast<const float4*>(rec_bytes + byte_off);
            const float4 b = *reinterpret_cast<const float4*>(rec_bytes + byte_off + 16);
            const float2 c = *reinterpret_cast<const float2*>(rec_bytes + byte_off + 32);
            const float dy = a.y - py;
            const v2f dx = a.x - px;
            const float cdy2 = b.x * dy * dy, bdy = a.w * dy;
            const v2f w = a.z * dx + bdy;
            const v2f sig = w * dx + cdy2;        // sigma * log2(e)
            const v2f e = sig + b.y;              // sigma' - log2(opacity)
            float al0 = fminf(0.999f, __builtin_amdgcn_exp2f(-e.x));
            float al1 = fminf(0.999f, __builtin_amdgcn_exp2f(-e.y));
            const bool hit0 = !(b.z > cut0) && !(sig.x < 0.f) && !(al0 < 1.f / 255.f);
            const bool hit1 = !(b.z > cut1) && !(sig.y < 0.f) && !(al1 < 1.f / 255.f);
            const v2f al = {hit0 ? al0 : 0.f, hit1 ? al1 : 0.f};
            o0 += b.w * al; o1 += c.x * al; o2 += c.y * al; o3 += b.z * al; ws += al;
        };
#pragma unroll
        for (int k = 0; k < FWD_VECS; k++) {
            const int n_k = min(cnt - 64 * k, 64);   // wave-uniform
            for (int j = 0; j + 1 < n_k; j += 2) {   // two survivors per trip: their records leave LDS together
                const int t0 = __builtin_amdgcn_readlane(addr[k], j), t1 = __builtin_amdgcn_readlane(addr[k], j + 1);
                blend(t0);
                blend(t1);
            }
            if (n_k > 0 && (n_k & 1)) blend(__builtin_amdgcn_readlane(addr[k], n_k - 1));
        }
    }
    // list parts 1.. -> LDS (over the records) -> part 0 adds them in order and stores
    FWD_STAMP(3);
    __syncthreads();   // every wave is done with the records
    float* part = reinterpret_cast<float*>(lds_rec);
    const int slot = (pix_half * 64 + lane) * 10;
    if (list_part) {
        float* q = part + (list_part - 1) * 1280 + slot;
        q[0] = o0.x; q[1] = o1.x; q[2] = o2.x; q[3] = o3.x; q[4] = ws.x;
        q[5] = o0.y; q[6] = o1.y; q[7] = o2.y; q[8] = o3.y; q[9] = ws.y;
    }
    __syncthreads();
    FWD_STAMP(4);
    if (!list_part) {
        const int pix = row * W + col;
#pragma unroll
        for (int k = 0; k < FWD_SPLIT - 1; k++) {
            const float* q = part + k * 1280 + slot;
            o0.x += q[0]; o1.x += q[1]; o2.x += q[2]; o3.x += q[3]; ws.x += q[4];
            o0.y += q[5]; o1.y += q[6]; o2.y += q[7]; o3.y += q[8]; ws.y += q[9];
        }
        const float4 c0 = make_float4(o0.x, o1.x, o2.x, o3.x);
        const float4 c1 = make_float4(o0.y, o1.y, o2.y, o3.y);
        const float w0 = ws.x, w1 = ws.y;
        if (in0) { render_colors[pix] = c0; render_alphas[pix] = w0; }
        if (in1) { render_colors[pix + 1] = c1; render_alphas[pix + 1] = w1; }
        if (fc.base_color) {
            // compose + L1 + image gradients of the two pixels, operation for operation what compose_l1_kernel does
            float lsum = 0.f;
            if (in0) lsum += compose_l1_pixel(fc, pix, c0, w0, cut0);
            if (in1) lsum += compose_l1_pixel(fc, pix + 1, c1, w1, cut1);
            lsum = wave_sum(lsum);
            if (lane == 0) atomicAdd(fc.loss, lsum * fc.inv_count);
        }
    }
    FWD_STAMP(5);
}


// ---------------------------------------------------------------------------------------------------------
// The same forward as a PERSISTENT launch (round 6; the round-5 review's item 3).  BUILT, BIT-IDENTICAL, MEASURED SLOWER -- kept behind
// gps_set_raster_fwd_persistent(1) with its equality test, off by default.  The idea: raster_ges_fwd_pk_kernel's workgroups pay three
// dependent loads (tile offsets -> list -> records) + barriers before they evaluate, all workgroups of a compute unit go through
// those phases in lock-step, and 1,200 tiles on 1,024 workgroup slots leave a nearly empty second round; so let PP_WGS_PER_CU
// workgroups per compute unit stay resident and walk a STATIC share of the tiles:
//   * dealing: the binning's tile_order (tiles by descending list length) is dealt in a snake -- pass p gives workgroup w the
//     tile at position p G + w (p even) or (p + 1) G - 1 - w (p odd): the longest list of one pass meets the shortest of the
//     next, no tickets, no atomics;
//   * cross-tile staging: the work is a stream of (tile, 512-entry batch) items.  While item s is evaluated out of LDS the
//     records of item s + 1 are on their way into registers (their list entries were fetched one item earlier still) and the
//     list entries of item s + 2 are being fetched (unconditional loads from clamped indices: a load under a lane mask, or a
//     select on its value, makes the compiler wait on the spot); the next tile's reference depths come in the same way;
//   * everything per (pixel, entry) -- staging arithmetic, culling, survivor partition, blend, the order partial sums are added
//     in -- is the per-tile kernel's, so the images are BIT-IDENTICAL to it (tests/test_splat_gpu.py).
// Measured (tools/probe/fwd_pp_ab.py, fwd_pp_stamps.py; 640x480 / 499 k list entries and 1200x680 / 951 k): 51.5 us against 42.7
// (per-tile, ordered) and 91.4 against 72.8.  The stamps say why: the look-ahead works (a further tile costs 11.6 / 12.7 us
// against 19-28 of residency in the per-tile launch), but (i) the FIRST tile of every workgroup -- the 768 longest lists --
// starts at t = 0 everywhere: 7 us of dependent loads during which no wave of the chip evaluates, then 16.5 us (median, 26 max)
// in which every wave evaluates at once and the compute units are VALU-bound (13.7 M of the launch's 18.3 M wave instructions
// = 11 us at full issue rate), then a 4 us tail per tile waiting for its slowest wave; (ii) 80 VGPRs (look-ahead registers) allow 6
// waves per SIMD, not 8.  What bounds the forward is its VALU work (15-18 us at these sizes) plus one synchronised start-up
// and per-tile tails, not the load chain of the later tiles; the per-tile launch already hides that chain behind the other
// three workgroups of its compute unit.
constexpr int PP_MAX_PASSES = 32;   // tiles one persistent workgroup can be dealt (more: the per-tile kernel)
constexpr int PP_WGS_PER_CU = 3;    // 8 waves each at <= 80 VGPRs and 27 KB of LDS

__global__ __launch_bounds__(FWD_THREADS, 2 * PP_WGS_PER_CU) void raster_ges_fwd_pp_kernel(
    const float4* __restrict__ recs, const float* __restrict__ ref_depth, int W, int H, int tw, int th,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    const int64_t* __restrict__ counts, float delta_depth, float4* __restrict__ render_colors,
    float* __restrict__ render_alphas, gps::FwdCompose fc, const int32_t* __restrict__ tile_order, gps::LaunchStamp stamp) {
    static_assert(FWD_TRIPS == 1, "one list entry per thread and batch");
    gps::StampScope timed(stamp);
    constexpr int PART_FLOATS = (FWD_SPLIT - 1) * 128 * 10;
    constexpr int REC_FLOATS = FWD_BATCH * 12;
    __shared__ float4 lds_rec[(REC_FLOATS > PART_FLOATS ? REC_FLOATS : PART_FLOATS) / 4];
    __shared__ uint16_t sidx[2][FWD_BATCH];
    __shared__ int scnt[2][FWD_SEGS];
    __shared__ int t_tile[PP_MAX_PASSES], t_lo[PP_MAX_PASSES], t_hi[PP_MAX_PASSES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int list_part = wave >> 1, pix_half = wave & 1;
    const int T = tw * th, G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int n_pass = (T + G - 1) / G;
    const int n_isects = (int)counts[0];
    if (tid < n_pass) {
        const int pos = (tid & 1) ? (tid + 1) * G - 1 - wg : tid * G + wg;
        int tile = -1, lo = 0, hi = 0;
        if (pos < T) {
            tile = tile_order ? tile_order[pos] : pos;
            lo = tile_offsets[tile];
            hi = (tile == T - 1) ? n_isects : tile_offsets[tile + 1];
        }
        t_tile[tid] = tile; t_lo[tid] = lo; t_hi[tid] = hi;
    }
    FWD_STAMP(0);
    __syncthreads();
    FWD_STAMP(1);
    // (positions past the last tile only occur in the last pass)
    const int n_mine = __builtin_amdgcn_readfirstlane(n_pass > 0 && t_tile[n_pass - 1] < 0 ? n_pass - 1 : n_pass);
    if (n_mine <= 0) return;
    constexpr float LOG2E = 1.4426950408889634f;
    const unsigned long long lt = lanemask_lt();

    // the item stream: (index into my tiles, batch start); uniform over the workgroup
    struct Item { int ti, bs; };
    auto item_next = [&](Item it) {
        if (it.ti >= n_mine) return it;
        if (it.bs + FWD_BATCH < t_hi[it.ti]) { it.bs += FWD_BATCH; return it; }
        it.ti++;
        it.bs = it.ti < n_mine ? t_lo[it.ti] : 0;
        return it;
    };
    // Every load of the look-ahead is UNCONDITIONAL, from a clamped index, and nothing is computed from its result before the item
    // that consumes it: a load under a lane mask (or a select on its value) makes the compiler wait for it on the spot -- the
    // first build of this kernel did, and ran behind its own "prefetches".  Validity travels separately, from the indices alone.
    auto has_entry = [&](Item it) -> bool { return it.ti < n_mine && it.bs + tid < t_hi[it.ti]; };
    auto load_id = [&](Item it) -> int {   // this thread's list entry of the item (some valid entry's id when it has none)
        const int idx = it.ti < n_mine ? it.bs + tid : 0;
        return flatten_ids[max(0, min(idx, n_isects - 1))];   // (no list at all: word 0 of the buffer, never used as an id -- has_entry)
    };
    auto pixel_of = [&](int tile, int& row, int& col) {
        const int ty = tile / tw, tx = tile - ty * tw;
        row = ty * 16 + pix_half * 8 + (lane >> 3); col = tx * 16 + 2 * (lane & 7);
    };
    auto load_depths = [&](int tile, float& d0, float& d1) {   // raw reference depths of the thread's two pixels (clamped into the image)
        int row, col;
        pixel_of(tile, row, col);
        const int r = min(row, H - 1);
        d0 = ref_depth[r * W + min(col, W - 1)];
        d1 = ref_depth[r * W + min(col + 1, W - 1)];
    };
    auto cuts_of = [&](int row, int col, float d0, float d1, float& c0, float& c1) {   // pixels outside the image never pass the depth test
        c0 = (row < H && col < W) ? d0 + delta_depth : -3.0e38f;
        c1 = (row < H && col + 1 < W) ? d1 + delta_depth : -3.0e38f;
    };

    Item A = {0, t_lo[0]}, B = item_next(A), C = item_next(B);
    const int idA = load_id(A);
    int idB = load_id(B);
    bool haveA = has_entry(A), haveB = has_entry(B);
    const size_t gA = haveA ? (size_t)idA : 0;   // (record 0 stands in where the thread has no entry: N > 0)
    float4 Ra = recs[3 * gA], Rb = recs[3 * gA + 1], Rc = recs[3 * gA + 2];
    int row, col;
    pixel_of(t_tile[0], row, col);
    float cut0, cut1, dn0, dn1;
    load_depths(t_tile[0], dn0, dn1);
    cuts_of(row, col, dn0, dn1, cut0, cut1);
    v2f o0 = {0.f, 0.f}, o1 = o0, o2 = o0, o3 = o0, ws = o0;
    const char* rec_bytes = reinterpret_cast<const char*>(lds_rec);

    while (A.ti < n_mine) {
        const int tile = t_tile[A.ti], hiA = t_hi[A.ti];
        const int ty = tile / tw, tx = tile - ty * tw;
        const bool empty = A.bs >= hiA;               // a tile without a list: straight to its epilogue
        const bool last_batch = empty || A.bs + FWD_BATCH >= hiA;
        const v2f px = {(float)col + 0.5f, (float)col + 1.5f};
        const float py = (float)row + 0.5f;
        __syncthreads();   // the previous item's records, lists and partial sums are consumed
        bool h0 = false, h1 = false;
        if (!empty) {
            // ---- stage item A (its records are in registers) + cull per 16 x 8 half, exactly as the per-tile kernel does
            if (haveA) {
                const int k = tid;
                lds_rec[3 * k] = make_float4(Ra.x, Ra.y, 0.5f * LOG2E * Ra.z, LOG2E * Ra.w);
                lds_rec[3 * k + 1] = make_float4(0.5f * LOG2E * Rb.x, -__log2f(Rb.y), Rb.z, Rb.w);
                lds_rec[3 * k + 2] = make_float4(Rc.x, Rc.y, 0.f, 0.f);
                const int xb = __float_as_int(Rc.z), yb = __float_as_int(Rc.w);
                const int x_lo = (int)(short)(xb & 0xffff), x_hi = xb >> 16, y_lo = (int)(short)(yb & 0xffff), y_hi = yb >> 16;
                const bool in_x = x_lo <= x_hi && y_lo <= y_hi && x_lo <= tx * 16 + 15 && x_hi >= tx * 16;
                h0 = in_x && y_lo <= ty * 16 + 7 && y_hi >= ty * 16;
                h1 = in_x && y_lo <= ty * 16 + 15 && y_hi >= ty * 16 + 8;
            }
        }
        // ---- the loads of the items behind: records of B (its entries arrived an item ago), entries of C, the next tile's depths
        const size_t gB = haveB ? (size_t)idB : 0;
        Ra = recs[3 * gB]; Rb = recs[3 * gB + 1]; Rc = recs[3 * gB + 2];
        const int idC = load_id(C);
        const bool haveC = has_entry(C);
        load_depths(t_tile[min(A.ti + 1, n_mine - 1)], dn0, dn1);
        if (!empty) {
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            const int r0 = __popcll(m0 & lt), r1 = __popcll(m1 & lt);
            if (lane == 0) { scnt[0][wave] = __popcll(m0); scnt[1][wave] = __popcll(m1); }
            __syncthreads();
            int pre0[FWD_SEGS + 1], pre1[FWD_SEGS + 1];
            pre0[0] = 0; pre1[0] = 0;
#pragma unroll
            for (int w = 0; w < FWD_SEGS; w++) { pre0[w + 1] = pre0[w] + scnt[0][w]; pre1[w + 1] = pre1[w] + scnt[1][w]; }
            int S = pix_half ? pre1[FWD_SEGS] : pre0[FWD_SEGS];
            int b0 = 0, b1 = 0;
#pragma unroll
            for (int w = 0; w < FWD_SEGS; w++)
                if (w == wave) { b0 = pre0[w]; b1 = pre1[w]; }
            if (h0) sidx[0][b0 + r0] = (uint16_t)(3 * tid);
            if (h1) sidx[1][b1 + r1] = (uint16_t)(3 * tid);
            __syncthreads();
            if (A.ti == 0 && A.bs == t_lo[0]) FWD_STAMP(2);
            S = __builtin_amdgcn_readfirstlane(S);
            const int lo = list_part * S / FWD_SPLIT, cnt = (list_part + 1) * S / FWD_SPLIT - lo;
            int addr[FWD_VECS];
#pragma unroll
            for (int k = 0; k < FWD_VECS; k++) addr[k] = (64 * k + lane < cnt) ? 16 * (int)sidx[pix_half][lo + 64 * k + lane] : 0;
            auto blend = [&](int byte_off) {
                const float4 a = *reinterpret_cast<const float4*>(rec_bytes + byte_off);
                const float4 b = *reinterpret_cast<const float4*>(rec_bytes + byte_off + 16);
                const float2 c = *reinterpret_cast<const float2*>(rec_bytes + byte_off + 32);
                const float dy = a.y - py;
                const v2f dx = a.x - px;
                const float cdy2 = b.x * dy * dy, bdy = a.w * dy;
                const v2f w = a.z * dx + bdy;
                const v2f sig = w * dx + cdy2;
                const v2f e = sig + b.y;
                float al0 = fminf(0.999f, __builtin_amdgcn_exp2f(-e.x));
                float al1 = fminf(0.999f, __builtin_amdgcn_exp2f(-e.y));
                const bool hit0 = !(b.z > cut0) && !(sig.x < 0.f) && !(al0 < 1.f / 255.f);
                const bool hit1 = !(b.z > cut1) && !(sig.y < 0.f) && !(al1 < 1.f / 255.f);
                const v2f al = {hit0 ? al0 : 0.f, hit1 ? al1 : 0.f};
                o0 += b.w * al; o1 += c.x * al; o2 += c.y * al; o3 += b.z * al; ws += al;
            };
#pragma unroll
            for (int k = 0; k < FWD_VECS; k++) {
                const int n_k = min(cnt - 64 * k, 64);
                for (int j = 0; j + 1 < n_k; j += 2) {
                    const int t0 = __builtin_amdgcn_readlane(addr[k], j), t1 = __builtin_amdgcn_readlane(addr[k], j + 1);
                    blend(t0);
                    blend(t1);
                }
                if (n_k > 0 && (n_k & 1)) blend(__builtin_amdgcn_readlane(addr[k], n_k - 1));
            }
        }
        if (A.ti == 0 && last_batch) FWD_STAMP(3);
        if (last_batch) {
            // ---- the tile's epilogue: list parts 1.. -> LDS (over the records) -> part 0 adds them in order and stores
            __syncthreads();
            if (A.ti == 0) FWD_STAMP(4);
            float* part = reinterpret_cast<float*>(lds_rec);
            const int slot = (pix_half * 64 + lane) * 10;
            if (list_part) {
                float* q = part + (list_part - 1) * 1280 + slot;
                q[0] = o0.x; q[1] = o1.x; q[2] = o2.x; q[3] = o3.x; q[4] = ws.x;
                q[5] = o0.y; q[6] = o1.y; q[7] = o2.y; q[8] = o3.y; q[9] = ws.y;
            }
            __syncthreads();
            if (!list_part) {
                const bool in0 = (row < H) && (col < W), in1 = (row < H) && (col + 1 < W);
                const int pix = row * W + col;
#pragma unroll
                for (int k = 0; k < FWD_SPLIT - 1; k++) {
                    const float* q = part + k * 1280 + slot;
                    o0.x += q[0]; o1.x += q[1]; o2.x += q[2]; o3.x += q[3]; ws.x += q[4];
                    o0.y += q[5]; o1.y += q[6]; o2.y += q[7]; o3.y += q[8]; ws.y += q[9];
                }
                const float4 c0 = make_float4(o0.x, o1.x, o2.x, o3.x);
                const float4 c1 = make_float4(o0.y, o1.y, o2.y, o3.y);
                const float w0 = ws.x, w1 = ws.y;
                if (in0) { render_colors[pix] = c0; render_alphas[pix] = w0; }
                if (in1) { render_colors[pix + 1] = c1; render_alphas[pix + 1] = w1; }
                if (fc.base_color) {
                    float lsum = 0.f;
                    if (in0) lsum += compose_l1_pixel(fc, pix, c0, w0, cut0);
                    if (in1) lsum += compose_l1_pixel(fc, pix + 1, c1, w1, cut1);
                    lsum = wave_sum(lsum);
                    if (lane == 0) atomicAdd(fc.loss, lsum * fc.inv_count);
                }
            }
            if (A.ti == 0) FWD_STAMP(5);
            o0 = {0.f, 0.f}; o1 = o0; o2 = o0; o3 = o0; ws = o0;
            if (A.ti + 1 < n_mine) { pixel_of(t_tile[A.ti + 1], row, col); cuts_of(row, col, dn0, dn1, cut0, cut1); }
        }
        haveA = haveB; haveB = haveC;
        idB = idC;
        A = B; B = C; C = item_next(C);
    }
    FWD_STAMP(6);
#ifdef GPS_FWD_STAMPS
    if (threadIdx.x == 0) gps_fwd_stamps_buf[blockIdx.x * 8 + 7] = (unsigned long long)n_mine;
#endif
}


__global__ __launch_bounds__(256) void zero_grads_kernel(int N, float* __restrict__ v_means2d,
                                                        float* __restrict__ v_conics, float* __restrict__ v_colors,
                                                        float* __restrict__ v_opacities) {
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < 4 * N; k += stride) {
        v_colors[k] = 0.f;
        if (k < 3 * N) v_conics[k] = 0.f;
        if (k < 2 * N) v_means2d[k] = 0.f;
        if (k < N) v_opacities[k] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward.  One wave64 task = 32 consecutive 32-pixel groups, as two contiguous runs of 16 (one per half-wave).
// Phase A: lanes 0..15 of each half fetch their step's group header and Gaussian record and park them in LDS --
//          ONE global round trip per task instead of one dependent chain per step.
// Phase B: 16 steps; every lane of a half reads its step's record from LDS, evaluates its pixel of the 2r x 2r box and
//          accumulates in registers.  BWD_INFLIGHT steps are evaluated before the first is accumulated; with the straight-line
//          evaluation below ONE step at a time is fastest (68 VGPRs, 7 waves per SIMD: 62.5 us against 65.5 for two steps
//          at 82 VGPRs / 5 waves and 70 for four): other waves cover the gather better than a second step in this one.  A
//          half reduces and writes its 10 sums only when the Gaussian changes.
//
// Per-pixel inputs: three gathers per pixel slot (ref_depth 4 B, v_render_colors 16 B, v_render_alphas 4 B).  Packing them into
// one 32-byte record per pixel (written by the compose kernel; 2 x dwordx4 from one sector) was built and measured SLOWER
// (84 vs 79 us at G = 668 k): the record array is 9.8 MB against 7.3 MB for the three arrays, and what limits the gathers is
// how much of the gradient image each XCD's 4 MB L2 holds, not the number of load instructions.
//
// LDS layout.  Reads are broadcasts (all 32 lanes of a half read one record: conflict free whatever the layout); the bank
// conflicts the counters show (SQ_LDS_BANK_CONFLICT) come from phase A's STORES -- 16 lanes each writing a 64-byte record.
// Records of a half are contiguous (stride 16 dwords: 8-way on a ds_write_b128); interleaving the halves per step (stride 32
// dwords) was measured worse (conflict cycles 3.1e6 -> 7.2e6 per launch), and either way they are < 3 % of the wave cycles.
//
// Flush.  Ten values x 32 lanes -> ten totals.  v_permlane16_swap exchanges the odd rows of one register with the even rows of
// another, so "swap, add" halves the lane span of TWO values at once: five swaps + five adds leave five registers whose two
// 16-lane rows hold one value each; four DPP row_shr adds per register finish the sums (lane 15 of a row); four DPP row_shl
// moves line the ten totals up in ten lanes so that ONE memory instruction writes them: ~45 VALU instead of 85 for ten full
// half-wave DPP reductions.
typedef float bwd_v4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_row_add(float v) {  // v + (v moved by a row-local DPP shift; lanes without a source add 0)
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float row_total_to_lane15(float v) {
    v = dpp_row_add<0x111>(v);  // row_shr:1
    v = dpp_row_add<0x112>(v);  // row_shr:2
    v = dpp_row_add<0x114>(v);  // row_shr:4
    v = dpp_row_add<0x118>(v);  // row_shr:8
    return v;
}
// rows of the result: {a over rows 0+1, b over rows 0+1, a over rows 2+3, b over rows 2+3} -- per half-wave: row 0 = a, row 1 = b
__device__ __forceinline__ float pair_rows(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lane 15 of each row -> lane 15 - K of the same row (row_shl:K), merged into `packed` there
template <int K>
__device__ __forceinline__ float place(float packed, float v, int lane16) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + K, 0xF, 0xF, true);
    return lane16 == 15 - K ? __int_as_float(moved) : packed;
}

struct Acc { float c0, c1, c2, c3, ka, kb, kc, mx, my, op; };

// Reduce the half-wave's 10 partial sums and add them to Gaussian g's gradient rows.
//   exclusive = this half-wave saw every pixel group of g (the segment neither touches the start nor the end of the
//   half's 16-group run) and the arrays were zeroed by this launch: a plain store, no read-modify-write at the L2.
//   otherwise: float atomics (another half-wave may hold the rest of g, or the caller accumulates across launches).
__device__ __forceinline__ void flush_acc(Acc& a, int g, int hl, bool exclusive, float* __restrict__ v_means2d,
                                          float* __restrict__ v_conics, float* __restrict__ v_colors,
                                          float* __restrict__ v_opacities) {
    const int lane16 = hl & 15, row = hl >> 4;
    // value index 2 j + row sits in register j: (c0,c1) (c2,c3) (ka,kb) (kc,mx) (my,op)
    float v = row_total_to_lane15(pair_rows(a.c0, a.c1));
    v = place<1>(v, row_total_to_lane15(pair_rows(a.c2, a.c3)), lane16);
    v = place<2>(v, row_total_to_lane15(pair_rows(a.ka, a.kb)), lane16);
    v = place<3>(v, row_total_to_lane15(pair_rows(a.kc, a.mx)), lane16);
    v = place<4>(v, row_total_to_lane15(pair_rows(a.my, a.op)), lane16);
    const int j = 15 - lane16;
    if (g >= 0 && j < 5) {
        const int k = 2 * j + row;  // which total this lane holds
        float* dst = k < 4 ? v_colors + 4 * (size_t)g + k
                   : k < 7 ? v_conics + 3 * (size_t)g + (k - 4)
                   : k < 9 ? v_means2d + 2 * (size_t)g + (k - 7)
                           : v_opacities + g;
        if (exclusive) *dst = v;
        else if (v != 0.f) atomicAdd(dst, v);
    }
    a.c0 = a.c1 = a.c2 = a.c3 = a.ka = a.kb = a.kc = a.mx = a.my = a.op = 0.f;
}

struct __attribute__((aligned(16))) BwdRec {
    float x, y, ca, cb;      // xy, conic a, b
    float cc, opac, r, g;    // conic c, opacity, colour r, g
    float b, depth;          // colour b, depth channel
    int gs_id, pid0;         // Gaussian id (-1 = no group), first pixel slot of the group inside the box
    int x0, y0, bw;          // x_min + 1, y_min + 1, box width 2r (y_max = y0 + bw - 1)
    float inv_bw;            // 1 / bw: (pid + 0.5) * inv_bw truncates to pid / bw exactly for pid < 2^16
};

#ifndef GPS_BWD_INFLIGHT
#define GPS_BWD_INFLIGHT 1
#endif
GPS_TUNABLE_REPORT(GPS_BWD_INFLIGHT, 1);
constexpr int BWD_INFLIGHT = GPS_BWD_INFLIGHT;
struct BwdPix { bool on; float alpha, vis, dx, dy, cut; float4 vc; float va; };

struct BwdPixArgs {  // kernel arguments
    const float* ref_depth;
    const float4* v_render_colors;
    const float* v_render_alphas;
    float delta_depth;
};
struct BwdPixSrc {  // the three per-pixel arrays as buffer resources (W * H elements each)
    __amdgpu_buffer_rsrc_t ref_depth, v_render_colors, v_render_alphas;
    float delta_depth;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pixel_buffer(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);  // raw dword buffer, stride 0
}

// Straight-line on purpose.  With the gathers under the alpha-test branch and their use under bwd_accum's branch, the
// compiler cannot tell at the loop's back edge whether a gather is still pending and guards the next step's address
// registers with s_waitcnt vmcnt(0) -- which (vmcnt counts in order) also waits for the gathers the previous step has just
// issued: one full memory round trip per step, SQ_WAIT_ANY 55 %.  Here every lane always loads (slots that fail the test
// read pixel 0: one extra address per wave), and bwd_landed() makes the arrival of a step's three values an unconditional
// point of the loop, so two steps' gathers are really in flight together.
__device__ __forceinline__ void bwd_eval(const BwdRec& R, int hl, int W, int H, const BwdPixSrc& src, BwdPix& o) {
    const uint32_t pid = (uint32_t)R.pid0 + (uint32_t)hl;
    const int q = (int)(((float)pid + 0.5f) * R.inv_bw);  // pid / bw
    const int j = R.x0 + ((int)pid - q * R.bw);
    const int i = R.y0 + q;
    const bool inside = (R.gs_id >= 0) && (i < H) && (j < W) && (i >= 0) && (j >= 0) && (q < R.bw);
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    o.dx = R.x - px; o.dy = R.y - py;
    const float sigma = 0.5f * (R.ca * o.dx * o.dx + R.cc * o.dy * o.dy) + R.cb * o.dx * o.dy;
    o.vis = __expf(-sigma);
    o.alpha = fminf(0.999f, R.opac * o.vis);
    // the {alpha >= 1/255} ellipse fills 45 % of the 2r x 2r box the groups enumerate (tools/raster_bench.py)
    o.on = inside && !((sigma < 0.f) || (o.alpha < 1.f / 255.f));
    // raw buffer loads: a 32-bit byte offset per lane instead of three 64-bit addresses, and a slot that fails the test
    // passes an out-of-range offset -- the load returns 0 without touching memory
    const uint32_t pix = o.on ? (uint32_t)(i * W + j) : 0x0FFFFFFFu;
    o.cut = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src.ref_depth, pix * 4u, 0, 0));  // (tested in bwd_accum)
    o.vc = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(src.v_render_colors, pix * 16u, 0, 0));
    o.va = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src.v_render_alphas, pix * 4u, 0, 0));
}

__device__ __forceinline__ void bwd_landed(BwdPix& p) {
    asm volatile("" : "+v"(p.cut), "+v"(p.va), "+v"(p.vc.x), "+v"(p.vc.y), "+v"(p.vc.z), "+v"(p.vc.w));
}

__device__ __forceinline__ void bwd_accum(const BwdRec& R, const BwdPix& p, float delta_depth, Acc& acc) {
    if (!p.on || R.depth > p.cut + delta_depth) return;
    acc.c0 += p.alpha * p.vc.x; acc.c1 += p.alpha * p.vc.y; acc.c2 += p.alpha * p.vc.z; acc.c3 += p.alpha * p.vc.w;
    const float v_alpha = R.r * p.vc.x + R.g * p.vc.y + R.b * p.vc.z + R.depth * p.vc.w + p.va;
    if (R.opac * p.vis <= 0.999f) {
        const float v_sigma = -R.opac * p.vis * v_alpha;
        acc.ka += 0.5f * v_sigma * p.dx * p.dx;
        acc.kb += v_sigma * p.dx * p.dy;
        acc.kc += 0.5f * v_sigma * p.dy * p.dy;
        acc.mx += v_sigma * (R.ca * p.dx + R.cb * p.dy);
        acc.my += v_sigma * (R.cb * p.dx + R.cc * p.dy);
        acc.op += p.vis * v_alpha;
    }
}

__global__ __launch_bounds__(256) void raster_ges_bwd_gs_kernel(
    const int32_t* __restrict__ group_gs_ids, const int32_t* __restrict__ group_starts,
    const float2* __restrict__ means2d, const float* __restrict__ conics, const float4* __restrict__ colors,
    const float* __restrict__ opacities, const int32_t* __restrict__ radiis, BwdPixArgs pix_args,
    const int64_t* __restrict__ counts, int W, int H, float* __restrict__ v_means2d, float* __restrict__ v_conics,
    float* __restrict__ v_colors, float* __restrict__ v_opacities, int plain_ok) {
    __shared__ BwdRec recs[4][2][16];  // [wave][half][step]
    const uint32_t n_px = (uint32_t)(W * H);
    const BwdPixSrc src = {pixel_buffer(pix_args.ref_depth, n_px * 4u), pixel_buffer(pix_args.v_render_colors, n_px * 16u),
                           pixel_buffer(pix_args.v_render_alphas, n_px * 4u), pix_args.delta_depth};
    const int n_groups = (int)counts[1];
    const int n_tasks = (n_groups + 31) >> 5;
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int half = lane >> 5, hl = lane & 31;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); give each XCD ONE contiguous eighth of the task
    // list instead of a stride-8 comb through all of it: Gaussians with neighbouring ids cover neighbouring pixels, so
    // an XCD then gathers from one band of the gradient image, which its 4 MB L2 can hold.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;   // gridDim.x % 8 == 0
    const int per_xcd = (n_tasks + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tasks, xcd_lo + per_xcd);
    const int wave_in_xcd = slot * 4 + wave_in_wg, waves_per_xcd = wgs_per_xcd * 4;
    BwdRec* my = recs[wave_in_wg][half];

    for (int task = xcd_lo + wave_in_xcd; task < xcd_hi; task += waves_per_xcd) {
        // ---- phase A: one record per step, fetched by lanes 0..15 of each half
        if (hl < 16) {
            const int gid = task * 32 + half * 16 + hl;
            BwdRec R;
            R.gs_id = -1;
            if (gid < n_groups) {
                const int g = group_gs_ids[gid];
                const int gstart = group_starts[gid];
                const float2 xy = means2d[g];
                const float4 c = colors[g];
                const int r = radiis[g];
                R.x = xy.x; R.y = xy.y;
                R.ca = conics[3 * g]; R.cb = conics[3 * g + 1]; R.cc = conics[3 * g + 2];
                R.opac = opacities[g];
                R.r = c.x; R.g = c.y; R.b = c.z; R.depth = c.w;
                R.gs_id = g;
                R.pid0 = (gid - gstart) * 32;
                R.x0 = (int)xy.x - r + 1; R.y0 = (int)xy.y - r + 1; R.bw = 2 * r; R.inv_bw = 1.0f / (float)(2 * r);
            }
            my[hl] = R;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // ---- phase B
        Acc acc;
        acc.c0 = acc.c1 = acc.c2 = acc.c3 = acc.ka = acc.kb = acc.kc = acc.mx = acc.my = acc.op = 0.f;
        int cur_g = -1;
        bool first_seg = true;  // the current segment began at step 0 of the run (it may continue a neighbour's)
        for (int s = 0; s < 16; s += BWD_INFLIGHT) {
            BwdPix p[BWD_INFLIGHT];
#pragma unroll
            for (int u = 0; u < BWD_INFLIGHT; ++u) bwd_eval(my[s + u], hl, W, H, src, p[u]);
#pragma unroll
            for (int u = 0; u < BWD_INFLIGHT; ++u) {
                const BwdRec R = my[s + u];
                if (R.gs_id != cur_g) {
                    // the segment that ends here started after step 0 and ends before the run does
                    if (cur_g >= 0)
                        flush_acc(acc, cur_g, hl, plain_ok && !first_seg, v_means2d, v_conics, v_colors, v_opacities);
                    first_seg = cur_g < 0;  // still no Gaussian seen (cannot happen after a real segment)
                    cur_g = R.gs_id;
                }
                bwd_landed(p[u]);
                bwd_accum(R, p[u], src.delta_depth, acc);
            }
        }
        if (cur_g >= 0) flush_acc(acc, cur_g, hl, false, v_means2d, v_conics, v_colors, v_opacities);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// persistent forward: OFF by default (measured slower, see the kernel's header); workgroups = PP_WGS_PER_CU per compute unit of the
// current device (asked once)
static int g_fwd_persistent = 0;
static int fwd_persistent_workgroups() {
    if (!__atomic_load_n(&g_fwd_persistent, __ATOMIC_RELAXED)) return 0;
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        (void)hipGetLastError();
        return n;
    }();
    return cus * PP_WGS_PER_CU;
}

namespace gps {

int raster_ges_fwd_rec_launch(int N, const float* records, const float* ref_depth_map, int width, int height,
                              const int32_t* tile_offsets, const int32_t* flatten_ids, const int64_t* counts, float delta_depth,
                              float* render_colors, float* render_alphas, const FwdCompose* compose, gps_stream stream,
                              const int32_t* tile_order) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0);
    GPS_REQUIRE(ref_depth_map && tile_offsets && flatten_ids && counts && render_colors && render_alphas);
    GPS_REQUIRE(N == 0 || records);
    FwdCompose fc = {};
    if (compose) {
        fc = *compose;
        GPS_REQUIRE(fc.base_color && fc.gt_rgb && fc.rgb && fc.loss && fc.v_render_colors && fc.v_render_alphas);
    }
    const int tw = gps_div_up(width, 16), th = gps_div_up(height, 16);
    // (experiment, gps_set_frame_chain_reserve bit 1: 14 KB of unused dynamic LDS on top of the 26.7 KB the kernel declares -> 3
    // workgroups of 8 waves per compute unit instead of the 4 that fill every wave slot)
    const size_t pad = (gps::frame_chain_reserve_bits() & 2) ? 14 * 1024 : 0;
    // the persistent launch (round 6) wherever its static dealing applies; gps_set_raster_fwd_persistent(0): the per-tile kernel
    const int n_wg = fwd_persistent_workgroups();
    if (FWD_TRIPS == 1 && N > 0 && n_wg > 0 && tw * th > n_wg / 2 && gps_div_up(tw * th, n_wg) <= PP_MAX_PASSES && pad == 0)
        launch_kernel(TK_RASTER_FWD, compose ? 1 : 0, raster_ges_fwd_pp_kernel, dim3(n_wg), dim3(FWD_THREADS), 0, (hipStream_t)stream,
                      (const float4*)records, ref_depth_map, width, height, tw, th, tile_offsets, flatten_ids, counts, delta_depth,
                      (float4*)render_colors, render_alphas, fc, tile_order);
    else
        launch_kernel(TK_RASTER_FWD, compose ? 1 : 0, raster_ges_fwd_pk_kernel, dim3(tw * th), dim3(FWD_THREADS), pad, (hipStream_t)stream,
                      (const float4*)records, ref_depth_map, width, height, tw, th, tile_offsets, flatten_ids, counts, delta_depth,
                      (float4*)render_colors, render_alphas, fc, tile_order);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int raster_ges_bwd_gs_launch(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                             const int32_t* radii, const float* ref_depth_map, int width, int height,
                             const int32_t* group_gs_ids, const int32_t* group_starts, const int64_t* counts, float delta_depth,
                             const float* v_render_colors, const float* v_render_alphas, float* v_means2d, float* v_conics,
                             float* v_colors, float* v_opacities, int zero_mode, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0);
    if (N == 0) return GPS_OK;
    GPS_REQUIRE(means2d && conics && colors && opacities && radii && ref_depth_map && group_gs_ids && group_starts &&
                counts && v_render_colors && v_render_alphas && v_means2d && v_conics && v_colors && v_opacities);
    hipStream_t s = (hipStream_t)stream;
    if (zero_mode == 0)
        zero_grads_kernel<<<min(2048, gps_div_up(4 * (int64_t)N, 256)), 256, 0, s>>>(N, v_means2d, v_conics, v_colors,
                                                                                     v_opacities);
#ifndef GPS_BWD_BLOCKS
#define GPS_BWD_BLOCKS 4096
#endif
    constexpr int bwd_blocks = GPS_BWD_BLOCKS;  // multiple of 8 (one contiguous task range per XCD), 16 workgroups per CU
    BwdPixArgs src = {ref_depth_map, (const float4*)v_render_colors, v_render_alphas, delta_depth};
    // plain stores where a half-wave owns a Gaussian: only if the buffers are known to be zero (filled here or by the caller)
    raster_ges_bwd_gs_kernel<<<bwd_blocks, 256, 0, s>>>(group_gs_ids, group_starts, (const float2*)means2d, conics,
                                                        (const float4*)colors, opacities, radii, src, counts, width, height,
                                                        v_means2d, v_conics, v_colors, v_opacities, zero_mode == 1 ? 0 : 1);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // namespace gps

extern "C" {

int gps_raster_ges_fwd(int N, const float* means2d, const float* conics, const float* colors, const float* opacities,
                       const float* ref_depth_map, int width, int height, int tile_size, const int32_t* tile_offsets,
                       const int32_t* flatten_ids, const int64_t* counts, float delta_depth, float* render_colors,
                       float* render_alphas, int32_t* last_ids, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && width > 0 && height > 0);
    GPS_REQUIRE(tile_size == 16);  // every shipped config uses 16 (raw_gs_model.h); other sizes are rejected loudly
    GPS_REQUIRE(ref_depth_map && tile_offsets && flatten_ids && counts && render_colors && render_alphas);
    GPS_REQUIRE(N == 0 || (means2d && conics && colors && opacities));
    const int tw = gps_div_up(width, tile_size), th = gps_div_up(height, tile_size);
    raster_ges_fwd_kernel<16><<<tw * th, TILE_THREADS, 0, (hipStream_t)stream>>>(
        (const float2*)means2d, conics, (const float4*)colors, opacities, ref_depth_map, width, height, tw, th,
        tile_offsets, flatten_ids, counts, delta_depth, (float4*)render_colors, render_alphas, last_ids);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

void gps_set_raster_fwd_persistent(int on) { __atomic_store_n(&g_fwd_persistent, on ? 1 : 0, __ATOMIC_RELAXED); }

int gps_raster_ges_fwd_rec(int N, const float* records, const float* ref_depth_map, int width, int height,
                           const int32_t* tile_offsets, const int32_t* flatten_ids, const int64_t* counts,
                           float delta_depth, float* render_colors, float* render_alphas, gps_stream stream) {
    return gps::raster_ges_fwd_rec_launch(N, records, ref_depth_map, width, height, tile_offsets, flatten_ids, counts,
                                          delta_depth, render_colors, render_alphas, nullptr, stream);
}

int gps_raster_ges_fwd_rec_ordered(int N, const float* records, const float* ref_depth_map, int width, int height,
                                   const int32_t* tile_offsets, const int32_t* flatten_ids, const int64_t* counts,
                                   float delta_depth, float* render_colors, float* render_alphas, const int32_t* tile_order,
                                   gps_stream stream) {
    return gps::raster_ges_fwd_rec_launch(N, records, ref_depth_map, width, height, tile_offsets, flatten_ids, counts,
                                          delta_depth, render_colors, render_alphas, nullptr, stream, tile_order);
}

int gps_raster_ges_bwd_gs(int N, const float* means2d, const float* conics, const float* colors,
                          const float* opacities, const int32_t* radii, const float* ref_depth_map, int width,
                          int height, const int32_t* group_gs_ids, const int32_t* group_starts, const int64_t* counts,
                          float delta_depth, const float* v_render_colors, const float* v_render_alphas,
                          float* v_means2d, float* v_conics, float* v_colors, float* v_opacities, int accumulate,
                          gps_stream stream) {
    return gps::raster_ges_bwd_gs_launch(N, means2d, conics, colors, opacities, radii, ref_depth_map, width, height,
                                         group_gs_ids, group_starts, counts, delta_depth, v_render_colors, v_render_alphas,
                                         v_means2d, v_conics, v_colors, v_opacities, accumulate ? 1 : 0, stream);
}

}  // extern "C"

// build-time tunables defined inside functions above (gps_build_flags)
GPS_TUNABLE_REPORT(GPS_BWD_BLOCKS, 4096);
