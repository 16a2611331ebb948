// Tile binning without a depth key, entirely on the device.
//
// gps_isect_tiles_no_depth <- gsplat::isect_tiles_tensor_no_depth +
//                             isect_offset_encode_tensor_no_depth
//                             (gsplat/rasterizer/isect_tiles_no_depth.cu:132-461)
//
// The reference does: count kernel -> torch cumsum x2 -> two .item() host syncs ->
// fill kernel -> cub radix sort of (int64 key, int32 value) -> offset kernel.
// Here n_isects / n_groups never leave the device (counts[]), buffers have a
// caller-chosen capacity, and the pipeline is
//   count (+ per-256 block sums) -> single-workgroup scan of block sums ->
//   expand (balanced, coalesced: one thread per OUTPUT element, binary search
//   in the block's LDS prefix) -> ONE stable counting-sort pass on the whole tile id
//   (up to 4,096 tiles: a per-(tile, 2048-item block) count table, one workgroup per
//   tile scans its row, the scatter ranks 11/12-bit digits with wave ballots; the tile
//   offsets are the exclusive scan of the tile totals, so no pass over the sorted keys
//   is needed).  More tiles than that: 2 LSD passes of <= 8 bits + an offsets pass.
// Sort keys are u32 tile ids; the int64 isect_ids of the reference API are
// materialised only if the caller asks for them.  The sort is stable, so inside
// a tile Gaussians stay in ascending index order exactly like cub's LSD sort.
#include "splat_bin.hpp"

namespace {

using gps::BIN_BLOCK;                   // Gaussians per count/expand workgroup
using gps::TileBox;
using gps::tile_bbox;
constexpr int SORT_THREADS = 256;       // 4 waves
constexpr int SORT_ITEMS = 8;           // items per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 2048 items per workgroup
constexpr int SCAN_THREADS = 1024;

// ---- workgroup-wide exclusive scan of one int per thread (blockDim multiple of 64, <= 1024) ----
__device__ __forceinline__ int block_excl_scan(int v, int* lds_wave_sums /*[17]*/, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    int incl = wave_incl_scan_i(v);
    if (lane == 63) lds_wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int s = lane < nwaves ? lds_wave_sums[lane] : 0;
        int si = wave_incl_scan_i(s);
        if (lane < nwaves) lds_wave_sums[lane] = si - s;
        if (lane == 63) lds_wave_sums[16] = si;
    }
    __syncthreads();
    int r = lds_wave_sums[wave] + incl - v;
    total = lds_wave_sums[16];
    __syncthreads();
    return r;
}

// ---------------- pass 1: per-Gaussian counts + per-block sums ----------------
__global__ __launch_bounds__(BIN_BLOCK) void count_kernel(int N, const float* __restrict__ means2d,
                                                         const int32_t* __restrict__ radii, int tile_size, int tw,
                                                         int th, int32_t* __restrict__ tiles_per_gauss,
                                                         int32_t* __restrict__ groups_per_gauss,
                                                         int32_t* __restrict__ blk_tiles,
                                                         int32_t* __restrict__ blk_groups,
                                                         int32_t* __restrict__ blk_vis,
                                                         const uint32_t* __restrict__ order,
                                                         int32_t* __restrict__ tiles_by_rank) {
    // order != NULL (depth-keyed binning of the `raw` method): thread i handles the Gaussian of depth rank i; its tile
    // count also goes to tiles_by_rank[i], which is what the expansion scans; no pixel groups are produced
    int i = blockIdx.x * BIN_BLOCK + threadIdx.x;
    int t = 0, g = 0, vis = 0;
    if (i < N) {
        const int gi = order ? (int)order[i] : i;
        int r = radii[gi];
        if (r > 0) {
            float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)gi);
            gps::tile_group_count(m.x, m.y, r, tile_size, tw, th, t, g);
            if (order) g = 0;
            vis = 1;
        }
        tiles_per_gauss[gi] = t;
        if (order) tiles_by_rank[i] = t;
        groups_per_gauss[i] = g;
    }
    gps::BinCountOut o = {tiles_per_gauss, groups_per_gauss, blk_tiles, blk_groups, blk_vis, tile_size, tw, th, {}};
    gps::bin_block_sums(o, t, g, vis);
}

// ---------------- single workgroup: exclusive scan of the block sums, totals -> counts ----------------
__global__ __launch_bounds__(SCAN_THREADS) void scan_blocks_kernel(int nblk, int32_t* __restrict__ blk_tiles,
                                                                  int32_t* __restrict__ blk_groups,
                                                                  const int32_t* __restrict__ blk_vis,
                                                                  int64_t isect_cap, int64_t group_cap,
                                                                  int64_t* __restrict__ counts) {
    __shared__ int ws[17];
    const int per = (nblk + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = threadIdx.x * per, hi = min(nblk, lo + per);
    int totals[2];
    for (int which = 0; which < 2; which++) {
        int32_t* a = which == 0 ? blk_tiles : blk_groups;
        int s = 0;
        for (int k = lo; k < hi; k++) s += a[k];
        int total;
        int base = block_excl_scan(s, ws, total);
        for (int k = lo; k < hi; k++) { int v = a[k]; a[k] = base; base += v; }
        totals[which] = total;
    }
    int s = 0;
    for (int k = lo; k < hi; k++) s += blk_vis[k];
    int vis_total;
    block_excl_scan(s, ws, vis_total);
    if (threadIdx.x == 0) {
        int64_t ni = totals[0], ng = totals[1];
        int64_t ovf = 0;
        if (ni > isect_cap) { ni = isect_cap; ovf = 1; }
        if (ng > group_cap) { ng = group_cap; ovf = 1; }
        // the overflow word is STICKY (only ever raised here): a host that looks at it once per keyframe still sees an
        // overflow of any launch in between; the caller zeroes it
        counts[0] = ni; counts[1] = ng; if (ovf) counts[2] = 1; counts[3] = vis_total;
    }
}

// ---------------- pass 2: expand (Gaussian, tile) pairs and the 32-pixel group table ----------------
__global__ __launch_bounds__(BIN_BLOCK) void expand_kernel(int N, const float* __restrict__ means2d,
                                                          const int32_t* __restrict__ radii, int tile_size, int tw,
                                                          int th, const int32_t* __restrict__ tiles_per_gauss,
                                                          const int32_t* __restrict__ groups_per_gauss,
                                                          const int32_t* __restrict__ blk_tiles,
                                                          const int32_t* __restrict__ blk_groups,
                                                          const int32_t* __restrict__ blk_vis, int64_t* __restrict__ counts,
                                                          int64_t isect_cap, int64_t group_cap, uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals,
                                                          int32_t* __restrict__ group_gs_ids,
                                                          int32_t* __restrict__ group_starts,
                                                          const uint32_t* __restrict__ order) {
    // order != NULL: thread i expands the Gaussian of depth rank i (tiles_per_gauss is then indexed by rank)
    // blk_tiles / blk_groups / blk_vis hold the per-block COUNTS of the count pass: every workgroup sums the blocks in front of
    // it itself (<= a few thousand ints, one coalesced read) instead of a single-workgroup scan launch in between, and the last
    // workgroup publishes the totals (counts[0..3], clamped to the capacities as before)
    __shared__ int ws[17];
    __shared__ int pre_t[BIN_BLOCK + 1];
    __shared__ int pre_g[BIN_BLOCK + 1];
    __shared__ uint32_t box_x0[BIN_BLOCK], box_y0[BIN_BLOCK], box_w[BIN_BLOCK], box_id[BIN_BLOCK];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * BIN_BLOCK + tid;
    int t = 0, g = 0;
    if (i < N) {
        const int gi = order ? (int)order[i] : i;
        box_id[tid] = (uint32_t)gi;
        t = tiles_per_gauss[i];
        g = groups_per_gauss[i];
        if (t > 0) {
            float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)gi);
            TileBox b = tile_bbox(m.x, m.y, radii[gi], tile_size, tw, th);
            box_x0[tid] = b.x0; box_y0[tid] = b.y0; box_w[tid] = b.x1 - b.x0;
        }
    }
    int tot_t, tot_g;
    int et = block_excl_scan(t, ws, tot_t);
    int eg = block_excl_scan(g, ws, tot_g);
    pre_t[tid] = et; pre_g[tid] = eg;
    if (tid == 0) { pre_t[BIN_BLOCK] = tot_t; pre_g[BIN_BLOCK] = tot_g; }
    __syncthreads();
    int pt = 0, pg = 0, pv = 0;
    for (int k = tid; k < (int)blockIdx.x; k += BIN_BLOCK) { pt += blk_tiles[k]; pg += blk_groups[k]; pv += blk_vis[k]; }
    int sum_t, sum_g, sum_v;
    block_excl_scan(pt, ws, sum_t);
    block_excl_scan(pg, ws, sum_g);
    block_excl_scan(pv, ws, sum_v);
    const int64_t base_t = sum_t, base_g = sum_g;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        int64_t ni = base_t + tot_t, ng = base_g + tot_g;
        int64_t ovf = 0;
        if (ni > isect_cap) { ni = isect_cap; ovf = 1; }
        if (ng > group_cap) { ng = group_cap; ovf = 1; }
        // the overflow word is STICKY (only ever raised here): a host that looks at it once per keyframe still sees an
        // overflow of any launch in between; the caller zeroes it
        counts[0] = ni; counts[1] = ng; if (ovf) counts[2] = 1; counts[3] = sum_v + blk_vis[blockIdx.x];
    }
    // one thread per output intersection
    for (int j = tid; j < tot_t; j += BIN_BLOCK) {
        int lo = 0, hi = BIN_BLOCK;  // largest src with pre_t[src] <= j
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (pre_t[mid] <= j) lo = mid; else hi = mid; }
        int k = j - pre_t[lo];
        uint32_t w = box_w[lo];
        uint32_t ty = box_y0[lo] + (uint32_t)k / w, tx = box_x0[lo] + (uint32_t)k % w;
        int64_t o = base_t + j;
        if (o < isect_cap) { keys[o] = ty * (uint32_t)tw + tx; vals[o] = box_id[lo]; }
    }
    // one thread per output group
    for (int j = tid; j < tot_g; j += BIN_BLOCK) {
        int lo = 0, hi = BIN_BLOCK;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (pre_g[mid] <= j) lo = mid; else hi = mid; }
        int64_t o = base_g + j;
        if (o < group_cap) {
            group_gs_ids[o] = blockIdx.x * BIN_BLOCK + lo;
            group_starts[o] = (int32_t)(base_g + pre_g[lo]);
        }
    }
}

// ---------------- stable LSD radix sort pass on `bits` bits starting at `shift` ----------------
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys,
                                                                 const int64_t* __restrict__ counts, int shift,
                                                                 int bits, int nblk_cap, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    const int n = (int)counts[0];
    const int base = blockIdx.x * SORT_TILE;
    if (base >= n) return;
    const int bins = 1 << bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    if (threadIdx.x < bins) h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        int idx = base + k * SORT_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < bins) hist[(size_t)threadIdx.x * nblk_cap + blockIdx.x] = h[threadIdx.x];
}

// One workgroup per digit: exclusive scan of that digit's per-block counts (row d of hist) and the digit total.
// The cross-digit base is added by the scatter kernel (it scans the <= 256 totals in LDS), so no single-workgroup
// serial scan over bins x blocks is needed.
__global__ __launch_bounds__(256) void radix_scan_kernel(const int64_t* __restrict__ counts, int nblk_cap,
                                                        uint32_t* __restrict__ hist, uint32_t* __restrict__ digit_total) {
    __shared__ int ws[17];
    const int n = (int)counts[0];
    const int nblk = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t* row = hist + (size_t)blockIdx.x * nblk_cap;
    const int per = (nblk + 255) / 256;
    const int lo = min(nblk, (int)threadIdx.x * per), hi = min(nblk, lo + per);
    int s = 0;
    for (int b = lo; b < hi; b++) s += (int)row[b];
    int tot;
    int run = block_excl_scan(s, ws, tot);
    for (int b = lo; b < hi; b++) { int v = (int)row[b]; row[b] = (uint32_t)run; run += v; }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = (uint32_t)tot;
}

// Each wave owns a contiguous 512-item slice; inside it items are visited in index order
// (iteration-major, lane-minor) so "earlier item, same digit" == stable rank.
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    const int64_t* __restrict__ counts, int shift,
                                                                    int bits, int nblk_cap,
                                                                    const uint32_t* __restrict__ hist,
                                                                    const uint32_t* __restrict__ digit_total,
                                                                    uint32_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t wavecnt[SORT_THREADS / 64][256];
    __shared__ uint32_t digitbase[256];
    const int n = (int)counts[0];
    const int base = blockIdx.x * SORT_TILE;
    if (base >= n) return;
    const int bins = 1 << bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < (SORT_THREADS / 64) * 256; k += SORT_THREADS) (&wavecnt[0][0])[k] = 0;
    __syncthreads();
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
    const int wbase = base + wave * (64 * SORT_ITEMS);
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        int idx = wbase + k * 64 + lane;
        bool valid = idx < n;
        key[k] = valid ? keys_in[idx] : 0u;
        val[k] = valid ? vals_in[idx] : 0u;
        uint32_t d = (key[k] >> shift) & mask;
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < bits; b++) {
            unsigned long long bal = __ballot(valid && ((d >> b) & 1u));
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t prev = wavecnt[wave][d];
        rank[k] = prev + (uint32_t)__popcll(same & lt);
        // highest lane of each digit group publishes the new running count
        if (valid && (same >> lane) == 1ull) wavecnt[wave][d] = prev + (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // exclusive scan of the digit totals (bins <= 256 == blockDim.x): every workgroup recomputes it, it is 256 adds
    __shared__ int dws[17];
    int dtot;
    const int dval = (int)threadIdx.x < bins ? (int)digit_total[threadIdx.x] : 0;
    const int dexcl = block_excl_scan(dval, dws, dtot);
    if ((int)threadIdx.x < bins) {
        uint32_t run = (uint32_t)dexcl + hist[(size_t)threadIdx.x * nblk_cap + blockIdx.x];
        digitbase[threadIdx.x] = run;
        uint32_t acc = 0;
        for (int w = 0; w < SORT_THREADS / 64; w++) { uint32_t c = wavecnt[w][threadIdx.x]; wavecnt[w][threadIdx.x] = acc; acc += c; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        int idx = wbase + k * 64 + lane;
        if (idx < n) {
            uint32_t d = (key[k] >> shift) & mask;
            uint32_t pos = digitbase[d] + wavecnt[wave][d] + rank[k];
            keys_out[pos] = key[k];
            vals_out[pos] = val[k];
        }
    }
}

// ---------------- one-pass stable counting sort on the whole tile id (n_tiles <= WIDE_MAX_BINS) ----------------
// The binning of an optimise iteration is launch-latency bound (10 launches of 5-12 us for 4 MB of keys), so the two 8-bit
// LSD passes + the offsets pass (7 launches) are replaced by 3: count table, row scan, scatter.  Same output order: a
// counting sort on the full key is the stable sort.
constexpr int WIDE_MAX_BINS = 4096;

__global__ __launch_bounds__(SORT_THREADS) void wide_hist_kernel(const uint32_t* __restrict__ keys,
                                                                const int64_t* __restrict__ counts, int bins,
                                                                int nblk_cap, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t wh[];  // [bins]
    const int n = (int)counts[0];
    const int base = blockIdx.x * SORT_TILE;
    if (base >= n) return;
    for (int b = threadIdx.x; b < bins; b += SORT_THREADS) wh[b] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        int idx = base + k * SORT_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&wh[keys[idx]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += SORT_THREADS) hist[(size_t)b * nblk_cap + blockIdx.x] = wh[b];
}

// Ranking as in radix_scatter_kernel with `bits`-wide digits; per-wave running counts are 16-bit (a wave owns 512 items).
// Workgroup 0 also writes the tile offsets = exclusive scan of the tile totals (isect_tiles_no_depth.cu:373-425: first
// index of each tile, n for the tiles behind the last one).
__global__ __launch_bounds__(SORT_THREADS) void wide_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   const int64_t* __restrict__ counts, int bins, int bits,
                                                                   int nblk_cap, const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ digit_total,
                                                                   uint32_t* __restrict__ keys_out /* may be NULL */,
                                                                   uint32_t* __restrict__ vals_out,
                                                                   int32_t* __restrict__ tile_offsets) {
    extern __shared__ uint32_t wlds[];
    uint32_t* digitbase = wlds;                                               // [bins]
    uint16_t* wavecnt = reinterpret_cast<uint16_t*>(wlds + bins);             // [4][bins]
    __shared__ int dws[17];
    const int n = (int)counts[0];
    const int base = blockIdx.x * SORT_TILE;
    if (base >= n && blockIdx.x != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // exclusive scan of the tile totals: thread t owns `per` consecutive tiles
    const int per = (bins + SORT_THREADS - 1) / SORT_THREADS;
    const int lo = min(bins, (int)threadIdx.x * per), hi = min(bins, lo + per);
    int sum = 0;
    if (n > 0)
        for (int b = lo; b < hi; b++) sum += (int)digit_total[b];
    int dtot;
    int run = block_excl_scan(sum, dws, dtot);
    for (int b = lo; b < hi; b++) {
        digitbase[b] = (uint32_t)run;
        if (blockIdx.x == 0) tile_offsets[b] = run;
        if (n > 0) run += (int)digit_total[b];
    }
    if (base >= n) return;  // (workgroup 0 of an empty launch: offsets only)
    for (int k = threadIdx.x; k < (SORT_THREADS / 64) * bins; k += SORT_THREADS) wavecnt[k] = 0;
    __syncthreads();
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
    const int wbase = base + wave * (64 * SORT_ITEMS);
    const unsigned long long lt = lanemask_lt();
    uint16_t* mycnt = wavecnt + wave * bins;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        int idx = wbase + k * 64 + lane;
        bool valid = idx < n;
        key[k] = valid ? keys_in[idx] : 0u;
        val[k] = valid ? vals_in[idx] : 0u;
        const uint32_t d = key[k];
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < bits; b++) {
            unsigned long long bal = __ballot(valid && ((d >> b) & 1u));
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t prev = mycnt[d];
        rank[k] = prev + (uint32_t)__popcll(same & lt);
        if (valid && (same >> lane) == 1ull) mycnt[d] = (uint16_t)(prev + (uint32_t)__popcll(same));
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += SORT_THREADS) {
        digitbase[b] += hist[(size_t)b * nblk_cap + blockIdx.x];
        uint32_t acc = 0;
        for (int w = 0; w < SORT_THREADS / 64; w++) { uint32_t c = wavecnt[w * bins + b]; wavecnt[w * bins + b] = (uint16_t)acc; acc += c; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        int idx = wbase + k * 64 + lane;
        if (idx < n) {
            const uint32_t d = key[k];
            uint32_t pos = digitbase[d] + mycnt[d] + rank[k];
            if (keys_out) keys_out[pos] = key[k];
            vals_out[pos] = val[k];
        }
    }
}

// ---------------- tile offsets (+ optional int64 copy of the sorted keys) ----------------
// isect_tiles_no_depth.cu:373-425
__global__ __launch_bounds__(256) void offsets_kernel(const uint32_t* __restrict__ keys,
                                                     const int64_t* __restrict__ counts, int n_tiles,
                                                     int32_t* __restrict__ offsets, int64_t* __restrict__ isect_ids,
                                                     const float* __restrict__ depths,
                                                     const int32_t* __restrict__ flatten_ids) {
    const int n = (int)counts[0];
    const int stride = gridDim.x * blockDim.x;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) {
        for (int t = gid; t < n_tiles; t += stride) offsets[t] = 0;
        return;
    }
    for (int idx = gid; idx < n; idx += stride) {
        int cur = (int)keys[idx];
        if (isect_ids)  // isect_tiles.cu:98-109: tile id in the high word, the depth's bit pattern in the low word
            isect_ids[idx] = depths ? (((int64_t)cur << 32) | (int64_t)(uint32_t)__float_as_int(depths[flatten_ids[idx]]))
                                    : (int64_t)cur;
        int prev = idx > 0 ? (int)keys[idx - 1] : -1;
        for (int t = prev + 1; t <= cur; t++) offsets[t] = idx;
        if (idx == n - 1)
            for (int t = cur + 1; t < n_tiles; t++) offsets[t] = n;
    }
}

// ---------------- depth mode helpers (isect_tiles.cu: key = tile << 32 | bits(depth)) ----------------
// Sorting 64-bit (tile, depth) keys over all I intersections would take 6 radix passes over I items.  Equivalent and much
// cheaper: sort the N Gaussians by depth once (4 passes over N << I items, invisible ones last), expand them in that order,
// then the usual stable sort by tile id (2 passes over I) leaves every tile's list in (depth, index) order.
__global__ __launch_bounds__(256) void depth_keys_kernel(int N, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        int64_t* __restrict__ count_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { count_n[0] = N; count_n[1] = 0; count_n[2] = 0; count_n[3] = 0; }
    if (i >= N) return;
    keys[i] = radii[i] > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;  // visible depths are positive floats
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void copy_u32_kernel(int n, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

struct Workspace {
    int32_t *groups_per_gauss, *blk_tiles, *blk_groups, *blk_vis;
    uint32_t *keys_a, *vals_a, *keys_b, *vals_b, *hist, *digit_total;
    uint32_t* order;         // depth mode: Gaussian id by depth rank
    int32_t* tiles_by_rank;  // depth mode: tile count by depth rank
    int64_t* count_n;        // depth mode: {N, 0, 0, 0} for the radix kernels
    int32_t* dummy_groups;
    char* sb_region;         // tables of the superblock binning (splat_bin_sb.hip)
    int nblkN, nblkI;
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t carve(Workspace* w, char* base, int N, int64_t cap) {
    size_t off = 0;
    int nblkN = gps_div_up(N > 0 ? N : 1, BIN_BLOCK);
    int nblkI = gps_div_up(cap > 0 ? cap : 1, SORT_TILE);
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return base ? base + o : nullptr; };
    char* p;
    // FIRST, at an offset that does not depend on N: the superblock binning's tables are state that lives across launches
    // (zero between them) while N changes with every add / prune
    p = take(gps::sb_tables_bytes()); if (w) w->sb_region = p;
    p = take((size_t)(N > 0 ? N : 1) * 4); if (w) w->groups_per_gauss = (int32_t*)p;
    p = take((size_t)nblkN * 4); if (w) w->blk_tiles = (int32_t*)p;
    p = take((size_t)nblkN * 4); if (w) w->blk_groups = (int32_t*)p;
    p = take((size_t)nblkN * 4); if (w) w->blk_vis = (int32_t*)p;
    p = take((size_t)cap * 4); if (w) w->keys_a = (uint32_t*)p;
    p = take((size_t)cap * 4); if (w) w->vals_a = (uint32_t*)p;
    p = take((size_t)cap * 4); if (w) w->keys_b = (uint32_t*)p;
    p = take((size_t)cap * 4); if (w) w->vals_b = (uint32_t*)p;
    p = take((size_t)WIDE_MAX_BINS * nblkI * 4); if (w) w->hist = (uint32_t*)p;   // [tile][2048-item block] count table
    p = take((size_t)WIDE_MAX_BINS * 4); if (w) w->digit_total = (uint32_t*)p;
    p = take((size_t)(N > 0 ? N : 1) * 4); if (w) w->order = (uint32_t*)p;
    p = take((size_t)(N > 0 ? N : 1) * 4); if (w) w->tiles_by_rank = (int32_t*)p;
    p = take(4 * sizeof(int64_t)); if (w) w->count_n = (int64_t*)p;
    p = take(256); if (w) w->dummy_groups = (int32_t*)p;
    if (w) { w->nblkN = nblkN; w->nblkI = nblkI; }
    return off;
}

}  // namespace

static int isect_impl(int N, const float* means2d, const int32_t* radii, const float* depths, int tile_size, int tile_width,
                      int tile_height, int64_t isect_capacity, int64_t group_capacity, int32_t* tiles_per_gauss,
                      int64_t* isect_ids, int32_t* flatten_ids, int32_t* group_gs_ids, int32_t* group_starts,
                      int32_t* tile_offsets, int64_t* counts, void* workspace, int64_t workspace_bytes, gps_stream stream,
                      bool counted);

namespace gps {

int isect_count_targets(int N, int64_t isect_capacity, int32_t* tiles_per_gauss, int tile_size, int tile_width, int tile_height,
                        void* workspace, int64_t workspace_bytes, bool superblock, BinCountOut* out) {
    GPS_REQUIRE(N >= 0 && isect_capacity > 0 && workspace && out && tiles_per_gauss);
    if (workspace_bytes < (int64_t)carve(nullptr, nullptr, N, isect_capacity)) return GPS_ERR_CAPACITY;
    Workspace w;
    carve(&w, (char*)workspace, N, isect_capacity);
    *out = {tiles_per_gauss, w.groups_per_gauss, w.blk_tiles, w.blk_groups, w.blk_vis, tile_size, tile_width, tile_height, {}};
    if (superblock) {
        GPS_REQUIRE(sb_supported(N, tile_width, tile_height));
        sb_tables_carve(w.sb_region, &out->sb);
        out->sb.sb_shift = sb_shift_for(N);
    }
    return GPS_OK;
}

int isect_workspace_tables(void* workspace, int64_t workspace_bytes, SbTables* t) {
    GPS_REQUIRE(workspace && t && workspace_bytes >= (int64_t)align_up(gps::sb_tables_bytes()));
    Workspace w;
    carve(&w, (char*)workspace, 0, 1);   // (the tables sit first, at an offset that does not depend on N)
    sb_tables_carve(w.sb_region, t);
    return GPS_OK;
}

int isect_tiles_no_depth_counted(int N, const float* means2d, const int32_t* radii, int tile_size, int tile_width,
                                 int tile_height, int64_t isect_capacity, int64_t group_capacity, int32_t* tiles_per_gauss,
                                 int32_t* flatten_ids, int32_t* group_gs_ids, int32_t* group_starts, int32_t* tile_offsets,
                                 int64_t* counts, void* workspace, int64_t workspace_bytes, gps_stream stream) {
    GPS_REQUIRE(group_gs_ids && group_starts);
    return isect_impl(N, means2d, radii, nullptr, tile_size, tile_width, tile_height, isect_capacity, group_capacity,
                      tiles_per_gauss, nullptr, flatten_ids, group_gs_ids, group_starts, tile_offsets, counts, workspace,
                      workspace_bytes, stream, true);
}

}  // namespace gps

extern "C" {

int64_t gps_isect_workspace_bytes(int N, int64_t isect_capacity) {
    if (N < 0 || isect_capacity < 0) return GPS_ERR_ARG;
    return (int64_t)carve(nullptr, nullptr, N, isect_capacity);
}

const int32_t* gps_isect_workspace_tile_order(void* workspace, int N, int64_t isect_capacity) {
    if (!workspace || N < 0 || isect_capacity <= 0) return nullptr;
    Workspace w;
    carve(&w, (char*)workspace, N, isect_capacity);
    gps::SbTables t;
    gps::sb_tables_carve(w.sb_region, &t);
    return t.tile_order;
}

int gps_isect_workspace_init(void* workspace, int64_t workspace_bytes, gps_stream stream) {
    GPS_REQUIRE(workspace && workspace_bytes > 0);
    return hipMemsetAsync(workspace, 0, (size_t)workspace_bytes, (hipStream_t)stream) == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}

}  // extern "C"

static int isect_impl(int N, const float* means2d, const int32_t* radii, const float* depths /* NULL: no-depth variant */,
                      int tile_size, int tile_width, int tile_height, int64_t isect_capacity, int64_t group_capacity,
                      int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids, int32_t* group_gs_ids,
                      int32_t* group_starts, int32_t* tile_offsets, int64_t* counts, void* workspace, int64_t workspace_bytes,
                      gps_stream stream, bool counted /* first pass already written (gps::isect_count_targets) */) {
    GPS_ENTER();
    GPS_REQUIRE(N >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0);
    GPS_REQUIRE(isect_capacity > 0 && isect_capacity < (1ll << 31) && group_capacity > 0 && group_capacity < (1ll << 31));
    GPS_REQUIRE(flatten_ids && tile_offsets && counts && workspace);
    GPS_REQUIRE(depths || (group_gs_ids && group_starts));
    GPS_REQUIRE(N == 0 || (means2d && radii && tiles_per_gauss));
    const int n_tiles = tile_width * tile_height;
    GPS_REQUIRE(n_tiles <= (1 << 16));
    if (workspace_bytes < gps_isect_workspace_bytes(N, isect_capacity)) return GPS_ERR_CAPACITY;
    GPS_REQUIRE(!depths || N <= isect_capacity);  // the Gaussian sort borrows the intersection buffers
    Workspace w;
    carve(&w, (char*)workspace, N, isect_capacity);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t* order = nullptr;
    if (depths && N > 0) {
        depth_keys_kernel<<<gps_div_up(N, 256), 256, 0, s>>>(N, radii, depths, w.keys_a, w.vals_a, w.count_n);
        uint32_t *ka = w.keys_a, *va = w.vals_a, *kb = w.keys_b, *vb = w.vals_b;
        const int nblkN_sort = gps_div_up(N, SORT_TILE);
        for (int pass = 0; pass < 4; pass++) {
            radix_hist_kernel<<<nblkN_sort, SORT_THREADS, 0, s>>>(ka, w.count_n, 8 * pass, 8, w.nblkI, w.hist);
            radix_scan_kernel<<<256, 256, 0, s>>>(w.count_n, w.nblkI, w.hist, w.digit_total);
            radix_scatter_kernel<<<nblkN_sort, SORT_THREADS, 0, s>>>(ka, va, w.count_n, 8 * pass, 8, w.nblkI, w.hist,
                                                                    w.digit_total, kb, vb);
            uint32_t* t = ka; ka = kb; kb = t;
            t = va; va = vb; vb = t;
        }
        copy_u32_kernel<<<gps_div_up(N, 256), 256, 0, s>>>(N, va, w.order);  // (4 passes: va == w.vals_a again)
        order = w.order;
    }
    int32_t* tpg_scan = order ? w.tiles_by_rank : tiles_per_gauss;
    if (!group_gs_ids) { group_gs_ids = w.dummy_groups; group_starts = w.dummy_groups; }

    if (N > 0 && !counted)
        count_kernel<<<w.nblkN, BIN_BLOCK, 0, s>>>(N, means2d, radii, tile_size, tile_width, tile_height,
                                                   tiles_per_gauss, w.groups_per_gauss, w.blk_tiles, w.blk_groups,
                                                   w.blk_vis, order, w.tiles_by_rank);
    if (N > 0)  // (prefix over the blocks + totals inside: no scan launch)
        expand_kernel<<<w.nblkN, BIN_BLOCK, 0, s>>>(N, means2d, radii, tile_size, tile_width, tile_height,
                                                    tpg_scan, w.groups_per_gauss, w.blk_tiles, w.blk_groups, w.blk_vis, counts,
                                                    isect_capacity, group_capacity, w.keys_a, w.vals_a, group_gs_ids,
                                                    group_starts, order);
    else  // nothing to expand: the totals are zeros
        scan_blocks_kernel<<<1, SCAN_THREADS, 0, s>>>(0, w.blk_tiles, w.blk_groups, w.blk_vis, isect_capacity, group_capacity,
                                                      counts);
    int bits_total = 1;
    while ((1 << bits_total) < n_tiles) bits_total++;
    const uint32_t* sorted_keys = nullptr;
    if (n_tiles <= WIDE_MAX_BINS) {
        // one counting-sort pass on the whole tile id; the offsets come out of the scatter
        uint32_t* keys_out = isect_ids ? w.keys_b : nullptr;
        wide_hist_kernel<<<w.nblkI, SORT_THREADS, (size_t)n_tiles * 4, s>>>(w.keys_a, counts, n_tiles, w.nblkI, w.hist);
        radix_scan_kernel<<<n_tiles, 256, 0, s>>>(counts, w.nblkI, w.hist, w.digit_total);
        wide_scatter_kernel<<<w.nblkI, SORT_THREADS, (size_t)n_tiles * (4 + 2 * (SORT_THREADS / 64)), s>>>(
            w.keys_a, w.vals_a, counts, n_tiles, bits_total, w.nblkI, w.hist, w.digit_total, keys_out, (uint32_t*)flatten_ids,
            tile_offsets);
        sorted_keys = keys_out;
    } else {
        int b1 = (bits_total + 1) / 2, b2 = bits_total - b1;
        radix_hist_kernel<<<w.nblkI, SORT_THREADS, 0, s>>>(w.keys_a, counts, 0, b1, w.nblkI, w.hist);
        radix_scan_kernel<<<1 << b1, 256, 0, s>>>(counts, w.nblkI, w.hist, w.digit_total);
        radix_scatter_kernel<<<w.nblkI, SORT_THREADS, 0, s>>>(w.keys_a, w.vals_a, counts, 0, b1, w.nblkI,
                                                                     w.hist, w.digit_total, w.keys_b, w.vals_b);
        radix_hist_kernel<<<w.nblkI, SORT_THREADS, 0, s>>>(w.keys_b, counts, b1, b2, w.nblkI, w.hist);
        radix_scan_kernel<<<1 << b2, 256, 0, s>>>(counts, w.nblkI, w.hist, w.digit_total);
        radix_scatter_kernel<<<w.nblkI, SORT_THREADS, 0, s>>>(w.keys_b, w.vals_b, counts, b1, b2, w.nblkI,
                                                                    w.hist, w.digit_total, w.keys_a, (uint32_t*)flatten_ids);
        sorted_keys = w.keys_a;
    }
    // the pass over the sorted keys is only needed for the int64 isect_ids of the operator API (and for the offsets of the
    // two-pass path)
    if (sorted_keys)
        offsets_kernel<<<512, 256, 0, s>>>(sorted_keys, counts, n_tiles, tile_offsets, isect_ids, depths, flatten_ids);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

extern "C" {

int gps_isect_tiles_no_depth(int N, const float* means2d, const int32_t* radii, int tile_size, int tile_width,
                             int tile_height, int64_t isect_capacity, int64_t group_capacity,
                             int32_t* tiles_per_gauss, int64_t* isect_ids, int32_t* flatten_ids,
                             int32_t* group_gs_ids, int32_t* group_starts, int32_t* tile_offsets, int64_t* counts,
                             void* workspace, int64_t workspace_bytes, gps_stream stream) {
    GPS_REQUIRE(group_gs_ids && group_starts);
    return isect_impl(N, means2d, radii, nullptr, tile_size, tile_width, tile_height, isect_capacity, group_capacity,
                      tiles_per_gauss, isect_ids, flatten_ids, group_gs_ids, group_starts, tile_offsets, counts, workspace,
                      workspace_bytes, stream, false);
}

int gps_isect_tiles(int N, const float* means2d, const int32_t* radii, const float* depths, int tile_size, int tile_width,
                    int tile_height, int64_t isect_capacity, int32_t* tiles_per_gauss, int64_t* isect_ids,
                    int32_t* flatten_ids, int32_t* tile_offsets, int64_t* counts, void* workspace, int64_t workspace_bytes,
                    gps_stream stream) {
    GPS_REQUIRE(N == 0 || depths);
    static const float dummy_depth = 1.0f;
    return isect_impl(N, means2d, radii, depths ? depths : &dummy_depth, tile_size, tile_width, tile_height, isect_capacity,
                      1 << 20, tiles_per_gauss, isect_ids, flatten_ids, nullptr, nullptr, tile_offsets, counts, workspace,
                      workspace_bytes, stream, false);
}

}  // extern "C"
