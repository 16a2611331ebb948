// Superblock binning: the fused model path's tile binning as two launches behind the preprocessing kernel (see splat_bin.hpp).
//
//   <- gsplat::isect_tiles_tensor_no_depth + isect_offset_encode_tensor_no_depth
//      (gsplat/rasterizer/isect_tiles_no_depth.cu:132-461): count, cumsum, fill keys, cub radix sort by tile id, offsets.
//
// Same result -- per tile the Gaussian ids in ascending order, tile start offsets -- as a counting sort keyed on the tile id
// whose "blocks" are runs of consecutive GAUSSIANS instead of runs of intersections, so that no (key, value) array is ever
// written or read: the histogram pass runs inside the preprocessing kernel on the bounding boxes still in registers, and the
// scatter re-derives a superblock's pairs from the same boxes.
#include <mutex>

#include "launch_timing.hpp"
#include "splat_bin.hpp"

namespace {

using gps::BIN_BLOCK;
using gps::SB_MAX;
using gps::SB_MAX_TILES;
using gps::SbTables;
using gps::TileBox;
using gps::tile_bbox;

constexpr int SCAT_THREADS = 512;
constexpr int SCAT_WAVES = SCAT_THREADS / 64;
constexpr int SCAT_ITEMS = 4;                              // pairs per thread per chunk
constexpr int SCAT_CHUNK = SCAT_THREADS * SCAT_ITEMS;      // 1024 pairs per chunk

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_excl_scan_i(int v, int& total) {
    const int incl = wave_incl_scan_i(v);
    total = __shfl(incl, 63, 64);
    return incl - v;
}

// ---- one wave per tile: exclusive prefix of the tile's row of C over the superblocks; C is left zero ----
__global__ __launch_bounds__(256) void sb_scan_kernel(int n_tiles, SbTables t, gps::LaunchStamp stamp) {
    gps::StampScope timed(stamp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    static_assert(SB_MAX == 512, "a lane owns 8 consecutive superblocks");
    if (tile < n_tiles) {
        uint4* crow = reinterpret_cast<uint4*>(t.C + (size_t)tile * SB_MAX) + 2 * lane;
        const uint4 a = crow[0], b = crow[1];
        crow[0] = make_uint4(0, 0, 0, 0); crow[1] = make_uint4(0, 0, 0, 0);
        const int s = (int)(a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w);
        int total;
        uint32_t run = (uint32_t)wave_excl_scan_i(s, total);
        uint4 pa, pb;
        pa.x = run; run += a.x; pa.y = run; run += a.y; pa.z = run; run += a.z; pa.w = run; run += a.w;
        pb.x = run; run += b.x; pb.y = run; run += b.y; pb.z = run; run += b.z; pb.w = run;
        uint4* prow = reinterpret_cast<uint4*>(t.P + (size_t)tile * SB_MAX) + 2 * lane;
        prow[0] = pa; prow[1] = pb;
        if (lane == 0) t.tile_total[tile] = (uint32_t)total;
    }
    // the list-key counts (row k of cls_count = key k over the superblocks): the rows behind the last tile, one wave each
    const int k = tile - n_tiles;
    if (k >= 0 && k < gps::BWD_KEYS) {
        int4* crow = reinterpret_cast<int4*>(t.cls_count + (size_t)k * SB_MAX) + 2 * lane;
        const int4 a = crow[0], b = crow[1];
        crow[0] = make_int4(0, 0, 0, 0); crow[1] = make_int4(0, 0, 0, 0);
        const int s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        int total;
        int run = wave_excl_scan_i(s, total);
        int4 pa, pb;
        pa.x = run; run += a.x; pa.y = run; run += a.y; pa.z = run; run += a.z; pa.w = run; run += a.w;
        pb.x = run; run += b.x; pb.y = run; run += b.y; pb.z = run; run += b.z; pb.w = run;
        int4* prow = reinterpret_cast<int4*>(t.cls_prefix + (size_t)k * SB_MAX) + 2 * lane;
        prow[0] = pa; prow[1] = pb;
        if (lane == 0) t.cls_prefix[gps::BWD_KEYS * SB_MAX + k] = total;   // totals row (n_visible = their sum)
    }
}

// ---- one workgroup per superblock: stable scatter of its (Gaussian, tile) pairs + the backward's class lists ----
// Everything that is "per tile" runs over the band of tile rows the superblock's boxes touch (consecutive Gaussians cover
// neighbouring pixels, so the band is a few rows of the tile grid, not all of it); what concerns the whole grid -- tile_offsets,
// the counts, the forward rasterizer's launch order -- is the extra (last) workgroup's.
__global__ __launch_bounds__(SCAT_THREADS) void sb_scatter_kernel(
    int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const int32_t* __restrict__ tiles_per_gauss,
    int tile_size, int tw, int th, SbTables t, int64_t isect_cap, int32_t* __restrict__ flatten_ids,
    int32_t* __restrict__ tile_offsets, int64_t* __restrict__ counts, int32_t* __restrict__ cls_ids,
    int32_t* __restrict__ cls_counts, int64_t cls_stride, gps::LaunchStamp stamp) {
    gps::StampScope timed(stamp);
    extern __shared__ uint32_t lds[];
    const int n_tiles = tw * th;
    if (blockIdx.x == gridDim.x - 1) {
        // The extra workgroup does what concerns the WHOLE tile grid, so that no scattering workgroup has to (round 3 gave these
        // duties to superblock 0, whose "band" then was every tile: its per-tile loops and ballot ranks ran over 1,200 tiles
        // instead of ~100 and it finished last): tile_offsets = exclusive scan of the tile totals, the counts, and the forward
        // rasterizer's launch order = tiles by descending list length (a counting sort over 64 length classes; the order inside a
        // class does not matter) -- 1,200 tiles are 1.2 rounds of the 1,024 tile workgroups the chip holds: with the longest
        // lists first, the part-filled second round is the short ones.
        __shared__ int hist[64];
        __shared__ int gws[SCAT_WAVES];
        const int tid_ = threadIdx.x, lane_ = tid_ & 63, wave_ = tid_ >> 6;
        if (tid_ < 64) hist[tid_] = 0;
        const int per = (n_tiles + SCAT_THREADS - 1) / SCAT_THREADS;
        const int lo = min(n_tiles, tid_ * per), hi = min(n_tiles, lo + per);
        int sum = 0;
        for (int b = lo; b < hi; b++) sum += (int)t.tile_total[b];
        const int incl = wave_incl_scan_i(sum);
        if (lane_ == 63) gws[wave_] = incl;
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < SCAT_WAVES; w++) { const int v = gws[w]; if (w < wave_) woff += v; total += v; }
        int64_t run = woff + incl - sum;
        for (int b = lo; b < hi; b++) {
            const int tt = (int)t.tile_total[b];
            tile_offsets[b] = (int)min(run, isect_cap);
            atomicAdd(&hist[63 - min(63, tt >> 5)], 1);
            run += tt;
        }
        if (tid_ == 0) {
            int64_t ni = total;
            if (ni > isect_cap) { ni = isect_cap; counts[2] = 1; }   // sticky overflow word, as the sorted-key path
            counts[0] = ni; counts[1] = 0;
            int nv = 0;
            for (int k = 0; k < gps::BWD_CLASSES; k++) {
                int nk = 0;
                for (int b = 0; b < gps::BWD_BANDS; b++) nk += t.cls_prefix[gps::BWD_KEYS * SB_MAX + k * gps::BWD_BANDS + b];
                if (cls_counts) cls_counts[k] = nk;
                nv += nk;
            }
            counts[3] = nv;
        }
        __syncthreads();
        if (tid_ < 64) {
            int tot_;
            const int ex = wave_excl_scan_i(hist[tid_], tot_);
            hist[tid_] = ex;
        }
        __syncthreads();
        for (int b = tid_; b < n_tiles; b += SCAT_THREADS) t.tile_order[atomicAdd(&hist[63 - min(63, (int)(t.tile_total[b] >> 5))], 1)] = b;
        return;
    }
    const int sb = blockIdx.x, sb_size = BIN_BLOCK << t.sb_shift;        // Gaussians per superblock
    uint32_t* base = lds;                                                 // [n_tiles] absolute start of this superblock's run in a tile
    uint16_t* wavecnt = reinterpret_cast<uint16_t*>(lds + n_tiles);       // [SCAT_WAVES][n_tiles] per-wave running counts of a chunk
    uint16_t* tot16 = wavecnt + SCAT_WAVES * n_tiles;                     // [n_tiles] a chunk's pairs per tile
    uint32_t* pre = lds + n_tiles + ((SCAT_WAVES + 1) * n_tiles + 1) / 2; // [sb_size + 1] exclusive prefix of the tile counts
    uint32_t* box = pre + sb_size + 1;                                    // [sb_size] x0 | y0 << 12 | width << 24
    __shared__ int ws[SCAT_WAVES], wlo[SCAT_WAVES], whi[SCAT_WAVES];
    __shared__ int key_wave[SCAT_WAVES][gps::BWD_KEYS];   // a trip's Gaussians per wave and list key (zero between trips)
    __shared__ int key_run[gps::BWD_KEYS];                // next free position of this superblock's run in a key's list piece
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = sb * sb_size;

    // ---- this superblock's Gaussians: tile counts -> exclusive prefix, boxes, the band of tile rows; class lists
    // class k's list = its bands in order, a band = the superblocks in order, a superblock's run in ascending id
    if (cls_ids && tid < gps::BWD_KEYS) {
        int start = t.cls_prefix[tid * SB_MAX + sb];
        for (int k = tid - tid % gps::BWD_BANDS; k < tid; k++) start += t.cls_prefix[gps::BWD_KEYS * SB_MAX + k];
        key_run[tid] = start;
        for (int w = 0; w < SCAT_WAVES; w++) key_wave[w][tid] = 0;
    }
    int carry = 0, row_lo = th, row_hi = 0;
    for (int j0 = 0; j0 < sb_size; j0 += SCAT_THREADS) {
        const int j = j0 + tid, g = g0 + j;
        const bool mine_j = j < sb_size;   // (a superblock can be smaller than the workgroup)
        int tcount = 0, r = 0;
        uint32_t bx = 0;
        if (mine_j && g < N) {
            tcount = tiles_per_gauss[g];
            r = radii[g];
            if (tcount > 0) {
                const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)g);
                const TileBox b = tile_bbox(m.x, m.y, r, tile_size, tw, th);
                bx = b.x0 | (b.y0 << 12) | ((b.x1 - b.x0) << 24);
                row_lo = min(row_lo, (int)b.y0); row_hi = max(row_hi, (int)b.y1);
            }
        }
        const int incl = wave_incl_scan_i(tcount);
        __syncthreads();   // (ws / key_wave of the previous trip consumed)
        if (lane == 63) ws[wave] = incl;
        // rank among the wave's Gaussians of the same list key: one ballot per DISTINCT key in the wave (a few: neighbours in id
        // are neighbours in the image and alike in size)
        const int key = (cls_ids && mine_j && g < N && r > 0) ? gps::bwd_key(r, tcount, bx >> 12 & 0xfff, th) : -1;
        int rank = 0;
        for (unsigned long long rem = __ballot(key >= 0); rem;) {
            const int kk = __shfl(key, __ffsll((long long)rem) - 1, 64);
            const unsigned long long m = __ballot(key == kk);
            if (key == kk) rank = __popcll(m & lanemask_lt());
            if (lane == 0) key_wave[wave][kk] = __popcll(m);
            rem &= ~m;
        }
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < SCAT_WAVES; w++) { const int v = ws[w]; if (w < wave) woff += v; total += v; }
        if (mine_j) { pre[j] = (uint32_t)(carry + woff + incl - tcount); box[j] = bx; }
        carry += total;
        if (cls_ids) {
            if (key >= 0) {
                int before = 0;
                for (int w = 0; w < wave; w++) before += key_wave[w][key];
                cls_ids[(key / gps::BWD_BANDS) * cls_stride + key_run[key] + before + rank] = g;
            }
            __syncthreads();
            if (tid < gps::BWD_KEYS) {
                int all = 0;
                for (int w = 0; w < SCAT_WAVES; w++) { all += key_wave[w][tid]; key_wave[w][tid] = 0; }
                key_run[tid] += all;
            }
        }
    }
    const int n_pairs = carry;
    row_lo = wave_min_i(row_lo); row_hi = wave_max_i(row_hi);
    __syncthreads();
    if (lane == 0) { wlo[wave] = row_lo; whi[wave] = row_hi; }
    if (tid == 0) pre[sb_size] = (uint32_t)carry;
    __syncthreads();
    for (int w = 0; w < SCAT_WAVES; w++) { row_lo = min(row_lo, wlo[w]); row_hi = max(row_hi, whi[w]); }
    int t_lo = min(row_lo, row_hi) * tw, t_hi = row_hi * tw;   // the band [t_lo, t_hi)
    if (t_hi < t_lo) t_hi = t_lo;
    const int nt = t_hi - t_lo;
    if (n_pairs == 0) return;

    // ---- tile starts of the band = (sum of the tile totals in front of it) + exclusive scan inside it, + this superblock's prefix
    {
        int before = 0;
        for (int b = tid; b < t_lo; b += SCAT_THREADS) before += (int)t.tile_total[b];
        const int per = (nt + SCAT_THREADS - 1) / SCAT_THREADS;
        const int lo = min(nt, tid * per), hi = min(nt, lo + per);
        int sum = 0;
        for (int b = lo; b < hi; b++) sum += (int)t.tile_total[t_lo + b];
        const int incl = wave_incl_scan_i(sum);
        before = wave_sum_i(before);
        __syncthreads();
        if (lane == 63) ws[wave] = incl;
        if (lane == 0) wlo[wave] = before;
        __syncthreads();
        int woff = 0, total = 0, front = 0;
        for (int w = 0; w < SCAT_WAVES; w++) { const int v = ws[w]; if (w < wave) woff += v; total += v; front += wlo[w]; }
        int run = front + woff + incl - sum;
        for (int b = lo; b < hi; b++) {
            base[b] = (uint32_t)run + t.P[(size_t)(t_lo + b) * SB_MAX + sb];
            run += (int)t.tile_total[t_lo + b];
        }
        (void)total;
    }
    // ---- chunks of SCAT_CHUNK pairs in Gaussian order; wave w owns pairs [256 w, 256 w + 256) of a chunk, visited
    // iteration-major, lane-minor: "earlier pair, same tile" == stable rank (as wide_scatter_kernel of splat_bin.hip)
    int bits = 1;
    while ((1 << bits) < max(nt, 2)) bits++;
    const unsigned long long lt = lanemask_lt();
    uint16_t* mycnt = wavecnt + wave * n_tiles;
    for (int c0 = 0; c0 < n_pairs; c0 += SCAT_CHUNK) {
        for (int k = tid; k < nt; k += SCAT_THREADS) {
#pragma unroll
            for (int w = 0; w < SCAT_WAVES; w++) wavecnt[w * n_tiles + k] = 0;
        }
        __syncthreads();
        uint32_t tile[SCAT_ITEMS], gid[SCAT_ITEMS], rank[SCAT_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAT_ITEMS; k++) {
            const int p = c0 + wave * (64 * SCAT_ITEMS) + k * 64 + lane;
            const bool valid = p < n_pairs;
            uint32_t d = 0;
            gid[k] = 0;
            if (valid) {
                int lo = 0, hi = sb_size;   // largest j with pre[j] <= p (Gaussians without tiles have equal prefixes: take the last)
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)pre[mid] <= p) lo = mid; else hi = mid; }
                const uint32_t b = box[lo];
                const uint32_t w = b >> 24, q = (uint32_t)(p - (int)pre[lo]);
                d = ((b >> 12) & 0xfffu) * (uint32_t)tw + (b & 0xfffu) + (q / w) * (uint32_t)tw + q % w - (uint32_t)t_lo;
                gid[k] = (uint32_t)(g0 + lo);
            }
            tile[k] = d;   // tile id inside the band
            unsigned long long same = __ballot(valid);
            for (int bb = 0; bb < bits; bb++) {
                const unsigned long long bal = __ballot(valid && ((d >> bb) & 1u));
                same &= ((d >> bb) & 1u) ? bal : ~bal;
            }
            const uint32_t prev = mycnt[d];
            rank[k] = prev + (uint32_t)__popcll(same & lt);
            if (valid && (same >> lane) == 1ull) mycnt[d] = (uint16_t)(prev + (uint32_t)__popcll(same));
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // per tile of the band: exclusive prefix over the waves, the chunk's total
        for (int b = tid; b < nt; b += SCAT_THREADS) {
            uint32_t acc = 0;
#pragma unroll
            for (int w = 0; w < SCAT_WAVES; w++) { const uint32_t c = wavecnt[w * n_tiles + b]; wavecnt[w * n_tiles + b] = (uint16_t)acc; acc += c; }
            tot16[b] = (uint16_t)acc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAT_ITEMS; k++) {
            const int p = c0 + wave * (64 * SCAT_ITEMS) + k * 64 + lane;
            if (p < n_pairs) {
                const uint32_t d = tile[k];
                const int64_t pos = (int64_t)base[d] + mycnt[d] + rank[k];
                if (pos < isect_cap) flatten_ids[pos] = (int32_t)gid[k];
            }
        }
        if (c0 + SCAT_CHUNK < n_pairs) {   // the next chunk's pairs of a tile go behind this chunk's
            __syncthreads();
            for (int b = tid; b < nt; b += SCAT_THREADS) base[b] += tot16[b];
        }
    }
}

}  // namespace

namespace gps {

// Round 5, measured and dropped: WINDOWED per-tile tables (a window of R tile rows of the superblock's band in LDS, one pass over the
// superblock's pairs per window; bit-equal).  At Replica's 75 x 43 tiles the whole-grid tables are 79 KB (one workgroup per compute
// unit) -- but a superblock's band is most of the image in the SLAM loop (radii up to 100 px, Gaussians of several keyframes'
// views): R = 8 / 16 / 24 rows: 143 / 84 / 65 us per launch against 62 for the whole grid (640x480: 44 us at R = 8 against 22).
// Dynamic LDS of sb_scatter_kernel: per-tile histogram + per-wave cursors (n_tiles words + (SCAT_WAVES + 1) n_tiles halves) + two
// words per Gaussian of a superblock.  It grows with the tile count (79 KB at 1200x680, up to 160 KB): above the default 64 KB a
// workgroup needs the opt-in attribute, which a device with less LDS per workgroup refuses.
static size_t sb_scatter_lds_bytes(int N, int n_tiles) {
    const size_t sb_size = (size_t)BIN_BLOCK << sb_shift_for(N);
    return ((size_t)n_tiles + ((SCAT_WAVES + 1) * (size_t)n_tiles + 1) / 2 + 2 * sb_size + 2) * 4;
}

// Everything the superblock binning needs, checked BEFORE the preprocessing kernel adds its counts to the persistent tables
// (isect_count_targets / strips_on): a configuration that fails here takes the sorted-key binning of splat_bin.hip instead.
bool sb_supported(int N, int tile_width, int tile_height) {
    if (N <= 0 || tile_width <= 0 || tile_height <= 0) return false;
    const int64_t n_tiles = (int64_t)tile_width * tile_height;
    // (the scatter packs a box's width and height into 8 bits each)
    if (n_tiles > SB_MAX_TILES || tile_width > 255 || tile_height > 255) return false;
    const int shift = sb_shift_for(N);
    const int64_t nblk = ((int64_t)N + BIN_BLOCK - 1) / BIN_BLOCK;
    if (((nblk + ((int64_t)1 << shift) - 1) >> shift) > SB_MAX) return false;
    const size_t lds = sb_scatter_lds_bytes(N, (int)n_tiles);
    if (lds > 160 * 1024) return false;
    if (lds <= 64 * 1024) return true;   // (no opt-in needed: the common sizes never take the lock below)
    // the opt-in for more than 64 KB of dynamic LDS is a per-DEVICE function attribute: granted once per device and size class
    // (monotonic; several host threads may ask)
    constexpr int MAX_DEVICES = 64;
    static std::mutex mu;
    static size_t granted[MAX_DEVICES] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(mu);
    if (lds <= granted[dev]) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    granted[dev] = lds;
    return true;
}

size_t sb_tables_bytes() {
    return (size_t)SB_MAX_TILES * SB_MAX * 4 * 2 + (size_t)SB_MAX_TILES * 4 * 2 + (size_t)(SB_MAX + 1) * BWD_KEYS * 4 * 2 + 2048;
}

// the tables back to "zero between launches" (after a failed launch of the preprocessing kernel or a rejected binning call)
int sb_tables_clear(const SbTables& t, gps_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool ok = hipMemsetAsync(t.C, 0, (size_t)SB_MAX_TILES * SB_MAX * 4, s) == hipSuccess &&
                    hipMemsetAsync(t.cls_count, 0, (size_t)(SB_MAX + 1) * BWD_KEYS * 4, s) == hipSuccess;
    return ok ? GPS_OK : GPS_ERR_LAUNCH;
}

void sb_tables_carve(char* base, SbTables* t) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += (bytes + 255) & ~(size_t)255; return p; };
    t->C = (uint32_t*)take((size_t)SB_MAX_TILES * SB_MAX * 4);
    t->P = (uint32_t*)take((size_t)SB_MAX_TILES * SB_MAX * 4);
    t->tile_total = (uint32_t*)take((size_t)SB_MAX_TILES * 4);
    t->cls_count = (int32_t*)take((size_t)(SB_MAX + 1) * BWD_KEYS * 4);
    t->cls_prefix = (int32_t*)take((size_t)(SB_MAX + 1) * BWD_KEYS * 4);
    t->tile_order = (int32_t*)take((size_t)SB_MAX_TILES * 4);
    t->sb_shift = 0;
}

int isect_tiles_superblock(int N, const float* means2d, const int32_t* radii, const BinCountOut& cnt, int64_t isect_capacity,
                           const int32_t* tiles_per_gauss, int32_t* flatten_ids, int32_t* tile_offsets, int64_t* counts,
                           int32_t* cls_ids, int32_t* cls_counts, int64_t cls_stride, gps_stream stream) {
    GPS_ENTER();
    const int n_tiles = cnt.tw * cnt.th;
    hipStream_t s = (hipStream_t)stream;
    // The preprocessing kernel has already added this launch's counts to the persistent tables, which only sb_scan_kernel clears:
    // an argument error from here on must not leave them dirty for the next binning (every size-dependent condition was checked
    // by sb_supported() before that kernel ran; what is left are the caller's pointers).
    const bool ok = N > 0 && cnt.sb.C && isect_capacity > 0 && means2d && radii && tiles_per_gauss && flatten_ids && tile_offsets &&
                    counts && (!cls_ids || (cls_counts && cls_stride >= N)) && cnt.sb.sb_shift == sb_shift_for(N) &&
                    sb_supported(N, cnt.tw, cnt.th);
    if (!ok) {
        if (cnt.sb.C) (void)sb_tables_clear(cnt.sb, stream);
        return GPS_ERR_ARG;
    }
    const int nblk = gps_div_up(N, BIN_BLOCK);
    const int n_sb = (nblk + (1 << cnt.sb.sb_shift) - 1) >> cnt.sb.sb_shift;
    launch_kernel(TK_SB_SCAN, 0, sb_scan_kernel, dim3(gps_div_up(n_tiles + BWD_KEYS, 4)), dim3(256), 0, s, n_tiles, cnt.sb);
    const size_t lds = sb_scatter_lds_bytes(N, n_tiles);
    launch_kernel(TK_SB_SCATTER, 0, sb_scatter_kernel, dim3(n_sb + 1), dim3(SCAT_THREADS), lds, s, N, means2d, radii, tiles_per_gauss,
                  cnt.tile_size, cnt.tw, cnt.th, cnt.sb, isect_capacity, flatten_ids, tile_offsets, counts, cls_ids, cls_counts, cls_stride);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // namespace gps
