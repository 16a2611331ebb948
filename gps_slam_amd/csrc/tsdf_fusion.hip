// TSDF voxel-block-hash fusion for gfx950: depth conversion, hash allocation + visible list,
// integration.  Compiled with -ffp-contract=off: voxel updates end in a truncating (short)(x*32767)
// store, so every mul/add must round exactly like the CPU engine (IEEE div/sqrt are hipcc defaults).
//
// What each kernel computes follows the reference's per-element code
//   InfiniTAM/ITMLib/Engines/Reconstruction/Shared/ITMSceneReconstructionEngine_Shared.h
// (line-level restatement: oracle/tsdf_oracle.c).  How it is parallelised is new:
//
//  * Allocation is DETERMINISTIC.  The reference CUDA kernel lets racing pixels overwrite
//    entriesAllocType/blockCoords (last writer wins, order undefined) and hands out blocks with
//    atomicSub (order undefined).  Here each pixel posts atomicMax(prio[slot], (pixel, step)) so
//    the winner is the last requester in scan order -- exactly what the single-threaded CPU
//    engine produces -- and the winner's block coordinates are re-derived from (pixel, step) in
//    the allocation sweep.  Blocks / excess entries are handed out by an ordered prefix sum over
//    the hash slots (ascending slot index, like ITMSceneReconstructionEngine_CPU.tpp:196-265),
//    so even `ptr` and `offset` match the CPU engine bit for bit.
//  * The visible list is built by an ordered stream compaction (count -> scan -> write) and stays
//    on the device: nothing is copied back to size the next launch (the reference blocks on a
//    12-byte cudaMemcpy, ..._CUDA.tcu:197); dependent kernels are persistent and grid-stride
//    over counters[GPS_TSDF_N_VISIBLE].
//  * Integration: one wave64 per visible block walking its 8 z-slices (coalesced 512-byte slice
//    accesses, 8-byte voxel per lane), ~8k blocks in flight to hide the per-block header latency.
#include "launch_timing.hpp"
#include "tsdf_common.hpp"

using namespace gpst;

namespace {

// ---------------------------------------------------------------- reset
__global__ __launch_bounds__(256) void reset_kernel(TsdfState s, uint32_t* __restrict__ bits) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nvox = (int64_t)s.n_blocks * BLK3;
    const uint64_t empty = 0x0000000000007FFFull;  // sdf = 32767, everything else 0
    uint64_t* v = reinterpret_cast<uint64_t*>(s.vba);
    for (int64_t i = tid; i < nvox; i += stride) v[i] = empty;
    const int n_total = s.n_buckets + s.n_excess;
    for (int64_t i = tid; i < n_total; i += stride) {
        gps_hash_entry e;
        e.pos[0] = e.pos[1] = e.pos[2] = 0; e.pad_ = 0; e.offset = 0; e.ptr = -2;
        s.hash[i] = e;
        s.visible_type[i] = 0;
        s.alloc_prio[i] = 0u;
    }
    for (int64_t i = tid; i < s.n_blocks; i += stride) s.vba_alloc_list[i] = (int32_t)i;
    for (int64_t i = tid; i < s.n_excess; i += stride) s.excess_list[i] = (int32_t)i;
    // min/max images: everything outside the 1/8-resolution window CreateExpectedDepths rewrites stays at this value
    const float2 mm_init = make_float2(FAR_AWAY, VERY_CLOSE);
    float2* mm = reinterpret_cast<float2*>(s.minmax);
    float2* fmm = reinterpret_cast<float2*>(s.fv_minmax);
    for (int64_t i = tid; i < (int64_t)s.width * s.height; i += stride) { mm[i] = mm_init; fmm[i] = mm_init; }
    for (int64_t i = tid; i < (s.n_buckets + 31) / 32; i += stride) bits[i] = 0u;
    if (tid < 16) {
        int v0 = 0;
        if (tid == GPS_TSDF_LAST_FREE_BLOCK) v0 = s.n_blocks - 1;
        if (tid == GPS_TSDF_LAST_FREE_EXCESS) v0 = s.n_excess - 1;
        s.counters[tid] = v0;
    }
}

// bucket-occupancy bitmap from the table (gps_tsdf_rebuild_index)
__global__ __launch_bounds__(256) void rebuild_bits_kernel(TsdfState s, uint32_t* __restrict__ bits) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool occ = idx < s.n_buckets && s.hash[idx].ptr != -2;
    const unsigned long long b = __ballot(occ);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && idx < s.n_buckets) bits[idx >> 5] = (uint32_t)b;
    if (lane == 32 && idx < s.n_buckets) bits[idx >> 5] = (uint32_t)(b >> 32);
}

// ---------------------------------------------------------------- view building
__global__ __launch_bounds__(256) void convert_depth_kernel(int P, const int16_t* __restrict__ in, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    int16_t d = in[i];
    out[i] = d <= 0 ? -1.0f : (float)d * (1.0f / 1000.0f) + 0.0f;
}

// ---------------------------------------------------------------- allocation requests
// Steps along the truncation band of one depth pixel (Shared.h:207-323).  The geometry is
// recomputed by the allocation sweep from (loc, step) via the same function, so both see the
// identical fp32 sequence.
struct BandWalk {
    float px, py, pz;      // current point in block units
    float dx, dy, dz;
    int n_steps;
};

__device__ __forceinline__ bool band_begin(const TsdfState& s, const Mat4& invM, int x, int y, BandWalk& w) {
    const float depth_measure = s.depth[x + y * s.width];
    const float mu = s.mu;
    if (depth_measure <= 0 || (depth_measure - mu) < 0 || (depth_measure - mu) < s.view_frustum_min ||
        (depth_measure + mu) > s.view_frustum_max)
        return false;
    const float inv_fx = 1.0f / s.fx, inv_fy = 1.0f / s.fy;
    const float oneOverVoxelSize = 1.0f / (s.voxel_size * BLK);
    float cz = depth_measure;
    float cx = cz * (((float)x - s.cx) * inv_fx);
    float cy = cz * (((float)y - s.cy) * inv_fy);
    float norm = sqrtf(cx * cx + cy * cy + cz * cz);
    float sc = 1.0f - mu / norm;
    float qx, qy, qz;
    mul_point(invM, cx * sc, cy * sc, cz * sc, 1.0f, qx, qy, qz);
    w.px = qx * oneOverVoxelSize; w.py = qy * oneOverVoxelSize; w.pz = qz * oneOverVoxelSize;
    sc = 1.0f + mu / norm;
    mul_point(invM, cx * sc, cy * sc, cz * sc, 1.0f, qx, qy, qz);
    float ex = qx * oneOverVoxelSize, ey = qy * oneOverVoxelSize, ez = qz * oneOverVoxelSize;
    w.dx = ex - w.px; w.dy = ey - w.py; w.dz = ez - w.pz;
    norm = sqrtf(w.dx * w.dx + w.dy * w.dy + w.dz * w.dz);
    w.n_steps = (int)ceilf(2.0f * norm);
    const float dn = (float)(w.n_steps - 1);
    w.dx /= dn; w.dy /= dn; w.dz /= dn;
    return true;
}

__global__ __launch_bounds__(256) void mark_previous_visible_kernel(TsdfState s) {
    GPS_FRAME_PRIO();
    const int n = s.counters[GPS_TSDF_N_VISIBLE];
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s.visible_type[s.visible_ids[i]] = 3;
}

__global__ __launch_bounds__(256) void alloc_request_kernel(TsdfState s, Mat4 invM) {
    GPS_FRAME_PRIO();
    const int loc = blockIdx.x * blockDim.x + threadIdx.x;
    if (loc >= s.width * s.height) return;
    const int y = loc / s.width, x = loc - y * s.width;
    BandWalk w;
    if (!band_begin(s, invM, x, y, w)) return;
    const int steps = min(w.n_steps, MAX_BAND_STEPS);
    for (int i = 0; i < steps; i++) {
        const int bx = (int)(short)floorf(w.px), by = (int)(short)floorf(w.py), bz = (int)(short)floorf(w.pz);
        int hashIdx = hash_index(bx, by, bz, s.n_buckets - 1);
        HashEntry he = load_entry(s.hash, hashIdx);
        bool found = false;
        if (entry_is(he, bx, by, bz) & (he.ptr >= -1)) {
            s.visible_type[hashIdx] = (he.ptr == -1) ? 2 : 1;
            found = true;
        }
        if (!found) {
            if (he.ptr >= -1) {
                while (he.offset >= 1) {
                    hashIdx = s.n_buckets + he.offset - 1;
                    he = load_entry(s.hash, hashIdx);
                    if (entry_is(he, bx, by, bz) & (he.ptr >= -1)) {
                        s.visible_type[hashIdx] = (he.ptr == -1) ? 2 : 1;
                        found = true;
                        break;
                    }
                }
            }
            // last requester in (pixel, step) order wins the slot, like the sequential CPU loop
            if (!found) atomicMax(&s.alloc_prio[hashIdx], ((uint32_t)(loc + 1) << BAND_STEP_BITS) | (uint32_t)i);
        }
        w.px += w.dx; w.py += w.dy; w.pz += w.dz;
    }
}

// ---------------------------------------------------------------- ordered sweeps over the hash slots
constexpr int SWEEP = 1024;  // slots per workgroup
constexpr int SCAN_THREADS = 256;  // the single-workgroup scan of the per-workgroup counts (<= 1,152 of them: 5 per thread)

__device__ __forceinline__ int block_excl_scan_1024(int v, int* ws /*[17]*/, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = wave_incl_scan_i(v);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int nw = (int)blockDim.x >> 6;  // (16 waves or fewer)
        int sv = lane < nw ? ws[lane] : 0;
        int si = wave_incl_scan_i(sv);
        if (lane < nw) ws[lane] = si - sv;
        if (lane == 63) ws[16] = si;
    }
    __syncthreads();
    int r = ws[wave] + incl - v;
    total = ws[16];
    __syncthreads();
    return r;
}

// The sweeps' workgroups: SWEEP slots each, walked by SWEEP_THREADS threads that own SLOTS_PER_THREAD CONSECUTIVE slots (so the
// order of a compaction is still the slot order).  Round 3: 256 threads x 4 slots instead of 1,024 x 1 -- a 16-wave workgroup
// needs four free wave slots on every SIMD of ONE compute unit at the same moment, and beside the map stream's rasterizer
// (8-wave workgroups that refill a unit as soon as one of them retires) it waited for that for up to 50 us (rocprof, overlap
// table: alloc_count / alloc_apply / visible_count 13-16 us on average against 8 alone, max 48-52); a 4-wave workgroup fits into
// the hole ANY retiring rasterizer workgroup leaves.
constexpr int SWEEP_THREADS = 256, SLOTS_PER_THREAD = SWEEP / SWEEP_THREADS;
static_assert(SLOTS_PER_THREAD == 4 && SWEEP_THREADS == 256, "block_excl_scan_4 is written for 4 waves x 4 slots");

// exclusive prefix of v[0..3] of every thread over the workgroup's 1,024 slots (thread-major), total to `total`
__device__ __forceinline__ void block_excl_scan_4(const int (&v)[4], int* ws /*[5]*/, int (&excl)[4], int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mine = v[0] + v[1] + v[2] + v[3];
    const int incl = wave_incl_scan_i(mine);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int a = ws[0], b = ws[1], c = ws[2], d = ws[3];
        ws[0] = 0; ws[1] = a; ws[2] = a + b; ws[3] = a + b + c; ws[4] = a + b + c + d;
    }
    __syncthreads();
    excl[0] = ws[wave] + incl - mine;
    excl[1] = excl[0] + v[0]; excl[2] = excl[1] + v[1]; excl[3] = excl[2] + v[2];
    total = ws[4];
    __syncthreads();
}

// request type of a slot given the PRE-allocation table: empty bucket -> 1 (ordered), occupied chain end -> 2
__device__ __forceinline__ int request_type(const TsdfState& s, int idx) {
    if (s.alloc_prio[idx] == 0u) return 0;
    return s.hash[idx].ptr < -1 ? 1 : 2;
}

__global__ __launch_bounds__(SWEEP_THREADS) void alloc_count_kernel(TsdfState s, int32_t* __restrict__ blk1, int32_t* __restrict__ blk2) {
    GPS_FRAME_PRIO();
    __shared__ int ws[5];
    const int idx0 = blockIdx.x * SWEEP + threadIdx.x * SLOTS_PER_THREAD;
    const int n_total = s.n_buckets + s.n_excess;
    int t1[4], t2[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = idx0 + j < n_total ? request_type(s, idx0 + j) : 0;
        t1[j] = t == 1; t2[j] = t == 2;
    }
    int tot1, tot2;
    block_excl_scan_4(t1, ws, e, tot1);
    block_excl_scan_4(t2, ws, e, tot2);
    if (threadIdx.x == 0) { blk1[blockIdx.x] = tot1; blk2[blockIdx.x] = tot2; }
}

// single workgroup: exclusive scan of up to two arrays of per-block counts, totals to out_tot[0..1]
__global__ __launch_bounds__(SCAN_THREADS) void scan_counts_kernel(int nblk, int32_t* __restrict__ a, int32_t* __restrict__ b,
                                                                  int32_t* __restrict__ out_tot, const ViewRec* __restrict__ views) {
    GPS_FRAME_PRIO();
    __shared__ int ws[17];
    if (views) {  // free-view batch: the sweep counts and the list length of view blockIdx.z
        a = views[blockIdx.z].scratch + 2 * nblk; b = nullptr; out_tot = views[blockIdx.z].counters + GPS_TSDF_N_VISIBLE_FREE;
    }
    const int per = (nblk + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(nblk, (int)threadIdx.x * per), hi = min(nblk, lo + per);
    for (int which = 0; which < 2; which++) {
        int32_t* arr = which == 0 ? a : b;
        if (arr == nullptr) continue;
        int sum = 0;
        for (int k = lo; k < hi; k++) sum += arr[k];
        int total;
        int run = block_excl_scan_1024(sum, ws, total);
        for (int k = lo; k < hi; k++) { int v = arr[k]; arr[k] = run; run += v; }
        if (threadIdx.x == 0) out_tot[which] = total;
    }
}

__global__ __launch_bounds__(SWEEP_THREADS) void alloc_apply_kernel(TsdfState s, Mat4 invM, const int32_t* __restrict__ blk1,
                                                                   const int32_t* __restrict__ blk2, uint32_t* __restrict__ bits) {
    GPS_FRAME_PRIO();
    __shared__ int ws[5];
    const int idx0 = blockIdx.x * SWEEP + threadIdx.x * SLOTS_PER_THREAD;
    const int n_total = s.n_buckets + s.n_excess;
    uint32_t prio[4];
    int t[4], t1[4], t2[4], c1[4], c2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        prio[j] = idx0 + j < n_total ? s.alloc_prio[idx0 + j] : 0u;
        t[j] = prio[j] == 0u ? 0 : (s.hash[idx0 + j].ptr < -1 ? 1 : 2);
        t1[j] = t[j] == 1; t2[j] = t[j] == 2;
    }
    int tot;
    block_excl_scan_4(t1, ws, c1, tot);
    block_excl_scan_4(t2, ws, c2, tot);
    if (__ballot((t[0] | t[1] | t[2] | t[3]) != 0) == 0ull) return;  // (wave-uniform: every lane of a wave with requests stays for the count reduction)
    const int base1 = blk1[blockIdx.x], base2 = blk2[blockIdx.x];
    // state of the sequential allocator when it reaches a slot (CPU.tpp:196-265): every earlier
    // type-1 request and the first E type-2 requests consumed one voxel block each.
    const int lastBlock = s.counters[GPS_TSDF_LAST_FREE_BLOCK], lastExcess = s.counters[GPS_TSDF_LAST_FREE_EXCESS];
    const int E = lastExcess + 1;
    int n_blocks_taken = 0, n_excess_taken = 0;  // this thread's allocations: added to the frame's scratch counters once per wave
    for (int j = 0; j < 4; j++) {
        if (t[j] == 0) continue;
        const int idx = idx0 + j;
        s.alloc_prio[idx] = 0u;  // leave the scratch clean for the next frame
        const int n1 = base1 + c1[j], n2 = base2 + c2[j];
        const int vbaIdx = lastBlock - (n1 + min(n2, E));
        const int exlIdx = lastExcess - n2;
        // winner's block coordinates, re-derived from (pixel, step)
        const int loc = (int)(prio[j] >> BAND_STEP_BITS) - 1, step = (int)(prio[j] & ((1u << BAND_STEP_BITS) - 1u));
        const int y = loc / s.width, x = loc - y * s.width;
        BandWalk w;
        band_begin(s, invM, x, y, w);
        for (int i = 0; i < step; i++) { w.px += w.dx; w.py += w.dy; w.pz += w.dz; }
        gps_hash_entry ne;
        ne.pos[0] = (short)floorf(w.px); ne.pos[1] = (short)floorf(w.py); ne.pos[2] = (short)floorf(w.pz);
        ne.pad_ = 0; ne.offset = 0;
        if (t[j] == 1) {
            if (vbaIdx >= 0) {
                ne.ptr = s.vba_alloc_list[vbaIdx];
                s.hash[idx] = ne;
                atomicOr(&bits[idx >> 5], 1u << (idx & 31));  // bucket head now non-empty (idx < n_buckets for type-1 requests)
                s.visible_type[idx] = 1;  // "new entry is visible" (Shared.h:311)
                n_blocks_taken++;
            } else {
                s.visible_type[idx] = 0;
            }
        } else {
            if (vbaIdx >= 0 && exlIdx >= 0) {
                ne.ptr = s.vba_alloc_list[vbaIdx];
                const int exlOffset = s.excess_list[exlIdx];
                s.hash[idx].offset = exlOffset + 1;
                s.hash[s.n_buckets + exlOffset] = ne;
                s.visible_type[s.n_buckets + exlOffset] = 1;
                n_blocks_taken++;
                n_excess_taken++;
            }
        }
    }
    int nb = n_blocks_taken, ne = n_excess_taken;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nb += __shfl_xor(nb, o, 64); ne += __shfl_xor(ne, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (nb) atomicAdd(&s.counters[GPS_TSDF_SCRATCH0], nb);
        if (ne) atomicAdd(&s.counters[GPS_TSDF_SCRATCH1], ne);
    }
}

// visibility predicates for the two ordered compactions
enum { VIS_LIVE = 0, VIS_FREE = 1 };

template <int MODE>
__device__ __forceinline__ bool slot_visible(const TsdfState& s, const Mat4& M, int idx, bool update) {
    if (MODE == VIS_LIVE) {
        // CPU.tpp:268-306: type 3 (visible last frame) is re-tested against the frustum, types 1/2 stay
        uint8_t vt = s.visible_type[idx];
        if (vt == 3) {
            const HashEntry he = load_entry(s.hash, idx);
            if (!block_visible(s, M, he.x, he.y, he.z)) vt = 0;
            if (update) s.visible_type[idx] = vt;
        }
        return vt > 0;
    } else {
        // Visualisation CPU.tpp:36-74 / buildCompleteVisibleList_device: every allocated block inside the frustum
        const HashEntry he = load_entry(s.hash, idx);
        return he.ptr >= 0 && block_visible(s, M, he.x, he.y, he.z);
    }
}

template <int MODE>
__global__ __launch_bounds__(SWEEP_THREADS) void visible_count_kernel(TsdfState s, Mat4 M, int32_t* __restrict__ blk,
                                                                     uint8_t* __restrict__ flags, const ViewRec* __restrict__ views) {
    GPS_FRAME_PRIO();
    __shared__ int ws[5];
    if (views) { apply_view(s, views[blockIdx.z]); M = views[blockIdx.z].M; blk = sweep_counts(s); flags = sweep_flags(s); }
    const int idx0 = blockIdx.x * SWEEP + threadIdx.x * SLOTS_PER_THREAD;
    const int n_total = s.n_buckets + s.n_excess;
    if (MODE == VIS_LIVE && blockIdx.x == 0 && threadIdx.x == 0) {
        // fold the allocation bookkeeping of this frame into the counters (ordered after alloc_apply)
        s.counters[GPS_TSDF_LAST_FREE_BLOCK] -= s.counters[GPS_TSDF_SCRATCH0];
        s.counters[GPS_TSDF_LAST_FREE_EXCESS] -= s.counters[GPS_TSDF_SCRATCH1];
        s.counters[GPS_TSDF_SCRATCH0] = 0;
        s.counters[GPS_TSDF_SCRATCH1] = 0;
    }
    int v[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j] = idx0 + j < n_total ? (slot_visible<MODE>(s, M, idx0 + j, true) ? 1 : 0) : 0;
        if (idx0 + j < n_total) flags[idx0 + j] = (uint8_t)v[j];
    }
    int tot;
    block_excl_scan_4(v, ws, e, tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SWEEP_THREADS) void visible_write_kernel(TsdfState s, const int32_t* __restrict__ blk,
                                                                     const uint8_t* __restrict__ flags,
                                                                     int32_t* __restrict__ out_ids, int cap,
                                                                     const ViewRec* __restrict__ views) {
    GPS_FRAME_PRIO();
    __shared__ int ws[5];
    if (views) { apply_view(s, views[blockIdx.z]); blk = sweep_counts(s); flags = sweep_flags(s); out_ids = s.fv_visible_ids; }
    const int idx0 = blockIdx.x * SWEEP + threadIdx.x * SLOTS_PER_THREAD;
    const int n_total = s.n_buckets + s.n_excess;
    int v[4], e[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = idx0 + j < n_total ? (flags[idx0 + j] != 0 ? 1 : 0) : 0;
    int tot;
    block_excl_scan_4(v, ws, e, tot);
    const int base = blk[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (v[j] && base + e[j] < cap) out_ids[base + e[j]] = idx0 + j;
}

// ---------------------------------------------------------------- integration
// One wave64 per visible block: the wave walks the 8 z-slices (lane = (x, y) of the 8x8 slice), so a slice is one
// coalesced 512-byte access and the per-block header chain (visible_ids -> hash entry) is paid once per 512 voxels
// while 8 waves per SIMD keep ~8k independent blocks in flight chip-wide (the per-block dependent-load latency, not
// bandwidth, bounded the earlier one-workgroup-per-block layout).
// Divergence.  The colour update (4 bilinear taps x 3 channels + exact divisions: ~100 VALU ops) applies only to voxels within
// a quarter of the band (~20 % of an allocated block), but with lane = voxel of a slice nearly every slice has SOME such
// voxel, so all 64 lanes sat through it 8 times per block (66 M wave instructions per launch, 180 per slice).  The colour
// work is therefore split off: phase 1 updates depth / weight for every slice and pushes the (slice, lane) pairs that need
// colour into a per-wave LDS queue (ballot + prefix); phase 2 drains the queue 64 tasks at a time with all lanes busy,
// re-deriving the projection with the same operations (bit-identical) and re-reading the voxel phase 1 just wrote.
template <bool FAST_DIV>
__device__ __forceinline__ bool project_voxel(const TsdfState& s, const Mat4& M, float pmx, float pmy, float pmz, int W, int H,
                                              float& cz, float& ix, float& iy) {
    float cx, cy;
    mul_point(M, pmx, pmy, pmz, 1.0f, cx, cy, cz);
    if (cz <= 0) return false;
    if (FAST_DIV && cz >= 1e-4f) {  // (wave-uniform in practice; tiny cz would overflow the unscaled sequence)
        const float rz = refined_rcp(cz);
        ix = div_shared(s.fx * cx, cz, rz) + s.cx;
        iy = div_shared(s.fy * cy, cz, rz) + s.cy;
    } else {
        ix = s.fx * cx / cz + s.cx;
        iy = s.fy * cy / cz + s.cy;
    }
    return !((ix < 1) || (ix > W - 2) || (iy < 1) || (iy > H - 2));
}

template <bool FAST_DIV>
__global__ __launch_bounds__(256) void integrate_kernel(TsdfState s, Mat4 M, gps::LaunchStamp stamp) {
    gps::StampScope timed(stamp);
    GPS_FRAME_PRIO();
    __shared__ uint16_t queue[4][BLK3];  // per wave: (slice << 6 | lane) of the voxels that take the colour update
    const int n_visible = s.counters[GPS_TSDF_N_VISIBLE];
    const int lane = threadIdx.x & 63;
    const int lx = lane & 7, ly = lane >> 3;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    uint16_t* q = queue[threadIdx.x >> 6];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int W = s.width, H = s.height;
    const float mu = s.mu;
    const float inv_mu = 1.0f / mu;                       // RN(1/mu), RN(1/255), RN(1/32767) for div_known
    const float inv_255 = 1.0f / 255.0f, inv_32767 = 1.0f / 32767.0f;
    const uchar4* img = reinterpret_cast<const uchar4*>(s.rgb);
    // the entry of the wave's next block and the list word of the one after it are fetched under the current block's loads
    // (clamped indices: the loads stay unconditional)
    HashEntry he_next = {};
    int id_after = 0;
    if (wave < n_visible) {
        he_next = load_entry(s.hash, s.visible_ids[wave]);
        id_after = s.visible_ids[min(wave + n_waves, n_visible - 1)];
    }
    for (int e = wave; e < n_visible; e += n_waves) {
        const HashEntry he = he_next;
        if (he.ptr < 0) {  // uniform across the wave
            he_next = load_entry(s.hash, id_after);
            id_after = s.visible_ids[min(e + 2 * n_waves, n_visible - 1)];
            continue;
        }
        uint64_t* blk = reinterpret_cast<uint64_t*>(s.vba + (size_t)he.ptr * BLK3) + lane;
        const float pmx = (float)(he.x * BLK + lx) * s.voxel_size;
        const float pmy = (float)(he.y * BLK + ly) * s.voxel_size;
        int n_tasks = 0;
        // ---- phase 1: depth / weight of every voxel.  A batch of slices is read unconditionally (coalesced) while their
        // projections run, and their depth gathers go out together: a few dependent round trips per block (entry, voxels +
        // depths, stores per batch) instead of two per slice -- the kernel is bound by that chain, not by bytes
#ifndef GPS_INTEGRATE_SLICES
#define GPS_INTEGRATE_SLICES 4
#endif
        // slices per batch of loads.  8 (the whole block at once): 88 VGPRs, 5 waves per SIMD, 73.1 us; 4: 70 VGPRs, 7 waves, 71.1 us;
        // 2: 66 VGPRs, 73.1 us (bench scene, 46 k visible blocks)
        constexpr int NS = GPS_INTEGRATE_SLICES;
#pragma unroll
        for (int z0 = 0; z0 < BLK; z0 += NS) {
        uint64_t raw8[NS];
        float cz8[NS], dm8[NS];
        int at8[NS];
        bool ok8[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) raw8[k] = blk[(z0 + k) * 64];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int lz = z0 + k;
            const float pmz = (float)(he.z * BLK + lz) * s.voxel_size;
            float ix = 0.f, iy = 0.f;
            ok8[k] = project_voxel<FAST_DIV>(s, M, pmx, pmy, pmz, W, H, cz8[k], ix, iy);
            at8[k] = ok8[k] ? (int)(ix + 0.5f) + (int)(iy + 0.5f) * W : 0;
        }
#pragma unroll
        for (int k = 0; k < NS; k++) dm8[k] = s.depth[at8[k]];
        if (z0 == 0) {
            he_next = load_entry(s.hash, id_after);
            id_after = s.visible_ids[min(e + 2 * n_waves, n_visible - 1)];
        }
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int lz = z0 + k;
            bool colour = false;
            const float dm = dm8[k];
            const float eta = dm - cz8[k];
            if (ok8[k] && dm > 0.0f && !(eta < -mu)) {
                uint64_t raw = raw8[k];
                // unpack {short sdf; uchar w_depth; uchar clr[3]; uchar w_color; pad}
                const int16_t sdf = (int16_t)(raw & 0xFFFF);
                const int oldW = (int)((raw >> 16) & 0xFF);
                float oldF = FAST_DIV ? div_known((float)sdf, 32767.0f, inv_32767) : (float)sdf / 32767.0f;
                const float eta_mu = FAST_DIV ? div_known(eta, mu, inv_mu) : eta / mu;
                float newF = (1.0f < eta_mu) ? 1.0f : eta_mu;
                int newW = 1;
                newF = oldW * oldF + newW * newF;
                newW = oldW + newW;
                if (FAST_DIV) { const float fw = (float)newW; newF = div_shared(newF, fw, refined_rcp(fw)); }
                else newF /= newW;
                newW = (newW < s.max_w) ? newW : s.max_w;
                raw = (raw & ~0xFFFFFFull) | (uint64_t)(uint16_t)(int16_t)(newF * 32767.0f) | ((uint64_t)(uint8_t)newW << 16);
                blk[lz * 64] = raw;
                colour = !((eta > mu) || (fabsf(eta_mu) > 0.25f));
            }
            const unsigned long long need = __ballot(colour);
            if (colour) q[n_tasks + __popcll(need & lt)] = (uint16_t)((lz << 6) | lane);
            n_tasks += __popcll(need);
        }
        }
        // ---- phase 2: colour of the queued voxels, 64 at a time (one wave's own LDS queue: a wave-level fence, no barrier)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < n_tasks; t0 += 64) {
            if (t0 + lane >= n_tasks) continue;
            const int id = q[t0 + lane];
            const int lz = id >> 6, l2 = id & 63;
            uint64_t* slot = reinterpret_cast<uint64_t*>(s.vba + (size_t)he.ptr * BLK3) + l2 + lz * 64;
            const float vx = (float)(he.x * BLK + (l2 & 7)) * s.voxel_size;
            const float vy = (float)(he.y * BLK + (l2 >> 3)) * s.voxel_size;
            const float vz = (float)(he.z * BLK + lz) * s.voxel_size;
            float cz, ix, iy;
            project_voxel<FAST_DIV>(s, M, vx, vy, vz, W, H, cz, ix, iy);  // passed in phase 1: same operations, same values
            uint64_t raw = *slot;
            // colour: rgb camera == depth camera (trafo_rgb_to_depth is identity, InfiniTAM_tools.cpp:6-10), so the
            // colour projection repeats the depth projection's rounding sequence exactly
            const int px = (int)floorf(ix), py = (int)floorf(iy);
            const float dx = ix - (float)px, dy = iy - (float)py;
            // the four taps go out together (1 <= ix <= W-2: all in the image); the reference reads b, c, d only when their
            // weight is non-zero and uses 0 otherwise
            const uchar4 a = img[px + py * W];
            uchar4 b = img[(px + 1) + py * W], c = img[px + (py + 1) * W], d = img[(px + 1) + (py + 1) * W];
            const uchar4 zero4 = make_uchar4(0, 0, 0, 0);
            if (!(dx != 0)) b = zero4;
            if (!(dy != 0)) c = zero4;
            if (!(dx != 0 && dy != 0)) d = zero4;
            const float oldWc = (float)((raw >> 48) & 0xFF);
            const float sumW = oldWc + 1.0f;
            const float maxWf = (float)(uint8_t)s.max_w;
            const float cw = (sumW < maxWf) ? sumW : maxWf;
            const float rsum = FAST_DIV ? refined_rcp(sumW) : 0.f;
            uint64_t packed = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float fa = k == 0 ? a.x : k == 1 ? a.y : a.z, fb = k == 0 ? b.x : k == 1 ? b.y : b.z;
                const float fc = k == 0 ? c.x : k == 1 ? c.y : c.z, fd = k == 0 ? d.x : k == 1 ? d.y : d.z;
                // ((a*(1-dx))*(1-dy) + (b*dx)*(1-dy)) + (c*(1-dx))*dy) + (d*dx)*dy  (ITMPixelUtils.h:25-26)
                const float m = ((fa * (1.0f - dx) * (1.0f - dy) + fb * dx * (1.0f - dy)) + fc * (1.0f - dx) * dy) +
                                fd * dx * dy;
                const float oldByte = (float)((raw >> (24 + 8 * k)) & 0xFF);
                const float meas = FAST_DIV ? div_known(m, 255.0f, inv_255) : m / 255.0f;
                const float oldC = FAST_DIV ? div_known(oldByte, 255.0f, inv_255) : oldByte / 255.0f;
                float newC = oldC * oldWc + meas * 1.0f;
                if (FAST_DIV) newC = div_shared(newC, sumW, rsum);
                else newC /= sumW;
                const float sc = newC * 255.0f;
                int vi = (int)((sc < 0) ? (sc - 0.5f) : (sc + 0.5f));
                vi = max(0, min(255, vi));
                packed |= (uint64_t)vi << (24 + 8 * k);
            }
            raw = (raw & ~0x00FFFFFFFF000000ull) | packed | ((uint64_t)(uint8_t)cw << 48);
            *slot = raw;
        }
        __builtin_amdgcn_wave_barrier();  // the queue is reused by the wave's next block
    }
}

}  // namespace

extern "C" {

int gps_tsdf_reset(const gps_tsdf_state* sp, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    reset_kernel<<<4096, 256, 0, (hipStream_t)stream>>>(s, bucket_bits(s));
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_rebuild_index(const gps_tsdf_state* sp, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    rebuild_bits_kernel<<<gps_div_up(s.n_buckets, 256), 256, 0, (hipStream_t)stream>>>(s, bucket_bits(s));
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_convert_depth(const gps_tsdf_state* sp, const int16_t* depth_mm, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && depth_mm != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    const int P = sp->width * sp->height;
    convert_depth_kernel<<<gps_div_up(P, 256), 256, 0, (hipStream_t)stream>>>(P, depth_mm, sp->depth);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_allocate(const gps_tsdf_state* sp, const float* M, const float* invM, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && M != nullptr && invM != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    // the (pixel, step) priority packs the step into BAND_STEP_BITS bits
    {
        const float band_blocks = 2.0f * s.mu / (s.voxel_size * BLK);
        GPS_REQUIRE((int)ceilf(2.0f * band_blocks) + 2 < MAX_BAND_STEPS);
        GPS_REQUIRE((int64_t)s.width * s.height + 1 < (1ll << (32 - BAND_STEP_BITS)));
    }
    Mat4 m = load_mat(M), im = load_mat(invM);
    hipStream_t st = (hipStream_t)stream;
    const int P = s.width * s.height;
    const int n_total = s.n_buckets + s.n_excess;
    const int nblk = gps_div_up(n_total, SWEEP);
    int32_t* blk1 = s.scan_scratch;
    int32_t* blk2 = s.scan_scratch + nblk;
    int32_t* blkv = s.scan_scratch + 2 * nblk;
    int32_t* tot = s.scan_scratch + 3 * nblk;  // [2] totals scratch
    mark_previous_visible_kernel<<<256, 256, 0, st>>>(s);
    alloc_request_kernel<<<gps_div_up(P, 256), 256, 0, st>>>(s, im);
    alloc_count_kernel<<<nblk, SWEEP_THREADS, 0, st>>>(s, blk1, blk2);
    scan_counts_kernel<<<1, SCAN_THREADS, 0, st>>>(nblk, blk1, blk2, tot, nullptr);
    alloc_apply_kernel<<<nblk, SWEEP_THREADS, 0, st>>>(s, im, blk1, blk2, bucket_bits(s));
    // ordered visible list (byte flags live behind the per-block counts in scan_scratch)
    uint8_t* flags = reinterpret_cast<uint8_t*>(s.scan_scratch + 3 * nblk + 16);
    visible_count_kernel<VIS_LIVE><<<nblk, SWEEP_THREADS, 0, st>>>(s, m, blkv, flags, nullptr);
    scan_counts_kernel<<<1, SCAN_THREADS, 0, st>>>(nblk, blkv, nullptr, &s.counters[GPS_TSDF_N_VISIBLE], nullptr);
    visible_write_kernel<<<nblk, SWEEP_THREADS, 0, st>>>(s, blkv, flags, s.visible_ids, s.n_blocks, nullptr);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

int gps_tsdf_find_visible(const gps_tsdf_state* sp, const float* M, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && M != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    Mat4 m = load_mat(M);
    hipStream_t st = (hipStream_t)stream;
    const int n_total = s.n_buckets + s.n_excess;
    const int nblk = gps_div_up(n_total, SWEEP);
    int32_t* blkv = s.scan_scratch + 2 * nblk;
    uint8_t* flags = reinterpret_cast<uint8_t*>(s.scan_scratch + 3 * nblk + 16);
    visible_count_kernel<VIS_FREE><<<nblk, SWEEP_THREADS, 0, st>>>(s, m, blkv, flags, nullptr);
    scan_counts_kernel<<<1, SCAN_THREADS, 0, st>>>(nblk, blkv, nullptr, &s.counters[GPS_TSDF_N_VISIBLE_FREE], nullptr);
    visible_write_kernel<<<nblk, SWEEP_THREADS, 0, st>>>(s, blkv, flags, s.fv_visible_ids, s.n_blocks, nullptr);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"

namespace gpst {
// ordered visible lists of n free views in three launches (grid.z = view)
int find_visible_batch(const TsdfState& s, int n, const ViewRec* table, hipStream_t st) {
    const int nblk = sweep_blocks(s);
    const Mat4 none = {};
    visible_count_kernel<VIS_FREE><<<dim3(nblk, 1, n), SWEEP_THREADS, 0, st>>>(s, none, nullptr, nullptr, table);
    scan_counts_kernel<<<dim3(1, 1, n), SCAN_THREADS, 0, st>>>(nblk, nullptr, nullptr, nullptr, table);
    visible_write_kernel<<<dim3(nblk, 1, n), SWEEP_THREADS, 0, st>>>(s, nullptr, nullptr, nullptr, s.n_blocks, table);
    return hipGetLastError() == hipSuccess ? GPS_OK : GPS_ERR_LAUNCH;
}
}  // namespace gpst

extern "C" {

int gps_tsdf_integrate(const gps_tsdf_state* sp, const float* M, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp != nullptr && M != nullptr);
    GPS_REQUIRE(state_valid(*sp));
    TsdfState s = *sp;
    // a fixed grid of waves strides over the visible list, one block per wave at a time; measured on the bench scene (46 k visible
    // blocks, us per launch): 1024 workgroups 71, 2048: 67, 4096: 63, 8192: 63.5, 16384: 66
    // the fast exact divisions need mu's significand not to be all ones (div_known_safe); any other mu takes the generic path
#ifndef GPS_INTEGRATE_WGS
#define GPS_INTEGRATE_WGS 4096
#endif
    if (div_known_safe(s.mu)) gps::launch_kernel(gps::TK_INTEGRATE, 0, integrate_kernel<true>, dim3(GPS_INTEGRATE_WGS), dim3(256), 0, (hipStream_t)stream, s, load_mat(M));
    else gps::launch_kernel(gps::TK_INTEGRATE, 0, integrate_kernel<false>, dim3(GPS_INTEGRATE_WGS), dim3(256), 0, (hipStream_t)stream, s, load_mat(M));
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"

// build-time tunables defined inside functions above (gps_build_flags)
GPS_TUNABLE_REPORT(GPS_INTEGRATE_SLICES, 4);
GPS_TUNABLE_REPORT(GPS_INTEGRATE_WGS, 4096);
