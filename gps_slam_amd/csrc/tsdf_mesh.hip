// Marching-cubes meshing of the voxel-block hash scene.
//   ITMMeshingEngine_CUDA.tcu:43-80, 101-134  (findAllocateBlocks + meshScene_device)
//   ITMMeshingEngine_Shared.h:279-471         (findPointNeighbors, sdfInterp, buildVertList)
//   ITMMesh.h:18-21                           (Triangle = p0 p1 p2 c0 c1 c2 clr, 21 floats)
//
// The reference's CUDA engine appends triangles with one global atomicAdd each, so their order changes from run to run;
// its CPU engine emits them in (hash entry, z, y, x, case-table) order.  Here the order is the CPU engine's, every run:
//   1. ordered list of allocated hash entries (count per 1024 slots -> scan -> write),
//   2. one workgroup per allocated block stages the 9x9x9 voxel neighbourhood (own block + the 7 blocks behind its +x/+y/+z
//      faces, resolved by 8 hash walks per block instead of 8 per voxel) in LDS and counts its triangles,
//   3. exclusive scan of the per-block counts,
//   4. the same workgroups run again, scan their 512 per-voxel counts and write the triangles at their final positions.
// Nothing synchronises with the host; the triangle count stays on the device.  Float arithmetic follows the CPU engine
// operation by operation (this file is built with -ffp-contract=off like the other TSDF kernels): positions and colours
// are bit-equal to the restatement in oracle/tsdf_oracle.c, which is bit-equal to the reference CPU engine.
#include "tsdf_common.hpp"

using namespace gpst;

namespace {

#include "mc_cases.inc"
__device__ const unsigned long long mc_cases_dev[256] = GPS_MC_CASES_INIT;

constexpr int SLOTS = 1024;   // hash slots per workgroup of the list sweeps
constexpr int NB = 9;         // staged neighbourhood edge
constexpr int NB3 = NB * NB * NB;
constexpr int STAGE = 256;    // triangles per LDS staging round of the emit pass (21.5 KB)

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

// exclusive scan over the NT threads of the workgroup; ws needs NT/64 + 1 ints
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int* ws, int& total) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int sv = lane < NW ? ws[lane] : 0;
        const int si = wave_incl_scan(sv);
        if (lane < NW) ws[lane] = si - sv;
        if (lane == 63) ws[NW] = si;
    }
    __syncthreads();
    const int r = ws[wave] + incl - v;
    total = ws[NW];
    __syncthreads();
    return r;
}

// ---- 1. ordered list of allocated entries
__global__ __launch_bounds__(SLOTS) void mesh_entries_count_kernel(TsdfState s, int32_t* __restrict__ blk) {
    __shared__ int ws[17];
    const int idx = blockIdx.x * SLOTS + threadIdx.x;
    const int n_total = s.n_buckets + s.n_excess;
    const int has = idx < n_total && s.hash[idx].ptr >= 0;
    int tot;
    block_excl_scan<SLOTS>(has, ws, tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// single workgroup: in-place exclusive scan of n counts, total to *out_total
__global__ __launch_bounds__(1024) void mesh_scan_kernel(const int32_t* __restrict__ n_ptr, int n_fixed, int32_t* __restrict__ arr,
                                                        int32_t* __restrict__ out_total) {
    __shared__ int ws[17];
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int sum = 0;
    for (int k = lo; k < hi; k++) sum += arr[k];
    int total;
    int run = block_excl_scan<1024>(sum, ws, total);
    for (int k = lo; k < hi; k++) { const int v = arr[k]; arr[k] = run; run += v; }
    if (threadIdx.x == 0) *out_total = total;
}

__global__ __launch_bounds__(SLOTS) void mesh_entries_write_kernel(TsdfState s, const int32_t* __restrict__ blk,
                                                                  int32_t* __restrict__ list) {
    __shared__ int ws[17];
    const int idx = blockIdx.x * SLOTS + threadIdx.x;
    const int n_total = s.n_buckets + s.n_excess;
    const int has = idx < n_total && s.hash[idx].ptr >= 0;
    int tot;
    const int pos = blk[blockIdx.x] + block_excl_scan<SLOTS>(has, ws, tot);
    if (has && pos < s.n_blocks) list[pos] = idx;
}

// ---- 2./4. per-block marching cubes
// findVoxel's hash walk (ITMRepresentationAccess.h:39-57): voxel-block pointer of block (bx,by,bz), or -1
__device__ __forceinline__ int find_block(const TsdfState& s, int bx, int by, int bz) {
    int idx = hash_index(bx, by, bz, s.n_buckets - 1);
    while (true) {
        const HashEntry e = load_entry(s.hash, idx);
        if (entry_is(e, bx, by, bz) && e.ptr >= 0) return e.ptr;
        if (e.offset < 1) return -1;
        idx = s.n_buckets + e.offset - 1;
    }
}

// corner k of the cube at (x,y,z): offsets in the reference's order (Shared.h:286-354)
__device__ __forceinline__ int corner_dx(int k) { return (0x66 >> k) & 1; }  // k = 1,2,5,6
__device__ __forceinline__ int corner_dy(int k) { return (0xCC >> k) & 1; }  // k = 2,3,6,7
__device__ __forceinline__ int corner_dz(int k) { return (0xF0 >> k) & 1; }  // k = 4..7
// end points of edge e, nibble-packed: 0-1 1-2 2-3 3-0 4-5 5-6 6-7 7-4 0-4 1-5 2-6 3-7
__device__ __forceinline__ int edge_a(int e) { return (int)((0x321076543210ull >> (4 * e)) & 0xF); }
__device__ __forceinline__ int edge_b(int e) { return (int)((0x765447650321ull >> (4 * e)) & 0xF); }

struct Corner { float px, py, pz, sdf, r, g, b; };

// sdfInterp (Shared.h:359-369) on one component pair
__device__ __forceinline__ float lerp_or_pick(float a, float b, int pick, float f) {
    return pick == 1 ? a : (pick == 2 ? b : a + f * (b - a));
}

template <bool EMIT>
__global__ __launch_bounds__(512) void mesh_block_kernel(TsdfState s, const int32_t* __restrict__ list,
                                                        const int32_t* __restrict__ n_list_ptr, int32_t* __restrict__ tri_count,
                                                        const int32_t* __restrict__ tri_base, int64_t max_triangles,
                                                        float* __restrict__ triangles) {
    __shared__ uint2 vox[NB3];       // raw voxels of the 9^3 neighbourhood
    __shared__ uint8_t present[NB3]; // the block holding that voxel exists (readVoxel's vmIndex != 0)
    __shared__ int nb_ptr[8];
    __shared__ int ws[9];
    __shared__ float stage[EMIT ? STAGE * 21 : 1];
    __shared__ int off[EMIT ? 512 : 1];
    __shared__ unsigned long long cs[EMIT ? 512 : 1];
    const int tid = threadIdx.x;
    const int n_list = min(*n_list_ptr, s.n_blocks);
    for (int b = blockIdx.x; b < n_list; b += gridDim.x) {
        const HashEntry he = load_entry(s.hash, list[b]);
        if (tid < 8) nb_ptr[tid] = tid == 0 ? he.ptr : find_block(s, he.x + (tid & 1), he.y + ((tid >> 1) & 1), he.z + (tid >> 2));
        __syncthreads();
        for (int q = tid; q < NB3; q += 512) {
            const int i = q % NB, j = (q / NB) % NB, k = q / (NB * NB);
            const int sel = (i >> 3) | ((j >> 3) << 1) | ((k >> 3) << 2);
            const int ptr = nb_ptr[sel];
            uint2 v = make_uint2(0u, 0u);
            if (ptr >= 0) v = reinterpret_cast<const uint2*>(s.vba)[(size_t)ptr * BLK3 + (i & 7) + (j & 7) * BLK + (k & 7) * BLK * BLK];
            vox[q] = v;
            present[q] = ptr >= 0;
        }
        __syncthreads();
        const int x = tid & 7, y = (tid >> 3) & 7, z = tid >> 6;
        const int base = x + y * NB + z * NB * NB;
        // findPointNeighbors: all 8 corners must exist and be touched (sdf != 1.0f <=> raw sdf != 32767)
        bool ok = true;
        int cube = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int q = base + corner_dx(k) + corner_dy(k) * NB + corner_dz(k) * NB * NB;
            const int raw = (int)(int16_t)(vox[q].x & 0xFFFFu);
            ok = ok && present[q] && raw != 32767;
            cube |= (raw < 0) << k;  // valueToFloat(sdf) < 0 <=> sdf < 0
        }
        const unsigned long long cases = ok ? mc_cases_dev[cube] : ~0ull;
        int n_tri = 0;
#pragma unroll
        for (int j = 0; j < 15; j += 3) n_tri += ((cases >> (4 * j)) & 0xF) != 0xF;
        int total;
        const int first = block_excl_scan<512>(n_tri, ws, total);
        if (!EMIT) {
            if (tid == 0) tri_count[b] = total;
        } else {
            // Dense emit.  Only a few of the 512 cubes of a block cut the surface (85 triangles per block on the bench scene),
            // so a thread-per-cube loop leaves most lanes idle behind the busiest one (first version: 660 us).  Instead the
            // per-cube offsets and case words go to LDS and the block's 3 * total triangle VERTICES are dealt to consecutive
            // threads (owner cube by binary search in the offsets); vertices are staged in LDS and copied out with
            // consecutive lanes on consecutive dwords -- the block's triangles are contiguous in the output.
            off[tid] = first;
            cs[tid] = cases;
            const int64_t out_base = (int64_t)tri_base[b];
            const int64_t room = max_triangles - 1 - out_base;  // the CPU engine stops advancing at noMaxTriangles - 1
            const int emit_total = (int)(room <= 0 ? 0 : (room < total ? room : total));
            __syncthreads();
            for (int r0 = 0; r0 < emit_total; r0 += STAGE) {
                const int n_round = min(STAGE, emit_total - r0);
                for (int q = tid; q < 3 * n_round; q += 512) {
                    const int tri = r0 + q / 3, c = q % 3;
                    int lo = 0, hi = 511;  // last cube whose first triangle index is <= tri (cubes without triangles repeat the offset)
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (off[mid] <= tri) lo = mid; else hi = mid - 1;
                    }
                    const int v = lo, t = tri - off[v];
                    const unsigned long long vc = cs[v];
                    const int vx = v & 7, vy = (v >> 3) & 7, vz = v >> 6;
                    const int vbase = vx + vy * NB + vz * NB * NB;
                    const int gx = he.x * BLK + vx, gy = he.y * BLK + vy, gz = he.z * BLK + vz;
                    float* o = stage + (tri - r0) * 21;
                    const int e = (int)((vc >> (4 * (3 * t + c))) & 0xF);
                    const int ka = edge_a(e), kb = edge_b(e);
                    const int qa = vbase + corner_dx(ka) + corner_dy(ka) * NB + corner_dz(ka) * NB * NB;
                    const int qb = vbase + corner_dx(kb) + corner_dy(kb) * NB + corner_dz(kb) * NB * NB;
                    const uint2 va = vox[qa], vb = vox[qb];
                    const float sa = (float)(int16_t)(va.x & 0xFFFFu) / 32767.0f, sb = (float)(int16_t)(vb.x & 0xFFFFu) / 32767.0f;
                    // sdfInterp: |v1| < 1e-5 -> p1; |v2| < 1e-5 -> p2; |v1 - v2| < 1e-5 -> p1; else p1 + (-v1 / (v2 - v1)) (p2 - p1)
                    const int pick = fabsf(0.0f - sa) < 0.00001f ? 1 : (fabsf(0.0f - sb) < 0.00001f ? 2 : (fabsf(sa - sb) < 0.00001f ? 1 : 0));
                    const float f = (0.0f - sa) / (sb - sa);
                    const float pax = (float)(gx + corner_dx(ka)), pay = (float)(gy + corner_dy(ka)), paz = (float)(gz + corner_dz(ka));
                    const float pbx = (float)(gx + corner_dx(kb)), pby = (float)(gy + corner_dy(kb)), pbz = (float)(gz + corner_dz(kb));
                    o[3 * c + 0] = lerp_or_pick(pax, pbx, pick, f) * s.voxel_size;
                    o[3 * c + 1] = lerp_or_pick(pay, pby, pick, f) * s.voxel_size;
                    o[3 * c + 2] = lerp_or_pick(paz, pbz, pick, f) * s.voxel_size;
                    const float ar = (float)(va.x >> 24) / 255.0f, ag = (float)(va.y & 0xFFu) / 255.0f, ab = (float)((va.y >> 8) & 0xFFu) / 255.0f;
                    const float br = (float)(vb.x >> 24) / 255.0f, bg = (float)(vb.y & 0xFFu) / 255.0f, bb = (float)((vb.y >> 8) & 0xFFu) / 255.0f;
                    o[9 + 3 * c + 0] = lerp_or_pick(ar, br, pick, f);
                    o[9 + 3 * c + 1] = lerp_or_pick(ag, bg, pick, f);
                    o[9 + 3 * c + 2] = lerp_or_pick(ab, bb, pick, f);
                    if (c == 0) {  // VoxelColorReader::uninterpolate at the cube's origin voxel
                        const uint2 v0 = vox[vbase];
                        o[18] = (float)(v0.x >> 24) / 255.0f; o[19] = (float)(v0.y & 0xFFu) / 255.0f; o[20] = (float)((v0.y >> 8) & 0xFFu) / 255.0f;
                    }
                }
                __syncthreads();
                float* dst = triangles + (out_base + r0) * 21;
                for (int k = tid; k < n_round * 21; k += 512) dst[k] = stage[k];
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

__global__ void mesh_finish_kernel(const int32_t* __restrict__ total, int64_t max_triangles, int64_t* __restrict__ counts) {
    const int64_t t = *total;
    counts[0] = t < max_triangles - 1 ? t : (max_triangles > 0 ? max_triangles - 1 : 0);  // noTotalTriangles
    counts[1] = t;                                                                         // triangles the scene holds
}

struct MeshWs { int32_t *blk, *list, *tri, *n_list, *total; };
inline int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }
inline int64_t carve_ws(MeshWs* w, char* base, const TsdfState& s) {
    const int nblk = gps_div_up(s.n_buckets + s.n_excess, SLOTS);
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off += align16(bytes); return p; };
    w->blk = (int32_t*)take((int64_t)nblk * 4);
    w->list = (int32_t*)take((int64_t)s.n_blocks * 4);
    w->tri = (int32_t*)take((int64_t)s.n_blocks * 4);
    w->n_list = (int32_t*)take(16);
    w->total = (int32_t*)take(16);
    return off;
}

}  // namespace

extern "C" {

int64_t gps_tsdf_mesh_workspace_bytes(const gps_tsdf_state* s) {
    if (!s || s->n_blocks <= 0 || s->n_buckets <= 0 || s->n_excess <= 0) return -1;
    MeshWs w;
    return carve_ws(&w, nullptr, *s);
}

int gps_tsdf_mesh_scene(const gps_tsdf_state* sp, int64_t max_triangles, float* triangles, int64_t* counts, void* workspace,
                        int64_t workspace_bytes, gps_stream stream) {
    GPS_ENTER();
    GPS_REQUIRE(sp && triangles && counts && workspace);
    GPS_REQUIRE(state_valid(*sp));
    GPS_REQUIRE(max_triangles >= 2);
    const TsdfState s = *sp;
    MeshWs w;
    if (workspace_bytes < carve_ws(&w, (char*)workspace, s)) return GPS_ERR_CAPACITY;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = gps_div_up(s.n_buckets + s.n_excess, SLOTS);
    mesh_entries_count_kernel<<<nblk, SLOTS, 0, st>>>(s, w.blk);
    mesh_scan_kernel<<<1, 1024, 0, st>>>(nullptr, nblk, w.blk, w.n_list);
    mesh_entries_write_kernel<<<nblk, SLOTS, 0, st>>>(s, w.blk, w.list);
    const int grid = 4096;  // grid-stride over the allocated blocks (their number is only known on the device)
    mesh_block_kernel<false><<<grid, 512, 0, st>>>(s, w.list, w.n_list, w.tri, nullptr, max_triangles, nullptr);
    mesh_scan_kernel<<<1, 1024, 0, st>>>(w.n_list, 0, w.tri, w.total);
    mesh_block_kernel<true><<<grid, 512, 0, st>>>(s, w.list, w.n_list, nullptr, w.tri, max_triangles, triangles);
    mesh_finish_kernel<<<1, 1, 0, st>>>(w.total, max_triangles, counts);
    GPS_LAUNCH_CHECK();
    return GPS_OK;
}

}  // extern "C"
