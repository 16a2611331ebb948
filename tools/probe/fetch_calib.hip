// What does rocprofv3's FETCH_SIZE count per memory-side read request on this gfx950?  The MI355X guide calibrates one case only
// (wide coalesced streaming reads report exactly half their bytes); tools/pmc_extract.py doubled FETCH_SIZE for every kernel, the
// gather-dominated ones (tracker evaluation, colour, raycast, expected depths) included.  This probe reads a buffer far larger than
// the 256 MiB Infinity Cache ONCE per pattern, so the bytes that must come from HBM are known:
//   stream16   16 B per lane, consecutive          (the guide's case)
//   stream4     4 B per lane, consecutive
//   gather4     4 B per lane, every lane its own 128-B line (a permutation of all lines: no line is touched twice)
//   gather8     8 B per lane, own line             (a voxel)
//   gather16   16 B per lane, own line             (a hash entry)
//   gather64   4 x 16 B = a whole 64-B half line per lane, own line
// usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -- ./fetch_calib     (tools/probe/fetch_calib.sh parses the database)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream16(const uint4* __restrict__ p, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void stream4(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) out[0] = acc;
}
// line = a permutation of [0, n_lines) (n_lines a power of two, odd multiplier); BYTES read at the start of that 128-byte line
template <int BYTES>
__global__ void gather(const uint32_t* __restrict__ p, uint32_t n_lines, uint32_t* out) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += gridDim.x * blockDim.x) {
        const uint32_t line = (i * 2654435761u + 12345u) & (n_lines - 1);
        const uint32_t* q = p + (size_t)line * 32;
        if (BYTES == 4) acc += q[0];
        else if (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2*>(q); acc += v.x ^ v.y; }
        else if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4*>(q); acc += v.x ^ v.y ^ v.z ^ v.w; }
        else { for (int k = 0; k < BYTES / 16; k++) { const uint4 v = reinterpret_cast<const uint4*>(q)[k]; acc += v.x ^ v.y ^ v.z ^ v.w; } }
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB
    const uint32_t n_lines = (uint32_t)(bytes / 128);
    void* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 16, block = 256;
    for (int rep = 0; rep < 2; rep++) {
        stream16<<<grid, block>>>((const uint4*)buf, bytes / 16, out);
        stream4<<<grid, block>>>((const uint32_t*)buf, bytes / 4, out);
        gather<4><<<grid, block>>>((const uint32_t*)buf, n_lines, out);
        gather<8><<<grid, block>>>((const uint32_t*)buf, n_lines, out);
        gather<16><<<grid, block>>>((const uint32_t*)buf, n_lines, out);
        gather<64><<<grid, block>>>((const uint32_t*)buf, n_lines, out);
        CK(hipDeviceSynchronize());
    }
    printf("{\"bytes\": %zu, \"lines\": %u}\n", bytes, n_lines);
    return 0;
}
