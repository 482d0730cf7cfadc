// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the trace kernel?
//
// MI355X_MICROARCH.md (HBM section) gives one calibration point -- a wide coalesced streaming read is reported at exactly half its
// bytes -- and says to calibrate every other pattern on a known byte count. The trace kernel's loads are 2-byte gathers (cube grid /
// voxel lookups), 4-byte gathers (light texels) and 32-byte records (palette entries); until round 4 its counters were doubled
// wholesale, "an upper bound for our narrow gathers" (VERDICT r03 weak 10, next 7c). Each kernel below touches a buffer far larger
// than the 256 MiB Infinity Cache exactly once, in a pattern whose distinct 64-byte sectors and 128-byte lines are known:
//
//   stream16    16 B per lane, coalesced                    (the guide's point: expect raw = bytes / 2)
//   dense2      2 B per lane, consecutive                   (128 B per wave)
//   gather2_64  one u16 per 64-byte sector  (stride  64 B)  (every sector distinct, two per 128-B line)
//   gather2_128 one u16 per 128-byte line   (stride 128 B)  (every line distinct)
//   gather2_256 one u16 per 256 B           (stride 256 B)  (every second line)
//   gather2_rnd one u16 at a hashed index over the whole buffer (distinct lines with overwhelming probability)
//   gather4_rnd one u32 likewise
//   store4      4 B per lane, coalesced stores              (WRITE_SIZE)
//
// Run under the counters (tools/measure_fetch_calib.sh):
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench/fetch_calib
// It prints the number of elements and the nominal bytes of every pattern; tools/fetch_calib_table.py joins them with the counter CSVs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void stream16(const uint4 *p, size_t n, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) *sink = a;
}
// (the u16 kernels compare with a value 16 bits CAN hold: with 0x12345678 the compiler proves the store dead and drops the loads --
//  the first run of this file "measured" 1.3 us kernels that way)
__global__ void dense2(const uint16_t *p, size_t n, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i];
    if (a == 0x5a5au) *sink = a;
}
// one u16 every `stride_elems` elements: lane i of the grid reads element i * stride_elems
template <int STRIDE_BYTES>
__global__ void gather2_stride(const uint16_t *p, size_t n_reads, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n_reads; i += (size_t)gridDim.x * blockDim.x) a ^= p[i * (size_t)(STRIDE_BYTES / 2)];
    if (a == 0x5a5au) *sink = a;
}
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void gather2_rnd(const uint16_t *p, size_t n_reads, size_t n_elems, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n_reads; i += (size_t)gridDim.x * blockDim.x) a ^= p[mix64(i) % n_elems];
    if (a == 0x5a5au) *sink = a;
}
__global__ void gather4_rnd(const uint32_t *p, size_t n_reads, size_t n_elems, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n_reads; i += (size_t)gridDim.x * blockDim.x) a ^= p[mix64(i) % n_elems];
    if (a == 0x12345678u) *sink = a;
}
__global__ void store4(uint32_t *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = (size_t)4 << 30;  // 4 GiB: 16x the Infinity Cache
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    const size_t n2 = bytes / 2;
    const size_t n_rnd = (size_t)1 << 24;  // 16.7 M random reads over 2 G elements: repeated lines are ~0.4 % of the reads
    // name, reads, nominal bytes moved if every read pulled: its own bytes / a 64-B sector / a 128-B line
    std::printf("pattern,reads,elem_bytes,distinct_64B_sectors,distinct_128B_lines\n");
    hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink);
    std::printf("stream16,%zu,16,%zu,%zu\n", bytes / 16, bytes / 64, bytes / 128);
    hipLaunchKernelGGL(dense2, grid, block, 0, 0, (const uint16_t *)buf, n2 / 4, sink);  // the first GiB
    std::printf("dense2,%zu,2,%zu,%zu\n", n2 / 4, bytes / 4 / 64, bytes / 4 / 128);
    hipLaunchKernelGGL(gather2_stride<64>, grid, block, 0, 0, (const uint16_t *)buf, bytes / 64, sink);
    std::printf("gather2_64,%zu,2,%zu,%zu\n", bytes / 64, bytes / 64, bytes / 128);
    hipLaunchKernelGGL(gather2_stride<128>, grid, block, 0, 0, (const uint16_t *)buf, bytes / 128, sink);
    std::printf("gather2_128,%zu,2,%zu,%zu\n", bytes / 128, bytes / 128, bytes / 128);
    hipLaunchKernelGGL(gather2_stride<256>, grid, block, 0, 0, (const uint16_t *)buf, bytes / 256, sink);
    std::printf("gather2_256,%zu,2,%zu,%zu\n", bytes / 256, bytes / 256, bytes / 256);
    hipLaunchKernelGGL(gather2_rnd, grid, block, 0, 0, (const uint16_t *)buf, n_rnd, n2, sink);
    std::printf("gather2_rnd,%zu,2,%zu,%zu\n", n_rnd, n_rnd, n_rnd);
    hipLaunchKernelGGL(gather4_rnd, grid, block, 0, 0, (const uint32_t *)buf, n_rnd, bytes / 4, sink);
    std::printf("gather4_rnd,%zu,4,%zu,%zu\n", n_rnd, n_rnd, n_rnd);
    hipLaunchKernelGGL(store4, grid, block, 0, 0, (uint32_t *)buf, bytes / 4 / 4);  // the first GiB
    std::printf("store4,%zu,4,%zu,%zu\n", bytes / 4 / 4, bytes / 4 / 64, bytes / 4 / 128);
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
