// Micro-benchmarks used to calibrate the in-kernel cycle counter and per-instruction costs on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ void k(double *out, uint32_t *ticks, int iters, const uint16_t *pool, uint32_t mask) {
    double a = threadIdx.x * 1e-9 + 1.0, b = 1.0000001, c = 0.5;
    float fa = threadIdx.x * 1e-3f + 1.0f, fb = 1.0001f;
    uint32_t idx = threadIdx.x * 977u + blockIdx.x * 131u;
    uint32_t t0 = (uint32_t)__builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 16 dependent f64 adds
#pragma unroll
            for (int j = 0; j < 16; j++) a = a + b;
        } else if (MODE == 1) {  // 16 dependent f32 adds
#pragma unroll
            for (int j = 0; j < 16; j++) fa = fa + fb;
        } else if (MODE == 2) {  // 16 dependent (cmp f64 + cndmask pair)
#pragma unroll
            for (int j = 0; j < 16; j++) { a = (a < c) ? b : a + 1e-30; c = c + 1.0; }
        } else if (MODE == 3) {  // dependent u16 loads (pointer chase through a table)
#pragma unroll
            for (int j = 0; j < 4; j++) idx = (idx * 31u + pool[idx & mask]) ;
        } else if (MODE == 4) {  // 16 independent f64 adds (4 chains)
            double a1 = a + 1, a2 = a + 2, a3 = a + 3;
#pragma unroll
            for (int j = 0; j < 4; j++) { a = a + b; a1 = a1 + b; a2 = a2 + b; a3 = a3 + b; }
            a = a + a1 + a2 + a3;
        }
    }
    uint32_t t1 = (uint32_t)__builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + fa + idx;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int threads, int iters, int per_iter, const uint16_t *pool, uint32_t mask) {
    double *out; uint32_t *ticks;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&ticks, 4 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, ticks, 10, pool, mask);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters, pool, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(blocks);
    hipMemcpy(h.data(), ticks, 4 * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-34s blocks %4d x %3d thr: %8.3f ms, %10.0f ticks/wave (%.3f GHz tick rate), %.2f ticks per op, %.2f ns per op\n", name, blocks,
           threads, ms, avg, avg / (ms * 1e6), avg / ((double)iters * per_iter), ms * 1e6 / ((double)iters * per_iter));
    hipFree(out); hipFree(ticks);
}

int main() {
    const uint32_t n = 1u << 19;  // 1 MB of u16: L2-resident
    std::vector<uint16_t> hp(n);
    for (uint32_t i = 0; i < n; i++) hp[i] = (uint16_t)(i * 2654435761u >> 13);
    uint16_t *pool; hipMalloc(&pool, n * 2); hipMemcpy(pool, hp.data(), n * 2, hipMemcpyHostToDevice);
    const int it = 20000;
    run<0>("dep f64 add, 1 wave/SIMD", 1, 64, it, 16, pool, n - 1);
    run<1>("dep f32 add, 1 wave", 1, 64, it, 16, pool, n - 1);
    run<2>("dep cmp_f64+cndmask+add, 1 wave", 1, 64, it, 16, pool, n - 1);
    run<4>("4 indep chains f64 add, 1 wave", 1, 64, it, 16, pool, n - 1);
    run<0>("dep f64 add, 2 waves/SIMD all CUs", 512, 256, it, 16, pool, n - 1);
    run<0>("dep f64 add, 4 waves/SIMD all CUs", 1024, 256, it, 16, pool, n - 1);
    run<3>("dep u16 load 1MB table, 1 wave", 1, 64, it / 4, 4, pool, n - 1);
    run<3>("dep u16 load 32KB table, 1 wave", 1, 64, it / 4, 4, pool, (1u << 14) - 1);
    run<3>("dep u16 load 1MB, 2 waves/SIMD all", 512, 256, it / 4, 4, pool, n - 1);
    return 0;
}
