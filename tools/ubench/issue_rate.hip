// issue_rate.hip -- throughput of the instruction mix of trace_image_kernel's DDA step on gfx950.
//
// the round-1 "VALU-issue-bound" claim priced a wave64 VALU instruction at 4 cycles,
// MI355X_MICROARCH.md prices v_fma_f32 at 2. This bench measures, per instruction kind, the cycles one SIMD
// spends per wave-instruction when 1/2/3/4/8 waves per SIMD issue INDEPENDENT instances of it (8 chains,
// so no dependency stall), on every CU at once. Output feeds bench.py's valu_issue.peak.
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

enum Op { ADD_U32, CNDMASK, CMP_F64, ADD_F64, FMA_F32, MOV_B32, MAD_U64_U32, MUL_LO_U32, MAD_U32_U24, MIN3_U32, SALU_AND,
          READLANE, LDS_READ_B32, LDS_READ_U16, MIX_DDA, MOV_B64, LSHL_ADD_U64, CMP_U32, VS_MIX, VS_MIX_CMP, N_OPS };
static const char *kNames[N_OPS] = {"v_add_u32", "v_cndmask_b32", "v_cmp_lt_f64", "v_add_f64", "v_fma_f32", "v_mov_b32",
                                    "v_mad_u64_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_min3_u32", "s_and_b64", "v_readlane_b32",
                                    "ds_read_b32", "ds_read_u16", "mix(3cmp64,3add64,8cnd,8int)", "v_mov_b64", "v_lshl_add_u64",
                                    "v_cmp_eq_u32", "16 v_add_f64 + 16 s_and_b64", "16 v_cmp_lt_f64 + 16 s_and_b64"};
// instructions per inner block (each block is 32 instructions, except MIX = 22)
static const int kPerBlock[N_OPS] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 22, 32, 32, 32, 32, 32};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t *ticks, int iters) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t c = blockIdx.x | 1u;
    double d0 = 1.0 + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    double dc = 1e-9;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fc = 1.0001f;
    unsigned long long s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    uint32_t la = (threadIdx.x * 4u) & 16383u;
    const uint32_t t0 = (uint32_t)__builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (OP == ADD_U32) {
            asm volatile(REP4("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == CNDMASK) {
            asm volatile(REP4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
        } else if (OP == CMP_F64) {
            asm volatile(REP4("v_cmp_lt_f64 %0, %4, %5\n v_cmp_lt_f64 %1, %5, %6\n v_cmp_lt_f64 %2, %6, %7\n v_cmp_lt_f64 %3, %7, %4\n"
                              "v_cmp_lt_f64 %0, %4, %6\n v_cmp_lt_f64 %1, %5, %7\n v_cmp_lt_f64 %2, %6, %4\n v_cmp_lt_f64 %3, %7, %5\n")
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        } else if (OP == ADD_F64) {
            asm volatile(REP4("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));
        } else if (OP == FMA_F32) {
            asm volatile(REP4("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fc));
        } else if (OP == MOV_B32) {
            asm volatile(REP4("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                              "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == MOV_B64) {
            asm volatile(REP4("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                              "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));
        } else if (OP == MAD_U64_U32) {
            unsigned long long m0 = a0, m1 = a1, m2 = a2, m3 = a3;
            asm volatile(REP8("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(c), "v"(a7) : "vcc");
            a0 = (uint32_t)m0; a1 = (uint32_t)m1; a2 = (uint32_t)m2; a3 = (uint32_t)m3;
        } else if (OP == MUL_LO_U32) {
            asm volatile(REP4("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == MAD_U32_U24) {
            asm volatile(REP4("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n"
                              "v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == MIN3_U32) {
            asm volatile(REP4("v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %1, %1, %8, %2\n v_min3_u32 %2, %2, %8, %3\n v_min3_u32 %3, %3, %8, %4\n"
                              "v_min3_u32 %4, %4, %8, %5\n v_min3_u32 %5, %5, %8, %6\n v_min3_u32 %6, %6, %8, %7\n v_min3_u32 %7, %7, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == CMP_U32) {
            asm volatile(REP4("v_cmp_eq_u32 %0, %4, %5\n v_cmp_eq_u32 %1, %5, %6\n v_cmp_eq_u32 %2, %6, %7\n v_cmp_eq_u32 %3, %7, %4\n"
                              "v_cmp_eq_u32 %0, %4, %6\n v_cmp_eq_u32 %1, %5, %7\n v_cmp_eq_u32 %2, %6, %4\n v_cmp_eq_u32 %3, %7, %5\n")
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if (OP == LSHL_ADD_U64) {
            asm volatile(REP4("v_lshl_add_u64 %0, %0, 1, %8\n v_lshl_add_u64 %1, %1, 1, %8\n v_lshl_add_u64 %2, %2, 1, %8\n v_lshl_add_u64 %3, %3, 1, %8\n"
                              "v_lshl_add_u64 %4, %4, 1, %8\n v_lshl_add_u64 %5, %5, 1, %8\n v_lshl_add_u64 %6, %6, 1, %8\n v_lshl_add_u64 %7, %7, 1, %8\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));
        } else if (OP == SALU_AND) {
            asm volatile(REP8("s_and_b64 %0, %0, %1\n s_and_b64 %1, %1, %2\n s_and_b64 %2, %2, %3\n s_and_b64 %3, %3, %0\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (OP == READLANE) {
            asm volatile(REP8("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9\n")
                         : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if (OP == LDS_READ_B32) {
            asm volatile(REP4("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                              "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(la) : "memory");
        } else if (OP == LDS_READ_U16) {
            // data-dependent (gather) 2-byte reads: the address of the next comes from the previous value
            asm volatile(REP4("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:2\n ds_read_u16 %2, %8 offset:4\n ds_read_u16 %3, %8 offset:6\n"
                              "ds_read_u16 %4, %8 offset:258\n ds_read_u16 %5, %8 offset:514\n ds_read_u16 %6, %8 offset:770\n ds_read_u16 %7, %8 offset:1026\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(la) : "memory");
            la = (la + (a0 & 1022u)) & 8190u;
        } else if (OP == VS_MIX) {
            // does scalar work issue beside vector work (different waves), or do both share one issue slot per SIMD?
            asm volatile(REP4("v_add_f64 %0, %0, %8\n s_and_b64 %9, %9, %10\n v_add_f64 %1, %1, %8\n s_and_b64 %10, %10, %11\n"
                              "v_add_f64 %2, %2, %8\n s_and_b64 %11, %11, %12\n v_add_f64 %3, %3, %8\n s_and_b64 %12, %12, %9\n")
                         REP4("v_add_f64 %4, %4, %8\n s_and_b64 %9, %9, %10\n v_add_f64 %5, %5, %8\n s_and_b64 %10, %10, %11\n"
                              "v_add_f64 %6, %6, %8\n s_and_b64 %11, %11, %12\n v_add_f64 %7, %7, %8\n s_and_b64 %12, %12, %9\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "scc");
        } else if (OP == VS_MIX_CMP) {
            asm volatile(REP4("v_cmp_lt_f64 vcc, %0, %1\n s_and_b64 %4, %4, vcc\n v_cmp_lt_f64 vcc, %1, %2\n s_and_b64 %5, %5, vcc\n"
                              "v_cmp_lt_f64 vcc, %2, %3\n s_and_b64 %6, %6, vcc\n v_cmp_lt_f64 vcc, %3, %0\n s_and_b64 %7, %7, vcc\n")
                         : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "vcc", "scc");
        } else if (OP == MIX_DDA) {
            // the arithmetic skeleton of one DDA step: 3 f64 compares, 3 f64 adds, 8 selects, 8 integer ops
            asm volatile("v_cmp_lt_f64 %12, %0, %1\n v_cmp_lt_f64 %13, %0, %2\n v_cmp_lt_f64 vcc, %1, %2\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, %12\n v_cndmask_b32 %6, %6, %7, %13\n v_cndmask_b32 %7, %7, %4, vcc\n"
                         "v_add_f64 %0, %0, %3\n v_add_f64 %1, %1, %3\n v_add_f64 %2, %2, %3\n"
                         "v_cndmask_b32 %8, %8, %9, vcc\n v_cndmask_b32 %9, %9, %10, %12\n v_cndmask_b32 %10, %10, %11, %13\n v_cndmask_b32 %11, %11, %8, vcc\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %10\n v_add_u32 %7, %7, %11\n"
                         "v_and_b32 %8, %8, %4\n v_or_b32 %9, %9, %5\n v_xor_b32 %10, %10, %6\n v_lshrrev_b32 %11, 1, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(dc), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "=s"(s0), "=s"(s1) : : "vcc");
        }
    }
    const uint32_t t1 = (uint32_t)__builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) +
                                                 (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + (uint32_t)(s0 + s1 + s2 + s3) + r0 + r1 + r2 + r3;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(int n_cus, int waves_per_simd, int iters) {
    const int blocks = n_cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    uint32_t *out, *ticks;
    hipMalloc(&out, sizeof(uint32_t) * blocks * 256);
    hipMalloc(&ticks, 16 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, ticks, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(blocks * 4);
    hipMemcpy(h.data(), ticks, 16 * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= (double)h.size();
    const double n_inst = (double)iters * kPerBlock[OP];
    // ticks one wave needed per instruction, and what the SIMD spent per wave-instruction with W waves sharing it
    printf("%-30s W=%d  %8.3f ms  %6.2f ticks/inst/wave  %6.2f cyc/inst/SIMD (ticks)  %6.2f cyc/inst/SIMD (wall @2.4GHz)\n", kNames[OP],
           waves_per_simd, ms, avg / n_inst, avg / n_inst / waves_per_simd, ms * 1e-3 * 2.4e9 / n_inst / waves_per_simd);
    hipFree(out);
    hipFree(ticks);
}

template <int OP>
void sweep(int n_cus) {
    const int ws[] = {1, 2, 3, 4, 8};
    for (int w : ws) run<OP>(n_cus, w, 4000);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cus = p.multiProcessorCount;
    printf("# %s, %d CUs, clock %d kHz; 256-thread blocks (one wave per SIMD), W blocks per CU\n", p.name, n_cus, p.clockRate);
    sweep<ADD_U32>(n_cus);
    sweep<CNDMASK>(n_cus);
    sweep<MOV_B32>(n_cus);
    sweep<MOV_B64>(n_cus);
    sweep<CMP_U32>(n_cus);
    sweep<CMP_F64>(n_cus);
    sweep<ADD_F64>(n_cus);
    sweep<FMA_F32>(n_cus);
    sweep<MAD_U32_U24>(n_cus);
    sweep<MUL_LO_U32>(n_cus);
    sweep<MAD_U64_U32>(n_cus);
    sweep<LSHL_ADD_U64>(n_cus);
    sweep<MIN3_U32>(n_cus);
    sweep<SALU_AND>(n_cus);
    sweep<READLANE>(n_cus);
    sweep<LDS_READ_B32>(n_cus);
    sweep<LDS_READ_U16>(n_cus);
    sweep<MIX_DDA>(n_cus);
    sweep<VS_MIX>(n_cus);
    sweep<VS_MIX_CMP>(n_cus);
    return 0;
}
