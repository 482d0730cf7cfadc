// issue_rate.hip -- instruction issue rates of one gfx950 SIMD, measured with the residency PINNED and VERIFIED.
//
// round 2's version launched "W 256-thread blocks per CU" and never checked where the blocks landed;
// its W sweep was not monotone and v_fma_f32 came out at 3.8 cycles where MI355X_MICROARCH.md says 2. This version
//   * launches ONE workgroup per CU: W * 256 threads (W waves on each of the CU's 4 SIMDs), with an LDS allocation of more
//     than half of the CU's 160 KB so that a second workgroup cannot share the CU (W = 8: two 1024-thread workgroups with
//     64 KB each);
//   * records HW_REG_HW_ID / XCC_ID of every wave and prints the histogram of waves per SIMD it actually got: a sweep line
//     is only meaningful if that histogram is a single spike at W;
//   * times each wave with s_memtime (shader clock) AND s_memrealtime (constant 100 MHz), and the launch with HIP events, so
//     the effective shader clock of the run is part of the output instead of an assumed 2.4 GHz;
//   * runs >= 2 ms per launch, 8 independent dependency chains per wave, loop overhead < 1 %.
// Output: cycles one SIMD spends per wave-instruction (shader cycles / (instructions per wave * waves on the SIMD)).
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)

enum Op { CND_VCC_E64, CMP_VCC_CND, ADDC_VCC, MOV_B32, ADD_U32, FMA_F32, PK_FMA_F32, ADD_F64, FMA_F64, CMP_F64, CNDMASK_VCC, CNDMASK_SGPR, MUL_LO_U32, SALU_AND, VALU_SALU, CMP_SALU,
          CMP_NOP_CND, DS_READ_B32, DDA_STEP, N_OPS };
static const char *kNames[N_OPS] = {"v_cndmask_b32_e64 (vcc named as the mask operand)", "v_cmp_eq_u32 vcc + v_cndmask vcc (compiler's e32 idiom)", "v_addc_co_u32 (vcc out, sgpr-pair carry in)", "v_mov_b32", "v_add_u32", "v_fma_f32", "v_pk_fma_f32", "v_add_f64", "v_fma_f64", "v_cmp_lt_f64 -> sgpr",
                                    "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr pair)", "v_mul_lo_u32", "s_and_b64", "v_add_f64 + s_and_b64 alternating",
                                    "v_cmp_lt_f64 vcc + s_and_b64 vcc (dependent pair)", "v_cmp_eq_u32 + s_nop 1 + v_cndmask (select idiom)",
                                    "ds_read_b32 (8 in flight, conflict-free)", "fast DDA step of trace_image_kernel (28 inst: 18 VALU, 10 SALU)"};
static const int kPerBlock[N_OPS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 72, 64, 28};

struct WaveRec {
    uint32_t hw_id, xcc_id;
    uint64_t t_shader, t_real;
};

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, WaveRec *rec, int iters) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t c = blockIdx.x | 1u;
    double d0 = 1.0 + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    double dc = 1e-9;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fc = 1.0001f;
    unsigned long long s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const uint32_t la = (threadIdx.x * 4u) & 16383u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (OP == CND_VCC_E64) {
            asm volatile("s_mov_b64 vcc, %9\n"
                         REP8("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(s0) : "vcc");
        } else if (OP == CMP_VCC_CND) {
            asm volatile(REP8("v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %0, %0, %6, vcc\n v_cmp_eq_u32 vcc, %5, %6\n v_cndmask_b32 %1, %1, %7, vcc\n"
                              "v_cmp_eq_u32 vcc, %6, %7\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_eq_u32 vcc, %7, %4\n v_cndmask_b32 %3, %3, %5, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (OP == ADDC_VCC) {
            asm volatile(REP8("v_addc_co_u32 %0, vcc, 0, %0, %8\n v_addc_co_u32 %1, vcc, 0, %1, %8\n v_addc_co_u32 %2, vcc, 0, %2, %8\n v_addc_co_u32 %3, vcc, 0, %3, %8\n v_addc_co_u32 %4, vcc, 0, %4, %8\n v_addc_co_u32 %5, vcc, 0, %5, %8\n v_addc_co_u32 %6, vcc, 0, %6, %8\n v_addc_co_u32 %7, vcc, 0, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0) : "vcc");
        } else if (OP == MOV_B32) {
            asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == ADD_U32) {
            asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == FMA_F32) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fc));
        } else if (OP == PK_FMA_F32) {
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));
        } else if (OP == ADD_F64) {
            asm volatile(REP8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));
        } else if (OP == FMA_F64) {
            asm volatile(REP8("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));
        } else if (OP == CMP_F64) {
            asm volatile(REP8("v_cmp_lt_f64 %0, %4, %5\n v_cmp_lt_f64 %1, %5, %6\n v_cmp_lt_f64 %2, %6, %7\n v_cmp_lt_f64 %3, %7, %4\n v_cmp_lt_f64 %0, %4, %6\n v_cmp_lt_f64 %1, %5, %7\n v_cmp_lt_f64 %2, %6, %4\n v_cmp_lt_f64 %3, %7, %5\n")
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        } else if (OP == CNDMASK_VCC) {
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == CNDMASK_SGPR) {
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(s0));
        } else if (OP == MUL_LO_U32) {
            asm volatile(REP8("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == SALU_AND) {
            asm volatile(REP8(REP2("s_and_b64 %0, %0, %1\n s_and_b64 %1, %1, %2\n s_and_b64 %2, %2, %3\n s_and_b64 %3, %3, %0\n"))
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (OP == VALU_SALU) {
            asm volatile(REP4("v_add_f64 %0, %0, %8\n s_and_b64 %9, %9, %10\n v_add_f64 %1, %1, %8\n s_and_b64 %10, %10, %11\n v_add_f64 %2, %2, %8\n s_and_b64 %11, %11, %12\n v_add_f64 %3, %3, %8\n s_and_b64 %12, %12, %9\n"
                              "v_add_f64 %4, %4, %8\n s_and_b64 %9, %9, %10\n v_add_f64 %5, %5, %8\n s_and_b64 %10, %10, %11\n v_add_f64 %6, %6, %8\n s_and_b64 %11, %11, %12\n v_add_f64 %7, %7, %8\n s_and_b64 %12, %12, %9\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "scc");
        } else if (OP == CMP_SALU) {
            asm volatile(REP8("v_cmp_lt_f64 vcc, %0, %1\n s_and_b64 %4, %4, vcc\n v_cmp_lt_f64 vcc, %1, %2\n s_and_b64 %5, %5, vcc\n v_cmp_lt_f64 vcc, %2, %3\n s_and_b64 %6, %6, vcc\n v_cmp_lt_f64 vcc, %3, %0\n s_and_b64 %7, %7, vcc\n")
                         : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "vcc", "scc");
        } else if (OP == CMP_NOP_CND) {
            // what the compiler emits for `x = (a == b) ? y : x` with the mask in an SGPR pair: compare, 2 wait states, select
            asm volatile(REP8("v_cmp_eq_u32 %8, %4, %5\n s_nop 1\n v_cndmask_b32 %0, %0, %6, %8\n v_cmp_eq_u32 %9, %5, %6\n s_nop 1\n v_cndmask_b32 %1, %1, %7, %9\n"
                              "v_cmp_eq_u32 %8, %6, %7\n s_nop 1\n v_cndmask_b32 %2, %2, %4, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(s0), "=&s"(s1));
        } else if (OP == DS_READ_B32) {
            asm volatile(REP8("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                              "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n")
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(la) : "memory");
        } else if (OP == DDA_STEP) {
            // the fast step of aic_trace.hip's stepping trip, verbatim in shape: the DDA step (axis from min(t_max), one exec-masked
            // run per axis), the bounds test, and the count; the 2-byte lookup is left out (this measures issue, not memory)
            unsigned long long sv, mx;
            asm volatile(
                "s_and_saveexec_b64 %[sv], %[m]\n\t"
                "v_min_f64 %[lt], %[tx], %[ty]\n\t"
                "v_min_f64 %[lt], %[lt], %[tz]\n\t"
                "v_cmp_eq_f64 %[mx], %[tz], %[lt]\n\t"
                "v_cmp_eq_f64 vcc, %[ty], %[lt]\n\t"
                "s_andn2_b64 vcc, vcc, %[mx]\n\t"
                "s_mov_b64 exec, %[mx]\n\t"
                "v_add_f64 %[tz], %[tz], %[td]\n\t"
                "v_add_u32 %[rz], -1, %[rz]\n\t"
                "v_add_u32 %[bo], %[bo], %[ss]\n\t"
                "v_mov_b32 %[lax], 2\n\t"
                "s_or_b64 %[mx], %[mx], vcc\n\t"
                "s_mov_b64 exec, vcc\n\t"
                "v_add_f64 %[ty], %[ty], %[td]\n\t"
                "v_add_u32 %[ry], -1, %[ry]\n\t"
                "v_add_u32 %[bo], %[bo], %[ss]\n\t"
                "v_mov_b32 %[lax], 1\n\t"
                "s_andn2_b64 exec, %[m], %[mx]\n\t"
                "v_add_f64 %[tx], %[tx], %[td]\n\t"
                "v_add_u32 %[rx], -1, %[rx]\n\t"
                "v_add_u32 %[bo], %[bo], %[ss]\n\t"
                "v_mov_b32 %[lax], 0\n\t"
                "s_mov_b64 exec, %[sv]\n\t"
                "v_min3_u32 %[k], %[rx], %[ry], %[rz]\n\t"
                "v_cmp_eq_u32 vcc, 0, %[k]\n\t"
                "s_and_b64 %[mx], vcc, %[m]\n\t"
                "s_andn2_b64 %[mx], %[m], %[mx]\n\t"
                "v_addc_co_u32 %[cnt], vcc, 0, %[cnt], %[mx]\n\t"
                : [tx] "+v"(d0), [ty] "+v"(d1), [tz] "+v"(d2), [lt] "+v"(d3), [rx] "+v"(a0), [ry] "+v"(a1), [rz] "+v"(a2), [bo] "+v"(a3), [lax] "+v"(a4),
                  [k] "+v"(a5), [cnt] "+v"(a6), [sv] "=&s"(sv), [mx] "=&s"(mx)
                : [td] "v"(dc), [ss] "v"(c), [m] "s"(~0ull)
                : "vcc", "scc");  // (without the SCC clobber the loop's own compare, placed before the block, is overwritten by it)
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) +
                                                 (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + (uint32_t)(s0 + s1 + s2 + s3);
    if ((threadIdx.x & 63u) == 0u) {
        WaveRec w;
        w.hw_id = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        w.xcc_id = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        w.t_shader = t1 - t0;
        w.t_real = r1 - r0;
        rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = w;
    }
}

struct Result {
    double cyc_shader, cyc_shader_max, cyc_wall, clock_ghz, ms;
    std::map<int, int> hist;  // waves on a SIMD -> number of SIMDs
};

template <int OP>
Result run(int n_cus, int W, int iters) {
    const int groups_per_cu = W == 8 ? 2 : 1;
    const int threads = 256 * (W == 8 ? 4 : W);
    const size_t lds = W == 8 ? (64u << 10) : (96u << 10);  // one (W = 8: two) workgroups per CU, no more
    const int blocks = n_cus * groups_per_cu;
    const int waves = blocks * threads / 64;
    uint32_t *out;
    WaveRec *rec;
    hipMalloc(&out, sizeof(uint32_t) * (size_t)blocks * threads);
    hipMalloc(&rec, sizeof(WaveRec) * waves);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, rec, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, rec, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("!! %s: %s\n", kNames[OP], hipGetErrorString(err));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<WaveRec> h(waves);
    hipMemcpy(h.data(), rec, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost);
    Result r;
    std::map<uint64_t, int> per_simd;
    double ts = 0, tr = 0, ts_max = 0;
    for (const WaveRec &w : h) {
        // HW_ID (gfx9): wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13 (gfx950: se 3 bits), ...
        const uint64_t simd = (w.hw_id >> 4) & 3u, cu = (w.hw_id >> 8) & 15u, sh = (w.hw_id >> 12) & 1u, se = (w.hw_id >> 13) & 7u, xcc = w.xcc_id & 15u;
        per_simd[(xcc << 24) | (se << 16) | (sh << 12) | (cu << 4) | simd]++;
        ts += (double)w.t_shader;
        ts_max = std::max(ts_max, (double)w.t_shader);
        tr += (double)w.t_real;
    }
    for (auto &kv : per_simd) r.hist[kv.second]++;
    ts /= waves;
    tr /= waves;
    const double n_inst = (double)iters * kPerBlock[OP];
    r.clock_ghz = ts / (tr * 10.0);              // shader ticks per ns (s_memrealtime: 100 MHz)
    r.cyc_shader = ts / n_inst / W;              // shader cycles the SIMD spent per wave-instruction (W waves share it): average wave
    r.cyc_shader_max = ts_max / n_inst / W;      // ... by the wave that took longest (issue arbitration favours the oldest wave: the others finish later)
    r.cyc_wall = ms * 1e6 * r.clock_ghz / n_inst / W;  // same from the launch's HIP-event time (includes launch + drain)
    r.ms = ms;
    hipFree(out);
    hipFree(rec);
    return r;
}

template <int OP>
void sweep(int n_cus) {
    const int ws[] = {1, 2, 3, 4, 8};
    for (int w : ws) {
        // aim at >= 2 ms per launch: the slowest streams need ~4 cycles per instruction per wave-slot
        const int iters = std::max(2000, (int)(6.0e6 / (kPerBlock[OP] * w)));
        const Result r = run<OP>(n_cus, w, iters);
        char hist[160];
        int n = 0;
        for (auto &kv : r.hist) n += snprintf(hist + n, sizeof(hist) - n, "%dx%d ", kv.second, kv.first);
        printf("%-52s W=%d %8.3f ms  clock %.3f GHz  %6.3f cyc/inst/SIMD (s_memtime, mean wave)  %6.3f (slowest wave)  %6.3f (events)   SIMDs x waves: %s\n", kNames[OP], w, r.ms, r.clock_ghz,
               r.cyc_shader, r.cyc_shader_max, r.cyc_wall, hist);
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cus = p.multiProcessorCount;
    printf("# %s, %d CUs, clock %d kHz; one workgroup of W*256 threads per CU (W = 8: two of 1024), residency from HW_ID shown per line\n", p.name, n_cus, p.clockRate);
    sweep<DDA_STEP>(n_cus);
    sweep<CND_VCC_E64>(n_cus);
    sweep<CMP_VCC_CND>(n_cus);
    sweep<ADDC_VCC>(n_cus);
    sweep<MOV_B32>(n_cus);
    sweep<ADD_U32>(n_cus);
    sweep<FMA_F32>(n_cus);
    sweep<PK_FMA_F32>(n_cus);
    sweep<ADD_F64>(n_cus);
    sweep<FMA_F64>(n_cus);
    sweep<CMP_F64>(n_cus);
    sweep<CNDMASK_VCC>(n_cus);
    sweep<CNDMASK_SGPR>(n_cus);
    sweep<MUL_LO_U32>(n_cus);
    sweep<SALU_AND>(n_cus);
    sweep<VALU_SALU>(n_cus);
    sweep<CMP_SALU>(n_cus);
    sweep<CMP_NOP_CND>(n_cus);
    sweep<DS_READ_B32>(n_cus);
    return 0;
}
