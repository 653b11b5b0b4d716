// Issue-rate probe for the vector ALU of one SIMD: how many clocks does a wave64 instruction occupy, and
// does a partly empty EXEC mask shorten it?  Every wave runs a long unrolled chain of one instruction kind
// (eight independent accumulators, so latency is hidden by the wave itself and by the other waves of the SIMD);
// the grid puts `wavesPerSimd` waves on every SIMD of the chip.
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip        run: ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

constexpr int kIters = 16384;
constexpr int kPerIter = 32;  // instructions of the probed kind per loop trip

enum Kind { K0, K1, K2, K3, K4, K5, K6, K7, K8, K9, K10, K11, K12, K13, K14, K15, K16, K17, K18, K19, K20, K21, K22, K23, K24, K25 };

template <int KIND>
__global__ void __launch_bounds__(512) probe(float* out, int activeLanes, const float* in) {
    const int lane = threadIdx.x & 63;
    float a0 = in[0] + lane, a1 = in[1], a2 = in[2], a3 = in[3], a4 = in[4], a5 = in[5], a6 = in[6], a7 = in[7];
    const float m = in[8], c = in[9];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 pm = {m, m}, pc = {c, c};
    double q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
    const double qm = m, qc = c;
    unsigned int u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3),
                 u4 = lane, u5 = lane + 1, u6 = lane + 2, u7 = lane + 3;
    const unsigned int um = __float_as_uint(m), uc = __float_as_uint(c);
    if (lane < activeLanes) {
        for (int it = 0; it < kIters; it++) {
#pragma unroll
            for (int k = 0; k < kPerIter / 8; k++) {
                if (KIND == K0)
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K1)
                    asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K2)
                    asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K3)
                    asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K4)
                    asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K5)
                    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                                 : "v"(pm), "v"(pc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K6)
                    asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                                 : "v"(pm), "v"(pc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K7)
                    asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K8)
                    asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K9)
                    asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K10)
                    asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[20:21], %1, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[20:21], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[20:21], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[20:21], %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K11)
                    asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K12)
                    asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K13)
                    asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K14)
                    asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K15)
                    asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K16)
                    asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K17)
                    asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K18)
                    asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K19)
                    asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K20)
                    asm volatile("v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 7\n v_alignbit_b32 %2, %2, %2, 7\n v_alignbit_b32 %3, %3, %3, 7\n v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n v_alignbit_b32 %7, %7, %7, 7"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K21)
                    asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                                 : "v"(um), "v"(uc)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K22)
                    asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K23)
                    asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K24)
                    asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(m), "v"(c)
                                 : "vcc", "scc", "s20", "s21");
                if (KIND == K25)
                    asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
                                 : "v"(qm), "v"(qc)
                                 : "vcc", "scc", "s20", "s21");
            }
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
              p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + __uint_as_float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) +
              (float)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, int wavesPerSimd, int activeLanes, float* out, const float* in, double ghz) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wavesPerSimd / 2;  // 512 threads = 8 waves = 2 per SIMD of one CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    probe<KIND><<<blocks, 512>>>(out, activeLanes, in);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<KIND><<<blocks, 512>>>(out, activeLanes, in);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instrPerSimd = (double)wavesPerSimd * kIters * kPerIter;
    const double clocks = ms * 1e-3 * ghz * 1e9;
    printf("%-12s waves/SIMD=%d active=%2d  %8.3f ms  %.2f clocks per wave-instruction (at %.2f GHz)\n", name,
           wavesPerSimd, activeLanes, ms, clocks / instrPerSimd, ghz);
}

template <int MIX>
__global__ void __launch_bounds__(512) probeMix(float* out, const float* in, int seed0, int seed1) {
    const int lane = threadIdx.x & 63;
    float a0 = in[0] + lane, a1 = in[1], a2 = in[2], a3 = in[3], a4 = in[4], a5 = in[5], a6 = in[6], a7 = in[7];
    const float m = in[8];
    int s0 = seed0, s1 = seed1;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (MIX == 0)  // 8 max_f32 + 8 scalar ALU, interleaved
                asm volatile("v_max_f32 %0, %0, %10\n s_add_u32 %8, %8, %9\n v_max_f32 %1, %1, %10\n s_xor_b32 %9, %9, %8\n v_max_f32 %2, %2, %10\n s_add_u32 %8, %8, %9\n v_max_f32 %3, %3, %10\n s_xor_b32 %9, %9, %8\n"
                             "v_max_f32 %4, %4, %10\n s_add_u32 %8, %8, %9\n v_max_f32 %5, %5, %10\n s_xor_b32 %9, %9, %8\n v_max_f32 %6, %6, %10\n s_add_u32 %8, %8, %9\n v_max_f32 %7, %7, %10\n s_xor_b32 %9, %9, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m) : "scc");
            if (MIX == 1)  // 8 max_f32 + 8 s_nop
                asm volatile("v_max_f32 %0, %0, %10\n s_nop 0\n v_max_f32 %1, %1, %10\n s_nop 0\n v_max_f32 %2, %2, %10\n s_nop 0\n v_max_f32 %3, %3, %10\n s_nop 0\n"
                             "v_max_f32 %4, %4, %10\n s_nop 0\n v_max_f32 %5, %5, %10\n s_nop 0\n v_max_f32 %6, %6, %10\n s_nop 0\n v_max_f32 %7, %7, %10\n s_nop 0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m) : "scc");
            if (MIX == 2)  // 8 add_f32 + 8 scalar ALU
                asm volatile("v_add_f32 %0, %0, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %1, %1, %10\n s_xor_b32 %9, %9, %8\n v_add_f32 %2, %2, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %3, %3, %10\n s_xor_b32 %9, %9, %8\n"
                             "v_add_f32 %4, %4, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %5, %5, %10\n s_xor_b32 %9, %9, %8\n v_add_f32 %6, %6, %10\n s_add_u32 %8, %8, %9\n v_add_f32 %7, %7, %10\n s_xor_b32 %9, %9, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m) : "scc");
            if (MIX == 3)  // 16 scalar ALU
                asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n"
                             "s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0"
                             : "+s"(s0), "+s"(s1) : : "scc");
            if (MIX == 4)  // 8 max_f32 with a taken branch after every second one
                asm volatile("v_max_f32 %0, %0, %10\n v_max_f32 %1, %1, %10\n s_branch 1f\n s_nop 0\n1:\n v_max_f32 %2, %2, %10\n v_max_f32 %3, %3, %10\n s_branch 2f\n s_nop 0\n2:\n"
                             "v_max_f32 %4, %4, %10\n v_max_f32 %5, %5, %10\n s_branch 3f\n s_nop 0\n3:\n v_max_f32 %6, %6, %10\n v_max_f32 %7, %7, %10\n s_branch 4f\n s_nop 0\n4:"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m) : "scc");
            if (MIX == 5)  // 8 max_f32 with exec save/restore pairs (as divergent code has)
                asm volatile("s_and_saveexec_b64 s[20:21], vcc\n v_max_f32 %0, %0, %10\n v_max_f32 %1, %1, %10\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], vcc\n v_max_f32 %2, %2, %10\n v_max_f32 %3, %3, %10\n s_or_b64 exec, exec, s[20:21]\n"
                             "s_and_saveexec_b64 s[20:21], vcc\n v_max_f32 %4, %4, %10\n v_max_f32 %5, %5, %10\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], vcc\n v_max_f32 %6, %6, %10\n v_max_f32 %7, %7, %10\n s_or_b64 exec, exec, s[20:21]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m) : "scc", "s20", "s21");
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 ^ s1);
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int MIX>
static void runMix(const char* name, int wavesPerSimd, float* out, const float* in, double ghz) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * wavesPerSimd / 2;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    probeMix<MIX><<<blocks, 512>>>(out, in, 3, 5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probeMix<MIX><<<blocks, 512>>>(out, in, 3, 5);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double groupsPerSimd = (double)wavesPerSimd * kIters * 4;  // groups of 8 vector (+ 8 other) instructions
    printf("%-34s waves/SIMD=%d  %8.3f ms  %.1f clocks per group\n", name, wavesPerSimd, ms, ms * 1e-3 * ghz * 1e9 / groupsPerSimd);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *out, *in;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&in, 4096));
    float h[64];
    for (int i = 0; i < 64; i++) h[i] = 1.0f + 0.001f * i;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    int khz = 0;
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock rate attribute: %.3f GHz\n", ghz);
    for (int w : {1, 4}) run<K0>("fma_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K1>("mul_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K2>("add_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K3>("max_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K4>("min3_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K5>("pk_fma_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K6>("pk_mul_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K7>("mov_b32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K8>("cndmask", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K9>("cmp_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K10>("cmp_sgpr", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K11>("add_u32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K12>("xor_b32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K13>("and_b32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K14>("lshlrev", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K15>("lshl_add", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K16>("add3_u32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K17>("mul_lo_u32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K18>("mul_u32_u24", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K19>("mad_u32_u24", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K20>("alignbit", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K21>("bfe_u32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K22>("cvt_f32_u32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K23>("rcp_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K24>("sqrt_f32", w, 64, out, in, ghz);
    for (int w : {1, 4}) run<K25>("fma_f64", w, 64, out, in, ghz);
    run<K0>("fma_f32", 4, 16, out, in, ghz);
    run<K8>("cndmask", 4, 16, out, in, ghz);
    for (int w : {1, 6}) runMix<0>("8 v_max_f32 + 8 salu", w, out, in, ghz);
    for (int w : {1, 6}) runMix<1>("8 v_max_f32 + 8 s_nop", w, out, in, ghz);
    for (int w : {1, 6}) runMix<2>("8 v_add_f32 + 8 salu", w, out, in, ghz);
    for (int w : {1, 6}) runMix<3>("16 salu", w, out, in, ghz);
    for (int w : {1, 6}) runMix<4>("8 v_max_f32 + 4 taken branches", w, out, in, ghz);
    for (int w : {1, 6}) runMix<5>("8 v_max_f32 + 4 exec save/restore", w, out, in, ghz);
    return 0;
}
