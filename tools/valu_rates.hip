// valu_rates.hip -- issue cost of the gfx950 vector instructions the pass-1 kernel is made of.
//
// One workgroup per CU (x waves-per-SIMD), every wave runs REPS x 8 independent copies of one instruction
// between two s_memtime reads.  Output: shader cycles per wave-instruction seen by ONE wave when 1, 2 and 4
// waves share a SIMD (the second and third figure divided by the wave count is the SIMD's issue cost).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define REPS 1024

#define K8(ASM) \
  ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)

// body: 8 independent instructions on r[0..7]; a, b, c are loop-invariant vector operands
#define DEFKERNEL(NAME, BODY)                                                                    \
  __global__ void __launch_bounds__(1024) NAME(uint64_t *out, unsigned seed, int reps)                       \
  { unsigned r[8];                                                                               \
    unsigned a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55aa;                        \
    unsigned long long q[8];                                                                     \
    __shared__ unsigned lds[4096];                                                               \
    lds[threadIdx.x] = 0; lds[threadIdx.x + 1024] = 0;                                           \
    for (int i = 0; i < 8; i++) { r[i] = a * (i + 1); q[i] = (unsigned long long) a * (i + 3); } \
    unsigned la = (threadIdx.x * 4) & 4095;                                                      \
    (void) la; (void) q; (void) b; (void) c;                                                     \
    __syncthreads();                                                                             \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    for (int it = 0; it < reps; it++) { BODY }                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    unsigned s = 0;                                                                              \
    for (int i = 0; i < 8; i++) s += r[i] + (unsigned) q[i] + (unsigned) (q[i] >> 32);           \
    if (s == 0x12345677u) out[1] = s;                                                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                   \
  }

#define A1(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_add_u32, K8(A1))
#define A2(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_xor_b32, K8(A2))
#define A3(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_and_or_b32, K8(A3))
#define A4(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_lshl_or_b32, K8(A4))
#define A5(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_lshl_add_u32, K8(A5))
#define A6(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_alignbit_b32, K8(A6))
#define A7(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_bcnt_u32_b32, K8(A7))
#define ONE8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
#define I8(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
DEFKERNEL(k_cndmask_b32, ONE8(I8))
#define I8b(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
DEFKERNEL(k_cndmask_e64_sgpr, ONE8(I8b))
#define I8c(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
DEFKERNEL(k_cmp_vcc_then_cndmask_vcc, ONE8(I8c))
#define I9(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n\t"
DEFKERNEL(k_cmp_eq_u32_vcc, ONE8(I9))
#define I9b(i) "v_cmp_eq_u32_e64 s[20:21], %" #i ", %8\n\t"
DEFKERNEL(k_cmp_eq_u32_sgpr, ONE8(I9b))
#define I9c(i) "v_cmp_eq_u32_e64 s[20:21], %" #i ", %8\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n\t"
DEFKERNEL(k_cmp_then_cndmask, ONE8(I9c))
#define I9d(i) "v_cmp_eq_u32_e64 s[20:21], %" #i ", %8\n\tv_cmp_lt_u32_e64 s[22:23], %" #i ", %9\n\ts_and_b64 s[24:25], s[20:21], s[22:23]\n\t"
DEFKERNEL(k_cmp_cmp_sand, ONE8(I9d))
#define A10(i) asm volatile("v_bfrev_b32 %0, %0" : "+v"(r[i]));
DEFKERNEL(k_bfrev_b32, K8(A10))
#define A11(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r[i]));
DEFKERNEL(k_bfe_u32, K8(A11))
#define A12(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_perm_b32, K8(A12))
#define A13(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_mad_u32_u24, K8(A13))
#define A14(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mul_u32_u24, K8(A14))
#define A15(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mul_lo_u32, K8(A15))
#define A16(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mul_hi_u32, K8(A16))
#define A17(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(b), "v"(c) : "vcc");
DEFKERNEL(k_mad_u64_u32, K8(A17))
#define A18(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
DEFKERNEL(k_lshlrev_b64, K8(A18))
#define A19(i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(b));
DEFKERNEL(k_lshrrev_b64_v, K8(A19))
#define A20(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(r[i]), "+v"(r[(i + 4) & 7]) : "v"(b), "v"(c) : "vcc");
DEFKERNEL(k_add64_pair, A20(0) A20(1) A20(2) A20(3))
#define A21(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cmp_lt_u64 vcc, %1, %0\n\tv_cmp_lt_u64 vcc, %0, %1\n\tv_cmp_lt_u64 vcc, %1, %0" : : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");
DEFKERNEL(k_cmp_lt_u64, A21(0) A21(2))
#define A22(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_min_u32, K8(A22))
#define A23(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_xad_u32, K8(A23))
#define A24(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_add3_u32, K8(A24))
#define A25(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mov_dpp_row_shr1, K8(A25))
#define A26(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mov_dpp_wave_shr1, K8(A26))
#define A26b(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mov_dpp_wave_shl1, K8(A26b))
#define A27(i) asm volatile("v_xor_b32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_xor_dpp_wave_shl1, K8(A27))
#define A28(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_pk_add_u16, K8(A28))
#define A29(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_fma_f32, K8(A29))
#define A30(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
DEFKERNEL(k_pk_fma_f32, A30(0) A30(2) A30(4) A30(6))
#define A31(i) asm volatile("ds_add_u32 %0, %1" : : "v"(la), "v"(r[i]) : "memory");
DEFKERNEL(k_ds_add_u32, K8(A31))
#define A32(i) asm volatile("ds_read_b64 %0, %1" : "=v"(q[i]) : "v"(la) : "memory");
DEFKERNEL(k_ds_read_b64, K8(A32))
#define A33(i) asm volatile("ds_read_b128 %0, %1" : "=v"(*(uint4 *) &q[i]) : "v"(la) : "memory");
DEFKERNEL(k_ds_read_b128_x4, A33(0) A33(2) A33(4) A33(6))
#define A34(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(r[i]) : "v"(la) : "memory");
DEFKERNEL(k_ds_bpermute_b32, K8(A34))
#define I35(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n\t"
DEFKERNEL(k_sub_co_u32, ONE8(I35))
#define A36(i) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(r[i]));
DEFKERNEL(k_lshrrev_b32, K8(A36))
#define I37(i) "s_and_b64 s[20:21], s[20:21], vcc\n\t"
DEFKERNEL(k_s_and_b64, ONE8(I37))
#define A38(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
DEFKERNEL(k_or3_b32, K8(A38))
#define A39(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(b));
DEFKERNEL(k_mov_b32, K8(A39))
#define I40(i) "v_readlane_b32 s20, %" #i ", 5\n\t"
DEFKERNEL(k_readlane, ONE8(I40))
#define A41(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(r[i]));
DEFKERNEL(k_ffbh_u32, K8(A41))
#define A42(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(b));
DEFKERNEL(k_mbcnt_lo, K8(A42))

typedef void (*kfn)(uint64_t *, unsigned, int);
struct Ent { const char *name; kfn f; int per_iter; };

int main()
{ std::vector<Ent> ks = {
#define E(N, P) { #N, N, P }
    E(k_add_u32, 8), E(k_xor_b32, 8), E(k_and_or_b32, 8), E(k_lshl_or_b32, 8), E(k_lshl_add_u32, 8), E(k_alignbit_b32, 8),
    E(k_bcnt_u32_b32, 8), E(k_cndmask_b32, 8), E(k_cmp_eq_u32_vcc, 8), E(k_cmp_eq_u32_sgpr, 8), E(k_cmp_then_cndmask, 16), E(k_cndmask_e64_sgpr, 8), E(k_cmp_vcc_then_cndmask_vcc, 16), E(k_cmp_cmp_sand, 24), E(k_bfrev_b32, 8), E(k_bfe_u32, 8),
    E(k_perm_b32, 8), E(k_mad_u32_u24, 8), E(k_mul_u32_u24, 8), E(k_mul_lo_u32, 8), E(k_mul_hi_u32, 8), E(k_mad_u64_u32, 8),
    E(k_lshlrev_b64, 8), E(k_lshrrev_b64_v, 8), E(k_add64_pair, 8), E(k_cmp_lt_u64, 8), E(k_min_u32, 8), E(k_xad_u32, 8),
    E(k_add3_u32, 8), E(k_mov_dpp_row_shr1, 8), E(k_mov_dpp_wave_shr1, 8), E(k_mov_dpp_wave_shl1, 8), E(k_xor_dpp_wave_shl1, 8),
    E(k_pk_add_u16, 8), E(k_fma_f32, 8), E(k_pk_fma_f32, 4), E(k_ds_add_u32, 8), E(k_ds_read_b64, 8), E(k_ds_read_b128_x4, 4),
    E(k_ds_bpermute_b32, 8), E(k_sub_co_u32, 8), E(k_lshrrev_b32, 8), E(k_s_and_b64, 8), E(k_or3_b32, 8), E(k_mov_b32, 8),
    E(k_readlane, 8), E(k_ffbh_u32, 8), E(k_mbcnt_lo, 8),
  };
  uint64_t *d; hipMalloc(&d, 64); uint64_t h[2];
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  printf("# cycles (s_memtime ticks) per wave-instruction, seen by one wave, with W waves per SIMD (all 4 SIMDs of every CU busy)\n");
  printf("# the SIMD's issue cost per instruction = value / W once W is large enough to saturate the pipe\n");
  printf("%-24s %8s %8s | %s\n", "instruction", "W=1", "W=8", "ns per wave-instruction per SIMD at W=8: slope between 4096 and 32768 loop trips (wall clock)");
  hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  for (auto &e : ks)
    { printf("%-24s ", e.name + 2); fflush(stdout);
      double res[2];
      for (int j = 0; j < 2; j++)
        { const int threads = j == 0 ? 256 : 1024, grid = j == 0 ? cus : 2 * cus;
          hipLaunchKernelGGL(e.f, dim3(grid), dim3(threads), 0, 0, d, 12345u, REPS);
          hipLaunchKernelGGL(e.f, dim3(grid), dim3(threads), 0, 0, d, 12345u, REPS);
          hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
          res[j] = (double) h[0] / ((double) REPS * e.per_iter);
        }
      double us[2]; int trips[2] = { 4096, 32768 };
      for (int j = 0; j < 2; j++)
        { hipLaunchKernelGGL(e.f, dim3(2 * cus), dim3(1024), 0, 0, d, 12345u, trips[j]);
          hipEventRecord(ea, 0);
          for (int i = 0; i < 3; i++) hipLaunchKernelGGL(e.f, dim3(2 * cus), dim3(1024), 0, 0, d, 12345u, trips[j]);
          hipEventRecord(eb, 0); hipEventSynchronize(eb);
          float ms = 0; hipEventElapsedTime(&ms, ea, eb);
          us[j] = ms * 1e3 / 3;
        }
      const double ns = (us[1] - us[0]) * 1e3 / (8.0 * (trips[1] - trips[0]) * e.per_iter);
      printf("%8.2f %8.2f | %7.3f ns  (%.0f / %.0f us)\n", res[0], res[1], ns, us[0], us[1]);
    }
  // s_memtime rate vs wall clock
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_mul_lo_u32, dim3(cus), dim3(1024), 0, 0, d, 1u, REPS);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("# k_mul_lo_u32 W=4: %.1f us per launch by events, %llu s_memtime ticks inside => %.1f ticks/us\n",
         ms * 1e3 / 20, (unsigned long long) h[0], (double) h[0] / (ms * 1e3 / 20));
  return 0;
}
