// issue_mix.hip -- do VALU and SALU instructions of DIFFERENT waves of a SIMD issue side by side on gfx950?
//
// 2 workgroups x 1024 threads per CU = 8 waves per SIMD.  Kernel `pure<0>`: all waves run a loop of 8 independent
// v_add_u32; `pure<1>`: 8 s_and_b64 (a dependent chain on one register pair: scalar results are forwarded in order);
// `mixed`: even waves the vector loop, odd waves the scalar loop.  If the two kinds share the issue slot, mixed takes
// (T_valu + T_salu) / 2; if they issue side by side, max(T_valu, T_salu) / 2.  `mixed2`: every wave alternates a vector
// and a scalar instruction (what a real kernel looks like inside one wave).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/issue_mix tools/issue_mix.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define V8 asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t" \
                        "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"     \
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(b));
#define S8 asm volatile("s_xor_b64 s[20:21], s[20:21], s[22:23]\n\ts_xor_b64 s[24:25], s[24:25], s[22:23]\n\t"               \
                        "s_xor_b64 s[26:27], s[26:27], s[22:23]\n\ts_xor_b64 s[28:29], s[28:29], s[22:23]\n\t"               \
                        "s_xor_b64 s[20:21], s[20:21], s[22:23]\n\ts_xor_b64 s[24:25], s[24:25], s[22:23]\n\t"               \
                        "s_xor_b64 s[26:27], s[26:27], s[22:23]\n\ts_xor_b64 s[28:29], s[28:29], s[22:23]"                   \
                        ::: "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
#define VS8 asm volatile("v_add_u32 %0, %0, %8\n\ts_xor_b64 s[20:21], s[20:21], s[22:23]\n\tv_add_u32 %1, %1, %8\n\ts_xor_b64 s[24:25], s[24:25], s[22:23]\n\t" \
                         "v_add_u32 %2, %2, %8\n\ts_xor_b64 s[26:27], s[26:27], s[22:23]\n\tv_add_u32 %3, %3, %8\n\ts_xor_b64 s[28:29], s[28:29], s[22:23]\n\t" \
                         "v_add_u32 %4, %4, %8\n\ts_xor_b64 s[20:21], s[20:21], s[22:23]\n\tv_add_u32 %5, %5, %8\n\ts_xor_b64 s[24:25], s[24:25], s[22:23]\n\t" \
                         "v_add_u32 %6, %6, %8\n\ts_xor_b64 s[26:27], s[26:27], s[22:23]\n\tv_add_u32 %7, %7, %8\n\ts_xor_b64 s[28:29], s[28:29], s[22:23]"     \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(b)                               \
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");

// MODE 0: vector, 1: scalar, 2: even waves vector / odd waves scalar, 3: both in every wave (16 instructions per trip)
template <int MODE> __global__ void __launch_bounds__(1024) k_mix(unsigned *out, unsigned seed, int trips)
{ unsigned r[8], b = seed * 3 + 1;
  for (int i = 0; i < 8; i++) r[i] = (seed + threadIdx.x) * (i + 1);
  const bool odd = (threadIdx.x >> 6) & 1;
  if (MODE == 0 || (MODE == 2 && !odd)) for (int it = 0; it < trips; it++) { V8 }
  if (MODE == 1 || (MODE == 2 && odd))  for (int it = 0; it < trips; it++) { S8 }
  if (MODE == 3)                        for (int it = 0; it < trips; it++) { VS8 }
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s += r[i];
  if (s == 0x12345677u) out[0] = s;
}

template <int MODE> static double slope_us(unsigned *d, int cus)
{ hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  double us[2]; const int trips[2] = { 4096, 32768 };
  for (int j = 0; j < 2; j++)
    { hipLaunchKernelGGL(k_mix<MODE>, dim3(2 * cus), dim3(1024), 0, 0, d, 12345u, trips[j]);
      hipEventRecord(ea, 0);
      for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_mix<MODE>, dim3(2 * cus), dim3(1024), 0, 0, d, 12345u, trips[j]);
      hipEventRecord(eb, 0); hipEventSynchronize(eb);
      float ms = 0; hipEventElapsedTime(&ms, ea, eb);
      us[j] = ms * 1e3 / 3;
    }
  return (us[1] - us[0]) / (32768 - 4096);           // microseconds per loop trip of the whole launch
}

int main()
{ unsigned *d; hipMalloc(&d, 64);
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const double tv = slope_us<0>(d, cus), ts = slope_us<1>(d, cus), tm = slope_us<2>(d, cus), tb = slope_us<3>(d, cus);
  printf("# 8 waves per SIMD, ns per loop trip of 8 instructions per wave (wall clock slope)\n");
  printf("all waves vector (8 v_add_u32)          %8.2f ns  = %.2f ns per wave-instruction per SIMD\n", tv * 1e3, tv * 1e3 / 64);
  printf("all waves scalar (8 s_xor_b64)          %8.2f ns  = %.2f ns per wave-instruction per SIMD\n", ts * 1e3, ts * 1e3 / 64);
  printf("4 vector waves + 4 scalar waves         %8.2f ns  (shared issue slot: %.2f, side by side: %.2f)\n", tm * 1e3,
         (tv + ts) * 1e3 / 2, (tv > ts ? tv : ts) * 1e3 / 2);
  printf("every wave 8 vector + 8 scalar          %8.2f ns  (shared issue slot: %.2f, side by side: %.2f)\n", tb * 1e3,
         (tv + ts) * 1e3, (tv > ts ? tv : ts) * 1e3);
  return 0;
}
