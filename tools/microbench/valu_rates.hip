// Issue rates of the VALU operations the scans are made of (gfx950): v_or_b32, v_or3_b32, v_bitop3_b32, v_bfi_b32,
// v_and_or_b32, v_bcnt_u32_b32, v_mul_u32_u24, v_lshl_add_u32 — eight independent chains per lane, 64 x 1024 threads per
// CU-filling launch.   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void rate(unsigned* out, unsigned seed, int iters) {
  unsigned a0 = threadIdx.x ^ seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  const unsigned b = seed * 2654435761u, c = seed ^ 0x5bd1e995u;
  for (int i = 0; i < iters; ++i) {
#define STEP(r)                                                                                           \
    if (OP == 0) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(b));                                  \
    if (OP == 1) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));                     \
    if (OP == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(b), "v"(c));      \
    if (OP == 3) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));                     \
    if (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));                  \
    if (OP == 5) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r) : "v"(b));                            \
    if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));                             \
    if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r) : "v"(b));                         \
    if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));                              \
    if (OP == 9) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(b));                                 \
    if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b));                                \
    if (OP == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));                                \
    if (OP == 12) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r));                                      \
    if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b));                       \
    if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b));                                    \
    if (OP == 15) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r));                                            \
    if (OP == 16) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(b));                                \
    if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r));                                       \
    if (OP == 18) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));                   \
    if (OP == 19) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(r), "v"(b) : "vcc");                    \
    if (OP == 20) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r) : "v"(b));              \
    if (OP == 21) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));                \
    if (OP == 22) asm volatile("v_readlane_b32 s12, %0, 3" : : "v"(r) : "s12");                           \
    if (OP == 23) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(r) : "v"(b));
    REP8(STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7))
  }
  out[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
static void run(const char* name, unsigned* d) {
  const int iters = 2000, blocks = 512;  // 2 workgroups of 16 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<OP>, blocks, 1024, 0, 0, d, 1u, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<OP>, blocks, 1024, 0, 0, d, 7u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)blocks * 16 * iters * 64;  // per launch
  // 1024 SIMDs; cycles per wave instruction and SIMD at 2.4 GHz
  printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction and SIMD (2.4 GHz assumed)\n", name, ms, ms * 1e-3 * 2.4e9 / (wave_instr / 1024.0));
}
int main() {
  unsigned* d;
  hipMalloc(&d, 512 * 1024 * 4);
  run<0>("v_or_b32", d);
  run<1>("v_or3_b32", d);
  run<2>("v_bitop3_b32", d);
  run<3>("v_bfi_b32", d);
  run<4>("v_and_or_b32", d);
  run<5>("v_bcnt_u32_b32", d);
  run<6>("v_mul_u32_u24", d);
  run<7>("v_lshl_add_u32", d);
  run<8>("v_mul_lo_u32", d);
  run<9>("v_xor_b32", d);
  run<10>("v_and_b32", d);
  run<11>("v_add_u32", d);
  run<12>("v_lshlrev_b32", d);
  run<13>("v_cndmask (vcc)", d);
  run<14>("v_mov_b32", d);
  run<15>("v_ffbl_b32", d);
  run<16>("v_min_u32", d);
  run<17>("v_bfe_u32", d);
  run<18>("v_add3_u32", d);
  run<19>("v_cmp_ne_u32", d);
  run<20>("v_cndmask (sgpr)", d);
  run<21>("v_mad_u32_u24", d);
  run<22>("v_readlane_b32", d);
  run<23>("v_mul_u24 sdwa", d);
  return 0;
}
