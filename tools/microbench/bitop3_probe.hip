// v_bitop3_b32's truth-table convention on gfx950: with a = 0xF0, b = 0xCC, c = 0xAA the low byte of the result IS the
// table when operand 0 is the most significant index bit.   hipcc --offload-arch=gfx950 -o bitop3_probe bitop3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TT>
__global__ void probe(unsigned* out, unsigned a, unsigned b, unsigned c) { out[0] = __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
int main() {
  unsigned* d;
  hipMalloc(&d, 4);
  unsigned h[4];
  hipLaunchKernelGGL(probe<0x70>, 1, 1, 0, 0, d, 0xF0u, 0xCCu, 0xAAu); hipMemcpy(&h[0], d, 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(probe<0x2A>, 1, 1, 0, 0, d, 0xF0u, 0xCCu, 0xAAu); hipMemcpy(&h[1], d, 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(probe<0x96>, 1, 1, 0, 0, d, 0xF0u, 0xCCu, 0xAAu); hipMemcpy(&h[2], d, 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(probe<0xE8>, 1, 1, 0, 0, d, 0xF0u, 0xCCu, 0xAAu); hipMemcpy(&h[3], d, 4, hipMemcpyDeviceToHost);
  printf("bitop3(0xF0, 0xCC, 0xAA): table 0x70 -> 0x%02X, 0x2A -> 0x%02X, 0x96 -> 0x%02X, 0xE8 -> 0x%02X  (operand 0 = most significant index bit: %s)\n",
         h[0] & 0xFF, h[1] & 0xFF, h[2] & 0xFF, h[3] & 0xFF, (h[0] & 0xFF) == 0x70 && (h[1] & 0xFF) == 0x2A ? "yes" : "NO");
  return 0;
}
