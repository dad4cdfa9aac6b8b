// xcd_atomics.hip — can the aggregate's LDS tables be flushed straight into XCD-PRIVATE tables in global memory with
// L2-level atomics (no slabs, no reduction kernel)?  Every workgroup reads the id of the XCD it runs on (HW_REG_XCC_ID)
// and adds into that XCD's table with workgroup-scope (no sc1: executed by the XCD's own L2) 64-bit atomics; the host
// then checks the sums exactly and times the kernel against the agent-scope form.
//   hipcc --offload-arch=gfx950 -O3 xcd_atomics.hip -o xcd_atomics && ./xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xFu; }  // HW_REG_XCC_ID[3:0]

template <int SCOPE>
__global__ __launch_bounds__(1024) void k_flush(unsigned long long* xtab, uint32_t words_per_xcd, int per_thread, uint32_t* wg_xcc, int spread) {
  const uint32_t x = xcc_id();
  if (threadIdx.x == 0) wg_xcc[blockIdx.x] = x;
  unsigned long long* tab = xtab + (size_t)x * words_per_xcd;
  uint32_t r = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
  for (int i = 0; i < per_thread; ++i) {
    // spread = 1: the flush pattern (thread = record, consecutive words); spread = 0: random words
    const uint32_t w = spread ? ((threadIdx.x * (uint32_t)per_thread + (uint32_t)i) % words_per_xcd) : ((r = r * 1664525u + 1013904223u) >> 8) % words_per_xcd;
    (void)__hip_atomic_fetch_add(tab + w, 1ull + (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
  }
}

int main() {
  const uint32_t W = 18000;  // 1000 throttles x 18 words
  unsigned long long* xtab;
  uint32_t* wg_xcc;
  CK(hipMalloc(&xtab, (size_t)16 * W * 8));
  CK(hipMalloc(&wg_xcc, 512 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int scope = 0; scope < 2; ++scope)
    for (int spread = 0; spread < 2; ++spread)
      for (int per : {4, 18}) {
        for (int grid : {256, 512}) {
          CK(hipMemset(xtab, 0, (size_t)16 * W * 8));
          float ms = 0;
          CK(hipEventRecord(e0, 0));
          if (scope == 0) hipLaunchKernelGGL(k_flush<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(1024), 0, 0, xtab, W, per, wg_xcc, spread);
          else hipLaunchKernelGGL(k_flush<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(1024), 0, 0, xtab, W, per, wg_xcc, spread);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
          std::vector<unsigned long long> h((size_t)16 * W);
          std::vector<uint32_t> hx(512);
          CK(hipMemcpy(h.data(), xtab, h.size() * 8, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hx.data(), wg_xcc, 512 * 4, hipMemcpyDeviceToHost));
          unsigned long long want_per_wg = 0;
          for (int i = 0; i < per; ++i) want_per_wg += 1ull + (unsigned long long)i;
          want_per_wg *= 1024ull;
          unsigned long long wg_of[16] = {0}, got[16] = {0};
          for (int b = 0; b < grid; ++b) wg_of[hx[b] & 15u]++;
          for (int x = 0; x < 16; ++x)
            for (uint32_t w = 0; w < W; ++w) got[x] += h[(size_t)x * W + w];
          bool exact = true;
          for (int x = 0; x < 16; ++x) exact &= got[x] == wg_of[x] * want_per_wg;
          printf("%s scope, %s words, %2d atomics / thread, grid %3d : %7.1f us  %6.1f atomics / ns  sums %s  workgroups per XCD:", scope == 0 ? "workgroup (L2)" : "agent         ",
                 spread ? "consecutive" : "random     ", per, grid, ms * 1e3, (double)grid * 1024 * per / (ms * 1e6), exact ? "EXACT" : "WRONG");
          for (int x = 0; x < 8; ++x) printf(" %llu", wg_of[x]);
          printf("\n");
        }
      }
  // a second kernel reads what the first one's L2 atomics left (kernel boundary = release / acquire): covered by the D2H above
  return 0;
}
