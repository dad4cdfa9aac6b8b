// LDS atomic throughput on gfx950 (one workgroup per CU, 16 waves): cycles per wave-instruction for ds_add_u64 /
// ds_add_u32 with 64 / 16 / 4 active lanes, distinct or identical addresses, and plain ds_read + ds_write for
// comparison.   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LDS __attribute__((address_space(3)))
extern __shared__ unsigned char smem[];

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters, int active, int same, int stride8) {
  LDS unsigned long long* t64 = (LDS unsigned long long*)smem;
  LDS uint32_t* t32 = (LDS uint32_t*)smem;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) t64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool on = lane < active;
  // address: per wave a private 64-slot window (no cross-wave conflicts) unless `same`
  uint32_t idx = same ? (uint32_t)(wave * 64) : (uint32_t)(wave * 64 + lane);
  idx = (idx * (uint32_t)stride8) & 16383u;
  const long long t0 = clock64();
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (on) {
      if (MODE == 0) (void)__hip_atomic_fetch_add(t64 + idx, (unsigned long long)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) (void)__hip_atomic_fetch_add(t32 + idx, (uint32_t)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 2) acc += __hip_atomic_fetch_add(t64 + idx, (unsigned long long)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 3) { unsigned long long v = t64[idx]; t64[idx] = v + it + 1; }
      if (MODE == 4) acc += t64[(idx + it) & 16383u];
    }
    if (MODE == 5) acc += (unsigned long long)__builtin_amdgcn_ds_bpermute(((lane + it) & 63) << 2, (int)acc + lane);
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
  if (acc == 0x123456789ull) out[0] = acc + t64[idx];
}

template <int MODE>
static void run(const char* name, int active, int same, int stride8) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8);
  const int iters = 4096;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, d, iters, active, same, stride8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, d, iters, active, same, stride8);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  s /= 256;
  // clock64 ticks at the 100 MHz reference on gfx9: report per-instruction time in ns as well via a wall clock below
  printf("%-28s active=%2d same=%d stride=%2d : %8.1f ticks/CU total, %6.3f ticks per wave-instruction (16 waves x %d iters)\n", name, active, same,
         stride8, s, s / (16.0 * iters), iters);
  hipFree(d);
}

template <int MODE>
static void timed(const char* name, int active, int same, int stride8, int threads = 1024) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8);
  const int iters = 4096;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 131072, 0, d, iters, active, same, stride8);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 131072, 0, d, iters, active, same, stride8);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s active=%2d same=%d stride=%2d waves=%2d : %7.3f ms -> %6.2f ns per wave-instruction per CU, %6.1f ns per instruction of one wave\n", name, active,
         same, stride8, threads / 64, ms, ms * 1e6 / ((threads / 64.0) * iters), ms * 1e6 / iters);
  hipFree(d);
}

int main() {
  for (int active : {64, 16, 4}) {
    timed<0>("ds_add_u64", active, 0, 1);
    timed<1>("ds_add_u32", active, 0, 1);
  }
  timed<0>("ds_add_u64 stride 10 (80B)", 64, 0, 10);
  timed<0>("ds_add_u64 stride 10 (80B)", 16, 0, 10);
  timed<0>("ds_add_u64 same address", 64, 1, 1);
  timed<0>("ds_add_u64 same address", 16, 1, 1);
  timed<1>("ds_add_u32 same address", 64, 1, 1);
  timed<2>("ds_add_rtn_u64", 64, 0, 1);
  timed<3>("ds_read_b64 + ds_write_b64", 64, 0, 1);
  timed<4>("ds_read_b64", 64, 0, 1);
  timed<5>("ds_bpermute_b32", 64, 0, 1);
  for (int threads : {64, 256, 512, 1024}) {
    timed<0>("ds_add_u64", 64, 0, 1, threads);
    timed<4>("ds_read_b64", 64, 0, 1, threads);
  }
  return 0;
}
