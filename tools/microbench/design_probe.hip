// design_probe.hip — round 3 measurements behind the fused reconcile kernel, the packed LDS folds and the single-pod
// PreFilter path (gfx950).   hipcc --offload-arch=gfx950 -O3 design_probe.hip -o design_probe && ./design_probe
//   grid barrier      : 256 x 1024-thread workgroups meet at an agent-scope counter (release fence / acquire spin)
//   global atomics    : 64-bit atomicAdd from every CU onto A distinct addresses (direct accumulation vs slabs)
//   LDS folds         : K independent ds_add_u64 per step and wave, 4 / 8 waves per SIMD
//   launch chain      : 1 / 3 dependent tiny kernels on one stream; launch + sync of one tiny kernel; a kernel that
//                       writes a pinned host word the host spins on (no hipStreamSynchronize)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#define LDS __attribute__((address_space(3)))
extern __shared__ unsigned char smem[];
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)

// ------------------------------------------------------------------ grid barrier
__global__ __launch_bounds__(1024) void k_barrier(unsigned int* ctr, unsigned long long* buf, int rounds, unsigned long long* out) {
  const long long t0 = clock64();
  unsigned int target = 0;
  for (int r = 0; r < rounds; ++r) {
    buf[(size_t)blockIdx.x * 1024 + threadIdx.x] = (unsigned long long)r;  // something the release has to publish
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __threadfence();
      (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // read what another workgroup published
    const unsigned long long v = __hip_atomic_load(buf + (size_t)((blockIdx.x + 1) % gridDim.x) * 1024 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != (unsigned long long)r) out[1] = 0xBADull;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(clock64() - t0);
}

// ------------------------------------------------------------------ global atomics
__global__ __launch_bounds__(1024) void k_gatomic(unsigned long long* tab, uint32_t n_addr, int per_thread) {
  uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    atomicAdd(tab + (x >> 8) % n_addr, 1ull);
  }
}

// ------------------------------------------------------------------ LDS folds
template <int K>
__global__ __launch_bounds__(1024) void k_ldsfold(unsigned long long* out, int iters) {
  LDS unsigned long long* t = (LDS unsigned long long*)smem;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) t[i] = 0;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    const uint32_t base = (x >> 10) & 1023u;  // a random 64-byte record
#pragma unroll
    for (int k = 0; k < K; ++k)
      (void)__hip_atomic_fetch_add(t + base * 8 + k, (unsigned long long)(it + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t[5];
}

// ------------------------------------------------------------------ launch chain
__global__ void k_tiny(unsigned long long* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
__global__ void k_flag(volatile unsigned long long* host_flag, unsigned long long v) {
  if (threadIdx.x == 0) {
    *host_flag = v;
    __threadfence_system();
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // ---- grid barrier
  {
    unsigned int* ctr;
    unsigned long long *buf, *out;
    CK(hipMalloc(&ctr, 4));
    CK(hipMalloc(&buf, 256 * 1024 * 8));
    CK(hipMalloc(&out, 16));
    CK(hipFuncSetAttribute((const void*)k_barrier, hipFuncAttributeMaxDynamicSharedMemorySize, 65 * 1024 * 2));
    for (int grid : {256, 128}) {
      for (int rounds : {1, 101}) {
        float ms[3];
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemsetAsync(ctr, 0, 4, s));
          CK(hipMemsetAsync(out, 0, 16, s));
          CK(hipEventRecord(e0, s));
          hipLaunchKernelGGL(k_barrier, dim3(grid), dim3(1024), 65 * 1024 * 2, s, ctr, buf, rounds, out);
          CK(hipEventRecord(e1, s));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms[rep], e0, e1));
        }
        unsigned long long h[2];
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("grid barrier: grid=%d rounds=%3d : %.1f us kernel (%.1f / %.1f), stale reads: %s\n", grid, rounds, ms[2] * 1e3, ms[0] * 1e3,
               ms[1] * 1e3, h[1] == 0xBADull ? "YES" : "none");
      }
    }
  }
  // ---- global atomics
  {
    unsigned long long* tab;
    CK(hipMalloc(&tab, 1 << 24));
    CK(hipMemset(tab, 0, 1 << 24));
    for (uint32_t n_addr : {18000u, 180000u, 1800000u}) {
      for (int per : {1, 4, 16}) {
        float ms = 0;
        hipLaunchKernelGGL(k_gatomic, dim3(256), dim3(1024), 0, s, tab, n_addr, per);
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_gatomic, dim3(256), dim3(1024), 0, s, tab, n_addr, per);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("global atomicAdd u64: %8u addresses, %7.0f k atomics : %8.1f us -> %6.2f atomics / ns\n", n_addr, 256.0 * 1024 * per / 1e3, ms * 1e3,
               256.0 * 1024 * per / (ms * 1e6));
      }
    }
  }
  // ---- LDS folds
  {
    unsigned long long* out;
    CK(hipMalloc(&out, 512 * 8));
    auto run = [&](auto kern, int K, int wgs_per_cu, int threads) -> int {
      const int iters = 2048;
      float ms = 0;
      const size_t lds = wgs_per_cu == 1 ? 128 * 1024 : 70 * 1024;
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(threads), lds, s, out, iters);
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(threads), lds, s, out, iters);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double waves = wgs_per_cu * threads / 64.0;
      printf("LDS fold: K=%d independent ds_add_u64 per step, %2.0f waves / CU : %7.3f ms -> %6.2f ns per wave-instruction per CU, %6.1f ns per step of one wave\n", K,
             waves, ms, ms * 1e6 / (waves * iters * K), ms * 1e6 / iters);
      return 0;
    };
    for (int wgs : {1, 2}) {
      if (run(k_ldsfold<1>, 1, wgs, 1024)) return 1;
      if (run(k_ldsfold<2>, 2, wgs, 1024)) return 1;
      if (run(k_ldsfold<4>, 4, wgs, 1024)) return 1;
      if (run(k_ldsfold<8>, 8, wgs, 1024)) return 1;
    }
  }
  // ---- launch chain
  {
    unsigned long long* d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    unsigned long long* hflag;
    CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped));
    *hflag = 0;
    for (int chain : {1, 3}) {
      std::vector<double> t;
      for (int rep = 0; rep < 2000; ++rep) {
        const double a = now_us();
        for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
        CK(hipStreamSynchronize(s));
        t.push_back(now_us() - a);
      }
      std::sort(t.begin(), t.end());
      printf("launch chain: %d tiny kernel(s) + hipStreamSynchronize : p50 %.1f us, p99 %.1f us\n", chain, t[t.size() / 2], t[t.size() * 99 / 100]);
    }
    {
      std::vector<double> t;
      for (int rep = 1; rep <= 2000; ++rep) {
        const double a = now_us();
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, hflag, (unsigned long long)rep);
        while (*(volatile unsigned long long*)hflag != (unsigned long long)rep) {
        }
        t.push_back(now_us() - a);
      }
      CK(hipStreamSynchronize(s));
      std::sort(t.begin(), t.end());
      printf("launch + host spin on a pinned word the kernel writes : p50 %.1f us, p99 %.1f us\n", t[t.size() / 2], t[t.size() * 99 / 100]);
    }
    {
      // device steady state of dependent launches: 300 tiny kernels back to back, one sync
      const double a = now_us();
      for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
      CK(hipStreamSynchronize(s));
      printf("300 dependent tiny kernels on one stream: %.2f us per launch (device-side launch-to-launch)\n", (now_us() - a) / 300);
    }
    {
      std::vector<double> t;
      unsigned long long hv = 0;
      for (int rep = 0; rep < 2000; ++rep) {
        const double a = now_us();
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
        CK(hipMemcpyAsync(&hv, d, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        t.push_back(now_us() - a);
      }
      std::sort(t.begin(), t.end());
      printf("launch + 8-byte D2H copy + hipStreamSynchronize : p50 %.1f us, p99 %.1f us\n", t[t.size() / 2], t[t.size() * 99 / 100]);
    }
  }
  return 0;
}
