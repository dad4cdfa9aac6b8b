// meet_litmus.hip — litmus / stress test of the multi-group meet of kt_reduce_finalize_packed (kt_kernels.hip): G groups on
// different XCDs add their sums of one throttle to the throttle's partial row with RETURNING relaxed atomics, the returned
// values feed the operand of the arrival ticket (ordering by data dependence, no release fence), and the group that
// arrives last takes the row by read-modify-writes (exchange with 0).  Every lane sequence below is the one of the
// kernel (thread = (record, dimension), DT = 8 lanes per throttle, lane 0 adds the pod count and takes the ticket).
//
// Zero tolerance: over `launches` launches of `rows` rows the last arriver must see the exact total of every word, every
// row must be taken exactly once, and rows + tickets must be left zeroed.  Variants:
//   rmw      the product's form (the last arriver reads with atomic exchange)
//   plain    the last arriver reads with plain loads (what round 2 did) — shown for contrast: may legitimately FAIL
//   fenced   release / acquire at agent scope around the ticket (the textbook form), timed against `rmw`
// Groups are consecutive workgroup ids (round-robin over the 8 XCDs), delayed by a per-(launch, group) pseudo-random spin.
//   hipcc --offload-arch=gfx950 -O3 meet_litmus.hip -o meet_litmus && ./meet_litmus [launches] [rows]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r_)); return 1; } } while (0)

constexpr int DT = 8, D = 8, STRIDE = 2 * D + 2;
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xFu; }
__device__ __forceinline__ uint32_t group_bits(bool b) {
  const uint64_t m = __ballot(b);
  const uint32_t lane = threadIdx.x & 63u;
  return (uint32_t)(m >> (lane & ~(uint32_t)(DT - 1))) & ((1u << DT) - 1u);
}
__host__ __device__ inline uint32_t mix(uint32_t a, uint32_t b) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
  h ^= h >> 15, h *= 0xC2B2AE35u, h ^= h >> 13;
  return h;
}
// what group g contributes to word j of row t in launch `it`
__host__ __device__ inline unsigned long long contrib(uint32_t it, uint32_t t, uint32_t g, uint32_t j) {
  return 1ull + (mix(it * 131u + g, t * 31u + j) & 0xFFFFFu);
}

template <int MODE>  // 0 rmw, 1 plain, 2 fenced
__global__ __launch_bounds__(64) void k_meet(unsigned long long* partial, uint32_t* arrive, uint32_t rows, uint32_t G, uint32_t it,
                                             unsigned long long* errors, unsigned long long* taken, uint32_t* xcd_seen) {
  const uint32_t x = threadIdx.x, d = x & (DT - 1), gq = x / DT;
  const uint32_t g = blockIdx.x % G, tile = blockIdx.x / G;  // consecutive workgroups (different XCDs) meet in the same rows
  const uint32_t t = tile * (64 / DT) + gq;
  const bool valid = t < rows;
  if (d == 0 && valid) atomicOr(xcd_seen + t, 1u << xcc_id());
  // a spin that differs per (launch, group): the groups do not arrive in lockstep
  const uint32_t spin = mix(it, blockIdx.x) & 0x3FFu;
  for (uint32_t i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(1);
  unsigned long long* prow = partial + (size_t)(valid ? t : 0) * STRIDE;
  const unsigned long long rec_pods = contrib(it, t, g, 2 * D), mine = contrib(it, t, g, d);
  const bool key_mark = (mix(it + g, t + d) & 3u) == 0u;
  unsigned long long seen = 0;
  if (valid) {
    if (d == 0) seen |= atomicAdd(prow + 2 * D, rec_pods);
    if (d < D) {
      seen |= atomicAdd(prow + d, mine);
      if (key_mark) seen |= atomicAdd(prow + D + d, 1ull);
    }
  }
  if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const uint32_t one = 1u + (group_bits(seen == ~0ull) != 0u ? 1u : 0u);
  uint32_t arrived = 0;
  if (valid && d == 0) arrived = __hip_atomic_fetch_add(arrive + t, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  arrived = (uint32_t)__shfl((int)arrived, (int)((x & 63u) & ~(uint32_t)(DT - 1)));
  const bool last = valid && arrived + 1u == G;
  if (last && d == 0) __hip_atomic_store(arrive + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!last) return;
  if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  unsigned long long pv, pc, p0 = 0;
  if (MODE == 1) {
    pv = prow[d], pc = prow[D + d];
    if (d == 0) p0 = prow[2 * D];
    prow[d] = 0ull, prow[D + d] = 0ull;
    if (d == 0) prow[2 * D] = 0ull;
  } else {
    pv = __hip_atomic_exchange(prow + d, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pc = __hip_atomic_exchange(prow + D + d, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == 0) p0 = __hip_atomic_exchange(prow + 2 * D, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned long long wv = 0, wc = 0, wp = 0;
  for (uint32_t gg = 0; gg < G; ++gg) {
    wv += contrib(it, t, gg, d);
    wc += (mix(it + gg, t + d) & 3u) == 0u ? 1ull : 0ull;
    wp += contrib(it, t, gg, 2 * D);
  }
  const uint32_t bad = (pv != wv ? 1u : 0u) + (pc != wc ? 1u : 0u) + ((d == 0 && p0 != wp) ? 1u : 0u);
  if (bad) atomicAdd(errors, (unsigned long long)bad);
  if (d == 0) atomicAdd(taken, 1ull);
}

int main(int argc, char** argv) {
  const uint32_t launches = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000u;
  const uint32_t rows = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024u;
  unsigned long long *partial, *errors, *taken;
  uint32_t *arrive, *xcd_seen;
  CK(hipMalloc(&partial, (size_t)rows * STRIDE * 8));
  CK(hipMalloc(&arrive, (size_t)rows * 4));
  CK(hipMalloc(&xcd_seen, (size_t)rows * 4));
  CK(hipMalloc(&errors, 8));
  CK(hipMalloc(&taken, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[3] = {"rmw (product)", "plain loads  ", "fenced       "};
  int rc = 0;
  for (uint32_t G : {2u, 8u, 16u}) {
    for (int mode = 0; mode < 3; ++mode) {
      CK(hipMemset(partial, 0, (size_t)rows * STRIDE * 8));
      CK(hipMemset(arrive, 0, (size_t)rows * 4));
      CK(hipMemset(xcd_seen, 0, (size_t)rows * 4));
      CK(hipMemset(errors, 0, 8));
      CK(hipMemset(taken, 0, 8));
      const uint32_t tiles = (rows + 64 / DT - 1) / (64 / DT);
      const dim3 grid(tiles * G), block(64);
      CK(hipEventRecord(e0, 0));
      for (uint32_t it = 0; it < launches; ++it) {
        if (mode == 0) hipLaunchKernelGGL(k_meet<0>, grid, block, 0, 0, partial, arrive, rows, G, it, errors, taken, xcd_seen);
        else if (mode == 1) hipLaunchKernelGGL(k_meet<1>, grid, block, 0, 0, partial, arrive, rows, G, it, errors, taken, xcd_seen);
        else hipLaunchKernelGGL(k_meet<2>, grid, block, 0, 0, partial, arrive, rows, G, it, errors, taken, xcd_seen);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h_err = 0, h_taken = 0;
      CK(hipMemcpy(&h_err, errors, 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&h_taken, taken, 8, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> hp((size_t)rows * STRIDE);
      std::vector<uint32_t> ha(rows), hx(rows);
      CK(hipMemcpy(hp.data(), partial, hp.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ha.data(), arrive, rows * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx.data(), xcd_seen, rows * 4, hipMemcpyDeviceToHost));
      unsigned long long left = 0;
      for (auto v : hp) left += v != 0;
      for (auto v : ha) left += v != 0;
      double xcds = 0;
      for (auto v : hx) xcds += __builtin_popcount(v);
      const bool ok = h_err == 0 && h_taken == (unsigned long long)rows * launches && left == 0;
      printf("G=%2u %s: %u launches x %u rows = %.2e meets, wrong words %llu, rows taken %llu (want %llu), words left non-zero %llu, "
             "XCDs per row %.1f, %.2f us / launch  -> %s\n", G, names[mode], launches, rows, (double)rows * launches, h_err, h_taken,
             (unsigned long long)rows * launches, left, xcds / rows, ms * 1e3 / launches, ok ? "EXACT" : "FAILED");
      if (!ok && mode != 1) rc = 1;
    }
  }
  return rc;
}
