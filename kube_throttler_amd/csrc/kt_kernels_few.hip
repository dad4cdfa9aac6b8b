// kt_kernels_few.hip — kt_check_few: PreFilter (plugin.go:148-215) for ONE pod (up to 8), the scheduler's actual call pattern.
//
// The sweep kernels (kt_kernels_check.hip) stage a chunk of the selector index in LDS and amortise that over thousands
// of tiles; for one pod the staging IS the cost, a 1024-thread workgroup with 70+ KB of LDS cannot start beside a
// reconcile sweep, and the result has to come back through a copy and a stream synchronisation.  This kernel is the
// opposite shape:
//   * grid = index chunks, ONE wave per workgroup, no LDS: it fits into the wave slots and registers a running sweep
//     leaves free, on the engine's high-priority stream;
//   * lane = (pod, word of the pod's namespace list in this chunk): all words of the list are decided in parallel,
//     straight from the chunk image in global memory (L2-resident), with the same exact term-bitmap algebra as
//     scan_tile (kt_scan.h): any / two / three / veto accumulators over the pod's atom rows;
//   * every match takes the full CheckThrottledFor comparison (classify(), kt_kernels_common.h) against the CheckRec;
//   * the class counters of the chunks meet by atomics in a scratch word; the last workgroup to arrive (ticket) writes
//     the summary words to PINNED HOST memory followed by a sequence number the caller spins on — no copy, no
//     hipStreamSynchronize — and leaves scratch and ticket zeroed for the next call.
// Programs with `slow` term shapes, slow-list throttles or overflow pods are not dispatched here (the host falls back
// to the staged small launch).
#include "kt_scan.h"

namespace kt {

struct FewArgs {
  const uint64_t* meta;  // pod tables
  const uint16_t* latom;
  const int64_t* req;
  const void* recs;
  const uint8_t* ns_valid;
  const unsigned char* blob;  // chunk images
  const BmChunk* chunks;
  unsigned long long* acc;  // [8] class counters meeting across chunks; zero between launches
  uint32_t* ticket;         // [1] arrival counter; zero between launches
  uint64_t* host_summary;   // pinned host: [8] summary words ...
  uint64_t* host_seq;       // ... and the sequence number written after them
  uint64_t seq;
  uint32_t n_chunks, n, lanes_per_pod;
  int32_t DS, T;
  int64_t rows[8];
};

template <int DT, int LA, bool VETO, int NEED>
__global__ __launch_bounds__(64) void kt_check_few(const FewArgs a) {
  const uint32_t lane = threadIdx.x;
  const uint32_t ci = blockIdx.x;
  const BmChunk ch = a.chunks[ci];
  const unsigned char* img = a.blob + ch.img_off;
  const uint32_t W = a.lanes_per_pod;  // power of two, >= 8
  const uint32_t i = lane / W, j = lane & (W - 1u);
  const bool in = i < a.n;
  const int64_t p = a.rows[in ? i : 0u];
  const uint64_t meta = a.meta[p];
  const bool on = in && ((meta >> kMetaStateShift) & kPodValid) != 0;
  const uint32_t ns = on ? (uint32_t)(meta & kMetaNsMask) : 0u;
  const uint32_t nz = (uint32_t)(meta >> kMetaNzShift);
  u32x4 raw[LA / 8];
  load_atoms<LA>(a.latom, p, raw);
  uint32_t ro[LA];
  atom_row_offsets<LA>(raw, ro);
  int64_t v[DT];
  load_requests<DT>(a.req, a.DS, p, v);
  const u32x2* nsl_rng = (const u32x2*)(img + ch.off_nsl_rng);
  const NsWord* nsl = (const NsWord*)(img + ch.off_nsl);
  const WordHdr* hdr = (const WordHdr*)(img + ch.off_hdr);
  const uint32_t* term_t = (const uint32_t*)(img + ch.off_term_t);
  const CheckRec<DT>* recs = (const CheckRec<DT>*)a.recs;
  const u32x2 rng = nsl_rng[ns];
  uint32_t k = rng.x + j;
  const uint32_t k1 = on ? rng.y : 0u;
  unsigned long long my = 0;
  for (; k < k1; k += W) {  // normally one trip: a namespace has a handful of words per chunk
    const u32x4 e = *(const u32x4*)(nsl + k);  // {w, -, mask lo, mask hi}
    const uint32_t w = e.x;
    const WordHdr h = hdr[w];
    uint64_t any = h.univ, two = 0, three = 0, four = 0, five = 0, vet = 0;
    const unsigned char* col = img + (size_t)w * ch.col_rows * 8u;  // the word's column of the `any` plane
    // (a word without a veto column reads the plane's all-zero column: kt_index.h, image layout)
    const unsigned char* colv = (ch.zero_col == 0u || (e.y & kNsWordVeto) != 0u) ? col + (size_t)ch.n_words * ch.col_rows * 8u : img + ch.zero_col;
#pragma unroll
    for (int l = 0; l < LA; ++l) {
      const uint64_t r = *(const unsigned long long*)(col + ro[l]);
      if (VETO) vet |= *(const unsigned long long*)(colv + ro[l]);
      if (NEED >= 5) five |= four & r;
      if (NEED >= 4) four |= three & r;
      if (NEED >= 3) three |= two & r;
      if (NEED >= 2) two |= any & r;
      any |= r;
    }
    uint64_t xx = any;
    if (NEED >= 2) xx = (any & ~h.m2) | (two & h.m2);
    if (NEED >= 3) xx = (xx & ~h.m3) | (three & h.m3);
    if (NEED >= 4) xx = (xx & ~h.m4) | (four & h.m4);
    if (NEED >= 5) xx = (xx & ~h.m5) | (five & h.m5);
    xx &= ~vet & ((uint64_t)e.z | (uint64_t)e.w << 32);
    uint32_t last_t = 0xFFFFFFFFu;
    while (xx) {
      const uint32_t bit = (uint32_t)__ffsll((unsigned long long)xx) - 1u;
      xx &= xx - 1ull;
      const uint32_t tt = term_t[w * 64u + bit];
      const uint32_t t = tt & kTermRowMask;
      // a throttle with several terms is reported once: its copies sit side by side in one word
      const bool dup = (tt & kTermAdj) && t == last_t;
      last_t = t;
      if (dup) continue;
      const uint32_t st = classify<DT>(recs + t, v, nz);
      my += st == 4u ? 1ull << 4 : st == 2u ? 1ull << 24 : st == 3u ? 1ull << 44 : 0ull;
    }
  }
  // the W lanes of a pod
  for (uint32_t o = W >> 1; o >= 1u; o >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)my, (int)o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(my >> 32), (int)o);
    my += (unsigned long long)lo | (unsigned long long)hi << 32;
  }
  if (a.n_chunks == 1u) {  // one workgroup saw everything: the words go straight to the host (1M x 1k: the usual case)
    if (in && j == 0u) {
      const bool err = on && a.ns_valid[ns] == 0;
      a.host_summary[i] = !on ? 0ull : err ? 2ull : (my | (my ? 1ull : 0ull));
    }
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if (in && j == 0u) {
    if (my) (void)__hip_atomic_fetch_add(a.acc + i, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // affectedClusterThrottles: the pod's Namespace object must exist (clusterthrottle_controller.go:273-276)
    if (ci == 0u && on && a.ns_valid[ns] == 0) (void)__hip_atomic_fetch_or(a.acc + i, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __threadfence();
  uint32_t arrived = 0;
  if (lane == 0) arrived = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  arrived = __builtin_amdgcn_readfirstlane(arrived);
  if (arrived + 1u != a.n_chunks) return;
  // ---- last workgroup: final form of the words, to the host; scratch back to zero
  if (lane < a.n) {
    const int64_t pr = a.rows[lane];
    const bool valid = ((a.meta[pr] >> kMetaStateShift) & kPodValid) != 0;
    const unsigned long long wv = __hip_atomic_load(a.acc + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long c = wv & ~3ull;
    a.host_summary[lane] = !valid ? 0ull : (wv & 2ull) ? 2ull : (c | (c ? 1ull : 0ull));
    __hip_atomic_store(a.acc + lane, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  if (lane == 0) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define KT_FEW_LAUNCH(DT_, LA_, VETO_, NEED_) hipLaunchKernelGGL((kt_check_few<DT_, LA_, VETO_, NEED_>), dim3(ix.n_chunks), dim3(64), 0, s, a)

// n <= 8 pod rows (host memory), summaries to pinned host memory + sequence number; false when the program / pods need
// a path this kernel does not have
bool launch_check_few(const PodTable& pods, int n, const int64_t* rows_host, const SelProgram& sp, const IndexDev& ix, const void* recs,
                      unsigned long long* acc, uint32_t* ticket, uint64_t* host_summary, uint64_t* host_seq, uint64_t seq, hipStream_t s) {
  // (ix.has_long: a throttle whose run of term numbers spans words — lane = (pod, word) cannot apply "reported once" across lanes)
  // (ix.max_need > 3: terms with four or five positive keys — the staged small launch has the NEED = 5 instantiation)
  if (n < 1 || n > 8 || ix.n_slow != 0 || ix.has_long || ix.max_need > 3u || ix.n_chunks == 0) return false;
  FewArgs a{};
  a.meta = pods.meta, a.latom = pods.latom, a.req = pods.req, a.recs = recs, a.ns_valid = sp.ns_valid;
  a.blob = ix.bm_blob, a.chunks = ix.bm_chunks, a.acc = acc, a.ticket = ticket, a.host_summary = host_summary, a.host_seq = host_seq;
  a.seq = seq, a.n_chunks = ix.n_chunks, a.n = (uint32_t)n, a.lanes_per_pod = n == 1 ? 64u : n == 2 ? 32u : n <= 4 ? 16u : 8u;
  a.DS = pods.DS, a.T = sp.T;
  for (int k = 0; k < 8; ++k) a.rows[k] = rows_host[k < n ? k : n - 1];
  const int DT = dt_bucket_ix(pods.D), LA = pods.LA;
#ifdef KT_FAST_BUILD
  KT_FEW_LAUNCH(8, 8, false, 2);
#else
  if (!ix.rich) { if (DT <= 8) KT_FEW_LAUNCH(8, 8, false, 2); else KT_FEW_LAUNCH(16, 8, false, 2); }
  else if (LA <= 8) { if (DT <= 8) KT_FEW_LAUNCH(8, 8, true, 3); else KT_FEW_LAUNCH(16, 8, true, 3); }
  else if (LA <= 16) { if (DT <= 8) KT_FEW_LAUNCH(8, 16, true, 3); else KT_FEW_LAUNCH(16, 16, true, 3); }
  else { if (DT <= 8) KT_FEW_LAUNCH(8, 32, true, 3); else KT_FEW_LAUNCH(16, 32, true, 3); }
#endif
  return true;
}

}  // namespace kt
