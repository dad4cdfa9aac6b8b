// kt_kernels.hip — hand-written HIP kernels for gfx950 (CDNA4, wave64), pod side: ingest (PodRequestResourceList on device), atom
// translation, scan lists and views, the pod event kernels.  Per-throttle finalize / check-record preparation and the DENSE pod x
// throttle scans: kt_kernels_finalize.hip; the indexed scans: kt_kernels_check.hip / kt_kernels_aggregate.hip (kt_scan.h).
//
// Everything here is integer / compare work on row tables in HBM: no MFMA, no floating point.
#include "kt_index_device.h"

namespace kt {

typedef uint32_t u128 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;

static inline int grid_for(int64_t n, int per_block = kBlock, int max_blocks = 256 * 8) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// ---------------------------------------------------------------------------------------------------
// Pod ingest: staged batch -> the pod tables (one row per pod), computing the pod's effective request on the way.
// Restates resourcelist.PodRequestResourceList (pkg/resourcelist/resourcelist.go:27-46):
//   ic = SetMax over initContainers (missing key => copy, :76-84); c = Add over containers (key created
//   even for +0, :48-54); c.SetMax(ic); c.Add(overhead) when overhead != nil.
// One thread per pod; D and the container count are tiny, the kernel is bound by the plane writes.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ingest_one(const PodTable& pods, const PodBatchDev& b, int64_t i) {
  const int D = pods.D, L = pods.L;
  const int64_t row = b.rows ? b.rows[i] : b.row0 + i;
  int64_t c[16], ic[16];
  uint32_t cp = 0, icp = 0;
#pragma unroll
  for (int d = 0; d < 16; ++d) c[d] = 0, ic[d] = 0;
  const uint32_t k0 = b.ctr_off[i] - b.ctr_base, k1 = b.ctr_off[i + 1] - b.ctr_base;
  for (uint32_t k = k0; k < k1; ++k) {
    const uint32_t pm = b.ctr_present[k];
    const bool init = b.ctr_init[k] != 0;
    for (int d = 0; d < D; ++d) {
      if (!((pm >> d) & 1u)) continue;
      const int64_t q = b.ctr_req[(int64_t)k * D + d];
      if (init) {
        ic[d] = ((icp >> d) & 1u) ? (ic[d] >= q ? ic[d] : q) : q;
      } else {
        c[d] += q;
      }
    }
    if (init) icp |= pm; else cp |= pm;
  }
  for (int d = 0; d < D; ++d)
    if ((icp >> d) & 1u) c[d] = ((cp >> d) & 1u) ? (c[d] >= ic[d] ? c[d] : ic[d]) : ic[d];
  cp |= icp;
  const uint32_t op = b.ovh_present[i];
  if (op >> 31) {
    for (int d = 0; d < D; ++d)
      if ((op >> d) & 1u) c[d] += b.ovh[i * D + d];
    cp |= op & 0xFFFFu;
  }
  cp &= (1u << D) - 1u;
  uint32_t nz = 0;
  for (int d = 0; d < D; ++d) nz |= (((cp >> d) & 1u) && c[d] != 0 ? 1u : 0u) << d;
  for (int d = 0; d < pods.DS; ++d) pods.req[(int64_t)row * pods.DS + d] = (d < D && ((cp >> d) & 1u)) ? c[d] : 0;
  pods.ns[row] = b.ns[i];
  pods.flags[row] = (b.flags[i] & 0xFu) | (cp << kPresentShift);
  // the record the indexed scans stream (kMeta*); the atom row follows from kt_translate_pods
  pods.meta[row] = (uint64_t)(b.ns[i] & (uint32_t)kMetaNsMask) | (uint64_t)(b.flags[i] & 0xFu) << kMetaStateShift |
                   (uint64_t)cp << kMetaPresentShift | (uint64_t)nz << kMetaNzShift;
  const uint32_t l0 = b.label_off[i] - b.label_base, l1 = b.label_off[i + 1] - b.label_base;
  for (int l = 0; l < pods.LS; ++l) {
    const bool have = l < L && l0 + l < l1;
    pods.lpair[(int64_t)row * pods.LS + l] = have ? b.label_pair[l0 + l] : 0u;
    pods.lkey[(int64_t)row * pods.LS + l] = have ? b.label_key[l0 + l] : 0u;
  }
}
__global__ __launch_bounds__(kBlock) void kt_ingest_pods(PodTable pods, PodBatchDev b) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < b.n; i += (int64_t)gridDim.x * kBlock) ingest_one(pods, b, i);
}

__global__ __launch_bounds__(kBlock) void kt_delete_pods(PodTable pods, int64_t n, const int64_t* rows) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    pods.flags[rows[i]] = 0;
    pods.meta[rows[i]] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------
// kt_translate_pods — a pod's labels as the ids of the atoms the selector program REFERENCES (kt_index.h): every
// (key,value) pair id and every key id is looked up in the index's open-addressing table; labels no selector mentions
// are dropped.  Runs over all pods after a program change and over the ingested rows after an upsert.  One thread per
// pod; the atom row is assembled in LDS and written with 16-byte stores.
// ---------------------------------------------------------------------------------------------------
// -> id | home slot << 16 (0: the program does not refer to the atom)
__device__ __forceinline__ uint32_t atom_id_of(const uint64_t* table, uint32_t mask, uint32_t atom) {
  uint32_t s = atom_slot(atom, mask);
  for (;;) {
    const uint64_t e = table[s];
    if (e == 0ull) return 0u;
    if ((uint32_t)e == atom) return (uint32_t)(e >> 32);
    s = (s + 1) & mask;
  }
}

// one pod row; out: the thread's LA slots of LDS scratch
template <int LA>
__device__ __forceinline__ void translate_one(const PodTable& pods, int64_t row, const uint64_t* table, uint32_t mask, int key_atoms,
                                              unsigned long long* n_overflow, uint16_t* out) {
  const int LS = pods.LS;
  uint32_t cnt = 0;
  // Every atom goes to the HOME slot of its key when that is free (kt_index.cpp: number_atoms_by_home_slot — the lanes of a
  // scan then read, slot by slot, atoms that were numbered side by side: distinct LDS banks); the few that find it taken
  // (two keys of one pod with the same home: only programs that name more keys than there are slots) take the free slots
  // in label order afterwards.  Any order is CORRECT: the scans accumulate the rows with symmetric functions.
  uint32_t taken = 0;           // slots in use
  unsigned long long late = 0;  // labels whose atom found its home taken
#pragma unroll
  for (int k = 0; k < LA; ++k) out[k] = 0;
  for (int l = 0; l < LS; ++l) {
    const uint32_t pr = pods.lpair[row * LS + l];
    if (pr == 0u) continue;  // empty slot
    // one atom per label: the pair when some selector names it, else the key atom when some selector names the key
    uint32_t e = atom_id_of(table, mask, pr);
    if (!e && key_atoms) e = atom_id_of(table, mask, kKeyAtom | pods.lkey[row * LS + l]);
    if (e) {
      const uint32_t home = (e >> 16) & (uint32_t)(LA - 1);
      if (!((taken >> home) & 1u)) out[home] = (uint16_t)e, taken |= 1u << home;
      else late |= 1ull << l;  // (LS <= 64)
      ++cnt;
    }
  }
  if (late != 0ull && cnt <= (uint32_t)LA) {
    for (int l = 0; l < LS; ++l) {
      if (!((late >> l) & 1ull)) continue;
      uint32_t e = atom_id_of(table, mask, pods.lpair[row * LS + l]);
      if (!e) e = atom_id_of(table, mask, kKeyAtom | pods.lkey[row * LS + l]);
      const uint32_t free_slot = (uint32_t)__ffs((int)~taken) - 1u;  // (cnt <= LA: there is one)
      out[free_slot] = (uint16_t)e, taken |= 1u << free_slot;
    }
  }
  const bool over = cnt > (uint32_t)LA;
  const uint64_t m = pods.meta[row];
  if (over) {
    if ((m >> kMetaStateShift) & kPodValid) atomicAdd(n_overflow, 1ull);
    pods.meta[row] = m | kMetaOverflow;
  } else if (m & kMetaOverflow) {
    pods.meta[row] = m & ~kMetaOverflow;
  }
  const kt_u32x4* src = (const kt_u32x4*)out;
  kt_u32x4* dst = (kt_u32x4*)(pods.latom + row * LA);
#pragma unroll
  for (int q = 0; q < LA / 8; ++q) dst[q] = src[q];
}

template <int LA>
__global__ __launch_bounds__(kBlock) void kt_translate_pods(PodTable pods, int64_t n, const int64_t* rows, int64_t row0,
                                                           const uint64_t* table, uint32_t mask, int key_atoms,
                                                           unsigned long long* n_overflow) {
  __shared__ __attribute__((aligned(16))) uint16_t out[kBlock][LA];
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n; i0 += (int64_t)gridDim.x * kBlock) {
    const int64_t i = i0 + threadIdx.x;
    if (i < n) translate_one<LA>(pods, rows ? rows[i] : row0 + i, table, mask, key_atoms, n_overflow, &out[threadIdx.x][0]);
  }
}

void launch_translate_pods(const PodTable& pods, int64_t n, const int64_t* rows_dev, int64_t row0, const IndexDev& ix,
                           unsigned long long* n_overflow, hipStream_t s) {
  if (n <= 0) return;
  const dim3 g(grid_for(n)), b(kBlock);
  if (pods.LA == 8) hipLaunchKernelGGL(kt_translate_pods<8>, g, b, 0, s, pods, n, rows_dev, row0, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow);
  else if (pods.LA == 16) hipLaunchKernelGGL(kt_translate_pods<16>, g, b, 0, s, pods, n, rows_dev, row0, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow);
  else hipLaunchKernelGGL(kt_translate_pods<32>, g, b, 0, s, pods, n, rows_dev, row0, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow);
}

__global__ __launch_bounds__(kBlock) void kt_gather_pod_requests(PodTable pods, int64_t n, const int64_t* rows,
                                                                int64_t* out_v, uint32_t* out_present) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int64_t row = rows ? rows[i] : i;
    for (int d = 0; d < pods.D; ++d) out_v[i * pods.D + d] = pods.req[(int64_t)row * pods.DS + d];
    out_present[i] = pods.flags[row] >> kPresentShift;
  }
}

// kt_compact_countable — the rows of the pods a reconcile has to look at (shouldCountIn, throttle_controller.go:217-219:
// valid, scheduled by the target scheduler, bound to a node; finished ones included — they still matter for selector
// errors), as a dense list: the aggregate scan then spends no lanes on pending / foreign pods.  Rebuilt only after pod
// events.  Order: ascending inside a 1024-pod block, blocks in arrival order (sums do not care).
__global__ __launch_bounds__(1024) void kt_compact_countable(PodTable pods, int64_t n, int64_t* out_rows, unsigned long long* out_n) {
  // one atomic per 1024-pod block: wave ballots -> LDS counts -> block base
  __shared__ uint32_t wcnt[16];
  __shared__ unsigned long long bbase;
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t n_round = (n + 1023) / 1024 * 1024;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024; i0 < n_round; i0 += (int64_t)gridDim.x * 1024) {
    const int64_t i = i0 + threadIdx.x;
    const uint32_t st = i < n ? (uint32_t)(pods.meta[i] >> kMetaStateShift) & 0xFu : 0u;
    const bool countable = (st & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
    const uint64_t mk = __ballot(countable);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(mk);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (int w = 0; w < 16; ++w) {
        const uint32_t c = wcnt[w];
        wcnt[w] = tot;  // exclusive prefix
        tot += c;
      }
      bbase = tot ? atomicAdd(out_n, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
    if (countable) out_rows[bbase + wcnt[wave] + rank] = i;
    __syncthreads();  // wcnt / bbase are rewritten by the next round
  }
}
void launch_compact_countable(const PodTable& pods, int64_t n, int64_t* out_rows, unsigned long long* out_n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(kt_compact_countable, dim3(grid_for(n, 1024, 2048)), dim3(1024), 0, s, pods, n, out_rows, out_n);
}

// ---------------------------------------------------------------------------------------------------
// kt_order_rows_by_ns — pod rows ordered by namespace (counting sort: histogram, scan, scatter), the scan order of the
// indexed kernels when the selector index has several chunks: the 64 pods of a tile then share their namespace's word
// list (all lanes advance and peel together instead of waiting for each other's words), and a workgroup's contiguous
// range of tiles touches only the chunks that hold words of ITS namespaces.
//   countable_only: the rows kt_compact_countable would list (the aggregate's scan), else every row in [0, n)
// Order inside a namespace is arrival order of the atomics (sums and per-pod verdicts do not care).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool order_takes(uint64_t meta, bool countable_only) {
  const uint32_t st = (uint32_t)(meta >> kMetaStateShift) & 0xFu;
  return !countable_only || (st & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
}
constexpr uint32_t kNsLdsKeys = 12288;      // histogram: namespaces whose u32 counters fit 48 KB of LDS
constexpr uint32_t kNsLdsKeysScatter = 4096;  // scatter: counter + 64-bit base per namespace (12 bytes) in 48 KB
// Both passes keep a workgroup's per-namespace counters in LDS when the namespaces fit (one global atomic per
// (workgroup, namespace present in its 8192 rows) instead of one per row on a few hot words); more namespaces than that
// fall back to plain global atomics.
constexpr int kNsRowsPerBlock = 8192;
__global__ __launch_bounds__(1024) void kt_ns_histogram(const uint64_t* meta, int64_t n, int countable_only, uint32_t n_keys,
                                                       unsigned long long* counts) {
  extern __shared__ uint32_t ns_hist[];
  const bool in_lds = n_keys <= kNsLdsKeys;
  for (int64_t b0 = (int64_t)blockIdx.x * kNsRowsPerBlock; b0 < n; b0 += (int64_t)gridDim.x * kNsRowsPerBlock) {
    if (in_lds) {
      for (uint32_t k = threadIdx.x; k < n_keys; k += 1024) ns_hist[k] = 0u;
      __syncthreads();
    }
    for (int64_t i = b0 + threadIdx.x; i < min(b0 + (int64_t)kNsRowsPerBlock, n); i += 1024) {
      const uint64_t m = meta[i];
      if (!order_takes(m, countable_only != 0)) continue;
      const uint32_t key = min((uint32_t)(m & kMetaNsMask), n_keys - 1u);
      if (in_lds) atomicAdd(ns_hist + key, 1u);
      else atomicAdd(counts + key, 1ull);
    }
    if (in_lds) {
      __syncthreads();
      for (uint32_t k = threadIdx.x; k < n_keys; k += 1024)
        if (ns_hist[k]) atomicAdd(counts + k, (unsigned long long)ns_hist[k]);
      __syncthreads();
    }
  }
}
// one workgroup: counts[k] -> first position of key k (exclusive prefix sum), *total = number of listed rows
__global__ __launch_bounds__(1024) void kt_ns_scan(unsigned long long* counts, uint32_t n_keys, unsigned long long* total) {
  __shared__ unsigned long long part[1024];
  const uint32_t per = (n_keys + 1023u) / 1024u;
  const uint32_t k0 = threadIdx.x * per, k1 = min(k0 + per, n_keys);
  unsigned long long sum = 0;
  for (uint32_t k = k0; k < k1; ++k) sum += counts[k];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int t = 0; t < 1024; ++t) {
      const unsigned long long c = part[t];
      part[t] = run;
      run += c;
    }
    *total = run;
  }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  for (uint32_t k = k0; k < k1; ++k) {
    const unsigned long long c = counts[k];
    counts[k] = run;
    run += c;
  }
}
__global__ __launch_bounds__(1024) void kt_ns_scatter(const uint64_t* meta, int64_t n, int countable_only, uint32_t n_keys,
                                                     unsigned long long* cursor, int64_t* out_rows) {
  extern __shared__ uint32_t ns_hist[];  // [n_keys] rows of the block per namespace, then its local cursor | [n_keys] x 2 base
  const bool in_lds = n_keys <= kNsLdsKeysScatter;
  unsigned long long* base = (unsigned long long*)(ns_hist + ((n_keys + 1u) & ~1u));
  for (int64_t b0 = (int64_t)blockIdx.x * kNsRowsPerBlock; b0 < n; b0 += (int64_t)gridDim.x * kNsRowsPerBlock) {
    const int64_t b1 = min(b0 + (int64_t)kNsRowsPerBlock, n);
    if (!in_lds) {
      for (int64_t i = b0 + threadIdx.x; i < b1; i += 1024) {
        const uint64_t m = meta[i];
        if (order_takes(m, countable_only != 0)) out_rows[atomicAdd(cursor + min((uint32_t)(m & kMetaNsMask), n_keys - 1u), 1ull)] = i;
      }
      continue;
    }
    for (uint32_t k = threadIdx.x; k < n_keys; k += 1024) ns_hist[k] = 0u;
    __syncthreads();
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 1024) {
      const uint64_t m = meta[i];
      if (order_takes(m, countable_only != 0)) atomicAdd(ns_hist + min((uint32_t)(m & kMetaNsMask), n_keys - 1u), 1u);
    }
    __syncthreads();
    // the block reserves its rows' range of every namespace with ONE global atomic, then hands out the slots locally
    for (uint32_t k = threadIdx.x; k < n_keys; k += 1024) {
      const uint32_t c = ns_hist[k];
      base[k] = c ? atomicAdd(cursor + k, (unsigned long long)c) : 0ull;
      ns_hist[k] = 0u;
    }
    __syncthreads();
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 1024) {
      const uint64_t m = meta[i];
      if (!order_takes(m, countable_only != 0)) continue;
      const uint32_t key = min((uint32_t)(m & kMetaNsMask), n_keys - 1u);
      out_rows[base[key] + atomicAdd(ns_hist + key, 1u)] = i;
    }
    __syncthreads();
  }
}
void launch_order_rows_by_ns(const PodTable& pods, int64_t n, bool countable_only, uint32_t n_keys, unsigned long long* cursor,
                             int64_t* out_rows, unsigned long long* out_n, hipStream_t s) {
  (void)hipMemsetAsync(cursor, 0, (size_t)n_keys * 8, s);
  if (n <= 0) {
    (void)hipMemsetAsync(out_n, 0, 8, s);
    return;
  }
  const dim3 g(grid_for(n, kNsRowsPerBlock, 2048)), b(1024);
  const size_t lds_hist = n_keys <= kNsLdsKeys ? (size_t)n_keys * 4 : 0;
  const size_t lds_scat = n_keys <= kNsLdsKeysScatter ? (size_t)((n_keys + 1u) & ~1u) * 4 + (size_t)n_keys * 8 : 0;
  hipLaunchKernelGGL(kt_ns_histogram, g, b, lds_hist, s, pods.meta, n, countable_only ? 1 : 0, n_keys, cursor);
  hipLaunchKernelGGL(kt_ns_scan, dim3(1), b, 0, s, cursor, n_keys, out_n);
  hipLaunchKernelGGL(kt_ns_scatter, g, b, lds_scat, s, pods.meta, n, countable_only ? 1 : 0, n_keys, cursor, out_rows);
}

// plan_wg_ranges — which records of a namespace-ordered list every workgroup of a scan owns: contiguous ranges of about
// n / G records whose ends are moved to a namespace boundary when one lies close enough.  A workgroup walks the index chunks
// that hold words of ITS namespaces; with ranges cut at fixed multiples of tiles nearly every workgroup straddled two
// namespaces and opened the chunks of both (configs[4]: 14.6 chunk passes per workgroup where one namespace needs 8).
// ns_end[k] = end of namespace k's records in the list (what kt_ns_scatter leaves in its cursor words); cap = the most
// records one workgroup may own (the packed fold's fields are proven for it).  range[g] .. range[g + 1]; range[G + 1] = the
// largest range handed out.  On the HOST (round 6): the walk is one thread's chain of dependent steps — G iterations of 64-bit
// divisions and a binary search — which ONE GPU thread took 388 us for (profiles/r05_r05g_cfg4_kernel_stats.csv: the only kernel
// of a view build that is not a stream), while the engine synchronises for the row count right there anyway; the host does it in
// microseconds from a copy of the n_keys cursor words.
void plan_wg_ranges(const unsigned long long* ns_end, uint32_t n_keys, int64_t n_, int G_, uint32_t* range) {
  const uint64_t n = (uint64_t)(n_ > 0 ? n_ : 0);
  const uint32_t G = (uint32_t)(G_ > 0 ? G_ : 0), cap = wg_range_cap(n_, G_);
  uint64_t pos = 0, largest = 0;
  range[0] = 0u;
  for (uint32_t g = 0; g < G; ++g) {
    const uint64_t left = n - pos, wgs = G - g;
    uint64_t end = pos;
    if (left > 0) {
      const uint64_t share = (left + wgs - 1) / wgs;
      // what the workgroups behind this one can still take bounds how little this one may take
      const uint64_t must = left > (wgs - 1) * (uint64_t)cap ? left - (wgs - 1) * (uint64_t)cap : 1;
      const uint64_t lo = pos + (must > share / 2 ? must : (share / 2 ? share / 2 : 1)), hi = pos + (left < cap ? left : cap);
      const uint64_t ideal = pos + share > hi ? hi : pos + share;
      end = ideal < lo ? lo : ideal;
      if (wgs > 1) {
        // the namespace boundaries around `ideal`: first k with ns_end[k] >= ideal
        uint32_t a = 0, b = n_keys;
        while (a < b) {
          const uint32_t m = (a + b) / 2;
          if (ns_end[m] < ideal) a = m + 1; else b = m;
        }
        uint64_t best = 0, best_d = ~0ull;
        for (uint32_t k = (a > 2 ? a - 2 : 0); k < n_keys && k <= a + 1; ++k) {
          const uint64_t e = ns_end[k];
          if (e < lo || e > hi) continue;
          const uint64_t d = e > ideal ? e - ideal : ideal - e;
          if (d < best_d) best_d = d, best = e;
        }
        if (best_d != ~0ull) end = best;
      } else {
        end = n;
      }
    }
    largest = end - pos > largest ? end - pos : largest;
    pos = end;
    range[g + 1] = (uint32_t)pos;
  }
  range[G + 1] = (uint32_t)largest;
}
uint32_t wg_range_cap(int64_t n, int G) {
  const int64_t share = (n + G - 1) / (G > 0 ? G : 1);
  return (uint32_t)(share + share / 4 + 64);
}

// kt_build_scan_view — the records the namespace-ordered scans stream, copied into scan order once per ordering so that
// a tile reads 64 consecutive records instead of gathering them through the row list on every chunk visit:
// meta word, atom row, and (aggregate view) the request row of every listed pod.
// One pod's record of a scan view (meta word, atom row, request row and / or packed request words) at position j
__device__ __forceinline__ void write_view_record(const PodTable& pods, int64_t p, int64_t j, uint64_t meta, uint64_t* v_meta, uint16_t* v_latom,
                                                  int64_t* v_req, const PackPlan& pk, uint64_t* v_pk) {
  const int LA = pods.LA, DS = pods.DS;
  v_meta[j] = meta;
  const u128* src = (const u128*)(pods.latom + p * LA);
  u128* dst = (u128*)(v_latom + j * LA);
  for (int q = 0; q < LA / 8; ++q) dst[q] = src[q];
  if (v_req) {
    const u128* rs = (const u128*)(pods.req + p * DS);
    u128* rd = (u128*)(v_req + j * DS);
    for (int q = 0; q < DS / 2; ++q) rd[q] = rs[q];
  }
  if (v_pk) {
    // ResourceAmountOfPod as packed words (PackPlan, kt_index.h): pod count 1 from bit 0 of word 0, every non-zero
    // request as its field; the plan proved that no value of this engine needs more bits than its field has
    uint64_t w[kPackMaxWords] = {1ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    for (int d = 0; d < pods.D; ++d) {
      const uint64_t f = (uint64_t)pods.req[p * DS + d] >> pk.shift[d];
      const uint64_t piece = pk.width[d] ? f << pk.pos[d] : 0ull;
      const uint32_t k = pk.word[d];
#pragma unroll
      for (uint32_t q = 0; q < kPackMaxWords; ++q) w[q] |= k == q ? piece : 0ull;  // (select chain: no indexed access to registers)
    }
    uint64_t* o = v_pk + j * pk.stride;
    o[0] = w[0], o[1] = w[1];
    if (pk.stride > 2u) o[2] = w[2], o[3] = w[3];
    if (pk.stride > 4u) o[4] = w[4], o[5] = w[5], o[6] = w[6], o[7] = w[7];
  }
}

__global__ __launch_bounds__(256) void kt_build_scan_view(PodTable pods, int64_t n, const int64_t* rows, uint64_t* v_meta,
                                                         uint16_t* v_latom, int64_t* v_req, const PackPlan pk, uint64_t* v_pk, int32_t* pos) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
    const int64_t p = rows[j];
    if (pos) pos[p] = (int32_t)j;  // where a later pod event finds the record (kt_patch_scan_views)
    write_view_record(pods, p, j, pods.meta[p], v_meta, v_latom, v_req, pk, v_pk);
  }
}

// kt_patch_scan_views — a pod event batch (upserts: the rows hold their NEW content; deletes: their meta word is 0)
// applied to the scan views in place instead of voiding them (every reconcile of the reference follows an event,
// throttle_controller.go:400-536; rebuilding lists and views costs a full pass over the pod tables per event):
//   * a pod that has a record keeps its position and gets its record rewritten — a pod that stopped being countable
//     stays as a record whose meta word says so (the scan's own `countable` test skips it);
//   * a newly countable pod is appended (row-ordered list) — or, in a namespace-ordered list, raises `dirty`: the
//     next scan then rebuilds (so does a pod that moved to another namespace).
__device__ __forceinline__ void patch_one(const PodTable& pods, int64_t p, const ViewPatch& v) {
  const uint64_t meta = pods.meta[p];
  const uint32_t st = (uint32_t)(meta >> kMetaStateShift) & 0xFu;
  const bool countable = (st & (kPodValid | kPodSchedMatch | kPodScheduled)) == (kPodValid | kPodSchedMatch | kPodScheduled);
  if (v.vc_meta) {
    int64_t pos = v.pos_c[p];
    if (pos < 0 && countable) {
      if (v.by_ns) {
        *v.dirty = 1u;  // has to be sorted in
      } else if (atomicCAS((int*)&v.pos_c[p], -1, -2) == -1) {
        // (the row is claimed first: a batch that names the same row twice must not append two records — ADVICE r3)
        pos = (int64_t)atomicAdd(v.n_c, 1ull);
        if (pos >= v.cap_c) {
          *v.dirty = 1u;
          v.pos_c[p] = -1;
          pos = -1;
        } else {
          v.vc_rows[pos] = p;
          v.pos_c[p] = (int32_t)pos;
        }
      } else {
        pos = -1;  // another entry of this batch appends the row (its record is written from the same table content)
      }
    }
    if (pos >= 0) {
      const uint64_t old = v.vc_meta[pos];
      if (v.by_ns && countable && ((old ^ meta) & kMetaNsMask) != 0) *v.dirty = 1u;  // moved to another namespace
      // a record that stops counting keeps its place in its namespace's run: the scans read the namespace range of a
      // workgroup off the first and last record of its tiles
      const uint64_t meta_w = countable ? meta : (meta & ~kMetaNsMask) | (old & kMetaNsMask);
      write_view_record(pods, p, pos, meta_w, v.vc_meta, v.vc_latom, v.vc_req, v.pk, v.vc_pk);
    }
  }
  if (v.va_meta) {
    const int64_t pos = p < v.rows_a ? (int64_t)v.pos_a[p] : -1;
    if (pos < 0) {
      *v.dirty = 1u;  // a row the list does not cover yet
    } else {
      const uint64_t old = v.va_meta[pos];
      if ((st & kPodValid) && ((old ^ meta) & kMetaNsMask) != 0) *v.dirty = 1u;
      const uint64_t meta_w = (st & kPodValid) ? meta : (meta & ~kMetaNsMask) | (old & kMetaNsMask);
      write_view_record(pods, p, pos, meta_w, v.va_meta, v.va_latom, nullptr, v.pk, nullptr);
    }
  }
}
__global__ __launch_bounds__(256) void kt_patch_scan_views(PodTable pods, int64_t n, const int64_t* rows, int64_t row0, const ViewPatch v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  patch_one(pods, rows ? rows[i] : row0 + i, v);
}

// The completion signal of the event kernels: everything the workgroup wrote is made visible device-wide (a release fence
// at system scope writes the XCD's L2 back), then the sequence number goes to the pinned word the host spins on —
// hipEventSynchronize on the event behind such a kernel took 35-40 us of the 47 us from a pod event to the PreFilter
// that sees it (`latency.upsert1_then_check1`); the event stays as the fallback and for the slot's reuse.
__device__ __forceinline__ void signal_host(unsigned long long* host_seq, unsigned long long seq) {
  if (!host_seq) return;
  __threadfence_system();
  __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// kt_feed_small — ONE launch for a pod informer event or a handful of them (n <= 256, one workgroup): what kt_ingest_pods,
// kt_translate_pods and kt_patch_scan_views do one after the other, per thread = pod (each phase only reads what the same
// thread wrote in the phase before), and the overflow counter as the batch left it goes straight to the pinned word the
// host reads after the event — three dependent launches and a copy less on the path from a pod event to the next PreFilter.
//   do_translate: the compiled program is current (else every row is translated when it is compiled)
//   has_patch   : there are scan views to patch
//   stage_src / stage_dst / stage_bytes: the batch lies in a pinned HOST slot; the workgroup first copies it to device
//                 scratch with all its threads (one trip over the link, 16 bytes per thread and pass) — `b` points into the
//                 copy — instead of every thread walking offsets -> containers -> requests over the link one dependent
//                 access after the other
template <int LA>
__global__ __launch_bounds__(kBlock) void kt_feed_small(PodTable pods, PodBatchDev b, const uint64_t* table, uint32_t mask, int key_atoms,
                                                        unsigned long long* n_overflow, int do_translate, int has_patch, const ViewPatch v,
                                                        unsigned long long* host_overflow, const u128* stage_src, u128* stage_dst, uint32_t stage_bytes,
                                                        unsigned long long* host_seq, unsigned long long seq) {
  __shared__ __attribute__((aligned(16))) uint16_t out[kBlock][LA];
  for (uint32_t o = threadIdx.x; o < (stage_bytes + 15u) / 16u; o += kBlock) stage_dst[o] = stage_src[o];
  __syncthreads();
  const int64_t i = threadIdx.x;
  if (i < b.n) {
    const int64_t row = b.rows ? b.rows[i] : b.row0 + i;
    ingest_one(pods, b, i);
    if (do_translate) translate_one<LA>(pods, row, table, mask, key_atoms, n_overflow, &out[threadIdx.x][0]);
    if (has_patch) patch_one(pods, row, v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (do_translate && host_overflow) *host_overflow = __hip_atomic_load(n_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    signal_host(host_seq, seq);
  }
}
// kt_feed_few — an informer event proper (n <= kFeedFewMax pods): ONE WAVE PER POD.  kt_feed_small runs a pod's ingest,
// translation and view patch in one THREAD — some forty dependent trips to memory for a single pod, 20 of the 28 us from
// the event to the kernel's completion signal.  Here the workgroup pulls the pinned slot into LDS (one trip over the
// link), then a pod's wave works lane-parallel: lane d sums dimension d over the containers (PodRequestResourceList),
// lane l copies and translates label l (one probe of the atom table per lane, all in flight together), the masks are
// ballots, the atom row is compacted by mbcnt; only the view patch stays one lane's work.
//   b: the batch as BYTE OFFSETS into the slot (the pointer fields hold offsets); has_rows: b.rows is a list
template <int LA>
__global__ __launch_bounds__(kBlock) void kt_feed_few(PodTable pods, PodBatchDev b, int has_rows, const uint64_t* table, uint32_t mask, int key_atoms,
                                                      unsigned long long* n_overflow, int do_translate, int has_patch, const ViewPatch v,
                                                      unsigned long long* host_overflow, const u128* slot, uint32_t slot_bytes,
                                                      unsigned long long* host_seq, unsigned long long seq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char few_lds[];  // the slot, then one atom row per wave
  const uint32_t n16 = (slot_bytes + 15u) / 16u;
  for (uint32_t o = threadIdx.x; o < n16; o += kBlock) ((u128*)few_lds)[o] = slot[o];
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if ((int64_t)wave < b.n) {
    auto at = [&](const void* off) -> const unsigned char* { return few_lds + (size_t)off; };
    const uint32_t i = wave;
    const int D = pods.D, DS = pods.DS, L = pods.L, LS = pods.LS;
    const int64_t row = has_rows ? ((const int64_t*)at(b.rows))[i] : b.row0 + i;
    const uint32_t* ctr_off = (const uint32_t*)at(b.ctr_off);
    const uint8_t* ctr_init = (const uint8_t*)at(b.ctr_init);
    const uint32_t* ctr_present = (const uint32_t*)at(b.ctr_present);
    const int64_t* ctr_req = (const int64_t*)at(b.ctr_req);
    // ---- PodRequestResourceList (resourcelist.go:27-46), lane = dimension: sum of the containers, max with every init
    //      container, plus the overhead — exactly ingest_one's arithmetic, one dimension per lane
    const int d = (int)lane;
    int64_t c = 0, ic = 0;
    bool cp = false, icp = false;
    const uint32_t k0 = ctr_off[i] - b.ctr_base, k1 = ctr_off[i + 1] - b.ctr_base;
    for (uint32_t k = k0; k < k1; ++k) {  // (wave-uniform)
      const uint32_t pm = ctr_present[k];
      const bool init = ctr_init[k] != 0;
      if (d < D && ((pm >> d) & 1u)) {
        const int64_t q = ctr_req[(int64_t)k * D + d];
        if (init) {
          ic = icp ? (ic >= q ? ic : q) : q;
          icp = true;
        } else {
          c += q;
          cp = true;
        }
      }
    }
    if (icp) c = cp ? (c >= ic ? c : ic) : ic;
    cp = cp || icp;
    const uint32_t op = ((const uint32_t*)at(b.ovh_present))[i];
    if ((op >> 31) && d < D && ((op >> d) & 1u)) {
      c += ((const int64_t*)at(b.ovh))[(int64_t)i * D + d];
      cp = true;
    }
    const uint32_t dmask = (1u << D) - 1u;
    const uint32_t cpm = (uint32_t)__ballot(d < D && cp) & dmask;
    const uint32_t nz = (uint32_t)__ballot(d < D && cp && c != 0) & dmask;
    if (d < DS) pods.req[row * DS + d] = (d < D && cp) ? c : 0;
    const uint32_t ns = ((const uint32_t*)at(b.ns))[i], fl = ((const uint32_t*)at(b.flags))[i];
    if (lane == 0) {
      pods.ns[row] = ns;
      pods.flags[row] = (fl & 0xFu) | (cpm << kPresentShift);
    }
    uint64_t meta = (uint64_t)(ns & (uint32_t)kMetaNsMask) | (uint64_t)(fl & 0xFu) << kMetaStateShift | (uint64_t)cpm << kMetaPresentShift |
                    (uint64_t)nz << kMetaNzShift;
    // ---- labels, lane = label slot
    const uint32_t* label_off = (const uint32_t*)at(b.label_off);
    const uint32_t l0 = label_off[i] - b.label_base, l1 = label_off[i + 1] - b.label_base;
    const bool have = (int)lane < L && l0 + lane < l1;
    const uint32_t pair = have ? ((const uint32_t*)at(b.label_pair))[l0 + lane] : 0u;
    const uint32_t key = have ? ((const uint32_t*)at(b.label_key))[l0 + lane] : 0u;
    if ((int)lane < LS) {
      pods.lpair[row * LS + lane] = pair;
      pods.lkey[row * LS + lane] = key;
    }
    if (do_translate) {
      // one atom per label: the pair when some selector names it, else the key atom when some selector names the key;
      // the atoms keep the labels' order (rank among the lanes that found one)
      uint32_t id = 0;
      if ((int)lane < LS && pair != 0u) {
        id = atom_id_of(table, mask, pair);
        if (!id && key_atoms) id = atom_id_of(table, mask, kKeyAtom | key);
      }
      const uint64_t mk = __ballot(id != 0u);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      const uint32_t cnt = (uint32_t)__popcll(mk);
      uint16_t* out = (uint16_t*)(few_lds + ((size_t)n16 * 16u) + (size_t)wave * LA * 2u);
      if (lane < (uint32_t)LA) out[lane] = 0;
      if (id != 0u && rank < (uint32_t)LA) out[rank] = (uint16_t)id;
      if (lane < (uint32_t)(LA / 8)) ((kt_u32x4*)(pods.latom + row * LA))[lane] = ((const kt_u32x4*)out)[lane];
      if (cnt > (uint32_t)LA) {
        if (lane == 0 && ((meta >> kMetaStateShift) & kPodValid)) atomicAdd(n_overflow, 1ull);
        meta |= kMetaOverflow;
      }
    }
    if (lane == 0) pods.meta[row] = meta;
    if (has_patch) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the wave's stores before the patch reads the rows back
      if (lane == 0) patch_one(pods, row, v);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (do_translate && host_overflow) *host_overflow = __hip_atomic_load(n_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    signal_host(host_seq, seq);
  }
}
// the same for deletes: kt_delete_pods + kt_patch_scan_views
__global__ __launch_bounds__(kBlock) void kt_unfeed_small(PodTable pods, int64_t n, const int64_t* rows, int has_patch, const ViewPatch v,
                                                          unsigned long long* host_seq, unsigned long long seq) {
  const int64_t i = threadIdx.x;
  if (i < n) {
    const int64_t p = rows[i];
    pods.flags[p] = 0;
    pods.meta[p] = 0;
    if (has_patch) patch_one(pods, p, v);
  }
  __syncthreads();
  if (threadIdx.x == 0) signal_host(host_seq, seq);
}
void launch_feed_small(const PodTable& pods, const PodBatchDev& b, const IndexDev& ix, unsigned long long* n_overflow, bool do_translate,
                       const ViewPatch* v, unsigned long long* host_overflow, const void* stage_src, void* stage_dst, uint32_t stage_bytes,
                       unsigned long long* host_seq, unsigned long long seq, hipStream_t s) {
  const ViewPatch vp = v ? *v : ViewPatch{};
  const int tr = do_translate ? 1 : 0, hp = v ? 1 : 0;
  const u128* ss = (const u128*)stage_src;
  u128* sd = (u128*)stage_dst;
  if (pods.LA == 8) hipLaunchKernelGGL(kt_feed_small<8>, dim3(1), dim3(kBlock), 0, s, pods, b, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow, tr, hp, vp, host_overflow, ss, sd, stage_bytes, host_seq, seq);
  else if (pods.LA == 16) hipLaunchKernelGGL(kt_feed_small<16>, dim3(1), dim3(kBlock), 0, s, pods, b, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow, tr, hp, vp, host_overflow, ss, sd, stage_bytes, host_seq, seq);
  else hipLaunchKernelGGL(kt_feed_small<32>, dim3(1), dim3(kBlock), 0, s, pods, b, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow, tr, hp, vp, host_overflow, ss, sd, stage_bytes, host_seq, seq);
}
void launch_feed_few(const PodTable& pods, const PodBatchDev& b_off, bool has_rows, const IndexDev& ix, unsigned long long* n_overflow, bool do_translate,
                     const ViewPatch* v, unsigned long long* host_overflow, const void* slot, uint32_t slot_bytes, unsigned long long* host_seq,
                     unsigned long long seq, hipStream_t s) {
  const ViewPatch vp = v ? *v : ViewPatch{};
  const int tr = do_translate ? 1 : 0, hp = v ? 1 : 0, hr = has_rows ? 1 : 0;
  const u128* sl = (const u128*)slot;
  const size_t lds = (((size_t)slot_bytes + 15u) & ~(size_t)15u) + (size_t)kFeedFewMax * (size_t)pods.LA * 2u;
  if (pods.LA == 8) hipLaunchKernelGGL(kt_feed_few<8>, dim3(1), dim3(kBlock), lds, s, pods, b_off, hr, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow, tr, hp, vp, host_overflow, sl, slot_bytes, host_seq, seq);
  else if (pods.LA == 16) hipLaunchKernelGGL(kt_feed_few<16>, dim3(1), dim3(kBlock), lds, s, pods, b_off, hr, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow, tr, hp, vp, host_overflow, sl, slot_bytes, host_seq, seq);
  else hipLaunchKernelGGL(kt_feed_few<32>, dim3(1), dim3(kBlock), lds, s, pods, b_off, hr, ix.atom_table, ix.atom_mask, (int)ix.has_key_atoms, n_overflow, tr, hp, vp, host_overflow, sl, slot_bytes, host_seq, seq);
}
void launch_unfeed_small(const PodTable& pods, int64_t n, const int64_t* rows, const ViewPatch* v, unsigned long long* host_seq,
                         unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(kt_unfeed_small, dim3(1), dim3(kBlock), 0, s, pods, n, rows, v ? 1 : 0, v ? *v : ViewPatch{}, host_seq, seq);
}

void launch_patch_scan_views(const PodTable& pods, int64_t n, const int64_t* rows, int64_t row0, const ViewPatch& v, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(kt_patch_scan_views, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pods, n, rows, row0, v);
}

void launch_build_scan_view(const PodTable& pods, int64_t n, const int64_t* rows, uint64_t* v_meta, uint16_t* v_latom,
                            int64_t* v_req, hipStream_t s, const PackPlan* pk, uint64_t* v_pk, int32_t* pos) {
  if (n <= 0) return;
  hipLaunchKernelGGL(kt_build_scan_view, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, pods, n, rows, v_meta, v_latom, v_req,
                     pk ? *pk : PackPlan(), pk ? v_pk : nullptr, pos);
}

// kt_sum_abs_requests — sum over the valid pod rows of |effective request| per dimension, exactly (two 32-bit limb sums
// per dimension: out[2d] = sum of the low halves, out[2d+1] = sum of the high halves; good for 2^32 pods).  Every sum
// the scans or the exchange between GPUs can form is a sum over a subset of these pods, so it is bounded by this total:
// the engine refuses a reconcile only when THIS exceeds the exact range — where the reference would promote to big
// decimals — instead of pricing every pod at the largest request ever seen.
__global__ __launch_bounds__(256) void kt_sum_abs_requests(PodTable pods, int64_t n, unsigned long long* out) {
  // lane = pod: its request row as 16-byte pieces (the lanes of a wave cover 64 consecutive rows: every fetched line is
  // used in full); limb sums per dimension in registers, met per block in LDS, one atomic per block and word
  const int D = pods.D, ppr = pods.DS / 2;  // pieces per row (<= 8)
  unsigned long long lo[16], hi[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) lo[d] = 0ull, hi[d] = 0ull;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
    if (!((pods.meta[p] >> kMetaStateShift) & kPodValid)) continue;
    const kt_i64x2* row = (const kt_i64x2*)(pods.req + p * pods.DS);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < ppr) {
        const kt_i64x2 v = row[q];
        const unsigned long long a = v.x < 0 ? 0ull - (unsigned long long)v.x : (unsigned long long)v.x;
        const unsigned long long b = v.y < 0 ? 0ull - (unsigned long long)v.y : (unsigned long long)v.y;
        lo[2 * q] += a & 0xFFFFFFFFull, hi[2 * q] += a >> 32, lo[2 * q + 1] += b & 0xFFFFFFFFull, hi[2 * q + 1] += b >> 32;
      }
  }
  __shared__ unsigned long long acc[32];
  if (threadIdx.x < 32) acc[threadIdx.x] = 0ull;
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 16; ++d)
    if (d < D) {
      if (lo[d]) atomicAdd(&acc[2 * d], lo[d]);
      if (hi[d]) atomicAdd(&acc[2 * d + 1], hi[d]);
    }
  __syncthreads();
  if (threadIdx.x < 2u * (unsigned)D && acc[threadIdx.x]) atomicAdd(out + threadIdx.x, acc[threadIdx.x]);
}
void launch_sum_abs_requests(const PodTable& pods, int64_t n, unsigned long long* out, hipStream_t s) {
  (void)hipMemsetAsync(out, 0, 32 * 8, s);
  if (n <= 0) return;
  hipLaunchKernelGGL(kt_sum_abs_requests, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, pods, n, out);
}

void launch_ingest_pods(const PodTable& pods, const PodBatchDev& b, hipStream_t s) {
  if (b.n <= 0) return;
  hipLaunchKernelGGL(kt_ingest_pods, dim3(grid_for(b.n)), dim3(kBlock), 0, s, pods, b);
}
void launch_delete_pods(const PodTable& pods, int64_t n, const int64_t* rows_dev, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(kt_delete_pods, dim3(grid_for(n)), dim3(kBlock), 0, s, pods, n, rows_dev);
}
void launch_gather_pod_requests(const PodTable& pods, int64_t n, const int64_t* rows_dev, int64_t* out_v,
                                uint32_t* out_present, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(kt_gather_pod_requests, dim3(grid_for(n)), dim3(kBlock), 0, s, pods, n, rows_dev, out_v,
                     out_present);
}

}  // namespace kt
