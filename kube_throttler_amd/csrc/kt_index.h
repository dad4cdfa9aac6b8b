// kt_index.h — inverted index from label atoms to selector terms, so that the pod x throttle scans do
// work proportional to (pods + candidate terms) instead of P x T.
//
// Every term of a live throttle (valid, responsible, no unconvertible podSelector) is filed under ONE
// anchor requirement:
//   In{key, values}  -> under every (key,value) pair id of the set            (atom = pair id)
//   Exists{key}      -> under the key                                         (atom = 0x80000000 | key id)
// A pod carries at most one value per key, so a matching term is reached through exactly one of the
// pod's labels.  Terms with no positive requirement (empty selector, only NotIn/DoesNotExist) are filed under
// row 0 ("every pod"); throttles that contain an unconvertible podSelector term go to a "slow" list and are
// walked term by term in order (error semantics of throttle_selector.go:30-42 depend on term order).
// The namespace side of every term (implicit namespace equality of a Throttle, throttle_controller.go:249;
// namespaceSelector of a ClusterThrottle) is pre-evaluated into SelProgram::ns_term_ok and enters as a bitmap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "kt_device.h"

namespace kt {

constexpr uint32_t kKeyAtom = 0x80000000u;

// TermRec flags
constexpr uint32_t kPostComplex = 0x1u;  // needs the generic requirement walk (term_match)
constexpr uint32_t kPostPair2 = 0x10u;   // pod must also carry `pair2`

// ---- bitmap form of the whole selector side ----------------------------------------------------------
// Every indexed term gets a number c; terms with the same namespace-admission set (a "class") are numbered
// contiguously — throttles ordered by the admission set of their first term, the terms of a throttle kept
// together — and a class of <= 64 terms never straddles a 64-bit word (larger classes start on a word boundary;
// unused numbers are padding), so a namespace only ever touches a few words of any bitmap:
//     candidates(pod)[w] = (rows[0][w] | OR_l rows[row_of(label_l)][w]) & nsrows[ns][w]   for w in nswords[ns]
// rows[0] = terms without a positive requirement, rows[1] = all zero (unknown atoms).  Atoms are found in
// 4-entry buckets (branch-free probe).  TermRec carries what a visit needs.
struct alignas(16) TermRec {
  uint32_t g, t, pair2, flags;  // flags: kPost* in the low byte, the chunk-local throttle rank above
};
// Bitmap-form extras of a term (flag kPostInline): up to two requirements besides the anchor, small enough to be
// decided from registers.  e[k] = {op (KT_OP_*; 0xFF = none), up to three atoms (pair ids for In / NotIn, the key id
// for Exists / DoesNotExist; kNoAtom = unused)}.
constexpr uint32_t kPostInline = 0x20u;  // TermX holds the remaining requirements
constexpr uint32_t kPostAdj = 0x40u;     // multi-term throttle: a match repeating the lane's previous throttle is dropped
constexpr uint32_t kNoAtom = 0xFFFFFFFFu;
struct alignas(16) TermX {
  uint32_t e[2][4];
};
struct alignas(16) AtomBucket {
  uint32_t atom[4];  // 0 = empty
  uint32_t row[4];
};
// bucket of an atom: multiplicative hash; the builder tries several odd multipliers per table size before doubling it
__host__ __device__ inline uint32_t atom_bucket(uint32_t atom, uint32_t mask, uint32_t mult) { return ((atom * mult) >> 9) & mask; }

struct ThrInfo {
  bool live;
  bool cluster;
  uint32_t ns;
};

// One LDS-sized slice of the bitmap form: 64-bit words [w0, w0 + n_words) of every row, with the TermRecs of those
// term numbers and the per-namespace lists of words that can hold candidates.  The terms of a throttle never
// straddle two chunks; `rank0 .. rank0 + n_thr` are the dense throttle ranks (term order) the chunk covers —
// TermRec::flags carries the chunk-local rank in bits 8+.
struct BmChunk {
  uint32_t w0, n_words;
  uint32_t rank0, n_thr;
  uint32_t img_off, img_bytes;  // image inside the blob (multiple of 16): rows first
  uint32_t off_nsrows, off_nsw_off, off_nsw, off_trec, off_trecx;  // relative to the image
  uint32_t stride;              // 64-bit words per row inside the image (odd: column reads spread over LDS banks)
  uint32_t slab_off;            // (aggregate) byte offset of this chunk's tables in the slab scratch / 16
  uint32_t pad[3];
};

struct HostIndex {
  std::vector<uint32_t> slow_thr;
  uint32_t bm_words = 0;  // W: 64-bit words per full bitmap row
  uint32_t bm_rows = 0;
  uint32_t bm_bucket_mask = 0, bm_bucket_mult = 0x9E3779B1u;
  std::vector<BmChunk> bm_chunks;
  std::vector<unsigned char> bm_images;  // chunk images back to back
  std::vector<uint32_t> bm_rank_t;       // dense throttle rank (term order) -> throttle row
  std::vector<AtomBucket> bm_buckets;
  uint32_t bm_max_img = 0, bm_max_thr = 0;
  uint64_t bm_slab_bytes = 0;  // aggregate scratch: one table per (chunk, workgroup)
  bool bm_has_key_rows = false;  // some term is anchored on an Exists requirement
  bool bm_has_inline = false;    // some term carries TermX
};

struct IndexDev {
  uint32_t* slow_thr = nullptr;
  uint32_t n_slow = 0;
  unsigned char* bm_blob = nullptr;  // chunk images
  BmChunk* bm_chunks = nullptr;
  uint32_t* bm_rank_t = nullptr;
  AtomBucket* bm_buckets = nullptr;
  std::vector<BmChunk> h_chunks;     // host copy (launch planning)
  uint32_t n_chunks = 0, bm_max_img = 0, bm_max_thr = 0, bm_bucket_bytes = 0;
  uint64_t bm_slab_bytes = 0;
  uint32_t bm_bucket_mask = 0, bm_bucket_mult = 0x9E3779B1u, bm_has_key_rows = 0, bm_has_inline = 0;
  size_t cap_bm_blob = 0, cap_bm_chunks = 0, cap_bm_rank_t = 0, cap_bm_buckets = 0, cap_slow = 0;
};

// agg_budget / chk_budget: LDS bytes left for (atom buckets + chunk image + table of the chunk's throttles, thr_bytes
// each) in kt_aggregate_bitmap and for (atom buckets + chunk image) in kt_check_bitmap
void build_index(HostIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint32_t>& term_thr,
                 const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
                 const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                 const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                 const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns,
                 const std::vector<uint32_t>& ns_term_ok, uint32_t gw, uint32_t agg_budget, uint32_t chk_budget, uint32_t thr_bytes);
hipError_t upload_index(const HostIndex& h, IndexDev& d, hipStream_t s);
void release_index(IndexDev& d);

struct PodTable;
struct SelProgram;
// LDS the two scan kernels need beside the atom buckets, the chunk image and (aggregate) the chunk's table
uint32_t aggregate_fixed_lds();
uint32_t check_fixed_lds();
// sp_dev: device-resident copy of sp.  Both return the symbol of the scan kernel they dispatched.
// after_scan (nullable) is invoked on the host right after the scan kernel is enqueued and before the slab
// reduction kernel (if any) — the engine uses it to bracket the two kernels with separate timing events.
// which pods an aggregate scan covers and how they enter the target buffer
struct AggScan {
  int64_t n = 0;                 // pods
  const int64_t* rows = nullptr; // device list of pod rows, or
  int64_t row0 = 0;              // ... the contiguous range row0 + [0, n)
  bool counts = false;           // exact per-key pod counts (incremental engines) instead of presence masks
  int sign = 1;                  // -1: remove the scanned pods' contribution (delta scans)
};
const char* launch_aggregate_indexed(const PodTable& pods, const AggScan& scan, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, bool keys, unsigned long long* partial, void* slab, hipStream_t s,
                              const std::function<void()>& after_scan = nullptr);
const char* launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const SelProgram* sp_dev, const IndexDev& ix, bool keys, const void* recs, uint64_t* summary,
                          uint8_t* status, hipStream_t s);

}  // namespace kt
