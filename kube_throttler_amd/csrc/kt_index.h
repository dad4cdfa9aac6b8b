// kt_index.h — inverted index from label atoms to selector terms, so that the pod x throttle scans do
// work proportional to (pods + candidate terms) instead of P x T.
//
// Every term of a live throttle (valid, responsible, no unconvertible podSelector) is filed under ONE
// anchor requirement:
//   In{key, values}  -> one posting per (key,value) pair id of the set        (atom = pair id)
//   Exists{key}      -> one posting under the key                             (atom = 0x80000000 | key id)
// scoped by namespace for namespaced Throttles (scope = ns + 1; Throttles(pod.Namespace).List is an
// implicit namespace-equality predicate, throttle_controller.go:249) and unscoped (scope = 0) for
// ClusterThrottles, whose namespaceSelector is pre-evaluated into SelProgram::ns_term_ok.
// A pod carries at most one value per key, so a matching term is reached through exactly one of the
// pod's labels.  Terms with no positive requirement (empty selector, only NotIn/DoesNotExist) go to
// per-namespace / cluster-wide "universal" lists; throttles that contain an unconvertible podSelector
// term go to a "slow" list and are walked term by term in order (error semantics of
// throttle_selector.go:30-42 depend on term order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "kt_device.h"

namespace kt {

struct alignas(16) IndexSlot {
  uint64_t key;  // (scope << 32) | atom ; 0 = empty
  uint32_t begin;
  uint32_t count;
};

constexpr uint32_t kKeyAtom = 0x80000000u;

// One posting = one candidate term, with everything the common case needs to decide the match inline:
// a term whose requirements are all single-value In (matchLabels with <= 2 pairs) is fully described by
// its anchor pair (implied by the hash key) and `pair2`; ClusterThrottle terms carry the 64-bit
// namespace admission mask when the engine holds <= 64 namespaces.
constexpr uint32_t kPostComplex = 0x1u;   // needs the generic requirement walk (term_match)
constexpr uint32_t kPostMulti = 0x2u;     // owning throttle has several terms: first-matching-term dedup
constexpr uint32_t kPostNsMask = 0x4u;    // cluster term, namespace test = bit `ns` of nsmask
constexpr uint32_t kPostNsBitmap = 0x8u;  // cluster term, namespace test = SelProgram::ns_term_ok
constexpr uint32_t kPostPair2 = 0x10u;    // pod must also carry `pair2`
struct alignas(16) Posting {
  uint32_t g;      // term
  uint32_t t;      // owning throttle row
  uint32_t pair2;
  uint32_t flags;
  uint64_t nsmask;
  uint64_t pad;
};

__host__ __device__ inline uint32_t index_hash(uint64_t key, uint32_t mask) {
  uint64_t h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return (uint32_t)h & mask;
}

// ---- bitmap form of the whole selector side (small-T regime) ----------------------------------------
// Every indexed term gets a number c; terms with the same namespace-admission set (a "class") are numbered
// contiguously and a class of <= 64 terms never straddles a 64-bit word (larger classes start on a word
// boundary; unused numbers are padding), so a namespace only ever touches a few 64-bit words of any bitmap:
//     candidates(pod)[w] = (rows[0][w] | OR_l rows[row_of(label_l)][w]) & nsrows[ns][w]   for w in nswords[ns]
// rows[0] = terms without a positive requirement, rows[1] = all zero (unknown atoms).  Atoms are found in
// 4-entry buckets (branch-free probe).  TermRec carries what a visit needs (same flags as Posting).
struct alignas(16) TermRec {
  uint32_t g, t, pair2, flags;
};
struct alignas(16) AtomBucket {
  uint32_t atom[4];  // 0 = empty
  uint32_t row[4];
};
__host__ __device__ inline uint32_t atom_bucket(uint32_t atom, uint32_t mask) { return ((atom * 0x9E3779B1u) >> 9) & mask; }

struct ThrInfo {
  bool live;
  bool cluster;
  uint32_t ns;
};

struct HostIndex {
  std::vector<IndexSlot> slots;
  uint32_t mask = 0;
  std::vector<Posting> postings;
  std::vector<uint32_t> uni_ns_off;  // [n_ns + 1]
  std::vector<uint32_t> uni_ns;
  std::vector<uint32_t> uni_cluster;
  std::vector<uint32_t> slow_thr;
  bool has_key_atoms = false;
  // bitmap form (valid when bm_words != 0)
  uint32_t bm_words = 0;   // W: 64-bit words per bitmap row
  uint32_t bm_stride = 0;  // row stride in 64-bit words (odd: column reads spread over LDS banks)
  uint32_t bm_rows = 0;
  uint32_t bm_bucket_mask = 0;
  std::vector<uint64_t> bm_row_bits;    // [bm_rows][bm_stride]
  std::vector<uint64_t> bm_nsrows;      // [n_ns][bm_stride]
  std::vector<uint32_t> bm_nswords_off; // [n_ns + 1]
  std::vector<uint32_t> bm_nswords;     // word indices a namespace can touch
  std::vector<AtomBucket> bm_buckets;
  std::vector<TermRec> bm_trec;
};

struct IndexDev {
  IndexSlot* slots = nullptr;
  Posting* postings = nullptr;
  uint32_t* uni_ns_off = nullptr;
  uint32_t* uni_ns = nullptr;
  uint32_t* uni_cluster = nullptr;
  uint32_t* slow_thr = nullptr;
  uint32_t mask = 0, n_uni_cluster = 0, n_slow = 0;
  uint32_t has_key_atoms = 0;
  uint32_t n_slots = 0, n_postings = 0;
  uint32_t n_cluster_postings = 0;  // postings filed under scope 0 come first in the array
  // bitmap form: ONE device blob holding the six tables back to back (16-byte aligned pieces, in the order
  // rows, nsrows, nswords_off, nswords, buckets, trec) — the kernels copy it to LDS with one streaming loop
  uint32_t bm_words = 0, bm_stride = 0, bm_bucket_mask = 0;
  unsigned char* bm_blob = nullptr;
  uint32_t bm_blob_bytes = 0;
  uint32_t bm_off[6] = {0, 0, 0, 0, 0, 0};  // byte offsets of the tables inside the blob
  size_t cap_bm_blob = 0;
  size_t cap_slots = 0, cap_postings = 0, cap_uni_ns_off = 0, cap_uni_ns = 0, cap_uni_cluster = 0, cap_slow = 0;
};

void build_index(HostIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint32_t>& term_thr,
                 const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
                 const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                 const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                 const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns,
                 const std::vector<uint32_t>& ns_term_ok, uint32_t gw);
hipError_t upload_index(const HostIndex& h, IndexDev& d, hipStream_t s);
void release_index(IndexDev& d);

struct PodTable;
struct SelProgram;
// slab: scratch for the per-block LDS tables of the aggregate kernel (nullptr => global atomics only)
size_t aggregate_slab_bytes(int T, int D);
// sp_dev: device-resident copy of sp.  Both return the symbol of the scan kernel they dispatched.
// after_scan (nullable) is invoked on the host right after the scan kernel is enqueued and before the slab
// reduction kernel (if any) — the engine uses it to bracket the two kernels with separate timing events.
const char* launch_aggregate_indexed(const PodTable& pods, int64_t n_rows, const SelProgram& sp, const SelProgram* sp_dev,
                              const IndexDev& ix, bool keys, unsigned long long* partial, void* slab, hipStream_t s,
                              const std::function<void()>& after_scan = nullptr);
const char* launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const SelProgram* sp_dev, const IndexDev& ix, bool keys, const void* recs, uint64_t* summary,
                          uint8_t* status, hipStream_t s);

}  // namespace kt
