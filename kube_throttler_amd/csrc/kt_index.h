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

__host__ __device__ inline uint32_t index_hash(uint64_t key, uint32_t mask) {
  uint64_t h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return (uint32_t)h & mask;
}

struct ThrInfo {
  bool live;
  bool cluster;
  uint32_t ns;
};

struct HostIndex {
  std::vector<IndexSlot> slots;
  uint32_t mask = 0;
  std::vector<uint32_t> postings;
  std::vector<uint32_t> uni_ns_off;  // [n_ns + 1]
  std::vector<uint32_t> uni_ns;
  std::vector<uint32_t> uni_cluster;
  std::vector<uint32_t> slow_thr;
  bool has_key_atoms = false;
};

struct IndexDev {
  IndexSlot* slots = nullptr;
  uint32_t* postings = nullptr;
  uint32_t* uni_ns_off = nullptr;
  uint32_t* uni_ns = nullptr;
  uint32_t* uni_cluster = nullptr;
  uint32_t* slow_thr = nullptr;
  uint32_t mask = 0, n_uni_cluster = 0, n_slow = 0;
  uint32_t has_key_atoms = 0;
  size_t cap_slots = 0, cap_postings = 0, cap_uni_ns_off = 0, cap_uni_ns = 0, cap_uni_cluster = 0, cap_slow = 0;
};

void build_index(HostIndex& out, const std::vector<uint32_t>& thr_term_off, const std::vector<uint32_t>& term_thr,
                 const std::vector<uint8_t>& term_flags, const std::vector<uint32_t>& term_req_off,
                 const std::vector<uint8_t>& req_op, const std::vector<uint32_t>& req_key,
                 const std::vector<uint32_t>& req_val_off, const std::vector<uint32_t>& req_val,
                 const std::function<ThrInfo(uint32_t)>& thr_info, uint32_t n_ns);
hipError_t upload_index(const HostIndex& h, IndexDev& d, hipStream_t s);
void release_index(IndexDev& d);

struct PodTable;
struct SelProgram;
void launch_aggregate_indexed(const PodTable& pods, int64_t n_rows, const SelProgram& sp, const IndexDev& ix,
                              bool keys, unsigned long long* partial, hipStream_t s);
void launch_check_indexed(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp,
                          const IndexDev& ix, bool keys, const void* recs, uint64_t* summary, uint8_t* status,
                          hipStream_t s);

}  // namespace kt
