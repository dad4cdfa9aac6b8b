// kt_device.h — device-visible table descriptors shared by the HIP kernels and the host engine.
// gfx950 only (wave64); no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace kt {

constexpr int kWave = 64;
constexpr int64_t kInf = INT64_MAX;

// KT_DEBUG_POISON=1: every device allocation of the engine is filled with 0xA5 before anything is written to it, so
// that a kernel that reads memory no launch wrote (round 3: the slab of an aggregate workgroup without tiles) gives
// wrong results on EVERY box instead of on the ones whose allocator hands back dirty pages.  The -m gpu suite is run
// under it once per round (tools/gpu_poison_suite.sh).
inline hipError_t kt_alloc_device(void** p, size_t bytes) {
  static const bool poison = [] { const char* v = getenv("KT_DEBUG_POISON"); return v && *v && *v != '0'; }();
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && poison && bytes) {
    e = hipMemset(*p, 0xA5, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  return e;
}

// pod_flags word in HBM: bits 0-3 = KT_POD_* state, bits 16-31 = request-key presence mask.
constexpr uint32_t kPodValid = 0x1u, kPodSchedMatch = 0x2u, kPodScheduled = 0x4u, kPodFinished = 0x8u;
constexpr int kPresentShift = 16;
// pod meta word (PodTable::meta), the one 8-byte record the indexed scans read per pod besides its atom row:
//   bits 0-26 namespace row | 27 kMetaOverflow | 28-31 KT_POD_* state | 32-47 request-key presence | 48-63 non-zero mask
constexpr uint64_t kMetaNsMask = 0x07FFFFFFull;
constexpr uint64_t kMetaOverflow = 1ull << 27;  // more relevant atoms than the atom row holds: decided by the dense path
constexpr int kMetaStateShift = 28, kMetaPresentShift = 32, kMetaNzShift = 48;

constexpr uint32_t kThrValid = 0x1u, kThrCluster = 0x2u, kThrResponsible = 0x4u, kThrCalcAtNonzero = 0x8u,
                   kThrThrottledPod = 0x10u;
constexpr uint8_t kTermPodSelInvalid = 0x1u, kTermNsSelInvalid = 0x2u;
constexpr uint8_t kOpIn = 0, kOpNotIn = 1, kOpExists = 2, kOpDoesNotExist = 3;
constexpr uint8_t kOvrParseError = 0x1u, kOvrBeginParsed = 0x2u;
constexpr int64_t kZeroTimeS = -62135596800LL;

// Pod state: one 16-byte-aligned ROW per pod in each table.  lane = pod reads its label row with LS/4 and
// its request row with DS/2 128-bit loads (a wave covers one contiguous 64 x row span: fully coalesced,
// no per-element address math, no per-element predicates); the request row is also what the
// (pod, throttle)-match lanes gather (D lanes read one pod's row as a single 64-byte transaction at D = 8).
// Unused label slots hold 0 ("no label"), padding dimensions hold 0 ("not requested").
struct PodTable {
  uint32_t* ns;     // [cap]
  uint32_t* flags;  // [cap]
  int64_t* req;     // [cap][DS]  effective request (ResourceAmountOfPod), 0 where absent
  uint32_t* lpair;  // [cap][LS]  (key,value) pair ids, 0 = empty slot
  uint32_t* lkey;   // [cap][LS]  key ids
  uint64_t* meta;   // [cap]      ns | state | presence | non-zero mask (kMeta*): what the indexed scans stream
  uint16_t* latom;  // [cap][LA]  ids of the pod's REFERENCED label atoms (kt_index.h), 0 = empty slot; rewritten by
                    //            kt_translate_pods whenever the selector program changes
  int64_t cap;
  int32_t D, L;
  int32_t DS, LS;   // row strides: DS = D rounded up to even, LS = 4 / 8 / 16 / 32 / 64 >= L
  int32_t LA;       // atom slots per pod (8 / 16 / 32), chosen per selector program
};
inline int req_stride(int D) { return (D + 1) & ~1; }
inline int label_stride(int L) { return L <= 4 ? 4 : L <= 8 ? 8 : L <= 16 ? 16 : L <= 32 ? 32 : 64; }

// ResourceAmount rows, row-major [n][D] like kt_amounts.
struct AmountTab {
  int64_t* v;
  uint32_t* present;
  int64_t* count;
  uint8_t* has_count;
};

// Compiled selector program of all throttles (rebuilt by the host whenever throttles/namespaces change).
struct SelProgram {
  const uint32_t* thr_term_off;  // [T+1] terms of throttle t: [off[t], off[t+1])
  const uint32_t* term_thr;      // [G]   owning throttle row
  const uint8_t* term_flags;     // [G]   kTerm*
  const uint32_t* term_req_off;  // [G+1] podSelector requirements
  const uint8_t* req_op;         // [R]
  const uint32_t* req_key;       // [R]
  const uint32_t* req_val_off;   // [R+1]
  const uint32_t* req_val;       // pair ids
  // bit g of row ns: term g can apply to pods of namespace ns, i.e. the owning throttle is valid and
  // responsible AND (Throttle: ns == thr.ns | ClusterThrottle: ns object exists and the term's
  // namespaceSelector converts and matches it).
  const uint32_t* ns_term_ok;    // [n_ns][gw]
  const uint8_t* ns_valid;       // [n_ns] Namespace object exists
  uint32_t gw;                   // words per ns row
  int32_t T;                     // throttle rows in use (1 + highest row)
  int32_t G;                     // terms
  int32_t n_ns;
};

// Throttle tables (row-major, T rows).
struct ThrTables {
  uint32_t* flags;  // kThr*
  AmountTab spec, calc, used, reserved;
  // status.used beyond int64 (resource.Quantity never overflows, resourcelist.go:48-54): the HIGH 64 bits of every value,
  // used.v holding the low 64 (two's complement); nullptr = every value is in range (sign extension of used.v)
  int64_t* used_hi;
  uint32_t* thrl_flag;
  uint32_t* thrl_has;
  uint64_t* status_msgs_fp;
  const uint64_t* spec_msgs_fp;
  const uint32_t* ovr_off;  // [T+1]
  const int64_t* ovr_begin_s;
  const int32_t* ovr_begin_ns;
  const int64_t* ovr_end_s;
  const int32_t* ovr_end_ns;
  const uint8_t* ovr_flags;
  AmountTab ovr_thr;
};

// Result of one reconcile pass (device), T rows.
struct ReconcileOut {
  AmountTab used, calc;
  int64_t* used_hi;  // nullable: high 64 bits of used.v (wide sums)
  uint8_t* calc_updated;
  uint32_t* thrl_flag;
  uint32_t* thrl_has;
  uint8_t* thrl_pod;
  uint8_t* error;
  int64_t* next_s;   // NextOverrideHappensIn as an instant: seconds (INT64_MAX = none) ...
  int32_t* next_ns;  // ... and nanoseconds
};

// Per-throttle record the check kernels consume (built by kt_prepare_check):
//   thr[d]   effective threshold, +inf where the threshold has no such key
//   head[d]  threshold - used - reserved  (minus 1 when isThrottledOnEqual), +inf likewise
// so that CheckThrottledFor's steps 1 and 4 become  nz(pod,d) && pod[d] > thr[d] / head[d]
// and steps 2+3 collapse into one bitmask (see DESIGN.md "Check algebra").
constexpr uint32_t kRecExceedsByCount = 0x1u, kRecActiveByCount = 0x2u, kRecInsufficientByCount = 0x4u;
// kRecTight: SOME pod of this engine could exceed thr[] / head[] in some dimension (judged against the per-dimension
// upper bound of all effective requests, ReqBound).  Without it the verdict of a matched pod follows from the count
// flags and (non-zero mask & active_mask) alone: neither its request row nor thr[] / head[] has to be read.
constexpr uint32_t kRecTight = 0x8u;
struct ReqBound {
  int64_t v[16];  // >= every pod's effective request per dimension (host-maintained; INT64_MAX = unknown)
};
// thr[] and head[] of a throttle share one 128-byte line at DT = 8 (the (match, dimension) lanes of the check
// kernels gather them as 16-byte pieces); {flags, active_mask} is ALSO kept as a compact array behind the
// records (rec_flags()), small enough to be staged in LDS.
template <int DT>
struct alignas(128) CheckRec {
  int64_t thr[DT];
  int64_t head[DT];
  uint32_t flags;
  uint32_t active_mask;
};
struct RecFlags {
  uint32_t flags, active_mask;
};
template <int DT>
__host__ __device__ inline RecFlags* rec_flags(void* recs, int T) { return (RecFlags*)((CheckRec<DT>*)recs + T); }
inline size_t recs_bytes(int T) { return (size_t)(T + 1) * (sizeof(CheckRec<16>) + sizeof(RecFlags)); }

// Wide sums: when the requests of the pods held add up beyond int64, a reconcile scans twice — once adding the low 32-bit
// limb of every request (limb 1), once the rest (limb 2: request >> 32, arithmetic) — and kt_finalize puts the two sums
// together in 128 bits.  limb 0 = the request itself.
__host__ __device__ inline int64_t limb_of(int64_t v, int limb) {
  return limb == 1 ? (int64_t)((uint64_t)v & 0xFFFFFFFFull) : limb == 2 ? (v >> 32) : v;
}

// layout of one throttle's row in the partial-used buffer (int64 words): v[D], present_count[D], pods, errors
__host__ __device__ inline int partial_stride(int D) { return 2 * D + 2; }
// ... and where its parts start (every kernel that writes or reads a partial row, and kt_partial_layout — the C-ABI query a
// host that runs its own collective lays its buffers out by — go through these)
__host__ __device__ inline int partial_off_presence(int D) { return D; }     // per-key contributor counts, after the D values
__host__ __device__ inline int partial_off_pods(int D) { return 2 * D; }     // pods counted
__host__ __device__ inline int partial_off_errors(int D) { return 2 * D + 1; }  // pods whose selector evaluation failed

}  // namespace kt
