// kt_launch.h — host-callable launchers of the HIP kernels (implemented in kt_kernels*.hip).
#pragma once
#include "kt_device.h"

namespace kt {

// Staged pod batch in device memory (row-major, as in kt_snapshot).
struct PodBatchDev {
  int64_t n;
  const int64_t* rows;  // nullable => row0 + i
  int64_t row0;
  const uint32_t* ns;
  const uint32_t* flags;
  const uint32_t* label_off;  // [n+1], relative to label_base
  const uint32_t* label_key;
  const uint32_t* label_pair;
  uint32_t label_base;        // value of label_off[0] in the host batch (arrays are copied from there)
  const uint32_t* ctr_off;    // [n+1]
  const uint8_t* ctr_init;
  const uint32_t* ctr_present;
  const int64_t* ctr_req;     // [n_ctr][D]
  uint32_t ctr_base;
  const uint32_t* ovh_present;
  const int64_t* ovh;         // [n][D]
};

struct IndexTables;  // kt_index.h

void launch_ingest_pods(const PodTable& pods, const PodBatchDev& b, hipStream_t s);
// dense list of the countable pod rows among [0, n) (out_n: device counter, zeroed by the caller)
void launch_compact_countable(const PodTable& pods, int64_t n, int64_t* out_rows, unsigned long long* out_n, hipStream_t s);
// pod rows of [0, n) ordered by namespace (all of them, or the countable ones); cursor: n_keys device words of scratch,
// n_keys = namespace capacity; out_n: device counter receiving the number of listed rows
void launch_order_rows_by_ns(const PodTable& pods, int64_t n, bool countable_only, uint32_t n_keys, unsigned long long* cursor,
                             int64_t* out_rows, unsigned long long* out_n, hipStream_t s);
// contiguous record ranges of a namespace-ordered list for the G workgroups of a scan, ends moved to namespace boundaries
// (host code, kt_kernels.hip: plan_wg_ranges): range[0 .. G], range[G + 1] = the largest range; no range holds more than
// wg_range_cap(n, G) records.  ns_end = a host copy of the cursor words launch_order_rows_by_ns leaves behind (end of every
// namespace's records); range: G + 2 host words
uint32_t wg_range_cap(int64_t n, int G);
void plan_wg_ranges(const unsigned long long* ns_end, uint32_t n_keys, int64_t n, int G, uint32_t* range);
// scan-ordered copies of meta / atom row (/ request row when v_req is given) of the listed rows
struct PackPlan;  // kt_index.h
// pk + v_pk (nullable): also the packed request words of every listed pod
// pos (nullable, one int32 per pod row, -1 = not listed): pos[row] = the record's position in the view
void launch_build_scan_view(const PodTable& pods, int64_t n, const int64_t* rows, uint64_t* v_meta, uint16_t* v_latom,
                            int64_t* v_req, hipStream_t s, const PackPlan* pk = nullptr, uint64_t* v_pk = nullptr, int32_t* pos = nullptr);
// a pod event batch applied to the scan views in place (kt_kernels.hip: kt_patch_scan_views)
struct ViewPatch;  // kt_index.h
void launch_patch_scan_views(const PodTable& pods, int64_t n, const int64_t* rows, int64_t row0, const ViewPatch& v, hipStream_t s);
// exact per-dimension sums of |request| over the valid rows of [0, n): out[2d] low-half sum, out[2d+1] high-half sum (32 words)
void launch_sum_abs_requests(const PodTable& pods, int64_t n, unsigned long long* out, hipStream_t s);
void launch_delete_pods(const PodTable& pods, int64_t n, const int64_t* rows_dev, hipStream_t s);
// small pod event batches (n <= kFeedSmallMax) as ONE launch: ingest + translate + view patch / delete + view patch
constexpr int64_t kFeedSmallMax = 256;
struct IndexDev;
void launch_feed_small(const PodTable& pods, const PodBatchDev& b, const IndexDev& ix, unsigned long long* n_overflow, bool do_translate,
                       const ViewPatch* v, unsigned long long* host_overflow, const void* stage_src, void* stage_dst, uint32_t stage_bytes,
                       unsigned long long* host_seq, unsigned long long seq, hipStream_t s);
// n <= kFeedFewMax pods, whose batch fits kFeedFewSlotMax bytes: one wave per pod (kt_feed_few); b_off holds BYTE OFFSETS into the slot
constexpr int64_t kFeedFewMax = 4;
constexpr uint32_t kFeedFewSlotMax = 32 * 1024;
void launch_feed_few(const PodTable& pods, const PodBatchDev& b_off, bool has_rows, const IndexDev& ix, unsigned long long* n_overflow, bool do_translate,
                     const ViewPatch* v, unsigned long long* host_overflow, const void* slot, uint32_t slot_bytes, unsigned long long* host_seq,
                     unsigned long long seq, hipStream_t s);
void launch_unfeed_small(const PodTable& pods, int64_t n, const int64_t* rows, const ViewPatch* v, unsigned long long* host_seq,
                         unsigned long long seq, hipStream_t s);
void launch_gather_pod_requests(const PodTable& pods, int64_t n, const int64_t* rows_dev, int64_t* out_v,
                                uint32_t* out_present, hipStream_t s);

void launch_aggregate_dense(const PodTable& pods, int64_t n_rows, const SelProgram& sp, bool keys,
                            unsigned long long* partial, hipStream_t s, int limb = 0);
// recs (nullable): also build the CheckRec<rec_DT> of every throttle for isThrottledOnEqual = rec_eq
// consume: the kernel leaves the partial rows zeroed behind
void launch_finalize(const ThrTables& tt, const SelProgram& sp, int D, unsigned long long* partial, bool consume,
                     int64_t now_s, int32_t now_ns, bool apply, const ReconcileOut& out, void* recs, int rec_DT, bool rec_eq,
                     const ReqBound& vmax, hipStream_t s, const uint8_t* row_mask = nullptr, unsigned long long* partial_hi = nullptr);
// partial_hi (nullable): wide sums — partial = sums of the requests' low 32-bit limbs, partial_hi = sums of their high parts
// row_mask (nullable, device, T bytes): rows with 0 are not reconciled — they keep and report their stored status
void launch_prepare_check(const ThrTables& tt, int T, int D, int DT, bool on_equal, void* recs, const ReqBound& vmax, hipStream_t s);
void launch_check_dense(const PodTable& pods, int64_t n, const int64_t* rows_dev, const SelProgram& sp, bool keys,
                        const void* recs, uint64_t* summary, uint8_t* status, hipStream_t s);

// sequential admission of a pod queue with reservation (kt_kernels_admit.hip); the mutable state (reserved amounts of
// all throttles) lives in LDS when it fits, else in `scratch` (admit_state_bytes)
size_t admit_state_bytes(int T, int D);
bool launch_admit(const PodTable& pods, int64_t n, const int64_t* rows_dev, const ThrTables& tt, int T, int D,
                  bool on_equal, bool commit, uint8_t* status, uint64_t* summary, void* scratch, bool force_global,
                  hipStream_t s);

inline int dt_bucket(int D) { return D <= 4 ? 4 : D <= 8 ? 8 : 16; }
inline int dt_bucket_ix(int D) { return D <= 8 ? 8 : 16; }  // indexed kernels: two instantiations
inline int lt_bucket(int L) { return L <= 8 ? 8 : 16; }

}  // namespace kt
