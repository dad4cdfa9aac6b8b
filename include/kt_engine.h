/*
 * kt_engine.h — C-ABI of the MI355X-native throttle-evaluation engine (libkt_engine.so).
 *
 * kube-throttler has no FFI of any kind today (pure Go, CGO_ENABLED=0 — Makefile:3,10), so this is a
 * NEW seam.  It is cut exactly where the Go plugin does its per-pod x per-throttle work, and each
 * entry point names the reference code whose body it replaces.  The outer plugin API stays untouched:
 *   PluginName / NewPlugin / PreFilter / Reserve / Unreserve  (pkg/scheduler_plugin/plugin.go:45,63,148,217,240)
 * The cgo stubs a maintainer would add are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int32_t: KT_OK (0) or a negative KT_ERR_*; text via kt_last_error().
 *   - the caller owns every pointer it passes; the engine copies during the call and retains nothing
 *     (cgo rule: C must not keep Go memory).  Output buffers are caller-allocated.
 *   - strings never cross: labels, namespaces, resource names are interned to ids by the caller
 *     (include/kt_snapshot.h documents the id spaces and the flat batch format).
 *   - rows are caller-managed dense indices: pod row in [0, pod_capacity), throttle row in
 *     [0, throttle_capacity), namespace id in [0, namespace_capacity).  Upserting a row replaces it.
 *   - `stream` arguments are a hipStream_t (NULL = the engine's own stream).  *_launch calls are
 *     asynchronous on that stream; results stay in HBM until a *_fetch call copies them out.
 *   - all entry points may be called from any OS thread; calls on one engine are serialised internally.
 */
#ifndef KT_ENGINE_H
#define KT_ENGINE_H

#include <stdint.h>
#include "kt_snapshot.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KT_OK 0
#define KT_ERR_INVALID_ARGUMENT (-1)
#define KT_ERR_OUT_OF_RANGE (-2)   /* row / id / dimension outside the configured capacity */
#define KT_ERR_DEVICE (-3)         /* HIP runtime error (message has the hipError string) */
#define KT_ERR_OVERFLOW_RISK (-4)  /* a value beyond 2^60 (upsert), or pod requests that ADD UP beyond 2^60 in one dimension
                                      (reconcile: where resource.Quantity would promote to big decimals): rescale it */
#define KT_ERR_NOT_READY (-5)      /* fetch without a preceding launch */
#define KT_ERR_NO_DEVICE (-6)      /* no gfx950 device visible: there is NO CPU fallback */
#define KT_ERR_UNSUPPORTED (-7)    /* the request exceeds what this entry point supports (message says what) */

/* per (pod, throttle) status — v1alpha1.CheckThrottleStatus (throttle_types.go:119-126) + not-affected / error */
#define KT_STATUS_NOT_AFFECTED 0
#define KT_STATUS_NOT_THROTTLED 1
#define KT_STATUS_ACTIVE 2
#define KT_STATUS_INSUFFICIENT 3
#define KT_STATUS_POD_REQUESTS_EXCEEDS_THRESHOLD 4
#define KT_STATUS_ERROR 255

/* per-pod summary word: what KubeThrottler.PreFilter derives (plugin.go:148-215).
 * bits 0-1  verdict: 0 framework.Success, 1 UnschedulableAndUnresolvable, 2 framework.Error
 * bits 4-23 #throttles[pod-requests-exceeds-threshold], 24-43 #throttles[active], 44-63 #throttles[insufficient] */
#define KT_VERDICT_SUCCESS 0
#define KT_VERDICT_UNSCHEDULABLE 1
#define KT_VERDICT_ERROR 2
#define KT_SUMMARY_VERDICT(w) ((uint32_t)((w) & 3u))
#define KT_SUMMARY_EXCEEDS(w) ((uint32_t)(((w) >> 4) & 0xFFFFFu))
#define KT_SUMMARY_ACTIVE(w) ((uint32_t)(((w) >> 24) & 0xFFFFFu))
#define KT_SUMMARY_INSUFFICIENT(w) ((uint32_t)(((w) >> 44) & 0xFFFFFu))

typedef struct kt_engine kt_engine; /* opaque */

typedef struct kt_config {
  int32_t n_dims;             /* D: resource dimensions (<= KT_MAX_DIMS) */
  int32_t max_labels;         /* labels kept per pod (<= KT_MAX_LABELS) */
  int64_t pod_capacity;       /* pod rows held in HBM */
  int32_t throttle_capacity;  /* Throttle + ClusterThrottle rows (< 2^20) */
  int32_t namespace_capacity;
  int32_t device;             /* HIP device ordinal; -1 = current device */
  int32_t kernel_variant;     /* low byte: 0 = default (indexed); 1 = dense P x T scan (reference shape, for cross-checks);
                                 | KT_VARIANT_INCREMENTAL: see below */
} kt_config;
/* Incremental event path, indexed kernels only: the engine keeps the per-throttle `used` partials of
 * its pod rows current across kt_upsert_pods / kt_delete_pods by delta scans over just the touched rows (old content
 * out, new content in — which also covers the label-change symmetric difference of throttle_controller.go:469-500),
 * so that a reconcile no longer rescans every pod: kt_aggregate_launch becomes a copy.  A change of throttles or
 * namespaces voids the partials; the next reconcile rescans once.  Results are identical to a full rescan. */
#define KT_VARIANT_INCREMENTAL 0x100

/* Library / device facts (for logs and bench JSON). */
const char* kt_version(void);

/* Replaces the controller construction in NewPlugin (plugin.go:63-146) for the evaluation state:
 * allocates the SoA tables in HBM.  Fails with KT_ERR_NO_DEVICE when no GPU is present. */
int32_t kt_engine_create(const kt_config* cfg, kt_engine** out);
int32_t kt_engine_destroy(kt_engine* e);
/* Last error text of this engine (or of kt_engine_create when e == NULL); valid until the next call. */
const char* kt_last_error(kt_engine* e);

/* ---- state feed: what the informer event handlers push (throttle_controller.go:400-536,
 *      clusterthrottle_controller.go:428-570).  A batch is a kt_snapshot whose sections may be empty
 *      (n_ns / n_pods / n_thr = 0).  rows == NULL means batch index i -> row i. ------------------------ */
int32_t kt_upsert_namespaces(kt_engine* e, const kt_snapshot* batch, const int32_t* ns_rows);
/* Pods: the effective request of each pod is computed ON DEVICE from the batch's containers —
 * resourcelist.PodRequestResourceList (pkg/resourcelist/resourcelist.go:27-46) + ResourceAmountOfPod
 * (resource_amount.go:71-76).
 * The arrays are copied during the call.  A batch that fits a 64 KB pinned slot (an informer event, or a few dozen
 * coalesced ones) does NOT wait for the device: the call enqueues one kernel and returns; every other entry point — a
 * kt_check issued right behind it included — first waits for the newest such call, so the order "feed the event, then
 * the next PreFilter sees it" holds (kt_delete_pods and kt_upsert_pod likewise).  Larger batches block as before.
 * A batch may name a pod row more than once (coalesced informer events — Add, then Update of one pod): the entries are applied
 * in batch order, the LAST one wins, exactly as if they had arrived in separate calls (incremental engines included). */
int32_t kt_upsert_pods(kt_engine* e, const kt_snapshot* batch, const int64_t* pod_rows);
/* Throttles: spec (threshold, overrides, selector), stored status and reserved amounts of each row. */
int32_t kt_upsert_throttles(kt_engine* e, const kt_snapshot* batch, const int32_t* thr_rows);
int32_t kt_delete_namespaces(kt_engine* e, int32_t n, const int32_t* ns_rows);
int32_t kt_delete_pods(kt_engine* e, int64_t n, const int64_t* pod_rows);
int32_t kt_delete_throttles(kt_engine* e, int32_t n, const int32_t* thr_rows);
/* Clears everything and ingests a whole snapshot (rows = indices). */
int32_t kt_load_snapshot(kt_engine* e, const kt_snapshot* s);

/* ---- single-object forms of the state feed: what ONE informer event handler call pushes (OnAdd / OnUpdate of a
 *      Pod, Throttle / ClusterThrottle, Namespace: throttle_controller.go:400-536, clusterthrottle_controller.go:428-570).
 *      Every pointer is a direct argument to pointer-free memory, no struct of pointers crosses the boundary: a cgo
 *      call may pass Go slices' backing arrays as they are (cgo pointer rule, also on the reference's go 1.20 — no
 *      runtime.Pinner needed).  Same semantics, validation and errors as the batch forms above. -------------------- */
int32_t kt_upsert_namespace(kt_engine* e, int32_t ns_row, int32_t exists, int32_t n_labels, const uint32_t* label_keys,
                            const uint32_t* label_pairs);
/* ctr_req: [n_ctr][D] (D = kt_config.n_dims); ovh: [D] or NULL; ovh_present: bit 31 = spec.overhead != nil */
int32_t kt_upsert_pod(kt_engine* e, int64_t pod_row, uint32_t ns, uint32_t flags, int32_t n_labels, const uint32_t* label_keys,
                      const uint32_t* label_pairs, int32_t n_ctr, const uint8_t* ctr_init, const uint32_t* ctr_present,
                      const int64_t* ctr_req, uint32_t ovh_present, const int64_t* ovh);
/* amounts of the throttle as four rows — 0 spec.threshold, 1 status.calculatedThreshold.threshold, 2 status.used,
 * 3 reserved: amt_v [4][D], amt_present [4], amt_count [4], amt_has_count [4].  Overrides: n_ovr rows (ovr_v [n_ovr][D]).
 * Selector: n_terms terms; term_preq_off / term_nreq_off [n_terms+1] index the two requirement pools, each given as
 * (n, op [n], key [n], val_off [n+1], val []) like kt_reqs.  The shapes are held against each other before anything is
 * stored (offset arrays start at 0 and never decrease, operators are KT_OP_*, masks name dimensions below n_dims, term flags
 * are KT_TERM_*): a slice passed in the wrong position answers KT_ERR_INVALID_ARGUMENT naming the argument (kt_last_error)
 * instead of feeding a wrong selector silently. */
int32_t kt_upsert_throttle(kt_engine* e, int32_t thr_row, uint32_t flags, uint32_t ns, const int64_t* amt_v,
                           const uint32_t* amt_present, const int64_t* amt_count, const uint8_t* amt_has_count,
                           uint32_t thrl_flag, uint32_t thrl_has, uint64_t status_msgs_fp, uint64_t spec_msgs_fp, int32_t n_ovr,
                           const int64_t* ovr_begin_s, const int32_t* ovr_begin_ns, const int64_t* ovr_end_s,
                           const int32_t* ovr_end_ns, const uint8_t* ovr_flags, const int64_t* ovr_v, const uint32_t* ovr_present,
                           const int64_t* ovr_count, const uint8_t* ovr_has_count, int32_t n_terms, const uint8_t* term_flags,
                           const uint32_t* term_preq_off, const uint32_t* term_nreq_off, uint32_t n_preq, const uint8_t* preq_op,
                           const uint32_t* preq_key, const uint32_t* preq_val_off, const uint32_t* preq_val, uint32_t n_nreq,
                           const uint8_t* nreq_op, const uint32_t* nreq_key, const uint32_t* nreq_val_off, const uint32_t* nreq_val);

/* Scheduler-side reserved amounts per throttle — the value reservedResourceAmount(nn) returns
 * (pkg/controllers/reserved_resource_amounts.go:113-126,148-156); the pod map itself stays in Go. */
int32_t kt_set_reserved(kt_engine* e, int32_t n, const int32_t* thr_rows, const kt_amounts* reserved);
/* Stored CR status as the informer cache holds it (what CheckThrottledFor reads, throttle_types.go:128-153). */
typedef struct kt_status {
  kt_amounts used;           /* status.used */
  kt_amounts calc;           /* status.calculatedThreshold.threshold */
  uint8_t* calc_at_nonzero;  /* !status.calculatedThreshold.calculatedAt.IsZero() ; as an OUTPUT: calculatedThreshold replaced, calculatedAt := now */
  uint32_t* thrl_flag;       /* status.throttled.resourceRequests values */
  uint32_t* thrl_has;        /* ... keys */
  uint8_t* thrl_pod;         /* status.throttled.resourceCounts.pod */
  uint64_t* msgs_fp;         /* fingerprint of status.calculatedThreshold.messages (0 = none); input only */
  uint8_t* error;            /* output only: reconcile returned an error for this throttle (selector) */
} kt_status;
int32_t kt_set_status(kt_engine* e, int32_t n, const int32_t* thr_rows, const kt_status* status);

/* ---- reconcile: [Cluster]ThrottleController.reconcile, aggregation part (throttle_controller.go:103-133,
 *      clusterthrottle_controller.go:106-136) for EVERY responsible throttle in one pass:
 *      affectedPods (:221-246 / :224-270) -> used = fold ResourceAmount.Add (resource_amount.go:91-110)
 *      -> CalculateThreshold(now) (throttle_types.go:65-106) -> throttled = IsThrottled(used, true)
 *      (resource_amount.go:127-159).  kt_reconcile_launch = aggregate + finalize on one GPU (it leaves the partial
 *      buffer zeroed: the sums are consumed by the finalize, the next scan starts from a clean buffer).
 *      Multi-GPU (pods row-sharded, throttles replicated): kt_aggregate_launch, all-reduce(sum, int64)
 *      over the buffer kt_partial_used_buffer returns, then kt_finalize_launch. -------------------------- */
#define KT_RECONCILE_APPLY 0x1u /* store the new status as the engine's stored status (UpdateStatus) */
int32_t kt_reconcile_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, void* stream);
/* The same for a SUBSET of throttle rows — the reference reconciles ONE throttle per workqueue key
 * (pkg/controllers/throttle_controller.go:84-133, controller.go:89-122): only the listed rows resolve
 * CalculateThreshold(now), get a new `used` / `throttled` and (with KT_RECONCILE_APPLY) have their stored status
 * replaced; every other row keeps its stored status and reports it unchanged (calc_at_nonzero = 0, no next-override
 * instant).  The scan itself still covers all throttles (it is one pass over the pods either way). */
int32_t kt_reconcile_rows_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, int32_t n,
                                 const int32_t* throttle_rows, void* stream);
int32_t kt_aggregate_launch(kt_engine* e, void* stream);
int32_t kt_partial_used_buffer(kt_engine* e, void** device_ptr, int64_t* n_int64);
/* Optional: aggregate into / finalize from a CALLER-owned device buffer of >= n_int64 words (e.g. the
 * storage of a framework tensor handed to RCCL) instead of the engine's own. NULL restores the default. */
int32_t kt_use_partial_buffer(kt_engine* e, void* device_ptr, int64_t n_int64);
int32_t kt_finalize_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, void* stream);
/* ---- multi-GPU without any framework: one process (or thread) per GPU, each with its own engine over its shard of
 *      the pod rows and a replica of the throttles.  The only exchange of a reconcile is the sum of the partial-`used`
 *      buffers; kt_comm_* runs it as ONE RCCL all-reduce (ncclInt64, ncclSum; xGMI between the GPUs of a node) on the
 *      stream the kernels run on:   kt_aggregate_launch -> kt_comm_allreduce_partial -> kt_finalize_launch.
 *      Rank 0 creates the 128-byte id and hands it to the other ranks by any channel the host has (the Go side would
 *      use its own RPC / a ConfigMap); librccl.so is loaded on the first kt_comm_* call. ------------------------------ */
#define KT_COMM_ID_BYTES 128
int32_t kt_comm_unique_id(void* out_id128);
int32_t kt_comm_init(kt_engine* e, int32_t rank, int32_t world, const void* id128);
/* in-place sum over all ranks of this engine's partial buffer (the caller's, if kt_use_partial_buffer set one) — exactly
 * the words the preceding kt_aggregate_launch filled.  KT_ERR_NOT_READY when no aggregate is pending, or when throttles /
 * namespaces changed since it ran (the ranks would disagree on the word count): aggregate again. */
int32_t kt_comm_allreduce_partial(kt_engine* e, void* stream);
/* A caller that sums the partial buffers with its OWN collective (kt_partial_used_buffer / kt_use_partial_buffer) declares
 * the number of ranks here, so that the engine's exact-range guard covers the sum over all of them (2^60 per rank up to 4
 * ranks, 2^62 / world beyond); kt_comm_init does it by itself. */
int32_t kt_set_exchange_world(kt_engine* e, int32_t world);
/* Wide sums (two blocks of limb sums, see kt_reconcile_fetch_used_hi) change the LAYOUT of the exchanged buffer, and
 * whether a rank's requests leave int64 is a local fact: with more than one rank an engine never goes wide by itself
 * (the aggregate answers KT_ERR_OVERFLOW_RISK instead, as rounds 1-2 did) — the host switches EVERY rank with
 * mode = 1 (always two blocks; exact either way), mode = 0 returns to the per-engine decision.  Not for incremental engines. */
int32_t kt_set_wide_sums(kt_engine* e, int32_t mode);
/* Words (int64) of the aggregate that is pending, and whether they are the two-block form: what a caller's own
 * collective has to sum.  KT_ERR_NOT_READY without a pending kt_aggregate_launch. */
int32_t kt_partial_words(kt_engine* e, int64_t* n_int64, int32_t* wide);
/* The layout of one throttle's row of the partial buffer, for n_dims resource dimensions (int64 words): `stride` words per
 * throttle row; the summed request values start at off_values (n_dims words), the per-key contributor counts — presence
 * travels as counts so that it can be summed — at off_presence (n_dims words), then the counted pods at off_pods and the
 * pods whose selector evaluation failed at off_errors (one word each).  With wide sums the buffer holds two such blocks of
 * throttle_rows x stride words (low 32-bit limbs, then the rest: kt_partial_words).  No engine and no device needed: a host
 * that exchanges the buffer with its own collective — or a test that builds one by hand — lays it out by THIS query, which
 * returns what the kernels compile against (partial_stride / partial_off_* in csrc/kt_device.h). */
int32_t kt_partial_layout(int32_t n_dims, int32_t* stride, int32_t* off_values, int32_t* off_presence, int32_t* off_pods,
                          int32_t* off_errors);
int32_t kt_comm_destroy(kt_engine* e);

/* Copies the last reconcile's result for throttle rows [0, n) into caller arrays (synchronises). */
int32_t kt_reconcile_fetch(kt_engine* e, int32_t n, const kt_status* out);
/* resource.Quantity never overflows (Add promotes to big decimals, pkg/resourcelist/resourcelist.go:48-54).  When the
 * requests of the pods an engine holds add up beyond int64 at the scale they were fed with, the reconcile sums their
 * low 32-bit limbs and the rest separately (two scans, both blocks cross the exchange: kt_partial_used_buffer then
 * reports twice the words) and kt_finalize joins them in 128 bits: `used`, the throttled flags and the check that
 * follows are exact up to 2^124.  kt_reconcile_fetch returns the LOW 64 bits of every value, this call the HIGH 64 bits
 * of rows [0, n) x n_dims (two's complement; out_any_wide, nullable: some value really left int64).  A status fed back
 * through kt_set_status / kt_upsert_throttles is int64.  Not available to KT_VARIANT_INCREMENTAL engines and
 * kt_admit_launch (KT_ERR_OVERFLOW_RISK / KT_ERR_UNSUPPORTED there). */
int32_t kt_reconcile_fetch_used_hi(kt_engine* e, int32_t n, int64_t* out_hi, int32_t* out_any_wide);
/* ThrottleSpecBase.NextOverrideHappensIn(now) (throttle_types.go:37-63) of the last reconcile for throttle rows
 * [0, n), as the INSTANT of the next override boundary (the controller's enqueueAfter delay is instant - now,
 * throttle_controller.go:201-208); has[i] = 0 when nothing lies ahead or the row was not reconciled. Synchronises. */
int32_t kt_reconcile_fetch_next_override(kt_engine* e, int32_t n, int64_t* next_s, int32_t* next_ns, uint8_t* has);

/* ---- check: KubeThrottler.PreFilter (plugin.go:148-215) = ThrottleController.CheckThrottled
 *      (throttle_controller.go:349-397) + ClusterThrottleController.CheckThrottled
 *      (clusterthrottle_controller.go:378-425) -> [Cluster]Throttle.CheckThrottledFor
 *      (throttle_types.go:128-153, clusterthrottle_types.go:30-55) for n pods at once against the stored
 *      status + reserved amounts.  pod_rows == NULL checks rows [0, n).  on_equal is the
 *      isThrottledOnEqual argument (PreFilter passes false). ------------------------------------------- */
#define KT_CHECK_STATUS_MATRIX 0x1u /* also produce the n x throttle_rows status matrix (parity / reason strings) */
int32_t kt_check_launch(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint32_t flags,
                        void* stream);
/* The PreFilter sweep of EVERY pod row against the stored status and the reconcile of every throttle as ONE pass over the
 * pod tables: what a host that re-evaluates the whole cluster calls instead of kt_check_launch(all rows) followed by
 * kt_reconcile_launch.  Results (kt_check_fetch for rows [0, pod rows in use), kt_reconcile_fetch) are bit for bit those
 * of that pair: the verdicts are PreFilter's against the status stored BEFORE this reconcile (plugin.go:148-215 reads the
 * informer cache the controllers write later), the new status is reconcile's (throttle_controller.go:84-133).  One
 * selector scan per pod instead of two where the program allows it (one index chunk, no slow list, requests that pack);
 * otherwise the two launches run one after the other inside the call.  flags: KT_RECONCILE_APPLY. */
int32_t kt_sweep_launch(kt_engine* e, int64_t now_s, int32_t now_ns, uint32_t flags, int32_t on_equal, void* stream);
/* out_summary [n] ; out_status [n][n_throttle_rows] (nullable; needs KT_CHECK_STATUS_MATRIX), where
 * n_throttle_rows = 1 + highest throttle row ever upserted (kt_throttle_rows). Synchronises. */
int32_t kt_check_fetch(kt_engine* e, int64_t n, uint64_t* out_summary, uint8_t* out_status);
/* kt_check_launch + kt_check_fetch as ONE critical section on the engine's own stream (out_status nullable: no
 * matrix is produced then).  This is the form a PreFilter shim calls when other threads use the engine concurrently
 * (Unreserve from binding goroutines plugin.go:240-257, reconcile workers controller.go:52-122): results of a
 * separate launch / fetch pair may be replaced by another thread's launch in between. */
/* With n <= 8 and out_status == NULL (one PreFilter call: the verdict; the status row is only needed to word the reasons
 * of a blocked pod) the call takes the few-pod path: it holds the engine lock SHARED, launches one wave per index chunk
 * on a high-priority stream of its own and spins on a sequence number the kernel writes to pinned host memory behind
 * the summary words — no copy, no stream synchronisation, and no waiting for the kernels of a reconcile another thread
 * launched (while those run it sees the status as stored before that reconcile: the CheckRecs are double-buffered). */
int32_t kt_check(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint64_t* out_summary,
                 uint8_t* out_status);
/* affectedPods for pods the caller names — pkg/controllers/throttle_controller.go:221-246, clusterthrottle_controller.go:224-270
 * restricted to n pod rows x m throttle rows: out[i * m + j] = 1 when throttle_rows[j]'s selector (its namespace side
 * included) matches pod_rows[i] as the engine holds it now, 0 when not, KT_STATUS_ERROR when the pod's PreFilter is an
 * error (unknown namespace object, an unconvertible selector reached first).  The caller adds shouldCountIn
 * (throttle_controller.go:217-219: it knows scheduler name and node of its pods).  This is what unreserveAffectedPods
 * (throttle_controller.go:135-155) iterates over: behind a reconcile, a reservation is released only for a pod that is in the
 * reconciled throttle's affected set — a pod whose labels changed after Reserve is not.  Cost: one small check launch, on the
 * engine's ONE check slot: a kt_check_launch that was pending is dropped (its kt_check_fetch answers KT_ERR_NOT_READY; kt_check,
 * the one-call form a shim uses beside other threads, is not affected). */
int32_t kt_affected_pods(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t m, const int32_t* throttle_rows, uint8_t* out);
/* ---- sequential admission with reservation: for i = 0..n-1 IN ORDER,
 *      PreFilter(pod_rows[i]) (plugin.go:148-215) and, on Success, Reserve(pod_rows[i]) (plugin.go:217-239 ->
 *      [Cluster]ThrottleController.Reserve, throttle_controller.go:271-300 -> reservedResourceAmounts.addPod,
 *      reserved_resource_amounts.go:66-77): ResourceAmountOfPod(pod) is added to the reserved amount of every
 *      throttle that affects the pod, and the following pods of the queue are checked against it.  One launch
 *      replaces n PreFilter + Reserve round trips.  Results are read with kt_check_fetch: out_summary[i] /
 *      out_status[i][*] are what PreFilter returned for pod i AT ITS TURN.
 *      flags: KT_ADMIT_COMMIT keeps the resulting reserved amounts in the engine (as if Reserve had been called
 *      for every admitted pod; read them back with kt_fetch_reserved); without it the call is a dry run.
 *      Every admitted pod ADDS its amount: the reference's cache is a map keyed by pod (reserved_resource_amounts.go:
 *      130-135), so a pod whose amount is already part of the reserved totals, or that occurs twice in the queue, must
 *      not be in it — the caller ends the queue there and takes that pod through kt_check + its own map (the C++ plugin
 *      mirror's AdmitQueue does exactly that).
 *      Limit: n x throttle_rows <= 2^31 (the status matrix).  The reserved amounts of all throttles live in LDS while
 *      they fit (throttle_rows x (8 x n_dims + 16) <= ~156 KB), beyond that in HBM (same results, L2 latency per
 *      pod). ---------------------------------------------------------------------------------------------------- */
#define KT_ADMIT_COMMIT 0x1u
int32_t kt_admit_launch(kt_engine* e, int64_t n, const int64_t* pod_rows, int32_t on_equal, uint32_t flags,
                        void* stream);
/* Current reserved amounts of n throttle rows (after kt_set_reserved / kt_admit_launch(KT_ADMIT_COMMIT)). */
int32_t kt_fetch_reserved(kt_engine* e, int32_t n, const int32_t* throttle_rows, const kt_amounts* out);
int32_t kt_throttle_rows(kt_engine* e, int32_t* out_rows);
/* Device pointer of the per-pod summary words of the last check (stays valid until the next check). */
int32_t kt_check_device_summary(kt_engine* e, void** device_ptr);

/* Effective per-pod requests as held in HBM (parity of the resourcelist summation): out_v [n][D]. */
int32_t kt_fetch_pod_requests(kt_engine* e, int64_t n, const int64_t* pod_rows, int64_t* out_v, uint32_t* out_present);

/* ---- measurement: HIP-event timing of the engine's own kernels on the stream they run on. ------------- */
#define KT_KERNEL_CHECK 0
#define KT_KERNEL_AGGREGATE 1
#define KT_KERNEL_FINALIZE 2
#define KT_KERNEL_PREPARE 3
#define KT_KERNEL_REDUCE 4 /* slab reduction that follows the aggregate scan kernel (when it privatises in LDS) */
#define KT_KERNEL_COUNT 5
int32_t kt_timing_enable(kt_engine* e, int32_t on);
/* Sum of durations (ms) and launch count since the last reset for one kernel family; synchronises. */
int32_t kt_timing_read(kt_engine* e, int32_t kernel, double* total_ms, int64_t* launches);
int32_t kt_timing_reset(kt_engine* e);
int32_t kt_synchronize(kt_engine* e, void* stream);
/* symbol of the HIP kernel LAST dispatched for a family (to match rocprofv3 kernel-trace rows) */
const char* kt_kernel_name(kt_engine* e, int32_t kernel);
/* event counters (-1: unknown counter) */
#define KT_COUNTER_FEW_CHECKS 0 /* kt_check calls served by the few-pod path (shared lock, no copy, no stream sync) */
#define KT_COUNTER_COMPILES 1   /* selector program compiles + index builds so far (a Throttle event that leaves every
                                   selector, namespace and flag of its rows as stored — a threshold edit, a status update —
                                   only re-uploads the throttle tables and does not count) */
#define KT_COUNTER_INDEX_CHUNKS 2 /* LDS-sized chunks of the compiled selector index (0 before the first compile) */
#define KT_COUNTER_INDEX_WORDS 3  /* 64-bit words of term numbers of the compiled program (all chunks) */
#define KT_COUNTER_NS_WORD_VISITS 4 /* sum over the namespace rows in use of the words a pod of that namespace visits */
#define KT_COUNTER_NS_ROWS 5      /* namespace rows the compiled program covers */
#define KT_COUNTER_NS_CHUNK_VISITS 6  /* sum over the namespace rows of the index chunks that hold a word list of the row: the chunk
                                         passes a namespace-ordered scan makes per namespace, summed */
#define KT_COUNTER_INDEX_IMAGE_WORDS 7 /* words over all chunk images (>= INDEX_WORDS: the grouped plan keeps copies of a word in
                                          the chunks of every group of namespaces that visits it) */
#define KT_COUNTER_SLOW_THROTTLES 8 /* throttles of the compiled program that are walked term by term instead of through the index
                                      (an unconvertible podSelector term; more than 512 selector terms) */
#define KT_COUNTER_PACKED_WORDS 9   /* 64-bit words per pod of the packed fold the last full aggregate scan ran with (1..8; 0: the
                                      plain fold — a negative request, sums beyond int64, fields that do not fit) */
int64_t kt_counter(kt_engine* e, int32_t which);
/* ---- More resource names than one engine has dimensions (KT_MAX_DIMS): PAGES.  The reference sums and compares any resource
 *      name (pkg/resourcelist/resourcelist.go:27-54, resource_amount.go:127-159).  The host builds the same cluster once per
 *      page of <= KT_MAX_DIMS names — every page engine holds every pod row and every throttle row, with the requests /
 *      thresholds of ITS names — and these two calls run a step on every page and combine the results.  The combination is
 *      exact: every step of CheckThrottledFor (throttle_types.go:128-153) is `count part OR exists a resource name ...`; the
 *      count part needs no name (every page computes it alike) and the name part of the cluster is the OR over the pages:
 *          exceeds <=> some page says exceeds; else active <=> some page says active; else insufficient <=> some page says so;
 *      a pod-level Error shows in every page.  (kube_throttler_amd/paging.py is the same statement in Python.) -------------- */
/* kt_check on every page, combined: out_status [n][throttle rows] (nullable) and out_summary [n] (nullable; verdict and the
 * three class counts of the COMBINED row).  The engines are called one after the other (each call is its own critical section). */
int32_t kt_paged_check(kt_engine* const* pages, int32_t n_pages, int64_t n, const int64_t* pod_rows, int32_t on_equal,
                       uint64_t* out_summary, uint8_t* out_status);
/* kt_reconcile_launch + kt_reconcile_fetch on every page: page_out[k] receives page k's result for throttle rows [0, n) —
 * `used`, calculatedThreshold and `throttled` are per resource name, each name comes from the page that owns it; the pod
 * counts and the pod flag are the same in every page.  replaced_any[i] (nullable) = calculatedThreshold replaced in some
 * page (it is replaced as a whole), error_any[i] (nullable) = the reconcile of row i failed.
 * Failure: with KT_RECONCILE_APPLY every page is first reconciled as a dry run — a page that cannot (LDS budgets, sums out of
 * range) fails the call before ANY page has stored a new status; an error after that (device errors) is returned once every
 * launched page's result has been drained, never with pending reconciles left behind. */
int32_t kt_paged_reconcile(kt_engine* const* pages, int32_t n_pages, int64_t now_s, int32_t now_ns, uint32_t flags, int32_t n,
                           const kt_status* page_out, uint8_t* replaced_any, uint8_t* error_any);

/* Development aid: the engine reads its A/B switches (KT_NO_* / KT_SYNC_INGEST ... environment variables, all off by default)
 * once, at kt_engine_create; a tool that flips one on a live engine calls this afterwards. */
int32_t kt_debug_reload_env(kt_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* KT_ENGINE_H */
