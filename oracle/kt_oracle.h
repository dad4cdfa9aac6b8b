/*
 * kt_oracle.h — CPU oracle for the kube-throttler hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (per pod -> list throttles -> build
 * selector -> match -> 4-step CheckThrottledFor over sparse name->Quantity maps; per throttle ->
 * scan pods -> fold Add), following /root/reference file:line as cited on every function in
 * kt_oracle.c.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (kube_throttler_amd, libkt_engine.so) never does.
 *
 * Parity status: the reference is Go and cannot be built here (no Go toolchain, ~110 un-vendored
 * modules), so this oracle is pinned against the reference's OWN test tables and integration
 * scenarios transcribed in tests/golden/ (G1-G5).  Behaviour that lives in
 * k8s.io/apimachinery v0.26.4 and is touched by no reference test (matchExpressions operators,
 * invalid-selector paths, Quantity beyond int64) is restated from its published semantics and is
 * "parity unpinned" — see DESIGN.md.
 */
#ifndef KT_ORACLE_H
#define KT_ORACLE_H

#include <stdint.h>
#include "../include/kt_snapshot.h"

#ifdef __cplusplus
extern "C" {
#endif

/* per (pod, throttle) status — CheckThrottleStatus (throttle_types.go:119-126) plus
 * "not affected" (selector/namespace/throttlerName did not match) and "error". */
#define KTO_NOT_AFFECTED 0
#define KTO_NOT_THROTTLED 1
#define KTO_ACTIVE 2
#define KTO_INSUFFICIENT 3
#define KTO_EXCEEDS 4
#define KTO_ERROR 255

/* per-pod summary word (what PreFilter derives, plugin.go:148-215):
 * bits 0-1 verdict (0 Success, 1 UnschedulableAndUnresolvable, 2 Error),
 * bits 4-23 #pod-requests-exceeds-threshold, 24-43 #active, 44-63 #insufficient. */
#define KTO_VERDICT_ALLOW 0
#define KTO_VERDICT_BLOCK 1
#define KTO_VERDICT_ERROR 2

typedef struct kto_ctx kto_ctx;

/* Builds the per-namespace indexes the informer caches provide (listers/.../throttle.go:49-100). */
kto_ctx* kto_create(const kt_snapshot* s);
void kto_destroy(kto_ctx* c);
/* TEST MODE: memoise ClusterThrottleSelectorTerm.MatchesToNamespace per (term, namespace) — a pure function of the two
 * objects that the literal loops re-evaluate for every pod.  Results are unchanged; only kto_check(mimic_log_args = 0),
 * kto_reconcile and kto_admit use it — the timed CPU baseline (mimic_log_args = 1) always runs the literal evaluation.
 * Returns 1 when the table is in place (0: too large / no terms: the literal evaluation stays). */
int kto_enable_ns_memo(kto_ctx* c);

/* resourcelist.PodRequestResourceList for n pods (rows NULL => 0..n-1). out_v [n][D]. */
int kto_pod_requests(kto_ctx* c, int64_t n, const int64_t* rows, int64_t* out_v, uint32_t* out_present);

/* KubeThrottler.PreFilter for n pods against the snapshot's stored status.
 * out_status [n][n_thr] (nullable), out_summary [n] (nullable).  mimic_log_args != 0 also performs
 * the work Go's eager klog argument evaluation does per affected throttle
 * (throttle_controller.go:376-386).  Returns 0. */
int kto_check(kto_ctx* c, int64_t n, const int64_t* rows, int on_equal, uint8_t* out_status,
              uint64_t* out_summary, int nthreads, int mimic_log_args);

/* One scheduling pass over a queue, in order: PreFilter, and on Success Reserve (which the next pods see).
 * out_reserved: nullable, n_thr rows — reserved totals after the pass. */
int kto_admit(kto_ctx* c, int64_t n, const int64_t* rows, int on_equal, uint8_t* out_status, uint64_t* out_summary,
              const kt_amounts* out_reserved);

typedef struct kto_reconcile_out {
  kt_amounts used;       /* new status.used                                   [n][D] ... */
  kt_amounts calc;       /* new status.calculatedThreshold.threshold                      */
  uint8_t* calc_updated; /* calculatedThreshold replaced (calculatedAt := now)            */
  uint32_t* thrl_flag;   /* new status.throttled.resourceRequests values                  */
  uint32_t* thrl_has;    /* ... keys                                                      */
  uint8_t* thrl_pod;     /* new status.throttled.resourceCounts.pod                       */
  uint8_t* error;        /* reconcile returned an error (selector); status left untouched */
} kto_reconcile_out;

/* [Cluster]ThrottleController.reconcile aggregation part for n throttles (rows NULL => 0..n-1);
 * out rows are indexed by position i (not by throttle row). */
int kto_reconcile(kto_ctx* c, int32_t n, const int32_t* rows, int64_t now_s, int32_t now_ns,
                  kto_reconcile_out* out, int nthreads);

/* resource.Quantity never overflows: status_used_hi (nullable, [n_thr][D]) = high 64 bits of the snapshot's status.used
 * values (thr_used.v are the low words); out_used_hi (nullable, [n][D] by output position) receives the high words of what
 * kto_reconcile computes — its error flag then no longer reports sums beyond int64.  The oracle keeps the pointers. */
void kto_set_wide(kto_ctx* c, const int64_t* status_used_hi, int64_t* out_used_hi);

/* ThrottleSpecBase.NextOverrideHappensIn for n throttles (rows NULL => 0..n-1), as the instant (has = 0: none). */
int kto_next_override(kto_ctx* c, int32_t n, const int32_t* rows, int64_t now_s, int32_t now_ns, int64_t* out_s,
                      int32_t* out_ns, uint8_t* out_has);

/* Function-level entry points used to replay the reference's unit-test tables (tests/test_oracle_unit_tables.py). */
int kto_unit_is_throttled(int D, const kt_amounts* threshold, const kt_amounts* used, int on_equal,
                          uint32_t* out_flag, uint32_t* out_has, uint8_t* out_pod);
int kto_unit_is_throttled_for(const kt_snapshot* s, int64_t pod_row, uint32_t flag, uint32_t has, int pod_flag);
int kto_unit_override_is_active(const kt_snapshot* s, uint32_t o, int64_t now_s, int32_t now_ns);
int kto_unit_calculate_threshold(const kt_snapshot* s, int32_t t, int64_t now_s, int32_t now_ns,
                                 const kt_amounts* out, uint8_t* out_any_err);
int kto_unit_selector_matches(const kt_snapshot* s, int32_t t, int64_t pod_row);

#ifdef __cplusplus
}
#endif
#endif
