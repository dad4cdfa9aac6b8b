// Package engine is the cgo binding of libkt_engine.so (include/kt_engine.h) for kube-throttler: the drop-in for the per-pod
// scan of PreFilter / CheckThrottled and the aggregation half of reconcile.
//
// NOT COMPILED HERE: the image this repository is built in has no Go toolchain (`go: command not found`), so this file has
// never seen `go build` or `go vet`.  It is written against go 1.20 (the reference's go.mod:3) and the header as committed;
// every C call below names the exports of include/kt_engine.h argument for argument — the same calls the test-suite makes
// through ctypes (kube_throttler_amd/engine.py) and plain C99 (tests/c/abi_flat_test.c), which DO run on the GPU box.
// A maintainer drops this directory into pkg/engine of the reference tree, applies integration/go/patches/*.patch and builds
// with the Makefile flip described there (CGO_ENABLED=1).
//
// cgo pointer rule: every slice handed to C is a direct argument to pointer-free memory and only for the duration of the
// call — the engine copies what it needs and retains nothing — so no runtime.Pinner (go >= 1.21) is required.
package engine

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -lkt_engine
#include <stdlib.h>
#include "kt_engine.h"
*/
import "C"

import (
	"fmt"
	"time"
	"unsafe"
)

// CheckThrottleStatus codes of the status matrix (schedulev1alpha1.CheckThrottleStatus*, throttle_types.go:108-126).
const (
	StatusNotAffected    uint8 = 0
	StatusNotThrottled   uint8 = 1
	StatusActive         uint8 = 2
	StatusInsufficient   uint8 = 3
	StatusPodExceeds     uint8 = 4
	StatusError          uint8 = 255
	VerdictSuccess             = 0 // summary word, bits 0-1
	VerdictUnschedulable       = 1
	VerdictError               = 2
)

// Engine owns one kt_engine (one GPU).  Methods are safe from any goroutine / OS thread (every export selects its device).
type Engine struct {
	h    *C.kt_engine
	dims int
}

// New creates an engine for `dims` resource names (<= 16), pods with up to maxLabels labels and the given row capacities.
func New(dims, maxLabels int, podCap int64, thrCap, nsCap int, incremental bool) (*Engine, error) {
	cfg := C.kt_config{n_dims: C.int32_t(dims), max_labels: C.int32_t(maxLabels), pod_capacity: C.int64_t(podCap),
		throttle_capacity: C.int32_t(thrCap), namespace_capacity: C.int32_t(nsCap), device: -1}
	if incremental {
		cfg.kernel_variant = C.KT_VARIANT_INCREMENTAL
	}
	var h *C.kt_engine
	if rc := C.kt_engine_create(&cfg, &h); rc != C.KT_OK {
		return nil, fmt.Errorf("kt_engine_create: %d: %s", int(rc), C.GoString(C.kt_last_error(nil)))
	}
	return &Engine{h: h, dims: dims}, nil
}

// Close releases the device memory of the engine.
func (e *Engine) Close() {
	if e.h != nil {
		C.kt_engine_destroy(e.h)
		e.h = nil
	}
}

func (e *Engine) err(rc C.int32_t) error {
	return fmt.Errorf("kt: %d: %s", int(rc), C.GoString(C.kt_last_error(e.h)))
}

func u32(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}
func u8(s []uint8) *C.uint8_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&s[0]))
}
func i64(s []int64) *C.int64_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.int64_t)(unsafe.Pointer(&s[0]))
}
func i32(s []int32) *C.int32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.int32_t)(unsafe.Pointer(&s[0]))
}
func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// ---- state feed: what the informer event handlers push (throttle_controller.go:400-536, clusterthrottle_controller.go:428-570)

// UpsertNamespace replaces the Namespace informer's Add / Update handlers for row nsRow.
func (e *Engine) UpsertNamespace(nsRow int32, exists bool, keys, pairs []uint32) error {
	if rc := C.kt_upsert_namespace(e.h, C.int32_t(nsRow), b2i(exists), C.int32_t(len(keys)), u32(keys), u32(pairs)); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// UpsertPod pushes ONE pod event (kt_upsert_pod: returns without waiting for the device, 5 us).  ctrReq is [len(ctrInit)][dims].
func (e *Engine) UpsertPod(row int64, ns, flags uint32, keys, pairs []uint32, ctrInit []uint8, ctrPresent []uint32,
	ctrReq []int64, ovhPresent uint32, ovh []int64) error {
	if rc := C.kt_upsert_pod(e.h, C.int64_t(row), C.uint32_t(ns), C.uint32_t(flags), C.int32_t(len(keys)), u32(keys), u32(pairs),
		C.int32_t(len(ctrInit)), u8(ctrInit), u32(ctrPresent), i64(ctrReq), C.uint32_t(ovhPresent), i64(ovh)); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// DeletePods replaces the pod informer's Delete handler.
func (e *Engine) DeletePods(rows []int64) error {
	if len(rows) == 0 {
		return nil
	}
	if rc := C.kt_delete_pods(e.h, C.int64_t(len(rows)), i64(rows)); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// Amounts is one ResourceAmount at the engine's scales: V[d] for the names in Present, the pod count when HasCount.
type Amounts struct {
	V        []int64 // [dims]
	Present  uint32
	Count    int64
	HasCount bool
}

// Override is one TemporaryThresholdOverride with its instants parsed once (temporary_threshold_override.go:40-70).
type Override struct {
	BeginS, EndS   int64
	BeginNs, EndNs int32
	Flags          uint8 // KT_OVR_*
	Threshold      Amounts
}

// Requirements is one pool of selector requirements (kt_reqs): Op[i], Key[i], values ValOff[i]..ValOff[i+1] of Val.
type Requirements struct {
	Op     []uint8
	Key    []uint32
	ValOff []uint32
	Val    []uint32
}

// ThrottleRow is everything kt_upsert_throttle takes for one Throttle / ClusterThrottle.
type ThrottleRow struct {
	Flags, Namespace           uint32
	Spec, Calc, Used, Reserved Amounts
	ThrlFlag, ThrlHas          uint32
	StatusMsgsFP, SpecMsgsFP   uint64
	Overrides                  []Override
	TermFlags                  []uint8
	TermPreqOff, TermNreqOff   []uint32 // [terms+1]
	PodReqs, NsReqs            Requirements
}

// UpsertThrottle replaces the Throttle / ClusterThrottle informer handlers (throttle_controller.go:401-430).  The engine holds
// the 37 argument shapes against each other and answers KT_ERR_INVALID_ARGUMENT naming the array on a mis-feed.
func (e *Engine) UpsertThrottle(row int32, t *ThrottleRow) error {
	D := e.dims
	amtV := make([]int64, 4*D)
	amtPresent := make([]uint32, 4)
	amtCount := make([]int64, 4)
	amtHas := make([]uint8, 4)
	for k, a := range []*Amounts{&t.Spec, &t.Calc, &t.Used, &t.Reserved} {
		copy(amtV[k*D:(k+1)*D], a.V)
		amtPresent[k], amtCount[k] = a.Present, a.Count
		if a.HasCount {
			amtHas[k] = 1
		}
	}
	n := len(t.Overrides)
	bs, es := make([]int64, n), make([]int64, n)
	bn, en := make([]int32, n), make([]int32, n)
	fl, oh := make([]uint8, n), make([]uint8, n)
	ov, oc := make([]int64, n*D), make([]int64, n)
	op := make([]uint32, n)
	for i, o := range t.Overrides {
		bs[i], es[i], bn[i], en[i], fl[i] = o.BeginS, o.EndS, o.BeginNs, o.EndNs, o.Flags
		copy(ov[i*D:(i+1)*D], o.Threshold.V)
		op[i], oc[i] = o.Threshold.Present, o.Threshold.Count
		if o.Threshold.HasCount {
			oh[i] = 1
		}
	}
	rc := C.kt_upsert_throttle(e.h, C.int32_t(row), C.uint32_t(t.Flags), C.uint32_t(t.Namespace), i64(amtV), u32(amtPresent),
		i64(amtCount), u8(amtHas), C.uint32_t(t.ThrlFlag), C.uint32_t(t.ThrlHas), C.uint64_t(t.StatusMsgsFP), C.uint64_t(t.SpecMsgsFP),
		C.int32_t(n), i64(bs), i32(bn), i64(es), i32(en), u8(fl), i64(ov), u32(op), i64(oc), u8(oh),
		C.int32_t(len(t.TermFlags)), u8(t.TermFlags), u32(t.TermPreqOff), u32(t.TermNreqOff),
		C.uint32_t(len(t.PodReqs.Op)), u8(t.PodReqs.Op), u32(t.PodReqs.Key), u32(t.PodReqs.ValOff), u32(t.PodReqs.Val),
		C.uint32_t(len(t.NsReqs.Op)), u8(t.NsReqs.Op), u32(t.NsReqs.Key), u32(t.NsReqs.ValOff), u32(t.NsReqs.Val))
	if rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// DeleteThrottles replaces the Throttle / ClusterThrottle Delete handlers.
func (e *Engine) DeleteThrottles(rows []int32) error {
	if len(rows) == 0 {
		return nil
	}
	if rc := C.kt_delete_throttles(e.h, C.int32_t(len(rows)), i32(rows)); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// cAmounts builds a kt_amounts over C memory for n rows (freed by the returned func): a struct of pointers may not point into
// Go memory on go 1.20, so the few calls that take kt_amounts / kt_status stage through C.malloc.
func (e *Engine) cAmounts(n int) (C.kt_amounts, func()) {
	D := e.dims
	a := C.kt_amounts{
		v:         (*C.int64_t)(C.calloc(C.size_t(n*D+1), 8)),
		present:   (*C.uint32_t)(C.calloc(C.size_t(n+1), 4)),
		count:     (*C.int64_t)(C.calloc(C.size_t(n+1), 8)),
		has_count: (*C.uint8_t)(C.calloc(C.size_t(n+1), 1)),
	}
	return a, func() {
		C.free(unsafe.Pointer(a.v))
		C.free(unsafe.Pointer(a.present))
		C.free(unsafe.Pointer(a.count))
		C.free(unsafe.Pointer(a.has_count))
	}
}

// SetReserved pushes reservedResourceAmount(nn) of the listed throttles (reserved_resource_amounts.go:113-126); the pod map
// itself stays in Go.
func (e *Engine) SetReserved(rows []int32, amounts []Amounts) error {
	n, D := len(rows), e.dims
	if n == 0 {
		return nil
	}
	ca, free := e.cAmounts(n)
	defer free()
	v := unsafe.Slice((*int64)(unsafe.Pointer(ca.v)), n*D)
	p := unsafe.Slice((*uint32)(unsafe.Pointer(ca.present)), n)
	c := unsafe.Slice((*int64)(unsafe.Pointer(ca.count)), n)
	h := unsafe.Slice((*uint8)(unsafe.Pointer(ca.has_count)), n)
	for i, a := range amounts {
		copy(v[i*D:(i+1)*D], a.V)
		p[i], c[i] = a.Present, a.Count
		if a.HasCount {
			h[i] = 1
		}
	}
	if rc := C.kt_set_reserved(e.h, C.int32_t(n), i32(rows), &ca); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// ---- admission: KubeThrottler.PreFilter (plugin.go:148-215)

// Verdict is what PreFilter calls first: ONE pod, the summary word only — the few-pod path of kt_check (shared engine lock, no
// copy, no stream synchronisation: 9-10 us; not faster than ONE CPU core doing one PreFilter at 1k throttles — the engine wins
// batched, see INTEGRATION.md).  Bits 0-1 of the word are the verdict, the three 20-bit fields count the throttles per class.
func (e *Engine) Verdict(podRow int64, onEqual bool) (uint64, error) {
	var summary C.uint64_t
	row := C.int64_t(podRow)
	if rc := C.kt_check(e.h, 1, &row, b2i(onEqual), &summary, nil); rc != C.KT_OK {
		return 0, e.err(rc)
	}
	return uint64(summary), nil
}

// ThrottleRows is 1 + the highest throttle row ever upserted: the width of a status row.
func (e *Engine) ThrottleRows() int {
	var t C.int32_t
	C.kt_throttle_rows(e.h, &t)
	return int(t)
}

// Check replaces the bodies of ThrottleController.CheckThrottled + ClusterThrottleController.CheckThrottled
// (throttle_controller.go:349-397, clusterthrottle_controller.go:378-425) for one or many pods: status[i*T+t] is the
// CheckThrottleStatus code of (pod i, throttle row t).  kt_check = launch + fetch under ONE engine lock.
func (e *Engine) Check(podRows []int64, onEqual bool) (summary []uint64, status []uint8, err error) {
	if len(podRows) == 0 {
		return nil, nil, nil
	}
	T := e.ThrottleRows()
	summary = make([]uint64, len(podRows))
	status = make([]uint8, len(podRows)*T+1)
	if rc := C.kt_check(e.h, C.int64_t(len(podRows)), i64(podRows), b2i(onEqual),
		(*C.uint64_t)(unsafe.Pointer(&summary[0])), u8(status)); rc != C.KT_OK {
		return nil, nil, e.err(rc)
	}
	return summary, status[:len(podRows)*T], nil
}

// AffectedPods answers "does throttle j's selector match pod i as held now" (affectedPods restricted to the named rows:
// throttle_controller.go:221-246) — what unreserveAffectedPods iterates over behind a reconcile.
func (e *Engine) AffectedPods(podRows []int64, throttleRows []int32) ([]uint8, error) {
	out := make([]uint8, len(podRows)*len(throttleRows)+1)
	if rc := C.kt_affected_pods(e.h, C.int64_t(len(podRows)), i64(podRows), C.int32_t(len(throttleRows)), i32(throttleRows), u8(out)); rc != C.KT_OK {
		return nil, e.err(rc)
	}
	return out[:len(podRows)*len(throttleRows)], nil
}

// Admit runs PreFilter + Reserve for rows[i] IN ORDER in one launch (plugin.go:148-239); a dry run unless commit.
func (e *Engine) Admit(rows []int64, onEqual, commit bool) (summary []uint64, status []uint8, err error) {
	var flags C.uint32_t
	if commit {
		flags = C.KT_ADMIT_COMMIT
	}
	if rc := C.kt_admit_launch(e.h, C.int64_t(len(rows)), i64(rows), b2i(onEqual), flags, nil); rc != C.KT_OK {
		return nil, nil, e.err(rc)
	}
	T := e.ThrottleRows()
	summary = make([]uint64, len(rows))
	status = make([]uint8, len(rows)*T+1)
	if rc := C.kt_check_fetch(e.h, C.int64_t(len(rows)), (*C.uint64_t)(unsafe.Pointer(&summary[0])), u8(status)); rc != C.KT_OK {
		return nil, nil, e.err(rc)
	}
	return summary, status[:len(rows)*T], nil
}

// ---- aggregation: [Cluster]ThrottleController.reconcile (throttle_controller.go:103-133, clusterthrottle_controller.go:106-136)

// Status is the reconcile result of n throttle rows at the engine's scales.
type Status struct {
	Used, Calc      []Amounts
	CalcReplaced    []bool   // calculatedThreshold was replaced (calculatedAt := now)
	ThrottledFlag   []uint32 // status.throttled.resourceRequests values ...
	ThrottledHas    []uint32 // ... and keys
	ThrottledPod    []bool   // status.throttled.resourceCounts.pod
	Err             []bool   // reconcile returned an error for the row (selector)
	NextOverride    []time.Time
	HasNextOverride []bool
}

// ReconcileRows replaces the aggregation of reconcile(key) for the dirty keys of a worker batch: affectedPods -> used = fold Add
// -> CalculateThreshold(now) -> throttled = IsThrottled(used, true), with KT_RECONCILE_APPLY the result becomes the stored status
// (what a successful UpdateStatus would feed back through the informer).  The scan is one pass over the pods whatever the
// number of keys: drain the workqueue into ONE call.  rows == nil reconciles every row (resync).
func (e *Engine) ReconcileRows(now time.Time, apply bool, rows []int32) (*Status, error) {
	var flags C.uint32_t
	if apply {
		flags = C.KT_RECONCILE_APPLY
	}
	var rc C.int32_t
	if rows == nil {
		rc = C.kt_reconcile_launch(e.h, C.int64_t(now.Unix()), C.int32_t(now.Nanosecond()), flags, nil)
	} else {
		rc = C.kt_reconcile_rows_launch(e.h, C.int64_t(now.Unix()), C.int32_t(now.Nanosecond()), flags, C.int32_t(len(rows)), i32(rows), nil)
	}
	if rc != C.KT_OK {
		return nil, e.err(rc)
	}
	n, D := e.ThrottleRows(), e.dims
	used, freeU := e.cAmounts(n)
	defer freeU()
	calc, freeC := e.cAmounts(n)
	defer freeC()
	alloc := func(sz int) unsafe.Pointer { return C.calloc(C.size_t(n+1), C.size_t(sz)) }
	st := C.kt_status{used: used, calc: calc,
		calc_at_nonzero: (*C.uint8_t)(alloc(1)), thrl_flag: (*C.uint32_t)(alloc(4)), thrl_has: (*C.uint32_t)(alloc(4)),
		thrl_pod: (*C.uint8_t)(alloc(1)), msgs_fp: nil, error: (*C.uint8_t)(alloc(1))}
	defer func() {
		for _, p := range []unsafe.Pointer{unsafe.Pointer(st.calc_at_nonzero), unsafe.Pointer(st.thrl_flag), unsafe.Pointer(st.thrl_has),
			unsafe.Pointer(st.thrl_pod), unsafe.Pointer(st.error)} {
			C.free(p)
		}
	}()
	if rc := C.kt_reconcile_fetch(e.h, C.int32_t(n), &st); rc != C.KT_OK {
		return nil, e.err(rc)
	}
	out := &Status{Used: make([]Amounts, n), Calc: make([]Amounts, n), CalcReplaced: make([]bool, n), ThrottledFlag: make([]uint32, n),
		ThrottledHas: make([]uint32, n), ThrottledPod: make([]bool, n), Err: make([]bool, n)}
	rd := func(a C.kt_amounts, i int) Amounts {
		v := unsafe.Slice((*int64)(unsafe.Pointer(a.v)), n*D)
		return Amounts{V: append([]int64(nil), v[i*D:(i+1)*D]...),
			Present:  unsafe.Slice((*uint32)(unsafe.Pointer(a.present)), n)[i],
			Count:    unsafe.Slice((*int64)(unsafe.Pointer(a.count)), n)[i],
			HasCount: unsafe.Slice((*uint8)(unsafe.Pointer(a.has_count)), n)[i] != 0}
	}
	for i := 0; i < n; i++ {
		out.Used[i], out.Calc[i] = rd(used, i), rd(calc, i)
		out.CalcReplaced[i] = unsafe.Slice((*uint8)(unsafe.Pointer(st.calc_at_nonzero)), n)[i] != 0
		out.ThrottledFlag[i] = unsafe.Slice((*uint32)(unsafe.Pointer(st.thrl_flag)), n)[i]
		out.ThrottledHas[i] = unsafe.Slice((*uint32)(unsafe.Pointer(st.thrl_has)), n)[i]
		out.ThrottledPod[i] = unsafe.Slice((*uint8)(unsafe.Pointer(st.thrl_pod)), n)[i] != 0
		out.Err[i] = unsafe.Slice((*uint8)(unsafe.Pointer(st.error)), n)[i] != 0
	}
	// NextOverrideHappensIn (throttle_types.go:37-63): the controller's enqueueAfter delay is instant - now
	sec, nsec, has := make([]int64, n+1), make([]int32, n+1), make([]uint8, n+1)
	if rc := C.kt_reconcile_fetch_next_override(e.h, C.int32_t(n), i64(sec), i32(nsec), u8(has)); rc != C.KT_OK {
		return nil, e.err(rc)
	}
	for i := 0; i < n; i++ {
		out.NextOverride = append(out.NextOverride, time.Unix(sec[i], int64(nsec[i])))
		out.HasNextOverride = append(out.HasNextOverride, has[i] != 0)
	}
	return out, nil
}

// ---- several GPUs / scheduler replicas: pods row-sharded, one int64 sum all-reduce per reconcile (SURVEY.md 8e)

// CommUniqueID is created on rank 0 and handed to the other ranks by the host's own channel (RPC, a ConfigMap).
func CommUniqueID() ([C.KT_COMM_ID_BYTES]byte, error) {
	var id [C.KT_COMM_ID_BYTES]byte
	if rc := C.kt_comm_unique_id(unsafe.Pointer(&id[0])); rc != C.KT_OK {
		return id, fmt.Errorf("kt_comm_unique_id: %d", int(rc))
	}
	return id, nil
}

// CommInit joins the RCCL communicator of `world` ranks.
func (e *Engine) CommInit(rank, world int, id [C.KT_COMM_ID_BYTES]byte) error {
	if rc := C.kt_comm_init(e.h, C.int32_t(rank), C.int32_t(world), unsafe.Pointer(&id[0])); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}

// ReconcileSharded is ReconcileRows(nil) across ranks: aggregate this rank's rows, sum the partials over xGMI, finalize replicated.
func (e *Engine) ReconcileSharded(now time.Time, apply bool) error {
	var flags C.uint32_t
	if apply {
		flags = C.KT_RECONCILE_APPLY
	}
	if rc := C.kt_aggregate_launch(e.h, nil); rc != C.KT_OK {
		return e.err(rc)
	}
	if rc := C.kt_comm_allreduce_partial(e.h, nil); rc != C.KT_OK {
		return e.err(rc)
	}
	if rc := C.kt_finalize_launch(e.h, C.int64_t(now.Unix()), C.int32_t(now.Nanosecond()), flags, nil); rc != C.KT_OK {
		return e.err(rc)
	}
	return nil
}
