// reference_go_bench_test.go — times the REAL reference functions (everpeace/kube-throttler, Go) on the hot path this
// repository replaces, for anyone with Go >= 1.20 and module access.
//
// UNVERIFIED: the build image of this repository has no Go toolchain (`go: command not found`) and no module cache, so
// this file has never been compiled.  It only uses exported API of pkg/apis/schedule/v1alpha1 as of the pinned commit:
//   ThrottleSelector.MatchesToPod            (throttle_selector.go:30)
//   ClusterThrottleSelector.MatchesToPod     (clusterthrottle_selector.go:44)
//   Throttle.CheckThrottledFor               (throttle_types.go:128)
//   ClusterThrottle.CheckThrottledFor        (clusterthrottle_types.go:30)
//   ResourceAmountOfPod / ResourceAmount.Add / ResourceAmount.IsThrottled   (resource_amount.go:71,91,127)
//
// Usage: copy this file to <kube-throttler checkout>/bench/ and run
//     go test -run xxx -bench PreFilter -benchtime 10x ./bench/
// It reports pod x throttle decisions per second for the shape of BASELINE.json configs[2] (scaled by KT_PODS /
// KT_THROTTLES), i.e. the loop of ThrottleController.CheckThrottled + ClusterThrottleController.CheckThrottled
// (throttle_controller.go:349-397, clusterthrottle_controller.go:378-425) without the informer caches: per pod, the
// Throttles of its namespace and every ClusterThrottle are matched and, when affected, classified.
// The synthetic objects follow the distributions of SURVEY.md 8d (not bit-identical to kt_workload.c: different PRNG
// consumption order) — the number is a throughput baseline, not a parity input.
package bench

import (
	"fmt"
	"os"
	"strconv"
	"testing"

	"github.com/everpeace/kube-throttler/pkg/apis/schedule/v1alpha1"
	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
)

type rng struct{ s uint64 }

func (r *rng) next() uint64 {
	r.s += 0x9E3779B97F4A7C15
	z := r.s
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
	z = (z ^ (z >> 27)) * 0x94D049BB133111EB
	return z ^ (z >> 31)
}
func (r *rng) below(n int) int      { return int(r.next() % uint64(n)) }
func (r *rng) chance(p float64) bool { return float64(r.next()>>11)/float64(1<<53) < p }

const (
	nNamespaces = 64
	nKeys       = 16
	nValues     = 16
	nLabels     = 8
)

var resourceNames = []corev1.ResourceName{"cpu", "memory", "ephemeral-storage", "amd.com/gpu", "hugepages-2Mi",
	"example.com/a", "example.com/b", "example.com/c"}
var presence = []float64{.9, .9, .3, .2, .15, .15, .15, .15}

func envInt(name string, def int) int {
	if v, err := strconv.Atoi(os.Getenv(name)); err == nil && v > 0 {
		return v
	}
	return def
}

func quantityFor(r *rng, d int) resource.Quantity {
	switch d {
	case 0:
		return *resource.NewMilliQuantity(int64(50*(1+r.below(80))), resource.DecimalSI)
	case 1:
		return *resource.NewQuantity(int64(64<<20)*int64(1+r.below(256)), resource.BinarySI)
	case 2:
		return *resource.NewQuantity(int64(1<<30)*int64(1+r.below(64)), resource.BinarySI)
	default:
		return *resource.NewQuantity(int64(1+r.below(8)), resource.DecimalSI)
	}
}

func makePods(n int) []*corev1.Pod {
	r := &rng{s: 0x6B7468726F74 + 2}
	pods := make([]*corev1.Pod, n)
	for i := range pods {
		labels := map[string]string{}
		for len(labels) < nLabels {
			labels[fmt.Sprintf("k%d", r.below(nKeys))] = fmt.Sprintf("v%d", r.below(nValues))
		}
		requests := corev1.ResourceList{}
		for d, name := range resourceNames {
			if r.chance(presence[d]) {
				requests[name] = quantityFor(r, d)
			}
		}
		pods[i] = &corev1.Pod{
			ObjectMeta: metav1.ObjectMeta{Namespace: fmt.Sprintf("ns%d", r.below(nNamespaces)), Name: fmt.Sprintf("pod%d", i), Labels: labels},
			Spec: corev1.PodSpec{SchedulerName: "my-scheduler",
				Containers: []corev1.Container{{Name: "c", Resources: corev1.ResourceRequirements{Requests: requests}}}},
		}
	}
	return pods
}

func makeThreshold(r *rng) v1alpha1.ResourceAmount {
	a := v1alpha1.ResourceAmount{ResourceRequests: corev1.ResourceList{}}
	if r.chance(.5) {
		a.ResourceCounts = &v1alpha1.ResourceCounts{Pod: 1 + r.below(5000)}
	}
	for d, name := range resourceNames {
		if r.chance(.6) {
			q := quantityFor(r, d)
			q.Set(q.Value() * int64(1+r.below(2000)))
			a.ResourceRequests[name] = q
		}
	}
	return a
}

func podSelector(r *rng) metav1.LabelSelector {
	ml := map[string]string{fmt.Sprintf("k%d", r.below(nKeys)): fmt.Sprintf("v%d", r.below(nValues))}
	if r.chance(.5) {
		ml[fmt.Sprintf("k%d", r.below(nKeys))] = fmt.Sprintf("v%d", r.below(nValues))
	}
	return metav1.LabelSelector{MatchLabels: ml}
}

func makeThrottles(n int) (map[string][]v1alpha1.Throttle, []v1alpha1.ClusterThrottle, map[string]*corev1.Namespace) {
	r := &rng{s: 0x6B7468726F74 + 3}
	namespaces := map[string]*corev1.Namespace{}
	for i := 0; i < nNamespaces; i++ {
		name := fmt.Sprintf("ns%d", i)
		namespaces[name] = &corev1.Namespace{ObjectMeta: metav1.ObjectMeta{Name: name,
			Labels: map[string]string{"kubernetes.io/metadata.name": name, "zone": fmt.Sprintf("z%d", i%4)}}}
	}
	byNs := map[string][]v1alpha1.Throttle{}
	var cluster []v1alpha1.ClusterThrottle
	for i := 0; i < n; i++ {
		base := v1alpha1.ThrottleSpecBase{ThrottlerName: "kube-throttler", Threshold: makeThreshold(r)}
		if i%2 == 0 {
			ns := fmt.Sprintf("ns%d", r.below(nNamespaces))
			byNs[ns] = append(byNs[ns], v1alpha1.Throttle{
				ObjectMeta: metav1.ObjectMeta{Namespace: ns, Name: fmt.Sprintf("t%d", i)},
				Spec: v1alpha1.ThrottleSpec{ThrottleSpecBase: base, Selector: v1alpha1.ThrottleSelector{
					SelecterTerms: []v1alpha1.ThrottleSelectorTerm{{PodSelector: podSelector(r)}}}}})
		} else {
			cluster = append(cluster, v1alpha1.ClusterThrottle{
				ObjectMeta: metav1.ObjectMeta{Name: fmt.Sprintf("c%d", i)},
				Spec: v1alpha1.ClusterThrottleSpec{ThrottleSpecBase: base, Selector: v1alpha1.ClusterThrottleSelector{
					SelecterTerms: []v1alpha1.ClusterThrottleSelectorTerm{{
						ThrottleSelectorTerm: v1alpha1.ThrottleSelectorTerm{PodSelector: podSelector(r)},
						NamespaceSelector:    metav1.LabelSelector{MatchLabels: map[string]string{"zone": fmt.Sprintf("z%d", r.below(4))}}}}}}})
		}
	}
	return byNs, cluster, namespaces
}

// BenchmarkPreFilter: one iteration = PreFilter's two CheckThrottled calls for every pod (isThrottledOnEqual=false,
// plugin.go:153,165) against empty reservations.
func BenchmarkPreFilter(b *testing.B) {
	nPods, nThr := envInt("KT_PODS", 20000), envInt("KT_THROTTLES", 1000)
	pods := makePods(nPods)
	byNs, cluster, namespaces := makeThrottles(nThr)
	reserved := v1alpha1.ResourceAmount{}
	blocked := 0
	b.ResetTimer()
	for it := 0; it < b.N; it++ {
		for _, pod := range pods {
			for i := range byNs[pod.Namespace] {
				thr := &byNs[pod.Namespace][i]
				if ok, err := thr.Spec.Selector.MatchesToPod(pod); err == nil && ok {
					if thr.CheckThrottledFor(pod, reserved, false) != v1alpha1.CheckThrottleStatusNotThrottled {
						blocked++
					}
				}
			}
			ns := namespaces[pod.Namespace]
			for i := range cluster {
				thr := &cluster[i]
				if ok, err := thr.Spec.Selector.MatchesToPod(pod, ns); err == nil && ok {
					if thr.CheckThrottledFor(pod, reserved, false) != v1alpha1.CheckThrottleStatusNotThrottled {
						blocked++
					}
				}
			}
		}
	}
	b.StopTimer()
	decisions := float64(b.N) * float64(nPods) * float64(nThr)
	b.ReportMetric(decisions/b.Elapsed().Seconds(), "decisions/s")
	_ = blocked
}

// BenchmarkReconcile: one iteration = the aggregation half of reconcile for every throttle (throttle_controller.go:116-133,
// clusterthrottle_controller.go:119-136): scan the pods in scope, match, fold ResourceAmountOfPod with ResourceAmount.Add,
// then IsThrottled(used, true).  Reported as pod x throttle pairs per second (P_counted x T), the unit of the engine's
// aggregation rate.
func BenchmarkReconcile(b *testing.B) {
	nPods, nThr := envInt("KT_PODS", 20000), envInt("KT_THROTTLES", 1000)
	pods := makePods(nPods)
	for i, p := range pods { // 60 % bound and running, like the synthetic snapshots
		if i%5 < 3 {
			p.Spec.NodeName = "node-1"
			p.Status.Phase = corev1.PodRunning
		}
	}
	byNs, cluster, namespaces := makeThrottles(nThr)
	podsByNs := map[string][]*corev1.Pod{}
	counted := 0
	for _, p := range pods {
		if p.Spec.NodeName != "" {
			podsByNs[p.Namespace] = append(podsByNs[p.Namespace], p)
			counted++
		}
	}
	throttled := 0
	b.ResetTimer()
	for it := 0; it < b.N; it++ {
		for ns := range byNs {
			for i := range byNs[ns] {
				thr := &byNs[ns][i]
				used := v1alpha1.ResourceAmount{}
				for _, p := range podsByNs[ns] {
					if ok, err := thr.Spec.Selector.MatchesToPod(p); err == nil && ok {
						used = used.Add(v1alpha1.ResourceAmountOfPod(p))
					}
				}
				if thr.Spec.Threshold.IsThrottled(used, true).ResourceCounts.Pod {
					throttled++
				}
			}
		}
		for i := range cluster {
			thr := &cluster[i]
			used := v1alpha1.ResourceAmount{}
			for nsName, ns := range namespaces {
				if ok, err := thr.Spec.Selector.MatchesToNamespace(ns); err != nil || !ok {
					continue
				}
				for _, p := range podsByNs[nsName] {
					if ok, err := thr.Spec.Selector.MatchesToPod(p, ns); err == nil && ok {
						used = used.Add(v1alpha1.ResourceAmountOfPod(p))
					}
				}
			}
			if thr.Spec.Threshold.IsThrottled(used, true).ResourceCounts.Pod {
				throttled++
			}
		}
	}
	b.StopTimer()
	b.ReportMetric(float64(b.N)*float64(counted)*float64(nThr)/b.Elapsed().Seconds(), "pairs/s")
	_ = throttled
}
