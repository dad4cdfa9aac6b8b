// kt_host.hpp — C++ mirror of kube-throttler's scheduler plugin on top of the engine's C-ABI.
//
// The reference's host side is Go (pkg/scheduler_plugin/plugin.go); Go is not available in this build
// image, so the layer above include/kt_engine.h is written in C++ with the SAME names, argument meaning
// and error behaviour for the hot path:
//     PluginName, NewPlugin(args), KubeThrottler::Name / PreFilter / Reserve / Unreserve
// plus the informer-side feed (OnPodAdd/..., what the event handlers of
// pkg/controllers/throttle_controller.go:400-536 push) and ReconcileAll (the aggregation part of
// [Cluster]ThrottleController.reconcile for every responsible throttle).
//
// Everything string-shaped lives here: label/namespace/resource interning, resource.Quantity and RFC3339
// parsing, LabelSelectorAsSelector validation, reason-string formatting.  The engine only sees ids and
// exact integers.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kt_engine.h"

namespace kth {

// ---- k8s.io/apimachinery pkg/api/resource.Quantity, exact (restated) ---------------
// value = nano * 1e-9 ; finer input is rounded away from zero at parse time.
// Format is the suffix family the text was written in; it only steers String() (quantity.go CanonicalizeBytes) and
// is inherited by a sum from the addend that first made it non-zero (Quantity.Add).
enum class Format : uint8_t { DecimalSI = 0, BinarySI = 1, DecimalExponent = 2 };
struct Quantity {
  __int128 nano = 0;
  Format format = Format::DecimalSI;
};
bool ParseQuantity(const std::string& text, Quantity* out, std::string* err);
// Quantity.String(): the canonical text the API server persists (status write-back).  BinarySI
// falls back to DecimalSI for |value| < 1024 and for non-integers; mantissa without trailing zeros, exponent a
// multiple of 3 (DecimalSI / DecimalExponent) or a power of 1024 (BinarySI).
std::string FormatQuantity(const Quantity& q);
// Quantity.Add's format rule: a zero receiver takes the addend's format.
inline void AddQuantity(Quantity* q, const Quantity& y) {
  if (q->nano == 0) q->format = y.format;
  q->nano += y.nano;
}
// value / 10^scale as an exact integer; false when not representable (or beyond int64).
bool ScaledValue(const Quantity& q, int scale, int64_t* out);
std::string FormatDecimalSI(const Quantity& q);

// k8s.io/apimachinery validation.IsQualifiedName / IsValidLabelValue (restated): what makes
// LabelSelectorAsSelector fail on a key or a value.
bool ValidLabelKey(const std::string& key);
bool ValidLabelValue(const std::string& value);

// Go time.Parse(time.RFC3339, text) -> (unix seconds, nanoseconds).
bool ParseRFC3339(const std::string& text, int64_t* sec, int32_t* nsec, std::string* err);

// ---- object model (core/v1 Pod / Namespace, schedule/v1alpha1 Throttle / ClusterThrottle) ---------------
using Labels = std::map<std::string, std::string>;
using ResourceList = std::map<std::string, std::string>;  // resource name -> Quantity text

struct Container {
  ResourceList requests;
};
struct Pod {
  std::string ns, name;
  Labels labels;
  std::string schedulerName, nodeName, phase;
  std::vector<Container> containers, initContainers;
  bool hasOverhead = false;
  ResourceList overhead;
  std::string Key() const { return ns + "/" + name; }
};
struct Namespace {
  std::string name;
  Labels labels;
};
struct LabelSelectorRequirement {
  std::string key, op;  // In | NotIn | Exists | DoesNotExist
  std::vector<std::string> values;
};
struct LabelSelector {
  Labels matchLabels;
  std::vector<LabelSelectorRequirement> matchExpressions;
};
struct SelectorTerm {
  LabelSelector podSelector;
  LabelSelector namespaceSelector;  // ClusterThrottle only
};
struct ResourceAmount {
  bool hasCounts = false;  // resourceCounts != nil
  int64_t pod = 0;
  ResourceList requests;
};
struct TemporaryThresholdOverride {
  std::string begin, end;
  ResourceAmount threshold;
};
struct Throttle {
  bool cluster = false;  // kind ClusterThrottle
  std::string ns, name, throttlerName;
  ResourceAmount threshold;
  std::vector<TemporaryThresholdOverride> overrides;
  std::vector<SelectorTerm> selectorTerms;
  // types.NamespacedName.String(): a ClusterThrottle renders as "/name" (plugin.go:289-295)
  std::string Key() const { return (cluster ? std::string() : ns) + "/" + name; }
};

// status written back by ReconcileAll (what UpdateStatus would persist)
struct ThrottleStatus {
  bool usedHasCounts = false;
  int64_t usedPod = 0;
  std::map<std::string, Quantity> used;
  std::map<std::string, bool> throttledRequests;
  bool throttledPod = false;
  bool calculatedThresholdUpdated = false;
  std::vector<std::string> messages;
  bool error = false;
  // ThrottleSpecBase.NextOverrideHappensIn as an instant: when the controller must reconcile this throttle again
  bool hasNextOverride = false;
  int64_t nextOverrideSec = 0;
  int32_t nextOverrideNsec = 0;
  // !apiequality.Semantic.DeepEqual(thr.Status, *newStatus) (throttle_controller.go:157): UpdateStatus is only called
  // when this is set; otherwise the reference logs "No need to update status"
  bool needsUpdate = false;
  // status.used.resourceRequests as the API server would persist it (Quantity.String())
  std::map<std::string, std::string> UsedStrings() const;
};
// apiequality.Semantic.DeepEqual on the parts of ThrottleStatus a reconcile can change besides calculatedThreshold
// (used, throttled): quantities compare by value (Cmp == 0), nil and empty maps are equal, resourceCounts nil != &{0}.
bool StatusSemanticEqual(const ThrottleStatus& a, const ThrottleStatus& b);

// framework.Code values used by the plugin
enum Code { Success = 0, Error = 1, UnschedulableAndUnresolvable = 3 };
// what PreFilter hands to fh.EventRecorder().Eventf (plugin.go:190-202)
struct Event {
  std::string type, reason, message;
};
struct Status {
  Code code = Success;
  std::vector<std::string> reasons;
  std::vector<Event> events;
  bool IsSuccess() const { return code == Success; }
};

// KubeThrottlerPluginArgs (pkg/scheduler_plugin/plugin_args.go:33-40) + engine sizing
struct PluginArgs {
  std::string name;                 // throttler name (required)
  std::string targetSchedulerName;  // required
  int64_t podCapacity = 1 << 16;
  int32_t throttleCapacity = 1024;
  int32_t namespaceCapacity = 256;
  int32_t maxLabels = KT_MAX_LABELS;
  // resource name -> decimal scale of its fixed-point dimension (e.g. {"cpu",-3}); names not listed are
  // assigned on first sight at scale 0 ("cpu" at -3)
  std::map<std::string, int> resourceScales;
};

extern const char* const PluginName;  // "kube-throttler" (plugin.go:45)

class KubeThrottler {
 public:
  ~KubeThrottler();
  const char* Name() const { return PluginName; }

  // ---- informer feed
  bool OnNamespaceAdd(const Namespace& ns, std::string* err);
  bool OnNamespaceDelete(const std::string& name, std::string* err);
  bool OnPodAdd(const Pod& pod, std::string* err);  // Add; also what the engine has to see of an Update
  // Update handler (throttle_controller.go:451-507, clusterthrottle_controller.go:483-539): when the pod counts in
  // (before or after) and its set of affected throttles changes, its reservation moves with it — removed from the
  // throttles it left, ADDED to the ones it joined (reserved_resource_amounts.go:92-111), until their next reconcile
  // counts it in `used` and un-reserves it.  Then the new object is fed to the engine like OnPodAdd.
  bool OnPodUpdate(const Pod& old_pod, const Pod& new_pod, std::string* err);
  // Delete handler: a scheduled pod that counts in is un-reserved first (throttle_controller.go:508-517)
  bool OnPodDelete(const std::string& key, std::string* err);
  bool OnThrottleAdd(const Throttle& thr, std::string* err);  // also Update; the stored status restarts empty until ReconcileAll
  bool OnThrottleDelete(const std::string& key, bool cluster, std::string* err);

  // ---- plugin.go:148-215
  Status PreFilter(const Pod& pod);
  // ---- plugin.go:217-257 (reservation bookkeeping stays host-side; totals go to the engine)
  Status Reserve(const Pod& pod);
  void Unreserve(const Pod& pod);

  // ---- one scheduling pass over a queue of pending pods (by Key(), fed through OnPodAdd) in order: PreFilter and,
  // on Success, Reserve — ONE engine launch (kt_admit_launch) instead of 2 x n calls; same reserved bookkeeping
  std::vector<Status> AdmitQueue(const std::vector<std::string>& pod_keys);

  // ---- reconcile of every responsible throttle at `now` (RFC3339); fills per-throttle status by Key()
  bool ReconcileAll(const std::string& now_rfc3339, std::map<std::string, ThrottleStatus>* out, std::string* err);

  // CheckThrottleStatus of (pod, throttle key) from the last PreFilter of that pod ("" = not affected)
  std::string LastStatusOf(const std::string& throttle_key) const;

  struct Impl;  // engine handle, dictionaries, reserved cache

 private:
  friend std::unique_ptr<KubeThrottler> NewPlugin(const PluginArgs&, std::string*);
  KubeThrottler() = default;
  std::unique_ptr<Impl> p_;
};

// NewPlugin (plugin.go:63-146): validates the args like DecodePluginArgs (plugin_args.go:42-60) and creates
// the engine.  Returns nullptr + err on failure (no GPU => KT_ERR_NO_DEVICE text).
std::unique_ptr<KubeThrottler> NewPlugin(const PluginArgs& args, std::string* err);

}  // namespace kth
